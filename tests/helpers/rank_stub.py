"""Stand-in body for the launcher test (tests/test_launch_cpu.py): what a rank of bench.py does around
its device work - join the group torch.distributed.run described in the environment, take part in one
collective per kind the ensemble path uses, rank 0 reports - with the gloo backend and no GPU."""
import json
import os
import sys

import torch
import torch.distributed as dist


def main():
    out_path = sys.argv[1]
    fail_rank = int(sys.argv[2]) if len(sys.argv) > 2 else -1
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["MASTER_ADDR"] == "127.0.0.1"
    if rank == fail_rank:
        raise SystemExit(3)
    dist.init_process_group("gloo")
    try:
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        parts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(parts, t)                       # bench.py: every rank's kernel time
        dist.all_reduce(t, op=dist.ReduceOp.MAX)        # bench.py: max-over-ranks wall time
        if rank == 0:
            with open(out_path, "w") as f:
                json.dump({"rccl_ranks": dist.get_world_size(), "max": float(t.item()),
                           "per_rank": [float(p.item()) for p in parts],
                           "local_rank": int(os.environ["LOCAL_RANK"])}, f)
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
