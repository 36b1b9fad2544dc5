"""Start one process per GPU for an ensemble job (the path shards along the
ensemble dimension only, one rank per GPU, RCCL for the code-object broadcast and the result gather).

The reference has no counterpart (a sweep there is a Python loop over ``rmtExe``); this is the
build's own entry into its multi-GPU path: ``python bench.py --gpus N`` - or any script that calls
``spawn_ranks`` - becomes N ranks of a ``torch.distributed`` job on this node.

Rules of the GPU boxes this runs on (see the task's environment notes):
  * the PARENT never touches the GPU (``torch.cuda.device_count()`` only counts devices); the ranks
    are started as child processes of ``python -m torch.distributed.run`` and the parent exits with
    their status - nothing is re-exec'ed after a process has initialised HIP;
  * the rendezvous is pinned to 127.0.0.1 (container host names may not resolve).
"""
import os
import signal
import socket
import subprocess
import sys


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_gpus():
    """Number of HIP devices this process could use - WITHOUT initialising the runtime."""
    try:
        import torch
        return int(torch.cuda.device_count())
    except Exception:
        return 0


def is_rank():
    """True inside a process started by torch.distributed.run / spawn_ranks."""
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def rank_command(nproc, script_argv, port=None, python=None):
    """The command line spawn_ranks runs - exactly the driver's own launch line for N > 1."""
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
            "--nproc-per-node", str(int(nproc)), "--master-addr", "127.0.0.1",
            "--master-port", str(int(port or free_port()))] + list(script_argv)


def spawn_ranks(nproc, script_argv, require_gpus=True, env=None, timeout=None):
    """Run ``script_argv`` (script path + its arguments) as ``nproc`` ranks on this node and return
    the job's exit status.  ``require_gpus``: refuse - loudly, before anything is started - when fewer
    than ``nproc`` HIP devices are visible (the ensemble path has no CPU fallback; the CPU tests pass
    False and use the gloo backend with a stand-in body)."""
    nproc = int(nproc)
    if nproc < 1:
        raise ValueError("need at least one rank (got %d)" % nproc)
    if is_rank():
        raise RuntimeError("spawn_ranks called from inside a rank (RANK=%s): ranks do not nest"
                           % os.environ.get("RANK"))
    if require_gpus:
        have = visible_gpus()
        if have < nproc:
            raise SystemExit("%d ranks need %d visible MI355X GPUs, this machine shows %d: the N2 integrator has "
                             "no CPU fallback and one rank per GPU is the only supported layout "
                             "(HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES restrict the count)" % (nproc, nproc, have))
    child_env = dict(os.environ)
    child_env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: required by RCCL on these hosts
    child_env.setdefault("OMP_NUM_THREADS", "1")
    child_env.update(env or {})
    # own session = own process group: a timeout ends exactly the launcher and the ranks it started
    proc = subprocess.Popen(rank_command(nproc, script_argv), env=child_env, start_new_session=True)
    try:
        return proc.wait(timeout=timeout)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)
        proc.wait()
        raise
