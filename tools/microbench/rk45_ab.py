import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench as B, numpy as np
from rmt_app_amd import plan
inputs = B.sweep_member_inputs(0, 256, total=2048)
mech = plan.Mechanism(inputs[0])
pairs = [plan.member_constants(mi, mech, 1024) for mi in inputs]
rows = np.array([r for _, r in pairs]); IV = np.array([plan.initial_state(nm, mech, 1024) for nm, _ in pairs])
for _ in range(3):
    r = B.adaptive_rk45(mech, rows, IV, 1024)
    print({k: round(v["accepted_node_steps_per_s"]/1e9, 3) for k, v in r.items()}, flush=True)
