#!/usr/bin/env python3
"""One shape of tools/rhs_stream_bench.py for rocprofv3: 16384 reactors x 1024 nodes (1.9 GB in + out) through rmt_n2_rhs,
12 launches (the bench.py `rhs_stream` measurement)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench as B                          # noqa: E402
from rmt_app_amd import plan               # noqa: E402

inputs = B.sweep_member_inputs(0, 1)
mech = plan.Mechanism(inputs[0])
print(B.rhs_stream(mech, inputs))
