"""Run under LD_PRELOAD=<asan runtime> with RMT_N2_LIBRARY=<librmt_n2_asan.so>: drives the C-ABI host
layer through its compile and error paths (no GPU needed) so AddressSanitizer sees every allocation,
string copy and cleanup of csrc/rmt_n2.cpp.  Prints ASAN_CABI_OK on success."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np

import inputs as INP
from rmt_app_amd import hipbind, plan

assert hipbind.LIB_PATH.endswith("librmt_n2_asan.so"), hipbind.LIB_PATH
L = hipbind.lib()
tpl = hipbind.kernel_template()
assert "rmt_n2_rk4_reg" in tpl

mech = plan.Mechanism(INP.ch4_input())
src = mech.source(tpl, False, 64, 1)
blob, log = hipbind.compile_source(src)                      # success path: code + log buffers, rmt_n2_free
assert blob[:4] == b"\x7fELF"
blob2, _ = hipbind.compile_source(src, "gfx950", "-DRMT_EXP_BITS=6 -ffast-math")   # option splitting
assert blob2[:4] == b"\x7fELF"
for bad in ("this is not HIP;", "#error stop\n", ""):        # compiler-error path: log copied into the message
    try:
        hipbind.compile_source(bad) if bad else hipbind.compile_source("int x = ;")
        raise SystemExit("compile of %r did not fail" % bad)
    except hipbind.RmtN2Error as e:
        assert "hiprtc" in str(e)
code, size, logp = C.c_void_p(), C.c_size_t(), C.c_void_p()
assert L.rmt_n2_compile(None, b"gfx950", b"", C.byref(code), C.byref(size), C.byref(logp)) != 0
assert b"null argument" in L.rmt_n2_last_error()

h = C.c_void_p()
assert L.rmt_n2_create(None, C.byref(h)) != 0
p = hipbind.Plan()
assert L.rmt_n2_create(C.byref(p), C.byref(h)) != 0 and b"ABI version" in L.rmt_n2_last_error()
p.abi_version = hipbind.ABI_VERSION
assert L.rmt_n2_create(C.byref(p), C.byref(h)) != 0 and b"bad plan sizes" in L.rmt_n2_last_error()
nm, row = plan.member_constants(INP.ch4_input(), mech, 20)
row = np.ascontiguousarray(row)
buf = C.create_string_buffer(blob, len(blob))
p.n_species, p.n_reactions, p.n_vars, p.n_nodes, p.n_members = mech.S, mech.R, mech.V + 3, 20, 1
assert L.rmt_n2_create(C.byref(p), C.byref(h)) != 0 and b"n_vars" in L.rmt_n2_last_error()
p.n_vars, p.block = mech.V, 100
assert L.rmt_n2_create(C.byref(p), C.byref(h)) != 0 and b"block" in L.rmt_n2_last_error()
p.block, p.nodes_per_thread = 64, 0
assert L.rmt_n2_create(C.byref(p), C.byref(h)) != 0 and b"nodes_per_thread" in L.rmt_n2_last_error()
p.nodes_per_thread = 1
assert L.rmt_n2_create(C.byref(p), C.byref(h)) != 0 and b"lacks code" in L.rmt_n2_last_error()
p.code_object, p.code_size = C.cast(buf, C.c_void_p), len(blob)
p.members = row.ctypes.data_as(C.POINTER(C.c_double))
rc = L.rmt_n2_create(C.byref(p), C.byref(h))
# The child is CPU-only by construction (the parent hides every device: sanitizers never run against the GPU on
# this pool), so a complete plan must end in the "no HIP device" error - with the handle left null
assert os.environ.get("HIP_VISIBLE_DEVICES") == "" and os.environ.get("ROCR_VISIBLE_DEVICES") == ""
assert rc != 0 and b"no HIP device" in L.rmt_n2_last_error() and not h.value
p.n_user_params = 65
assert L.rmt_n2_create(C.byref(p), C.byref(h)) != 0 and b"n_user_params" in L.rmt_n2_last_error()
# null-handle guards of every entry point
null = C.c_void_p()
st = hipbind.Stats()
assert L.rmt_n2_rhs(null, 0.0, null, null) != 0
assert L.rmt_n2_rk4(null, null, 0.0, 1e-6, 1) != 0
assert L.rmt_n2_multistep(null, null, 0.0, 1e-6, 3, 0) != 0
assert L.rmt_n2_rk45(null, null, 0.0, 1.0, 1e-6, 1e-9, 1e-6, 10, null) != 0
assert L.rmt_n2_ros4(null, null, 0.0, 1.0, 1e-6, 1e-9, 1e-6, 10, null) != 0
assert L.rmt_n1_profile(null, None, null, 2, 1e-6, 1e-9, 1e-6, 10, null) != 0
assert L.rmt_n2_status(null, None) != 0
assert L.rmt_n2_set_members(null, None) != 0
assert L.rmt_n2_set_stream(null, null) != 0
assert L.rmt_n2_set_mode(null, 0) != 0
ms = C.c_float()
assert L.rmt_n2_last_kernel_ms(null, C.byref(ms)) != 0
ci = C.c_int()
assert L.rmt_n2_last_geometry(null, C.byref(ci), C.byref(ci)) != 0
L.rmt_n2_destroy(null)
print("ASAN_CABI_OK")
