"""The C ABI with no torch anywhere: raw hipMalloc'ed pointers through ctypes, exactly what a
binding written inside the reference would do."""
import ctypes as C
import os

import numpy as np
import pytest

import inputs as INP
from oracle import n2_oracle as O
from rmt_app_amd import hipbind, plan

pytestmark = pytest.mark.gpu


def test_plain_ctypes_roundtrip():
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    hip.hipDeviceSynchronize.argtypes = []
    H2D, D2H = 1, 2
    L = hipbind.lib()

    mi = INP.dme_notebook_input()
    N = 200
    mech = plan.Mechanism(mi)
    nm, row = plan.member_constants(mi, mech, N)
    block, npt = 256, 1
    src = mech.source(hipbind.kernel_template(), False, block, npt)
    code, size, log = C.c_void_p(), C.c_size_t(), C.c_void_p()
    assert L.rmt_n2_compile(src.encode(), b"gfx950", b"", C.byref(code), C.byref(size), C.byref(log)) == 0

    p = hipbind.Plan()
    p.abi_version, p.n_species, p.n_reactions, p.n_vars = hipbind.ABI_VERSION, mech.S, mech.R, mech.V
    p.n_nodes, p.n_members, p.fp32, p.block, p.nodes_per_thread = N, 1, 0, block, npt
    p.code_object, p.code_size = code, size
    rows = np.ascontiguousarray(row.reshape(1, -1))
    p.members = rows.ctypes.data_as(C.POINTER(C.c_double))
    h = C.c_void_p()
    assert L.rmt_n2_create(C.byref(p), C.byref(h)) == 0, L.rmt_n2_last_error()

    y0 = np.ascontiguousarray(plan.initial_state(nm, mech, N))
    nbytes = y0.nbytes
    d_y, d_f = C.c_void_p(), C.c_void_p()
    assert hip.hipMalloc(C.byref(d_y), nbytes) == 0 and hip.hipMalloc(C.byref(d_f), nbytes) == 0
    assert hip.hipMemcpy(d_y, y0.ctypes.data, nbytes, H2D) == 0
    assert L.rmt_n2_rhs(h, 0.0, d_y, d_f) == 0
    assert L.rmt_n2_rk4(h, d_y, 0.0, 2e-6, 30) == 0
    flags = (C.c_uint32*1)()
    assert L.rmt_n2_status(h, flags) == 0 and flags[0] == 0
    f = np.empty_like(y0)
    y = np.empty_like(y0)
    assert hip.hipMemcpy(f.ctypes.data, d_f, nbytes, D2H) == 0
    assert hip.hipMemcpy(y.ctypes.data, d_y, nbytes, D2H) == 0
    ms = C.c_float()
    assert L.rmt_n2_last_kernel_ms(h, C.byref(ms)) == 0 and ms.value > 0

    pr = O.setup_n2(mi, N)
    fo = O.make_rhs_vec(pr)
    want_f = fo(0.0, pr["IV"])
    want_y = O.rk4(0.0, 30*2e-6, 30, pr["IV"], fo, keep=False)
    sc = np.max(np.abs(want_f.reshape(7, N)), axis=1, keepdims=True)
    assert np.max(np.abs(f.reshape(7, N) - want_f.reshape(7, N))/sc) < 1e-12
    sc = np.max(np.abs(want_y.reshape(7, N)), axis=1, keepdims=True)
    assert np.max(np.abs(y.reshape(7, N) - want_y.reshape(7, N))/sc) < 1e-12

    # error paths: N too large for the register stepper when forced, bad dt
    assert L.rmt_n2_rk4(h, d_y, 0.0, -1.0, 1) != 0 and b"dt > 0" in L.rmt_n2_last_error()
    L.rmt_n2_destroy(h)
    L.rmt_n2_free(code)
    hip.hipFree(d_y), hip.hipFree(d_f)


def test_plain_ctypes_stiff_stepper_and_stats():
    """rmt_n2_ros4 + rmt_n2_stats through raw device pointers: the reference test case to t = 0.1 s
    against the tight reference run (golden G4), no torch involved"""
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    L = hipbind.lib()
    mi = INP.dme_script_input()
    N = 20
    mech = plan.Mechanism(mi)
    nm, row = plan.member_constants(mi, mech, N)
    src = mech.source(hipbind.kernel_template(), False, 64, 1, None, {"RMT_WITH_ROS4": "1"})
    code, size, log = C.c_void_p(), C.c_size_t(), C.c_void_p()
    assert L.rmt_n2_compile(src.encode(), b"gfx950", b"", C.byref(code), C.byref(size), C.byref(log)) == 0
    p = hipbind.Plan()
    p.abi_version, p.n_species, p.n_reactions, p.n_vars = hipbind.ABI_VERSION, mech.S, mech.R, mech.V
    p.n_nodes, p.n_members, p.fp32, p.block, p.nodes_per_thread = N, 1, 0, 64, 1
    p.code_object, p.code_size = code, size
    rows = np.ascontiguousarray(row.reshape(1, -1))
    p.members = rows.ctypes.data_as(C.POINTER(C.c_double))
    h = C.c_void_p()
    assert L.rmt_n2_create(C.byref(p), C.byref(h)) == 0, L.rmt_n2_last_error()
    y0 = np.ascontiguousarray(plan.initial_state(nm, mech, N))
    d_y, d_st = C.c_void_p(), C.c_void_p()
    assert hip.hipMalloc(C.byref(d_y), y0.nbytes) == 0 and hip.hipMalloc(C.byref(d_st), C.sizeof(hipbind.Stats)) == 0
    assert hip.hipMemcpy(d_y, y0.ctypes.data, y0.nbytes, 1) == 0
    assert L.rmt_n2_ros4(h, d_y, 0.0, 0.1, 1e-6, 1e-9, 1e-5, 10**6, d_st) == 0, L.rmt_n2_last_error()
    flags = (C.c_uint32*1)()
    assert L.rmt_n2_status(h, flags) == 0 and flags[0] == 0          # synchronises
    st = hipbind.Stats()
    y = np.empty_like(y0)
    assert hip.hipMemcpy(C.byref(st), d_st, C.sizeof(hipbind.Stats), 2) == 0
    assert hip.hipMemcpy(y.ctypes.data, d_y, y0.nbytes, 2) == 0
    assert st.t_end == 0.1 and 20 < st.accepted < 400 and st.rejected < 40
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g4_tight_dme_script_lsoda.npz"))
    ref = np.concatenate([g["dataYCons1_0"].reshape(6, N), g["dataYTemp1_0"].reshape(1, N)])
    assert np.max(np.abs(y.reshape(7, N)[:, -1] - ref[:, -1])/np.abs(ref[:, -1])) < 1e-6
    L.rmt_n2_destroy(h)
    L.rmt_n2_free(code)
    hip.hipFree(d_y), hip.hipFree(d_st)
