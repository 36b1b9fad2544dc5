"""CPU-only tests of the product's host side: lowering, plan constants vs the reference goldens,
C-ABI library exports, hipRTC cross-compilation of the generated kernels, and the host
emulation of the generated source against the reference RHS goldens.  No GPU compute."""
import ctypes
import json
import math
import os
import re

import numpy as np
import pytest

import inputs as INP
from oracle import n2_oracle as O
from oracle.hostemu import HostEmu
from rmt_app_amd import compdb, hipbind, lowering, plan
from rmt_app_amd.n2 import choose_geometry, pack_interval

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def relerr(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.max(np.abs(a - b)/np.maximum(np.abs(b), 1e-300))


def rowwise_err(a, b, V):
    a = np.asarray(a, float).reshape(V, -1)
    b = np.asarray(b, float).reshape(V, -1)
    den = np.max(np.abs(b), axis=1)
    den[den == 0] = 1.0
    return np.max(np.max(np.abs(a - b), axis=1)/den)


@pytest.fixture(scope="module")
def template():
    return hipbind.kernel_template()


# ----------------------------------------------------------------------------- C ABI
def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "rmt_n2.h")).read()
    declared = set(re.findall(r"\b(rmt_n[12]_[a-z0-9_]+)\s*\(", hdr))
    assert {"rmt_n1_profile", "rmt_n2_ros4", "rmt_n2_multistep", "rmt_n2_set_members"} <= declared
    assert {"rmt_n2_create", "rmt_n2_rhs", "rmt_n2_rk4", "rmt_n2_rk45", "rmt_n2_status",
            "rmt_n2_destroy", "rmt_n2_last_error", "rmt_n2_compile"} <= declared
    L = ctypes.CDLL(hipbind.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    assert hipbind.lib().rmt_n2_abi_version() == hipbind.ABI_VERSION == 2
    # the embedded device template = the per-family files of csrc/kernels/ in the order of kernels/ORDER
    tpl = hipbind.kernel_template()
    kdir = os.path.join(ROOT, "rmt_app_amd", "csrc", "kernels")
    names = [ln.split("#", 1)[0].strip() for ln in open(os.path.join(kdir, "ORDER"))]
    names = [n for n in names if n]
    assert sorted(names) == sorted(f for f in os.listdir(kdir) if f.endswith(".inc"))
    assert tpl == "".join(open(os.path.join(kdir, n)).read() for n in names)


def test_create_rejects_bad_plans_and_reports_errors():
    L = hipbind.lib()
    p = hipbind.Plan()
    h = ctypes.c_void_p()
    assert L.rmt_n2_create(ctypes.byref(p), ctypes.byref(h)) != 0
    assert b"ABI version" in L.rmt_n2_last_error()
    p.abi_version = hipbind.ABI_VERSION
    assert L.rmt_n2_create(ctypes.byref(p), ctypes.byref(h)) != 0
    assert b"bad plan sizes" in L.rmt_n2_last_error()
    with pytest.raises(hipbind.RmtN2Error):
        hipbind.compile_source("this is not HIP")


def test_cabi_error_paths_under_asan():
    """SURVEY section 5 (sanitizer row): the C-ABI host layer built with -fsanitize=address
    (`make -C rmt_app_amd/csrc asan`; CPU build - GPU ASan is not available on this pool) is driven
    through its hipRTC compile, error and cleanup paths in a child process that preloads the runtime."""
    import subprocess
    import sys
    csrc = os.path.join(ROOT, "rmt_app_amd", "csrc")
    r = subprocess.run(["make", "-C", csrc, "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    rt = subprocess.run(["make", "-s", "-C", csrc, "asan-rt"], capture_output=True, text=True).stdout.strip()
    assert os.path.exists(rt), rt
    # the child never sees a GPU, whatever box this runs on: sanitizer runs stay on the CPU build
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",
               RMT_N2_LIBRARY=os.path.join(ROOT, "rmt_app_amd", "librmt_n2_asan.so"),
               HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "asan_cabi_paths.py")],
                       capture_output=True, text=True, env=env, timeout=600)
    assert "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0 and "ASAN_CABI_OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from rmt_app_amd import rmtExe
    with pytest.raises(hipbind.RmtN2Error):
        rmtExe(INP.dme_notebook_input(ivp="hip-rk4"))


@pytest.mark.parametrize("name", list(INP.ALL_N2_INPUTS))
def test_generated_kernels_cross_compile_for_gfx950(name, template):
    mech = plan.Mechanism(INP.ALL_N2_INPUTS[name]())
    block, npt = choose_geometry(1024, mech.V)
    blob = hipbind.compile_cached(mech.source(template, False, block, npt),
                                  mech.digest(template, False, block, npt), "gfx950")
    assert blob[:4] == b"\x7fELF" and len(blob) > 4096
    for sym in (b"rmt_n2_rhs", b"rmt_n2_rk4_reg", b"rmt_n2_rk4_mem"):
        assert sym in blob


# ----------------------------------------------------------------------------- lowering
@pytest.mark.parametrize("name", list(INP.ALL_N2_INPUTS))
def test_trace_is_bit_identical_to_direct_lambda_evaluation(name):
    mi = INP.ALL_N2_INPUTS[name]()
    S = len(mi['feed']['components']['shell'])
    rr = mi['reaction-rates']
    low = lowering.trace(rr['VARS'], rr['RATES'], S)
    rng = np.random.default_rng(3)
    for _ in range(25):
        T, P = rng.uniform(450, 1000), rng.uniform(1e5, 6e6)
        x = rng.random(S) + 0.01
        x /= x.sum()
        C = x*P/(8.314472*T)
        want = O.reaction_rate_exe((T, P, x, C), rr['VARS'], rr['RATES'])
        assert low.evaluate(T, P, x, C) == [float(w) for w in want]


def test_dme_dag_statistics_and_source_shape():
    rr = INP.dme_notebook_input()['reaction-rates']
    low = lowering.trace(rr['VARS'], rr['RATES'], 6)
    st = low.stats()
    assert st["exp"] == 8 and st["log"] == 1 and st["log10"] == 1 and st["sqrt"] == 1 and st["exp10"] == 1
    src = low.emit()
    assert "rmt_kinetics" in src and src.count("rmt_exp(") == 8 and "r[2] =" in src
    assert not low.uses("C0") and low.uses("x0") and low.uses("T") and low.uses("P")


def test_closures_and_direct_imports_are_rebound():
    from math import exp as my_exp, log

    def helper(k0, Ea, x):
        return k0*my_exp(-Ea/(x['R_CONST']*x['T']))

    VARS = {"A": 2.5, "k": lambda x: helper(3.0, 1.0e4, x), "lnT": lambda x: log(x['T'], 10)}
    RATES = {"r1": lambda x: x['k']*x['SpCoi'][1]**1.5*np.sqrt(x['MoFri'][0]) + x['A']*x['lnT']}
    low = lowering.trace(VARS, RATES, 2)
    T, P, x, C = 600.0, 2e5, np.array([0.3, 0.7]), np.array([10.0, 30.0])
    want = O.reaction_rate_exe((T, P, x, C), VARS, RATES)[0]
    assert abs(low.evaluate(T, P, x, C)[0] - want) <= 1e-15*abs(want)
    assert "rmt_pow(" in low.emit()


def test_untraceable_constructs_raise_not_fallback():
    with pytest.raises(lowering.LoweringError):
        lowering.trace({}, {"r": lambda x: 1.0 if x['T'] > 500 else 2.0}, 2)
    with pytest.raises(lowering.LoweringError):
        lowering.trace({}, {"r": lambda x: math.gamma(x['T'])}, 2)
    with pytest.raises(lowering.LoweringError):
        lowering.trace({}, {"r": lambda x: x['MoFri'][x['T']]}, 2)


def test_domain_checks_are_emitted():
    low = lowering.trace({}, {"r": lambda x: math.log(x['P'] - 1e5)/x['MoFri'][0] + math.exp(x['T'])}, 1)
    src = low.emit()
    assert "RMT_CHECK_POS(flag," in src and "RMT_CHECK_DEN(flag, x[0])" in src and "RMT_CHECK_EXP(flag, T)" in src


# ----------------------------------------------------------------------------- plan vs reference
@pytest.mark.parametrize("name", list(INP.ALL_N2_INPUTS))
def test_plan_constants_vs_reference_setup(name):
    with open(os.path.join(G, "g1_setup.json")) as f:
        g = json.load(f)[name]
    mi = INP.ALL_N2_INPUTS[name]()
    mech = plan.Mechanism(mi)
    nm, row = plan.member_constants(mi, mech, 20)
    c, bc, dap = g["const"], g["constBC1"], g["DimensionlessAnalysisParams"]
    assert mech.MW == c["MoWei"] and mech.V == c["varNo"] and mech.R == c["reactionListNo"]
    assert np.max(np.abs(mech.StHeRe25 - np.array(c["StHeRe25"]))) < 1e-9
    assert mech.reactionListSorted == g["reactionListSorted"]
    assert mech.reactionStochCoeff == g["reactionStochCoeff"]
    for k, ref in (("CrSeAr", c["CrSeAr"]), ("GaMiVi", c["GaMiVi"]), ("dz", c["dz"]),
                   ("SuGaVe0", bc["SuGaVe0"]), ("GaDe0", bc["GaDe0"]), ("GaCpMeanMix0", bc["GaCpMeanMix0"]),
                   ("SpCo0", bc["SpCo0"]), ("vf", dap["vf"]), ("Cpf", dap["Cpf"]), ("Cpif", dap["Cpif"]),
                   ("GaMaCoTe0", dap["GaMaCoTe0"]), ("GaHeCoTe0", dap["GaHeCoTe0"]),
                   ("EfHeTrAr", g["ExHe"]["EfHeTrAr"])):
        assert relerr(nm[k], ref) < 1e-14, k
    np.testing.assert_array_equal(plan.initial_state(nm, mech, 20), g["IV"])
    F = plan.MEMBER_FIELDS
    assert row[F["INV_DZ"]] == 19 and row[F["TM"]] == g["ExHe"]["MeTe"]
    assert relerr(row[F["F1"]], dap["vf"]/(mi["reactor"]["BeVoFr"]*dap["zf"])) < 1e-15
    np.testing.assert_allclose(row[F["CIN"]:], np.array(g["IV"]).reshape(c["varNo"], 20)[:mech.S, 0], rtol=1e-16)


def test_component_table_vs_reference_probes():
    with open(os.path.join(G, "g7_helpers.json")) as f:
        g = json.load(f)
    assert list(compdb.componentSymbolList) == g["components"]
    assert [compdb.COMPONENTS[s].MW for s in g["components"]] == g["MW"]
    assert [compdb.COMPONENTS[s].dHf25 for s in g["components"]] == g["dHf25"]
    mf, MW = np.array(g["molefrac"]), np.array(g["MW"])
    for p in g["probes"]:
        T = p["T"]
        assert relerr([compdb.cp_value(s, T) for s in g["components"]], p["Cp"]) < 1e-15
        vis = np.array([compdb.viscosity(s, T) for s in g["components"]])
        assert relerr(vis, p["GaVii"]) < 1e-15
        assert relerr(plan.wilke(vis, mf, MW), p["GaMiVi"]) < 1e-14


def test_unknown_component_and_bad_inputs():
    mi = INP.dme_notebook_input()
    mi["feed"]["components"]["shell"] = ["H2", "Xe"]
    with pytest.raises(Exception, match="Component database is not up to date!"):
        plan.Mechanism(mi)
    mi = INP.dme_notebook_input()
    mi["feed"]["concentration"] = [1.0, 2.0]
    with pytest.raises(ValueError):
        plan.member_constants(mi, plan.Mechanism(INP.dme_notebook_input()), 20)
    from rmt_app_amd import rmtExe
    with pytest.raises(NotImplementedError):
        rmtExe(dict(INP.dme_notebook_input(), model="M9"))


# ----------------------------------------------------------------------------- generated source on CPU
RHS_CASES = [("dme_nb", 20), ("dme_nb", 100), ("dme_nb", 1024), ("dme_script", 20),
             ("dme_script", 100), ("ch4", 20), ("ch4", 100), ("syn12", 20), ("syn12", 100)]


@pytest.mark.parametrize("name,zNo", RHS_CASES)
def test_generated_node_physics_vs_reference_rhs(name, zNo, template):
    """The same generated translation unit the GPU compiles, built for the host (test-only
    emulation, oracle/hostemu_driver.cpp) reproduces the reference RHS goldens."""
    g = np.load(os.path.join(G, "g2_rhs.npz"))
    Y, F = g["%s_%d_y" % (name, zNo)], g["%s_%d_f" % (name, zNo)]
    mi = INP.ALL_N2_INPUTS[name]()
    mech = plan.Mechanism(mi)
    _, row = plan.member_constants(mi, mech, zNo)
    emu = HostEmu(mech.source(template), tag=name)
    out, flags = emu.rhs(Y, np.tile(row, (len(Y), 1)), zNo)
    assert not flags.any()
    for k in range(len(Y)):
        assert rowwise_err(out[k], F[k], mech.V) < 1e-12, k


def test_generated_isothermal_variant(template):
    g = np.load(os.path.join(G, "g2_rhs.npz"))
    mi = INP.dme_notebook_input(process_type="iso-thermal")
    mech = plan.Mechanism(mi)
    assert mech.V == 6 and "#define RMT_ISO 1" in mech.prelude()
    _, row = plan.member_constants(mi, mech, 20)
    emu = HostEmu(mech.source(template), tag="dme_iso")
    Y, F = g["dme_nb_iso_20_y"], g["dme_nb_iso_20_f"]
    out, _ = emu.rhs(Y, np.tile(row, (len(Y), 1)), 20)
    for k in range(len(Y)):
        assert rowwise_err(out[k], F[k], 6) < 1e-12


def test_emulated_rk4_vs_reference_trajectory(template):
    g = np.load(os.path.join(G, "g3_rk4.npz"))
    mi = INP.dme_script_input()
    mech = plan.Mechanism(mi)
    nm, row = plan.member_constants(mi, mech, 20)
    emu = HostEmu(mech.source(template), tag="dme_script")
    traj = g["dme_script_20_traj"]
    y, flags = emu.rk4(plan.initial_state(nm, mech, 20), row, 20, 1e-5, 200)
    assert not flags.any()
    scale = np.max(np.abs(traj[:, -1].reshape(7, 20)), axis=1, keepdims=True)
    assert np.max(np.abs(y[0].reshape(7, 20) - traj[:, -1].reshape(7, 20))/scale) < 1e-11


def test_device_flags_in_emulation(template):
    mi = INP.dme_notebook_input()
    mech = plan.Mechanism(mi)
    nm, row = plan.member_constants(mi, mech, 20)
    emu = HostEmu(mech.source(template), tag="dme_nb")
    y = plan.initial_state(nm, mech, 20).reshape(7, 20).copy()
    y[6, 5] = -1.5          # T = Tf*(1-1.5) < 0  -> log(T) domain error in Ln_KP1
    _, flags = emu.rhs(y.flatten(), row, 20)
    assert flags[0] & lowering.FLAG_DOMAIN


def test_result_packing_matches_reference_schema():
    g4 = np.load(os.path.join(G, "g4_tight_dme_script_bdf.npz"))
    mi = INP.dme_script_input()
    mech = plan.Mechanism(mi)
    nm, _ = plan.member_constants(mi, mech, 20)
    Y = np.concatenate([g4["dataYCons1_4"].flatten(), g4["dataYTemp1_4"].flatten()])
    d = pack_interval(Y, nm, mech, 20, 0.5, "N2")
    for key in ("dataYs", "dataYCons1", "dataYCons2", "dataYTemp1", "dataYTemp2", "dataXs"):
        np.testing.assert_allclose(d[key], g4[key + "_4"], rtol=1e-15, atol=0)
    assert d["labelList"] == INP.DME_COMPONENTS + ["Temperature"] and d["indexList"] == [6, 7, 6]
    assert d["dataShape"] == () and d["successStatus"] is True and d["modelId"] == "N2"


@pytest.mark.parametrize("name", ["dme_script", "dme_nb"])
def test_emulated_full_run_vs_tight_scipy_reference(name, template):
    """Accuracy contract of the explicit stepper, checked on the CPU with the host build of the
    generated source: RK4 dt=2.5e-6 s over the whole 0.5 s transient vs the reference integrated
    by LSODA at rtol=1e-10 (golden G4): outlet <= 1e-8 relative (requirement: 1e-6)."""
    g = np.load(os.path.join(G, "g4_tight_%s_lsoda.npz" % name))
    mi = INP.ALL_N2_INPUTS[name]()
    mech = plan.Mechanism(mi)
    nm, row = plan.member_constants(mi, mech, 20)
    emu = HostEmu(mech.source(template), tag=name)
    y = plan.initial_state(nm, mech, 20)
    worst = 0.0
    for k in range(5):
        y, fl = emu.rk4(y, row, 20, 2.5e-6, 40000)
        y = y[0]
        assert not fl.any()
        d = pack_interval(y, nm, mech, 20, 0.1*(k + 1), "N2")
        a, b = d["dataYs"][:, -1], g["dataYs_%d" % k][:, -1]
        worst = max(worst, np.max(np.abs(a - b)/np.abs(b)))
    assert worst < 1e-8, worst


def test_generated_node_physics_near_steady_states(template):
    from parity import backward_ok
    g = np.load(os.path.join(G, "g2_rhs.npz"))
    Y, F = g["dme_script_20_transient_y"], g["dme_script_20_transient_f"]
    mi = INP.dme_script_input()
    mech = plan.Mechanism(mi)
    _, row = plan.member_constants(mi, mech, 20)
    emu = HostEmu(mech.source(template), tag="dme_script")
    out, flags = emu.rhs(Y, np.tile(row, (len(Y), 1)), 20)
    fv = O.make_rhs_vec(O.setup_n2(mi, 20))
    for k in range(len(Y)):
        ok, d = backward_ok(out[k], F[k], fv, Y[k], 7)
        assert ok, (k, d)


def test_generated_n1_node_function_vs_reference(template):
    """Steady-state sibling N1: the generated rmt_n1_rhs (host build) against the reference's
    modelEquationN1 outputs captured in golden G6, and the N1 member row against the oracle setup."""
    g = np.load(os.path.join(G, "g6_n1.npz"))
    mi = INP.n1_notebook_input()
    mech = plan.Mechanism(mi)
    nm, row = plan.member_constants_n1(mi, mech)
    pr = O.setup_n1(mi)
    assert relerr(1.0/row[plan.MEMBER1_FIELDS["INV_HECOTE"]], pr["GaHeCoTe0"]) < 1e-14
    assert relerr(1.0/row[plan.MEMBER1_FIELDS["INV_MACOTE"]], pr["GaMaCoTe0"][0]) < 1e-14
    emu = HostEmu(mech.source(template), tag="dme_n1")
    Y, F = g["rhs_y"], g["rhs_f"]
    out, flags = emu.n1_rhs(Y, np.tile(row, (len(Y), 1)))
    assert not flags.any()
    for k in range(len(Y)):
        # k = 2 is the outlet state, close to chemical equilibrium: the rates are differences of
        # nearly equal forward/backward terms there (same conditioning issue as tests/parity.py)
        assert relerr(out[k], F[k]) < (1e-11 if k < 2 else 1e-7), (k, out[k], F[k])


def test_n1_analytic_jacobian_vs_forward_differences(template):
    """Model N1's rmt_n1_rhs_jac (host build of the generated source): its right-hand side equals rmt_n1_rhs and
    its Jacobian - rates differentiated symbolically by T, P, x_i, C_i, the rest by hand - equals the (S+2)
    forward differences it replaces to FD accuracy, at the reference-generated G6 states."""
    g = np.load(os.path.join(G, "g6_n1.npz"))
    mi = INP.n1_notebook_input()
    mech = plan.Mechanism(mi)
    _, row = plan.member_constants_n1(mi, mech)
    emu = HostEmu(mech.source(template, defines={"RMT_WITH_N1": "1"}), tag="dme_n1jac", openmp=False)
    Y = g["rhs_y"]
    jan, jfd, fan, fref = emu.n1_jac(Y, np.tile(row, (len(Y), 1)))
    assert np.max(np.abs(fan - fref)/np.maximum(np.abs(fref), 1e-300)) < 1e-12
    for k in range(len(Y)):
        scale = np.max(np.abs(jfd[k]), axis=1, keepdims=True)          # row-relative: the rows differ by 1e6 in size
        assert np.max(np.abs(jan[k] - jfd[k])/scale) < 1e-4, (k, np.max(np.abs(jan[k] - jfd[k])/scale))   # (FD noise of the trace species)


def test_benchmark_mesh_golden_g8_vs_emulated_rk4(template):
    """Golden G8 (SciPy DOP853, rtol 1e-10, on the oracle's vectorised RHS at zNo = 1024;
    tools/make_mesh_golden.py) cross-checked on the CPU by an independent integration: the host
    build of the generated kernel source under the reference's RK4 at dt = 2e-6 s, 50 000 steps to
    the first output time t = 0.1 s (about 20 s on one core)."""
    p = os.path.join(G, "g8_mesh1024_dme_nb_dop853.npz")
    if not os.path.exists(p):
        pytest.skip("golden G8 not generated")
    g = np.load(p)
    assert int(g["zNo"]) == 1024 and str(g["method"]) == "DOP853" and float(g["rtol"]) <= 1e-10
    assert int(g["done"]) == len(g["times"]) == g["states"].shape[0] and g["states"].shape[1] == 7*1024
    pr = O.setup_n2(INP.dme_notebook_input(), 1024)
    for k in range(int(g["done"])):          # physical profiles
        pk = O.pack_interval(g["states"][k], pr, float(g["times"][k]))
        assert np.allclose(np.sum(pk["dataYs"][:6], axis=0), 1.0, atol=1e-12)
        assert 520.0 < pk["dataYs"][6].min() and pk["dataYs"][6].max() < 700.0
    mi = INP.dme_notebook_input()
    mech = plan.Mechanism(mi)
    nm, row = plan.member_constants(mi, mech, 1024)
    emu = HostEmu(mech.source(template), tag="dme_nb")
    y, flags = emu.rk4(plan.initial_state(nm, mech, 1024), row, 1024, 2e-6, 50000)
    assert not flags.any()
    assert rowwise_err(y[0], g["states"][0], 7) < 1e-7


# ----------------------------------------------------------------------------- analytic node Jacobian
@pytest.mark.parametrize("name", list(INP.ALL_N2_INPUTS))
def test_symbolic_gradient_of_the_rate_dag_vs_central_differences(name):
    """lowering.Lowered.gradient: d rate / d (T, x_i, C_i) of the device DAG against central differences
    of the same DAG (host evaluation), every non-zero partial; partials the gradient omits are zero."""
    mech = plan.Mechanism(INP.ALL_N2_INPUTS[name]())
    dag = mech.device_dag()
    gr = dag.gradient()
    assert "P" not in gr.wrt and gr.n_rates == mech.R
    rng = np.random.default_rng(7)
    S = mech.S
    for trial in range(3):
        x = rng.uniform(0.05, 1.0, S)
        x /= x.sum()
        T, P = float(rng.uniform(480, 650)), float(rng.uniform(2e6, 6e6))
        C = x*P/(8.314472*T)
        r, dr = gr.evaluate_all(T, P, list(x), list(C))
        np.testing.assert_allclose(r, dag.evaluate(T, P, list(x), list(C)), rtol=1e-14)
        for q in range(mech.R):
            for k in ["T"] + ["x%d" % i for i in range(S)] + ["C%d" % i for i in range(S)]:
                def at(sign, h=1e-6):
                    T2, x2, C2 = T, list(x), list(C)
                    if k == "T":
                        T2 = T*(1 + sign*h)
                    elif k[0] == "x":
                        x2[int(k[1:])] *= 1 + sign*h
                    else:
                        C2[int(k[1:])] *= 1 + sign*h
                    return dag.evaluate(T2, P, x2, C2)[q]
                base = T if k == "T" else (x[int(k[1:])] if k[0] == "x" else C[int(k[1:])])
                fd = (at(1) - at(-1))/(2e-6*base)
                an = dr[q].get(k, 0.0)
                assert abs(fd - an) <= 1e-7*max(abs(fd), abs(an)) + 1e-9*abs(r[q])/base, (q, k, fd, an)


@pytest.mark.parametrize("name,zNo", [("dme_nb", 20), ("dme_script", 20), ("ch4", 20), ("syn12", 20), ("dme_nb", 1024)])
def test_analytic_node_jacobian_vs_forward_differences(name, zNo, template):
    """rmt_node_jac (the stiff stepper's analytic -d f_z/d y_z, host build of the generated source) against
    the V forward differences of the node function it replaces, at the reference-generated G2 states: equal
    to FD accuracy (~1e-6 of the node's largest entry) wherever no concentration sits on the 1e-30 clamp
    (at the kink the one-sided difference and the derivative legitimately differ)."""
    g = np.load(os.path.join(G, "g2_rhs.npz"))
    mi = INP.ALL_N2_INPUTS[name]()
    mech = plan.Mechanism(mi)
    emu = HostEmu(mech.source(template, defines={"RMT_WITH_ROS4": "1"}), tag="jac_" + name, openmp=False)
    _, row = plan.member_constants(mi, mech, zNo)
    V = mech.V
    checked = 0
    for Y in g["%s_%d_y" % (name, zNo)]:
        jan, jfd = emu.node_jac(Y, row, zNo)
        ok = np.all(Y.reshape(V, zNo)[:mech.S] > 1e-30, axis=0)
        ok[1:] &= ok[:-1]                       # the upstream node's clamp enters through `up`
        scale = np.max(np.abs(jfd), axis=(1, 2), keepdims=True)
        assert np.max((np.abs(jan - jfd)/scale)[ok]) < 2e-5
        checked += int(ok.sum())
    assert checked >= zNo


# ----------------------------------------------------------------------------- round-2 host logic
def test_rk45_geometry_and_lds_budget():
    """on-chip RK45 geometry: reactors that fit one workgroup get the on-chip kernel with as many of its four
    long-lived vectors in LDS as fit 136 KiB; longer ones fall back to the memory-resident kernel's block."""
    from rmt_app_amd.n2 import rk45_block, rk45_geometry
    assert rk45_geometry(7, 1024) == (512, 2, {"RMT_RK45_LDS": "2"})
    assert rk45_geometry(7, 20) == (64, 1, {"RMT_RK45_LDS": "2"})
    assert rk45_geometry(13, 512) == (256, 2, {"RMT_RK45_LDS": "2"})
    # longer reactors (model N2): chunks of the on-chip size on several CUs ...
    assert rk45_geometry(7, 4096) == (512, 2, {"RMT_RK45_LDS": "2"})
    assert rk45_geometry(13, 1024) == (256, 2, {"RMT_RK45_LDS": "2"})
    assert rk45_geometry(7, 16384) == (512, 2, {"RMT_RK45_LDS": "2"})
    # ... unless that takes more chunks than a team may have, or the caller asks for the memory-resident kernel
    assert rk45_geometry(7, 1024*65) == (rk45_block(7, 1024*65), 1, {})
    assert rk45_geometry(7, 4096, chain=False) == (rk45_block(7, 4096), 1, {})
    assert rk45_geometry(13, 1024, chain=False) == (rk45_block(13, 1024), 1, {})
    for V, N in ((7, 1024), (13, 512), (4, 700), (8, 1024)):
        block, npt, defs = rk45_geometry(V, N)
        assert block*npt >= N and int(defs["RMT_RK45_LDS"])*V*block*npt*8 <= 136*1024
    for V, N in ((7, 4096), (13, 1024), (7, 16384)):
        block, npt, defs = rk45_geometry(V, N)
        assert int(defs["RMT_RK45_LDS"])*V*block*npt*8 <= 136*1024
    # a known ensemble size that leaves CUs idle: one node per lane, finer chunks, all of them co-resident
    assert rk45_geometry(7, 1024, E=256)[:2] == (512, 2)          # the device is full: one workgroup per reactor
    assert rk45_geometry(7, 1024, E=128)[:2] == (512, 1)          # 2 chunks x 128 reactors
    assert rk45_geometry(7, 1024, E=64)[:2] == (256, 1)           # 4 chunks x 64
    assert rk45_geometry(7, 1024, E=1)[:2] == (256, 1)
    assert rk45_geometry(7, 4096, E=32)[:2] == (512, 1)           # 8 chunks x 32
    assert rk45_geometry(7, 4096, E=8)[:2] == (256, 1)
    assert rk45_geometry(7, 4096, E=64)[:2] == (512, 2)           # 4 chunks x 64 fill the device already
    assert rk45_geometry(7, 16384, E=1)[:2] == (256, 1)           # 64 chunks = RMT_N2_MAX_CHUNKS
    assert rk45_geometry(13, 1024, E=64)[:2] == (256, 1)
    assert rk45_geometry(13, 512, E=64)[:2] == (128, 1)
    assert rk45_geometry(13, 1024, E=256)[:2] == (256, 2)
    assert rk45_geometry(7, 200, E=1)[:2] == rk45_geometry(7, 200)[:2]
    for V, N, E in ((7, 1024, 64), (13, 1024, 64), (7, 16384, 1), (7, 4096, 8)):
        block, npt, defs = rk45_geometry(V, N, E=E)
        assert E*(-(-N//(block*npt))) <= 256 and -(-N//(block*npt)) <= 64


def test_device_stats_totals():
    """device-stats: adaptive steppers report the SUM of accepted steps over the members as `steps` and
    steps*zNo node-steps (round 1 multiplied by the member count once more); RODAS4 = 6 stage evaluations +
    the node Jacobian per attempted step."""
    from rmt_app_amd.n2 import finish_stats
    acc, rej = np.array([10, 12, 9]), np.array([1, 0, 2])
    st = finish_stats({"steps": 0, "accepted": acc.copy(), "rejected": rej.copy()}, "hip-ros4", 3, 5, 20, 7)
    assert st["steps"] == 31 and st["node_steps"] == 31*20 and st["rhs_evals"] == int(np.sum((6 + 7)*(acc + rej)))
    st = finish_stats({"steps": 0, "accepted": acc.copy(), "rejected": rej.copy()}, "hip-rk45", 3, 5, 20, 7)
    assert st["rhs_evals"] == int(np.sum(6*(acc + rej))) + 3*5 and st["node_steps"] == 31*20
    st = finish_stats({"steps": 1000, "accepted": None}, "hip-rk4", 3, 5, 20, 7)
    assert st["node_steps"] == 1000*20*3


def test_isa_statistics_of_a_cross_compiled_kernel(template):
    """rmt_app_amd/isa.py (what bench.py derives its fp64 op count from): the step loop of rmt_n2_rk4_reg is
    found, dominated by fp64 VALU, and the kernel digest depends on the kernel's code only."""
    from rmt_app_amd import isa
    mech = plan.Mechanism(INP.dme_notebook_input())
    blob = hipbind.compile_cached(mech.source(template, False, 64, 1), mech.digest(template, False, 64, 1), "gfx950")
    st = isa.kernel_stats(blob, "rmt_n2_rk4_reg")
    loop = st["step_loop"]
    assert 1000 < loop["valu_f64"] < 2500 and loop["valu_f64"] > 0.6*loop["valu"] and loop["scratch"] == 0
    assert loop["instructions"] < st["whole"]["instructions"]
    res = isa.kernel_resources(blob, "rmt_n2_rk4_reg")
    assert res["vgpr_count"] <= 256 and res["private_segment_fixed_size"] == 0
    # the kernel digest identifies the kernel's machine code: reproducible across compiles of one source,
    # different as soon as the kernel's instructions change
    # (two fresh compiles: a cached blob may come from the OTHER hipRTC of this image - a process that imported
    # torch first runs torch's bundled ROCm 7.0 compiler, otherwise /opt/rocm's 7.2 one)
    blob1, _ = hipbind.compile_source(mech.source(template, False, 64, 1))
    blob1b, _ = hipbind.compile_source(mech.source(template, False, 64, 1))
    d1 = isa.kernel_stats(blob1, "rmt_n2_rk4_reg")["kernel_digest"]
    assert isa.kernel_stats(blob1b, "rmt_n2_rk4_reg")["kernel_digest"] == d1
    blob2, _ = hipbind.compile_source(mech.source(template, False, 64, 1, None, {"RMT_DPP": "0"}))
    assert isa.kernel_stats(blob2, "rmt_n2_rk4_reg")["kernel_digest"] != d1


def test_traffic_record_is_keyed_by_kernel_digest():
    import bench
    rec = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    (dig, ent), = list(rec.items())[:1]
    assert bench.tracked_traffic(dig, ent["members"], ent["nodes"])[0] == ent["bytes_per_launch"]
    assert bench.tracked_traffic("0"*24, ent["members"], ent["nodes"]) == (None, None)
    assert bench.tracked_traffic(dig, ent["members"] + 1, ent["nodes"]) == (None, None)
    ref = bench.reference_cpu_rate(1024)
    assert 100 < ref["value"] < 5000 and ref["cores"] == 1


def test_default_compile_options_and_caller_override():
    """rmt_n2_compile adds the library's default options (part of the cache key through hipbind.hiprtc_tag); an
    `extra_opts` that sets the machine-LICM switch itself replaces the default instead of repeating it (an -mllvm
    switch may be given once)."""
    from rmt_app_amd import hipbind
    opts = hipbind.lib().rmt_n2_compile_options().decode()
    assert "-O3" in opts and "-disable-machine-licm" in opts
    src = 'extern "C" __global__ void k(double* a) { for (int i = 0; i < 8; ++i) a[i] = a[i]*1.25 + 3.0; }'
    blob, _ = hipbind.compile_source(src)
    again, _ = hipbind.compile_source(src, extra_opts="-mllvm -disable-machine-licm")
    assert blob[:4] == b"\x7fELF" and again[:4] == b"\x7fELF"
    assert len(hipbind.hiprtc_tag()) == 8


# ----------------------------------------------------------------------------- MODEL_SETTING['GaMaCoTe0'] (golden G11)
@pytest.fixture
def gamacote_fix():
    from rmt_app_amd import MODEL_SETTING
    old = MODEL_SETTING["GaMaCoTe0"]
    MODEL_SETTING["GaMaCoTe0"] = "FIX"
    try:
        yield
    finally:
        MODEL_SETTING["GaMaCoTe0"] = old


def test_model_setting_n2_raises_like_the_reference(gamacote_fix, capsys):
    """Under MODEL_SETTING['GaMaCoTe0'] != "MAX" the reference's N2 run ends in numpy's ValueError (the RHS assigns
    the feed-concentration array to one element, pbHomoReactor.py:3901-3904) - recorded from the reference in golden
    G11; rmtExe here raises the same exception (printing it first, PyREMOT/rmt.py:78-80), and M2 is unaffected."""
    from rmt_app_amd import rmtExe
    g = json.load(open(os.path.join(G, "g11_model_setting.json")))
    assert g["N2"]["raises"] == "ValueError" and g["N1"]["raises"] is None and g["M2"]["raises"] is None
    with pytest.raises(ValueError) as e:
        rmtExe(INP.dme_notebook_input())
    assert str(e.value) == g["N2"]["message"]
    assert g["N2"]["message"] in capsys.readouterr().out
    mech = plan.Mechanism(INP.m2_dme_input())
    plan.member_constants_m2(INP.m2_dme_input(), mech, 20)            # no setting involved


def test_model_setting_n1_per_species_scaling_vs_reference(gamacote_fix, template):
    """Model N1 under the per-species scaling: the generated node function (RMT_N1_SCALE_FIX) and the oracle against
    the reference's modelEquationN1 probes recorded under that setting (golden G11), and the analytic Jacobian
    against forward differences."""
    g = np.load(os.path.join(G, "g11_n1_fix.npz"))
    mi = INP.n1_notebook_input()
    mech = plan.Mechanism(mi)
    nm, row = plan.member_constants_n1(mi, mech)
    pr = O.setup_n1(mi, gamacote="FIX")
    assert relerr(nm["GaMaCoTe0"], pr["GaMaCoTe0"]) < 1e-14 and np.ndim(nm["SpCoi0_Set"]) == 1
    Y, F = g["rhs_y"], g["rhs_f"]
    emu = HostEmu(mech.source(template, defines={"RMT_N1_SCALE_FIX": "1", "RMT_WITH_N1": "1"}), tag="dme_n1fix", openmp=False)
    out, flags = emu.n1_rhs(Y, np.tile(row, (len(Y), 1)))
    assert not flags.any()
    for k in range(len(Y)):
        assert relerr(O.n1_rhs(0.37, Y[k], pr), F[k]) < 1e-12, k
        assert relerr(out[k], F[k]) < (1e-11 if k < 2 else 1e-7), (k, out[k], F[k])
    # ... and it is a different function from the "MAX" one
    g6 = np.load(os.path.join(G, "g6_n1.npz"))
    assert relerr(F[0], g6["rhs_f"][0]) > 0.1
    jan, jfd, fan, fref = emu.n1_jac(Y, np.tile(row, (len(Y), 1)))
    assert np.max(np.abs(fan - fref)/np.maximum(np.abs(fref), 1e-300)) < 1e-12
    for k in range(len(Y)):
        scale = np.max(np.abs(jfd[k]), axis=1, keepdims=True)
        assert np.max(np.abs(jan[k] - jfd[k])/scale) < 1e-4, k
    # the reference's own default-tolerance profile under the setting vs the oracle run the same way
    ref = O.run_n1(mi, zNo=100, method="LSODA", gamacote="FIX")
    assert np.max(np.abs(ref["dataYs"] - g["dataYs"])/np.abs(g["dataYs"])) < 5e-3


# ----------------------------------------------------------------------------- ensemble result packing
@pytest.mark.parametrize("process_type", ["non-iso-thermal", "iso-thermal"])
def test_batched_packing_equals_per_member_packing(process_type):
    """pack_intervals (all members of a sweep at once) against pack_interval (sortResult5 per member): every key,
    shape, dtype and bit."""
    from rmt_app_amd.ensemble import expand_members
    from rmt_app_amd.n2 import pack_intervals
    base = INP.dme_notebook_input(process_type=process_type)
    members = expand_members(base, {"temperature": np.linspace(503, 543, 9), "pressure": [3e6, 5e6]})
    mech = plan.Mechanism(base)
    named = [plan.member_constants(m, mech, 40)[0] for m in members]
    Y = np.random.default_rng(5).random((len(members), mech.V*40))
    one = [pack_interval(Y[e], named[e], mech, 40, 0.1, "N2") for e in range(len(members))]
    for a, b in zip(one, pack_intervals(Y, named, mech, 40, 0.1, "N2")):
        assert list(a) == list(b)
        for k in a:
            if isinstance(a[k], np.ndarray):
                assert a[k].shape == b[k].shape and a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k]), k
            else:
                assert a[k] == b[k], k


def test_ensemble_output_outlet_keeps_only_the_last_node():
    """solver-config "ensemble-output": "outlet" (host-emulation stand-in for the device): every dataPack entry holds
    the outlet column of the full-profile run, same schema, one axial point."""
    import emu_device
    from rmt_app_amd import n2, rmtExe

    def run(**extra):
        mi = INP.dme_notebook_input(ivp="hip-rk4", period=2e-4)
        mi["solver-config"].update(dict({"quiet": True, "dt": 2e-6, "zNo": 24, "tNo": 2,
                                         "ensemble": {"temperature": [513.0, 533.0], "pressure": [4e6, 5e6]}}, **extra))
        real, n2.N2Device = n2.N2Device, emu_device.EmuDevice
        try:
            return rmtExe(mi)["resModel"]["ensemble"]
        finally:
            n2.N2Device = real
    full, out = run(), run(**{"ensemble-output": "outlet"})
    assert len(full) == len(out) == 4
    for f, o in zip(full, out):
        for k in range(2):
            a, b = f["dataPack"][k], o["dataPack"][k]
            assert list(a) == list(b) and b["dataYs"].shape == (7, 1) and list(b["dataXs"]) == [1.0]
            for key in ("dataYs", "dataYCons1", "dataYCons2", "dataYTemp2"):
                np.testing.assert_array_equal(b[key][:, 0], a[key][:, -1])
            assert b["dataYTemp1"][0] == a["dataYTemp1"][-1] and b["dataTime"] == a["dataTime"]
    with pytest.raises(ValueError):
        run(**{"ensemble-output": "everything"})


# ----------------------------------------------------------------------------- cache of the temperature-only rate constants
def test_kcache_plan_and_emission_for_the_test_mechanisms(template):
    """lowering.Lowered.kcache_plan: the temperature-only exp / log roots of every test mechanism, their cache slots,
    and the generated source with the cached section still builds for the host (where the section is discarded) and
    reproduces the reference RHS."""
    dme = plan.Mechanism(INP.dme_notebook_input())
    p = dme.device_dag().kcache_plan()
    kinds = [p["kind"][r] if isinstance(p["kind"][r], str) else "lin" for r in p["roots"]]
    assert kinds.count("lin") == 6 and kinds.count("log") == 1 and kinds.count("gen") == 3 and p["slots"] == 14
    assert dme.kcache_slots() == 14 and dme.kcache_fits(False, 512, 2) and not dme.kcache_fits(True, 512, 2)
    syn = plan.Mechanism(INP.syn12_input())
    assert syn.kcache_slots() == 9                         # 1/T_ref + eight Arrhenius constants
    assert plan.Mechanism(INP.ch4_input()).kcache_slots() == 0
    src = dme.source(template, defines={"RMT_KCACHE": "1"})
    assert "#define RMT_KC_SLOTS 14" in src and "MODE == 2" in src and "kc.leave(" in src
    g = np.load(os.path.join(G, "g2_rhs.npz"))
    Y, F = g["dme_nb_20_y"], g["dme_nb_20_f"]
    _, row = plan.member_constants(INP.dme_notebook_input(), dme, 20)
    out, flags = HostEmu(src, tag="dme_kc").rhs(Y, np.tile(row, (len(Y), 1)), 20)
    assert not flags.any()
    for k in range(len(Y)):
        assert rowwise_err(out[k], F[k], dme.V) < 1e-12
    from rmt_app_amd.n2 import device_source
    with pytest.raises(ValueError):                        # the 12-species geometry keeps its RK4 vectors in LDS: no room
        device_source(syn, row if False else plan.member_constants(INP.syn12_input(), syn, 1024)[1], 1024,
                      defines={"RMT_KCACHE": "1"})


def test_kcache_basis_decomposition_of_the_equilibrium_constants():
    """kcache_plan("basis"): ln K of the three DME equilibrium constants as a linear combination of T^n and log T - the
    ratio K(T2)/K(T1) of the DAG's own evaluation equals exp of the same combination of the basis functions' differences
    (what the cached path computes, kernels' MODE 2), and the emitted source carries the difference formulas."""
    import math
    dme = plan.Mechanism(INP.dme_notebook_input())
    dag = dme.device_dag()
    p = dag.kcache_plan("basis")
    based = [r for r in p["roots"] if isinstance(p["kind"][r], tuple) and p["kind"][r][0] == "basis"]
    assert len(based) == 3 and p["slots"] == 12 and p["tslot"] == 11 and not p["outside_exp"]
    sub = lowering.Lowered(dag.g, based, dag.S)
    x = [0.5, 0.2, 0.05, 0.2, 0.03, 0.02]
    T1, T2 = 521.3, 521.3 + 0.0612
    K1, K2 = sub.evaluate(T1, 5e6, x, x), sub.evaluate(T2, 5e6, x, x)
    for r, k1, k2 in zip(based, K1, K2):
        sg, lnb = dag._EXP_ROOTS[dag.g.nodes[r][0]]
        d = 0.0
        for key, c in p["kind"][r][1].items():
            d += c*((math.log(T2) - math.log(T1)) if key == ("log",) else (T2**key[1] - T1**key[1]))
        assert abs(k2/k1 - math.exp(sg*lnb*d)) < 2e-13*abs(k2/k1)
    src = dme.source(hipbind_template(), defines={"RMT_KCACHE": "1", "RMT_KCACHE_GEN": "2"})
    assert "#define RMT_KC_SLOTS 12" in src and "kc_dp3" in src and "kc_dm3" in src and "rmt_exp_sel<KC::enabled>(" in src
    # a mechanism without such constants: nothing changes; exponents that do not decompose stay out of the cache
    syn = plan.Mechanism(INP.syn12_input())
    assert syn.kcache_slots("basis") == syn.kcache_slots(False) == 9 and syn.kcache_small_exp("basis")


def test_kcache_plans_of_a_mechanism_with_every_kind_of_exponent():
    """Arrhenius -> "lin", polynomial / log exponent -> "basis", exp(-sqrt(T)/40) does not decompose and stays out of the
    cache under the "basis" policy (two slots under policy True), an exp of the composition is never a root: with it the
    caching kernel must keep the big exp table (no RMT_KC_SMALL_EXP, hence no equilibrium constants in the cache)."""
    from rmt_app_amd.n2 import kcache_choice
    for comp in (True, False):
        mech = plan.Mechanism(INP.ch4_arrhenius_input(composition_exp=comp))
        dag = mech.device_dag()
        p = dag.kcache_plan("basis")
        kinds = sorted(k if isinstance(k, str) else k[0] for k in p["kind"].values())
        assert kinds == ["basis", "lin", "log"] and p["outside_exp"]         # (the sqrt exponent is outside either way)
        assert sorted(k if isinstance(k, str) else k[0] for k in dag.kcache_plan(True)["kind"].values()) == \
            ["gen", "gen", "lin", "log"]
        assert not mech.kcache_small_exp("basis")
        defs, lds = kcache_choice(mech, 20, False, 64, 1, None, None)
        assert defs == {"RMT_KCACHE": "1", "RMT_KCACHE_GEN": "0", "RMT_KC_REFRESH": "8"} and lds is None
        src = mech.source(hipbind_template(), block=64, npt=1, defines=defs)
        assert "#define RMT_KC_SLOTS 3" in src                                 # 1/T_ref, the Arrhenius constant, log T_ref
        with pytest.raises(ValueError):
            from rmt_app_amd.n2 import device_source
            _, row = plan.member_constants(INP.ch4_arrhenius_input(composition_exp=comp), mech, 20)
            device_source(mech, row, 20, block=64, npt=1, defines={"RMT_KCACHE": "1", "RMT_KCACHE_GEN": "2",
                                                                    "RMT_KC_SMALL_EXP": "1"})


def test_kcache_basis_decomposition_random_exponents():
    """lowering's basis decomposition against the DAG's own evaluation for random exponents a/T + b ln T + c T + d T^2 +
    e/T^2 + f T^3 (written in different algebraic forms: products, quotients, powers, nested sums)."""
    rng = np.random.default_rng(20260412)
    for trial in range(12):
        c = rng.normal(size=7)*np.array([3000.0, 4.0, 2e-3, 3e-6, 5e4, 4e-9, 1.0])
        forms = [      # (the logarithm is handed in: the tracer rebinds `math` in the rate lambda itself only)
            lambda T, c, log: c[0]/T + c[1]*log(T) + c[2]*T + c[3]*T**2 + c[4]/T**2 + c[5]*T*T*T + c[6],
            lambda T, c, log: (c[0] + c[2]*T*T)/T + log(T)*c[1] - (-c[3])*(T*T) + c[4]*(1.0/T)**2 + (c[5]*T)*T**2 + c[6],
            lambda T, c, log: ((c[0] + c[4]/T)/T + c[6]) + (c[1]*log(T) + T*(c[2] + T*(c[3] + T*c[5]))),
        ]
        F = forms[trial % 3]
        f = lambda T, F=F, c=c: F(T, c, math.log)
        low = lowering.trace({}, {"r1": lambda x, F=F, c=c: math.exp(F(x['T'], c, math.log))}, 1)
        if trial % 2:
            low = low.optimize()                  # the device build's strength-reduced DAG decomposes as well
        p = low.kcache_plan("basis")
        (r,) = [r for r in p["roots"] if isinstance(p["kind"][r], tuple) and p["kind"][r][0] == "basis"]
        T1, T2 = 480.0 + 20.0*trial, 480.0 + 20.0*trial + 0.05
        d = sum(cf*((math.log(T2) - math.log(T1)) if key == ("log",) else (T2**key[1] - T1**key[1]))
                for key, cf in p["kind"][r][1].items())
        assert abs(d - (f(T2) - f(T1))) < 1e-12*max(1.0, abs(f(T1))), (trial, d, f(T2) - f(T1))


def hipbind_template():
    from rmt_app_amd import hipbind
    return hipbind.kernel_template()


def test_refresh_period_and_compile_options_rules():
    """Host rules that mirror the kernels': n2.kc_period (steps between two moves of the cache's reference point: at most
    RMT_KC_REFRESH, and no reference point older than 13 us of model time) and n2.compile_options (the register-pressure
    trackers for 512 x 2 code objects only - not with optional kernel families, not for RK45 builds, not twice)."""
    from rmt_app_amd.n2 import compile_options, kc_period, rk45_geometry
    d = {"RMT_KC_REFRESH": "8"}
    assert [kc_period(d, dt) for dt in (1e-6, 2e-6, 2.5e-6, 5e-6, 1e-5, 1e-4)] == [8, 6, 5, 2, 1, 1]
    assert kc_period({}, 2e-6) == 1 and kc_period({"RMT_KC_REFRESH": "4"}, 1e-6) == 4 and kc_period(d, 0.0) == 1
    assert kc_period({"RMT_KC_REFRESH": "8", "RMT_KC_MAX_AGE": "4e-6"}, 2e-6) == 2
    tr = "-mllvm -amdgpu-use-amdgpu-trackers=1"
    assert compile_options(512, 2) == tr and compile_options(512, 2, (), "-O3") == "-O3 " + tr
    assert compile_options(512, 2, ("ros4",)) == "" and compile_options(256, 1) == "" and compile_options(128, 1, (), "-g") == "-g"
    assert compile_options(512, 2, (), tr) == tr
    block, npt, defs = rk45_geometry(7, 1024, E=256)
    assert (block, npt) == (512, 2) and compile_options(block, npt, (), "", defs) == ""


def test_kcache_choice_and_the_code_object_it_builds():
    """n2.kcache_choice: the cache is switched on for the measured geometry only (512 x 2, model N2, fp64, a reactor that
    fits the workgroup), with y_n in LDS; an explicit RMT_KCACHE or another lds_state is left alone.  The code object then
    carries the plain stepper as rmt_n2_rk4_reg_redo and its cached step loop has no scratch access."""
    from rmt_app_amd import hipbind, isa
    from rmt_app_amd.n2 import device_source, kcache_choice
    dme = plan.Mechanism(INP.dme_notebook_input())
    on = ({"RMT_KCACHE": "1", "RMT_KCACHE_GEN": "2", "RMT_KC_SMALL_EXP": "1", "RMT_KC_NODE_MAJOR": "1",
           "RMT_KC_REFRESH": "8"}, 1)
    assert kcache_choice(dme, 1024, False, 512, 2, None, None) == on
    assert kcache_choice(dme, 1000, False, 512, 2, 1, {"X": "1"}) == (dict(on[0], X="1"), 1)
    assert kcache_choice(dme, 1024, False, 512, 2, 0, None) == ({}, 0)                  # the caller's lds_state wins
    assert kcache_choice(dme, 1024, False, 512, 2, None, {"RMT_KCACHE": "0"}) == ({"RMT_KCACHE": "0"}, None)
    assert kcache_choice(dme, 1024, True, 512, 2, None, None) == ({}, None)             # fp32
    chain = ({"RMT_KCACHE_CHAIN": "1", "RMT_KCACHE_GEN": "0"}, 1)
    assert kcache_choice(dme, 4096, False, 512, 2, None, None) == chain                 # chained reactor: the chunks' cache
    assert kcache_choice(dme, 4096, False, 512, 2, None, {"RMT_KCACHE_CHAIN": "0"}) == ({"RMT_KCACHE_CHAIN": "0"}, None)
    assert kcache_choice(dme, 4096, False, 128, 1, None, None) == (chain[0], None)      # the small chunks of ONE long reactor
    assert kcache_choice(dme, 1024, False, 256, 1, None, None) == (chain[0], None)      # a small ensemble, chained
    assert kcache_choice(dme, 1024, False, 1024, 1, None, None) == ({}, None)           # one workgroup, another geometry
    assert kcache_choice(plan.Mechanism(INP.syn12_input()), 1024, False, 256, 2, None, None) == ({}, None)      # V = 13
    assert kcache_choice(plan.Mechanism(INP.ch4_input()), 1024, False, 512, 2, None, None) == ({}, None)   # nothing to cache
    m2 = plan.Mechanism(INP.m2_dme_input())                                             # model M2: 512 x 2 only, Arrhenius constants
    assert kcache_choice(m2, 1024, False, 512, 2, None, None) == ({"RMT_KCACHE": "1", "RMT_KCACHE_GEN": "0", "RMT_KC_REFRESH": "8"}, 1)
    assert kcache_choice(m2, 4096, False, 512, 2, None, None) == ({}, None) and kcache_choice(m2, 100, False, 128, 1, None, None) == ({}, None)
    _, row = plan.member_constants(INP.dme_notebook_input(), dme, 1024)
    block, npt, defs, src, key = device_source(dme, np.tile(row, (256, 1)), 1024)
    assert (block, npt) == (512, 2) and defs["RMT_KCACHE"] == "1" and "#define RMT_LDS_STATE 1" in src
    blob = hipbind.compile_cached(src, key, "gfx950")
    st = isa.kernel_stats(blob, "rmt_n2_rk4_reg")
    assert st["step_loop"]["scratch"] == 0 and st["step_loop"]["valu"] < 4100           # (both versions of stage 1)
    from bench import executed_step_mix
    mix, note = executed_step_mix(st["step_loop"], "rmt_n2_rk4_reg", defs, lambda extra: hipbind.compile_cached(
        *device_source(dme, np.tile(row, (256, 1)), 1024, defines=extra)[3:5], "gfx950"))
    assert mix["valu"] < 3150 and mix["scratch"] == 0 and "refresh step" in note         # (the plain stepper: 3689)
    assert isa.kernel_stats(blob, "rmt_n2_rk4_reg_redo")["whole"]["valu"] > 3000        # the plain stepper, same object
    lds = isa.kernel_resources(blob, "rmt_n2_rk4_reg")["group_segment_fixed_size"]      # the plan's LDS estimate holds
    assert lds <= 160*1024 and abs(lds - dme._kcache_lds(False, 512, 2, 1, "basis", True, True)) < 4096
    assert isa.kernel_resources(blob, "rmt_n2_rk4_reg_redo")["group_segment_fixed_size"] < 80*1024    # (the big exp table)
    _, row4 = plan.member_constants(INP.dme_notebook_input(), dme, 4096)
    block, npt, defs, src, key = device_source(dme, np.tile(row4, (256, 1)), 4096)
    assert (block, npt) == (512, 2) and defs["RMT_KCACHE_CHAIN"] == "1" and "#define RMT_LDS_STATE_CHAIN 1" in src
    blob = hipbind.compile_cached(src, key, "gfx950")
    assert isa.kernel_stats(blob, "rmt_n2_rk4_chain")["step_loop"]["scratch"] == 0
    assert isa.kernel_stats(blob, "rmt_n2_rk4_chain_redo")["whole"]["valu"] > 3000
    with pytest.raises(ValueError):
        dme.source(hipbind.kernel_template(), defines={"RMT_KCACHE": "1", "RMT_KCACHE_THR": "0.5"})


# ----------------------------------------------------------------------------- stiff stepper, quad layout (V > 8)
def test_ros4_quad_layout_has_no_sweep_loop_spills():
    """What the layout is for: the code object of the 12-species stiff stepper keeps the sweep loops free of scratch
    accesses (the node-per-lane kernel spilled 1000-1500 VGPRs per lane, 2-3.8 KB of scratch)."""
    import re
    import subprocess
    import tempfile
    from rmt_app_amd import hipbind
    from rmt_app_amd.n2 import device_source
    mech = plan.Mechanism(INP.syn12_input())
    _, row = plan.member_constants(INP.syn12_input(), mech, 512)
    _, _, defs, src, key = device_source(mech, np.tile(row, (64, 1)), 512, block=256, npt=1, features=("ros4",))
    assert defs["RMT_ROS_QUAD"] == "1"
    blob = hipbind.compile_cached(src, key, "gfx950")
    with tempfile.NamedTemporaryFile(suffix=".hsaco") as f:
        f.write(blob)
        f.flush()
        notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True,
                               text=True, check=True).stdout
    for kern in ("rmt_n2_ros4_mem", "rmt_n2_ros4_chain"):
        blk = [b for b in notes.split("- .agpr_count") if ".name:           %s" % kern in b or re.search(r"\.name:\s+%s\b" % kern, b)][0]
        scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1))
        assert scratch < 1024, (kern, scratch)           # the once-per-step Jacobian assembly still spills a little
