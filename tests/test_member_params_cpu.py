"""Ensemble members may differ in kinetic constants (scalar reaction-rates.VARS entries) - the reference
copies VARS into the lambdas' namespace on every call (PyREMOT/docs/rmtReaction.py:44-51), so a sweep over a
catalyst density or a rate constant is an ordinary loop over rmtExe there.  Here those constants become
per-reactor columns of the member row (plan.Mechanism(params=...)); anything else that differs from the base
mechanism must RAISE instead of silently running the base kinetics.  CPU side: detection, lowering, the generated
source under host emulation against the oracle (which calls each member's own lambdas like the reference)."""
import numpy as np
import pytest

import inputs as INP
from oracle import n2_oracle as O
from oracle.hostemu import HostEmu
from rmt_app_amd import ensemble as ENS
from rmt_app_amd import hipbind, lowering, plan
from parity import rowwise_err

CABEDE = [1171.2, 500.0, 1800.0]


def _sweep(base, values=CABEDE):
    return ENS.expand_members(base, [{"reaction-rates": {"VARS": {"CaBeDe": v}}} for v in values])


def test_varying_scalar_vars_are_found_in_vars_order():
    base = INP.dme_notebook_input()
    base["reaction-rates"]["VARS"]["k0_scale"] = 1.0
    members = ENS.expand_members(base, [
        {"reaction-rates": {"VARS": {"k0_scale": 2.0}}},
        {"reaction-rates": {"VARS": {"CaBeDe": 500.0}}, "operating-conditions": {"temperature": 530.0}},
        {}])
    assert ENS.member_parameters(base, members) == ["CaBeDe", "k0_scale"]      # VARS order, not discovery order
    assert ENS.member_parameters(base, ENS.expand_members(base, {"temperature": [510.0, 530.0]})) == []
    # equal values are not parameters, numpy scalars count as scalars
    same = ENS.expand_members(base, [{"reaction-rates": {"VARS": {"CaBeDe": np.float64(1171.2)}}}])
    assert ENS.member_parameters(base, same) == []


@pytest.mark.parametrize("override,key", [
    ({"reactions": {"R1": "CO2 + 3H2 <=> CH3OH + H2O"}}, "'reactions'"),
    ({"feed": {"components": {"shell": ["H2", "CO2", "H2O", "CO", "CH3OH"]}}}, "feed.components.shell"),
    ({"operating-conditions": {"period": 1.0}}, "operating-conditions.period"),
    ({"operating-conditions": {"process-type": "iso-thermal"}}, "operating-conditions.process-type"),
    ({"model": "M2"}, "'model'"),
    ({"reaction-rates": {"RATES": {"r1": lambda x: 0.0}}}, "reaction-rates.RATES"),
    ({"reaction-rates": {"VARS": {"K1": lambda x: 1.0}}}, "reaction-rates.VARS['K1']"),
    ({"reaction-rates": {"VARS": {"extra": 1.0}}}, "reaction-rates.VARS"),
])
def test_differences_the_member_row_cannot_express_raise(override, key):
    base = INP.dme_notebook_input()
    members = ENS.expand_members(base, [{}, override])
    with pytest.raises(ValueError) as e:
        ENS.member_parameters(base, members)
    assert "member 1" in str(e.value) and key in str(e.value)


def test_rebuilt_but_identical_lambdas_are_accepted():
    """A member built by calling the same input factory again carries NEW lambda objects with the same code."""
    base = INP.dme_notebook_input()
    other = INP.dme_notebook_input()
    other["reaction-rates"] = INP.dme_kinetics(900.0)
    assert ENS.member_parameters(base, [base, other]) == ["CaBeDe"]


def test_parametrised_trace_is_bit_identical_to_the_members_own_lambdas():
    rr = INP.dme_kinetics(1171.2)
    low = lowering.trace(rr["VARS"], rr["RATES"], 6, params=["CaBeDe"])
    assert low.uses("u0")
    rng = np.random.default_rng(11)
    for v in CABEDE:
        own = INP.dme_kinetics(v)
        for _ in range(10):
            T, P = rng.uniform(450, 900), rng.uniform(1e5, 6e6)
            x = rng.random(6) + 0.01
            x /= x.sum()
            C = x*P/(8.314472*T)
            want = O.reaction_rate_exe((T, P, x, C), own["VARS"], own["RATES"])
            assert low.evaluate(T, P, x, C, U=[v]) == [float(w) for w in want]
    with pytest.raises(lowering.LoweringError):
        lowering.trace(rr["VARS"], rr["RATES"], 6, params=["K1"])          # a lambda is not a scalar constant
    # the gradient treats a parameter like the pressure: frozen
    grad = low.optimize().gradient()
    assert all(not w.startswith("u") for w in grad.wrt)


def test_generated_source_with_a_parameter_column_vs_oracle_rhs():
    """RHS of a 3-member CaBeDe sweep through the generated translation unit (host emulation) against the
    oracle evaluating every member with ITS OWN lambdas, at the reference-generated transient states of G2."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g2_rhs.npz"))
    zNo = 20
    Y = g["dme_nb_%d_y" % zNo]
    base = INP.dme_notebook_input()
    members = _sweep(base)
    params = ENS.member_parameters(base, members)
    assert params == ["CaBeDe"]
    mech = plan.Mechanism(base, params=params)
    assert mech.NU == 1 and mech.row_width == 16 + 6 + 1 and "#define RMT_NU 1" in mech.prelude()
    rows = np.array([plan.member_constants(mi, mech, zNo)[1] for mi in members])
    assert list(rows[:, -1]) == CABEDE
    src = mech.source(hipbind.kernel_template())
    assert "U[0]" in src
    emu = HostEmu(src, tag="cabede")
    for k in (1, 2):                                           # two mid-transient states
        out, flags = emu.rhs(np.tile(Y[k], (3, 1)), rows, zNo)
        assert not flags.any()
        for e, mi in enumerate(members):
            want = O.make_rhs_vec(O.setup_n2(mi, zNo))(0.0, Y[k])
            assert rowwise_err(out[e], want, mech.V) < 1e-12, (k, e)
        assert rowwise_err(out[1], out[0], mech.V) > 1e-3      # the members really differ
    # and the literal kernel of member 1 alone gives the same numbers to rounding
    m1 = plan.Mechanism(members[1])
    lit, _ = HostEmu(m1.source(hipbind.kernel_template()), tag="cabede_lit").rhs(
        Y[1], plan.member_constants(members[1], m1, zNo)[1], zNo)
    par, _ = emu.rhs(np.tile(Y[1], (3, 1)), rows, zNo)
    assert rowwise_err(par[1], lit[0], mech.V) < 1e-13


def _run(mi):
    import emu_device
    from rmt_app_amd import n2, rmtExe
    real_device, n2.N2Device = n2.N2Device, emu_device.EmuDevice
    try:
        return rmtExe(mi)["resModel"]
    finally:
        n2.N2Device = real_device


def test_rmtexe_cabede_sweep_equals_single_runs():
    """rmtExe (host-emulation stand-in for the device): a 3-member CaBeDe sweep in ONE launch equals three single
    runs with the same parameter list bit for bit, the literal-kernel single runs to rounding, and every member's
    result really depends on its own CaBeDe."""
    def base_input():
        mi = INP.dme_notebook_input(ivp="hip-rk4", period=2e-4)
        mi["solver-config"].update({"quiet": True, "dt": 2e-6, "zNo": 32, "tNo": 2})
        return mi
    mi = base_input()
    mi["solver-config"]["ensemble"] = [{"reaction-rates": {"VARS": {"CaBeDe": v}}} for v in CABEDE]
    sweep = _run(mi)["ensemble"]
    assert len(sweep) == 3
    for e, v in enumerate(CABEDE):
        single = base_input()
        single["reaction-rates"]["VARS"]["CaBeDe"] = v
        literal = _run(single)["dataPack"]
        single["solver-config"]["vars-as-parameters"] = ["CaBeDe"]
        param = _run(single)["dataPack"]
        for k in range(2):
            np.testing.assert_array_equal(sweep[e]["dataPack"][k]["dataYs"], param[k]["dataYs"])
            assert np.max(np.abs(sweep[e]["dataPack"][k]["dataYs"] - literal[k]["dataYs"])
                          / np.abs(literal[k]["dataYs"])) < 1e-12
    t_out = [m["dataPack"][1]["dataYs"][6, 3] for m in sweep]
    d = np.diff([t_out[1], t_out[0], t_out[2]])                # ordered by CaBeDe: 500, 1171.2, 1800
    assert np.all(d < 0) or np.all(d > 0)                      # monotone in the catalyst density
    # a member with another reaction set is refused, not run with the base kinetics
    mi["solver-config"]["ensemble"] = [{}, {"reactions": {"R1": "CO2 + 3H2 <=> CH3OH + H2O"}}]
    with pytest.raises(ValueError) as e:
        _run(mi)
    assert "member 1" in str(e.value)
