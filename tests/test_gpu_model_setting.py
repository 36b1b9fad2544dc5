"""GPU: MODEL_SETTING['GaMaCoTe0'] != "MAX" (PyREMOT/docs/modelSetting.py:10-18).  What the reference does under that
setting was recorded from the reference itself (tools/make_golden.py setting -> golden G11): its N2 run raises
numpy's ValueError on the first RHS evaluation (pbHomoReactor.py:3901-3904 assigns an array to an element), model
N1 runs with per-species scaling (:2819-2821, 3159-3162; solResultAnalysis.py:226-231), M2 never reads the setting.
The device build mirrors all three."""
import json
import os

import numpy as np
import pytest

import inputs as INP
from oracle import n2_oracle as O
from rmt_app_amd import MODEL_SETTING, rmtExe

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture
def gamacote_fix():
    old = MODEL_SETTING["GaMaCoTe0"]
    MODEL_SETTING["GaMaCoTe0"] = "FIX"
    try:
        yield
    finally:
        MODEL_SETTING["GaMaCoTe0"] = old


def test_n1_profile_under_per_species_scaling(gamacote_fix):
    from scipy.integrate import solve_ivp
    g = np.load(os.path.join(G, "g11_n1_fix.npz"))
    mi = INP.n1_notebook_input()
    d = rmtExe(mi)["resModel"][0]
    assert d["dataYs"].shape == g["dataYs"].shape == (8, 101)
    pr = O.setup_n1(mi, gamacote="FIX")
    tight = solve_ivp(lambda t, y: O.n1_rhs(t, y, pr), [0, 1], pr["IV"], method="LSODA", rtol=1e-11, atol=1e-13,
                      t_eval=np.linspace(0, 1, 101))
    conc = tight.y[:6]*pr["SpCoi0_Set"].reshape(-1, 1)
    want = np.concatenate([conc/conc.sum(0), (tight.y[6]*pr["Pf"]).reshape(1, -1),
                           (tight.y[7]*pr["Tf"] + pr["Tf"]).reshape(1, -1)])
    assert np.max(np.abs(d["dataYs"] - want)/np.abs(want)) < 1e-6
    # the reference's own run (default LSODA tolerances, ~1e-4 accurate) and its packed concentrations
    assert np.max(np.abs(d["dataYs"] - g["dataYs"])/np.abs(g["dataYs"])) < 5e-3
    assert np.max(np.abs(d["dataYCons2"] - g["dataYCons2"])/np.maximum(np.abs(g["dataYCons2"]), 1e-12)) < 5e-3
    # and it is not the "MAX" profile
    g6 = np.load(os.path.join(G, "g6_n1.npz"))
    assert np.max(np.abs(d["dataYs"] - g6["dataYs"])/np.abs(g6["dataYs"])) > 1e-2


def test_n2_raises_and_m2_runs_under_the_setting(gamacote_fix):
    rec = json.load(open(os.path.join(G, "g11_model_setting.json")))
    with pytest.raises(ValueError) as e:
        rmtExe(INP.dme_notebook_input(ivp="hip-rk4", period=1e-4))
    assert str(e.value) == rec["N2"]["message"]
    m2 = INP.m2_dme_input(ivp="hip-rk4", period=1e-3)
    m2["solver-config"].update({"quiet": True, "dt": 2e-6, "zNo": 32, "tNo": 1})
    with_setting = rmtExe(m2)["resModel"]["dataPack"][0]["dataYs"]
    MODEL_SETTING["GaMaCoTe0"] = "MAX"
    np.testing.assert_array_equal(with_setting, rmtExe(m2)["resModel"]["dataPack"][0]["dataYs"])
