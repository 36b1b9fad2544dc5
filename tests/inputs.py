"""Model inputs used by the parity tests, the golden-vector generator and bench.py.

These are *data*: the ``modelInput`` dicts a PyREMOT user writes (schema: SURVEY.md
Appendix B).  Sources of the numbers (reference files, read as text):

* ``dme_notebook_input``  – PyREMOT/jupyter-notebook/PyREMOT TEST2.ipynb (model N2; the README
  dynamic example; canonical config-2 input).  ``n1_notebook_input`` is TEST1.ipynb (model N1).
* ``dme_script_input``    – PyREMOT/tests/test_rmt_N1_DME.py:25-274 (model "N2" despite the
  file name; feed built through float32 mole fractions + 7-decimal rounding,
  PyREMOT/data/initData.py:11-69, reactor constants PyREMOT/data/inputDataReactor.py:9-39).
* ``ch4_input``           – PyREMOT/tests/test_rmt_N2_CH4.py:21-251 (3 species / 1 reaction,
  adiabatic, rate uses SpCoi), with the missing ``display-result`` key added.
* ``syn12_input``         – this repo's own 12-species / 8-reaction synthetic mechanism
  (BASELINE.json configs[4]); power-law Arrhenius lambdas built in a loop (closures).

Nothing here imports the reference.
"""
import math

import numpy as np

R_CONST = 8.314472  # PyREMOT/core/constants.py:8

DME_COMPONENTS = ["H2", "CO2", "H2O", "CO", "CH3OH", "DME"]

DME_REACTIONS_SPACED = {
    "R1": "CO2 + 3H2 <=> CH3OH + H2O",
    "R2": "CO + H2O <=> H2 + CO2",
    "R3": "2CH3OH <=> DME + H2O",
}
DME_REACTIONS_COMPACT = {
    "R1": "CO2+3H2<=>CH3OH+H2O",
    "R2": "CO+H2O<=>H2+CO2",
    "R3": "2CH3OH<=>DME+H2O",
}


def dme_kinetics(CaBeDe):
    """The DME VARS/RATES dicts (same in TEST2.ipynb and test_rmt_N1_DME.py:129-200)."""
    varis0 = {
        "CaBeDe": CaBeDe,
        "RT": lambda x: x['R_CONST']*x['T'],
        "K1": lambda x: 35.45*math.exp(-1.7069e4/x['RT']),
        "K2": lambda x: 7.3976*math.exp(-2.0436e4/x['RT']),
        "K3": lambda x: 8.2894e4*math.exp(-5.2940e4/x['RT']),
        "KH2": lambda x: 0.249*math.exp(3.4394e4/x['RT']),
        "KCO2": lambda x: 1.02e-7*math.exp(6.74e4/x['RT']),
        "KCO": lambda x: 7.99e-7*math.exp(5.81e4/x['RT']),
        "Ln_KP1": lambda x: 4213/x['T'] - 5.752 *
        math.log(x['T']) - 1.707e-3*x['T'] + 2.682e-6 *
        (math.pow(x['T'], 2)) - 7.232e-10*(math.pow(x['T'], 3)) + 17.6,
        "KP1": lambda x: math.exp(x['Ln_KP1']),
        "log_KP2": lambda x: 2167/x['T'] - 0.5194 *
        math.log10(x['T']) + 1.037e-3*x['T'] - 2.331e-7 *
        (math.pow(x['T'], 2)) - 1.2777,
        "KP2": lambda x: math.pow(10, x['log_KP2']),
        "Ln_KP3": lambda x: 4019/x['T'] + 3.707 *
        math.log(x['T']) - 2.783e-3*x['T'] + 3.8e-7 *
        (math.pow(x['T'], 2)) - 6.56e-4/(math.pow(x['T'], 3)) - 26.64,
        "KP3": lambda x: math.exp(x['Ln_KP3']),
        "yi_H2": lambda x: x['MoFri'][0],
        "yi_CO2": lambda x: x['MoFri'][1],
        "yi_H2O": lambda x: x['MoFri'][2],
        "yi_CO": lambda x: x['MoFri'][3],
        "yi_CH3OH": lambda x: x['MoFri'][4],
        "yi_DME": lambda x: x['MoFri'][5],
        "PH2": lambda x: x['P']*(x['yi_H2'])*1e-5,
        "PCO2": lambda x: x['P']*(x['yi_CO2'])*1e-5,
        "PH2O": lambda x: x['P']*(x['yi_H2O'])*1e-5,
        "PCO": lambda x: x['P']*(x['yi_CO'])*1e-5,
        "PCH3OH": lambda x: x['P']*(x['yi_CH3OH'])*1e-5,
        "PCH3OCH3": lambda x: x['P']*(x['yi_DME'])*1e-5,
        "ra1": lambda x: x['PCO2']*x['PH2'],
        "ra2": lambda x: 1 + (x['KCO2']*x['PCO2']) + (x['KCO']*x['PCO']) + math.sqrt(x['KH2']*x['PH2']),
        "ra3": lambda x: (1/x['KP1'])*((x['PH2O']*x['PCH3OH'])/(x['PCO2']*(math.pow(x['PH2'], 3)))),
        "ra4": lambda x: x['PH2O'] - (1/x['KP2'])*((x['PCO2']*x['PH2'])/x['PCO']),
        "ra5": lambda x: (math.pow(x['PCH3OH'], 2)/x['PH2O'])-(x['PCH3OCH3']/x['KP3']),
    }
    rates0 = {
        "r1": lambda x: 1000*x['K1']*(x['ra1']/(math.pow(x['ra2'], 3)))*(1-x['ra3'])*x['CaBeDe'],
        "r2": lambda x: 1000*x['K2']*(1/x['ra2'])*x['ra4']*x['CaBeDe'],
        "r3": lambda x: 1000*x['K3']*x['ra5']*x['CaBeDe'],
    }
    return {"VARS": varis0, "RATES": rates0}


def dme_notebook_input(model="N2", ivp="default", process_type="non-iso-thermal", period=0.5):
    """TEST2.ipynb (N2) / TEST1.ipynb (N1) literal input."""
    CaBeDe = 1171.2
    mi = {
        "model": model,
        "operating-conditions": {
            "pressure": 5000000,
            "temperature": 523,
            "process-type": process_type,
            "period": period,
        },
        "feed": {
            "volumetric-flowrate": 0.000228,
            "concentration": [574.8978, 287.4489, 1.15e-02, 287.4489, 1.15e-02, 1.15e-02],
            "components": {"shell": list(DME_COMPONENTS)},
        },
        "reactions": dict(DME_REACTIONS_COMPACT),
        "reaction-rates": dme_kinetics(CaBeDe),
        "external-heat": {"OvHeTrCo": 50, "MeTe": 523},
        "reactor": {
            "ReInDi": 0.0381, "ReLe": 1, "PaDi": 0.002, "BeVoFr": 0.39,
            "CaBeDe": CaBeDe, "CaDe": 1920, "CaSpHeCa": 960,
        },
        "solver-config": {"ivp": ivp, "display-result": "False"},
    }
    if model == "N1":
        del mi["operating-conditions"]["period"]
    return mi


def n1_notebook_input(ivp="default"):
    return dme_notebook_input(model="N1", ivp=ivp)


def _feed_concentration_rounded(MoFri, P, T):
    """calConcentration (PyREMOT/data/initData.py:42-69): kmol/m^3 rounded to 7 decimals."""
    Ci = np.zeros(len(MoFri))
    for i in range(len(MoFri)):
        Ci[i] = (P/(R_CONST*T))*MoFri[i]/1000
    return np.round(Ci, 7)


def dme_script_input(ivp="default", process_type="non-iso-thermal", period=0.5):
    """PyREMOT/tests/test_rmt_N1_DME.py input (model N2)."""
    P = 5*1e6
    T = 523
    # setFeedMoleFraction(1, 0.5) -> float32 array (initData.py:11-39)
    y0_H2O = y0_CH3OH = y0_DME = 0.00001
    tmf0 = 1 - (y0_H2O + y0_CH3OH + y0_DME)
    COx = tmf0/(1 + 1)
    y0_H2 = 1*COx
    y0_CO2 = 0.5*COx
    y0_CO = COx - y0_CO2
    feedMoFr = np.array([y0_H2, y0_CO2, y0_H2O, y0_CO, y0_CH3OH, y0_DME], dtype=np.float32)
    ct0 = _feed_concentration_rounded(feedMoFr, P, T)
    ct0_CONV = 1e3*ct0
    # reactor constants (inputDataReactor.py)
    rea_D, rea_L, bed_por = 0.0381, 1, 0.39
    cat_d, cat_rho, cat_Cp = 0.002, 1982, 960
    bulk_rho = cat_rho*(1 - bed_por)
    SuGaVe = 0.2
    InGaVe = SuGaVe/bed_por
    rea_CSA = bed_por*(math.pi*(rea_D**2)/4)
    VoFlRa = InGaVe*rea_CSA
    U = 100
    return {
        "model": "N2",
        "operating-conditions": {
            "pressure": P, "temperature": T, "period": period, "process-type": process_type,
        },
        "feed": {
            "volumetric-flowrate": VoFlRa,
            "concentration": ct0_CONV,
            "components": {"shell": list(DME_COMPONENTS)},
        },
        "reactions": dict(DME_REACTIONS_SPACED),
        "reaction-rates": dme_kinetics(bulk_rho),
        "external-heat": {"OvHeTrCo": U, "EfHeTrAr": 4/rea_D, "MeTe": T - 1},
        "reactor": {
            "ReInDi": rea_D, "ReLe": rea_L, "PaDi": cat_d, "BeVoFr": bed_por,
            "CaBeDe": bulk_rho, "CaDe": cat_rho, "CaSpHeCa": cat_Cp/1000,
        },
        "solver-config": {"ivp": ivp, "display-result": "False"},
    }



def m2_dme_input(ivp="LSODA", period=10):
    """PyREMOT/tests/test_rmt_DME.py input (model M2, the dimensional dynamic model:
    concentrations in kmol/m^3, feed viscosity given, catalyst thermal mass in the energy balance)."""
    P = 5*1e6
    T = 523
    y0_H2O = y0_CH3OH = y0_DME = 0.00001
    tmf0 = 1 - (y0_H2O + y0_CH3OH + y0_DME)
    COx = tmf0/(1 + 1)
    y0_H2 = 1*COx
    y0_CO2 = 0.5*COx
    y0_CO = COx - y0_CO2
    feedMoFr = np.array([y0_H2, y0_CO2, y0_H2O, y0_CO, y0_CH3OH, y0_DME], dtype=np.float32)
    ct0 = _feed_concentration_rounded(feedMoFr, P, T)          # [kmol/m^3]
    rea_D, rea_L, bed_por = 0.0381, 1, 0.39
    cat_d, cat_rho, cat_Cp = 0.002, 1982, 960
    bulk_rho = cat_rho*(1 - bed_por)
    SuGaVe = 0.2
    InGaVe = SuGaVe/bed_por
    rea_CSA = bed_por*(math.pi*(rea_D**2)/4)
    VoFlRa = InGaVe*rea_CSA
    return {
        "model": "M2",
        "operating-conditions": {"pressure": P, "temperature": T, "period": period},
        "feed": {
            "mole-fraction": 0, "molar-flowrate": 0, "molar-flux": 0,
            "volumetric-flowrate": VoFlRa,
            "concentration": ct0,
            "mixture-viscosity": 1e-5,
            "components": {"shell": list(DME_COMPONENTS), "tube": [], "medium": []},
        },
        "reactions": dict(DME_REACTIONS_SPACED),
        "reaction-rates": dme_kinetics(bulk_rho),
        "external-heat": {"OvHeTrCo": 50, "EfHeTrAr": 4/rea_D, "MeTe": 523},
        "reactor": {
            "ReInDi": rea_D, "ReLe": rea_L, "PaDi": cat_d, "BeVoFr": bed_por,
            "CaBeDe": bulk_rho, "CaDe": cat_rho, "CaSpHeCa": cat_Cp/1000,
        },
        "solver-config": {"ivp": ivp},
    }

def ch4_input(ivp="default", period=10):
    """PyREMOT/tests/test_rmt_N2_CH4.py input (3 species, 1 reaction, adiabatic)."""
    P = 3*1e5
    T = 973
    bed_por = 0.39
    cat_rho = 1982
    bulk_rho = cat_rho*(1 - bed_por)
    rea_dia = 0.007
    MoFri0 = np.array([1 - (0.05 + 0.05), 0.05, 0.05])
    ct0 = _feed_concentration_rounded(MoFri0, P, T)
    SuGaVe = 0.01
    InGaVe = SuGaVe/bed_por
    rea_CSA = bed_por*(math.pi*(rea_dia**2)/4)
    VoFlRa = InGaVe*rea_CSA
    varis0 = {
        "k0": 0.0072*1e-1,
        "y_CH4": lambda x: x['MoFri'][0],
        "C_CH4": lambda x: x['SpCoi'][0],
    }
    rates0 = {"r1": lambda x: x['k0']*(x['C_CH4']**2)}
    return {
        "model": "N2",
        "operating-conditions": {
            "pressure": P, "temperature": T, "period": period,
            "process-type": "non-iso-thermal",
        },
        "feed": {
            "volumetric-flowrate": VoFlRa,
            "concentration": 1e3*ct0,
            "components": {"shell": ["CH4", "C2H4", "H2"], "tube": [], "medium": []},
        },
        "reactions": {"R1": "2CH4 <=> C2H4 + 2H2"},
        "reaction-rates": {"VARS": varis0, "RATES": rates0},
        "external-heat": {"OvHeTrCo": 50, "EfHeTrAr": 4/rea_dia, "MeTe": 0},
        "reactor": {
            "ReInDi": rea_dia, "ReLe": 1, "PaDi": 0.002, "BeVoFr": bed_por,
            "CaBeDe": bulk_rho, "CaDe": cat_rho, "CaSpHeCa": 960/1000,
        },
        "solver-config": {"ivp": ivp, "display-result": "False"},
    }


SYN12_COMPONENTS = ["CO2", "H2", "CH3OH", "H2O", "CO", "DME", "N2", "CH4", "C2H4", "C3H6",
                    "C3H8", "C4H10"]  # = componentSymbolList order (data/componentData.py:435)

SYN12_REACTIONS = {
    "R1": "CO2 + 3H2 <=> CH3OH + H2O",
    "R2": "CO + H2O <=> H2 + CO2",
    "R3": "2CH3OH <=> DME + H2O",
    "R4": "CO + 3H2 <=> CH4 + H2O",
    "R5": "2CH4 <=> C2H4 + 2H2",
    "R6": "C3H8 <=> C3H6 + H2",
    "R7": "C4H10 <=> C3H6 + CH4",
    "R8": "C2H4 + CH4 <=> C3H8",
}


def _parse_side(side):
    import re
    out = []
    for coef, sym in re.findall(r"([0-9.]*)([a-zA-Z0-9.]+)", side.replace(" ", "")):
        out.append((sym, float(coef) if coef else 1.0))
    return out


def syn12_kinetics(seed=20260410):
    """8 reversible power-law Arrhenius rates; constants from default_rng(seed).

    r_k = k0_k*exp(-Ea_k/(R T)) * ( prod_reactants p_i^nu - (1/Keq_k) prod_products p_j^nu ),
    p in bar, result mol/m^3/s.  Lambdas are created in a loop and close over Python floats.
    """
    rng = np.random.default_rng(seed)
    varis0 = {"RT": lambda x: x['R_CONST']*x['T']}
    for i, s in enumerate(SYN12_COMPONENTS):
        varis0["p_" + s] = (lambda idx: (lambda x: x['P']*x['MoFri'][idx]*1e-5))(i)
    rates0 = {}
    for k, (name, expr) in enumerate(SYN12_REACTIONS.items()):
        lhs, rhs = expr.replace("<", "").replace(">", "").split("=")
        reac, prod = _parse_side(lhs), _parse_side(rhs)
        k0 = float(10.0**rng.uniform(-2.0, -0.5))
        Ea = float(rng.uniform(2.0e4, 4.5e4))
        Keq = float(10.0**rng.uniform(-1.0, 1.0))
        varis0["k_" + name] = (lambda k0=k0, Ea=Ea: (lambda x: k0*math.exp(-Ea/x['RT'])))()

        def make_rate(name=name, reac=reac, prod=prod, Keq=Keq):
            def rate(x):
                f = 1.0
                for s, n in reac:
                    f = f*x["p_" + s]**n
                b = 1.0
                for s, n in prod:
                    b = b*x["p_" + s]**n
                return 1000*x["k_" + name]*(f - b/Keq)
            return rate
        rates0["r" + str(k + 1)] = make_rate()
    return {"VARS": varis0, "RATES": rates0}


def ch4_arrhenius_input(ivp="default", period=10, composition_exp=True):
    """The CH4 case with a rate law that has what the caching steppers tell apart: an Arrhenius constant (exponent linear
    in 1/T), an equilibrium-type constant whose exponent is a polynomial in T, 1/T and log T, an exponent that does NOT
    decompose (exp of sqrt T) and - with composition_exp - an exp of the composition, which is not a function of the
    temperature at all (it needs the big exp table whatever the path)."""
    import math
    mi = ch4_input(ivp, period)
    damp = (lambda x: math.exp(-3.0*x['y_CH4'])) if composition_exp else (lambda x: 1.0)
    mi["reaction-rates"]["RATES"] = {
        "r1": lambda x: (x['k0']*math.exp(9000.0/973.0 - 9000.0/x['T'])*(x['C_CH4']**2)*damp(x)
                         / (1.0 + 0.01*math.exp(2.0 - 1500.0/x['T'] + 0.3*math.log(x['T']) - 1e-3*x['T'] + 2e-7*x['T']**2))
                         * (1.0 + 1e-3*math.exp(-math.sqrt(x['T'])/40.0)))}
    return mi


def syn12_input(ivp="default", period=0.5):
    P = 2.0e6
    T = 600
    y0 = np.array([0.10, 0.30, 0.02, 0.02, 0.15, 0.02, 0.20, 0.08, 0.03, 0.03, 0.03, 0.02])
    y0 = y0/y0.sum()
    conc = (P/(R_CONST*T))*y0
    return {
        "model": "N2",
        "operating-conditions": {
            "pressure": P, "temperature": T, "period": period,
            "process-type": "non-iso-thermal",
        },
        "feed": {
            "volumetric-flowrate": 0.000228,
            "concentration": conc,
            "components": {"shell": list(SYN12_COMPONENTS)},
        },
        "reactions": dict(SYN12_REACTIONS),
        "reaction-rates": syn12_kinetics(),
        "external-heat": {"OvHeTrCo": 50, "MeTe": 590},
        "reactor": {
            "ReInDi": 0.0381, "ReLe": 1, "PaDi": 0.002, "BeVoFr": 0.39,
            "CaBeDe": 1171.2, "CaDe": 1920, "CaSpHeCa": 960,
        },
        "solver-config": {"ivp": ivp, "display-result": "False"},
    }


ALL_N2_INPUTS = {
    "dme_nb": dme_notebook_input,
    "dme_script": dme_script_input,
    "ch4": ch4_input,
    "syn12": syn12_input,
}
