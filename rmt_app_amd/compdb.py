"""Component property table of the device build (12 species).

Values are those of the reference's database (PyREMOT/data/componentData.py:11-86 MW and dHf25,
:119-405 Cp(T) polynomials; PyREMOT/data/dataGasViscosity.py:9-141 viscosity correlations),
stored here as numeric coefficient rows - the form the kernel generator needs - instead of the
reference's expression strings that are eval()-ed on every call (PyREMOT/docs/rmtThermo.py:37).

Cp(T) = a + b*T + c*T^2 + d*T^3           [kJ/kmol/K = J/mol/K]
visc eq.1: mu = A*1e-6*T^B/(1 + C/T + D/T^2)   [Pa.s]   (PyREMOT/docs/gasTransPor.py:137-154)
visc eq.2: mu = A*T^B/(1 + C/T)                [Pa.s]   (DME expression string, :157-168)
"""
from collections import namedtuple

Component = namedtuple("Component", "symbol MW cp dHf25 vis_eq vis")

_ROWS = [
    #  symbol    MW      Cp a        b           c            d            dHf25     eq  viscosity params
    ("CO2",   44.01, (22.243,  5.98E-02,  -3.50E-05,   7.46E-09),  -393.51, 1, (4.719875, 0.373279, 512.686300, -6119.961)),
    ("H2",     2.0,  (26.879,  4.35E-03,  -3.30E-07,   0.0),          0.0,  1, (0.169104, 0.692485, -7.634394, 467.120)),
    ("CH3OH", 32.04, (19.038,  9.15E-02,  -1.22E-05,  -8.03E-09),  -200.7,  1, (0.477915, 0.641076, 284.838034, -3230.713)),
    ("H2O",   18.01, (29.163,  1.45E-02,  -2.02E-06,   0.0),       -241.820, 1, (0.501246, 0.709247, 869.465599, -90063.891)),
    ("CO",    28.01, (27.113,  6.55E-03,  -1.00E-06,   0.0),       -110.53, 1, (0.734306, 0.588574, 52.318660, 1018.822)),
    ("DME",   46.07, (19.8,    0.17,      -5.66e-5,    0.0),       -184.1,  2, (2.68e-7, 0.3975, 534.0)),
    ("N2",    28.0,  (28.883, -1.57E-03,   8.08E-06,  -2.87E-09),     0.0,  1, (0.847662, 0.574033, 75.437536, 56.771)),
    ("CH4",   16.04, (19.875,  5.021E-02,  1.268E-05, -11.004E-09), -74.90, 1, (1.119178, 0.493234, 214.627200, -3952.087)),
    ("C2H4",  28.05, (3.950,   15.628E-02, -8.339E-05, 17.657E-09),  52.32, 1, (1.503552, 0.456140, 288.342422, 73.362)),
    ("C3H6",  42.08, (3.151,   23.812E-02, -12.176E-05, 24.603E-09), 20.4,  1, (0.876767, 0.520871, 293.618650, -182.857)),
    ("C3H8",  44.1,  (-4.042,  30.456E-02, -15.711E-05, 31.716E-09), -103.9, 1, (0.173966, 0.734798, 143.207060, -7147.859)),
    ("C4H10", 58.12, (-7.908,  41.573E-02, -22.992E-05, 49.875E-09), -126.2, 1, (0.075828, 0.837082, 67618677.0, -2141.762)),
]

COMPONENTS = {r[0]: Component(*r) for r in _ROWS}
componentSymbolList = tuple(r[0] for r in _ROWS)   # same name as PyREMOT/data/componentData.py:435


def cp_value(sym, T):
    """Cp_i(T) with the reference's left-to-right evaluation of its expression string; a missing
    cubic (quadratic) term is really absent there, so it is skipped rather than added as 0."""
    a, b, c, d = COMPONENTS[sym].cp
    v = a + b*T + c*(T**2)
    if d != 0.0:
        v = v + d*(T**3)
    return v


def viscosity(sym, T):
    c = COMPONENTS[sym]
    if c.vis_eq == 1:
        A, B, C, D = c.vis
        return A*1e-6*(T**B)/(1 + C*(1/T) + D*(T**-2))
    A, B, C = c.vis
    return A*(T**B)/(1 + (C/T))
