"""Host front-end of the N2 path: modelInput -> mechanism tables + packed per-reactor constants
+ generated kernel source.

Mirrors the *setup* half of PackedBedHomoReactorClass.runN2 (PyREMOT/docs/pbHomoReactor.py:3334-3580)
and of rmtCoreClass.initReaction / initComponentData (PyREMOT/docs/rmtCore.py:129-183), but instead
of building nested dicts that the RHS re-reads on every call (:3741-3825) it produces

  * ``Mechanism``  - everything that is identical for all members of an ensemble and becomes
                     compile-time constants of the generated kernel (species, stoichiometry, MW,
                     Cp polynomials, heats of reaction, the lowered rate lambdas);
  * member rows    - 16+S(+NU) doubles per reactor (layout: csrc/kernels/00_config_math.inc ``M_*``) holding the
                     operating-point dependent scalars, pre-combined so the kernel does no
                     redundant work (e.g. the Ergun march coefficient of SURVEY.md section 5), followed by
                     the NU scalar ``reaction-rates.VARS`` entries that differ between the members of an
                     ensemble (``Mechanism(params=...)``): the reference copies the user's VARS constants into
                     the namespace on every call (rmtReaction.py:44-51), so a sweep over a catalyst density or
                     an Arrhenius constant is an ordinary sweep there - here those entries stay symbolic in the
                     lowering and are read from the member row.
"""
import hashlib
import math
import re

import numpy as np

from . import compdb
from .lowering import trace
from .settings import MODEL_SETTING, PROCESS_SETTING

R_CONST = 8.314472          # PyREMOT/core/constants.py:8
PI_CONST = math.pi          # :14
Tref = 273.15 + 25.00       # :23

MEMBER_FIXED = 16
# index of each scalar in a member row (must match csrc/kernels/00_config_math.inc)
MEMBER_FIELDS = {
    "CMAX": 0, "TF": 1, "P0": 2, "THETA_IN": 3, "ALPHA_K": 4, "BETA": 5, "RHO_K": 6,
    "INV_CP0": 7, "F1": 8, "FT": 9, "INV_DZ": 10, "INV_MACOTE": 11, "INV_HECOTE": 12,
    "UA": 13, "TM": 14, "CIN": 16,
}

_SPECIES_RE = re.compile(r"([0-9.]*)([a-zA-Z0-9.]+)")


def parse_reaction(expr):
    """'CO2 + 3H2 <=> CH3OH + H2O' -> ([(sym, -coeff)...], [(sym, +coeff)...]).
    Grammar of buildReactionCoefficient (PyREMOT/docs/rmtUtility.py:171-220): strip '<', '>' and
    blanks, split on '=', then an optional numeric prefix and a symbol per term."""
    sides = expr.replace("<", "").replace(">", "").replace(" ", "").split("=")
    if len(sides) < 2:
        raise ValueError("reaction %r has no '=' separator" % (expr,))
    reac = [(s, -1.0*float(c) if c else -1.0) for c, s in _SPECIES_RE.findall(sides[0])]
    prod = [(s, float(c) if c else 1.0) for c, s in _SPECIES_RE.findall(sides[1])]
    return reac, prod


def reaction_tables(reactionDict):
    """reactionListSorted / reactionStochCoeff in the reference's shapes (rmtUtility.py:171-249)."""
    srt, vec = [], []
    for expr in reactionDict.values():
        reac, prod = parse_reaction(expr)
        srt.append({"reactants": [{"symbol": s, "coeff": c} for s, c in reac],
                    "products": [{"symbol": s, "coeff": c} for s, c in prod]})
        vec.append([[s, float(c)] for s, c in reac + prod])
    return srt, vec


def build_component_list(componentDataDict):
    """buildComponentList (rmtUtility.py:312-340): shell + tube + medium, de-duplicated."""
    out = []
    for key in ("shell", "tube", "medium"):
        part = componentDataDict.get(key)
        if part:
            out.extend(part)
    return list(dict.fromkeys(out))


def wilke(visc, x, MW):
    """Wilke mixing rule, calMixturePropertyM1 (PyREMOT/docs/gasTransPor.py:229-274)."""
    n = len(visc)
    phi = np.ones((n, n))
    for i in range(n):
        for j in range(i + 1, n):
            A = 1 + math.sqrt(visc[i]/visc[j])*((MW[j]/MW[i])**(1/4))
            phi[i, j] = (A**2)/math.sqrt(8*(1 + (MW[i]/MW[j])))
            phi[j, i] = (visc[j]/visc[i])*(MW[i]/MW[j])*phi[i, j]
    tot = 0.0
    for i in range(n):
        tot += visc[i]*x[i]/np.sum(x*phi[i, :])
    return tot


# RMT_KCACHE_GEN: what a caching stepper does with constants whose exponent is not linear in 1/T (equilibrium constants):
# "0" evaluates them in full, "1" caches value and exponent (two slots), "2" caches the value and forms the exponent's
# change from the differences of its basis functions T^n, log T (one slot; lowering.Lowered.kcache_plan)
KCACHE_GEN = {"0": False, "1": True, "2": "basis"}


class Mechanism:
    """Ensemble-invariant part of a model: becomes literals in the generated kernel."""

    optimize = True     # class-wide switch: emit the strength-reduced kinetics DAG

    def __init__(self, modelInput, params=()):
        """``params``: names of scalar ``reaction-rates.VARS`` entries that are per-reactor parameters (columns
        16+S.. of the member row) instead of literals of the generated kernel."""
        mi = modelInput
        self.params = tuple(params)
        self.NU = len(self.params)
        self.compList = list(mi['feed']['components']['shell'])
        for s in build_component_list(mi['feed']['components']):
            if s not in compdb.componentSymbolList:
                raise Exception("Component database is not up to date!")       # rmt.py:55-57
        self.S = len(self.compList)
        # "M2" (dimensional dynamic model, pbReactor.py:552-1165) has no process-type key: it is
        # always non-iso-thermal; everything else on this path is the dimensionless N2/N1 pair
        self.model = "M2" if mi.get('model') == "M2" else "N2"
        if self.model == "M2":
            self.processType = PROCESS_SETTING['NON-ISO-THER']
        else:
            self.processType = mi['operating-conditions']['process-type']
        self.iso = self.processType == PROCESS_SETTING['ISO-THER']
        self.V = self.S if self.iso else self.S + 1
        self.reactionDict = dict(mi['reactions'])
        self.reactionListSorted, self.reactionStochCoeff = reaction_tables(self.reactionDict)
        self.R = len(self.reactionStochCoeff)
        self.MW = [compdb.COMPONENTS[s].MW for s in self.compList]
        # dense stoichiometric matrix: componentFormationRate (rmtReaction.py:64-97) sums every
        # occurrence of a species in a reaction; calEnthalpyChangeOfReaction (rmtThermo.py:258-312)
        # does the same with the Cp's, so one matrix serves both.
        self.nu = np.zeros((self.R, self.S))
        for k, rx in enumerate(self.reactionStochCoeff):
            for sym, c in rx:
                if sym in self.compList:
                    self.nu[k, self.compList.index(sym)] += c
        self.cp_coeff = np.array([compdb.COMPONENTS[s].cp for s in self.compList], dtype=float)
        self.cp_ref = np.array([compdb.cp_value(s, Tref) for s in self.compList])
        self.StHeRe25 = np.array([self._standard_heat(expr) for expr in self.reactionDict.values()])
        rr = mi['reaction-rates']
        self.lowered = trace(rr['VARS'], rr['RATES'], self.S, R_CONST, params=self.params)
        if len(self.lowered.outputs) != self.R:
            raise ValueError("%d rate expressions for %d reactions" % (len(self.lowered.outputs), self.R))

    @staticmethod
    def _standard_heat(expr):
        """calStandardEnthalpyOfReaction (rmtThermo.py:129-198) [kJ/kmol == J/mol]."""
        reac, prod = parse_reaction(expr)
        hr = np.sum(np.array([compdb.COMPONENTS[s].dHf25*(-c) for s, c in reac
                              if s in compdb.COMPONENTS]))
        hp = np.sum(np.array([compdb.COMPONENTS[s].dHf25*c for s, c in prod
                              if s in compdb.COMPONENTS]))
        return (hp - hr)*1000.00

    # ------------------------------------------------------------------ kernel source
    def prelude(self, fp32=False, block=1024, npt=1, lds_state=None, defines=None):
        def arr(vals):
            return "{" + ", ".join("real(%r)" % float(v) for v in vals) + "}"
        S, R = self.S, self.R
        lines = [
            "// generated by rmt_app_amd.plan.Mechanism.prelude - do not edit",
            "#define RMT_S %d" % S,
            "#define RMT_R %d" % R,
            "#define RMT_NU %d" % self.NU,
            "#define RMT_ISO %d" % (1 if self.iso else 0),
            "#define RMT_MODEL %d" % (2 if self.model == "M2" else 0),
            "#define RMT_FP32 %d" % (1 if fp32 else 0),
            "#define RMT_BLOCK %d" % block,
            "#define RMT_NPT %d" % npt,
            "#define RMT_LDS_STATE %d" % self.lds_state(fp32, block, npt, lds_state),
            "#define RMT_LDS_STATE_CHAIN %d" % self.lds_state(fp32, block, npt, lds_state, chained=True),
        ] + ([] if (block > 64 or "RMT_EXP_BITS" in (defines or {})) else [
            # one-wave workgroups (small meshes, big ensembles): the 16 KiB exp table would cap the CU
            # at 9 resident waves (measured -30 % at N=20, E=2048); they keep the 64-entry table
            "#define RMT_EXP_BITS 6"]) + ["#define %s %s" % (k, v) for k, v in sorted((defines or {}).items())] + [
            "typedef %s real;" % ("float" if fp32 else "double"),
            "__device__ static const real RMT_MW[RMT_S] = %s;" % arr(self.MW),
            "__device__ static const real RMT_DH25[RMT_R] = %s;" % arr(self.StHeRe25),
        ]

        def lincomb(coeffs, name):
            terms = []
            for idx, c in enumerate(coeffs):
                if c == 0:
                    continue
                mag = "%s[%d]" % (name, idx) if abs(c) == 1 else "real(%r) * %s[%d]" % (abs(float(c)), name, idx)
                terms.append((" - " if c < 0 else " + ") + mag)
            if not terms:
                return "real(0)"
            out = "".join(terms)
            return out[3:] if out.startswith(" + ") else "-" + out[3:]

        lines.append("__device__ __forceinline__ void rmt_species_source(const real* __restrict__ r, "
                     "real* __restrict__ s) {")
        for i in range(S):
            lines.append("    s[%d] = %s;" % (i, lincomb(self.nu[:, i], "r")))
        lines.append("}")
        lines.append("__device__ __forceinline__ void rmt_reaction_dcp(const real* __restrict__ c, "
                     "real* __restrict__ d) {")
        for k in range(R):
            lines.append("    d[%d] = %s;" % (k, lincomb(self.nu[k, :], "c")))
        lines.append("}")
        lines.append("__device__ __forceinline__ real rmt_cp_mean(const int i, const real T) {")
        # (Cp(Tref) + a + b T + c T^2 + d T^3)/2 (rmtThermo.py:52-75) in Horner form with the 1/2 and
        # Cp(Tref) folded into the coefficients: 3 fma per species
        for i in range(S):
            a, b, c, d = (float(v) for v in self.cp_coeff[i])
            k0, k1, k2, k3 = 0.5*(float(self.cp_ref[i]) + a), 0.5*b, 0.5*c, 0.5*d
            if d != 0.0:
                e = "((real(%r) * T + real(%r)) * T + real(%r)) * T + real(%r)" % (k3, k2, k1, k0)
            elif c != 0.0:
                e = "(real(%r) * T + real(%r)) * T + real(%r)" % (k2, k1, k0)
            else:
                e = "real(%r) * T + real(%r)" % (k1, k0)
            lines.append("    if (i == %d) return %s;" % (i, e))
        lines.append("    return real(0);")
        lines.append("}")
        # d/dT of the above (analytic node Jacobian of the stiff stepper)
        lines.append("__device__ __forceinline__ real rmt_cp_mean_dT(const int i, const real T) {")
        for i in range(S):
            a, b, c, d = (float(v) for v in self.cp_coeff[i])
            k1, k2, k3 = 0.5*b, 0.5*c, 0.5*d
            if d != 0.0:
                e = "(real(%r) * T + real(%r)) * T + real(%r)" % (3.0*k3, 2.0*k2, k1)
            elif c != 0.0:
                e = "real(%r) * T + real(%r)" % (2.0*k2, k1)
            else:
                e = "real(%r)" % k1
            lines.append("    if (i == %d) return %s;" % (i, e))
        lines.append("    return real(0);")
        lines.append("}")
        return "\n".join(lines) + "\n"

    def device_dag(self):
        """The DAG that is printed for the device: the traced one after strength reduction
        (lowering.Lowered.optimize); ``self.lowered`` stays the bit-exact trace."""
        if getattr(self, "_opt", None) is None:
            self._opt = self.lowered.optimize() if self.optimize else self.lowered
        return self._opt

    def lds_state(self, fp32, block, npt, want=None, chained=False):
        """How many of the two long-lived RK4 vectors (y_n, K accumulator) the on-chip stepper
        keeps in LDS instead of VGPRs.  Default: as many as fit in 126 KiB of the CU's 160 KiB
        (16 KiB go to the exp table, the rest to the scan/hand-over buffers),
        except for the single-workgroup kernel at 512 threads x 2 nodes with V <= 8, whose 249
        VGPRs hold everything without scratch (measured 12.3 vs 11.9 G node-steps/s); the chained
        kernel always prefers LDS (6.4 vs 3.4)."""
        per = self.V*block*npt*(4 if fp32 else 8)
        fit = min(2, (126*1024)//per)
        if want is not None:
            return min(int(want), fit)
        if not chained and block == 512 and npt == 2 and self.V <= 8 and not fp32:
            return 1 if self.model == "M2" else 0      # (M2's node functions need more registers: 13.7 / 13.4 / 11.8 G
            #                                              node-steps/s with 1 / 2 / 0 vectors in LDS)
        return fit

    def kcache_slots(self, gen=True):
        """doubles per mesh node the cache of the temperature-only rate constants needs (0: nothing to cache).  `gen`:
        the policy for constants whose exponent is not linear in 1/T, see lowering.Lowered.kcache_plan / KCACHE_GEN."""
        p = self.device_dag().kcache_plan(gen)
        return p["slots"] if p else 0

    def _kcache_lds(self, fp32, block, npt, state_vectors, gen, small_exp, node_major=False):
        """LDS bytes of a caching RK4 stepper: `state_vectors` of its long-lived vectors, the cache, the exp table (16 KiB;
        512 B in the small form a caching kernel may keep, and for one-wave workgroups) and the exchange buffers."""
        slots = self.kcache_slots(gen)
        if not slots or fp32 or self.model not in ("N2", "M2"):
            return None
        table = 512 if (small_exp or block <= 64) else 16384
        nodes = block*npt
        cache = nodes*((slots + 1) & ~1)*8 + nodes//4*8 if node_major else nodes*slots*8     # (RMT_KC_NODE_MAJOR: padded rows)
        return state_vectors*self.V*nodes*8 + cache + table + 4096

    def kcache_small_exp(self, gen):
        """True when a caching kernel can keep the 64-entry exp table: every table-driven exp of the mechanism is a cached
        constant, i.e. full evaluations (two more multiply-adds each) only happen when the reference point moves."""
        p = self.device_dag().kcache_plan(gen)
        return bool(p) and not p["outside_exp"]

    def kcache_fits(self, fp32, block, npt, lds_state=None, gen=True, small_exp=False, node_major=False):
        """True when the on-chip RK4 stepper can keep that cache in LDS beside the vectors it keeps there (model N2,
        fp64): within 158 KiB of the CU's 160."""
        need = self._kcache_lds(fp32, block, npt, self.lds_state(fp32, block, npt, lds_state), gen, small_exp, node_major)
        return need is not None and need <= 159*1024

    def kcache_fits_chain(self, fp32, block, npt, lds_state=None, gen=True, small_exp=False, node_major=False):
        """The same cache in the chained RK4 stepper (RMT_KCACHE_CHAIN): beside the chunk's RK4 vectors in LDS."""
        need = self._kcache_lds(fp32, block, npt, self.lds_state(fp32, block, npt, lds_state, chained=True), gen, small_exp,
                                node_major)
        return need is not None and need <= 159*1024

    def source(self, template, fp32=False, block=1024, npt=1, lds_state=None, defines=None):
        """Complete translation unit: prelude + template with the lowered kinetics spliced in."""
        if "RMT_KINETICS_SOURCE" not in template:
            raise ValueError("kernel template lacks the RMT_KINETICS_SOURCE marker")
        kin = self.device_dag().emit("rmt_kinetics", const_table=bool((defines or {}).get("RMT_KINETICS_KTAB")),
                                     kcache=(str((defines or {}).get("RMT_KCACHE", "0")) == "1"
                                             or str((defines or {}).get("RMT_KCACHE_CHAIN", "0")) == "1"),
                                     kcache_gen=KCACHE_GEN[str((defines or {}).get("RMT_KCACHE_GEN", "1"))],
                                     kcache_thr=(defines or {}).get("RMT_KCACHE_THR"))
        if (defines or {}).get("RMT_WITH_ROS4"):
            # the stiff stepper's node Jacobian is analytic: rates AND their partials by T, x_i, C_i
            kin += self.device_dag().gradient().emit_jac("rmt_kinetics_jac")
        if (defines or {}).get("RMT_WITH_N1"):
            # steady-state model N1: the pressure is a state variable, so the partials by P as well
            dag = self.device_dag()
            wrt = sorted({dag.g.nodes[i][1] for i in dag.live if dag.g.nodes[i][0] == "in"})
            kin += dag.gradient(wrt=wrt).emit_jac("rmt_kinetics_jacp", with_p=True)
        body = template.replace("RMT_KINETICS_SOURCE", kin, 1)
        return self.prelude(fp32, block, npt, lds_state, defines) + body

    @property
    def row_width(self):
        """doubles per member row: 16 fixed scalars, S inlet values, NU user parameters."""
        return MEMBER_FIXED + self.S + self.NU

    def new_row(self, modelInput):
        """Zeroed member row with the user-parameter columns (the member's own values of ``self.params``) filled."""
        row = np.zeros(self.row_width)
        if self.NU:
            VARS = modelInput['reaction-rates']['VARS']
            for k, nm in enumerate(self.params):
                row[MEMBER_FIXED + self.S + k] = float(VARS[nm])
        return row

    def digest(self, template, fp32, block, npt, lds_state=None, defines=None):
        h = hashlib.sha256()
        h.update(self.source(template, fp32, block, npt, lds_state, defines).encode())
        return h.hexdigest()[:24]


GAMACOTE_N2_ERROR = "setting an array element with a sequence."


def check_model_setting_n2():
    """MODEL_SETTING['GaMaCoTe0'] != "MAX" on the N2 path: the reference's modelEquationN2 then assigns the whole
    feed-concentration ARRAY to one element (``SpCoi0_Set = SpCoi0``, pbHomoReactor.py:3901-3904, where the other
    models write ``SpCoi0[i]``) and numpy raises ValueError on the first RHS evaluation - recorded from the reference
    itself in tests/golden/g11_model_setting.json.  There is no N2 result under that setting to reproduce, so the
    drop-in raises the same exception (before any device work) instead of inventing a per-species model the
    reference never ran.  Model N1 DOES run under it (member_constants_n1, n1.py); M2 never reads the setting."""
    if MODEL_SETTING['GaMaCoTe0'] != "MAX":
        raise ValueError(GAMACOTE_N2_ERROR)


def member_constants(modelInput, mech, zNo):
    """One reactor's scalars: the arithmetic of runN2's setup block (pbHomoReactor.py:3341-3466),
    returned both as a dict of named reference quantities (for parity tests and result packing)
    and as the packed row the kernels read.  (The scaling is the "MAX" one, GaMaCoTe0[i] = (vf/zf) max(SpCoi0);
    run_n2 refuses any other MODEL_SETTING like the reference's RHS does, see check_model_setting_n2.)"""
    mi = modelInput
    P = mi['operating-conditions']['pressure']
    T = mi['operating-conditions']['temperature']
    ReSpec = mi['reactor']
    ReInDi, ReLe = ReSpec['ReInDi'], ReSpec['ReLe']
    PaDi, BeVoFr = ReSpec['PaDi'], ReSpec['BeVoFr']
    CrSeAr = PI_CONST*(ReInDi**2)/4                                   # :3377
    VoFlRa0 = mi['feed']['volumetric-flowrate']
    SpCoi0 = np.array(mi['feed']['concentration'], dtype=float)
    if SpCoi0.shape != (mech.S,):
        raise ValueError("feed.concentration must have one entry per shell component")
    SpCo0 = np.sum(SpCoi0)
    InGaVe0 = VoFlRa0/(CrSeAr*BeVoFr)                                 # :3391
    SuGaVe0 = InGaVe0*BeVoFr                                          # :3393
    MoFri0 = SpCoi0/np.sum(SpCoi0)                                    # :3396
    MoWei = np.array(mech.MW, dtype=float)
    ExHe = mi['external-heat']
    Tm, U = ExHe['MeTe'], ExHe['OvHeTrCo']
    a = 4/ReInDi                                                      # :3411 (EfHeTrAr ignored)
    GaVii0 = np.array([compdb.viscosity(s, T) for s in mech.compList])
    GaMiVi = wilke(GaVii0, MoFri0, MoWei)                             # :3415-3416
    GaCpMeanList0 = np.array([(compdb.cp_value(s, Tref) + compdb.cp_value(s, T))*0.50
                              for s in mech.compList])                # :3420
    GaCpMeanMix0 = np.dot(MoFri0, GaCpMeanList0)                      # :3422
    MiMoWe0 = np.dot(MoFri0, MoWei)*1e-3                              # :3426
    GaDe0 = MiMoWe0*SpCo0                                             # :3429
    dz = 1/(zNo - 1)                                                  # :3439 (DoLe = 1)
    Tf, Pf, vf, zf, Cpf = T, P, SuGaVe0, ReLe, GaCpMeanMix0
    Cmax = np.max(SpCoi0)
    GaMaCoTe0 = (vf/zf)*np.repeat(Cmax, mech.S)                       # :3462-3464 ("MAX")
    GaHeCoTe0 = (GaDe0*vf*Tf*(Cpf/MiMoWe0)/zf)                        # :3466
    ergA = 150*GaMiVi*SuGaVe0/(PaDi**2)                               # :3970-3973 with v == vf
    ergB = ((1 - BeVoFr)**2)/(BeVoFr**3)
    ergD = (1 - BeVoFr)/(BeVoFr**3)
    named = {
        "CrSeAr": CrSeAr, "SpCoi0": SpCoi0, "SpCo0": SpCo0, "SuGaVe0": SuGaVe0, "GaMiVi": GaMiVi,
        "GaDe0": GaDe0, "GaCpMeanMix0": GaCpMeanMix0, "MiMoWe0": MiMoWe0, "dz": dz,
        "Tf": Tf, "Pf": Pf, "vf": vf, "zf": zf, "Cpif": GaCpMeanList0, "Cpf": Cpf,
        "GaMaCoTe0": GaMaCoTe0, "GaHeCoTe0": GaHeCoTe0, "EfHeTrAr": a, "Cmax": Cmax,
        "P0": P, "T0": T, "VoFlRa0": VoFlRa0, "U": U, "Tm": Tm,
    }
    row = mech.new_row(mi)
    F = MEMBER_FIELDS
    row[F["CMAX"]] = Cmax
    row[F["TF"]] = Tf
    row[F["P0"]] = P
    row[F["THETA_IN"]] = (T - Tf)/Tf                                   # :4108
    row[F["ALPHA_K"]] = dz*1.75*(SuGaVe0**2)*ergD/(PaDi*R_CONST)
    row[F["BETA"]] = -dz*ergA*ergB
    row[F["RHO_K"]] = 1.0/(R_CONST*GaDe0)
    row[F["INV_CP0"]] = 1.0/GaCpMeanMix0
    row[F["F1"]] = 1/(BeVoFr*(zf/vf))                                  # const_F1, :4075
    row[F["FT"]] = vf/zf
    row[F["INV_DZ"]] = float(zNo - 1)
    # pre-combined on the host (fewer per-reactor scalars in the kernel): FM = F1/GaMaCoTe0 and
    # GAIN_K = F1/(GaHeCoTe0 * RHO_K * INV_CP0), see rmt_node_post
    row[F["INV_MACOTE"]] = row[F["F1"]]/GaMaCoTe0[0]
    row[F["INV_HECOTE"]] = row[F["F1"]]/(GaHeCoTe0*row[F["RHO_K"]]*row[F["INV_CP0"]])
    row[F["UA"]] = U*a
    row[F["TM"]] = Tm
    row[F["CIN"]:F["CIN"] + mech.S] = SpCoi0/Cmax                      # :4090
    return named, row


def member_constants_m2(modelInput, mech, zNo):
    """Model M2 (pbReactor.py:552-700 setup, :845-1165 RHS): the packed row keeps the N2 layout
    (MEMBER_FIELDS) with the meanings listed above the M2 node functions in csrc/kernels/21_node_m2.inc."""
    mi = modelInput
    P = mi['operating-conditions']['pressure']
    T = mi['operating-conditions']['temperature']
    ReSpec = mi['reactor']
    ReInDi, ReLe = ReSpec['ReInDi'], ReSpec['ReLe']
    PaDi, BeVoFr = ReSpec['PaDi'], ReSpec['BeVoFr']
    CrSeAr = PI_CONST*(ReInDi**2)/4                                   # :592
    VoFlRa0 = mi['feed']['volumetric-flowrate']
    SpCoi0 = np.array(mi['feed']['concentration'], dtype=float)       # [kmol/m^3]
    if SpCoi0.shape != (mech.S,):
        raise ValueError("feed.concentration must have one entry per shell component")
    SpCo0 = np.sum(SpCoi0)
    GaMiVi = mi['feed']['mixture-viscosity']                          # :620
    ExHe = mi['external-heat']
    dz = ReLe/(zNo - 1)                                               # :629
    InGaVe0 = VoFlRa0/(CrSeAr*BeVoFr)                                 # :973
    ergB = ((1 - BeVoFr)**2)/(BeVoFr**3)
    ergD = (1 - BeVoFr)/(BeVoFr**3)
    named = {"CrSeAr": CrSeAr, "SpCoi0": SpCoi0, "SpCo0": SpCo0, "GaMiVi": GaMiVi, "dz": dz,
             "P0": P, "T0": T, "VoFlRa0": VoFlRa0, "ReLe": ReLe, "InGaVe0": InGaVe0}
    row = mech.new_row(mi)
    F = MEMBER_FIELDS
    row[F["CMAX"]] = 1.0
    row[F["TF"]] = T
    row[F["P0"]] = P
    row[F["THETA_IN"]] = T                                            # T0, :1151
    row[F["ALPHA_K"]] = dz*1.75*ergD/PaDi
    row[F["BETA"]] = dz*150*GaMiVi*ergB/(PaDi**2)
    row[F["RHO_K"]] = InGaVe0*BeVoFr*P/SpCo0
    row[F["INV_CP0"]] = (1 - BeVoFr)*ReSpec['CaDe']*ReSpec['CaSpHeCa']   # :1123
    row[F["F1"]] = 1/BeVoFr                                           # :1121
    row[F["FT"]] = BeVoFr
    row[F["INV_DZ"]] = 1.0/dz
    row[F["UA"]] = ExHe['OvHeTrCo']*ExHe['EfHeTrAr']*1e-3             # rmtUtility.py:445-450
    row[F["TM"]] = ExHe['MeTe']
    row[F["CIN"]:F["CIN"] + mech.S] = SpCoi0                          # :1135
    return named, row


def m2_newton_sweeps(rows, mech, zNo):
    """Newton sweeps the M2 pressure march needs (RMT_M2_NEWTON), from the relative pressure drop
    of the feed state over the bed (largest over the members): the error contracts like
    e' ~ 0.01..0.05 e^2 starting from e0 = drop, and the kernel accepts a last update < 3e-7."""
    rows = np.asarray(rows, dtype=float).reshape(-1, mech.row_width)
    F = MEMBER_FIELDS
    worst = 0.0
    for r in rows:
        c = r[F["CIN"]:F["CIN"] + mech.S]
        ct = np.sum(c)
        M = np.dot(c/ct, mech.MW)*1e-3
        sup, P0 = r[F["RHO_K"]]*ct, r[F["P0"]]
        worst = max(worst, zNo*(r[F["BETA"]]*sup/P0 + r[F["ALPHA_K"]]*(M*ct)*sup*sup/P0**2)/P0)
    return 2 if worst < 2e-3 else 3 if worst < 0.12 else 4 if worst < 0.3 else 5


def initial_state_m2(named, mech, zNo):
    """IV2D flattened (pbReactor.py:641-653): feed concentrations and feed temperature everywhere."""
    IV = np.zeros((mech.V, zNo))
    for i in range(mech.S):
        IV[i, :] = named["SpCoi0"][i]
    IV[mech.S, :] = named["T0"]
    return IV.flatten()


MEMBER1_FIELDS = {
    "CMAX": 0, "TF": 1, "PF": 2, "SPCO0": 3, "ERGA": 4, "ERGC": 5, "SUGAVE0": 6, "RHO_K": 7,
    "INV_CP0": 8, "EPS": 9, "INV_MACOTE": 10, "INV_HECOTE": 11, "UA": 12, "TM": 13, "GADE0": 14, "CIN": 16,
}


def member_constants_n1(modelInput, mech):
    """Packed constants of the steady-state model N1: the setup block of runN1
    (pbHomoReactor.py:2694-2900) - identical to runN2's apart from vf = VoFlRa0/CrSeAr - and the
    h-independent factors of modelEquationN1 (:3017-3314); layout M1_* in csrc/kernels/22_node_n1.inc.
    The row is the same for both values of MODEL_SETTING['GaMaCoTe0']: under the per-species scaling the kernel
    (RMT_N1_SCALE_FIX, set by n1.run_n1) derives scale_i = CMAX*CIN_i and 1/GaMaCoTe0[i] = INV_MACOTE/CIN_i from it;
    the named dict carries "SpCoi0_Set" / "GaMaCoTe0" as the reference would have them (:2819-2821, 3159-3160)."""
    nm, _ = member_constants(modelInput, mech, 2)
    mi = modelInput
    ReSpec = mi['reactor']
    PaDi, BeVoFr = ReSpec['PaDi'], ReSpec['BeVoFr']
    vf = nm["VoFlRa0"]/nm["CrSeAr"]                                     # :2763
    zf, Pf = nm["zf"], nm["Pf"]
    GaMaCoTe0 = (vf/zf)*nm["Cmax"]
    GaHeCoTe0 = (nm["GaDe0"]*vf*nm["Tf"]*(nm["Cpf"]/nm["MiMoWe0"])/zf)
    ergB = ((1 - BeVoFr)**2)/(BeVoFr**3)
    ergD = (1 - BeVoFr)/(BeVoFr**3)
    row = mech.new_row(mi)
    F = MEMBER1_FIELDS
    row[F["CMAX"]], row[F["TF"]], row[F["PF"]], row[F["SPCO0"]] = nm["Cmax"], nm["Tf"], Pf, nm["SpCo0"]
    row[F["ERGA"]] = 150*nm["GaMiVi"]*ergB/(PaDi**2)/(Pf/zf)             # :3206-3220
    row[F["ERGC"]] = 1.75*ergD/PaDi/(Pf/zf)
    row[F["SUGAVE0"]] = (nm["VoFlRa0"]/(nm["CrSeAr"]*BeVoFr))*BeVoFr    # InGaVe0*eps, :3100-3102
    row[F["RHO_K"]] = 1.0/(R_CONST*nm["GaDe0"])
    row[F["INV_CP0"]] = 1.0/nm["GaCpMeanMix0"]
    row[F["EPS"]] = BeVoFr
    row[F["INV_MACOTE"]] = 1.0/GaMaCoTe0
    row[F["INV_HECOTE"]] = 1.0/GaHeCoTe0
    row[F["UA"]] = nm["U"]*nm["EfHeTrAr"]
    row[F["TM"]] = nm["Tm"]
    row[F["GADE0"]] = nm["GaDe0"]
    row[F["CIN"]:F["CIN"] + mech.S] = nm["SpCoi0"]/nm["Cmax"]
    fix = MODEL_SETTING['GaMaCoTe0'] != "MAX"
    nm = dict(nm, vf=vf, SpCoi0_Set=np.array(nm["SpCoi0"], dtype=float) if fix else nm["Cmax"],
              GaMaCoTe0=(vf/zf)*(np.array(nm["SpCoi0"], dtype=float) if fix else np.repeat(nm["Cmax"], mech.S)))
    return nm, row


def uniform_columns(rows):
    """(first row, bool mask of the columns that hold one value in every row)."""
    rows = np.asarray(rows, dtype=float)
    return rows[0].copy(), np.all(rows == rows[0], axis=0)


def uniform_member_defines(rows, S, vals=None, mask=None):
    """Prelude #defines (RMT_MC_<FIELD>) for the member fields that are identical in every row:
    they become literals of the kernel (see rmt_load_member in csrc/kernels/10_member.inc).  ``vals`` /
    ``mask`` override the locally computed ones (multi-rank ensembles agree on them first).  ``rows``: 2-D
    [E][row width]; the user-parameter columns beyond 16+S are always read at run time."""
    if vals is None or mask is None:
        rows = np.asarray(rows, dtype=float)
        vals, mask = uniform_columns(rows if rows.ndim == 2 else rows.reshape(1, -1))
    out = {}
    for name, idx in MEMBER_FIELDS.items():
        if name == "CIN":
            if np.all(mask[idx:idx + S]):
                out["RMT_MC_CIN"] = "{" + ", ".join(repr(float(v)) for v in vals[idx:idx + S]) + "}"
        elif mask[idx]:
            out["RMT_MC_" + name] = repr(float(vals[idx]))
    return out


def initial_states(named_list, mech, zNo, init=None):
    """[E][V*zNo]: initial_state (or ``init``, e.g. initial_state_m2) of every member at once.  Both reference forms
    fill each row of the (V, zNo) matrix with ONE value, so the E x V matrix of those values (taken from a 1-node
    evaluation of the per-member function) is broadcast along the mesh - bit-identical to stacking the per-member
    arrays, without E Python-level fills of a zNo-long matrix (0.3 ms each at zNo = 1024: 0.6 s for a 2048-member sweep)."""
    init = init or initial_state
    first = np.array([init(nm, mech, 1) for nm in named_list], dtype=np.float64).reshape(len(named_list), mech.V, 1)
    return np.ascontiguousarray(np.broadcast_to(first, (len(named_list), mech.V, zNo))).reshape(len(named_list), mech.V*zNo)


def initial_state(named, mech, zNo):
    """IV2D flattened (pbHomoReactor.py:3483-3497)."""
    IV = np.zeros((mech.V, zNo))
    for i in range(mech.S):
        IV[i, :] = named["SpCoi0"][i]/np.max(named["SpCoi0"])
    return IV.flatten()
