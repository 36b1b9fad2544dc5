"""Device-side driver of model N2: the replacement for the time loop of
PackedBedHomoReactorClass.runN2 (PyREMOT/docs/pbHomoReactor.py:3556-3704).

The reference calls scipy's solve_ivp once per output interval with a Python RHS
(:3589-3610); here each interval is ONE launch of a device-resident explicit stepper
(fixed-step RK4 or adaptive Dormand-Prince RK45) over all mesh nodes of all ensemble members.
torch is used only as the owner of device memory and streams.
"""
import ctypes as C
from timeit import default_timer as timer

import os

import numpy as np

from . import hipbind, plan
from .lowering import FLAG_DIV0, FLAG_DOMAIN, FLAG_NONFINITE, FLAG_OVERFLOW, FLAG_STEP
from .settings import DEVICE_DEFAULTS, ROUND_FUN_ACCURACY, solverSetting

FLAG_PRESSURE = 32
DEVICE_IVPS = ("hip-rk4", "hip-rk45", "hip-ros4", "hip-auto", "AM", "hip-ab3")
FEATURE_DEFINES = {"ros4": "RMT_WITH_ROS4", "n1": "RMT_WITH_N1"}


N_CUS = 256          # MI355X; the C library clamps the team count to the device's real CU count


def choose_geometry(N, V, fp32=False, E=None):
    """(block, nodes_per_thread) of the generated kernels for E reactors of N nodes.
    The on-chip stepper holds block*npt nodes per workgroup; longer reactors are chained over
    several workgroups (rmt_n2_rk4_chain).  The table is what measured fastest on MI355X
    (profiles/round1_chain.md, round2_shapes.md):
      * big ensembles: 512 threads x 2 nodes (1024-node chunks, 2 waves/SIMD);
      * ensembles that would leave at least half of the 256 CUs idle with one workgroup per 1024 nodes:
        the reactors are cut into 128-, 256- or 512-node chunks (the smallest that keeps E x chunks <= 256
        workgroups, all resident), e.g. ONE 4096-node reactor over 32 CUs (10.7 us/step vs 217 us on one CU),
        64 x 1024 nodes as 4 chunks each (1.55x the one-CU-per-reactor rate)."""
    if N > 256 and E is not None and 2*E*(-(-N//1024)) <= N_CUS:
        for block, npt in ((128, 1), (256, 1), (256, 2)):
            if E*(-(-N//(block*npt))) <= N_CUS:
                return block, npt
    for block in (64, 128, 256, 512):
        if N <= block:
            return block, 1
    return (512, 2) if V <= 8 else (512, 1)


def ros4_quad(mech, fp32=False):
    """True when the stiff stepper runs in its one-node-on-four-lanes layout (kernels/61_ros4_quad.inc): mechanisms
    wider than 8 variables, whose V x V node inverse does not fit one lane's registers (model N2, fp64)."""
    return mech.V > 8 and getattr(mech, "model", "N2") == "N2" and not fp32


def ros4_block(V, N, fp32=False, quad=None):
    """Workgroup size for the stiff stepper: at most 256 threads (one V x V block inverse per lane),
    and the block's five stage vectors G_1..G_5 must fit in LDS next to the 16 KiB exp table and the
    hand-over buffers (5*V*block*sizeof(real) <= 140 KiB).  (Quad layout, V > 8 in fp64: 256 threads = 64 nodes.)"""
    if quad is None:
        quad = V > 8 and not fp32
    if quad:
        return 256 if N > 32 else 64*((4*N + 63)//64)
    block = min(256, 64*((N + 63)//64))
    while block > 64 and 5*V*block*(4 if fp32 else 8) > 140*1024:
        block //= 2
    return block


def rk45_block(V, N, fp32=False):
    """Workgroup size for the adaptive explicit stepper: the largest (<= 256) whose seven stage
    derivatives of a node block fit in LDS (7*V*block*sizeof(real) <= 112 KiB, csrc RMT_RK45_KLDS)."""
    block = min(256, 64*((N + 63)//64))
    while block > 64 and 7*V*block*(4 if fp32 else 8) > 112*1024:
        block //= 2
    return block


MAX_CHUNKS = 64          # include/rmt_n2.h RMT_N2_MAX_CHUNKS


def rk45_geometry(V, N, fp32=False, chain=True, E=None):
    """(block, nodes_per_thread, defines) for the adaptive explicit stepper.  The on-chip kernels keep 4
    long-lived vectors per node, RMT_RK45_LDS of them in LDS (V*block*npt reals each, at most 136 KiB
    together), the rest in VGPRs - measured on MI355X (profiles/round2_rk45.md): 512 x 2 with 2 vectors in LDS
    for V <= 8, 256 x 2 for the 12-species mechanism.  A reactor that fits one such workgroup runs
    rmt_n2_rk45_reg; a longer one is cut into chunks of that size on as many CUs (rmt_n2_rk45_chain; `chain`),
    up to MAX_CHUNKS; beyond that the memory-resident kernel (rk45_block).
    `E` (reactors on this GPU, when the caller knows it): an ensemble that leaves CUs idle with those chunks is cut
    finer - ONE node per lane, chunks of 512 / 256 nodes (256 / 128 for V > 8) - as long as all chunks of all
    reactors stay co-resident: a step is latency-bound per lane, so half the work per lane on twice the CUs is
    30-70 % faster (profiles/round2_rk45_chain.md)."""
    size = 4 if fp32 else 8
    big = V <= 8
    if big and N <= 1024:
        block, npt = choose_geometry(N, V, fp32)
    elif N <= 512:
        block, npt = (256, 2) if N > 256 else (64*((N + 63)//64), 1)
    elif chain and -(-N//(1024 if big else 512)) <= min(MAX_CHUNKS, N_CUS):
        block, npt = (512, 2) if big else (256, 2)
    else:
        return rk45_block(V, N, fp32), 1, {}
    if chain and E is not None and N > 256:
        for blk in ((256, 512) if big else (128, 256)):          # finest first
            C = -(-N//blk)
            if C >= 2 and blk < block*npt and E*C <= N_CUS and C <= MAX_CHUNKS:
                block, npt = blk, 1
                break
    slots = 2
    while slots > 0 and slots*V*block*npt*size > 136*1024:
        slots -= 1
    if slots == 0:
        return rk45_block(V, N, fp32), 1, {}
    return block, npt, {"RMT_RK45_LDS": str(slots)}


KC_REFRESH = 8          # csrc/kernels/50_rk4.inc RMT_KC_REFRESH: the cache's reference point moves every 8th step ...
KC_MAX_AGE = 1.3e-5     # ... RMT_KC_MAX_AGE: or sooner, so that no reference point is older than this much model time [s]


def kc_period(defines, dt):
    """steps between two moves of the reference point of a caching one-workgroup RK4 stepper at step size dt (the kernel's
    own rule, csrc/kernels/50_rk4.inc rmt_rk4_reg_body)."""
    K = int(defines.get("RMT_KC_REFRESH", 1))
    age = float(defines.get("RMT_KC_MAX_AGE", KC_MAX_AGE))
    return 1 if K <= 1 or not dt > 0 else int(min(float(K), max(1.0, age/dt)))


def kcache_choice(mech, N, fp32, block, npt, lds_state, defines):
    """(defines, lds_state) with the RK4 steppers' cache of the temperature-only rate constants switched on where it has
    been measured to pay (csrc/kernels/50_rk4.inc rmt_rk4_reg_body / rmt_rk4_chain_body; profiles/round3_kcache.md), model
    N2 in fp64 (the list is for at most 8 variables per node; wider mechanisms: see the table in the body):

    * one workgroup per reactor at 512 x 2 (RMT_KCACHE; the bench shape, 1.62e10 -> 1.95e10 node-steps/s) and at the
      small geometries of short reactors (2048 x 20 nodes at 64 x 1: 3.2e9 -> 5.0e9, 2048 x 128 at 128 x 1: 1.21e10 ->
      1.53e10, 1024 x 256 at 256 x 1: 1.47e10 -> 1.58e10; profiles/round3_kcache.md): 1/T_ref,
      log T_ref, T_ref, the Arrhenius constants AND the equilibrium constants (RMT_KCACHE_GEN 2: one slot each, the
      exponent's change from the differences of its basis functions T^n, log T) when the mechanism's exponents decompose
      that way and every table-driven exp is then a cached constant (the kernel keeps the 64-entry exp table, which makes
      the room in LDS); else without the equilibrium constants (RMT_KCACHE_GEN 0).  y_n moves to LDS (lds_state 1) to make
      room in the register file; the reference point moves every 8th step;
    * chained workgroups, every geometry (RMT_KCACHE_CHAIN, RMT_KCACHE_GEN 0): 256 x 4096 nodes at 512 x 2 +10 %, 128 x
      1024 at 256 x 2 +8 %, 64 x 1024 at 256 x 1 +5 %, ONE 4096-node reactor at 128 x 1 +7.5 %.

    An explicit "RMT_KCACHE" / "RMT_KCACHE_CHAIN" in `defines` (0 or 1) is left alone, and so is an lds_state that does
    not leave the room."""
    defs = dict(defines or {})
    chained = int(N) > int(block)*int(npt)
    key = "RMT_KCACHE_CHAIN" if chained else "RMT_KCACHE"
    geo = (int(block), int(npt))
    model = getattr(mech, "model", "N2")
    if key in defs or fp32 or model not in ("N2", "M2"):
        return defs, lds_state
    if model == "M2":
        # the dimensional model: its one-workgroup stepper at 512 x 2 (which keeps y_n in LDS anyway), Arrhenius constants
        # only - with the equilibrium constants its step loop spills (28 scratch accesses)
        if chained or geo != (512, 2) or lds_state not in (None, 1) or mech.V > 8 \
                or not mech.kcache_fits(fp32, block, npt, 1, gen=False):
            return defs, lds_state
        defs.update({key: "1", "RMT_KCACHE_GEN": "0"})
        defs.setdefault("RMT_KC_REFRESH", str(KC_REFRESH))
        return defs, 1
    # where the cache measured faster (profiles/round3_kcache.md) -> the lds_state it needs (None: the geometry's default).
    # At most 8 variables per node (DME): every chained geometry; one workgroup per reactor at 512 x 2, at 64 / 128 / 256
    # threads and at 512 x 1 with y_n in LDS (1.52e10 -> 1.72e10; with the geometry's own lds_state it is SLOWER).  Wider mechanisms (12 species, V = 13): 64 x 1 (8.7e9 ->
    # 1.21e10), 512 x 1 (1.07e10 -> 1.16e10) and the chained 512 x 1 (9.9e9 -> 1.04e10), with y_n in LDS; 128 x 1 is
    # SLOWER there (8.8e9 -> 7.1e9).
    if mech.V <= 8:
        good = {geo: (1 if geo == (512, 2) else lds_state)} if chained and geo[1] <= 2 else \
            {(512, 2): 1, (64, 1): lds_state, (128, 1): lds_state, (256, 1): lds_state, (512, 1): 1}
    else:
        good = {(512, 1): 1} if chained else {(64, 1): lds_state, (512, 1): 1}
    if geo not in good or (good[geo] == 1 and lds_state not in (None, 1)):
        return defs, lds_state
    want = good[geo]
    if chained:
        if not mech.kcache_fits_chain(fp32, block, npt, want, gen=False):
            return defs, lds_state
        defs.update({key: "1", "RMT_KCACHE_GEN": "0"})
        return defs, want
    # equilibrium constants too where that measured faster: everywhere but at 128 x 1 (1.53e10 with the Arrhenius
    # constants alone against 1.39e10 - the wider kernel loses a wave of occupancy there) and 512 x 1 (1.72e10 against 1.52e10)
    if geo not in ((128, 1), (512, 1)) and mech.kcache_small_exp("basis") and mech.kcache_slots("basis") > mech.kcache_slots(False) \
            and mech.kcache_fits(fp32, block, npt, want, gen="basis", small_exp=True, node_major=True):
        # (a node's slots side by side in LDS: one address register per node; with slot-major rows of 8 KiB the far slots
        # need registers of their own and the 512 x 2 step loop spills - 1.86e10 against 1.95e10 node-steps/s)
        defs.update({key: "1", "RMT_KCACHE_GEN": "2", "RMT_KC_SMALL_EXP": "1", "RMT_KC_NODE_MAJOR": "1"})
    elif mech.kcache_fits(fp32, block, npt, want, gen=False):
        defs.update({key: "1", "RMT_KCACHE_GEN": "0"})
    else:
        return defs, lds_state
    defs.setdefault("RMT_KC_REFRESH", str(KC_REFRESH))
    return defs, want


def device_source(mech, members, N, fp32=False, block=None, npt=None, lds_state=None, defines=None,
                  specialize=None, features=(), have_code=False):
    """What N2Device compiles for (mechanism, member rows, mesh): geometry, the prelude #defines (optional
    kernel families, model-M2 sweeps, sweep-invariant member fields as literals) and the translation unit.
    Needs no GPU - `precompile` uses it to fill the in-tree code-object cache ahead of time."""
    members = np.ascontiguousarray(members, dtype=np.float64)
    if members.ndim == 1:
        members = members.reshape(1, -1)
    E = members.shape[0]
    b, n = choose_geometry(int(N), mech.V, fp32, E)
    block, npt = int(block or b), int(npt or n)
    defs, lds_state = kcache_choice(mech, N, fp32, block, npt, lds_state, defines)
    # optional kernel families: "ros4" (stiff stepper), "n1" (steady-state model); their
    # unrolled VxV linear algebra is most of the JIT time, so they are compiled on demand
    for f in features:
        defs[FEATURE_DEFINES[f]] = "1"
    if "ros4" in features and "RMT_ROS_QUAD" not in defs and ros4_quad(mech, fp32) and npt == 1:
        defs["RMT_ROS_QUAD"] = "1"            # wide mechanism: one node on four lanes (the C library is told below)
    if getattr(mech, "model", "N2") == "M2" and "RMT_M2_NEWTON" not in defs and not have_code:
        defs["RMT_M2_NEWTON"] = str(plan.m2_newton_sweeps(members, mech, int(N)))
    # sweep-invariant member fields become literals (frees SGPRs); a single reactor is NOT
    # specialised by default - every new operating point would cost a 2-3 s JIT
    if specialize is None:
        specialize = E >= 2
    if specialize:
        defs.update(plan.uniform_member_defines(members, mech.S))
    # "RMT_KCACHE": "1" (kcache_choice above, or the caller's own): the on-chip RK4 stepper caches the temperature-only
    # rate constants per node in LDS - that has to fit
    gen = plan.KCACHE_GEN[str(defs.get("RMT_KCACHE_GEN", "1"))]
    small = str(defs.get("RMT_KC_SMALL_EXP", "0")) == "1"
    if small and not mech.kcache_small_exp(gen):
        raise ValueError("RMT_KC_SMALL_EXP=1: the mechanism has table-driven exp evaluations that are not cached constants")
    if str(defs.get("RMT_KCACHE_CHAIN", "0")) == "1" and not mech.kcache_fits_chain(
            fp32, block, npt, lds_state, gen, small, str(defs.get("RMT_KC_NODE_MAJOR", "0")) == "1"):
        raise ValueError("RMT_KCACHE_CHAIN=1: the cache of the temperature-only rate constants (%d doubles per node) does "
                         "not fit beside the chunk's RK4 vectors (model N2, fp64)" % mech.kcache_slots(gen))
    if str(defs.get("RMT_KCACHE", "0")) == "1" and not (
            int(N) <= block*npt and mech.kcache_fits(fp32, block, npt, lds_state, gen, small,
                                                     str(defs.get("RMT_KC_NODE_MAJOR", "0")) == "1")):
        raise ValueError("RMT_KCACHE=1: the cache of the temperature-only rate constants (%d doubles per node) does not "
                         "fit this geometry (needs the on-chip RK4 stepper with its vectors in registers, model N2, fp64)"
                         % mech.kcache_slots())
    tpl = hipbind.kernel_template()
    # (the user's lds_state, possibly None, is what selects the per-kernel defaults)
    src = mech.source(tpl, fp32, block, npt, lds_state, defs)
    key = mech.digest(tpl, fp32, block, npt, lds_state, defs)
    return block, npt, defs, src, key


def compile_options(block, npt, features=(), extra_opts="", defines=None):
    """hipRTC options beyond the library's own (csrc/rmt_n2.cpp rmt_n2_compile) for a code object of this geometry.  The
    RK4 steppers at 512 x 2 - two waves per SIMD at the register limit - are scheduled with LLVM's AMDGPU register
    pressure trackers (`-amdgpu-use-amdgpu-trackers=1`): bench 2.018e10 -> 2.045e10 node-steps/s, chained 256 x 4096
    nodes 1.53e10 -> 1.58e10; the stiff stepper (-1.5 %) and the small chunks of ONE long reactor (-1.9 %) are SLOWER
    with it, and so is the on-chip RK45 stepper on the bench sweep (6.90e9 -> 6.71e9; tools/microbench/exp_r3ae.sh,
    rk45_ab.py): code objects with optional kernel families or built for RK45 (rk45_geometry's "RMT_RK45_LDS") keep the
    default."""
    auto = "-mllvm -amdgpu-use-amdgpu-trackers=1" if ((int(block), int(npt)) == (512, 2) and not features
                                                       and "RMT_RK45_LDS" not in (defines or {})) else ""
    if os.environ.get("RMT_N2_NO_TRACKERS"):          # (A/B measurements)
        auto = ""
    if not auto or "amdgpu-use-amdgpu-trackers" in (extra_opts or ""):
        return extra_opts or ""
    return ("%s %s" % (extra_opts, auto)).strip()


def precompile(mech, members, N, arch="gfx950", extra_opts="", **kw):
    """Cross-compile (hipRTC, no GPU needed) the code object N2Device(mech, members, N, **kw) will load and
    leave it in the in-tree cache; returns its cache key."""
    block, npt, _, src, key = device_source(mech, members, N, **kw)
    hipbind.compile_cached(src, key, arch, compile_options(block, npt, kw.get("features", ()), extra_opts,
                                                           kw.get("defines")))
    return key


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise hipbind.RmtN2Error("no HIP device visible: the N2 integrator has no CPU fallback")
    return torch


class N2Device:
    """One compiled mechanism + E packed member rows on one GPU."""

    def __init__(self, mech, members, N, fp32=False, block=None, npt=None, device=None,
                 extra_opts="", lds_state=None, defines=None, code=None, specialize=None,
                 features=()):
        torch = _torch()
        self.torch = torch
        self.mech, self.N, self.fp32 = mech, int(N), bool(fp32)
        members = np.ascontiguousarray(members, dtype=np.float64)
        if members.ndim == 1:
            members = members.reshape(1, -1)
        assert members.shape[1] == mech.row_width, "member rows must hold 16 + S + NU doubles"
        self.E = members.shape[0]
        self.members = members
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.features = tuple(features)
        self.block, self.npt, self.defines, src, key = device_source(
            mech, members, self.N, self.fp32, block, npt, lds_state, defines, specialize, self.features,
            have_code=code is not None)
        _, lds_state = kcache_choice(mech, self.N, self.fp32, self.block, self.npt, lds_state, defines)
        self.lds_state = mech.lds_state(self.fp32, self.block, self.npt, lds_state)
        arch = torch.cuda.get_device_properties(self.device).gcnArchName.split(":")[0]
        if code is None:      # an ensemble rank may receive rank 0's code object instead
            code = hipbind.compile_cached(src, key, arch, compile_options(self.block, self.npt, self.features, extra_opts,
                                                                          self.defines))
        self._code = C.create_string_buffer(code, len(code))
        p = hipbind.Plan()
        p.abi_version = hipbind.ABI_VERSION
        p.n_species, p.n_reactions, p.n_vars = mech.S, mech.R, mech.V
        p.n_nodes, p.n_members, p.fp32 = self.N, self.E, int(self.fp32)
        p.block, p.nodes_per_thread = self.block, self.npt
        p.n_user_params = mech.NU
        self.ros_quad = str(self.defines.get("RMT_ROS_QUAD", "0")) == "1"
        p.ros4_nodes_per_block = self.block//4 if self.ros_quad else 0
        p.code_object = C.cast(self._code, C.c_void_p)
        p.code_size = len(code)
        p.members = members.ctypes.data_as(C.POINTER(C.c_double))
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            hipbind.check(hipbind.lib().rmt_n2_create(C.byref(p), C.byref(h)))
        self.h = h
        # node-function evaluations one Jacobian of the stiff stepper costs: the analytic Jacobian comes with the
        # stage-1 evaluation (rates and their partials in ONE fused pass, rmt_node_jac) and is charged as one more;
        # the forward-difference form (RMT_ROS_JAC_FD 1) costs V columns, +1 in model M2 for the upwind coupling
        fd = str(self.defines.get("RMT_ROS_JAC_FD", "0")) == "1"
        self.jacobian_evals = (mech.V + (1 if getattr(mech, "model", "N2") == "M2" else 0)) if fd else 1
        self.dtype = torch.float32 if self.fp32 else torch.float64
        self._stats = torch.zeros((self.E, 4), dtype=torch.float64, device=self.device)
        self.use_current_stream()

    # -- plumbing
    def use_current_stream(self):
        s = self.torch.cuda.current_stream(self.device).cuda_stream
        hipbind.check(hipbind.lib().rmt_n2_set_stream(self.h, C.c_void_p(s)))

    def set_members(self, members):
        """Replace the per-reactor constant rows (same E) without recompiling - e.g. the next
        point of a sweep.  Not available when member fields were baked into the kernel as literals
        (``specialize``): those fields would silently keep their old values."""
        if any(k.startswith("RMT_MC_") for k in self.defines):
            raise hipbind.RmtN2Error("this kernel was specialised on its member rows "
                                     "(N2Device(..., specialize=False) keeps them run-time)")
        members = np.ascontiguousarray(members, dtype=np.float64).reshape(self.E, -1)
        assert members.shape[1] == self.mech.row_width
        self.members = members
        hipbind.check(hipbind.lib().rmt_n2_set_members(self.h, members.ctypes.data_as(C.POINTER(C.c_double))))

    def set_mode(self, mode):
        hipbind.check(hipbind.lib().rmt_n2_set_mode(self.h, {"auto": 0, "reg": 1, "mem": 2, "chain": 3}[mode]))

    def close(self):
        if getattr(self, "h", None):
            hipbind.lib().rmt_n2_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def to_device(self, y):
        t = self.torch.as_tensor(np.ascontiguousarray(y), dtype=self.dtype)
        return t.reshape(self.E, self.mech.V*self.N).contiguous().to(self.device)

    def _chk_state(self, y):
        assert y.is_cuda and y.dtype == self.dtype and y.is_contiguous()
        assert y.numel() == self.E*self.mech.V*self.N, "state must be [E][V][N]"

    # -- hot path entry points
    def rhs(self, y, t=0.0):
        self._chk_state(y)
        out = self.torch.empty_like(y)
        hipbind.check(hipbind.lib().rmt_n2_rhs(self.h, float(t), C.c_void_p(y.data_ptr()),
                                               C.c_void_p(out.data_ptr())))
        return out

    def rk4(self, y, dt, nsteps, t0=0.0):
        """In place: nsteps RK4 steps of size dt."""
        self._chk_state(y)
        hipbind.check(hipbind.lib().rmt_n2_rk4(self.h, C.c_void_p(y.data_ptr()), float(t0), float(dt),
                                               int(nsteps)))

    def multistep(self, y, dt, nsteps, method="PreCorr3", t0=0.0):
        """In place: the reference's AdBash3 / PreCorr3 (odeSolver.py:43-102), nsteps >= 3."""
        self._chk_state(y)
        hipbind.check(hipbind.lib().rmt_n2_multistep(
            self.h, C.c_void_p(y.data_ptr()), float(t0), float(dt), int(nsteps),
            {"AdBash3": 0, "PreCorr3": 1}[method]))

    def rk45(self, y, t0, t1, rtol, atol, h0, max_steps):
        self._chk_state(y)
        hipbind.check(hipbind.lib().rmt_n2_rk45(self.h, C.c_void_p(y.data_ptr()), float(t0), float(t1),
                                                float(rtol), float(atol), float(h0), int(max_steps),
                                                C.c_void_p(self._stats.data_ptr())))

    def ros4(self, y, t0, t1, rtol, atol, h0, max_steps):
        """In place: stiff integration (RODAS4, order 4(3)) from t0 to t1 with per-reactor step control
        (needs a code object generated with features=("ros4",) and block <= 512; 256 is fastest)."""
        self._chk_state(y)
        if "ros4" not in self.features:
            raise hipbind.RmtN2Error("create the device with features=('ros4',) to use the stiff stepper")
        hipbind.check(hipbind.lib().rmt_n2_ros4(self.h, C.c_void_p(y.data_ptr()), float(t0), float(t1),
                                                float(rtol), float(atol), float(h0), int(max_steps),
                                                C.c_void_p(self._stats.data_ptr())))

    def n1_profile(self, rows1, nout, rtol, atol, h0, max_steps):
        """Steady-state model N1 (needs features=("n1",)): one profile per member row of ``rows1`` (layout M1_*),
        sampled at z* = k/(nout-1); returns the host array [E][nout][S+2] (S+1 when iso-thermal)."""
        torch = self.torch
        rows1 = np.ascontiguousarray(rows1, dtype=np.float64).reshape(self.E, self.mech.row_width)
        V1 = self.mech.S + (1 if self.mech.iso else 2)
        out = torch.zeros((self.E, int(nout), V1), dtype=torch.float64, device=self.device)
        hipbind.check(hipbind.lib().rmt_n1_profile(
            self.h, rows1.ctypes.data_as(C.POINTER(C.c_double)), C.c_void_p(out.data_ptr()), int(nout),
            float(rtol), float(atol), float(h0), int(max_steps), C.c_void_p(self._stats.data_ptr())))
        return out.cpu().numpy()

    def rk45_stats(self):
        raw = self._stats.cpu().numpy()
        return {"t_end": raw[:, 0].copy(), "h_last": raw[:, 1].copy(),
                "accepted": raw[:, 2].copy().view(np.int64), "rejected": raw[:, 3].copy().view(np.int64)}

    def status(self):
        flags = np.zeros(self.E, dtype=np.uint32)
        hipbind.check(hipbind.lib().rmt_n2_status(self.h, flags.ctypes.data_as(C.POINTER(C.c_uint32))))
        return flags

    def last_geometry(self):
        """(workgroups per reactor, teams) of the last rk4 / rk45 / ros4 launch - what the library's auto mode chose."""
        c, t = C.c_int(), C.c_int()
        hipbind.check(hipbind.lib().rmt_n2_last_geometry(self.h, C.byref(c), C.byref(t)))
        return c.value, t.value

    def fallbacks(self):
        """Reactor-launches the cached RK4 steppers handed to their plain twins so far (rmt_n2_fallbacks): a reactor whose
        temperature left the range of its cached rate constants during a launch is integrated again in full."""
        n = C.c_uint64()
        hipbind.check(hipbind.lib().rmt_n2_fallbacks(self.h, C.byref(n)))
        return int(n.value)

    def last_kernel_ms(self):
        ms = C.c_float()
        hipbind.check(hipbind.lib().rmt_n2_last_kernel_ms(self.h, C.byref(ms)))
        return ms.value

    def raise_on_flags(self):
        """Turn device status words into the exception the reference's Python path raises inside
        the user lambdas (SURVEY.md section 5 'Failure detection')."""
        flags = self.status()
        bad = np.nonzero(flags)[0]
        if len(bad) == 0:
            return
        f = int(flags[bad[0]])
        where = "reactor %d of %d (flags=0x%x)" % (bad[0], self.E, f)
        if f & FLAG_DOMAIN:
            raise ValueError("math domain error in " + where)
        if f & FLAG_DIV0:
            raise ZeroDivisionError("float division by zero in " + where)
        if f & FLAG_OVERFLOW:
            raise OverflowError("math range error in " + where)
        if f & FLAG_STEP:
            raise RuntimeError("adaptive step control failed (step underflow / max steps) in " + where)
        if f & FLAG_NONFINITE:
            raise FloatingPointError("state became NaN/Inf in " + where + " - step size too large?")
        if f & FLAG_PRESSURE:
            raise RuntimeError("model M2: the Newton sweeps of the pressure march did not converge in "
                               + where + " - pass defines={'RMT_M2_NEWTON': 4} (pressure drop > 15 % of P)")
        raise RuntimeError("device error in " + where)


class AutoStepper:
    """ivp "hip-auto": the device counterpart of LSODA's automatic method switching (what the reference's default
    `solve_ivp(..., method="LSODA")` does, pbHomoReactor.py:3576, 3609), decided per output interval and for the
    whole ensemble at once:

      * the first interval starts with a PROBE: at most `auto-probe-steps` Dormand-Prince steps.  If they reach
        the end of the interval the problem is not stiff at this scale and the explicit pair carries on.
      * otherwise the step size the controller settled on tells how many explicit steps the interval would cost
        ((t1 - t0) / min h_last: the explicit pair sits at its stability limit ~3.3/|lambda| on a stiff problem).
        Up to `auto-max-explicit-steps` the interval is redone with the explicit pair, beyond that the
        Rosenbrock stepper takes over - for good (chemistry gets stiffer as the bed heats up, not milder).
      * an explicit interval that exhausts its step budget is redone with the stiff stepper as well, and one that
        needed more than `auto-max-explicit-steps` steps is the last explicit one.

    Two devices (the on-chip RK45 geometry and the Rosenbrock kernel family) share the state tensor."""

    def __init__(self, dev_rk45, dev_ros4):
        """dev_ros4: the stiff device, or a zero-argument factory for it (single-process runs build - and JIT-
        compile - the Rosenbrock kernel family only when the problem turns out to be stiff)."""
        self.d45, self._dr = dev_rk45, dev_ros4
        self.mode = None                    # None: undecided, "rk45", "ros4"
        self.last = dev_rk45
        self.rhs_evals = 0
        self.choices = []
        self.jacobian_evals = dev_rk45.jacobian_evals
        self.prev_steps = 0
        self._started = {"rk45": False, "ros4": False}

    @property
    def dr(self):
        if callable(self._dr):
            self._dr = self._dr()
        return self._dr

    def to_device(self, y):
        return self.d45.to_device(y)

    def close(self):
        self.d45.close()
        if not callable(self._dr):
            self._dr.close()

    def rk45_stats(self):
        return self.last.rk45_stats()

    def raise_on_flags(self):
        self.last.raise_on_flags()

    def _run(self, which, y, t0, t1, cfg, max_steps):
        dev = self.d45 if which == "rk45" else self.dr
        first = not self._started[which]
        self._started[which] = True
        if which == "rk45":
            h0 = float(cfg.get('h0', DEVICE_DEFAULTS['rk45-h0']))
            dev.rk45(y, t0, t1, float(cfg.get('rtol', DEVICE_DEFAULTS['auto-rk45-rtol'])),
                     float(cfg.get('atol', DEVICE_DEFAULTS['auto-rk45-atol'])), h0 if first else -h0, int(max_steps))
        else:
            h0 = float(cfg.get('h0', DEVICE_DEFAULTS['ros4-h0']))
            dev.ros4(y, t0, t1, float(cfg.get('rtol', DEVICE_DEFAULTS['ros4-rtol'])),
                     float(cfg.get('atol', DEVICE_DEFAULTS['ros4-atol'])), h0 if first else -h0, int(max_steps))
        self.last = dev
        return dev

    def _explicit(self, y, t0, t1, cfg, budget):
        """One explicit attempt; True when every reactor reached t1 (other failures are left for raise_on_flags)."""
        dev = self._run("rk45", y, t0, t1, cfg, budget)
        st = dev.rk45_stats()
        tried = st["accepted"] + st["rejected"]
        self.rhs_evals += int(np.sum(6*tried) + dev.E)
        done = bool(np.all(st["t_end"] >= t1))
        if not done:
            flags = dev.status()            # read and clear: only the step budget may be set
            other = flags & ~np.uint32(FLAG_STEP)
            if other.any():                 # a genuine failure: put it back for the caller to raise
                raise _flag_error(int(other[np.nonzero(other)[0][0]]), int(np.nonzero(other)[0][0]), dev.E)
        return done, st

    def advance(self, y, t0, t1, cfg):
        hard_max = int(cfg.get('max-steps', DEVICE_DEFAULTS['rk45-max-steps']))
        if self.mode != "ros4":
            backup = y.clone()
            if self.mode is None:
                done, st = self._explicit(y, t0, t1, cfg, min(hard_max, DEVICE_DEFAULTS['auto-probe-steps']))
                if done:
                    self.mode = "rk45"
                else:
                    hmin = float(np.min(st["h_last"]))
                    est = (t1 - t0)/max(hmin, 1e-300)
                    y.copy_(backup)
                    self._started["rk45"] = False
                    self.mode = "rk45" if est <= DEVICE_DEFAULTS['auto-max-explicit-steps'] else "ros4"
                    if self.mode == "rk45":
                        done, st = self._explicit(y, t0, t1, cfg, min(hard_max, int(4*est) + 200))
                        if not done:
                            y.copy_(backup)
                            self.mode = "ros4"
            else:
                done, st = self._explicit(y, t0, t1, cfg, min(hard_max, 4*self.prev_steps + 200))
                if not done:
                    y.copy_(backup)
                    self.mode = "ros4"
            if self.mode == "rk45":
                self.prev_steps = int(np.max(st["accepted"] + st["rejected"]))
                self.choices.append("rk45")
                if self.prev_steps > DEVICE_DEFAULTS['auto-max-explicit-steps']:
                    self.mode = "ros4"          # the interval just taken was too expensive explicitly: hand over
                return
        dev = self._run("ros4", y, t0, t1, cfg, hard_max)
        st = dev.rk45_stats()
        self.rhs_evals += int(np.sum((6 + dev.jacobian_evals)*(st["accepted"] + st["rejected"])))
        self.choices.append("ros4")


def _flag_error(f, idx, E):
    where = "reactor %d of %d (flags=0x%x)" % (idx, E, f)
    if f & FLAG_DOMAIN:
        return ValueError("math domain error in " + where)
    if f & FLAG_DIV0:
        return ZeroDivisionError("float division by zero in " + where)
    if f & FLAG_OVERFLOW:
        return OverflowError("math range error in " + where)
    if f & FLAG_NONFINITE:
        return FloatingPointError("state became NaN/Inf in " + where + " - step size too large?")
    return RuntimeError("device error in " + where)


# --------------------------------------------------------------------------- result packing
def pack_interval(Y, named, mech, zNo, t_end, modelId):
    """One dataPack entry (pbHomoReactor.py:3630-3678; sortResult5, solResultAnalysis.py:252-301)."""
    S = mech.S
    Y = np.reshape(np.asarray(Y, dtype=np.float64), (mech.V, zNo))
    conc_dl = Y[:-1] if not mech.iso else Y[:]
    # dataYCons1 is dataYs_Reshaped[:-1] for BOTH process types in the reference (:3636): with V = S
    # (iso-thermal) that silently drops the last species; the schema is the spec, so it is mirrored
    cons1 = Y[:-1]
    temp_dl = Y[-1] if not mech.iso else np.repeat(0, zNo).reshape(zNo)
    conc = conc_dl*named["Cmax"]
    T_dl_row = Y[-1, :].reshape((1, zNo)) if not mech.iso else np.repeat(0, zNo).reshape((1, zNo))
    Treal = T_dl_row*named["Tf"] + named["Tf"]
    mofr = conc/np.sum(conc, axis=0)
    labelList = list(mech.compList) + ["Temperature"]
    return {
        "modelId": modelId, "processType": mech.processType, "successStatus": True,
        "dataShape": np.array(t_end).shape, "labelList": labelList, "indexList": [S, S + 1, S],
        "dataTime": t_end, "dataXs": np.linspace(0, 1, zNo),
        "dataYCons1": cons1, "dataYCons2": conc, "dataYTemp1": temp_dl, "dataYTemp2": Treal,
        "dataYs": np.concatenate((mofr, Treal), axis=0),
    }


def pack_intervals(Yg, named, mech, zNo, t_end, modelId):
    """pack_interval for EVERY member of an ensemble at once (Yg: [E][V*zNo]) - the arithmetic of sortResult5 as
    array operations over the member axis; entry e equals pack_interval(Yg[e], named[e], ...) bit for bit (same
    elementwise operations, the species sum taken in the same order).  A 2048-member sweep packs in ~10 ms per output
    time instead of 2048 Python-level passes (27 us each: several times the 50 ms the device needs for the sweep)."""
    S, E = mech.S, len(named)
    Y = np.reshape(np.asarray(Yg, dtype=np.float64), (E, mech.V, zNo))
    cmax = np.array([nm["Cmax"] for nm in named], dtype=np.float64).reshape(E, 1, 1)
    tf = np.array([nm["Tf"] for nm in named], dtype=np.float64).reshape(E, 1, 1)
    conc_dl = Y[:, :-1] if not mech.iso else Y
    conc = conc_dl*cmax
    if mech.iso:
        T_dl = np.zeros((E, 1, zNo), dtype=np.int64)          # np.repeat(0, zNo) of the reference: integer zeros
    else:
        T_dl = Y[:, -1:, :]
    Treal = T_dl*tf + tf
    mofr = conc/np.sum(conc, axis=1, keepdims=True)
    dataYs = np.concatenate((mofr, Treal), axis=1)
    labelList = list(mech.compList) + ["Temperature"]
    xs = np.linspace(0, 1, zNo) if zNo > 1 else np.array([1.0])
    shape = np.array(t_end).shape
    return [{
        "modelId": modelId, "processType": mech.processType, "successStatus": True,
        "dataShape": shape, "labelList": list(labelList), "indexList": [S, S + 1, S],
        "dataTime": t_end, "dataXs": xs,
        "dataYCons1": Y[e, :-1], "dataYCons2": conc[e], "dataYTemp1": T_dl[e, 0], "dataYTemp2": Treal[e],
        "dataYs": dataYs[e],
    } for e in range(E)]


def _progress(i, total, quiet):
    if quiet:
        return
    pct = ("{0:.1f}").format(100*(i/float(total)))
    filled = int(50*i//total)
    print('\rProgress: |%s| %s%% Complete' % ('█'*filled + '-'*(50 - filled), pct),
          end="\r" if i < total else "\n")


def compile_mechanism(mech, N, fp32=False, block=None, npt=None, lds_state=None, defines=None,
                      arch="gfx950", E=None, extra_opts=""):
    """Code object for (mechanism, mesh size, members per rank) - what ensemble rank 0 compiles
    and broadcasts; pass the same E/block/npt/lds_state/defines to N2Device(code=...)."""
    b, n = choose_geometry(N, mech.V, fp32, E)
    block, npt = int(block or b), int(npt or n)
    defines, lds_state = kcache_choice(mech, N, fp32, block, npt, lds_state, defines)
    tpl = hipbind.kernel_template()
    return hipbind.compile_cached(mech.source(tpl, fp32, block, npt, lds_state, defines),
                                  mech.digest(tpl, fp32, block, npt, lds_state, defines), arch,
                                  compile_options(block, npt, (), extra_opts, defines))


def resolve_ivp(ivp):
    """the reference's "default" is SciPy's LSODA (pbHomoReactor.py:3576) - Adams / BDF with AUTOMATIC
    stiffness detection: it maps to "hip-auto" (AutoStepper below: explicit Dormand-Prince while the problem is
    not stiff, the Rosenbrock stepper when it is); SciPy's always-implicit choices BDF / Radau map to the
    device's stiff Rosenbrock stepper, the explicit pairs to the device Dormand-Prince stepper."""
    ivp = {"default": "hip-auto", "LSODA": "hip-auto", "BDF": "hip-ros4", "Radau": "hip-ros4",
           "RK45": "hip-rk45", "RK23": "hip-rk45", "DOP853": "hip-rk45"}.get(ivp, ivp)
    if ivp not in DEVICE_IVPS:
        raise ValueError("`ivp` must be one of %s, 'default' or a SciPy method name (got %r)"
                         % (DEVICE_IVPS, ivp))
    return ivp


def device_arch():
    """Architecture the JIT targets: the visible GPU's, gfx950 when there is none (cross-compile)."""
    try:
        import torch
        if torch.cuda.is_available():
            return torch.cuda.get_device_properties(torch.cuda.current_device()).gcnArchName.split(":")[0]
    except Exception:
        pass
    return "gfx950"


def device_cls():
    """The device class the ensemble path instantiates (looked up at call time: the gloo CPU tests put a
    host-emulation stand-in here, tests/emu_device.py)."""
    return N2Device


def mechanism_for(modelInput, inputs, cfg):
    """The ONE compiled mechanism of a launch: the base input's, with every scalar VARS constant that differs
    between the members (ensemble.member_parameters - which also refuses members that differ in anything the
    member row cannot express) or that solver-config "vars-as-parameters" names kept as a per-reactor parameter."""
    params = list(cfg.get('vars-as-parameters', ()))
    if len(inputs) > 1 or inputs[0] is not modelInput:
        from .ensemble import member_parameters
        varying = member_parameters(modelInput, inputs)
        VARS = modelInput['reaction-rates']['VARS']
        params = [k for k in VARS if k in params or k in varying] + [k for k in params if k not in VARS]
    return plan.Mechanism(modelInput, params=params)


def open_members(mech, inputs, zNo, pack, init, sync=None, fp32=False, block=None, npt=None, defines=None,
                 features=()):
    """Device + initial state for the members THIS process integrates.

    Single process: all of ``inputs``.  As one rank of a torch.distributed job (``sync``, see
    ensemble.RankSync): the rank's contiguous block; rank 0 compiles, every rank loads the broadcast
    code object, sweep-invariant member fields agreed over all ranks become kernel literals.
    Returns (device, named constants of the local members, local initial states [E_local][V*N])."""
    if sync is None:
        pairs = [pack(mi, mech, zNo) for mi in inputs]
        rows = np.array([r for _, r in pairs])
        IV = plan.initial_states([nm for nm, _ in pairs], mech, zNo, init)
        dev = device_cls()(mech, rows, zNo, fp32=fp32, block=block, npt=npt, defines=defines, features=features)
        return dev, [nm for nm, _ in pairs], IV
    # Multi-rank: every rank-LOCAL phase (packing, the rank-0 compile inside DistributedEnsemble, loading the
    # module and allocating on the device) runs under ensemble.guarded / agree: a failure on one rank is raised
    # on every rank instead of leaving the others in the next collective until the backend's timeout.
    from .ensemble import DistributedEnsemble, guarded
    E_geo = max(sync.counts)                       # one geometry for all ranks (block sizes differ by <= 1 member)
    b, n = choose_geometry(zNo, mech.V, fp32, E_geo)
    block, npt = int(block or b), int(npt or n)
    defs = dict(defines or {})
    for f in features:
        defs[FEATURE_DEFINES[f]] = "1"
    if "ros4" in features and "RMT_ROS_QUAD" not in defs and ros4_quad(mech, fp32) and npt == 1:
        defs["RMT_ROS_QUAD"] = "1"
    if getattr(mech, "model", "N2") == "M2" and "RMT_M2_NEWTON" not in defs:
        sweeps = guarded(sync, lambda: plan.m2_newton_sweeps(
            np.array([pack(mi, mech, zNo)[1] for mi in inputs[sync.lo:sync.hi]]), mech, zNo))
        defs["RMT_M2_NEWTON"] = str(sync.max_int(sweeps))
    arch = device_arch()
    ens = DistributedEnsemble(
        mech, inputs, zNo, group=sync.group, device=sync.device,
        compile_fn=lambda mdef: compile_mechanism(mech, zNo, fp32, block, npt, None, {**defs, **mdef}, arch, E_geo))
    dev = guarded(sync, device_cls(), mech, ens.rows, zNo, fp32=fp32, block=block, npt=npt,
                  defines={**defs, **ens.member_defines}, specialize=False, code=ens.code, features=features)
    return dev, ens.named, ens.IV


def open_auto(mech, inputs, zNo, pack, init, sync, fp32, defines, block=None, npt=None):
    """The two devices of ivp "hip-auto" (explicit pair in its on-chip geometry, Rosenbrock family) behind one
    AutoStepper; an explicit `block` / `nodes-per-thread` of the solver-config applies to the explicit device."""
    if block is None:
        b45, n45, d45 = rk45_geometry(mech.V, zNo, fp32, E=len(inputs) if sync is None else max(sync.counts))
    else:
        b45, n45, d45 = block, npt, {}
    dev45, named_local, IV = open_members(mech, inputs, zNo, pack, init, sync, fp32=fp32, block=b45, npt=n45,
                                          defines={**(defines or {}), **d45})
    def make_ros4():
        return open_members(mech, inputs, zNo, pack, init, sync, fp32=fp32, block=ros4_block(mech.V, zNo, fp32),
                            npt=1, defines=defines, features=("ros4",))[0]
    if sync is None:
        return AutoStepper(dev45, make_ros4), named_local, IV       # built only if the problem turns out stiff
    try:        # multi-rank: creation involves collectives and the ranks may decide differently -> build it now
        devr = make_ros4()
    except Exception:
        dev45.close()
        raise
    return AutoStepper(dev45, devr), named_local, IV


def finish_stats(stats, ivp, n_members, tNo, zNo, jacobian_evals):
    """Totals of the device-stats record from the per-member step counts."""
    if stats.get("accepted") is not None:
        # adaptive steppers: "steps" is the sum over the members; per attempted step the Dormand-Prince
        # pair costs 6 RHS evaluations (FSAL; +1 for the first step of a launch), RODAS4 6 stage
        # evaluations + the node Jacobian (jacobian_evals node-function evaluations)
        tried = stats["accepted"] + stats["rejected"]
        stats["steps"] = int(np.sum(stats["accepted"]))
        if ivp == "hip-rk45":
            stats["rhs_evals"] = int(np.sum(6*tried) + n_members*tNo)
        else:
            stats["rhs_evals"] = int(np.sum((6 + jacobian_evals)*tried))
        stats["node_steps"] = stats["steps"]*zNo
    else:
        stats["node_steps"] = stats["steps"]*zNo*n_members
    return stats


def gather_stats(stats, sync, ivp, tNo, zNo, jacobian_evals):
    """Rank 0: the record for the WHOLE ensemble (per-member step counts gathered); other ranks keep theirs."""
    if stats.get("accepted") is not None:
        acc, rej = sync.gather(stats["accepted"]), sync.gather(stats["rejected"])
        if acc is not None:
            stats["accepted"], stats["rejected"] = acc, rej
    n = sync.n_total if sync.rank == 0 else sync.hi - sync.lo
    stats["ranks"] = sync.world
    return finish_stats(stats, ivp, n, tNo, zNo, jacobian_evals)


def outlet_only(cfg):
    """solver-config "ensemble-output": "outlet" - only the outlet node of every member leaves the device (and, in a
    multi-rank job, travels to rank 0) per output time: E x V doubles instead of the E x V x zNo state (117 MB per
    output time for 2048 x 1024 nodes).  The dataPack entries keep their schema with ONE axial column (dataXs = [1])."""
    out = cfg.get('ensemble-output', 'profile')
    if out not in ('profile', 'outlet'):
        raise ValueError("solver-config 'ensemble-output' must be 'profile' or 'outlet' (got %r)" % (out,))
    return out == 'outlet'


PIPELINE_BYTES = 1 << 30       # pinned host memory one batch of queued output intervals may hold (integrate_intervals)


def integrate_intervals(dev, y, cfg, ivp, opTSpan, n_members, zNo, quiet, on_interval, sync=None, outlet=False):
    """The reference's time loop (pbHomoReactor.py:3589-3690, pbReactor.py:711-762): one device
    launch per output interval; ``on_interval(i, t1, Y_host)`` packs the end state ([E][V*zNo], or [E][V] = the
    outlet node with ``outlet``).  With ``sync`` (multi-rank ensemble) a failure on any rank is raised on every
    rank before the next gather."""
    tNo = len(opTSpan) - 1
    stats = {"steps": 0, "rhs_evals": 0, "node_steps": 0, "accepted": None, "rejected": None}
    _progress(0, tNo + 1, quiet)
    def launch(i, t0, t1):
        if ivp == "hip-rk4":
            dt_req = float(cfg.get('dt', DEVICE_DEFAULTS['rk4-dt']))
            n = max(1, int(round((t1 - t0)/dt_req)))
            dev.rk4(y, (t1 - t0)/n, n, t0)
            stats["steps"] += n
            stats["rhs_evals"] += 4*n
        elif ivp in ("AM", "hip-ab3"):
            # the reference's plug point: PreCorr3 with n fixed steps per output interval,
            # n from solverSetting['T1']['ode-solver']['PreCorr3']['n'] (pbHomoReactor.py:3572,3601)
            n = int(cfg.get('n', solverSetting['T1']['ode-solver']['PreCorr3']['n']))
            dev.multistep(y, abs(t1 - t0)/n, n, "PreCorr3" if ivp == "AM" else "AdBash3", t0)
            stats["steps"] += n
            stats["rhs_evals"] += (2*n + 6) if ivp == "AM" else (n + 8)
        elif ivp == "hip-auto":
            dev.advance(y, t0, t1, cfg)
        elif ivp == "hip-ros4":
            # first interval: h0 for every reactor; later ones RESUME (h0 < 0): each reactor starts from
            # the step its own controller proposed at the end of the previous interval (stats.h_last)
            h0 = float(cfg.get('h0', DEVICE_DEFAULTS['ros4-h0']))
            dev.ros4(y, t0, t1, float(cfg.get('rtol', DEVICE_DEFAULTS['ros4-rtol'])),
                     float(cfg.get('atol', DEVICE_DEFAULTS['ros4-atol'])), h0 if i == 0 else -h0,
                     int(cfg.get('max-steps', DEVICE_DEFAULTS['rk45-max-steps'])))
        else:
            h0 = float(cfg.get('h0', DEVICE_DEFAULTS['rk45-h0']))
            dev.rk45(y, t0, t1, float(cfg.get('rtol', DEVICE_DEFAULTS['rk45-rtol'])),
                     float(cfg.get('atol', DEVICE_DEFAULTS['rk45-atol'])), h0 if i == 0 else -h0,
                     int(cfg.get('max-steps', DEVICE_DEFAULTS['rk45-max-steps'])))
        if check:
            dev.raise_on_flags()

    check = True
    adaptive = ivp in ("hip-rk45", "hip-ros4", "hip-auto")
    # One process, a stepper that needs no host decision between the intervals: the launches of SEVERAL output intervals
    # are queued back to back, each followed in the stream by the copy of its end state into pinned host memory (and of
    # its step counters), and the host packs while the device integrates.  With a synchronisation per interval the device
    # idled - and clocked down - while the host packed: 0.18 s for the bench's 256-member sweep against 0.04 s of kernel
    # time.  The status words are sticky, so one look at the end of a batch raises what any of its launches flagged.
    # Batches are bounded by PIPELINE_BYTES of pinned memory.
    if sync is None and ivp != "hip-auto" and tNo > 1 and getattr(y, "is_cuda", False) and hasattr(dev, "_stats"):
        import torch
        E_loc = y.shape[0]
        per = E_loc*(y.shape[1]//zNo if outlet else y.shape[1])*y.element_size()
        batch = int(max(1, min(tNo, PIPELINE_BYTES//max(per, 1))))
        check = False
        for lo in range(0, tNo, batch):
            hi = min(tNo, lo + batch)
            hosts, counters, landed = [], [], []
            for i in range(lo, hi):
                t0, t1 = float(opTSpan[i]), float(opTSpan[i + 1])
                _progress(i + 1, tNo + 1, quiet)
                launch(i, t0, t1)
                src = y.reshape(E_loc, -1, zNo)[:, :, zNo - 1] if outlet else y
                host = torch.empty(src.shape, dtype=src.dtype, pin_memory=True)
                host.copy_(src, non_blocking=True)               # stream-ordered: reads y before the next launch writes it
                hosts.append(host)
                if adaptive:
                    c = torch.empty(dev._stats.shape, dtype=dev._stats.dtype, pin_memory=True)
                    c.copy_(dev._stats, non_blocking=True)
                    counters.append(c)
                ev = torch.cuda.Event()
                ev.record()
                landed.append(ev)
            for k, i in enumerate(range(lo, hi)):                 # packing interval i while the device is at i+1, i+2, ...
                landed[k].synchronize()
                if adaptive:
                    raw = counters[k].numpy()
                    acc, rej = raw[:, 2].copy().view(np.int64), raw[:, 3].copy().view(np.int64)
                    stats["accepted"] = acc if stats["accepted"] is None else stats["accepted"] + acc
                    stats["rejected"] = rej if stats["rejected"] is None else stats["rejected"] + rej
                on_interval(i, float(opTSpan[i + 1]), hosts[k].numpy().astype(np.float64))
            dev.raise_on_flags()                                  # (sticky status words: whatever a launch of the batch flagged)
        return finish_stats(stats, ivp, n_members, tNo, zNo, dev.jacobian_evals)

    for i in range(tNo):
        t0, t1 = float(opTSpan[i]), float(opTSpan[i + 1])
        _progress(i + 1, tNo + 1, quiet)
        if sync is None:
            launch(i, t0, t1)
        else:                                   # whatever goes wrong on one rank is raised on every rank
            err = None
            try:
                launch(i, t0, t1)
            except Exception as e:              # noqa: BLE001 - re-raised on every rank by agree()
                err = e
            sync.agree(err)
        if adaptive:
            st = dev.rk45_stats()
            stats["accepted"] = st["accepted"] if stats["accepted"] is None else stats["accepted"] + st["accepted"]
            stats["rejected"] = st["rejected"] if stats["rejected"] is None else stats["rejected"] + st["rejected"]
        if outlet:
            E_loc = y.shape[0]
            Yh = y.reshape(E_loc, -1, zNo)[:, :, zNo - 1].contiguous().cpu().numpy().astype(np.float64)
        else:
            Yh = y.cpu().numpy().astype(np.float64)
        on_interval(i, t1, Yh)
    if ivp == "hip-auto":
        stats["method-per-interval"] = list(dev.choices)
    if sync is not None:
        stats = gather_stats(stats, sync, ivp, tNo, zNo, dev.jacobian_evals)
    else:
        stats = finish_stats(stats, ivp, n_members, tNo, zNo, dev.jacobian_evals)
    if ivp == "hip-auto":
        stats["rhs_evals"] = dev.rhs_evals          # includes the probe and any abandoned explicit attempt
    return stats


def run_n2(modelInput, members_inputs=None):
    """runN2 on the device.  ``members_inputs``: optional list of modelInput dicts (one per
    ensemble member, same mechanism); default = the single reactor described by modelInput."""
    start = timer()
    cfg = modelInput['solver-config']
    displayResult = cfg['display-result'] == "True"        # KeyError like the reference (:3337)
    ivp = resolve_ivp(cfg['ivp'])
    zNo = int(cfg.get('zNo', solverSetting['N2']['zNo']))
    tNo = int(cfg.get('tNo', solverSetting['N2']['tNo']))
    fp32 = cfg.get('dtype', 'fp64') in ('fp32', 'float32')
    quiet = bool(cfg.get('quiet', False))
    opT = modelInput['operating-conditions']['period']
    modelId = modelInput['model']

    plan.check_model_setting_n2()          # the reference's N2 RHS raises under any setting but "MAX" - so does this
    inputs = list(members_inputs) if members_inputs else [modelInput]
    mech = mechanism_for(modelInput, inputs, cfg)
    from .ensemble import active_ranks, guarded
    sync = active_ranks(len(inputs)) if members_inputs else None       # one rank of a torchrun job?
    block, npt = cfg.get('block'), cfg.get('nodes-per-thread')
    if ivp == "hip-ros4" and block is None:
        block, npt = ros4_block(mech.V, zNo, fp32), 1
    # "strict-flags": test the Python-exception conditions on every RK stage (default: stage 1 only)
    defines = {"RMT_CHECK_ALL_STAGES": "1"} if cfg.get('strict-flags') else {}
    if ivp == "hip-rk45" and block is None:
        block, npt, geo_defs = rk45_geometry(mech.V, zNo, fp32, E=len(inputs) if sync is None else max(sync.counts))
        defines.update(geo_defs)
    if ivp == "hip-auto":
        dev, named_local, IV = open_auto(mech, inputs, zNo, plan.member_constants, plan.initial_state, sync, fp32,
                                         defines, block, npt)
    else:
        dev, named_local, IV = open_members(mech, inputs, zNo, plan.member_constants, plan.initial_state, sync,
                                            fp32=fp32, block=block, npt=npt, defines=defines,
                                            features=("ros4",) if ivp == "hip-ros4" else ())
    # the process that returns the results (rank 0, or the only one) packs EVERY member
    packer = sync is None or sync.rank == 0
    try:
        if sync is None:
            named = named_local
        else:
            named = guarded(sync, lambda: [plan.member_constants(mi, mech, zNo)[0] for mi in inputs] if packer else [])
        y = guarded(sync, dev.to_device, IV)
        packs = [[] for _ in named]

        outlet = outlet_only(cfg) if members_inputs else False

        def on_interval(i, t1, Yh):
            Yg = Yh if sync is None else sync.gather(Yh)               # [E_total][V*N] (or [V]: outlet) on rank 0
            if Yg is not None:
                if len(named) == 1 and not outlet:
                    packs[0].append(pack_interval(Yg[0], named[0], mech, zNo, t1, modelId))
                else:
                    for e, pk in enumerate(pack_intervals(Yg, named, mech, 1 if outlet else zNo, t1, modelId)):
                        packs[e].append(pk)
        stats = integrate_intervals(dev, y, cfg, ivp, np.linspace(0, opT, tNo + 1), len(named_local),
                                    zNo, quiet or not packer, on_interval, sync, outlet)
    finally:
        dev.close()
    elapsed = np.round(timer() - start, ROUND_FUN_ACCURACY)
    resPack = {"computation-time": elapsed, "dataPack": packs[0] if packs else [], "device-stats": stats}
    if members_inputs:
        # multi-rank: rank 0 holds the whole sweep, the other ranks None (and an empty dataPack)
        resPack["ensemble"] = [{"dataPack": p} for p in packs] if packer else None
    if sync is not None:
        resPack["ensemble-shard"] = {"rank": sync.rank, "world": sync.world, "members": [sync.lo, sync.hi]}
    if displayResult and packer:
        from .plotting import plot_results_dynamic
        plot_results_dynamic(resPack, tNo)
    return resPack
