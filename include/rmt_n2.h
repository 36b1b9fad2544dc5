/*
 * rmt_n2.h - C ABI of the MI355X-native integrator for PyREMOT's dynamic packed-bed model "N2".
 *
 * The reference (sinagilassi/rmt-app, pure Python) has no FFI; its boundary for this path is the
 * Python-level contract listed in SURVEY.md section 8(b).  Each entry point below names the
 * reference interface it replaces (paths relative to the reference repo):
 *
 *   rmt_n2_compile      - (new) lowers nothing itself: turns the host-generated kernel source
 *                         (lowered `reaction-rates` lambdas + mechanism tables) into a gfx950 code
 *                         object.  Replaces the per-call Python evaluation of the lambdas in
 *                         reactionRateExe, PyREMOT/docs/rmtReaction.py:11-61.
 *   rmt_n2_create       - the setup half of PackedBedHomoReactorClass.runN2,
 *                         PyREMOT/docs/pbHomoReactor.py:3334-3580 (paramsSet construction): takes
 *                         the packed per-reactor constants instead of nested dicts.
 *   rmt_n2_rhs          - PackedBedHomoReactorClass.modelEquationN2(t, y, paramsSet),
 *                         PyREMOT/docs/pbHomoReactor.py:3706-4134 (one RHS evaluation).
 *   rmt_n2_rk4          - RK4(t0, tn, n, y0, f, params), PyREMOT/solvers/odeSolver.py:17-40, as used
 *                         from the `ivp == "AM"` plug point pbHomoReactor.py:3598-3607 (only the
 *                         last column - the end state - is produced, which is all runN2 keeps,
 *                         :3630, :3685).
 *   rmt_n2_multistep    - AdBash3 / PreCorr3, PyREMOT/solvers/odeSolver.py:43-102; PreCorr3 is what
 *                         runN2 calls for ivp == "AM" (pbHomoReactor.py:3598-3601).
 *   rmt_n2_rk45         - scipy.integrate.solve_ivp(funSet, t, IV, method=..., args=(paramsSet,))
 *                         at pbHomoReactor.py:3609-3610, restricted to an explicit embedded pair
 *                         (Dormand-Prince 5(4)) with per-reactor step control.
 *   rmt_n2_ros4         - the same solve_ivp call site with a STIFF method (the reference's default is
 *                         LSODA, pbHomoReactor.py:3576): linearly-implicit Rosenbrock method RODAS4
 *                         (6 stages, order 4(3), L-stable; Hairer & Wanner) with per-reactor step
 *                         control; SURVEY.md section 8(f) rank 2.  Also integrates model M2
 *                         (pbReactor.py:719) when the code object was generated for it.
 *   rmt_n1_profile      - PackedBedHomoReactorClass.runN1 / modelEquationN1 (pbHomoReactor.py:2694-3314):
 *                         the steady-state model N1, integrated along z* for E reactors at once
 *                         (one per lane) with the same Rosenbrock scheme; SURVEY.md 8(f) rank 1.
 *   (model M2)          - PackedBedReactorClass.runM2 / modelEquationM2 (pbReactor.py:552-1165) uses the
 *                         SAME entry points: the generated prelude selects its node functions
 *                         (RMT_MODEL 2), state rows are kmol/m^3 and K; SURVEY.md 8(f) rank 3.
 *   rmt_n2_status       - the exceptions Python raises inside the user lambdas / `raise` at
 *                         pbHomoReactor.py:3614-3626, as per-reactor flag words.
 *
 * Conventions: every function returns 0 on success, non-zero on error (rmt_n2_last_error() gives
 * a thread-local message).  `y`, `dydt` and `flags` are DEVICE pointers owned by the caller;
 * plan contents are HOST memory, copied during rmt_n2_create.  Work is enqueued on the stream set
 * with rmt_n2_set_stream (default: the null stream) and is NOT synchronised by these calls.
 * State layout: y[E][V][N] = the reference's row-major flattening of the (V, N) matrix
 * (pbHomoReactor.py:3483-3497, 3873) for each of E independent reactors; V = S (+1 unless
 * iso-thermal); real = double, or float when the code object was generated with fp32.
 * A handle is bound to the device that was current at create time; handles are not thread-safe,
 * distinct handles are independent.  No global state besides the last-error string.
 */
#ifndef RMT_N2_H
#define RMT_N2_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RMT_N2_ABI_VERSION 2

/* status bits written by the kernels (OR-ed per reactor) */
#define RMT_N2_FLAG_DOMAIN 1u    /* Python: ValueError("math domain error")          */
#define RMT_N2_FLAG_DIV0 2u      /* Python: ZeroDivisionError                         */
#define RMT_N2_FLAG_OVERFLOW 4u  /* Python: OverflowError("math range error")         */
#define RMT_N2_FLAG_NONFINITE 8u /* a state/derivative became NaN or Inf              */
#define RMT_N2_FLAG_STEP 16u     /* rk45: step size underflow / max steps exceeded    */
#define RMT_N2_FLAG_PRESSURE 32u /* model M2: Newton sweeps of the pressure march did not converge */
/* Contract of the three Python-exception bits: the explicit steppers test the conditions on which the
 * reference's lambdas would raise on the FIRST stage of every step (f(y_n)); an exception that exists
 * only at a trial-stage state surfaces as RMT_N2_FLAG_NONFINITE when it poisons the state (the lean
 * fp64 division and exp of the kernels turn an intermediate inf into NaN - 1/(1+inf) is NaN, not 0 - so
 * the usual ways of mapping an overflow back to a finite rate do poison it).  Code objects
 * generated with RMT_CHECK_ALL_STAGES=1 (solver-config "strict-flags") test every stage.
 * Every entry point taking a handle runs on the device that was current at rmt_n2_create and
 * restores the caller's current device before returning. */

/* chained steppers (rmt_n2_ros4 / rmt_n2_rk45 / rmt_n2_rk4 with one reactor over several CUs): ring depth of a tagged-word link
 * (= RMT_RING of csrc/kernels/30_lanes_links.inc) and the largest number of chunks one reactor is cut into */
#define RMT_N2_RING 512
#define RMT_N2_MAX_CHUNKS 64

/* member row layout: doubles per reactor = 16 + S + NU (see rmt_app_amd/csrc/kernels/00_config_math.inc M_*):
 * 16 fixed operating-point scalars, S inlet values, then the NU = plan.n_user_params scalar constants of the
 * user's reaction-rates.VARS that differ between the reactors of the ensemble (the reference copies VARS into
 * the rate lambdas' namespace on every call, PyREMOT/docs/rmtReaction.py:44-51; the generated kinetics read
 * them as U[k]) */
#define RMT_N2_MEMBER_FIXED 16

typedef struct rmt_n2_plan {
    int32_t abi_version;     /* RMT_N2_ABI_VERSION */
    int32_t n_species;       /* S */
    int32_t n_reactions;     /* R (informational) */
    int32_t n_vars;          /* V = S or S+1 */
    int32_t n_nodes;         /* N (zNo) */
    int32_t n_members;       /* E */
    int32_t fp32;            /* 1: real = float */
    int32_t block;           /* RMT_BLOCK the code object was generated with */
    int32_t nodes_per_thread;/* RMT_NPT the code object was generated with */
    int32_t n_user_params;   /* NU (RMT_NU of the code object; 0 = every VARS constant is a literal of the kernel) */
    int32_t ros4_nodes_per_block; /* mesh nodes one workgroup of the stiff stepper covers: 0 = `block` (one node per lane);
                              * block / 4 for code objects generated with RMT_ROS_QUAD (one node on four lanes, the
                              * layout for mechanisms wider than 8 variables, kernels/61_ros4_quad.inc) */
    int32_t reserved;
    const void* code_object; /* gfx950 code object from rmt_n2_compile (host memory) */
    size_t code_size;
    const double* members;   /* host [E][16+S+NU] packed constants */
} rmt_n2_plan;

typedef struct rmt_n2_handle rmt_n2_handle;

typedef struct rmt_n2_stats {   /* per reactor, written by rmt_n2_rk45 (device memory) */
    double t_end;
    double h_last;
    int64_t accepted;
    int64_t rejected;
} rmt_n2_stats;

/* hipRTC: source text -> code object for `arch` (e.g. "gfx950"); works without a GPU.
 * *code is malloc'ed (free with rmt_n2_free); *log (may be NULL) receives the compiler log. */
int rmt_n2_compile(const char* source, const char* arch, const char* extra_opts, void** code,
                   size_t* code_size, char** log);
void rmt_n2_free(void* p);
/* the device template that rmt_n2_compile expects to follow the generated prelude */
const char* rmt_n2_kernel_template(void);
/* path of the hipRTC library this process compiles with (a process that loaded PyTorch first uses the one
 * PyTorch bundles, otherwise /opt/rocm's; both report version 9.0 but generate different code): part of the key
 * of any code-object cache */
const char* rmt_n2_hiprtc_path(void);
/* the options rmt_n2_compile adds to every compilation besides --offload-arch and `extra_opts` (an `extra_opts`
 * that sets -mllvm ...machine-licm... itself replaces the default of that switch): the other part of a cache key */
const char* rmt_n2_compile_options(void);

int rmt_n2_create(const rmt_n2_plan* plan, rmt_n2_handle** out);
void rmt_n2_destroy(rmt_n2_handle* h);
int rmt_n2_set_stream(rmt_n2_handle* h, void* hip_stream);
/* replace the per-member constants (same E) without recompiling; fields that the code object's
 * prelude baked in as literals (#define RMT_MC_<FIELD>) are not affected */
int rmt_n2_set_members(rmt_n2_handle* h, const double* members);

int rmt_n2_rhs(rmt_n2_handle* h, double t, const void* y, void* dydt);
int rmt_n2_rk4(rmt_n2_handle* h, void* y_inout, double t0, double dt, int64_t nsteps);
/* method: 0 = AdBash3, 1 = PreCorr3 (needs nsteps >= 3, like the reference) */
int rmt_n2_multistep(rmt_n2_handle* h, void* y_inout, double t0, double dt, int64_t nsteps, int method);
/* h0 > 0: first step of every reactor.  h0 < 0: "resume" - every reactor starts from the h_last
 * its previous launch left in stats_out[e] (|h0| where that is not a positive finite number);
 * h_last is the controller's proposal for the step AFTER t1 (a last step clipped to t1 does not
 * shrink it), so consecutive output intervals chain without a restart transient. */
int rmt_n2_rk45(rmt_n2_handle* h, void* y_inout, double t0, double t1, double rtol, double atol,
                double h0, int64_t max_steps, rmt_n2_stats* stats_out);
int rmt_n2_ros4(rmt_n2_handle* h, void* y_inout, double t0, double t1, double rtol, double atol,
                double h0, int64_t max_steps, rmt_n2_stats* stats_out);
/* (tuning experiments: the environment variable RMT_N2_ROS4_CHUNKS=c overrides the number of chunks rmt_n2_ros4 cuts a
 * reactor into, 1 = one workgroup per reactor; unset = the library's estimate) */
/* members1: HOST [E][16+S+NU] rows (layout M1_* in csrc/kernels/22_node_n1.inc); out: DEVICE double [E][nout][S+2]
 * (S+1 when iso-thermal) = the state at z* = k/(nout-1); stats: DEVICE [E] */
int rmt_n1_profile(rmt_n2_handle* h, const double* members1, void* out, int nout, double rtol,
                   double atol, double h0, int64_t max_steps, rmt_n2_stats* stats_out);
/* copies the E flag words to host memory (synchronises the stream) and clears them on device */
int rmt_n2_status(rmt_n2_handle* h, uint32_t* flags_host);
/* which stepper rmt_n2_rk4 / rk45 / ros4 use: 0 = auto (on-chip if N fits one workgroup, else chained
 * workgroups - rk45: code objects that hold the on-chip stepper, at most RMT_N2_MAX_CHUNKS chunks of
 * block*nodes_per_thread nodes - else memory; ros4: one workgroup per reactor unless cutting the reactors
 * into chunks on several CUs, the teams working through the ensemble in rounds, is estimated to be faster),
 * 1 = on-chip single workgroup, 2 = one workgroup per reactor with the state in memory, 3 = chained workgroups */
int rmt_n2_set_mode(rmt_n2_handle* h, int mode);
/* how the last rk4 / rk45 / ros4 launch was laid out: workgroups (chunks) per reactor - 1 = one workgroup per reactor -
 * and the number of teams that worked through the ensemble (= E when every reactor had its own workgroup) */
int rmt_n2_last_geometry(rmt_n2_handle* h, int* chunks, int* teams);
/* Code objects whose on-chip RK4 steppers cache the temperature-only rate constants between the stages of a step
 * (kernels/50_rk4.inc) integrate a reactor whose temperature moved out of the cache's range during a launch again with
 * the plain stepper, inside the same rmt_n2_rk4 call (results are those of the plain stepper; the cost is that launch
 * twice).  How many reactor-launches that has happened to since rmt_n2_create (0 for every other code object);
 * synchronises the handle's stream. */
int rmt_n2_fallbacks(rmt_n2_handle* h, uint64_t* count);
/* timing of the last rk4/rk45/rhs launch in ms (HIP events on the handle's stream; synchronises) */
int rmt_n2_last_kernel_ms(rmt_n2_handle* h, float* ms);

const char* rmt_n2_last_error(void);
int rmt_n2_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RMT_N2_H */
