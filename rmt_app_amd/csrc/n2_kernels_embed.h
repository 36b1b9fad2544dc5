R"RMTSRC(// =====================================================================================
// rmt N2 device template (gfx950 / CDNA4, wave64).  Compiled at run time by hipRTC after a
// host-generated prelude that defines:
//
//   RMT_S, RMT_R            species / reactions            RMT_ISO   1 = iso-thermal (V = S)
//   RMT_FP32                1 = state & kinetics in fp32   RMT_BLOCK threads per workgroup
//   RMT_NPT                 nodes per thread of the register-resident steppers
//   typedef ... real;       double or float
//   RMT_MW[S]               molecular weights [g/mol]
//   rmt_species_source(r,s) s = nu^T r, rmt_reaction_dcp(cp,d) d = nu cp   (sparse, unrolled)
//   rmt_cp_mean(i,T)        0.5*(Cp_i(Tref)+Cp_i(T)) with only the non-zero polynomial terms
//   RMT_DH25[R]             standard heats of reaction [J/mol]
//   rmt_kinetics(T,P,x,C,r,flag)   lowered user rate lambdas
//
// What it computes: the method-of-lines right-hand side of PyREMOT's model N2
// (modelEquationN2, PyREMOT/docs/pbHomoReactor.py:3706-4134; equations restated in SURVEY.md
// Appendix A) for one axial mesh node per lane (RMT_NPT consecutive nodes per lane), and
// explicit time steppers on top of it (RK4 tableau of PyREMOT/solvers/odeSolver.py:17-40;
// Dormand-Prince 5(4) with per-reactor step control).
//
// Layout: state y[E][V][N] (the reference's row-major (V,N) flattening per reactor, reactors
// stacked), per-member constants members[E][RMT_NM] (see M_* below).  One workgroup integrates
// one reactor: the sequential Ergun pressure march  P[z+1] = a_z P[z] + b  (:3979) is an affine
// prefix scan (wave shuffles + one LDS exchange per stage), the upwind neighbour z-1 comes from
// lane-1 (shuffle) or the previous wave (same LDS exchange).  No inter-workgroup communication.
// =====================================================================================

#define RMT_V (RMT_S + (RMT_ISO ? 0 : 1))
#ifndef RMT_MEMBER_LDS
#define RMT_MEMBER_LDS 0
#endif
#ifndef RMT_STAGE_UNROLL
#define RMT_STAGE_UNROLL 1
#endif
#define RMT_NW (RMT_BLOCK / 64)
#define RMT_NM (16 + RMT_S)

#define RMT_FLAG_DOMAIN 1u
#define RMT_FLAG_DIV0 2u
#define RMT_FLAG_OVERFLOW 4u
#define RMT_FLAG_NONFINITE 8u
#define RMT_FLAG_STEP 16u

// member row layout (doubles); host side: rmt_app_amd/plan.py MEMBER_FIELDS
#define M_CMAX 0        // max(SpCoi0)                     [mol/m^3]
#define M_TF 1          // feed temperature                [K]
#define M_P0 2          // inlet pressure                  [Pa]
#define M_THETA_IN 3    // (T0-Tf)/Tf
#define M_ALPHA_K 4     // dz*1.75*vf^2*ergD/(dp*R):   a_z = 1 - ALPHA_K*M_z/T_z
#define M_BETA 5        // -dz*ergA*ergB
#define M_RHO_K 6       // 1/(R*GaDe0):                rho* = P*M/T*RHO_K
#define M_INV_CP0 7     // 1/GaCpMeanMix0
#define M_F1 8          // vf/(eps*zf)
#define M_FT 9          // vf/zf
#define M_INV_DZ 10     // N-1
#define M_INV_MACOTE 11 // 1/GaMaCoTe0
#define M_INV_HECOTE 12 // 1/GaHeCoTe0
#define M_UA 13         // U*a
#define M_TM 14         // medium temperature, 0 = adiabatic
#define M_CIN 16        // S inlet values SpCoi0[i]/Cmax

typedef double preal;   // the pressure scan is always carried in fp64

#define RMT_EPS real(1e-30)
#define RMT_TREF real(298.15)

// ------------------------------------------------------------------ math wrappers
__device__ __forceinline__ double rmt_expm1(double x) { return expm1(x); }
__device__ __forceinline__ double rmt_log10(double x) { return log10(x); }
__device__ __forceinline__ double rmt_log2(double x) { return log2(x); }
__device__ __forceinline__ double rmt_log1p(double x) { return log1p(x); }
__device__ __forceinline__ double rmt_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ double rmt_abs(double x) { return fabs(x); }
__device__ __forceinline__ double rmt_pow(double x, double y) { return pow(x, y); }
__device__ __forceinline__ double rmt_sin(double x) { return sin(x); }
__device__ __forceinline__ double rmt_cos(double x) { return cos(x); }
__device__ __forceinline__ double rmt_tan(double x) { return tan(x); }
__device__ __forceinline__ double rmt_tanh(double x) { return tanh(x); }
__device__ __forceinline__ double rmt_sinh(double x) { return sinh(x); }
__device__ __forceinline__ double rmt_cosh(double x) { return cosh(x); }
__device__ __forceinline__ double rmt_atan(double x) { return atan(x); }
__device__ __forceinline__ double rmt_min(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ double rmt_max(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ float rmt_expm1(float x) { return expm1f(x); }
__device__ __forceinline__ float rmt_log1p(float x) { return log1pf(x); }
__device__ __forceinline__ float rmt_abs(float x) { return fabsf(x); }
__device__ __forceinline__ float rmt_pow(float x, float y) { return powf(x, y); }
__device__ __forceinline__ float rmt_sin(float x) { return sinf(x); }
__device__ __forceinline__ float rmt_cos(float x) { return cosf(x); }
__device__ __forceinline__ float rmt_tan(float x) { return tanf(x); }
__device__ __forceinline__ float rmt_tanh(float x) { return tanhf(x); }
__device__ __forceinline__ float rmt_sinh(float x) { return sinhf(x); }
__device__ __forceinline__ float rmt_cosh(float x) { return coshf(x); }
__device__ __forceinline__ float rmt_atan(float x) { return atanf(x); }
__device__ __forceinline__ float rmt_min(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ float rmt_max(float a, float b) { return fmaxf(a, b); }

// ---- status flags ---------------------------------------------------------------------------
// The generated kinetics test the conditions on which Python would raise (log(<=0), x/0, exp
// overflow ...) through RMT_CHECK_POS / _NONNEG / _DEN / _EXP (and the generic RMT_CHECK).
// RMT_FLAGS_MODE 0 (device default): one v_cmp into an SGPR pair OR-ed into a wave-wide lane mask
//   with a scalar instruction; masks are collapsed to status bits at the end of the launch.
// RMT_FLAGS_MODE 1 (host emulation; selectable on the device): one fp64 min/max into per-lane
//   running extremes.  On gfx950 this cost 8 more live VGPRs and pushed the 512x2 kernel into
//   scratch (10.1 -> 6.9 G node-steps/s), so it is not the default there.
#define RMT_EXP_LIMIT 709.782712893384
#ifndef RMT_FLAGS_MODE
#ifdef RMT_HOST_EMULATION
#define RMT_FLAGS_MODE 1
#else
#define RMT_FLAGS_MODE 0
#endif
#endif
#if RMT_FLAGS_MODE == 1
struct rmt_flags_t {
    real pos_min;       // smallest argument that must be  > 0  (log)
    real nn_min;        // smallest argument that must be >= 0  (sqrt)
    real den_min;       // smallest |denominator|
    real exp_max;       // largest exp() argument
    unsigned bits;      // anything else (pow domain), per lane
};
__device__ __forceinline__ void rmt_flags_clear(rmt_flags_t& f) {
    f.pos_min = f.nn_min = f.den_min = real(__builtin_inf());
    f.exp_max = real(-__builtin_inf());
    f.bits = 0u;
}
__device__ __forceinline__ void rmt_flags_merge(rmt_flags_t& into, const rmt_flags_t& f) {
    into.pos_min = rmt_min(into.pos_min, f.pos_min);
    into.nn_min = rmt_min(into.nn_min, f.nn_min);
    into.den_min = rmt_min(into.den_min, f.den_min);
    into.exp_max = rmt_max(into.exp_max, f.exp_max);
    into.bits |= f.bits;
}
__device__ __forceinline__ unsigned rmt_flags_bits(const rmt_flags_t& f) {
    return f.bits | ((f.pos_min <= real(0) || f.nn_min < real(0)) ? RMT_FLAG_DOMAIN : 0u) |
           ((f.den_min == real(0)) ? RMT_FLAG_DIV0 : 0u) |
           ((f.exp_max > real(RMT_EXP_LIMIT)) ? RMT_FLAG_OVERFLOW : 0u);
}
#define RMT_CHECK_POS(f, x) (f).pos_min = rmt_min((f).pos_min, (x))
#define RMT_CHECK_NONNEG(f, x) (f).nn_min = rmt_min((f).nn_min, (x))
#define RMT_CHECK_DEN(f, x) (f).den_min = rmt_min((f).den_min, rmt_abs(x))
#define RMT_CHECK_EXP(f, x) (f).exp_max = rmt_max((f).exp_max, (x))
#define RMT_CHECK(f, cond, bit) (f).bits |= ((cond) ? (bit) : 0u)
#else
struct rmt_flags_t { unsigned long long dom, div0, ovf; };
template <unsigned BIT>
__device__ __forceinline__ void rmt_check(rmt_flags_t& f, const bool c) {
    const unsigned long long m = __ballot(c);
    if (BIT == RMT_FLAG_DOMAIN) f.dom |= m;
    else if (BIT == RMT_FLAG_DIV0) f.div0 |= m;
    else f.ovf |= m;
}
__device__ __forceinline__ void rmt_flags_clear(rmt_flags_t& f) { f.dom = f.div0 = f.ovf = 0ull; }
__device__ __forceinline__ void rmt_flags_merge(rmt_flags_t& into, const rmt_flags_t& f) {
    into.dom |= f.dom;
    into.div0 |= f.div0;
    into.ovf |= f.ovf;
}
__device__ __forceinline__ unsigned rmt_flags_bits(const rmt_flags_t& f) {
    return (f.dom ? RMT_FLAG_DOMAIN : 0u) | (f.div0 ? RMT_FLAG_DIV0 : 0u) | (f.ovf ? RMT_FLAG_OVERFLOW : 0u);
}
#define RMT_CHECK(f, cond, bit) rmt_check<bit>(f, (cond))
#define RMT_CHECK_POS(f, x) rmt_check<RMT_FLAG_DOMAIN>(f, (x) <= real(0))
#define RMT_CHECK_NONNEG(f, x) rmt_check<RMT_FLAG_DOMAIN>(f, (x) < real(0))
#define RMT_CHECK_DEN(f, x) rmt_check<RMT_FLAG_DIV0>(f, (x) == real(0))
#define RMT_CHECK_EXP(f, x) rmt_check<RMT_FLAG_OVERFLOW>(f, (x) > real(RMT_EXP_LIMIT))
#endif

// ---- lean fp64 division / reciprocal / log (RMT_FAST_MATH, default on) -----------------------
// gfx950 costs in fp64 VALU ops (llvm-objdump of the ocml versions): x/y 11 (+v_rcp_f64), log 76,
// log10 83, exp 19, sqrt 14.  The versions below drop the scale/fixup and double-double work that
// only matters for denormal/huge operands or the last half ulp: rcp 5, div 8, log ~30 ops, each
// accurate to ~1 ulp on normal numbers - far inside the 1e-12 parity budget of the RHS.
#ifndef RMT_FAST_MATH
#define RMT_FAST_MATH 1
#endif
#if RMT_FAST_MATH && !defined(RMT_HOST_EMULATION)
__device__ __forceinline__ double rmt_rcp(double b) {
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    r = fma(fma(-b, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double rmt_div(double a, double b) {
    const double r = rmt_rcp(b);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
}
__device__ __forceinline__ double rmt_log(double x) {     // after fdlibm e_log.c (x > 0, finite)
    int e = __builtin_amdgcn_frexp_exp(x);
    double m = __builtin_amdgcn_frexp_mant(x);            // [0.5, 1)
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;
    e = lo ? e - 1 : e;
    const double f = m - 1.0;
    const double s = rmt_div(f, 2.0 + f);
    const double dk = (double)e;
    const double z = s * s, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01),
                                     2.857142874366239149e-01), 6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    return dk * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
}
#else
__device__ __forceinline__ double rmt_rcp(double b) { return 1.0 / b; }
__device__ __forceinline__ double rmt_div(double a, double b) { return a / b; }
__device__ __forceinline__ double rmt_log(double x) { return log(x); }
#endif
#if RMT_FAST_MATH && !defined(RMT_HOST_EMULATION)
// exp(x) = 2^m * 2^(j/64) * e^r,  k = rint(x*64/ln2) = 64 m + j,  |r| <= ln2/128: a 64-entry table
// (staged in LDS by rmt_math_init) and a degree-5 polynomial - 12 fp64 ops against 19 + 6 range
// selects in the ocml version (the exp is ~40 % of the DME kernel's fp64 work).  Overflow is
// flagged by the generated checks before the call; results below 2^-1022 flush through ldexp.
__shared__ double rmt_exp_lds[64];
__device__ static const double RMT_EXP_TAB[64] = {
    0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0,
    0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0,
    0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,
    0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0,
    0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0,
    0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0,
    0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0,
    0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,
    0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0,
    0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0,
    0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,
    0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0,
    0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0,
    0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,
    0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0,
};
__device__ __forceinline__ void rmt_math_init() {
    if (threadIdx.x < 64) rmt_exp_lds[threadIdx.x] = RMT_EXP_TAB[threadIdx.x];
    __syncthreads();
}
__device__ __forceinline__ double rmt_exp(double x) {
    const double kd = __builtin_rint(x * 92.33248261689366);            // 64/ln2
    const int k = (int)kd;
    double r = fma(-kd, 0.010830424696249145, x);                       // ln2/64 hi
    r = fma(-kd, 3.623510646634843e-19, r);                             // ln2/64 lo
    double p = fma(r, 8.3333333333333332e-03, 4.1666666666666664e-02);  // 1/120, 1/24
    p = fma(p, r, 1.6666666666666666e-01);
    p = fma(p, r, 0.5);
    p = p * r;
    p = fma(p, r, r);                                                   // e^r - 1
    const double t = rmt_exp_lds[k & 63];
    return __builtin_ldexp(fma(t, p, t), k >> 6);
}
__device__ __forceinline__ double rmt_exp10(double x) { return rmt_exp(x * 2.302585092994046); }
__device__ __forceinline__ double rmt_exp2(double x) { return rmt_exp(x * 0.6931471805599453); }
#else
__device__ __forceinline__ void rmt_math_init() {}
__device__ __forceinline__ double rmt_exp(double x) { return exp(x); }
__device__ __forceinline__ double rmt_exp10(double x) { return exp10(x); }
__device__ __forceinline__ double rmt_exp2(double x) { return exp2(x); }
#endif
#if RMT_FAST_MATH && !defined(RMT_HOST_EMULATION)
// fp32: the hardware transcendental instructions (v_rcp_f32, v_exp_f32, v_log_f32, v_sqrt_f32; ~1 ulp)
__device__ __forceinline__ float rmt_rcp(float b) { return __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ float rmt_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ float rmt_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float rmt_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float rmt_exp10(float x) { return __builtin_amdgcn_exp2f(x * 3.32192809488736235f); }
__device__ __forceinline__ float rmt_log(float x) { return __builtin_amdgcn_logf(x) * 0.693147180559945309f; }
__device__ __forceinline__ float rmt_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float rmt_log10(float x) { return __builtin_amdgcn_logf(x) * 0.301029995663981195f; }
__device__ __forceinline__ float rmt_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
#else
__device__ __forceinline__ float rmt_rcp(float b) { return 1.0f / b; }
__device__ __forceinline__ float rmt_div(float a, float b) { return a / b; }
__device__ __forceinline__ float rmt_exp(float x) { return expf(x); }
__device__ __forceinline__ float rmt_exp2(float x) { return exp2f(x); }
__device__ __forceinline__ float rmt_exp10(float x) { return exp10f(x); }
__device__ __forceinline__ float rmt_log(float x) { return logf(x); }
__device__ __forceinline__ float rmt_log2(float x) { return log2f(x); }
__device__ __forceinline__ float rmt_log10(float x) { return log10f(x); }
__device__ __forceinline__ float rmt_sqrt(float x) { return sqrtf(x); }
#endif

RMT_KINETICS_SOURCE

// ------------------------------------------------------------------ per-member constants
struct RmtMember {
    real cmax, tf, theta_in, rho_k, inv_cp0, f1, ft, inv_dz, inv_macote, inv_hecote, ua, tm;
    preal p0, alpha_k, beta;
    real cin[RMT_S];
};

// A member field whose value is identical for every reactor of the launch can be baked into the
// kernel as a literal (prelude: #define RMT_MC_<FIELD> value): literals are rematerialised with
// s_mov instead of occupying (and spilling) SGPRs for the whole time loop.
__device__ __forceinline__ void rmt_load_member(const double* __restrict__ row, RmtMember& m) {
#ifdef RMT_MC_CMAX
    m.cmax = real(RMT_MC_CMAX);
#else
    m.cmax = real(row[M_CMAX]);
#endif
#ifdef RMT_MC_TF
    m.tf = real(RMT_MC_TF);
#else
    m.tf = real(row[M_TF]);
#endif
#ifdef RMT_MC_THETA_IN
    m.theta_in = real(RMT_MC_THETA_IN);
#else
    m.theta_in = real(row[M_THETA_IN]);
#endif
#ifdef RMT_MC_RHO_K
    m.rho_k = real(RMT_MC_RHO_K);
#else
    m.rho_k = real(row[M_RHO_K]);
#endif
#ifdef RMT_MC_INV_CP0
    m.inv_cp0 = real(RMT_MC_INV_CP0);
#else
    m.inv_cp0 = real(row[M_INV_CP0]);
#endif
#ifdef RMT_MC_F1
    m.f1 = real(RMT_MC_F1);
#else
    m.f1 = real(row[M_F1]);
#endif
#ifdef RMT_MC_FT
    m.ft = real(RMT_MC_FT);
#else
    m.ft = real(row[M_FT]);
#endif
#ifdef RMT_MC_INV_DZ
    m.inv_dz = real(RMT_MC_INV_DZ);
#else
    m.inv_dz = real(row[M_INV_DZ]);
#endif
#ifdef RMT_MC_INV_MACOTE
    m.inv_macote = real(RMT_MC_INV_MACOTE);
#else
    m.inv_macote = real(row[M_INV_MACOTE]);
#endif
#ifdef RMT_MC_INV_HECOTE
    m.inv_hecote = real(RMT_MC_INV_HECOTE);
#else
    m.inv_hecote = real(row[M_INV_HECOTE]);
#endif
#ifdef RMT_MC_UA
    m.ua = real(RMT_MC_UA);
#else
    m.ua = real(row[M_UA]);
#endif
#ifdef RMT_MC_TM
    m.tm = real(RMT_MC_TM);
#else
    m.tm = real(row[M_TM]);
#endif
#ifdef RMT_MC_P0
    m.p0 = RMT_MC_P0;
#else
    m.p0 = row[M_P0];
#endif
#ifdef RMT_MC_ALPHA_K
    m.alpha_k = RMT_MC_ALPHA_K;
#else
    m.alpha_k = row[M_ALPHA_K];
#endif
#ifdef RMT_MC_BETA
    m.beta = RMT_MC_BETA;
#else
    m.beta = row[M_BETA];
#endif
#ifdef RMT_MC_CIN
    {
        const double cin_[RMT_S] = RMT_MC_CIN;
#pragma unroll
        for (int i = 0; i < RMT_S; ++i) m.cin[i] = real(cin_[i]);
    }
#else
#pragma unroll
    for (int i = 0; i < RMT_S; ++i) m.cin[i] = real(row[M_CIN + i]);
#endif
}

// ------------------------------------------------------------------ node physics
// Phase A (before the pressure scan): clamp, real concentrations, mole fractions, T, mixture MW,
// Ergun affine coefficient.  pbHomoReactor.py:3897-3928, 3960-3979.
struct RmtNode {
    real x[RMT_S];
    real C[RMT_S];
    real T, M;
    real MoT;      // M/T, shared by the Ergun coefficient and the EOS density
};

__device__ __forceinline__ preal rmt_node_pre(const RmtMember& m, const real* __restrict__ ys,
                                              RmtNode& nd) {
    real ctot = real(0);
#pragma unroll
    for (int i = 0; i < RMT_S; ++i) {
        const real cc = rmt_max(ys[i], RMT_EPS);          // :3899
        nd.C[i] = cc * m.cmax;                            // :3903 (MAX scaling)
        ctot += nd.C[i];
    }
    const real inv_ctot = rmt_rcp(ctot);
    real mw = real(0);
#pragma unroll
    for (int i = 0; i < RMT_S; ++i) {
        nd.x[i] = nd.C[i] * inv_ctot;                     // :3927
        mw += nd.x[i] * RMT_MW[i];
    }
    nd.M = mw * real(1e-3);                               // :3960  [kg/mol]
#if RMT_ISO
    nd.T = m.tf;
#else
    nd.T = ys[RMT_S] * m.tf + m.tf;                       // :3914
#endif
    // P[z+1] = P[z] + dz*(-(ergA*ergB + 1.75*rho*v^2/dp*ergD)),  rho = P*M/(R*T)   (:3964-3979)
    const preal mot = rmt_div(preal(nd.M), preal(nd.T));
    nd.MoT = real(mot);
    return preal(1) - m.alpha_k * mot;
}

// Phase B (pressure known): kinetics, species source, Cp, heat of reaction, wall exchange,
// upwind balances.  pbHomoReactor.py:3989-4128.  `up` = clamped state of node z-1 (or inlet).
__device__ __forceinline__ void rmt_node_post(const RmtMember& m, const RmtNode& nd,
                                              const real* __restrict__ ys,
                                              const real* __restrict__ up, const preal Pz,
                                              real* __restrict__ k, rmt_flags_t& flag) {
    const real P = real(Pz);
    // (1) convective parts first: after this the stage state and the upstream values are dead,
    //     which keeps the register footprint of the kinetics small.
#pragma unroll
    for (int i = 0; i < RMT_S; ++i)
        k[i] = -m.f1 * ((ys[i] - up[i]) * m.inv_dz);                    // :4086-4098, convective term
#if !RMT_ISO
    const real T = nd.T;
    k[RMT_S] = -m.ft * ((ys[RMT_S] - up[RMT_S]) * m.inv_dz);            // :4104-4116
    // (2) everything of the energy balance that does not need the rates
    real hq[RMT_R];
    real cpm = real(0);
    {
        real cpbar[RMT_S];
#pragma unroll
        for (int i = 0; i < RMT_S; ++i) {
            cpbar[i] = rmt_cp_mean(i, T);                               // rmtThermo.py:52-75
            cpm += nd.x[i] * cpbar[i];                                  // :4013
        }
        real dcp[RMT_R];
        rmt_reaction_dcp(cpbar, dcp);                                   // sparse nu cpbar
#pragma unroll
        for (int q = 0; q < RMT_R; ++q) hq[q] = dcp[q] * (T - RMT_TREF) + RMT_DH25[q];   // :4025-4028
    }
    const real qm = (m.tm == real(0)) ? real(0) : m.ua * (m.tm - T);    // rmtUtility.py:438-445
    const real rho_s = (P * nd.MoT) * m.rho_k;                          // :3964-3966
    const real cp_s = cpm * m.inv_cp0;                                  // :4016
    const real gain = rmt_div(m.f1 * m.inv_hecote, rho_s * cp_s);       // const_T2/GaHeCoTe0, :4077,4118
#endif
    // (3) kinetics and the source terms
    real r[RMT_R];
    rmt_kinetics(nd.T, P, nd.x, nd.C, r, flag);                         // :3989-3992
    real src[RMT_S];
    rmt_species_source(r, src);                                         // :4000 (sparse nu^T r)
    const real fm = m.f1 * m.inv_macote;
#pragma unroll
    for (int i = 0; i < RMT_S; ++i) k[i] += fm * src[i];                // :4098
#if !RMT_ISO
    real qr = real(0);
#pragma unroll
    for (int q = 0; q < RMT_R; ++q) qr += r[q] * hq[q];                 // :4032
    k[RMT_S] += gain * (qm - qr);                                       // :4118-4126
#endif
}

// ------------------------------------------------------------------ steady-state model N1: node function
// SURVEY.md section 8(f) rank 1: PackedBedHomoReactorClass.runN1 / modelEquationN1
// (pbHomoReactor.py:2694-3314), the steady sibling of N2 - an ODE in the dimensionless length z*
// for u = [c_1..c_S, P*, theta] - integrated by the reference with solve_ivp(LSODA) at
// t_eval = linspace(0, 1, zNo+1) (:2931).  Along z the chemistry is as stiff as in time, so the
// same Rosenbrock(4,3) scheme is used, here with a dense (S+2)^2 finite-difference Jacobian and ONE
// REACTOR PER LANE (an ensemble of N1 profiles); steps are clipped to the output grid.
//   member row (doubles, host: plan.member_constants_n1): see M1_* below.
#define RMT_V1 (RMT_S + 1 + (RMT_ISO ? 0 : 1))
#define RMT_NM1 (16 + RMT_S)
#define M1_CMAX 0
#define M1_TF 1
#define M1_PF 2          // = P0
#define M1_SPCO0 3
#define M1_ERGA 4        // 150 mu ergB/dp^2 * zf/Pf      (times SuGaVe)
#define M1_ERGC 5        // 1.75 ergD/dp * zf/Pf          (times rho SuGaVe^2)
#define M1_SUGAVE0 6
#define M1_RHO_K 7       // 1/(R GaDe0)
#define M1_INV_CP0 8
#define M1_EPS 9
#define M1_INV_MACOTE 10
#define M1_INV_HECOTE 11
#define M1_UA 12
#define M1_TM 13
#define M1_GADE0 14
#define M1_CIN 16

__device__ __forceinline__ void rmt_n1_rhs(const double* __restrict__ mr, const real (&u)[RMT_V1],
                                           real (&du)[RMT_V1], rmt_flags_t& flag) {
    const real cmax = real(mr[M1_CMAX]), tf = real(mr[M1_TF]), pf = real(mr[M1_PF]);
    real C[RMT_S], x[RMT_S];
    real ctot = real(0);
#pragma unroll
    for (int i = 0; i < RMT_S; ++i) { C[i] = u[i] * cmax; ctot += C[i]; }      // :3123 (no clamp in N1)
    const real inv_ctot = rmt_rcp(ctot);
    real mw = real(0);
#pragma unroll
    for (int i = 0; i < RMT_S; ++i) { x[i] = C[i] * inv_ctot; mw += x[i] * RMT_MW[i]; }
    const real M = mw * real(1e-3);
    const real P = u[RMT_S] * pf;                                              // :3135
#if RMT_ISO
    const real T = tf;
#else
    const real T = u[RMT_S + 1] * tf + tf;
#endif
    // interstitial velocity from the EOS (rmtUtility.py:404-421): InGaVe* = (Ctot/SpCo0)(P0/P)
    const real vstar = rmt_div(ctot * pf, real(mr[M1_SPCO0]) * P);
    const real su = vstar * real(mr[M1_SUGAVE0]);
    const real rho = rmt_div(P * M, real(8.314472) * T);                        // rmtThermo.py:353
    const real rho_s = rho * rmt_rcp(real(mr[M1_GADE0]));
    du[RMT_S] = -(real(mr[M1_ERGA]) * su + real(mr[M1_ERGC]) * rho * su * su);  // :3206-3220
    real r[RMT_R];
    rmt_kinetics(T, P, x, C, r, flag);
    real src[RMT_S];
    rmt_species_source(r, src);
    const real iv = rmt_rcp(vstar);
#pragma unroll
    for (int i = 0; i < RMT_S; ++i) du[i] = iv * src[i] * real(mr[M1_INV_MACOTE]);   // :3283-3289
#if !RMT_ISO
    real cpbar[RMT_S], cpm = real(0);
#pragma unroll
    for (int i = 0; i < RMT_S; ++i) { cpbar[i] = rmt_cp_mean(i, T); cpm += x[i] * cpbar[i]; }
    real dcp[RMT_R];
    rmt_reaction_dcp(cpbar, dcp);
    real qr = real(0);
#pragma unroll
    for (int q = 0; q < RMT_R; ++q) qr += r[q] * (dcp[q] * (T - RMT_TREF) + RMT_DH25[q]);
    const real tm = real(mr[M1_TM]);
    const real qm = (tm == real(0)) ? real(0) : real(mr[M1_UA]) * (tm - T);
    const real cp_eff = cpm * real(mr[M1_INV_CP0]) * real(mr[M1_EPS]);
    du[RMT_S + 1] = rmt_div((qm - qr) * real(mr[M1_INV_HECOTE]), rho_s * cp_eff * vstar);   // :3284,3298
#endif
}

#ifndef RMT_HOST_EMULATION
// ---- cross-lane moves on the DPP path (no LDS crossbar round trip) ----------------------------
// RMT_DPP 1: v_mov_b32 ... row_shr / row_bcast / wave_shr (gfx9 DPP controls) for the 64-lane
// scan and the lane-1 neighbour fetch; RMT_DPP 0: __shfl_up (ds_bpermute_b32).
#ifndef RMT_DPP
#define RMT_DPP 1
#endif
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double rmt_dpp(const double old, const double v) {
    int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float rmt_dpp(const float old, const float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
// value of lane-1 (lane 0 keeps its own value, like __shfl_up(v, 1))
template <typename T>
__device__ __forceinline__ T rmt_lane_up1(const T v) {
#if RMT_DPP
    return rmt_dpp<0x138, 0xf>(v, v);                 // wave_shr:1
#else
    return __shfl_up(v, 1);
#endif
}

// ------------------------------------------------------------------ affine maps P -> a*P + b
struct RmtAff { preal a, b; };
__device__ __forceinline__ RmtAff rmt_then(const RmtAff first, const RmtAff second) {
    RmtAff o;
    o.a = second.a * first.a;
    o.b = second.a * first.b + second.b;
    return o;
}

struct RmtShared {
    preal tot_a[2][RMT_NW];
    preal tot_b[2][RMT_NW];
    real bnd[2][RMT_NW][RMT_V];
    real red[2][RMT_NW];
    double cin[2][RMT_V + 1];     // chained workgroups: upstream record (up[V], P) of this stage
    int abort[2];
};

// ---- chained workgroups (one reactor spread over C workgroups) --------------------------------
// Information only flows downstream (upwind stencil, pressure march from the inlet), so the
// workgroups of a reactor form a producer->consumer chain with NO grid barrier: per RHS stage,
// chunk c hands chunk c+1 one record {clamped state of its last node, pressure after its last
// node} through a ring of RMT_CHAIN_DEPTH slots in global memory.  Protocol = the fence-free
// form of the CDNA guide (MI355X_MICROARCH "Valid forms", table row 1): every payload byte is
// stored sc1 (agent-scope relaxed atomic store = write-through) by ONE wave, that wave drains
// vmcnt, then ONE lane stores the sequence flag sc1; the consumer's polling wave reads the flag
// and then the payload with sc1 loads (agent-scope relaxed atomic loads, L1-bypassing).  A
// consumed-counter gives back-pressure.  (The release/acquire-fence form measured ~3 us more per
// stage.)  Every spin is bounded; a timeout poisons both counters so that the whole
// chain drains and the launch ends with RMT_FLAG_STEP instead of hanging.
#ifndef RMT_CHAIN_DEPTH
#define RMT_CHAIN_DEPTH 8
#endif
#define RMT_CHAIN_POISON (~0ull)
#define RMT_CHAIN_SPINS (1u << 22)
#define RMT_SYNC_STRIDE 32        // u64 words per link: [0] published seq, [16] consumed seq (own 128-B lines)

struct RmtChainCtx {
    bool has_in, has_out;
    unsigned long long* sync_in;      // link c-1 -> c
    unsigned long long* sync_out;     // link c -> c+1
    const double* slots_in;
    double* slots_out;
    unsigned long long q;             // sequence number of the current stage (1, 2, ...)
    int abort;                        // workgroup-uniform, valid after the stage's barrier
};

__device__ __forceinline__ int rmt_chain_wait(unsigned long long* word, const unsigned long long need) {
    unsigned spins = 0;
    for (;;) {
        const unsigned long long v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v == RMT_CHAIN_POISON) return 1;
        if (v >= need) return 0;
        if (++spins > RMT_CHAIN_SPINS) return 1;
        __builtin_amdgcn_s_sleep(2);
    }
}

struct RmtLocal {       // what a node's balances see of the rest of the reactor
    preal P;            // pressure at the node
    real up[RMT_V];     // clamped state of the upstream node (inlet values for node 0)
};

struct RmtCarry {       // workgroup-uniform hand-over between consecutive node blocks
    preal P;            // pressure at the first node of the block
    real up[RMT_V];     // clamped state of the node just upstream of the block (inlet for block 0)
};

__device__ __forceinline__ void rmt_carry_inlet(const RmtMember& m, RmtCarry& c) {
    c.P = m.p0;
#pragma unroll
    for (int i = 0; i < RMT_S; ++i) c.up[i] = m.cin[i];                 // :4090
#if !RMT_ISO
    c.up[RMT_S] = m.theta_in;                                           // :4108
#endif
}

// RHS for the NPT consecutive nodes owned by this thread (nodes base+tid*NPT ...).
// `nvalid` = how many of them exist (< NPT only at the reactor's end).  One __syncthreads().
// buf = LDS ping-pong index; callers alternate it between consecutive calls.
template <int NPT, bool CARRY_OUT, bool CHAIN = false>
__device__ __forceinline__ void rmt_rhs_block(const RmtMember& m, RmtShared& sh, const int buf,
                                              const real (&ys)[NPT][RMT_V], const int nvalid,
                                              RmtCarry& carry, real (&k)[NPT][RMT_V],
                                              rmt_flags_t& flag, RmtChainCtx* ctx = nullptr,
                                              RmtLocal* loc_out = nullptr) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    RmtNode nd[NPT];
    RmtAff loc[NPT];
    RmtAff mine = {preal(1), preal(0)};
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const preal a = rmt_node_pre(m, ys[j], nd[j]);
        loc[j].a = (j < nvalid) ? a : preal(1);
        loc[j].b = (j < nvalid) ? m.beta : preal(0);
        mine = rmt_then(mine, loc[j]);
    }
    // inclusive scan of the per-lane maps over the wave
    RmtAff inc = mine;
#if RMT_DPP
    {   // lanes whose source is out of range / masked receive the identity map (a = 1, b = 0)
#define RMT_SCAN_STEP(CTRL, MASK)                                               \
        {                                                                       \
            RmtAff prev;                                                        \
            prev.a = rmt_dpp<CTRL, MASK>(preal(1), inc.a);                      \
            prev.b = rmt_dpp<CTRL, MASK>(preal(0), inc.b);                      \
            inc = rmt_then(prev, inc);                                          \
        }
        RMT_SCAN_STEP(0x111, 0xf)      // row_shr:1
        RMT_SCAN_STEP(0x112, 0xf)      // row_shr:2
        RMT_SCAN_STEP(0x114, 0xf)      // row_shr:4
        RMT_SCAN_STEP(0x118, 0xf)      // row_shr:8
        RMT_SCAN_STEP(0x142, 0xa)      // row_bcast:15 -> rows 1, 3
        RMT_SCAN_STEP(0x143, 0xc)      // row_bcast:31 -> rows 2, 3
#undef RMT_SCAN_STEP
    }
#else
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        RmtAff prev;
        prev.a = __shfl_up(inc.a, d);
        prev.b = __shfl_up(inc.b, d);
        if (lane >= d) inc = rmt_then(prev, inc);
    }
#endif
    // clamped state of my last node, for my downstream neighbour
    real last[RMT_V];
#pragma unroll
    for (int i = 0; i < RMT_S; ++i) last[i] = rmt_max(ys[NPT - 1][i], RMT_EPS);      // :4093
#if !RMT_ISO
    last[RMT_S] = ys[NPT - 1][RMT_S];                                                // :4111
#endif
    if (lane == 63) {
        sh.tot_a[buf][wave] = inc.a;
        sh.tot_b[buf][wave] = inc.b;
#pragma unroll
        for (int i = 0; i < RMT_V; ++i) sh.bnd[buf][wave][i] = last[i];
    }
    RmtAff exc;
    exc.a = rmt_lane_up1(inc.a);
    exc.b = rmt_lane_up1(inc.b);
    if (lane == 0) { exc.a = preal(1); exc.b = preal(0); }
    real up[RMT_V];
#pragma unroll
    for (int i = 0; i < RMT_V; ++i) up[i] = rmt_lane_up1(last[i]);
    if (CHAIN) {
        if (wave == 0) {                          // all waiting is done by one wave, before the barrier
            int st = 0;
            double rec = 0.0;
            if (ctx->has_in) {
                if (lane == 0) st = rmt_chain_wait(ctx->sync_in, ctx->q);
                st = __shfl(st, 0);
                if (lane <= RMT_V) {              // sc1 load (bypasses this CU's L1): no acquire fence needed
                    rec = __longlong_as_double((long long)__hip_atomic_load(
                        (const unsigned long long*)ctx->slots_in +
                            (size_t)(ctx->q % RMT_CHAIN_DEPTH) * (RMT_V + 1) + lane,
                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    sh.cin[buf][lane] = rec;
                }
            }
            if (ctx->has_out && lane == 0 && ctx->q > RMT_CHAIN_DEPTH)      // slot free again?
                st |= rmt_chain_wait(ctx->sync_out + 16, ctx->q - RMT_CHAIN_DEPTH);
            if (lane == 0) sh.abort[buf] = st;
        }
    }
    __syncthreads();
    if (CHAIN) {
        if (ctx->has_in) {                        // chunk 0 keeps the inlet carry it was given
            carry.P = sh.cin[buf][RMT_V];
#pragma unroll
            for (int i = 0; i < RMT_V; ++i) carry.up[i] = real(sh.cin[buf][i]);
        }
        ctx->abort = sh.abort[buf];
        if (ctx->has_in && threadIdx.x == 0)      // record consumed (it sits in LDS now)
            __hip_atomic_store(ctx->sync_in + 16, ctx->q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // pressure entering this wave = carry.P pushed through the totals of the waves before it
    preal pw = carry.P;
    for (int w = 0; w < wave; ++w) pw = sh.tot_a[buf][w] * pw + sh.tot_b[buf][w];
    if (lane == 0) {
        if (wave == 0) {
#pragma unroll
            for (int i = 0; i < RMT_V; ++i) up[i] = carry.up[i];
        } else {
#pragma unroll
            for (int i = 0; i < RMT_V; ++i) up[i] = sh.bnd[buf][wave - 1][i];
        }
    }
    preal P = exc.a * pw + exc.b;
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        // lanes beyond the reactor's end carry the (valid) inlet state, so their checks are not masked
        rmt_node_post(m, nd[j], ys[j], up, P, k[j], flag);
        if (loc_out) {                                   // frozen neighbourhood of node j (Jacobian)
            loc_out[j].P = P;
#pragma unroll
            for (int i = 0; i < RMT_V; ++i) loc_out[j].up[i] = up[i];
        }
        P = loc[j].a * P + loc[j].b;
        if (j + 1 < NPT) {
#pragma unroll
            for (int i = 0; i < RMT_S; ++i) up[i] = rmt_max(ys[j][i], RMT_EPS);
#if !RMT_ISO
            up[RMT_S] = ys[j][RMT_S];
#endif
        }
    }
    if (CARRY_OUT) {
        preal pe = pw;
        for (int w = wave; w < RMT_NW; ++w) pe = sh.tot_a[buf][w] * pe + sh.tot_b[buf][w];
        carry.P = pe;
#pragma unroll
        for (int i = 0; i < RMT_V; ++i) carry.up[i] = sh.bnd[buf][RMT_NW - 1][i];
    }
    if (CHAIN) {
        if (ctx->has_out && wave == RMT_NW - 1 && !ctx->abort) {     // publish this stage's record
            const preal pe = sh.tot_a[buf][wave] * pw + sh.tot_b[buf][wave];
            if (lane <= RMT_V) {                  // sc1 (write-through) stores, drained before the flag
                const double v = (lane == RMT_V) ? (double)pe : (double)sh.bnd[buf][RMT_NW - 1][lane < RMT_V ? lane : 0];
                __hip_atomic_store((unsigned long long*)ctx->slots_out +
                                       (size_t)(ctx->q % RMT_CHAIN_DEPTH) * (RMT_V + 1) + lane,
                                   (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0)
                __hip_atomic_store(ctx->sync_out, ctx->q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ------------------------------------------------------------------ helpers for the kernels
__device__ __forceinline__ void rmt_safe_state(const RmtMember& m, real* __restrict__ v) {
#pragma unroll
    for (int i = 0; i < RMT_S; ++i) v[i] = m.cin[i];
#if !RMT_ISO
    v[RMT_S] = m.theta_in;
#endif
}

__device__ __forceinline__ unsigned rmt_finite_flag(const real* __restrict__ v) {
    unsigned f = 0u;
#pragma unroll
    for (int i = 0; i < RMT_V; ++i) f |= __builtin_isfinite(v[i]) ? 0u : RMT_FLAG_NONFINITE;
    return f;
}

// ===================================================================== kernel: one RHS evaluation
// dydt[e] = f(y[e]) for every reactor e = blockIdx.x; any N (blocks of RMT_BLOCK nodes, carry
// handed from block to block).  Replaces one call of modelEquationN2 (pbHomoReactor.py:3706).
extern "C" __global__ __launch_bounds__(RMT_BLOCK) void rmt_n2_rhs(
        const real* __restrict__ y, real* __restrict__ dydt, const double* __restrict__ members,
        const int N, unsigned* __restrict__ flags) {
    __shared__ RmtShared sh;
    rmt_math_init();
    const int e = blockIdx.x;
#if RMT_MEMBER_LDS
    __shared__ RmtMember m;      // read back with broadcast ds_reads: frees ~2*(15+S) SGPRs
    if (threadIdx.x == 0) rmt_load_member(members + (size_t)e * RMT_NM, m);
    __syncthreads();
#else
    RmtMember m;
    rmt_load_member(members + (size_t)e * RMT_NM, m);
#endif
    RmtCarry carry;
    rmt_carry_inlet(m, carry);
    const real* ye = y + (size_t)e * RMT_V * N;
    real* de = dydt + (size_t)e * RMT_V * N;
    rmt_flags_t flag;
    rmt_flags_clear(flag);
    unsigned lflag = 0u;
    int ph = 0;
    for (int base = 0; base < N; base += RMT_BLOCK, ph ^= 1) {
        const int node = base + (int)threadIdx.x;
        const bool valid = node < N;
        real ys[1][RMT_V], k[1][RMT_V];
        rmt_safe_state(m, ys[0]);
        if (valid) {
#pragma unroll
            for (int i = 0; i < RMT_V; ++i) ys[0][i] = ye[(size_t)i * N + node];
        }
        rmt_rhs_block<1, true>(m, sh, ph, ys, valid ? 1 : 0, carry, k, flag);
        if (valid) {
#pragma unroll
            for (int i = 0; i < RMT_V; ++i) de[(size_t)i * N + node] = k[0][i];
            lflag |= rmt_finite_flag(k[0]);
        }
    }
    lflag |= rmt_flags_bits(flag);
    if (lflag) atomicOr(&flags[e], lflag);
}

// ===================================================================== kernel: RK4, state on chip
// nsteps classic RK4 steps of size h (tableau and update order of odeSolver.py:17-40) for reactor
// e = blockIdx.x with N <= RMT_BLOCK*RMT_NPT nodes.  The state is read from HBM once, kept on
// chip for all steps and written once: HBM traffic is 2*V*sizeof(real) per node per LAUNCH.
// RMT_LDS_STATE selects where the two long-lived RK4 vectors live (only the stage input and the
// current K are always in VGPRs):  2 = y_n and the K accumulator in LDS,  1 = y_n in LDS,
// 0 = both in VGPRs.  Each thread only touches its own nodes' LDS slots, so no barrier is needed.
#ifndef RMT_LDS_STATE
#define RMT_LDS_STATE 0
#endif
#define RMT_NODES_WG (RMT_BLOCK * RMT_NPT)

extern "C" __global__ __launch_bounds__(RMT_BLOCK) void rmt_n2_rk4_reg(
        real* __restrict__ y, const double* __restrict__ members, const int N, const double h_,
        const long long nsteps, unsigned* __restrict__ flags) {
    __shared__ RmtShared sh;
    rmt_math_init();
#if RMT_LDS_STATE >= 1
    __shared__ real s_y0[RMT_V][RMT_NODES_WG];
#endif
#if RMT_LDS_STATE >= 2
    __shared__ real s_acc[RMT_V][RMT_NODES_WG];
#endif
    const int e = blockIdx.x;
#if RMT_MEMBER_LDS
    __shared__ RmtMember m;      // read back with broadcast ds_reads: frees ~2*(15+S) SGPRs
    if (threadIdx.x == 0) rmt_load_member(members + (size_t)e * RMT_NM, m);
    __syncthreads();
#else
    RmtMember m;
    rmt_load_member(members + (size_t)e * RMT_NM, m);
#endif
    RmtCarry carry;
    rmt_carry_inlet(m, carry);
    real* ye = y + (size_t)e * RMT_V * N;
    const int node0 = (int)threadIdx.x * RMT_NPT;
    int nvalid = N - node0;
    nvalid = nvalid < 0 ? 0 : (nvalid > RMT_NPT ? RMT_NPT : nvalid);
    real ys[RMT_NPT][RMT_V], k[RMT_NPT][RMT_V];
#if RMT_LDS_STATE < 1
    real y0[RMT_NPT][RMT_V];
#define Y0(j, i) y0[j][i]
#else
#define Y0(j, i) s_y0[i][node0 + (j)]
#endif
#if RMT_LDS_STATE < 2
    real acc[RMT_NPT][RMT_V];
#define ACC(j, i) acc[j][i]
#else
#define ACC(j, i) s_acc[i][node0 + (j)]
#endif
#pragma unroll
    for (int j = 0; j < RMT_NPT; ++j) {
        rmt_safe_state(m, ys[j]);
        if (j < nvalid) {
#pragma unroll
            for (int i = 0; i < RMT_V; ++i) ys[j][i] = ye[(size_t)i * N + node0 + j];
        }
#pragma unroll
        for (int i = 0; i < RMT_V; ++i) Y0(j, i) = ys[j][i];
    }
    const real h = real(h_), hh = real(0.5 * h_), h6 = real(h_ / 6.0);
    rmt_flags_t flag;
    rmt_flags_clear(flag);
    unsigned lflag = 0u;
    for (long long step = 0; step < nsteps; ++step) {
        // One copy of the RHS code serves the four stages (a fully unrolled step is ~64 KB of
        // instructions - the size of the instruction cache); s is wave-uniform.
#if RMT_STAGE_UNROLL
#pragma unroll
#else
#pragma unroll 1
#endif
        for (int s = 0; s < 4; ++s) {
            rmt_rhs_block<RMT_NPT, false>(m, sh, s & 1, ys, nvalid, carry, k, flag);  // K_{s+1} = f(ys)
            if (s < 3) {
                const real cn = (s == 2) ? h : hh;                // stage inputs y+K1 h/2, y+K2 h/2, y+K3 h
                const real wk = (s == 0) ? real(1) : real(2);     // weights 1,2,2,(1)
#pragma unroll
                for (int j = 0; j < RMT_NPT; ++j)
#pragma unroll
                    for (int i = 0; i < RMT_V; ++i) {
                        ACC(j, i) = (s == 0) ? k[j][i] : ACC(j, i) + wk * k[j][i];
                        ys[j][i] = Y0(j, i) + k[j][i] * cn;
                    }
            } else {
#pragma unroll
                for (int j = 0; j < RMT_NPT; ++j)
#pragma unroll
                    for (int i = 0; i < RMT_V; ++i) {             // y + h(K1+2K2+2K3+K4)/6
                        const real yn = Y0(j, i) + h6 * (ACC(j, i) + k[j][i]);
                        Y0(j, i) = yn;
                        ys[j][i] = yn;
                    }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < RMT_NPT; ++j) {
        if (j < nvalid) {
#pragma unroll
            for (int i = 0; i < RMT_V; ++i) ye[(size_t)i * N + node0 + j] = ys[j][i];
            lflag |= rmt_finite_flag(ys[j]);
        }
    }
    lflag |= rmt_flags_bits(flag);
    if (lflag) atomicOr(&flags[e], lflag);
#undef Y0
#undef ACC
}

// ===================================================================== kernel: RK4 on chip, chained workgroups
// Same on-chip stepper for N > RMT_BLOCK*RMT_NPT (or to spread one reactor over several CUs):
// reactor e is cut into C chunks of RMT_NODES_WG nodes; the grid is T teams of C workgroups
// (T*C <= number of CUs, so every workgroup is resident), team t integrates reactors
// t, t+T, t+2T, ...  The hand-over between consecutive chunks is described at RmtChainCtx.
extern "C" __global__ __launch_bounds__(RMT_BLOCK) void rmt_n2_rk4_chain(
        real* __restrict__ y, const double* __restrict__ members, const int N, const int E,
        const int C, const int T, const double h_, const long long nsteps,
        unsigned long long* __restrict__ sync, double* __restrict__ slots,
        unsigned* __restrict__ flags) {
    __shared__ RmtShared sh;
    rmt_math_init();
#if RMT_LDS_STATE >= 1
    __shared__ real s_y0[RMT_V][RMT_NODES_WG];
#endif
#if RMT_LDS_STATE >= 2
    __shared__ real s_acc[RMT_V][RMT_NODES_WG];
#endif
    const int team = blockIdx.x / C, c = blockIdx.x % C;
    RmtChainCtx ctx;
    ctx.has_in = c > 0;
    ctx.has_out = c < C - 1;
    const size_t link = (size_t)team * C + c;
    ctx.sync_in = sync + (link - (c > 0 ? 1 : 0)) * RMT_SYNC_STRIDE;
    ctx.sync_out = sync + link * RMT_SYNC_STRIDE;
    ctx.slots_in = slots + (link - (c > 0 ? 1 : 0)) * RMT_CHAIN_DEPTH * (RMT_V + 1);
    ctx.slots_out = slots + link * RMT_CHAIN_DEPTH * (RMT_V + 1);
    ctx.q = 0ull;
    ctx.abort = 0;
    const int node0 = c * RMT_NODES_WG + (int)threadIdx.x * RMT_NPT;
    int nvalid = N - node0;
    nvalid = nvalid < 0 ? 0 : (nvalid > RMT_NPT ? RMT_NPT : nvalid);
    const int lnode0 = (int)threadIdx.x * RMT_NPT;
    const real h = real(h_), hh = real(0.5 * h_), h6 = real(h_ / 6.0);
    bool dead = false;
    for (int e = team; e < E && !dead; e += T) {
        RmtMember m;
        rmt_load_member(members + (size_t)e * RMT_NM, m);
        RmtCarry carry;
        rmt_carry_inlet(m, carry);
        real* ye = y + (size_t)e * RMT_V * N;
        real ys[RMT_NPT][RMT_V], k[RMT_NPT][RMT_V];
#if RMT_LDS_STATE < 1
        real y0[RMT_NPT][RMT_V];
#define Y0(j, i) y0[j][i]
#else
#define Y0(j, i) s_y0[i][lnode0 + (j)]
#endif
#if RMT_LDS_STATE < 2
        real acc[RMT_NPT][RMT_V];
#define ACC(j, i) acc[j][i]
#else
#define ACC(j, i) s_acc[i][lnode0 + (j)]
#endif
#pragma unroll
        for (int j = 0; j < RMT_NPT; ++j) {
            rmt_safe_state(m, ys[j]);
            if (j < nvalid) {
#pragma unroll
                for (int i = 0; i < RMT_V; ++i) ys[j][i] = ye[(size_t)i * N + node0 + j];
            }
#pragma unroll
            for (int i = 0; i < RMT_V; ++i) Y0(j, i) = ys[j][i];
        }
        rmt_flags_t flag;
        rmt_flags_clear(flag);
        unsigned lflag = 0u;
        for (long long step = 0; step < nsteps && !dead; ++step) {
#pragma unroll 1
            for (int s = 0; s < 4; ++s) {
                ctx.q += 1ull;
                rmt_rhs_block<RMT_NPT, false, true>(m, sh, s & 1, ys, nvalid, carry, k, flag, &ctx);
                if (ctx.abort) { dead = true; break; }
                if (s < 3) {
                    const real cn = (s == 2) ? h : hh;
                    const real wk = (s == 0) ? real(1) : real(2);
#pragma unroll
                    for (int j = 0; j < RMT_NPT; ++j)
#pragma unroll
                        for (int i = 0; i < RMT_V; ++i) {
                            ACC(j, i) = (s == 0) ? k[j][i] : ACC(j, i) + wk * k[j][i];
                            ys[j][i] = Y0(j, i) + k[j][i] * cn;
                        }
                } else {
#pragma unroll
                    for (int j = 0; j < RMT_NPT; ++j)
#pragma unroll
                        for (int i = 0; i < RMT_V; ++i) {
                            const real yn = Y0(j, i) + h6 * (ACC(j, i) + k[j][i]);
                            Y0(j, i) = yn;
                            ys[j][i] = yn;
                        }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < RMT_NPT; ++j) {
            if (j < nvalid) {
#pragma unroll
                for (int i = 0; i < RMT_V; ++i) ye[(size_t)i * N + node0 + j] = ys[j][i];
                lflag |= rmt_finite_flag(ys[j]);
            }
        }
        lflag |= rmt_flags_bits(flag);
        if (dead) lflag |= RMT_FLAG_STEP;
        if (lflag) atomicOr(&flags[e], lflag);
#undef Y0
#undef ACC
    }
    if (dead && threadIdx.x == 0) {       // let the rest of the chain drain
        __hip_atomic_store(ctx.sync_out, RMT_CHAIN_POISON, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (c > 0)
            __hip_atomic_store(ctx.sync_in + 16, RMT_CHAIN_POISON, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ===================================================================== kernel: RK4, state in memory
// Same integrator for any N: the workgroup walks its reactor in blocks of RMT_BLOCK nodes per stage.
// work = 3 arrays [E][V][N]: stage state A, stage state B, K accumulator.  Every thread only ever
// touches its own nodes' entries, so no global-memory synchronisation is needed.
extern "C" __global__ __launch_bounds__(RMT_BLOCK) void rmt_n2_rk4_mem(
        real* __restrict__ y, real* __restrict__ work, const double* __restrict__ members,
        const int N, const int E, const double h_, const long long nsteps,
        unsigned* __restrict__ flags) {
    __shared__ RmtShared sh;
    rmt_math_init();
    const int e = blockIdx.x;
#if RMT_MEMBER_LDS
    __shared__ RmtMember m;      // read back with broadcast ds_reads: frees ~2*(15+S) SGPRs
    if (threadIdx.x == 0) rmt_load_member(members + (size_t)e * RMT_NM, m);
    __syncthreads();
#else
    RmtMember m;
    rmt_load_member(members + (size_t)e * RMT_NM, m);
#endif
    const size_t per = (size_t)RMT_V * N, tot = per * E;
    real* ye = y + e * per;
    real* wa = work + e * per;
    real* wb = work + tot + e * per;
    real* wacc = work + 2 * tot + e * per;
    const real h = real(h_), hh = real(0.5 * h_), h6 = real(h_ / 6.0);
    rmt_flags_t flag;
    rmt_flags_clear(flag);
    unsigned lflag = 0u;
    int ph = 0;
    for (long long step = 0; step < nsteps; ++step) {
#pragma unroll 1
        for (int s = 0; s < 4; ++s) {
            const real* src = (s == 0) ? ye : ((s == 2) ? wb : wa);
            real* dst = (s == 1) ? wb : wa;
            const real cn = (s == 2) ? h : hh;
            const real wk = (s == 1 || s == 2) ? real(2) : real(1);
            RmtCarry carry;
            rmt_carry_inlet(m, carry);
            for (int base = 0; base < N; base += RMT_BLOCK, ph ^= 1) {
                const int node = base + (int)threadIdx.x;
                const bool valid = node < N;
                real ys[1][RMT_V], k[1][RMT_V];
                rmt_safe_state(m, ys[0]);
                if (valid) {
#pragma unroll
                    for (int i = 0; i < RMT_V; ++i) ys[0][i] = src[(size_t)i * N + node];
                }
                rmt_rhs_block<1, true>(m, sh, ph, ys, valid ? 1 : 0, carry, k, flag);
                if (valid) {
#pragma unroll
                    for (int i = 0; i < RMT_V; ++i) {
                        const size_t o = (size_t)i * N + node;
                        const real a = (s == 0) ? k[0][i] : wacc[o] + wk * k[0][i];
                        if (s < 3) {
                            wacc[o] = a;
                            dst[o] = ye[o] + k[0][i] * cn;
                        } else {
                            const real yn = ye[o] + h6 * a;
                            ye[o] = yn;
                            lflag |= __builtin_isfinite(yn) ? 0u : RMT_FLAG_NONFINITE;
                        }
                    }
                }
            }
        }
    }
    lflag |= rmt_flags_bits(flag);
    if (lflag) atomicOr(&flags[e], lflag);
}

// ===================================================================== kernel: adaptive RK45, state in memory
// Dormand-Prince 5(4) (the pair behind SciPy's "RK45", which the reference reaches through
// solve_ivp(method=...) at pbHomoReactor.py:3609) with PER-REACTOR step control: every workgroup
// carries its own t, h and accept/reject history.  Controller (restated in oracle/n2_oracle.py
// rk45, which the parity tests compare against):
//   err = max_i |e_i| / (atol + rtol*max(|y_i|, |ynew_i|));  accept iff err <= 1
//   h_new = h * clip(0.9*err^-0.2, 0.2, 5)  (no growth after a rejection);  a stage that produces a
//   non-finite value (Python would raise inside a lambda: SURVEY.md Appendix C) rejects with h/4.
// work = 8 arrays [E][V][N]: K1..K7 and the trial state.  FSAL: K7 of an accepted step is K1 of
// the next.  Each thread only touches its own nodes' entries.
__device__ static const double RMT_DP_A[7][6] = {
    {0, 0, 0, 0, 0, 0},
    {1.0 / 5, 0, 0, 0, 0, 0},
    {3.0 / 40, 9.0 / 40, 0, 0, 0, 0},
    {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0, 0},
    {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0, 0},
    {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656, 0},
    {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84}};
__device__ static const double RMT_DP_E[7] = {      // b5 - b4
    35.0 / 384 - 5179.0 / 57600, 0, 500.0 / 1113 - 7571.0 / 16695, 125.0 / 192 - 393.0 / 640,
    -2187.0 / 6784 + 92097.0 / 339200, 11.0 / 84 - 187.0 / 2100, -1.0 / 40};

__device__ __forceinline__ double rmt_block_max(RmtShared& sh, const int buf, double v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmax(v, __shfl_xor(v, d));
    if ((threadIdx.x & 63) == 0) sh.red[buf][threadIdx.x >> 6] = v;
    __syncthreads();
    double r = sh.red[buf][0];
    for (int w = 1; w < RMT_NW; ++w) r = fmax(r, sh.red[buf][w]);
    return r;
}

extern "C" __global__ __launch_bounds__(RMT_BLOCK) void rmt_n2_rk45_mem(
        real* __restrict__ y, real* __restrict__ work, const double* __restrict__ members,
        const int N, const int E, const double t0, const double t1, const double rtol,
        const double atol, const double h0, const long long max_steps,
        double* __restrict__ stats /* [E][4]: t_end, h_last, accepted(i64), rejected(i64) */,
        unsigned* __restrict__ flags) {
    __shared__ RmtShared sh;
    rmt_math_init();
    const int e = blockIdx.x;
    RmtMember m;
    rmt_load_member(members + (size_t)e * RMT_NM, m);
    const size_t per = (size_t)RMT_V * N, tot = per * E;
    real* ye = y + e * per;
    real* wk = work + e * per;                  // K_{j+1}[o] = wk[j*tot + o]
    real* yn = work + 7 * tot + e * per;
    rmt_flags_t flag, trial;
    rmt_flags_clear(flag);
    unsigned lflag = 0u;
    int ph = 0, rp = 0;
    double t = t0, h = h0;
    long long nacc = 0, nrej = 0;
    bool have_k1 = false;
    while (t < t1 && nacc + nrej < max_steps) {
        bool last = false;
        if (t + h >= t1) { h = t1 - t; last = true; }
        rmt_flags_clear(trial);
        double errloc = 0.0;
        bool bad = false;
#pragma unroll 1
        for (int s = have_k1 ? 1 : 0; s < 7; ++s) {
            // stage input: y + h*sum_j a[s][j] K_j ; s == 6 is the 5th-order solution itself
            RmtCarry carry;
            rmt_carry_inlet(m, carry);
            for (int base = 0; base < N; base += RMT_BLOCK, ph ^= 1) {
                const int node = base + (int)threadIdx.x;
                const bool valid = node < N;
                real ys[1][RMT_V], k[1][RMT_V];
                rmt_safe_state(m, ys[0]);
                if (valid) {
#pragma unroll
                    for (int i = 0; i < RMT_V; ++i) {
                        const size_t o = (size_t)i * N + node;
                        double acc = 0.0;
                        for (int j = 0; j < s; ++j) acc += RMT_DP_A[s][j] * (double)wk[(size_t)j * tot + o];
                        const double v = (double)ye[o] + h * acc;
                        ys[0][i] = real(v);
                        if (s == 6) yn[o] = real(v);
                    }
                }
                rmt_rhs_block<1, true>(m, sh, ph, ys, valid ? 1 : 0, carry, k, trial);
                if (valid) {
#pragma unroll
                    for (int i = 0; i < RMT_V; ++i) {
                        const size_t o = (size_t)i * N + node;
                        wk[(size_t)s * tot + o] = k[0][i];
                        bad |= !__builtin_isfinite(k[0][i]);
                        if (s == 6) {
                            double ev = 0.0;
                            for (int j = 0; j < 7; ++j) ev += RMT_DP_E[j] * (double)wk[(size_t)j * tot + o];
                            const double sc = atol + rtol * fmax(fabs((double)ye[o]), fabs((double)ys[0][i]));
                            errloc = fmax(errloc, fabs(h * ev) / sc);
                        }
                    }
                }
            }
            // a non-finite stage derivative: give up on this step early (wave-uniform decision)
            const double anybad = rmt_block_max(sh, rp, bad ? 1.0 : 0.0);
            rp ^= 1;
            if (anybad > 0.0) { bad = true; break; }
            have_k1 = true;     // K1 = f(y) is in place once stage 0 has run
        }
        double err = 0.0;
        if (!bad) {
            err = rmt_block_max(sh, rp, __builtin_isfinite(errloc) ? errloc : 1.0e300);
            rp ^= 1;
        }
        double fac;
        if (bad) {
            fac = 0.25;
            ++nrej;
        } else if (err <= 1.0) {
            t = last ? t1 : t + h;
            ++nacc;
            fac = (err == 0.0) ? 5.0 : fmin(5.0, fmax(0.2, 0.9 * pow(err, -0.2)));
            // accept: y <- ynew, K1 <- K7 (FSAL)
            for (int base = 0; base < N; base += RMT_BLOCK) {
                const int node = base + (int)threadIdx.x;
                if (node < N) {
#pragma unroll
                    for (int i = 0; i < RMT_V; ++i) {
                        const size_t o = (size_t)i * N + node;
                        ye[o] = yn[o];
                        wk[o] = wk[6 * tot + o];
                    }
                }
            }
            rmt_flags_merge(flag, trial);
        } else {
            ++nrej;
            fac = fmax(0.2, 0.9 * pow(err, -0.2));     // K1 = f(y) stays valid after a rejection
        }
        h = fmax(h * fac, 1e-14);
    }
    if (t < t1) lflag |= RMT_FLAG_STEP;
    lflag |= rmt_flags_bits(flag);
    if (threadIdx.x == 0) {
        stats[(size_t)e * 4 + 0] = t;
        stats[(size_t)e * 4 + 1] = h;
        ((long long*)stats)[(size_t)e * 4 + 2] = nacc;
        ((long long*)stats)[(size_t)e * 4 + 3] = nrej;
    }
    if (lflag) atomicOr(&flags[e], lflag);
}

// ===================================================================== kernel: Adams multistep, state in memory
// The reference's other two hand-written integrators (PyREMOT/solvers/odeSolver.py:43-102):
// AdBash3 (method 0) and the AB3 predictor / AM4 corrector PreCorr3 (method 1) that runN2 selects
// with ivp == "AM" (pbHomoReactor.py:3598-3601).  Same start-up as there: two RK4 steps give
// y_1, y_2; K2 = f(y_0), K1 = f(y_1); then for i = 2..n-1
//   K3,K2 <- K2,K1; K1 = f(y_i); y_{i+1} = y_i + h(23K1-16K2+5K3)/12            [AB3 / predictor]
//   K0 = f(y_{i+1}); y_{i+1} = y_i + h(9K0+19K1-5K2+K3)/24                      [PreCorr3 only]
// work = 8 arrays [E][V][N]: RK4 scratch (3), K ring (3), K0, predictor.
__device__ __forceinline__ void rmt_eval_mem(const RmtMember& m, RmtShared& sh, int& ph,
                                             const real* src, real* dst,   // may alias (in-place)
                                             const int N, rmt_flags_t& flag) {
    RmtCarry carry;
    rmt_carry_inlet(m, carry);
    for (int base = 0; base < N; base += RMT_BLOCK, ph ^= 1) {
        const int node = base + (int)threadIdx.x;
        const bool valid = node < N;
        real ys[1][RMT_V], k[1][RMT_V];
        rmt_safe_state(m, ys[0]);
        if (valid) {
#pragma unroll
            for (int i = 0; i < RMT_V; ++i) ys[0][i] = src[(size_t)i * N + node];
        }
        rmt_rhs_block<1, true>(m, sh, ph, ys, valid ? 1 : 0, carry, k, flag);
        if (valid) {
#pragma unroll
            for (int i = 0; i < RMT_V; ++i) dst[(size_t)i * N + node] = k[0][i];
        }
    }
}

__device__ __forceinline__ void rmt_rk4_step_mem(const RmtMember& m, RmtShared& sh, int& ph,
                                                 real* __restrict__ ye, real* __restrict__ wa,
                                                 real* __restrict__ wb, real* __restrict__ wacc,
                                                 const int N, const real h, rmt_flags_t& flag) {
    const real hh = real(0.5) * h, h6 = h / real(6);
#pragma unroll 1
    for (int s = 0; s < 4; ++s) {
        const real* src = (s == 0) ? ye : ((s == 2) ? wb : wa);
        real* dst = (s == 1) ? wb : wa;
        rmt_eval_mem(m, sh, ph, src, (s == 0) ? wacc : dst, N, flag);   // K into wacc (s=0) or dst
        for (int base = 0; base < N; base += RMT_BLOCK) {
            const int node = base + (int)threadIdx.x;
            if (node < N) {
#pragma unroll
                for (int i = 0; i < RMT_V; ++i) {
                    const size_t o = (size_t)i * N + node;
                    if (s == 0) {
                        wa[o] = ye[o] + wacc[o] * hh;                    // acc = K1 stays in wacc
                    } else if (s == 1) {
                        const real k = wb[o];
                        wacc[o] += real(2) * k;
                        wb[o] = ye[o] + k * hh;
                    } else if (s == 2) {
                        const real k = wa[o];
                        wacc[o] += real(2) * k;
                        wa[o] = ye[o] + k * h;
                    } else {
                        ye[o] += h6 * (wacc[o] + wa[o]);
                    }
                }
            }
        }
    }
}

extern "C" __global__ __launch_bounds__(RMT_BLOCK) void rmt_n2_multistep_mem(
        real* __restrict__ y, real* __restrict__ work, const double* __restrict__ members,
        const int N, const int E, const double h_, const long long nsteps, const int method,
        unsigned* __restrict__ flags) {
    __shared__ RmtShared sh;
    rmt_math_init();
    const int e = blockIdx.x;
    RmtMember m;
    rmt_load_member(members + (size_t)e * RMT_NM, m);
    const size_t per = (size_t)RMT_V * N, tot = per * E;
    real* ye = y + e * per;
    real* wa = work + e * per;
    real* wb = work + tot + e * per;
    real* wacc = work + 2 * tot + e * per;
    real* k1 = work + 3 * tot + e * per;
    real* k2 = work + 4 * tot + e * per;
    real* k3 = work + 5 * tot + e * per;
    real* k0 = work + 6 * tot + e * per;
    real* yp = work + 7 * tot + e * per;
    const real h = real(h_);
    rmt_flags_t flag;
    rmt_flags_clear(flag);
    unsigned lflag = 0u;
    int ph = 0;
    rmt_eval_mem(m, sh, ph, ye, k2, N, flag);                    // K2 = f(y_0)
    rmt_rk4_step_mem(m, sh, ph, ye, wa, wb, wacc, N, h, flag);   // y_1
    rmt_eval_mem(m, sh, ph, ye, k1, N, flag);                    // K1 = f(y_1)
    rmt_rk4_step_mem(m, sh, ph, ye, wa, wb, wacc, N, h, flag);   // y_2
    for (long long i = 2; i < nsteps; ++i) {
        real* t = k3; k3 = k2; k2 = k1; k1 = t;                  // K3 = K2; K2 = K1
        rmt_eval_mem(m, sh, ph, ye, k1, N, flag);                // K1 = f(y_i)
        real* dst = (method == 1) ? yp : ye;
        for (int base = 0; base < N; base += RMT_BLOCK) {
            const int node = base + (int)threadIdx.x;
            if (node < N) {
#pragma unroll
                for (int v = 0; v < RMT_V; ++v) {
                    const size_t o = (size_t)v * N + node;
                    dst[o] = ye[o] + h * (real(23) * k1[o] - real(16) * k2[o] + real(5) * k3[o]) / real(12);
                }
            }
        }
        if (method == 1) {
            rmt_eval_mem(m, sh, ph, yp, k0, N, flag);            // K0 = f(predictor)
            for (int base = 0; base < N; base += RMT_BLOCK) {
                const int node = base + (int)threadIdx.x;
                if (node < N) {
#pragma unroll
                    for (int v = 0; v < RMT_V; ++v) {
                        const size_t o = (size_t)v * N + node;
                        ye[o] = ye[o] + h * (real(9) * k0[o] + real(19) * k1[o] - real(5) * k2[o] + k3[o]) / real(24);
                    }
                }
            }
        }
    }
    for (int base = 0; base < N; base += RMT_BLOCK) {
        const int node = base + (int)threadIdx.x;
        if (node < N) {
#pragma unroll
            for (int v = 0; v < RMT_V; ++v)
                lflag |= __builtin_isfinite(ye[(size_t)v * N + node]) ? 0u : RMT_FLAG_NONFINITE;
        }
    }
    lflag |= rmt_flags_bits(flag);
    if (lflag) atomicOr(&flags[e], lflag);
}

// ===================================================================== kernel: stiff integrator (Rosenbrock), state in memory
// SURVEY.md section 8(f) rank 2.  The explicit steppers are stability-limited to dt ~ 3e-6 s on
// the hot DME bed (250 000 steps for 0.5 s).  This kernel is a linearly-implicit 4th-order
// Rosenbrock method with an embedded 3rd-order error estimate - the Kaps-Rentrop scheme with
// Shampine's parameters (gamma = 1/2; 3 RHS evaluations, 4 linear solves per step) - and
// per-reactor step control like rmt_n2_rk45_mem.
//   Jacobian: the method of lines RHS of node z depends on its own state, on the clamped state of
// node z-1 (upwind) and - weakly - on all upstream nodes through the pressure march.  J is taken
// as block lower-bidiagonal: D_z = d f_z / d y_z by forward differences of the node function at
// frozen (P_z, upstream state), L_z = d f_z / d y_{z-1} = diag(F1/dz or FT/dz) (0 where the clamp
// is active); the pressure coupling is dropped (measured: same step counts and accuracy as the
// full finite-difference Jacobian, tools/ros_prototype.py).
//   Solve (I/(gamma h) - J) x = b: per node x_z = Ainv_z (b_z + L_z x_{z-1}), Ainv_z the explicit
// VxV inverse (Gauss-Jordan in registers); the upstream coupling is resolved by Jacobi sweeps
// x <- p + Ainv L x_up whose contraction factor is <= c/(1/(gamma h) + c), c = F1/dz (h is capped
// so that this is <= 1/2); node blocks are walked in order, so the block boundary value is exact.
// work = 8 vector arrays [E][V][N] (F, G1..G4, stage state, new state, b) + Ainv [E][V*V][N] +
// per-node clamp masks.
#define RMT_ROS_GAM 0.5
__device__ __forceinline__ void rmt_shift_up(RmtShared& sh, const int buf, const real (&x)[RMT_V],
                                             const real (&carry_in)[RMT_V], real (&xup)[RMT_V],
                                             real (&carry_out)[RMT_V]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < RMT_V; ++i) xup[i] = rmt_lane_up1(x[i]);
    if (lane == 63) {
#pragma unroll
        for (int i = 0; i < RMT_V; ++i) sh.bnd[buf][wave][i] = x[i];
    }
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < RMT_V; ++i) xup[i] = (wave == 0) ? carry_in[i] : sh.bnd[buf][wave - 1][i];
    }
#pragma unroll
    for (int i = 0; i < RMT_V; ++i) carry_out[i] = sh.bnd[buf][RMT_NW - 1][i];
}

// in-place inverse of a VxV matrix held in registers (Gauss-Jordan, no pivoting: the matrix is
// I/(gamma h) - D with a dominant positive diagonal); returns the smallest |pivot|
__device__ __forceinline__ real rmt_invert(real (&a)[RMT_V][RMT_V]) {
    real pmin = real(__builtin_inf());
#pragma unroll
    for (int p = 0; p < RMT_V; ++p) {
        pmin = rmt_min(pmin, rmt_abs(a[p][p]));
        const real ip = rmt_rcp(a[p][p]);
        a[p][p] = real(1);
#pragma unroll
        for (int c = 0; c < RMT_V; ++c) a[p][c] *= ip;
#pragma unroll
        for (int r = 0; r < RMT_V; ++r) {
            if (r != p) {
                const real f = a[r][p];
                a[r][p] = real(0);
#pragma unroll
                for (int c = 0; c < RMT_V; ++c) a[r][c] -= f * a[p][c];
            }
        }
    }
    return pmin;
}

// F = f(src) for all node blocks; if WITH_JAC also Ainv(h) and the clamp masks of the upwind coupling
template <bool WITH_JAC>
__device__ __forceinline__ void rmt_ros_eval(const RmtMember& m, RmtShared& sh, int& ph,
                                             const real* src, real* dstF, real* ainv, unsigned* mask,
                                             const int N, const real inv_gh, rmt_flags_t& flag,
                                             real& pivmin) {
    RmtCarry carry;
    rmt_carry_inlet(m, carry);
    for (int base = 0; base < N; base += RMT_BLOCK, ph ^= 1) {
        const int node = base + (int)threadIdx.x;
        const bool valid = node < N;
        real ys[1][RMT_V], k[1][RMT_V];
        rmt_safe_state(m, ys[0]);
        if (valid) {
#pragma unroll
            for (int i = 0; i < RMT_V; ++i) ys[0][i] = src[(size_t)i * N + node];
        }
        RmtLocal lc[1];
        rmt_rhs_block<1, true, false>(m, sh, ph, ys, valid ? 1 : 0, carry, k, flag, nullptr,
                                      WITH_JAC ? lc : nullptr);
        if (valid) {
#pragma unroll
            for (int i = 0; i < RMT_V; ++i) dstF[(size_t)i * N + node] = k[0][i];
        }
        if (WITH_JAC) {
            real a[RMT_V][RMT_V];
            rmt_flags_t scratch_flag;
            rmt_flags_clear(scratch_flag);
#pragma unroll
            for (int c = 0; c < RMT_V; ++c) {                    // column c of D_z by a forward difference
                real yp[RMT_V], kp[RMT_V];
#pragma unroll
                for (int i = 0; i < RMT_V; ++i) yp[i] = ys[0][i];
                const real d = real(RMT_FP32 ? 3e-4 : 1.5e-8) * rmt_max(rmt_abs(ys[0][c]), real(1e-3));
                yp[c] += d;
                const real id = rmt_rcp(yp[c] - ys[0][c]);
                RmtNode ndp;
                (void)rmt_node_pre(m, yp, ndp);
                rmt_node_post(m, ndp, yp, lc[0].up, lc[0].P, kp, scratch_flag);
#pragma unroll
                for (int r = 0; r < RMT_V; ++r) a[r][c] = -(kp[r] - k[0][r]) * id;
            }
#pragma unroll
            for (int i = 0; i < RMT_V; ++i) a[i][i] += inv_gh;   // A = I/(gamma h) - D
            const real pv = rmt_invert(a);
            if (valid) {
                pivmin = rmt_min(pivmin, pv);
#pragma unroll
                for (int r = 0; r < RMT_V; ++r)
#pragma unroll
                    for (int c = 0; c < RMT_V; ++c) ainv[(size_t)(r * RMT_V + c) * N + node] = a[r][c];
                unsigned mk = 0u;                                 // upwind coupling active? (:4093 clamp)
                if (node > 0) {
#pragma unroll
                    for (int i = 0; i < RMT_S; ++i) mk |= (lc[0].up[i] > RMT_EPS) ? (1u << i) : 0u;
#if !RMT_ISO
                    mk |= 1u << RMT_S;
#endif
                }
                mask[node] = mk;
            }
        }
    }
}

// x = (I/(gamma h) - J)^-1 b with the stored inverses; b and x may alias
__device__ __forceinline__ void rmt_ros_solve(const RmtMember& m, RmtShared& sh, int& ph,
                                              const real* b, real* x, const real* ainv,
                                              const unsigned* mask, const int N, const int sweeps) {
    real carry[RMT_V];
#pragma unroll
    for (int i = 0; i < RMT_V; ++i) carry[i] = real(0);          // nothing upstream of node 0
    const real cs = m.f1 * m.inv_dz, ct = m.ft * m.inv_dz;       // d f_z / d y_{z-1}
    for (int base = 0; base < N; base += RMT_BLOCK) {
        const int node = base + (int)threadIdx.x;
        const bool valid = node < N;
        real ai[RMT_V][RMT_V], p[RMT_V], xv[RMT_V], l[RMT_V];
        unsigned mk = 0u;
        if (valid) mk = mask[node];
#pragma unroll
        for (int r = 0; r < RMT_V; ++r) {
            real bv = real(0);
            if (valid) bv = b[(size_t)r * N + node];
            xv[r] = bv;                                           // reuse xv as b for the moment
#pragma unroll
            for (int c = 0; c < RMT_V; ++c) ai[r][c] = valid ? ainv[(size_t)(r * RMT_V + c) * N + node] : real(0);
            l[r] = ((mk >> r) & 1u) ? ((r < RMT_S) ? cs : ct) : real(0);
        }
#pragma unroll
        for (int r = 0; r < RMT_V; ++r) {
            real acc = real(0);
#pragma unroll
            for (int c = 0; c < RMT_V; ++c) acc += ai[r][c] * xv[c];
            p[r] = acc;                                           // p = Ainv b
        }
        // fold the coupling into the matrix once: ai <- Ainv diag(l)
#pragma unroll
        for (int r = 0; r < RMT_V; ++r)
#pragma unroll
            for (int c = 0; c < RMT_V; ++c) ai[r][c] *= l[c];
#pragma unroll
        for (int r = 0; r < RMT_V; ++r) xv[r] = p[r];
        real cout[RMT_V];
        for (int it = 0; it <= sweeps; ++it, ph ^= 1) {           // last pass only publishes the boundary
            real xup[RMT_V];
            rmt_shift_up(sh, ph, xv, carry, xup, cout);
            if (it < sweeps) {
#pragma unroll
                for (int r = 0; r < RMT_V; ++r) {
                    real acc = p[r];
#pragma unroll
                    for (int c = 0; c < RMT_V; ++c) acc += ai[r][c] * xup[c];
                    xv[r] = acc;
                }
            }
        }
        // ragged last block: the boundary value is not needed any more
#pragma unroll
        for (int i = 0; i < RMT_V; ++i) carry[i] = cout[i];
        if (valid) {
#pragma unroll
            for (int r = 0; r < RMT_V; ++r) x[(size_t)r * N + node] = xv[r];
        }
    }
}

extern "C" __global__ __launch_bounds__(RMT_BLOCK) void rmt_n2_ros4_mem(
        real* __restrict__ y, real* __restrict__ work, unsigned* __restrict__ maskbuf,
        const double* __restrict__ members, const int N, const int E, const double t0,
        const double t1, const double rtol, const double atol, const double h0,
        const long long max_steps, double* __restrict__ stats, unsigned* __restrict__ flags) {
    __shared__ RmtShared sh;
    rmt_math_init();
    const int e = blockIdx.x;
    RmtMember m;
    rmt_load_member(members + (size_t)e * RMT_NM, m);
    const size_t per = (size_t)RMT_V * N, tot = per * E;
    real* ye = y + e * per;
    real* F = work + 0 * tot + e * per;
    real* G1 = work + 1 * tot + e * per;
    real* G2 = work + 2 * tot + e * per;
    real* G3 = work + 3 * tot + e * per;
    real* G4 = work + 4 * tot + e * per;
    real* YS = work + 5 * tot + e * per;
    real* B = work + 6 * tot + e * per;
    real* ainv = work + 7 * tot + (size_t)e * RMT_V * RMT_V * N;
    unsigned* mask = maskbuf + (size_t)e * N;
    // Kaps-Rentrop / Shampine parameters (gamma = 1/2)
    const double A21 = 2.0, A31 = 48.0 / 25, A32 = 6.0 / 25, C21 = -8.0, C31 = 372.0 / 25, C32 = 12.0 / 5,
                 C41 = -112.0 / 125, C42 = -54.0 / 125, C43 = -2.0 / 5, B1 = 19.0 / 9, B2 = 0.5,
                 B3 = 25.0 / 108, B4 = 125.0 / 108, E1 = 17.0 / 54, E2 = 7.0 / 36, E4 = 125.0 / 108;
    const double cmax = (double)rmt_max(m.f1, m.ft) * (double)m.inv_dz;
    const double hcap = 1.0 / (RMT_ROS_GAM * cmax);               // contraction factor <= 1/2
    rmt_flags_t flag, trial;
    rmt_flags_clear(flag);
    unsigned lflag = 0u;
    int ph = 0, rp = 0;
    double t = t0, h = fmin(h0, hcap);
    long long nacc = 0, nrej = 0;
    while (t < t1 && nacc + nrej < max_steps) {
        bool last = false;
        if (t + h >= t1) { h = t1 - t; last = true; }
        rmt_flags_clear(trial);
        const real inv_gh = real(1.0 / (RMT_ROS_GAM * h));
        const double rho = cmax / (1.0 / (RMT_ROS_GAM * h) + cmax);
        int sweeps = (int)ceil(-27.6 / log(rho)) + 1;             // rho^sweeps <= 1e-12
        sweeps = sweeps < 2 ? 2 : (sweeps > 60 ? 60 : sweeps);
        real pivmin = real(__builtin_inf());
        const real ih = real(1.0 / h);
        // stage 1
        rmt_ros_eval<true>(m, sh, ph, ye, F, ainv, mask, N, inv_gh, trial, pivmin);
        rmt_ros_solve(m, sh, ph, F, G1, ainv, mask, N, sweeps);
        // stage 2
        for (int base = 0; base < N; base += RMT_BLOCK) {
            const int node = base + (int)threadIdx.x;
            if (node < N) {
#pragma unroll
                for (int i = 0; i < RMT_V; ++i) {
                    const size_t o = (size_t)i * N + node;
                    YS[o] = ye[o] + real(A21) * G1[o];
                }
            }
        }
        rmt_ros_eval<false>(m, sh, ph, YS, F, ainv, mask, N, inv_gh, trial, pivmin);
        for (int base = 0; base < N; base += RMT_BLOCK) {
            const int node = base + (int)threadIdx.x;
            if (node < N) {
#pragma unroll
                for (int i = 0; i < RMT_V; ++i) {
                    const size_t o = (size_t)i * N + node;
                    B[o] = F[o] + real(C21) * G1[o] * ih;
                }
            }
        }
        rmt_ros_solve(m, sh, ph, B, G2, ainv, mask, N, sweeps);
        // stage 3 (and 4: same RHS)
        for (int base = 0; base < N; base += RMT_BLOCK) {
            const int node = base + (int)threadIdx.x;
            if (node < N) {
#pragma unroll
                for (int i = 0; i < RMT_V; ++i) {
                    const size_t o = (size_t)i * N + node;
                    YS[o] = ye[o] + real(A31) * G1[o] + real(A32) * G2[o];
                }
            }
        }
        rmt_ros_eval<false>(m, sh, ph, YS, F, ainv, mask, N, inv_gh, trial, pivmin);
        for (int base = 0; base < N; base += RMT_BLOCK) {
            const int node = base + (int)threadIdx.x;
            if (node < N) {
#pragma unroll
                for (int i = 0; i < RMT_V; ++i) {
                    const size_t o = (size_t)i * N + node;
                    B[o] = F[o] + (real(C31) * G1[o] + real(C32) * G2[o]) * ih;
                }
            }
        }
        rmt_ros_solve(m, sh, ph, B, G3, ainv, mask, N, sweeps);
        for (int base = 0; base < N; base += RMT_BLOCK) {
            const int node = base + (int)threadIdx.x;
            if (node < N) {
#pragma unroll
                for (int i = 0; i < RMT_V; ++i) {
                    const size_t o = (size_t)i * N + node;
                    B[o] = F[o] + (real(C41) * G1[o] + real(C42) * G2[o] + real(C43) * G3[o]) * ih;
                }
            }
        }
        rmt_ros_solve(m, sh, ph, B, G4, ainv, mask, N, sweeps);
        // new state (into YS) and error estimate
        double errloc = 0.0;
        bool bad = false;
        for (int base = 0; base < N; base += RMT_BLOCK) {
            const int node = base + (int)threadIdx.x;
            if (node < N) {
#pragma unroll
                for (int i = 0; i < RMT_V; ++i) {
                    const size_t o = (size_t)i * N + node;
                    const double g1 = G1[o], g2 = G2[o], g3 = G3[o], g4 = G4[o], yo = ye[o];
                    const double yn = yo + B1 * g1 + B2 * g2 + B3 * g3 + B4 * g4;
                    const double er = E1 * g1 + E2 * g2 + E4 * g4;
                    YS[o] = real(yn);
                    bad |= !__builtin_isfinite(yn);
                    errloc = fmax(errloc, fabs(er) / (atol + rtol * fmax(fabs(yo), fabs(yn))));
                }
            }
        }
        bad |= !(pivmin > real(0));
        const double worst = rmt_block_max(sh, rp, (bad || !__builtin_isfinite(errloc)) ? 1.0e300 : errloc);
        rp ^= 1;
        double fac;
        if (worst >= 1.0e300) {
            fac = 0.25;
            ++nrej;
        } else if (worst <= 1.0) {
            t = last ? t1 : t + h;
            ++nacc;
            fac = (worst > 1.89e-4) ? 0.9 * pow(worst, -0.25) : 1.5 * 5.0;     // 4th order: grow <= 7.5x
            fac = fmin(fac, 5.0);
            for (int base = 0; base < N; base += RMT_BLOCK) {
                const int node = base + (int)threadIdx.x;
                if (node < N) {
#pragma unroll
                    for (int i = 0; i < RMT_V; ++i) {
                        const size_t o = (size_t)i * N + node;
                        ye[o] = YS[o];
                    }
                }
            }
            rmt_flags_merge(flag, trial);
        } else {
            ++nrej;
            fac = fmax(0.2, 0.9 * pow(worst, -1.0 / 3.0));
        }
        h = fmin(fmax(h * fac, 1e-14), hcap);
    }
    if (t < t1) lflag |= RMT_FLAG_STEP;
    lflag |= rmt_flags_bits(flag);
    if (threadIdx.x == 0) {
        stats[(size_t)e * 4 + 0] = t;
        stats[(size_t)e * 4 + 1] = h;
        ((long long*)stats)[(size_t)e * 4 + 2] = nacc;
        ((long long*)stats)[(size_t)e * 4 + 3] = nrej;
    }
    if (lflag) atomicOr(&flags[e], lflag);
}

// ===================================================================== kernel: steady-state model N1 (batched)
// (node function rmt_n1_rhs and the M1_* row layout: see the node-physics section above)
template <int NV>
__device__ __forceinline__ real rmt_invert_n(real (&a)[NV][NV]) {
    real pmin = real(__builtin_inf());
#pragma unroll
    for (int p = 0; p < NV; ++p) {
        pmin = rmt_min(pmin, rmt_abs(a[p][p]));
        const real ip = rmt_rcp(a[p][p]);
        a[p][p] = real(1);
#pragma unroll
        for (int c = 0; c < NV; ++c) a[p][c] *= ip;
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            if (r != p) {
                const real f = a[r][p];
                a[r][p] = real(0);
#pragma unroll
                for (int c = 0; c < NV; ++c) a[r][c] -= f * a[p][c];
            }
        }
    }
    return pmin;
}

extern "C" __global__ __launch_bounds__(64) void rmt_n1_ros4(
        const double* __restrict__ members1, double* __restrict__ out /* [E][nout][V1] */,
        const int E, const int nout, const double rtol, const double atol, const double h0,
        const long long max_steps, double* __restrict__ stats, unsigned* __restrict__ flags) {
    rmt_math_init();
    const int e = blockIdx.x * 64 + (int)threadIdx.x;
    const bool live = e < E;
    const double* mr = members1 + (size_t)(live ? e : 0) * RMT_NM1;
    const double A21 = 2.0, A31 = 48.0 / 25, A32 = 6.0 / 25, C21 = -8.0, C31 = 372.0 / 25, C32 = 12.0 / 5,
                 C41 = -112.0 / 125, C42 = -54.0 / 125, C43 = -2.0 / 5, B1 = 19.0 / 9, B2 = 0.5,
                 B3 = 25.0 / 108, B4 = 125.0 / 108, E1 = 17.0 / 54, E2 = 7.0 / 36, E4 = 125.0 / 108;
    real u[RMT_V1];
#pragma unroll
    for (int i = 0; i < RMT_S; ++i) u[i] = real(mr[M1_CIN + i]);              // :2831-2839
    u[RMT_S] = real(1);
#if !RMT_ISO
    u[RMT_S + 1] = real(0);
#endif
    rmt_flags_t flag, trial;
    rmt_flags_clear(flag);
    double z = 0.0, h = h0;
    long long nacc = 0, nrej = 0;
    int kout = 0;
    if (live) {
#pragma unroll
        for (int i = 0; i < RMT_V1; ++i) out[((size_t)e * nout + 0) * RMT_V1 + i] = (double)u[i];
    }
    kout = 1;
    bool active = live;
    // every lane runs its own step sequence; the loop ends when no lane of the wave is active
    while (__any(active)) {
        if (active) {
            const double zt = (double)kout / (double)(nout - 1);
            bool hit = false;
            double hs = h;
            if (z + hs >= zt) { hs = zt - z; hit = true; }
            rmt_flags_clear(trial);
            real f1[RMT_V1], a[RMT_V1][RMT_V1];
            rmt_n1_rhs(mr, u, f1, trial);
#pragma unroll
            for (int c = 0; c < RMT_V1; ++c) {
                real up[RMT_V1], fp[RMT_V1];
#pragma unroll
                for (int i = 0; i < RMT_V1; ++i) up[i] = u[i];
                const real d = real(RMT_FP32 ? 3e-4 : 1.5e-8) * rmt_max(rmt_abs(u[c]), real(1e-3));
                up[c] += d;
                const real id = rmt_rcp(up[c] - u[c]);
                rmt_flags_t sf;
                rmt_flags_clear(sf);
                rmt_n1_rhs(mr, up, fp, sf);
#pragma unroll
                for (int r = 0; r < RMT_V1; ++r) a[r][c] = -(fp[r] - f1[r]) * id;
            }
            const real inv_gh = real(1.0 / (RMT_ROS_GAM * hs)), ih = real(1.0 / hs);
#pragma unroll
            for (int i = 0; i < RMT_V1; ++i) a[i][i] += inv_gh;
            const real pv = rmt_invert_n<RMT_V1>(a);
            real g1[RMT_V1], g2[RMT_V1], g3[RMT_V1], g4[RMT_V1], b[RMT_V1], us[RMT_V1], f3[RMT_V1];
#define RMT_MATVEC(dst, src)                                                        \
            _Pragma("unroll") for (int r_ = 0; r_ < RMT_V1; ++r_) {                 \
                real acc_ = real(0);                                                \
                _Pragma("unroll") for (int c_ = 0; c_ < RMT_V1; ++c_) acc_ += a[r_][c_] * src[c_]; \
                dst[r_] = acc_;                                                     \
            }
            RMT_MATVEC(g1, f1)
#pragma unroll
            for (int i = 0; i < RMT_V1; ++i) us[i] = u[i] + real(A21) * g1[i];
            rmt_n1_rhs(mr, us, f3, trial);
#pragma unroll
            for (int i = 0; i < RMT_V1; ++i) b[i] = f3[i] + real(C21) * g1[i] * ih;
            RMT_MATVEC(g2, b)
#pragma unroll
            for (int i = 0; i < RMT_V1; ++i) us[i] = u[i] + real(A31) * g1[i] + real(A32) * g2[i];
            rmt_n1_rhs(mr, us, f3, trial);
#pragma unroll
            for (int i = 0; i < RMT_V1; ++i) b[i] = f3[i] + (real(C31) * g1[i] + real(C32) * g2[i]) * ih;
            RMT_MATVEC(g3, b)
#pragma unroll
            for (int i = 0; i < RMT_V1; ++i)
                b[i] = f3[i] + (real(C41) * g1[i] + real(C42) * g2[i] + real(C43) * g3[i]) * ih;
            RMT_MATVEC(g4, b)
#undef RMT_MATVEC
            double worst = 0.0;
            bool bad = !(pv > real(0));
            real un[RMT_V1];
#pragma unroll
            for (int i = 0; i < RMT_V1; ++i) {
                const double yn = (double)u[i] + B1 * (double)g1[i] + B2 * (double)g2[i] + B3 * (double)g3[i] + B4 * (double)g4[i];
                const double er = E1 * (double)g1[i] + E2 * (double)g2[i] + E4 * (double)g4[i];
                un[i] = real(yn);
                bad |= !__builtin_isfinite(yn);
                worst = fmax(worst, fabs(er) / (atol + rtol * fmax(fabs((double)u[i]), fabs(yn))));
            }
            if (bad || !__builtin_isfinite(worst)) {
                h = fmax(hs * 0.25, 1e-14);
                ++nrej;
            } else if (worst <= 1.0) {
                z = hit ? zt : z + hs;
                ++nacc;
#pragma unroll
                for (int i = 0; i < RMT_V1; ++i) u[i] = un[i];
                rmt_flags_merge(flag, trial);
                const double fac = fmin((worst > 1.89e-4) ? 0.9 * pow(worst, -0.25) : 7.5, 5.0);
                h = (hit ? h : hs) * fac;                       // a clipped step does not shrink h
                if (hit) {
#pragma unroll
                    for (int i = 0; i < RMT_V1; ++i) out[((size_t)e * nout + kout) * RMT_V1 + i] = (double)u[i];
                    ++kout;
                }
            } else {
                ++nrej;
                h = fmax(hs * fmax(0.2, 0.9 * pow(worst, -1.0 / 3.0)), 1e-14);
            }
            if (kout >= nout || nacc + nrej >= max_steps) active = false;
        }
    }
    if (live) {
        unsigned lf = rmt_flags_bits(flag);
        if (kout < nout) lf |= RMT_FLAG_STEP;
        stats[(size_t)e * 4 + 0] = z;
        stats[(size_t)e * 4 + 1] = h;
        ((long long*)stats)[(size_t)e * 4 + 2] = nacc;
        ((long long*)stats)[(size_t)e * 4 + 3] = nrej;
        if (lf) atomicOr(&flags[e], lf);
    }
}
#endif  // RMT_HOST_EMULATION
)RMTSRC"
