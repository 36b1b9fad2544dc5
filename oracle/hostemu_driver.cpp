// TEST INFRASTRUCTURE ONLY (lives under oracle/, never linked into the product library).
//
// Compiles the SAME generated source the GPU gets (prelude + the csrc/kernels/*.inc template with the lowered
// kinetics) for the host with g++, using only its per-node physics (rmt_node_pre/rmt_node_post)
// and a plain sequential driver: pressure marched node by node exactly like the reference's loop
// (PyREMOT/docs/pbHomoReactor.py:3892-3979), RK4 as PyREMOT/solvers/odeSolver.py:17-40.
// Purpose: (1) CPU-side check of the lowering + kernel arithmetic against the oracle before any
// GPU run, (2) the "fair" multi-core CPU baseline of bench.py (OpenMP over ensemble members).
#include <cmath>
#include <cstddef>
#include <cstring>
#include <vector>
#define RMT_HOST_EMULATION 1
#define __device__
#define __forceinline__ inline
#define __restrict__
#ifndef INFINITY
#define INFINITY __builtin_inf()
#endif
using std::trunc;
#include RMT_GENERATED_SOURCE

static void emu_rhs_one(const RmtMember& m, const real* y, real* dydt, int N, rmt_flags_t& flag) {
    preal P = m.p0;
    real up[RMT_V];
    for (int i = 0; i < RMT_S; ++i) up[i] = m.cin[i];
#if !RMT_ISO
    up[RMT_S] = m.theta_in;
#endif
    for (int z = 0; z < N; ++z) {
        real ys[RMT_V], k[RMT_V];
        for (int i = 0; i < RMT_V; ++i) ys[i] = y[(size_t)i * N + z];
        RmtNode nd;
        const auto a = rmt_node_pre(m, ys, nd);
        rmt_node_post(m, nd, ys, up, P, k, flag);
        for (int i = 0; i < RMT_V; ++i) dydt[(size_t)i * N + z] = k[i];
        P = rmt_pressure_next(m, a, P);      // N2: affine (pbHomoReactor.py:3979); M2: EOS velocity (pbReactor.py:1055)
        for (int i = 0; i < RMT_S; ++i) up[i] = rmt_max(ys[i], RMT_EPS);
#if !RMT_ISO
        up[RMT_S] = ys[RMT_S];
#endif
    }
}

#ifdef _OPENMP
#include <omp.h>
extern "C" int emu_set_threads(int n) { if (n > 0) omp_set_num_threads(n); return omp_get_max_threads(); }
#else
extern "C" int emu_set_threads(int) { return 1; }
#endif

extern "C" int emu_sizes(int* S, int* R, int* V, int* fp32) {
    *S = RMT_S; *R = RMT_R; *V = RMT_V; *fp32 = RMT_FP32;
    return 0;
}

extern "C" void emu_rhs(const real* y, real* dydt, const double* members, int N, int E, unsigned* flags) {
#pragma omp parallel for schedule(static)
    for (int e = 0; e < E; ++e) {
        RmtMember m;
        rmt_load_member(members + (size_t)e * RMT_NM, m);
        rmt_flags_t f;
        rmt_flags_clear(f);
        emu_rhs_one(m, y + (size_t)e * RMT_V * N, dydt + (size_t)e * RMT_V * N, N, f);
        flags[e] |= rmt_flags_bits(f);
    }
}

extern "C" void emu_n1_rhs(const real* u, real* du, const double* members1, int E, unsigned* flags) {
    for (int e = 0; e < E; ++e) {
        real uu[RMT_V1], dd[RMT_V1];
        for (int i = 0; i < RMT_V1; ++i) uu[i] = u[(size_t)e * RMT_V1 + i];
        rmt_flags_t f;
        rmt_flags_clear(f);
        rmt_n1_rhs(members1 + (size_t)e * RMT_NM1, uu, dd, f);
        for (int i = 0; i < RMT_V1; ++i) du[(size_t)e * RMT_V1 + i] = dd[i];
        flags[e] |= rmt_flags_bits(f);
    }
}

#if RMT_WITH_N1
// Model N1: analytic -d du/d u (rmt_n1_rhs_jac) next to the forward differences of rmt_n1_rhs it replaces, and the
// right-hand side both functions return; jan / jfd are [E][V1][V1], fan / fref [E][V1].
extern "C" void emu_n1_jac(const real* u, const double* members1, int E, double* jan, double* jfd, double* fan,
                           double* fref) {
    rmt_noflags_t nof;
    for (int e = 0; e < E; ++e) {
        real uu[RMT_V1], f0[RMT_V1], fa[RMT_V1], a[RMT_V1][RMT_V1];
        for (int i = 0; i < RMT_V1; ++i) uu[i] = u[(size_t)e * RMT_V1 + i];
        const double* mr = members1 + (size_t)e * RMT_NM1;
        rmt_n1_rhs(mr, uu, f0, nof);
        rmt_n1_rhs_jac(mr, uu, fa, a, nof);
        for (int r = 0; r < RMT_V1; ++r) {
            fan[(size_t)e * RMT_V1 + r] = (double)fa[r];
            fref[(size_t)e * RMT_V1 + r] = (double)f0[r];
            for (int c = 0; c < RMT_V1; ++c) jan[((size_t)e * RMT_V1 + r) * RMT_V1 + c] = (double)a[r][c];
        }
        for (int c = 0; c < RMT_V1; ++c) {
            real up[RMT_V1], fp[RMT_V1];
            for (int i = 0; i < RMT_V1; ++i) up[i] = uu[i];
            const real d = real(RMT_FP32 ? 3e-4 : 1.5e-8) * rmt_max(rmt_abs(uu[c]), real(1e-3));
            up[c] += d;
            rmt_n1_rhs(mr, up, fp, nof);
            for (int r = 0; r < RMT_V1; ++r)
                jfd[((size_t)e * RMT_V1 + r) * RMT_V1 + c] = -(double)(fp[r] - f0[r]) / (double)(up[c] - uu[c]);
        }
    }
}
#endif

extern "C" void emu_rk4(real* y, const double* members, int N, int E, double h_, long long nsteps,
                        unsigned* flags) {
    const real h = real(h_), hh = real(0.5 * h_), h6 = real(h_ / 6.0);
#pragma omp parallel for schedule(dynamic, 1)
    for (int e = 0; e < E; ++e) {
        RmtMember m;
        rmt_load_member(members + (size_t)e * RMT_NM, m);
        const size_t n = (size_t)RMT_V * N;
        real* y0 = y + e * n;
        std::vector<real> ys(n), k(n), acc(n);
        rmt_flags_t f;
        rmt_flags_clear(f);
        for (long long s = 0; s < nsteps; ++s) {
            emu_rhs_one(m, y0, k.data(), N, f);
            for (size_t i = 0; i < n; ++i) { acc[i] = k[i]; ys[i] = y0[i] + k[i] * hh; }
            emu_rhs_one(m, ys.data(), k.data(), N, f);
            for (size_t i = 0; i < n; ++i) { acc[i] += real(2) * k[i]; ys[i] = y0[i] + k[i] * hh; }
            emu_rhs_one(m, ys.data(), k.data(), N, f);
            for (size_t i = 0; i < n; ++i) { acc[i] += real(2) * k[i]; ys[i] = y0[i] + k[i] * h; }
            emu_rhs_one(m, ys.data(), k.data(), N, f);
            for (size_t i = 0; i < n; ++i) y0[i] += h6 * (acc[i] + k[i]);
        }
        flags[e] |= rmt_flags_bits(f);
    }
}

#if RMT_WITH_ROS4
// Analytic node Jacobian of the stiff stepper (rmt_node_jac) next to the forward-difference one it
// replaces, for every node of ONE reactor state: jan / jfd are [N][V][V] = -d f_r / d y_c at the
// frozen (P, upstream state) the RHS evaluation sees.
extern "C" void emu_node_jac(const real* y, const double* member, int N, double* jan, double* jfd,
                             double* lan, double* lfd) {      // lan / lfd: [N][V], model M2 only (may be null)
    (void)lan;
    (void)lfd;
    RmtMember m;
    rmt_load_member(member, m);
    rmt_noflags_t nof;
    preal P = m.p0;
    real up[RMT_V];
    for (int i = 0; i < RMT_S; ++i) up[i] = m.cin[i];
#if !RMT_ISO
    up[RMT_S] = m.theta_in;
#endif
    for (int z = 0; z < N; ++z) {
        real ys[RMT_V], k[RMT_V];
        for (int i = 0; i < RMT_V; ++i) ys[i] = y[(size_t)i * N + z];
        RmtNode nd;
        const auto a0 = rmt_node_pre(m, ys, nd);
        rmt_node_post(m, nd, ys, up, P, k, nof);
        real a[RMT_V][RMT_V], r[RMT_R];
#if RMT_MODEL == 0
        rmt_node_jac(m, nd, ys, P, a, r, nof);
#else
        real lco[RMT_V], kpu[RMT_V];
        rmt_node_jac(m, nd, ys, up, P, a, r, lco, nof);
        if (lfd) {                                  // d f_r / d up_r next to its forward difference
            for (int c = 0; c < RMT_V; ++c) {
                real upp[RMT_V];
                for (int i = 0; i < RMT_V; ++i) upp[i] = up[i];
                const real d = real(RMT_FP32 ? 3e-4 : 1.5e-8) * rmt_max(rmt_abs(up[c]), real(1e-3));
                upp[c] += d;
                rmt_node_post(m, nd, ys, upp, P, kpu, nof);
                lan[(size_t)z * RMT_V + c] = (double)lco[c];
                lfd[(size_t)z * RMT_V + c] = (double)(kpu[c] - k[c]) / (double)(upp[c] - up[c]);
            }
        }
#endif
        for (int rr = 0; rr < RMT_V; ++rr)
            for (int c = 0; c < RMT_V; ++c) jan[((size_t)z * RMT_V + rr) * RMT_V + c] = (double)a[rr][c];
        for (int c = 0; c < RMT_V; ++c) {
            real yp[RMT_V], kp[RMT_V];
            for (int i = 0; i < RMT_V; ++i) yp[i] = ys[i];
            const real d = real(RMT_FP32 ? 3e-4 : 1.5e-8) * rmt_max(rmt_abs(ys[c]), real(1e-3));
            yp[c] += d;
            RmtNode ndp;
            (void)rmt_node_pre(m, yp, ndp);
            rmt_node_post(m, ndp, yp, up, P, kp, nof);
            for (int rr = 0; rr < RMT_V; ++rr)
                jfd[((size_t)z * RMT_V + rr) * RMT_V + c] = -(double)(kp[rr] - k[rr]) / (double)(yp[c] - ys[c]);
        }
        P = rmt_pressure_next(m, a0, P);
        for (int i = 0; i < RMT_S; ++i) up[i] = rmt_max(ys[i], RMT_EPS);
#if !RMT_ISO
        up[RMT_S] = ys[RMT_S];
#endif
    }
}
#endif
