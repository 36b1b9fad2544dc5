// Second issue microbenchmark for gfx950: what do the OTHER fp64 VALU instructions of the node functions cost next to an
// FMA - v_ldexp_f64, v_rcp_f64, v_sqrt_f64, v_frexp_*, conversions, compares, v_max - and a dependent LDS look-up?
// (profiles/round2_issue_model.md prices FMAs, 32-bit VALU, lane moves and SALU; the lean exp / log / div of
// kernels/00_config_math.inc are built from the instructions measured here.)
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/microbench/issue2 tools/microbench/issue2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define OP2(ins, a) asm volatile(ins " %0, %0, %1" : "+v"(a) : "v"(x));
#define OP1(ins, a) asm volatile(ins " %0, %0" : "+v"(a));
#define OPI(ins, a) asm volatile(ins " %0, %0, %1" : "+v"(a) : "v"(ione));
#define CVT(a, w) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(w) : "v"(a));
#define CMP(a) asm volatile("v_cmp_lt_f64 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(w0) : "v"(a), "v"(x), "v"(w1) : "vcc");
#define FMA(a) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y));

#define DECL double a0 = d[0], a1 = d[1], a2 = d[2], a3 = d[3], a4 = d[4], a5 = d[5], a6 = d[6], a7 = d[7]; \
    const double x = d[8], y = d[9]; const int ione = iters >> 30; int w0 = threadIdx.x, w1 = 1; (void)ione; (void)y;
#define FIN out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + w0 + w1;
#define ALL8(M) M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7)
#define X8(B) B B B B B B B B
#define KERNEL(name, BODY) __global__ __launch_bounds__(1024) void name(const double* d, double* out, int iters) { \
    DECL for (int i = 0; i < iters; ++i) { X8(BODY) } FIN }

#define M_MUL(a) OP2("v_mul_f64", a)
#define M_ADD(a) OP2("v_add_f64", a)
#define M_MAX(a) OP2("v_max_f64", a)
#define M_LDEXP(a) OPI("v_ldexp_f64", a)
#define M_RCP(a) OP1("v_rcp_f64", a)
#define M_SQRT(a) OP1("v_sqrt_f64", a)
#define M_RSQ(a) OP1("v_rsq_f64", a)
#define M_FREXPM(a) OP1("v_frexp_mant_f64", a)
#define M_RNDNE(a) OP1("v_rndne_f64", a)
#define M_CVT(a) CVT(a, w0)
#define M_CMP(a) CMP(a)
#define M_FMA(a) FMA(a)
// four FMAs between two slow instructions: does the slow pipe overlap with FMA issue?
#define B_RCP_F FMA(a0) M_RCP(a4) FMA(a1) FMA(a2) M_RCP(a5) FMA(a3) FMA(a0) M_RCP(a6) FMA(a1) FMA(a2) M_RCP(a7) FMA(a3)
#define B_LDEXP_F FMA(a0) M_LDEXP(a4) FMA(a1) FMA(a2) M_LDEXP(a5) FMA(a3) FMA(a0) M_LDEXP(a6) FMA(a1) FMA(a2) M_LDEXP(a7) FMA(a3)
#define B_SQRT_F FMA(a0) M_SQRT(a4) FMA(a1) FMA(a2) M_SQRT(a5) FMA(a3) FMA(a0) M_SQRT(a6) FMA(a1) FMA(a2) M_SQRT(a7) FMA(a3)

KERNEL(k_fma, ALL8(M_FMA))
KERNEL(k_mul, ALL8(M_MUL))
KERNEL(k_add, ALL8(M_ADD))
KERNEL(k_max, ALL8(M_MAX))
KERNEL(k_ldexp, ALL8(M_LDEXP))
KERNEL(k_rcp, ALL8(M_RCP))
KERNEL(k_sqrt, ALL8(M_SQRT))
KERNEL(k_rsq, ALL8(M_RSQ))
KERNEL(k_frexpm, ALL8(M_FREXPM))
KERNEL(k_rndne, ALL8(M_RNDNE))
KERNEL(k_cvt, ALL8(M_CVT))
KERNEL(k_cmp, ALL8(M_CMP))
KERNEL(k_rcp_f, B_RCP_F)
KERNEL(k_ldexp_f, B_LDEXP_F)
KERNEL(k_sqrt_f, B_SQRT_F)

// a table look-up whose address depends on the running value (the exp's 2^(j/2048) read), 8 independent chains
__global__ __launch_bounds__(1024) void k_lds(const double* d, double* out, int iters) {
    __shared__ double tab[2048];
    for (int j = threadIdx.x; j < 2048; j += blockDim.x) tab[j] = 1.0 + 1e-9 * j;
    __syncthreads();
    DECL
    for (int i = 0; i < iters; ++i) {
#define L(a) a = fma(a, x, tab[__double2loint(a + 6755399441055744.0) & 2047]);
        X8(ALL8(L))
    }
    FIN
}

struct K { const char* name; void (*fn)(const double*, double*, int); int n; };
#define CHK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r), __LINE__); exit(1); } } while (0)

int main() {
    K ks[] = {{"8 v_fma_f64", k_fma, 8}, {"8 v_mul_f64", k_mul, 8}, {"8 v_add_f64", k_add, 8}, {"8 v_max_f64", k_max, 8},
              {"8 v_ldexp_f64", k_ldexp, 8}, {"8 v_rcp_f64", k_rcp, 8}, {"8 v_sqrt_f64", k_sqrt, 8}, {"8 v_rsq_f64", k_rsq, 8},
              {"8 v_frexp_mant_f64", k_frexpm, 8}, {"8 v_rndne_f64", k_rndne, 8}, {"8 v_cvt_i32_f64", k_cvt, 8},
              {"8 (v_cmp_lt_f64 + v_cndmask_b32)", k_cmp, 16}, {"8 fma + 4 v_rcp_f64", k_rcp_f, 12},
              {"8 fma + 4 v_ldexp_f64", k_ldexp_f, 12}, {"8 fma + 4 v_sqrt_f64", k_sqrt_f, 12},
              {"8 (add + cvt + ds_read_b64 + fma), address from the value", k_lds, 32}};
    double h[10] = {1, 1, 1, 1, 1, 1, 1, 1, 0.999999, 1e-9};
    double *d, *out;
    CHK(hipMalloc(&d, sizeof h));
    CHK(hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice));
    CHK(hipMalloc(&out, 256 * 1024 * sizeof(double)));
    int clk = 0;
    CHK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0));
    const int iters = 20000;
    printf("| body (x8 per iteration) | waves/SIMD | ns per body | cycles per body @%.2f GHz | cycles / instruction |\n|---|---|---|---|---|\n", clk * 1e-6);
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (auto& k : ks)
        for (int block : {256, 512, 1024}) {
            k.fn<<<256, block>>>(d, out, 100);
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(e0));
            k.fn<<<256, block>>>(d, out, iters);
            CHK(hipEventRecord(e1));
            CHK(hipEventSynchronize(e1));
            float ms = 0;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            const double ns = ms * 1e6 / (iters * 8.0), cyc = ns * clk * 1e-6;
            printf("| %s | %d | %.2f | %.1f | %.2f |\n", k.name, block / 256, ns, cyc, cyc / k.n);
        }
    return 0;
}
