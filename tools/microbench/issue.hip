// Issue-model microbenchmark for gfx950 (MI355X): how do SALU / DPP / lane-move instructions share issue
// slots with fp64 VALU at 1, 2 and 4 waves per SIMD?  Answers the question behind profiles/round2_issue_model.md:
// is the step loop of rmt_n2_rk4_reg (3722 VALU + ~1000 SALU per wave) bound by VALU issue or by total issue?
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/microbench/issue tools/microbench/issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define FMA(a) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y));
#define SMOV(s) asm volatile("s_mov_b32 %0, 0x3ff12345" : "=s"(s));
#define DPP(d, s) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(s));
#define RDL(s, v) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s) : "v"(v));
#define VMOV(d, s) asm volatile("v_mov_b32 %0, %1" : "=v"(d) : "v"(s));
#define SNOP asm volatile("s_nop 0");

#define DECL double a0 = d[0], a1 = d[1], a2 = d[2], a3 = d[3], a4 = d[4], a5 = d[5], a6 = d[6], a7 = d[7]; \
    const double x = d[8], y = d[9]; int s0 = 0, s1 = 0, s2 = 0, s3 = 0; int w0 = threadIdx.x, w1 = 1;
#define FIN out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + s0 + s1 + s2 + s3 + w0 + w1;

// 8 independent fp64 FMAs
#define F8 FMA(a0) FMA(a1) FMA(a2) FMA(a3) FMA(a4) FMA(a5) FMA(a6) FMA(a7)
// 8 FMAs each followed by one s_mov
#define F8S8 FMA(a0) SMOV(s0) FMA(a1) SMOV(s1) FMA(a2) SMOV(s2) FMA(a3) SMOV(s3) FMA(a4) SMOV(s0) FMA(a5) SMOV(s1) FMA(a6) SMOV(s2) FMA(a7) SMOV(s3)
#define F8S4 FMA(a0) SMOV(s0) FMA(a1) FMA(a2) SMOV(s1) FMA(a3) FMA(a4) SMOV(s2) FMA(a5) FMA(a6) SMOV(s3) FMA(a7)
#define F8S2 FMA(a0) SMOV(s0) FMA(a1) FMA(a2) FMA(a3) FMA(a4) SMOV(s2) FMA(a5) FMA(a6) FMA(a7)
// groups: 8 FMAs then 8 s_movs
#define F8_S8 F8 SMOV(s0) SMOV(s1) SMOV(s2) SMOV(s3) SMOV(s0) SMOV(s1) SMOV(s2) SMOV(s3)
// dependent chain of 8 on one accumulator
#define D8 FMA(a0) FMA(a0) FMA(a0) FMA(a0) FMA(a0) FMA(a0) FMA(a0) FMA(a0)
// 2 chains
#define D8x2 FMA(a0) FMA(a1) FMA(a0) FMA(a1) FMA(a0) FMA(a1) FMA(a0) FMA(a1)
#define D8x4 FMA(a0) FMA(a1) FMA(a2) FMA(a3) FMA(a0) FMA(a1) FMA(a2) FMA(a3)
// dependent chain with s_movs in the shadow
#define D8S8 FMA(a0) SMOV(s0) FMA(a0) SMOV(s1) FMA(a0) SMOV(s2) FMA(a0) SMOV(s3) FMA(a0) SMOV(s0) FMA(a0) SMOV(s1) FMA(a0) SMOV(s2) FMA(a0) SMOV(s3)
// 8 FMAs + 4 DPP moves / 4 v_mov / 4 readlane
#define F8P4 FMA(a0) DPP(w0, w1) FMA(a1) FMA(a2) DPP(w0, w1) FMA(a3) FMA(a4) DPP(w0, w1) FMA(a5) FMA(a6) DPP(w0, w1) FMA(a7)
#define F8V4 FMA(a0) VMOV(w0, w1) FMA(a1) FMA(a2) VMOV(w0, w1) FMA(a3) FMA(a4) VMOV(w0, w1) FMA(a5) FMA(a6) VMOV(w0, w1) FMA(a7)
#define F8R4 FMA(a0) RDL(s0, w1) FMA(a1) FMA(a2) RDL(s1, w1) FMA(a3) FMA(a4) RDL(s2, w1) FMA(a5) FMA(a6) RDL(s3, w1) FMA(a7)
#define F8N8 FMA(a0) SNOP FMA(a1) SNOP FMA(a2) SNOP FMA(a3) SNOP FMA(a4) SNOP FMA(a5) SNOP FMA(a6) SNOP FMA(a7) SNOP
// only s_movs
#define S8 SMOV(s0) SMOV(s1) SMOV(s2) SMOV(s3) SMOV(s0) SMOV(s1) SMOV(s2) SMOV(s3)
#define V8 VMOV(w0, w1) VMOV(w0, w1) VMOV(w0, w1) VMOV(w0, w1) VMOV(w0, w1) VMOV(w0, w1) VMOV(w0, w1) VMOV(w0, w1)

#define X8(B) B B B B B B B B
#define KERNEL(name, BODY) __global__ __launch_bounds__(1024) void name(const double* d, double* out, int iters) { \
    DECL for (int i = 0; i < iters; ++i) { X8(BODY) } FIN }

KERNEL(k_f8, F8)
KERNEL(k_f8s8, F8S8)
KERNEL(k_f8s4, F8S4)
KERNEL(k_f8s2, F8S2)
KERNEL(k_f8_s8, F8_S8)
KERNEL(k_d8, D8)
KERNEL(k_d8x2, D8x2)
KERNEL(k_d8x4, D8x4)
KERNEL(k_d8s8, D8S8)
KERNEL(k_f8p4, F8P4)
KERNEL(k_f8v4, F8V4)
KERNEL(k_f8r4, F8R4)
KERNEL(k_f8n8, F8N8)
KERNEL(k_s8, S8)
KERNEL(k_v8, V8)

struct K { const char* name; void (*fn)(const double*, double*, int); int valu, other; };
#define CHK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r), __LINE__); exit(1); } } while (0)

int main() {
    K ks[] = {{"8 fma (indep)", k_f8, 8, 0}, {"8 fma + 8 s_mov interleaved", k_f8s8, 8, 8}, {"8 fma + 4 s_mov", k_f8s4, 8, 4},
              {"8 fma + 2 s_mov", k_f8s2, 8, 2}, {"8 fma then 8 s_mov", k_f8_s8, 8, 8}, {"8 fma one chain", k_d8, 8, 0},
              {"8 fma two chains", k_d8x2, 8, 0}, {"8 fma four chains", k_d8x4, 8, 0}, {"8 fma one chain + 8 s_mov", k_d8s8, 8, 8},
              {"8 fma + 4 v_mov_dpp", k_f8p4, 12, 0}, {"8 fma + 4 v_mov_b32", k_f8v4, 12, 0}, {"8 fma + 4 v_readlane", k_f8r4, 12, 0},
              {"8 fma + 8 s_nop", k_f8n8, 8, 8}, {"8 s_mov", k_s8, 0, 8}, {"8 v_mov_b32", k_v8, 8, 0}};
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
            printf("| %s | %d | %.2f | %.1f | %.2f |\n", k.name, block / 256, ns, cyc, cyc / (k.valu + k.other));
        }
    return 0;
}
