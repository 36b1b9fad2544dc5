// Accuracy of a lean fp64 square root (v_rsq_f64 + one Goldschmidt step + one or two residual corrections, no scaling
// for denormal / huge arguments) against the correctly rounded library sqrt, gfx950.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/microbench/sqrt_acc tools/microbench/sqrt_acc.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <int CORR> __device__ double lean_sqrt(double a) {
    const double r = __builtin_amdgcn_rsq(a);
    double g = a * r, h = 0.5 * r;
    const double e = fma(-h, g, 0.5);
    g = fma(g, e, g); h = fma(h, e, h);
    for (int c = 0; c < CORR; ++c) g = fma(fma(-g, g, a), h, g);
    return (a == 0.0) ? 0.0 : g;
}
__global__ void k(const double* x, double* y1, double* y2, double* y0, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { y1[i] = lean_sqrt<1>(x[i]); y2[i] = lean_sqrt<2>(x[i]); y0[i] = sqrt(x[i]); }
}
int main() {
    const int n = 1 << 22;
    std::vector<double> x(n), a(n), b(n), c(n);
    srand(7);
    for (int i = 0; i < n; ++i) x[i] = ldexp(1.0 + rand() / (double)RAND_MAX, (rand() % 600) - 300);
    x[0] = 0.0; x[1] = 1.0; x[2] = 4.0; x[3] = 2.0;
    double *dx, *d1, *d2, *d0;
    hipMalloc(&dx, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8); hipMalloc(&d0, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, d1, d2, d0, n);
    hipMemcpy(a.data(), d1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), d2, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), d0, n * 8, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0; long bad1 = 0, bad2 = 0;
    for (int i = 0; i < n; ++i) {
        const double ref = c[i];
        if (ref == 0.0) { if (a[i] != 0.0 || b[i] != 0.0) printf("zero wrong\n"); continue; }
        const double r1 = fabs(a[i] - ref) / ref, r2 = fabs(b[i] - ref) / ref;
        e1 = fmax(e1, r1); e2 = fmax(e2, r2); bad1 += a[i] != ref; bad2 += b[i] != ref;
    }
    printf("one correction: max rel err %.3e, %ld of %d differ from the library result\n", e1, bad1, n);
    printf("two corrections: max rel err %.3e, %ld of %d differ\n", e2, bad2, n);
    return 0;
}
