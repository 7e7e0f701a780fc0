// Diagnostic (not part of the product): accuracy of v_rcp_f64 and of one / two Newton steps on it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(double *out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // x spread over several binades, mantissas dense
    const double x = ldexp(1.0 + (double)i / n, (i % 7) * 9 - 27) * ((i & 1) ? 1.0 : 3.0);
    const double r0 = __builtin_amdgcn_rcp(x);
    const double r1 = fma(r0, fma(-x, r0, 1.0), r0);
    const double r2 = fma(r1, fma(-x, r1, 1.0), r1);
    const double e = fma(-x, r0, 1.0);
    const double rc = fma(r0, fma(e, e, e), r0);   // cubic
    const double ex = 1.0 / x;
    out[4 * i + 0] = fabs(r0 - ex) / ex; out[4 * i + 1] = fabs(r1 - ex) / ex; out[4 * i + 2] = fabs(r2 - ex) / ex; out[4 * i + 3] = fabs(rc - ex) / ex;
}
int main() {
    const int n = 1 << 22;
    double *d; hipMalloc(&d, sizeof(double) * 4 * n);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, n);
    double *h = new double[4 * n]; hipMemcpy(h, d, sizeof(double) * 4 * n, hipMemcpyDeviceToHost);
    double m[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; i++) for (int j = 0; j < 4; j++) m[j] = fmax(m[j], h[4 * i + j]);
    printf("max rel error: v_rcp_f64 %.3e (2^%.1f) | +1 Newton %.3e | +2 Newton %.3e | cubic %.3e   (eps = 2.22e-16)\n", m[0], log2(m[0]), m[1], m[2], m[3]);
    return 0;
}
