// Diagnostic (not part of the product): in-kernel shader clock = d(s_memtime) / d(s_memrealtime) * 100 MHz,
// and the cost of a barrier + LDS round trip, for a lightly loaded GPU (1 workgroup) and a loaded one.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_clock(unsigned long long *out, int iters) {
    __shared__ double buf[256];
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    double x = threadIdx.x * 1e-3 + 1.0;
    for (int i = 0; i < iters; i++) {
        buf[threadIdx.x] = x;
        __syncthreads();
        x = fma(x, 0.999, buf[(threadIdx.x + 17) & 255] * 1e-6);
        __syncthreads();
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[blockIdx.x * 3] = t1 - t0; out[blockIdx.x * 3 + 1] = r1 - r0; out[blockIdx.x * 3 + 2] = (unsigned long long)(x * 1000); }
}
__global__ void k_rcp(unsigned long long *out, int iters) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    double x = threadIdx.x * 1e-3 + 1.5;
    for (int i = 0; i < iters; i++) { double r = __builtin_amdgcn_rcp(x); r = fma(r, fma(-x, r, 1.0), r); r = fma(r, fma(-x, r, 1.0), r); x = r + 1.0; }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)(x * 1000); }
}
int main() {
    unsigned long long *d, h[3 * 1024];
    hipMalloc(&d, sizeof h);
    for (int blocks : {1, 1, 256, 1024}) {
        for (int rep = 0; rep < 3; rep++) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a);
            hipLaunchKernelGGL(k_clock, dim3(blocks), dim3(256), 0, 0, d, 20000);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            hipMemcpy(h, d, sizeof(unsigned long long) * 3, hipMemcpyDeviceToHost);
            printf("blocks %4d: %.3f ms  shader cycles %llu realtime ticks %llu -> %.0f MHz ; %.1f cycles per (2 barriers + LDS round trip + fma) iteration\n",
                   blocks, ms, h[0], h[1], 100.0 * h[0] / h[1], (double)h[0] / 20000);
        }
    }
    hipLaunchKernelGGL(k_rcp, dim3(1), dim3(64), 0, 0, d, 10000);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("rcp+2 newton dependent chain: %.1f cycles per iteration\n", (double)h[0] / 10000);
    return 0;
}
