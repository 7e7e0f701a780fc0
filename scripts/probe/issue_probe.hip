// Diagnostic (not part of the product): issue cost of fp64 FMA / fp64 MFMA streams on gfx950, in s_memtime cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4v __attribute__((ext_vector_type(4)));
__global__ void k_fma_indep(unsigned long long *out, int iters) {
    double a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 1e-3 + i;
    const double m = 0.999 + threadIdx.x * 1e-9, c = 1e-6;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) a[j] = fma(a[j], m, c);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < 8; i++) s += a[i];
    if ((threadIdx.x & 63) == 0) { out[2 * (threadIdx.x >> 6)] = t1 - t0; out[2 * (threadIdx.x >> 6) + 1] = (unsigned long long)s; }
}
__global__ void k_fma_dep(unsigned long long *out, int iters) {
    double a = threadIdx.x * 1e-3;
    const double m = 0.999 + threadIdx.x * 1e-9, c = 1e-6;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) a = fma(a, m, c);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) { out[2 * (threadIdx.x >> 6)] = t1 - t0; out[2 * (threadIdx.x >> 6) + 1] = (unsigned long long)a; }
}
__global__ void k_fma_masked(unsigned long long *out, int iters, int active) {
    double a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 1e-3 + i;
    const double m = 0.999 + threadIdx.x * 1e-9, c = 1e-6;
    unsigned long long t0 = 0, t1 = 0;
    if ((int)(threadIdx.x & 63) < active) {
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int j = 0; j < 8; j++) a[j] = fma(a[j], m, c);
        }
        t1 = __builtin_amdgcn_s_memtime();
    }
    double s = 0; for (int i = 0; i < 8; i++) s += a[i];
    if ((threadIdx.x & 63) == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)s; }
}
template <int NACC>
__global__ void k_mfma(unsigned long long *out, int iters) {
    double4v acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = double4v{0, 0, 0, 0};
    const double a = threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-4;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < NACC; j++) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][3];
    if ((threadIdx.x & 63) == 0) { out[2 * (threadIdx.x >> 6)] = t1 - t0; out[2 * (threadIdx.x >> 6) + 1] = (unsigned long long)s; }
}
// sustained fp64 MFMA on the whole chip: shader clock (s_memtime ticks per 100 MHz s_memrealtime tick) and cycles per MFMA
__global__ void k_mfma_chip(unsigned long long *out, int iters) {
    double4v acc[9];
    for (int i = 0; i < 9; i++) acc[i] = double4v{0, 0, 0, 0};
    const double a = threadIdx.x * 1e-3 + blockIdx.x, b = 1.0 - threadIdx.x * 1e-4;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 9; j++) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0; for (int i = 0; i < 9; i++) s += acc[i][0] + acc[i][3];
    if (threadIdx.x == 0) { out[3 * blockIdx.x] = t1 - t0; out[3 * blockIdx.x + 1] = r1 - r0; out[3 * blockIdx.x + 2] = (unsigned long long)s; }
}
// layout check: C = A(16x4) * B(4x16) with integer data
__global__ void k_layout(double *out) {
    const int l = threadIdx.x;
    const double a = 1.0 + (l & 15) + 100.0 * (l >> 4);      // A[i][k] = 1 + i + 100k
    const double b = 1.0 + 2.0 * (l & 15) + 1000.0 * (l >> 4);  // B[k][j] = 1 + 2j + 1000k
    double4v acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; r++) out[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}
int main() {
    unsigned long long *d, h[64];
    hipMalloc(&d, sizeof h);
    const int it = 2000;
    for (int threads : {64, 256, 512, 1024}) {
        hipLaunchKernelGGL(k_fma_indep, dim3(1), dim3(threads), 0, 0, d, it);
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("indep DFMA  threads %4d: %.2f cycles per wave-instruction (wave 0)\n", threads, (double)h[0] / (8.0 * it));
    }
    for (int active : {64, 48, 32, 16, 8, 1}) {
        hipLaunchKernelGGL(k_fma_masked, dim3(1), dim3(64), 0, 0, d, it, active);
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("indep DFMA, %2d active lanes: %.2f cycles per wave-instruction\n", active, (double)h[0] / (8.0 * it));
    }
    hipLaunchKernelGGL(k_fma_dep, dim3(1), dim3(64), 0, 0, d, it);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("dependent DFMA chain: %.2f cycles per instruction\n", (double)h[0] / (8.0 * it));
    for (int threads : {64, 256, 512}) {
        hipLaunchKernelGGL(k_mfma<1>, dim3(1), dim3(threads), 0, 0, d, it);
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("MFMA f64 16x16x4 dependent (1 acc) threads %4d: %.2f cycles per instruction\n", threads, (double)h[0] / (1.0 * it));
        hipLaunchKernelGGL(k_mfma<4>, dim3(1), dim3(threads), 0, 0, d, it);
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("MFMA f64 16x16x4 independent (4 acc) threads %4d: %.2f cycles per instruction\n", threads, (double)h[0] / (4.0 * it));
    }
    {
        unsigned long long *d2, *h2 = new unsigned long long[3 * 1024];
        hipMalloc(&d2, sizeof(unsigned long long) * 3 * 1024);
        for (int blocks : {1, 256, 1024}) for (int threads : {256, 512}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            const int iters = 20000;
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_mfma_chip, dim3(blocks), dim3(threads), 0, 0, d2, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h2, d2, sizeof(unsigned long long) * 3 * blocks, hipMemcpyDeviceToHost);
            const double flop = 2.0 * 1024 * 9.0 * iters * (threads / 64) * blocks;
            printf("fp64 MFMA chip load: %4d blocks x %d threads: %.2f ms, %.1f TFLOP/s, block 0: clock %.0f MHz, %.1f cycles per MFMA (its wave 0)\n",
                   blocks, threads, ms, flop / ms / 1e9, 100.0 * h2[0] / h2[1], (double)h2[0] / (9.0 * iters));
        }
    }
    double *o, ho[256];
    hipMalloc(&o, sizeof ho);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, o);
    hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
        double ref = 0; for (int k = 0; k < 4; k++) ref += (1.0 + i + 100.0 * k) * (1.0 + 2.0 * j + 1000.0 * k);
        if (ref != ho[i * 16 + j]) bad++;
    }
    printf("f64 MFMA layout check (A[l&15][l>>4], B[l>>4][l&15], D row=(l>>4)+4r col=l&15): %d mismatches\n", bad);
    return 0;
}
