// How fast does the MI355X hand out workgroups?  N workgroups of T threads with L bytes of LDS, each alive for ~C cycles.
//   hipcc --offload-arch=gfx950 -O3 scripts/probe/dispatch_probe.hip -o /tmp/dispatch_probe && /tmp/dispatch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_spin(int cycles, int touch, double *sink) {
    extern __shared__ double lds[];
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (touch) lds[threadIdx.x] = 1.0;
    while ((long long)(__builtin_readcyclecounter() - t0) < cycles) __builtin_amdgcn_s_sleep(8);
    if (cycles < 0) sink[blockIdx.x] = lds[0];
}

int main() {
    double *sink;
    hipMalloc(&sink, 1 << 20);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_spin), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int Ns[] = {256, 1024, 2048, 4096, 8192};
    const int Ts[] = {64, 256};
    const int Ls[] = {0, 16 * 1024};
    const int Cs[] = {0, 5000, 20000};
    printf("%8s %6s %8s %8s %10s %14s\n", "WGs", "thr", "LDS", "cycles", "us/launch", "ns per WG");
    for (int T : Ts) for (int L : Ls) for (int C : Cs) for (int N : Ns) {
        for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k_spin, dim3(N), dim3(T), L, 0, C, L > 0, sink);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_spin, dim3(N), dim3(T), L, 0, C, L > 0, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%8d %6d %8d %8d %10.2f %14.1f\n", N, T, L, C, 1e3 * ms / 20, 1e6 * ms / 20 / N);
    }
    return 0;
}
