// LDS fp64 atomic throughput on MI355X: cycles per wave-instruction of ds_add_f64 (64 lanes, distinct addresses) against plain ds_write_b64 /
// ds_read_b64 + add + write, for 1 .. 8 wavefronts on a CU and a few address patterns.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics scripts/probe/lds_atomic_probe.hip -o scripts/probe/lds_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void k_probe(int iters, int stride, int rowstep, unsigned long long *out, double *sink) {
    __shared__ double lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *base = lds + wave * 1024;
    double acc = 0.0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        double *p = base + ((lane * stride + it * rowstep) & 1023);
        if (MODE == 0) atomicAdd(p, 1.0);                       // ds_add_f64
        else if (MODE == 1) *p = (double)it;                   // ds_write_b64
        else if (MODE == 2) { const double v = *p; *p = v + 1.0; }   // read, add, write (same lane owns the address: no atomicity needed)
        else if (MODE == 3) acc += *p;                         // ds_read_b64
    }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
    if (acc == 12345.678) sink[0] = acc;
}

int main() {
    unsigned long long *out; double *sink;
    hipMalloc(&out, 1 << 16); hipMalloc(&sink, 64);
    const char *names[] = {"ds_add_f64", "ds_write_b64", "read+add+write", "ds_read_b64"};
    const int iters = 2000;
    printf("%-16s %6s %7s %8s %22s\n", "op", "waves", "stride", "rowstep", "cycles / wave-instr");
    for (int mode = 0; mode < 4; mode++)
        for (int waves : {1, 4, 8})
            for (int stride : {1, 37, 21, 36})
                for (int rowstep : {64, 0}) {
                    unsigned long long h[16] = {0};
                    for (int rep = 0; rep < 2; rep++) {
                        if (mode == 0) hipLaunchKernelGGL(k_probe<0>, dim3(1), dim3(64 * waves), 0, 0, iters, stride, rowstep, out, sink);
                        if (mode == 1) hipLaunchKernelGGL(k_probe<1>, dim3(1), dim3(64 * waves), 0, 0, iters, stride, rowstep, out, sink);
                        if (mode == 2) hipLaunchKernelGGL(k_probe<2>, dim3(1), dim3(64 * waves), 0, 0, iters, stride, rowstep, out, sink);
                        if (mode == 3) hipLaunchKernelGGL(k_probe<3>, dim3(1), dim3(64 * waves), 0, 0, iters, stride, rowstep, out, sink);
                        hipDeviceSynchronize();
                    }
                    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
                    unsigned long long mx = 0;
                    for (int w = 0; w < waves; w++) mx = h[w] > mx ? h[w] : mx;
                    // all waves issue `iters` instructions each: LDS-pipe cycles per instruction = longest wave / (iters * waves)
                    printf("%-16s %6d %7d %8d %12.1f per wave, %6.1f per instr of the CU\n", names[mode], waves, stride, rowstep, (double)mx / iters, (double)mx / iters / waves);
                }
    return 0;
}
