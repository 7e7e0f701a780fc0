// Probe: how fast do fire-and-forget fp64 atomic adds to the SAME addresses drain on MI355X (agent scope, -munsafe-fp-atomics)?
// Decides whether a frame-major Schur complement (every frame adds straight into S) can work: the camera x camera blocks of S are hit by EVERY frame.
//   hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics scripts/probe/atomic_probe.hip -o /tmp/atomic_probe && /tmp/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// each wavefront adds `nblk` 6x6 blocks (lanes 0..35: one entry each; 6 lanes share a 48-byte row segment); block b of wavefront w lands on
// block ((w * spread_w + b) % distinct) of a row panel of `distinct` blocks
__global__ void k_add(double *S, int n_pad, int nblk, int distinct, int spread_w) {
    const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (lane >= 36) return;
    const int i = lane / 6, j = lane % 6;
    for (int b = 0; b < nblk; b++) {
        const int blk = (int)(((long long)w * spread_w + b) % distinct);
        const int br = blk / (n_pad / 6), bc = blk % (n_pad / 6);
        atomicAdd(S + (size_t)(6 * br + i) * n_pad + 6 * bc + j, 1.0);
    }
}

int main() {
    const int n_pad = 288;
    double *S;
    hipMalloc(&S, sizeof(double) * n_pad * n_pad);
    hipMemset(S, 0, sizeof(double) * n_pad * n_pad);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    struct Case { int waves, nblk, distinct, spread; const char *what; };
    const Case cases[] = {
        {500, 36, 36, 0, "500 wavefronts x the same 36 blocks (config 3 camera x camera, depth 500)"},
        {2000, 36, 36, 0, "2000 wavefronts x the same 36 blocks (config 4, depth 2000)"},
        {500, 36, 2304, 36, "500 wavefronts x 36 blocks, spread over 2304 (depth ~8)"},
        {2000, 36, 2304, 36, "2000 wavefronts x 36 blocks, spread over 2304 (depth ~31)"},
        {500, 66, 2304, 7, "500 x 66 blocks spread (depth ~14)"},
        {125, 36, 36, 0, "125 wavefronts x the same 36 blocks (depth 125)"},
        {32, 36, 36, 0, "32 wavefronts x the same 36 blocks (depth 32)"},
        {500, 1, 1, 0, "500 wavefronts x ONE block (depth 500)"},
        {2000, 1, 1, 0, "2000 wavefronts x ONE block (depth 2000)"},
        {500, 36, 36, 1, "500 wavefronts x the same 36 blocks, every wavefront starting one block further (depth 500)"},
        {250, 36, 36, 1, "250 ... rotated (depth 250)"},
        {125, 36, 36, 1, "125 ... rotated (depth 125)"},
        {125, 36, 36, 7, "125 ... rotated by 7 (depth 125)"},
        {63, 36, 36, 1, "63 ... rotated (depth 63)"},
        {32, 36, 36, 1, "32 ... rotated (depth 32)"},
        {125, 120, 120, 1, "125 wavefronts x the same 120 blocks rotated (depth 125)"},
        {125, 120, 120, 0, "125 wavefronts x the same 120 blocks in step (depth 125)"},
        {500, 0, 1, 0, "empty (launch cost)"},
    };
    for (const Case &c : cases) {
        for (int wg : {64, 256}) {
            const int per = wg / 64, grid = (c.waves + per - 1) / per;
            for (int k = 0; k < 5; k++) hipLaunchKernelGGL(k_add, dim3(grid), dim3(wg), 0, 0, S, n_pad, c.nblk, c.distinct, c.spread);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            const int reps = 50;
            for (int k = 0; k < reps; k++) hipLaunchKernelGGL(k_add, dim3(grid), dim3(wg), 0, 0, S, n_pad, c.nblk, c.distinct, c.spread);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%-80s wg %3d: %7.2f us per launch\n", c.what, wg, 1e3 * ms / reps);
        }
    }
    return 0;
}
