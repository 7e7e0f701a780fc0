// MultiCamMapper::remove_distortions (libs/multicam_mapper.cpp:554-578) for the corners of one camera:
// cv::undistortPoints(points, out, K, dist, noArray(), P = K).  One thread per corner; fp64 inside, float in / out.
// The algorithm is OpenCV 3.2's cvUndistortPoints (the reference's tested version, README.md:11), restated from its published
// definition -- OpenCV itself is not part of the reference tree: normalise with K, five fixed-point iterations of the inverse
// of the (rational + tangential + thin-prism) distortion model, re-project with P = K.
#include <hip/hip_runtime.h>

#include "../host/internal.h"
#include "hostcopy.h"

namespace aar {

struct UndistortArgs {
    double K[9];
    double k[AAR_MAX_DIST];   // k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4
    long long n;
    const float *in;
    float *out;
};

__global__ void __launch_bounds__(256) k_undistort(const UndistortArgs a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const double fx = a.K[0], fy = a.K[4], cx = a.K[2], cy = a.K[5];
    const double ifx = 1.0 / fx, ify = 1.0 / fy;
    const double *k = a.k;
    double x = ((double)a.in[2 * i] - cx) * ifx, y = ((double)a.in[2 * i + 1] - cy) * ify;
    const double x0 = x, y0 = y;
#pragma unroll 1
    for (int it = 0; it < 5; it++) {
        const double r2 = x * x + y * y;
        const double icdist = (1.0 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1.0 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        const double dx = 2.0 * k[2] * x * y + k[3] * (r2 + 2.0 * x * x) + k[8] * r2 + k[9] * r2 * r2;
        const double dy = k[2] * (r2 + 2.0 * y * y) + 2.0 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
        x = (x0 - dx) * icdist;
        y = (y0 - dy) * icdist;
    }
    // P = K (R = identity): [xx yy ww] = K [x y 1]
    const double xx = a.K[0] * x + a.K[1] * y + a.K[2];
    const double yy = a.K[3] * x + a.K[4] * y + a.K[5];
    const double ww = 1.0 / (a.K[6] * x + a.K[7] * y + a.K[8]);
    a.out[2 * i] = (float)(xx * ww);
    a.out[2 * i + 1] = (float)(yy * ww);
}

}  // namespace aar

extern "C" int aar_undistort_points(const double K[9], const double *dist, int32_t n_dist, int64_t n_points, const float *uv_in,
                                    float *uv_out, int32_t device_id) {
    using namespace aar;
    if (!K || (n_dist > 0 && !dist) || n_dist < 0 || n_dist > AAR_MAX_DIST || n_points < 0 || (n_points > 0 && (!uv_in || !uv_out)))
        return set_error(AAR_ERR_INVALID, "aar_undistort_points: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return set_error(AAR_ERR_NO_DEVICE, "no HIP device available; this library has no CPU path");
    if (device_id < 0 || device_id >= ndev) return set_error(AAR_ERR_INVALID, "device_id %d out of range (%d devices)", device_id, ndev);
    if (n_points == 0) return AAR_OK;
    if (hipSetDevice(device_id) != hipSuccess) return set_error(AAR_ERR_HIP, "hipSetDevice failed");
    UndistortArgs a;
    for (int i = 0; i < 9; i++) a.K[i] = K[i];
    for (int i = 0; i < AAR_MAX_DIST; i++) a.k[i] = i < n_dist ? dist[i] : 0.0;
    a.n = n_points;
    float *d_in = nullptr, *d_out = nullptr;
    const size_t bytes = sizeof(float) * 2 * (size_t)n_points;
    hipError_t e = hipMalloc((void **)&d_in, bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&d_out, bytes);
    if (e == hipSuccess) e = (hipError_t)h2d(d_in, uv_in, bytes, nullptr);   // (page-locked staging: hostcopy.h)
    if (e == hipSuccess) {
        a.in = d_in; a.out = d_out;
        hipLaunchKernelGGL(k_undistort, dim3((unsigned)((n_points + 255) / 256)), dim3(256), 0, 0, a);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = (hipError_t)d2h(uv_out, d_out, bytes, nullptr);
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess) return set_error(AAR_ERR_HIP, "aar_undistort_points: %s", hipGetErrorString(e));
    return AAR_OK;
}
