// Internal helpers shared by the host translation units of libaar (not part of the C ABI).
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/aar.h"

namespace aar {

int set_error(int code, const char *fmt, ...);  // stores the thread's message, returns code

// Index bookkeeping of the pose vectors (fill_io_vec_* / *_vec2mats, libs/multicam_mapper.cpp:500-552)
struct PoseLayout {
    int C = 0, M = 0, F = 0, rc = 0, rm = 0;
    bool oc = true, om = true, of = true;
    bool oi = false;   // optimize_cam_intrinsics: the vectors end with 9 per camera (fx cx fy cy d0..d4), root camera included
    int64_t full_cam0() const { return 0; }
    int64_t full_mk0() const { return 6LL * (C - 1); }
    int64_t full_fr0() const { return 6LL * (C - 1) + 6LL * (M - 1); }
    int64_t full_intr0() const { return full_fr0() + 6LL * F; }
    int64_t full_len() const { return full_fr0() + 6LL * F + (oi ? 9LL * C : 0); }
    int64_t z_len() const { return (oc ? 6LL * (C - 1) : 0) + (om ? 6LL * (M - 1) : 0) + (of ? 6LL * F : 0) + (oi ? 9LL * C : 0); }
    int64_t z_intr0() const { return oi ? z_len() - 9LL * C : -1; }
    int64_t z_cam0() const { return oc ? 0 : -1; }
    int64_t z_mk0() const { return om ? (oc ? 6LL * (C - 1) : 0) : -1; }
    int64_t z_fr0() const { return of ? (oc ? 6LL * (C - 1) : 0) + (om ? 6LL * (M - 1) : 0) : -1; }
    int cam_slot(int c) const { return c == rc ? -1 : (c < rc ? c : c - 1); }
    int mk_slot(int m) const { return m == rm ? -1 : (m < rm ? m : m - 1); }
};

aar_dataset *dataset_alloc(int C, int M, int F, int64_t N, bool with_truth);

}  // namespace aar
