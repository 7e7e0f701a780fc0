// aar_dataset lifetime, error reporting, shard planning.  Host only.
#include <cstring>
#include <string>

#include "internal.h"
#include "se3.h"

namespace aar {

static thread_local std::string g_last_error;

int set_error(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

template <class T>
static T *zalloc(int64_t n) {
    return static_cast<T *>(calloc(n > 0 ? (size_t)n : 1, sizeof(T)));
}

aar_dataset *dataset_alloc(int C, int M, int F, int64_t N, bool with_truth) {
    aar_dataset *d = zalloc<aar_dataset>(1);
    d->num_cams = C; d->num_markers = M; d->num_frames = F; d->num_obs = N;
    d->cam_ids = zalloc<int32_t>(C);
    d->marker_ids = zalloc<int32_t>(M);
    d->frame_ids = zalloc<int32_t>(F);
    d->image_sizes = zalloc<int32_t>(2LL * C);
    d->cam_mats = zalloc<double>(9LL * C);
    d->dist_coeffs = zalloc<double>(5LL * C);
    d->obs_frame = zalloc<int32_t>(N);
    d->obs_cam = zalloc<int32_t>(N);
    d->obs_marker = zalloc<int32_t>(N);
    d->obs_uv = zalloc<float>(8 * N);
    const int64_t len = 6LL * (C - 1) + 6LL * (M - 1) + 6LL * F;
    d->x_full = zalloc<double>(len);
    d->x_truth = with_truth ? zalloc<double>(len) : nullptr;
    d->optimize_cam_poses = d->optimize_marker_poses = d->optimize_object_poses = 1;
    d->optimize_cam_intrinsics = 0;  // apps/find_solution.cpp:140
    return d;
}

}  // namespace aar

extern "C" {

const char *aar_last_error(void) { return aar::g_last_error.c_str(); }

void aar_dataset_free(aar_dataset *d) {
    if (!d) return;
    free(d->cam_ids); free(d->marker_ids); free(d->frame_ids); free(d->image_sizes);
    free(d->cam_mats); free(d->dist_coeffs);
    free(d->obs_frame); free(d->obs_cam); free(d->obs_marker); free(d->obs_uv);
    free(d->x_full); free(d->x_truth);
    free(d);
}

int64_t aar_dataset_full_len(const aar_dataset *d) {
    return 6LL * (d->num_cams - 1) + 6LL * (d->num_markers - 1) + 6LL * d->num_frames;
}

void aar_problem_desc_from_dataset(const aar_dataset *d, aar_problem_desc *p) {
    memset(p, 0, sizeof *p);
    p->num_cams = d->num_cams; p->num_markers = d->num_markers; p->num_frames = d->num_frames;
    p->root_cam = d->root_cam; p->root_marker = d->root_marker;
    p->cam_mats = d->cam_mats;
    p->marker_size = d->marker_size;
    p->num_obs = d->num_obs;
    p->obs_frame = d->obs_frame; p->obs_cam = d->obs_cam; p->obs_marker = d->obs_marker;
    p->obs_uv = d->obs_uv;
    p->optimize_cam_poses = d->optimize_cam_poses;
    p->optimize_marker_poses = d->optimize_marker_poses;
    p->optimize_object_poses = d->optimize_object_poses;
    p->residual_mode = AAR_RES_F32;
    p->with_huber = 0;
    p->device_id = 0;
    p->comm = nullptr;
}

void aar_rodrigues_vec2mat(const double w[3], double R[9]) { aar::rodrigues_vec2mat(w, R); }
void aar_rodrigues_mat2vec(const double R[9], double w[3]) { aar::rodrigues_mat2vec(R, w); }

// Contiguous frame ranges with (nearly) equal observation counts: rank r's range ends at the first
// frame where the running total reaches (r+1)/world of all observations.  Deterministic, so every rank
// computes the same plan from the same counts.
int aar_plan_shards(int32_t num_frames, const int64_t *obs_per_frame, int32_t world, int32_t *begin) {
    if (num_frames < 0 || world < 1 || !begin || (num_frames > 0 && !obs_per_frame))
        return aar::set_error(AAR_ERR_INVALID, "aar_plan_shards: bad arguments");
    int64_t total = 0;
    for (int f = 0; f < num_frames; f++) total += obs_per_frame[f];
    begin[0] = 0;
    int64_t run = 0;
    int f = 0;
    for (int r = 1; r < world; r++) {
        // smallest f with run >= total*r/world (ties resolved towards the earlier frame boundary)
        const double target = (double)total * (double)r / (double)world;
        while (f < num_frames && (double)run + 0.5 * (double)obs_per_frame[f] < target) run += obs_per_frame[f++];
        begin[r] = f;
    }
    begin[world] = num_frames;
    return AAR_OK;
}

}  // extern "C"
