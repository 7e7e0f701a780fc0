// Internal interface between the Initializer's host logic (host/initializer.cpp) and its kernels (csrc/init_kernels.hip).
// Not part of the C ABI.  All matrices are the 3x4 top of the reference's 4x4 cv::Mat (row-major, 12 doubles).
#pragma once
#include <cstdint>

#include "internal.h"

namespace aar {

struct InitDevice;  // device buffers of one Initializer run

int initdev_create(int32_t device_id, InitDevice **out);
void initdev_destroy(InitDevice *);

// aruco::solvePnP_ for every detection (3rdparty/aruco/aruco/ippe.cpp:118-223): the two poses of detection d stay on the
// device as pose 2d (smaller error) and 2d+1; the float errors and the corners undistorted with P = K
// (MultiCamMapper::remove_distortions, libs/multicam_mapper.cpp:554-578) come back to the host.
int initdev_ippe(InitDevice *, const aar_cam_model *cams, int32_t n_cams, float marker_size, int64_t n_det, const float *uv,
                 const int32_t *det_cam, float *e1, float *e2, float *uv_undistorted);

// Candidate sets between cameras (type 0) or markers (type 1), Initializer::fill_transformation_sets
// (libs/initializer.cpp:95-125): candidate k combines poses a[k] (first object) and b[k] (second object); sets are the
// ranges [set_begin[s], set_begin[s+1]).  Votes every set (find_best_transformation, :151-193) and returns per set the
// first-minimum candidate, its summed error and its transform.
int initdev_pair_vote(InitDevice *, int type, int64_t n_cand, const int32_t *a, const int32_t *b, int64_t n_sets,
                      const int64_t *set_begin, double marker_size, int64_t *best, double *weight, double *best_T);

// Initializer::fill_transformation_set + init_object_transforms (libs/initializer.cpp:73-93,453-465): candidate k is pose
// cand_pose[k] seen by camera index cand_cam[k] of marker index cand_marker[k]; to_root_* hold [count][12] transforms (identity
// for the roots); one set per frame.
int initdev_object_vote(InitDevice *, int64_t n_cand, const int32_t *cand_pose, const int32_t *cand_cam,
                        const int32_t *cand_marker, int32_t n_cams, const double *to_root_cam, int32_t n_markers,
                        const double *to_root_marker, int64_t n_sets, const int64_t *set_begin, double marker_size,
                        int64_t *best, double *weight, double *best_T);

}  // namespace aar
