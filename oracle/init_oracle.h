/*
 * oracle/init_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's Initializer (libs/initializer.cpp) and of the planar square pose solver it calls,
 * aruco::solvePnP_ (3rdparty/aruco/aruco/ippe.cpp:118-124 -> solvePoseOfCentredSquare :141-223).  Only tests/ may load it.
 *
 * PARITY UNPINNED against OpenCV / aruco themselves: both reference files need OpenCV (cv::Mat, cv::undistortPoints,
 * cv::Rodrigues), which is not in this image, so neither can be compiled here and the reference holds no golden vectors for
 * them.  IPPE is pinned against its two defining equations evaluated in independent numpy
 * (tests/test_initializer.py::test_oracle_ippe_satisfies_the_defining_equations_of_ippe); the rest of the restatement is
 * checked by properties: exact planar poses are recovered with ~zero
 * reprojection error, the two IPPE solutions are ordered by error, the vote picks the consistent candidate, the spanning
 * tree reaches every connected camera / marker, and the poses it hands to the LM converge to the ground truth.
 *
 * Container semantics are kept (std::map / std::set iteration order, first-minimum ties, operator[] creating empty frames);
 * matrices are 4x4 row-major doubles as cv::Mat(4,4,CV_64FC1).
 */
#ifndef INIT_ORACLE_H
#define INIT_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_cam_model {
    double K[9];      /* row-major camera matrix                           */
    double dist[12];  /* k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4 (unused = 0)  */
    int32_t n_dist;
} orc_cam_model;

/* cv::undistortPoints(src, dst, K, dist) WITHOUT R / P: normalised coordinates, float out (ippe.cpp:167). */
void orc_undistort_normalized(const orc_cam_model *cam, int64_t n, const float *in, float *out);

/* aruco::solvePnP_(size, corners, K, dist) for n markers of one camera.  T1/T2: [n][16] 4x4 poses (marker -> camera),
 * rounded to float as getRTMatrix(..., CV_32F) does (ippe.cpp:40-93,122); e1 <= e2: float reprojection errors. */
void orc_ippe_square(float marker_size, const orc_cam_model *cam, int64_t n, const float *uv, double *T1, double *e1,
                     double *T2, double *e2);

/* Initializer::find_best_transformation (libs/initializer.cpp:151-193) on one candidate set: returns the index of the
 * first minimum (-1 for an empty set), *weight = its summed corner distance; cost[n] (optional) = every candidate's sum. */
int64_t orc_vote(double marker_size, int64_t n, const double *T, const double *T1inv, const double *T2inv, double *cost,
                 double *weight);

/* 4x4 inverse as cv::Mat::inv() (LU with partial pivoting). */
void orc_inv4(const double *A, double *Ainv);

/* The whole Initializer(detections, marker_size, cam_configs, excluded_cams) constructor (libs/initializer.cpp:64-71):
 * obtain_pose_estimations + init_transforms.  Detections in file order (frame, camera slot, detection).  Returns a handle. */
void *orc_init_run(int32_t num_cam_slots, int32_t num_frames, int64_t n_det, const int32_t *det_frame,
                   const int32_t *det_cam, const int32_t *det_id, const float *det_uv, double marker_size,
                   const orc_cam_model *cams, const int32_t *excluded, int32_t n_excluded, double threshold,
                   int32_t min_detections);
/* apps/track.cpp:88-123 per frame: transforms_to_root_cam / _marker are GIVEN (set_transforms_to_root_*), then
 * obtain_pose_estimations + init_object_transforms.  Same handle as orc_init_run. */
void *orc_init_object_poses(int32_t num_cam_slots, int32_t num_frames, int64_t n_det, const int32_t *det_frame,
                            const int32_t *det_cam, const int32_t *det_id, const float *det_uv, double marker_size,
                            const orc_cam_model *cams, int32_t n_fixed_cams, const int32_t *cam_ids, const double *T_cam,
                            int32_t n_fixed_markers, const int32_t *marker_ids, const double *T_marker, double threshold,
                            int32_t min_detections);
/* counts[0..2] = cameras, markers, frames with a pose; counts[3] = frames kept in frame_cam_markers;
 * counts[4] = root camera id, counts[5] = root marker id */
void orc_init_counts(const void *h, int32_t counts[6]);
/* ids ascending; T_*: [count][16] transforms_to_root_cam / transforms_to_root_marker / object_transforms */
void orc_init_get(const void *h, int32_t *cam_ids, double *T_cam, int32_t *marker_ids, double *T_marker, int32_t *frame_ids,
                  double *T_object, int32_t *kept_frame_ids);
void orc_init_free(void *h);

#ifdef __cplusplus
}
#endif
#endif
