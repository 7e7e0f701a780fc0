/*
 * oracle/ba_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's sparse-LM bundle-adjustment path
 * (MultiCamMapper::solve -> ucoslam::SparseLevMarq<double>).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (automatic-ar_amd/) never links, imports or calls it.
 *
 * Parity pin: the reference has no tests / golden vectors for this path (SURVEY.md
 * section 4), and libs/multicam_mapper.cpp cannot be compiled here (OpenCV absent).
 * The solver half is pinned to the reference itself: oracle/ref_harness.cpp compiles the
 * reference's own libs/sparselevmarq.h + vendored Eigen in place (oracle/_ref/) and
 * tests/ check this restatement's LM loop, J^T J and LDL^T solve against it, both live
 * and through tests/golden/ fixtures.  The residual/Jacobian half restates OpenCV
 * arithmetic (cv::Rodrigues, cv::Mat::inv, the projection, cv::undistortPoints) from the
 * published definitions and is pinned against an INDEPENDENT implementation of those
 * definitions (scipy.spatial.transform.Rotation, numpy.linalg.inv, a numpy projection,
 * the Newton-inverted distortion model: tests/golden/make_primitives.py ->
 * tests/golden/g0_primitives.npz, tests/test_primitives_pin.py) -- not against OpenCV
 * itself, which is absent from the image and from the reference tree.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).
 */
#ifndef BA_ORACLE_H
#define BA_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One bundle-adjustment problem in the reference's own terms.
 * Observation order = residual row order of eval_curr_solution
 * (libs/multicam_mapper.cpp:1001-1007): ascending frame, ascending camera, detection order.
 * Indices are ranks of the ids in ascending std::map order (libs/multicam_mapper.h:95-105). */
typedef struct orc_problem {
    int32_t num_cams, num_markers, num_frames;
    int32_t root_cam, root_marker;       /* indices of the root camera / root marker          */
    const double *K;                     /* [num_cams][9] row-major 3x3 camera matrices        */
    double marker_size;                  /* MultiCamMapper::marker_size (a float widened)      */
    int64_t num_obs;                     /* marker observations; residual rows = 8*num_obs     */
    const int32_t *obs_frame, *obs_cam, *obs_marker;
    const float *obs_uv;                 /* [num_obs][8]: x0 y0 x1 y1 x2 y2 x3 y3 (undistorted) */
    int32_t opt_cams, opt_markers, opt_frames;   /* MultiCamMapper::Config (intrinsics off)    */
    int32_t with_huber;                  /* libs/multicam_mapper.cpp:1014-1019                 */
    float huber_delta;
    int32_t opt_intrinsics;              /* Config::optimize_cam_intrinsics: z ends with 9 per camera -- fx cx fy cy d0..d4
                                            (fill_io_vec_cam_intrinsics, :488-498); the projection then uses the pinhole matrix
                                            intrinsics_vec2mats rebuilds from z (:580-593: no skew), the five distortion entries
                                            never reach project_marker (their Jacobian columns are exact zeros)              */
    const double *dist;                  /* [num_cams][5] or NULL (zeros): only carried through z                            */
} orc_problem;

enum { ORC_RES_F32 = 0,   /* reference-faithful: projections rounded to float, float subtraction */
       ORC_RES_F64 = 1 }; /* same formula kept in double                                          */
enum { ORC_JAC_NUMERIC_F32 = 0, /* reference-faithful central differences, delta=1e-3, float projections */
       ORC_JAC_NUMERIC_F64 = 1, /* central differences, delta=1e-6, double projections                   */
       ORC_JAC_ANALYTIC    = 2, /* closed-form SE(3) Jacobian                     */
       ORC_JAC_TRACK       = 3 };/* SparseLevMarq::calcDerivates (libs/sparselevmarq.h:165-220): central differences,
                                    der_epsilon = 1e-3, double residuals, entries with |d| <= 1e-4 dropped (track()) */

typedef struct orc_lm_params {           /* ucoslam::SparseLevMarq<T>::Params, libs/sparselevmarq.h:30-50 */
    int32_t max_iters;
    double min_error, min_step_error_diff, min_average_step_error_diff, tau;
    int32_t huber_fixed;   /* 0: solve() semantics (delta = 10 then optCallBack's schedule); 1: track() semantics: the
                              problem's huber_delta is used as is for the whole solve (no step callback is installed) */
} orc_lm_params;

typedef struct orc_lm_iter {             /* one step() of libs/sparselevmarq.h:349-430 */
    double err;                          /* currErr after the step                      */
    double mu;                           /* damping after the step                      */
    double gain;
    double delta_norm;                   /* ||delta|| of the last trial                 */
    int32_t accepted;
    int32_t tries;
} orc_lm_iter;

/* length of the full default-Config pose vector: 6(C-1)+6(M-1)+6F (libs/multicam_mapper.cpp:239-250) */
int64_t orc_full_len(const orc_problem *p);
/* length of z for the problem's Config flags */
int64_t orc_num_vars(const orc_problem *p);
/* z <-> x_full helpers (optimised groups only, reference order cams|markers|frames|intrinsics); the intrinsics part of z comes
 * from p->K / p->dist on the way in and is read back with orc_get_intrinsics */
void orc_extract_z(const orc_problem *p, const double *x_full, double *z);
void orc_merge_z(const orc_problem *p, const double *x_full, const double *z, double *x_out);
void orc_get_intrinsics(const orc_problem *p, const double *z, double *K_out /* [C][9] */, double *dist_out /* [C][5] */);
/* triplet capacity orc_jacobian needs: 8 rows x (18 pose + 9 intrinsics columns) per observation */
int64_t orc_jac_capacity(const orc_problem *p);

/* cv::Rodrigues restatement (call sites libs/multicam_mapper.cpp:470,478) */
void orc_rodrigues_vec2mat(const double w[3], double R[9]);
void orc_rodrigues_mat2vec(const double R[9], double w[3]);

/* error_function / eval_curr_solution: libs/multicam_mapper.cpp:731-737,996-1028 */
void orc_residuals(const orc_problem *p, const double *x_full, const double *z, int res_mode, double *r);
/* jacobian_function: libs/multicam_mapper.cpp:739-801,803-994. Triplets (row, col, val); capacity orc_jac_capacity. */
int64_t orc_jacobian(const orc_problem *p, const double *x_full, const double *z, int jac_mode,
                     int32_t *rows, int32_t *cols, double *vals);
/* dense J^T J (P x P, row-major) and B = -J^T r (libs/sparselevmarq.h:355-367); small problems only */
void orc_normal_equations_dense(const orc_problem *p, const double *x_full, const double *z, int jac_mode,
                                int res_mode, double *JtJ, double *B);
/* (JtJ + mu I) delta = B through this file's own sparse LDL^T (restating :384-400) */
int orc_damped_solve(const orc_problem *p, const double *x_full, const double *z, int jac_mode, int res_mode,
                     double mu, double *delta);
/* SparseLevMarq<double>::solve(z,f,J): libs/sparselevmarq.h:440-472 (Appendix B). Returns final currErr. */
double orc_lm_solve(const orc_problem *p, const double *x_full, double *z_inout, const orc_lm_params *prm,
                    int jac_mode, int res_mode, orc_lm_iter *trace, int32_t trace_cap, int32_t *n_iters,
                    int32_t num_threads);
/* fp64 reprojection statistics from z: per-corner RMSE sqrt(sum r^2/(4N)) and mean euclidean corner distance */
void orc_reproj_stats(const orc_problem *p, const double *x_full, const double *z, double *rmse,
                      double *mean_dist, double *sum_sq);

/* MultiCamMapper::remove_distortions for one camera: cv::undistortPoints(src, dst, K, dist, noArray(), P = K)
 * (libs/multicam_mapper.cpp:570).  OpenCV is not in the reference tree (third-party, tested version 3.2.0, README.md:11): this
 * restates the published algorithm of that version's cvUndistortPoints -- normalise with K, five fixed-point iterations of
 * the inverse distortion model (k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4), re-project with P = K; fp64 inside, float in / out.
 * Pinned against the published model inverted by Newton (tests/test_primitives_pin.py), not against OpenCV itself.
 * in / out: [n][2]. */
void orc_undistort_points(const double K[9], const double *dist, int n_dist, int64_t n, const float *in, float *out);
/* The forward model (what cv::projectPoints applies to a normalised point): ideal pixel -> distorted pixel, in fp64.  Test
 * helper for the round trip distort(undistort(p)) = p. */
void orc_distort_points(const double K[9], const double *dist, int n_dist, int64_t n, const double *in, double *out);

#ifdef __cplusplus
}
#endif
#endif
