/*
 * include/aar.h -- C ABI of the MI355X-native sparse-LM bundle-adjustment path.
 *
 * This is the drop-in boundary for ONE path of HSarham/automatic-ar:
 *   MultiCamMapper::solve()  (libs/multicam_mapper.cpp:419-428)
 *     -> ucoslam::SparseLevMarq<double>::solve / init / step  (libs/sparselevmarq.h:440-472,238-249,349-430)
 *     -> MultiCamMapper::error_function / jacobian_function   (libs/multicam_mapper.cpp:731-801)
 * The reference has no FFI of its own (it is one C++ process); the entry points below are what a
 * binding of that path would need, and automatic-ar_amd/host/multicam_mapper.{h,cpp} is the C++ class
 * with the reference's method names that calls them.
 *
 * Conventions: plain pointers and sizes, caller-owned host arrays, no exceptions.  Every function that
 * returns int returns AAR_OK (0) or a negative aar_status; aar_last_error() gives the message of the
 * calling thread's last failure.  All compute entry points run hand-written HIP kernels on gfx950 and
 * FAIL (AAR_ERR_NO_DEVICE) when no GPU is present -- there is no CPU fallback.
 *
 * Pose vectors.  `x_full` is always the reference's default-Config pose vector
 *   [ (C-1) cameras | (M-1) markers | F frames ] x (rx,ry,rz,tx,ty,tz)
 * (fill_io_vec_cams/markers/object_poses, libs/multicam_mapper.cpp:500-522: ascending index, root camera
 * and root marker skipped) i.e. the pose part of the `.solution` vector (:1085-1089).  Which groups are
 * optimised is a flag (MultiCamMapper::Config, libs/multicam_mapper.h:75-81); fixed groups keep their
 * values.  With optimize_cam_intrinsics (off in find_solution, apps/find_solution.cpp:140; on in the reference's default
 * Config) the vectors carry the 9-per-camera intrinsics block behind the poses: see aar_problem_desc.
 */
#ifndef AAR_H
#define AAR_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum aar_status {
    AAR_OK = 0,
    AAR_ERR_INVALID = -1,      /* bad argument / malformed problem                        */
    AAR_ERR_NO_DEVICE = -2,    /* no HIP device: the product has no CPU path              */
    AAR_ERR_HIP = -3,          /* HIP runtime error                                        */
    AAR_ERR_UNSUPPORTED = -4,  /* outside the implemented scope            */
    AAR_ERR_NUMERIC = -5,      /* non-positive pivot in a Cholesky factorisation           */
    AAR_ERR_IO = -6,           /* file could not be opened / short read                    */
    AAR_ERR_COMM = -7          /* RCCL failure                                             */
} aar_status;

const char *aar_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Data set = everything MultiCamMapper holds after init(): ids, intrinsics, undistorted detections
 * and the pose vector.  Mirrors the content of a `.solution` file (libs/multicam_mapper.cpp:1053-1099).
 * Arrays are owned by the library (free with aar_dataset_free).
 * ------------------------------------------------------------------------------------------- */
typedef struct aar_dataset {
    int32_t num_cams, num_markers, num_frames;
    int32_t root_cam, root_marker;   /* INDICES (rank of the root id in ascending id order)          */
    int32_t *cam_ids;                /* [C] ascending                                                 */
    int32_t *marker_ids;             /* [M] ascending                                                 */
    int32_t *frame_ids;              /* [F] ascending                                                 */
    int32_t *image_sizes;            /* [C][2] width,height                                           */
    double *cam_mats;                /* [C][9] row-major K                                            */
    double *dist_coeffs;             /* [C][5] (carried for the file format; unused by the projection) */
    double marker_size;              /* MultiCamMapper::marker_size: (double)(float)size               */
    int64_t num_obs;                 /* marker observations in reference residual order                */
    int32_t *obs_frame, *obs_cam, *obs_marker; /* [N] indices                                          */
    float *obs_uv;                   /* [N][8] undistorted corners x0 y0 .. x3 y3                       */
    double *x_full;                  /* [6(C-1)+6(M-1)+6F] current pose vector                          */
    double *x_truth;                 /* same layout, ground truth (synthetic data only, else NULL)      */
    int32_t optimize_cam_poses, optimize_marker_poses, optimize_object_poses, optimize_cam_intrinsics;
} aar_dataset;

void aar_dataset_free(aar_dataset *);
int64_t aar_dataset_full_len(const aar_dataset *);   /* 6(C-1)+6(M-1)+6F */

/* Deterministic synthetic multi-camera / multi-marker sequence. */
typedef struct aar_synth_desc {
    int32_t num_cams, num_markers, num_frames;
    uint64_t seed;              /* 20190219 + config index                                             */
    double marker_size;         /* metres (0.05)                                                       */
    double noise_px;            /* corner noise sigma (0.3)                                            */
    double init_rot_sigma;      /* rad, perturbation of every rotation-vector component (0.02)         */
    double init_trans_sigma;    /* m, perturbation of every translation component (0.01)               */
    double init_scale;          /* multiplies both sigmas (1.0)                                        */
    double cam_arc_deg;         /* 0: cameras on a full ring around the scene; > 0: side by side on an arc of that
                                   many degrees centred on camera 0 (config 1, the box-like case: 50)   */
    double min_view_cos;        /* a marker is seen when cos(normal, view ray) >= this (0 = default 0.8;
                                   config 1: 0.35)                                                      */
} aar_synth_desc;
void aar_synth_default(aar_synth_desc *, int32_t config_index); /* BASELINE.json configs[1..4] -> 2..5; 1 = a 3-camera /
                                                                   6-marker stand-in for configs[0] (the box data set) */
int aar_synth_generate(const aar_synth_desc *, aar_dataset **out);

/* File formats of the path.
 *  .solution          libs/multicam_mapper.cpp:1053-1099 (write) / :1124-1205 (read)
 *  .solution.yaml     libs/multicam_mapper.cpp:1233-1268  (cv::FileStorage YAML 1.0 dialect)
 *  aruco.detections   libs/multicam_mapper.cpp:216-237, libs/initializer.cpp:316-348               */
int aar_solution_read(const char *path, aar_dataset **out);
/* flags: AAR_SOLUTION_REFERENCE_INDEXING files the detections as the reference's deserialize_frame_cam_markers does
 * (libs/multicam_mapper.cpp:1101-1122): the f-th frame record under frame id f and the c-th camera record of a frame under
 * camera id c -- the LOOP COUNTERS, not the ids the file stores (:1117-1119).  Same result as the default (ids honoured) when
 * frame / camera ids are 0..n-1 and every frame lists every camera; on `-subseqs` or `-exclude-cams` solutions it is what the
 * reference's track / overlay would see.  A frame counter that is not a stored frame id is AAR_ERR_INVALID. */
#define AAR_SOLUTION_REFERENCE_INDEXING 1
int aar_solution_read_ex(const char *path, int32_t flags, aar_dataset **out);
int aar_solution_write(const char *path, const aar_dataset *);
int aar_solution_write_yaml(const char *path, const aar_dataset *);
int aar_detections_write(const char *path, const aar_dataset *);

/* Camera calibration file <dir>/<cam>/calib.{xml,yml,yaml}: the four keys CamConfig::read_from_file takes from a
 * cv::FileStorage (libs/cam_config.cpp:52-80): image_width, image_height, camera_matrix (3x3), distortion_coefficients
 * (up to AAR_MAX_DIST values in OpenCV's order k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4; the rest of dist is zero-filled).
 * XML and YAML 1.0 dialects of cv::FileStorage. */
#define AAR_MAX_DIST 12
int aar_cam_config_read(const char *path, double K[9], double dist[AAR_MAX_DIST], int32_t *n_dist, int32_t *width, int32_t *height);

/* MultiCamMapper::remove_distortions for the corners of ONE camera (libs/multicam_mapper.cpp:554-578):
 * cv::undistortPoints(points, out, K, dist, noArray(), P = K) -- the fixed-point inversion of the distortion model (five
 * iterations, as OpenCV 3.2 does for a coefficient vector) followed by re-projection with K; float in, float out, fp64
 * inside.  Runs on the device (uv_in / uv_out are host pointers, [n_points][2]); uv_out may alias uv_in. */
int aar_undistort_points(const double K[9], const double *dist, int32_t n_dist, int64_t n_points, const float *uv_in,
                         float *uv_out, int32_t device_id);

/* ---------------------------------------------------------------------------------------------
 * Initializer (libs/initializer.cpp): raw detections + calibrations -> the initial solution MultiCamMapper(Initializer&)
 * holds before solve() (apps/find_solution.cpp:118-147).  The pose solver, the candidate sets and the n^2 votes run on the
 * device; the spanning trees (a handful of nodes) on the host.
 * ------------------------------------------------------------------------------------------- */
typedef struct aar_cam_model {       /* CamConfig (libs/cam_config.h)                                          */
    double K[9];                     /* camera_matrix, row-major                                                */
    double dist[AAR_MAX_DIST];       /* distortion_coefficients, OpenCV order, zero-filled                      */
    int32_t n_dist;
    int32_t width, height;
} aar_cam_model;
/* CamConfig::read_cam_configs (libs/cam_config.cpp:80-95): <folder>/<dir>/calib.{xml,yml,yaml} for every sub-directory;
 * camera index = position of the directory in ASCENDING NAME order (the reference takes readdir order).  Free with free(). */
int aar_cam_configs_read(const char *folder, aar_cam_model **out, int32_t *n_cams);
/* dir_order: AAR_DIR_ORDER_NAME (ascending name, "." / ".." skipped: the default above) or AAR_DIR_ORDER_READDIR -- the order
 * readdir lists the sub-directories in, "." and ".." included, exactly as the reference's get_dirs_list (libs/filesystem.cpp:5-17). */
enum { AAR_DIR_ORDER_NAME = 0, AAR_DIR_ORDER_READDIR = 1 };
int aar_cam_configs_read_ex(const char *folder, int32_t dir_order, aar_cam_model **out, int32_t *n_cams);

typedef struct aar_detections {      /* content of an `aruco.detections` file                                   */
    int32_t num_cams;                /* camera slots per frame record                                           */
    int32_t num_frames;              /* frame records                                                           */
    int64_t num_det;
    int32_t *det_frame, *det_cam, *det_id;   /* [num_det], file order: frame, camera slot, detection            */
    float *det_uv;                   /* [num_det][8] RAW (distorted) corners                                    */
} aar_detections;
/* Initializer::read_detections_file (libs/initializer.cpp:316-362).  subseqs = first,last frame pairs
 * (MultiCamMapper::read_subseqs); frames before/between the sub-sequences are emptied, as the reference does. */
int aar_detections_read(const char *path, const int32_t *subseqs, int32_t n_subseqs, aar_detections **out);
void aar_detections_free(aar_detections *);
/* <folder>/subseqs.txt: whitespace-separated frame numbers (libs/multicam_mapper.cpp read_subseqs).  Free with free(). */
int aar_subseqs_read(const char *path, int32_t **out, int32_t *n);

/* aruco::solvePnP_(size, corners, K, dist) (3rdparty/aruco/aruco/ippe.cpp:118-223) for n markers of one camera, on the
 * device: T1 / T2 [n][16] = the two 4x4 marker->camera poses rounded to float as getRTMatrix(.., CV_32F) does, err1 <= err2
 * the float reprojection errors in normalised image units.  uv: [n][8] raw corners (host). */
int aar_ippe_square(double marker_size, const aar_cam_model *cam, int64_t n, const float *uv, double *T1, double *err1,
                    double *T2, double *err2, int32_t device_id);

/* Initializer::find_best_transformation (libs/initializer.cpp:151-193) for n_sets candidate sets at once, on the device.
 * Candidates of set s are [set_begin[s], set_begin[s+1]) of T / T1inv / T2inv ([n][16] row-major 4x4, host).
 * best[s] = index inside the set of the first minimum (-1: empty set), weight[s] = its summed corner distance,
 * cost (optional, [n]) = every candidate's sum. */
int aar_vote_transforms(double marker_size, int64_t n_sets, const int64_t *set_begin, const double *T, const double *T1inv,
                        const double *T2inv, int64_t *best, double *weight, double *cost, int32_t device_id);

typedef struct aar_init_params {
    double marker_size;              /* metres                                                                  */
    double threshold;                /* second IPPE pose kept when err2/err1 < threshold (2.0, initializer.h:52)  */
    int32_t min_detections;          /* frames with fewer detections are dropped (2, initializer.h:51)           */
    int32_t n_excluded;
    const int32_t *excluded_cams;    /* camera slots to ignore (-exclude-cams)                                   */
    int32_t device_id;
} aar_init_params;
void aar_init_default_params(aar_init_params *);
/* Initializer(detections, marker_size, cam_configs, excluded_cams) followed by MultiCamMapper(Initializer&)
 * (libs/initializer.cpp:64-71, libs/multicam_mapper.cpp:252-254,281-331): the data set the reference writes as
 * initial.solution -- ids, intrinsics, corners undistorted with P = K, and the initial pose vector.  cams[i] belongs to
 * camera slot i.  Fails (AAR_ERR_INVALID) when a camera or marker is not connected to the root by co-visibility, where
 * the reference runs into std::map::at. */
int aar_initializer_run(const aar_detections *, const aar_cam_model *cams, int32_t n_cams, const aar_init_params *,
                        aar_dataset **out);
/* The Initializer as apps/track.cpp uses it (:70-89,117-120), for a whole recording at once: the camera and marker
 * transforms of `solution` are GIVEN (set_transforms_to_root_cam / _marker), only obtain_pose_estimations and
 * init_object_transforms run, and the result is what MultiCamMapper::init(object_poses, fcm) holds (libs/multicam_mapper.cpp:
 * 272-279): the solution's cameras / markers / intrinsics, the frames with >= min_detections usable detections, their
 * initial object poses and undistorted corners, optimize flags (0,0,1).  Follow with aar_problem_create + aar_track.
 * Detections of cameras or markers the solution does not hold are dropped. */
int aar_initializer_object_poses(const aar_dataset *solution, const aar_detections *, const aar_cam_model *cams, int32_t n_cams,
                                 const aar_init_params *, aar_dataset **out);

/* cv::Rodrigues as used at libs/multicam_mapper.cpp:470,478 (R row-major 3x3) */
void aar_rodrigues_vec2mat(const double w[3], double R[9]);
void aar_rodrigues_mat2vec(const double R[9], double w[3]);

/* Frame-range partition for `world` ranks, balanced by observation count:
 * rank r owns frames [begin[r], begin[r+1]).  begin has world+1 entries. */
int aar_plan_shards(int32_t num_frames, const int64_t *obs_per_frame, int32_t world, int32_t *begin);

/* ---------------------------------------------------------------------------------------------
 * RCCL communicator (multi-GPU only).  Rank 0 makes an id, the launcher hands it to every rank
 * (bench.py: torch.distributed broadcast), each rank creates its communicator on its own GPU.
 * ------------------------------------------------------------------------------------------- */
typedef struct aar_comm aar_comm;
#define AAR_COMM_ID_BYTES 128
int aar_comm_make_id(char id[AAR_COMM_ID_BYTES]);
int aar_comm_create(const char id[AAR_COMM_ID_BYTES], int32_t world_size, int32_t rank, int32_t device_id,
                    aar_comm **out);
void aar_comm_destroy(aar_comm *);
/* What the communicator has moved so far (bench.py reports it): ranks_seen = ncclCommCount of the RCCL communicator,
 * system_allreduce_bytes = payload of ONE all-reduce of the reduced system (packed lower triangle of S | rhs | g0 | scalars). */
typedef struct aar_comm_stats {
    int32_t world_size, rank, ranks_seen;
    int64_t allreduce_calls, allreduce_bytes, system_allreduce_bytes;
} aar_comm_stats;
int aar_comm_get_stats(const aar_comm *, aar_comm_stats *out);
/* In-process stand-in for a communicator (bring-up and tests on a 1-GPU box): `world_size` host threads of one process,
 * each with its own aar_problem on the same GPU, exchange through the group instead of RCCL; the sharded path itself
 * (frame ranges, every all-reduce, the final gather) is unchanged.  Every rank's calls must be made concurrently, one
 * thread per rank, exactly as separate processes would. */
typedef struct aar_local_group aar_local_group;
int aar_local_group_create(int32_t world_size, aar_local_group **out);
void aar_local_group_destroy(aar_local_group *);
int aar_comm_create_local(aar_local_group *group, int32_t rank, int32_t device_id, aar_comm **out);

/* ---------------------------------------------------------------------------------------------
 * The problem on the device.
 * ------------------------------------------------------------------------------------------- */
typedef struct aar_problem aar_problem;

enum { AAR_RES_F32 = 0,   /* reference-faithful: projection rounded to float, float subtraction
                             (cv::Point2f store + libs/multicam_mapper.cpp:1012-1013)               */
       AAR_RES_F64 = 1 }; /* same formula kept in double                                            */

typedef struct aar_problem_desc {
    int32_t num_cams, num_markers, num_frames;
    int32_t root_cam, root_marker;            /* indices                                            */
    const double *cam_mats;                   /* [C][9]                                             */
    double marker_size;
    int64_t num_obs;
    const int32_t *obs_frame, *obs_cam, *obs_marker;  /* frame-nondecreasing (reference order)      */
    const float *obs_uv;                      /* [N][8]                                             */
    int32_t optimize_cam_poses, optimize_marker_poses, optimize_object_poses;
    int32_t optimize_cam_intrinsics;          /* MultiCamMapper::Config's fourth flag (libs/multicam_mapper.h:75-81): every vector of this
                                                 problem -- x_full, z, delta -- then ENDS with 9 entries per camera, fx cx fy cy d0..d4
                                                 (fill_io_vec_cam_intrinsics, libs/multicam_mapper.cpp:488-498: all cameras, the root too), i.e.
                                                 x_full is the whole `.solution` vector (:1085-1089); cam_mats then only supplies nothing but
                                                 its shape -- the projection uses the pinhole matrix intrinsics_vec2mats rebuilds from those
                                                 four numbers (:580-593, no skew), the five distortion entries are carried along untouched
                                                 (their Jacobian columns are exact zeros in the reference: project_marker ignores them)     */
    int32_t residual_mode;                    /* AAR_RES_F32 | AAR_RES_F64                          */
    int32_t with_huber;                       /* MultiCamMapper::set_with_huber (libs/multicam_mapper.cpp:31-33,1014-1019): residual
                                                 rows scaled by sqrt(rho(e)/e); aar_lm_solve then also runs optCallBack's delta
                                                 schedule (:412-417): 10 at the start of solve(), -7.5/500 per step down to 2.5   */
    int32_t device_id;
    aar_comm *comm;                           /* NULL = single GPU; else observations are sharded by
                                                 frame range over the communicator's ranks           */
} aar_problem_desc;

void aar_problem_desc_from_dataset(const aar_dataset *, aar_problem_desc *);
int aar_problem_create(const aar_problem_desc *, aar_problem **out);   /* = aar_problem_create_ex(desc, NULL, out): solver AUTO */

/* How the damped normal equations of a try are solved -- the counterpart of configuring the reference's solver object through
 * SparseLevMarq::Params / setParams (libs/sparselevmarq.h:30-50,60-66): PER PROBLEM, fixed when the problem is created; two problems
 * of one process may differ.  The reference itself knows one way only (Eigen::SimplicialLDLT, :394-400): AAR_SOLVER_DIRECT is that
 * step to rounding.
 *   AAR_SOLVER_DIRECT  per-frame elimination (Schur complement) + dense blocked LDL^T of the reduced system: the reference's step
 *   AAR_SOLVER_SPCG    the same Schur complement, then block-Jacobi-preconditioned CG on the EXPLICIT reduced system, one wavefront
 *                      per camera / marker (csrc/spcg_kernels.hip), stopped at a relative residual pcg_eta: an inexact LM step.  A solve
 *                      that needs more than pcg_max_it iterations or whose hand-over times out (device shared with another
 *                      process) is redone with the direct chain automatically.  Needs 6 (C + M [+ C]) <= 1344.
 *   AAR_SOLVER_PCG     no Schur complement at all: CG THROUGH the frame blocks (csrc/pcg_kernels.hip); with a communicator the frames'
 *                      blocks stay on their ranks and every CG iteration all-reduces 8 n bytes (nothing O(n^3) is replicated).  At forcing terms
 *                      pcg_eta >= 1e-4 (not deterministic) the frame-entity coupling blocks are kept in fp32 (half the bytes of every CG pass): a rounding
 *                      of 6e-8 there is invisible beside such a forcing term (final poses unchanged: DESIGN.md section 6); tighter terms get fp64 blocks
 *   AAR_SOLVER_AUTO    THE DEFAULT (a NULL options pointer, aar_solver_default_options): the fastest of the three for the problem's size as measured on
 *                      MI355X (profiles/r05_auto_crossover.txt): one tile of unknowns (up to 16 cameras + markers) DIRECT; SPCG wherever
 *                      it fits (up to 224 cameras + markers), except on long sequences whose frames each see many entities, where PCG -- which never
 *                      forms the Schur complement -- overtakes it (from 96 entities on: (entity, frame) incidences x (incidences per frame - 30) >= 4e6, e.g.
 *                      BASELINE.json's 16-camera / 200-marker / 5000-frame configuration); PCG also where SPCG does not fit.
 * Inexact solvers stop an inner solve at a relative residual pcg_eta (PCG: |r| <= eta |b|; SPCG: sqrt(r^T M^-1 r) <= eta sqrt(b^T M^-1 b), M = the
 * block-Jacobi preconditioner).  The LM trajectory is then not the reference's step for step, and -- the reference's stopping rule being loose (its own
 * last step still moves the poses by ~3e-3) -- where a run ends along weakly determined directions depends on every step's accuracy.  The DEFAULT forcing
 * terms (SPCG 3e-4, PCG 5e-3) are therefore chosen for the final POSES: as transforms they agree with the DIRECT solver's to ~2e-6 in the rotation-matrix
 * entries and ~1e-6 m in the translations at BASELINE.json's configurations (tests/test_gpu_solvers.py asserts 1e-5), final reprojection error within
 * 1e-7 px; the direct path itself is ~1e-4 away from the reference-faithful CPU run (analytic against central-difference float Jacobian).
 * pcg_eta_loose > pcg_eta makes a forcing SEQUENCE (pcg_eta_loose while the last accepted LM step still took more than pcg_eta_switch of the error
 * away): faster, and measurably further from the direct run's poses (0.1 -> 0.02, round 4's default: 6e-4) -- opt-in.
 * AAR_SOLVER_DIRECT is the opt-out for callers who want the reference's every step (trace parity).
 * deterministic: every sum the default path leaves to fp64 atomics is taken in a fixed order (as the reference's ascending-row
 * accumulation is, libs/sparselevmarq.h:291-303): two runs give the same bits; slower.
 * Environment variables AAR_SOLVER (direct|spcg|pcg|auto), AAR_DETERMINISTIC, AAR_PCG_ETA, AAR_PCG_MAX_IT apply to problems whose caller
 * left the corresponding field at its default (NULL options, or AUTO / 0): tuning and bisecting only; aar_solver_stats.env_overrides
 * reports what they changed.  An explicitly set field always wins. */
enum { AAR_SOLVER_DIRECT = 0, AAR_SOLVER_PCG = 1, AAR_SOLVER_SPCG = 2, AAR_SOLVER_AUTO = 3 };
typedef struct aar_solver_options {
    uint32_t struct_size;                     /* sizeof(aar_solver_options) of the caller: fields beyond it keep their defaults      */
    int32_t solver;                           /* AAR_SOLVER_*                                                                        */
    int32_t deterministic;                    /* 0 | 1                                                                               */
    int32_t pcg_max_it;                       /* iteration cap of an inner CG solve; 0 = default (PCG 200; SPCG 64 up to four tiles of unknowns,
                                                 128 above -- 128 is also its maximum)                                                */
    double pcg_eta;                           /* forcing term of the inner solves; 0 = default (SPCG 3e-4, PCG 5e-3; aar_solver_stats.pcg_eta reports it)  */
    double pcg_eta_loose;                     /* > pcg_eta: forcing term of the EARLY LM steps (a forcing sequence); 0 = none (default)            */
    double pcg_eta_switch;                    /* an LM step is "early" while the last accepted step took more than this share of the error
                                                 away; 0 = default (0.01)                                                              */
    double pcg_abs_tol;                       /* ABSOLUTE tolerance of an inner solve beside the relative one (both must hold), in the units of the pose
                                                 vector (radians / metres): what the inexact solve may leave out of the step along a weakly determined
                                                 direction, measured in the preconditioner's norm (r^T M^-1 r <= tol^2 mu).  0 = default (SPCG 2e-5, PCG 5e-5); it only binds where the step is large while the damping is small (far
                                                 starts, tau << 1): there the CG runs on, or hands the try to the direct chain at its iteration cap       */
} aar_solver_options;
void aar_solver_default_options(aar_solver_options *);   /* struct_size set, AUTO, not deterministic, default forcing sequence / cap */
int aar_problem_create_ex(const aar_problem_desc *, const aar_solver_options *, aar_problem **out);
/* what the problem runs with (AUTO resolved), and what its inner solver has done so far.  The CALLER sets struct_size = sizeof(aar_solver_stats)
 * before the call; the library fills at most that many bytes, so fields appended LATER do not break a caller built against this layout.
 * ABI note: struct_size itself arrived in round 5, as the FIRST member -- an incompatible change against the round-4 struct (which began with `solver` and
 * carried a `reserved` word): a binary built against round 4 must be rebuilt.  From here on the struct only grows at its end. */
enum { AAR_ENV_SOLVER = 1, AAR_ENV_DETERMINISTIC = 2, AAR_ENV_PCG_ETA = 4, AAR_ENV_PCG_MAX_IT = 8 };
typedef struct aar_solver_stats {
    uint32_t struct_size;                     /* in: sizeof(aar_solver_stats) of the caller                                            */
    int32_t solver;                           /* AAR_SOLVER_DIRECT | _PCG | _SPCG: never AUTO                                         */
    int32_t deterministic;
    int32_t last_iterations;                  /* CG iterations of the last damped solve (0 for DIRECT)                                */
    int64_t total_iterations, solves;         /* since the problem was created                                                        */
    int64_t fallbacks;                        /* SPCG: tries redone with the direct chain (iteration cap, hand-over time-out)          */
    double pcg_eta;                           /* forcing term of the inner solves                                                      */
    int32_t pcg_max_it;
    int32_t env_overrides;                    /* AAR_ENV_* bits: fields an environment variable changed for this problem               */
    int64_t same_xcd_solves;                  /* SPCG: solves whose wavefronts all ran on one XCD (hand-overs through that XCD's L2: the fast case)   */
    double pcg_eta_loose;                     /* forcing term of the early LM steps (0: pcg_eta throughout)                            */
    double pcg_eta_switch;
    double pcg_abs_tol;
} aar_solver_stats;
int aar_problem_get_solver_stats(aar_problem *, aar_solver_stats *out);
/* TESTING ONLY: fault injection for the solvers' fall-back paths.  AAR_TEST_HOOK_SPCG_DROP: the CG wavefront of shared entity `value` never
 * shows up (what a device shared with another process can do): every hand-over times out, the try is redone by the direct chain; -1 clears. */
enum { AAR_TEST_HOOK_SPCG_DROP = 1 };
int aar_problem_set_test_hook(aar_problem *, int32_t hook, int32_t value);
void aar_problem_destroy(aar_problem *);
int64_t aar_problem_full_len(const aar_problem *);    /* length of x_full (+ 9 per camera with optimize_cam_intrinsics) */
int64_t aar_problem_num_vars(const aar_problem *);    /* length of the reference's z for the Config  */
int64_t aar_problem_local_obs(const aar_problem *);   /* observations owned by this rank             */
/* MultiCamMapper::hubberDelta: the delta the next residual evaluations use (only meaningful with with_huber) */
int aar_problem_set_huber_delta(aar_problem *, float delta);
float aar_problem_get_huber_delta(const aar_problem *);

/* error_function (libs/multicam_mapper.cpp:731-737): r (8*num_obs doubles, reference row order; may be
 * NULL; single-GPU only when non-NULL) and sum of squares (all ranks). */
int aar_eval_residuals(aar_problem *, const double *x_full, double *r, double *sum_sq);

/* J^T J and B = -J^T r of libs/sparselevmarq.h:355-367 for the analytic Jacobian, assembled DENSE in the
 * reference's z ordering (P x P row-major, P = aar_problem_num_vars).  For checking the block
 * accumulation kernels on small problems (single GPU).  Either output may be NULL. */
int aar_eval_normal_equations(aar_problem *, const double *x_full, double *JtJ, double *B, double *sum_sq);

/* delta of (J^T J + mu I) delta = B through the device Schur-complement + dense Cholesky path
 * (libs/sparselevmarq.h:384-400), in z ordering. */
int aar_eval_damped_step(aar_problem *, const double *x_full, double mu, double *delta);

/* ucoslam::SparseLevMarq<T>::Params (libs/sparselevmarq.h:30-50) with the values
 * MultiCamMapper::init installs (libs/multicam_mapper.cpp:326-330). */
typedef struct aar_lm_params {
    int32_t max_iters;                    /* 10000                                                    */
    double min_error;                     /* 1e-5                                                     */
    double min_step_error_diff;           /* 0                                                        */
    double min_average_step_error_diff;   /* 1e-4                                                     */
    double tau;                           /* 1                                                        */
    int32_t verbose;
} aar_lm_params;
void aar_lm_default_params(aar_lm_params *);

typedef struct aar_lm_iter {              /* one step() */
    double err, mu, gain, delta_norm;
    int32_t accepted, tries;
} aar_lm_iter;

typedef struct aar_lm_report {
    int32_t iterations;                   /* step() calls made                                        */
    int32_t stop_code;                    /* mustExit of libs/sparselevmarq.h:458-461 (0 = maxIters)  */
    double initial_err, final_err;        /* sum of squared residuals                                 */
    double final_mu;
    double solve_seconds;                 /* wall time of the loop (device-synchronised)              */
    int64_t trial_points;                 /* residual evaluations inside step()                        */
    aar_lm_iter *trace;                   /* optional caller array                                     */
    int32_t trace_cap;
} aar_lm_report;

/* step-by-step mode: SparseLevMarq::init / step / getCurrentSolution (libs/sparselevmarq.h:238-249,349-437) */
int aar_lm_init(aar_problem *, const double *x_full, const aar_lm_params *);
int aar_lm_step(aar_problem *, aar_lm_iter *out);
int aar_lm_get_solution(aar_problem *, double *x_full, double *err);
/* SparseLevMarq::solve(z, f, J) (libs/sparselevmarq.h:440-472): x_full in/out.  The step callback and the stop function below
 * are honoured exactly where the reference calls them (:446-451, :463). */
int aar_lm_solve(aar_problem *, double *x_full, const aar_lm_params *, aar_lm_report *);

/* The solver seam of ucoslam::SparseLevMarq<T> (libs/sparselevmarq.h:118-123).
 *  setStepCallBackFunc: called on the host after every step() with curr_z -- the reference's z for the problem's Config
 *    (mats2eVec order, aar_problem_num_vars doubles).  want_z = 0 spares the device -> host copy of curr_z for callbacks that
 *    ignore it, as MultiCamMapper::optCallBack does (libs/multicam_mapper.cpp:412-417); z is then NULL.
 *  setStopFunction: when set, solve() runs do { step(); callback; } while (!stop(curr_z)) with NO iteration cap and none of the
 *    error-based exits (:444-450); nonzero = stop.
 * NULL clears.  A with_huber problem without a step callback runs the mapper's own pair -- hubberDelta = 10 at the start of
 * solve(), optCallBack's schedule per step; a caller that installs a callback owns both (aar_problem_set_huber_delta).
 * On a sharded problem every rank must install the same kind of callback (fetching curr_z is a collective). */
typedef void (*aar_lm_step_callback)(void *ctx, const double *z, int64_t num_vars);
typedef int (*aar_lm_stop_function)(void *ctx, const double *z, int64_t num_vars);
int aar_lm_set_step_callback(aar_problem *, aar_lm_step_callback fn, void *ctx, int32_t want_z);
int aar_lm_set_stop_function(aar_problem *, aar_lm_stop_function fn, void *ctx);
/* z <-> x_full for the problem's Config (mats2eVec / eVec2Mats, libs/multicam_mapper.cpp:445-461,595-606): extract copies the
 * optimised groups of x_full into z; merge writes z back into x_full and leaves the fixed groups alone. */
int aar_problem_extract_z(const aar_problem *, const double *x_full, double *z);
int aar_problem_merge_z(const aar_problem *, const double *z, double *x_full);

/* MultiCamMapper::track() (libs/multicam_mapper.cpp:430-443) for every frame of the problem at once: cameras and markers
 * stay at their x_full values, each frame's object pose is refined on its own by the LM of SparseLevMarq::solve(z, f)
 * (libs/sparselevmarq.h:223-228) over error_function_tracking (libs/multicam_mapper.cpp:678-729: double residuals, Huber
 * weights with the problem's current delta when with_huber).  The whole loop of a frame runs on the device.
 * iterations / final_err: optional [num_frames] outputs (step() calls made, final sum of squares per frame). */
int aar_track(aar_problem *, double *x_full, const aar_lm_params *, int32_t *iterations, double *final_err);

/* per-stage device time of the last aar_lm_solve, seconds, in the reference's verbose-timer vocabulary
 * (libs/sparselevmarq.h:425) extended with the stages that only exist here */
typedef struct aar_stage_times {
    double unpack, jacobian_normal_eq, schur, chol, backsub, residual, control, allreduce, total;
    int64_t launches;
} aar_stage_times;
int aar_get_stage_times(aar_problem *, aar_stage_times *);
/* Stage timers on / off (also on with AAR_STAGE_TIMERS=1 in the environment when the problem is created): every stage of a
 * step is then bracketed by two HIP events on the library's stream and waited for, which serialises host and device -- a
 * diagnostic mode (bench.py's `amdahl` object, the verbose stage line), not the production path.  Resets the accumulators. */
int aar_set_stage_timers(aar_problem *, int on);

/* out[0] = CG iterations of the last damped solve, out[1] = their running total since the problem was created (zeros for AAR_SOLVER_DIRECT);
 * aar_problem_get_solver_stats says more */
int aar_problem_pcg_iterations(aar_problem *, int32_t out[2]);

/* Per-kernel device time: when profiling is on, every kernel launch of this problem is bracketed by two HIP
 * events on the library's own stream (the stream the kernels run on) and the elapsed times are accumulated
 * per kernel.  bench.py's roofline figures come from here.  Switching profiling on resets the accumulators. */
#define AAR_NUM_KERNELS 17
int aar_set_kernel_profiling(aar_problem *, int on);
int aar_get_kernel_times(aar_problem *, double seconds[AAR_NUM_KERNELS], int64_t launches[AAR_NUM_KERNELS]);
const char *aar_kernel_name(int kernel_id);

/* fp64 reprojection statistics at x_full (device): per-corner RMSE sqrt(sum r^2 / 4N), sum r^2 */
int aar_reproj_stats(aar_problem *, const double *x_full, double *rmse, double *sum_sq);

int aar_device_count(void);
int aar_device_synchronize(void);

#ifdef __cplusplus
}
#endif
#endif
