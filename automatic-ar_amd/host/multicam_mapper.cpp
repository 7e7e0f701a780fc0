// MultiCamMapper over the C ABI (see multicam_mapper.h).  Only packing / unpacking of parameter
// vectors and file I/O happen on the host; residuals, Jacobians and the LM solve are HIP kernels.
#include "multicam_mapper.h"

#include <cstdio>
#include <cstring>
#include <functional>
#include <iostream>
#include <stdexcept>
#include <string>

#include "internal.h"
#include "se3.h"

namespace aar {

MultiCamMapper::MultiCamMapper() {
    solver_params.maxIters = 10000;  // libs/multicam_mapper.cpp:337-343
    solver_params.min_average_step_error_diff = 1e-4;
    memset(&last_report, 0, sizeof last_report);
}

MultiCamMapper::MultiCamMapper(aar_dataset *dataset) : MultiCamMapper() {
    data_ = dataset;
    solver_params.verbose = true;  // :327
    if (data_) {
        config_.optimize_cam_poses = data_->optimize_cam_poses != 0;
        config_.optimize_marker_poses = data_->optimize_marker_poses != 0;
        config_.optimize_object_poses = data_->optimize_object_poses != 0;
        config_.optimize_cam_intrinsics = data_->optimize_cam_intrinsics != 0;
        mats2eVec();
    }
}

MultiCamMapper::MultiCamMapper(Initializer &initializer) : MultiCamMapper(initializer.release()) {}

static Rigid from44(const Mat44 &m) {
    Rigid T;
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) T.R[r * 3 + c] = m[r * 4 + c];
        T.t[r] = m[r * 4 + 3];
    }
    return T;
}

MultiCamMapper::MultiCamMapper(size_t root_c, const std::map<int, Mat44> &T_to_root_cam, size_t root_m, const std::map<int, Mat44> &T_to_root_marker,
                               const std::map<int, Mat44> &obj_transforms, const FrameCamMarkers &fcm, float m_size,
                               std::vector<aar_cam_model> &cam_confs)
    : MultiCamMapper() {
    init(root_c, T_to_root_cam, root_m, T_to_root_marker, obj_transforms, fcm, m_size, cam_confs);
}

// frames (ascending id), their poses, and the detections in the reference's residual order -- frame, camera, detection order
// (libs/multicam_mapper.cpp:1001-1007) -- minus what fill_iteration_arrays erases: cameras / markers without a transform (:356-367)
void MultiCamMapper::load_frames(const std::map<int, Mat44> &object_poses, const FrameCamMarkers &fcm) {
    aar_dataset *old = data_;
    std::map<int, int> cam_index, marker_index, frame_index;
    for (int c = 0; c < old->num_cams; c++) cam_index[old->cam_ids[c]] = c;
    for (int m = 0; m < old->num_markers; m++) marker_index[old->marker_ids[m]] = m;
    int fi = 0;
    for (const auto &kv : object_poses) frame_index[kv.first] = fi++;
    struct O { int f, c, m; const float *uv; };
    std::vector<O> obs;
    for (const auto &fr : fcm) {
        auto f = frame_index.find(fr.first);
        if (f == frame_index.end()) throw std::runtime_error("MultiCamMapper::init: detections of frame " + std::to_string(fr.first) + " without an object pose");   // the reference: std::map::at
        for (const auto &cm : fr.second) {
            auto c = cam_index.find(cm.first);
            if (c == cam_index.end()) continue;
            for (const Marker &mk : cm.second) {
                auto m = marker_index.find(mk.id);
                if (m == marker_index.end()) continue;
                obs.push_back({f->second, c->second, m->second, mk.corners});
            }
        }
    }
    const int C = old->num_cams, M = old->num_markers, F = (int)object_poses.size();
    aar_dataset *d = dataset_alloc(C, M, F, (int64_t)obs.size(), false);
    memcpy(d->cam_ids, old->cam_ids, sizeof(int32_t) * C);
    memcpy(d->marker_ids, old->marker_ids, sizeof(int32_t) * M);
    memcpy(d->image_sizes, old->image_sizes, sizeof(int32_t) * 2 * C);
    memcpy(d->cam_mats, old->cam_mats, sizeof(double) * 9 * C);
    memcpy(d->dist_coeffs, old->dist_coeffs, sizeof(double) * 5 * C);
    d->root_cam = old->root_cam; d->root_marker = old->root_marker; d->marker_size = old->marker_size;
    d->optimize_cam_poses = old->optimize_cam_poses; d->optimize_marker_poses = old->optimize_marker_poses;
    d->optimize_object_poses = old->optimize_object_poses; d->optimize_cam_intrinsics = old->optimize_cam_intrinsics;
    const int64_t shared = 6LL * (C - 1) + 6LL * (M - 1);
    memcpy(d->x_full, old->x_full, sizeof(double) * shared);
    fi = 0;
    for (const auto &kv : object_poses) {
        d->frame_ids[fi] = kv.first;
        rigid_to_pose(from44(kv.second), d->x_full + shared + 6LL * fi);
        fi++;
    }
    for (size_t i = 0; i < obs.size(); i++) {
        d->obs_frame[i] = obs[i].f; d->obs_cam[i] = obs[i].c; d->obs_marker[i] = obs[i].m;
        memcpy(d->obs_uv + 8 * i, obs[i].uv, sizeof(float) * 8);
    }
    drop_problem();
    aar_dataset_free(old);
    data_ = d;
    remove_distortions();
    mats2eVec();
}

void MultiCamMapper::init(const std::map<int, Mat44> &object_poses, const FrameCamMarkers &fcm) {   // libs/multicam_mapper.cpp:272-279
    if (!data_) throw std::runtime_error("MultiCamMapper::init(object_poses, fcm): the mapper holds no cameras / markers yet");
    load_frames(object_poses, fcm);
}

void MultiCamMapper::init(size_t root_c, const std::map<int, Mat44> &T_to_root_cam, size_t root_m, const std::map<int, Mat44> &T_to_root_marker,
                          const std::map<int, Mat44> &object_poses, const FrameCamMarkers &fcm, float m_size,
                          const std::vector<aar_cam_model> &cam_confs) {   // libs/multicam_mapper.cpp:281-335
    if (T_to_root_cam.empty() || T_to_root_marker.empty()) throw std::runtime_error("MultiCamMapper::init: no cameras / markers");
    if (!T_to_root_cam.count((int)root_c) || !T_to_root_marker.count((int)root_m)) throw std::runtime_error("MultiCamMapper::init: root id without a transform");
    const int C = (int)T_to_root_cam.size(), M = (int)T_to_root_marker.size();
    aar_dataset *d = dataset_alloc(C, M, 0, 0, false);
    cam_models_.assign(C, aar_cam_model());
    int i = 0;
    for (const auto &kv : T_to_root_cam) {   // ascending id = MatArray order (libs/multicam_mapper.h:95-105)
        const int id = kv.first;
        if (id < 0 || id >= (int)cam_confs.size()) { aar_dataset_free(d); throw std::runtime_error("MultiCamMapper::init: no calibration for camera id " + std::to_string(id)); }   // cam_configs[cam_id], :314
        d->cam_ids[i] = id;
        if (id == (int)root_c) d->root_cam = i;
        const aar_cam_model &cc = cam_confs[id];
        memcpy(d->cam_mats + 9 * i, cc.K, sizeof cc.K);
        for (int j = 0; j < 5; j++) d->dist_coeffs[5 * i + j] = cc.dist[j];
        d->image_sizes[2 * i] = cc.width; d->image_sizes[2 * i + 1] = cc.height;
        cam_models_[i] = cc;
        i++;
    }
    i = 0;
    for (const auto &kv : T_to_root_marker) {
        d->marker_ids[i] = kv.first;
        if (kv.first == (int)root_m) d->root_marker = i;
        i++;
    }
    PoseLayout L;
    L.C = C; L.M = M; L.F = 0; L.rc = d->root_cam; L.rm = d->root_marker;
    i = 0;
    for (const auto &kv : T_to_root_cam) { if (i != L.rc) rigid_to_pose(from44(kv.second), d->x_full + L.full_cam0() + 6LL * L.cam_slot(i)); i++; }
    i = 0;
    for (const auto &kv : T_to_root_marker) { if (i != L.rm) rigid_to_pose(from44(kv.second), d->x_full + L.full_mk0() + 6LL * L.mk_slot(i)); i++; }
    d->marker_size = (double)m_size;   // float parameter, libs/multicam_mapper.h:20
    d->optimize_cam_intrinsics = 1;    // a fresh mapper holds the default Config (libs/multicam_mapper.h:75-81)
    drop_problem();
    aar_dataset_free(data_);
    data_ = d;
    config_ = Config();
    solver_params = SparseLevMarq<double>::Params();   // :326-330
    solver_params.verbose = true;
    solver_params.maxIters = 10000;
    solver_params.min_average_step_error_diff = 1e-4;
    load_frames(object_poses, fcm);
    // eval_curr_solution + "the very initial error" (:331-333); the pose groups only, as error_function evaluates them
    const Config keep = config_;
    config_.optimize_cam_intrinsics = false;
    drop_problem();
    if (ensure_problem()) { config_ = keep; throw std::runtime_error(aar_last_error()); }
    double e0 = 0;
    const int rc = aar_eval_residuals(problem_, data_->x_full, nullptr, &e0);
    config_ = keep;
    drop_problem();   // (built for the pose groups only)
    if (rc) throw std::runtime_error(aar_last_error());
    std::cout << "the very initial error: " << e0 << std::endl;
}

MultiCamMapper::~MultiCamMapper() {
    drop_problem();
    aar_dataset_free(data_);
}

void MultiCamMapper::drop_problem() {
    solver.detach();   // the solver keeps the raw handle between init() / step() calls: never let it outlive the problem
    if (problem_) aar_problem_destroy(problem_);
    problem_ = nullptr;
}

namespace detail {
EvalProbe &eval_probe() {
    static thread_local EvalProbe p;
    return p;
}
aar_problem *current_problem(const MultiCamMapper *owner) { return owner ? owner->problem_ : nullptr; }
aar_problem *bind_problem(MultiCamMapper *owner, std::vector<double> &x_full) {
    if (!owner || !owner->data_) throw std::runtime_error("SparseLevMarq: the evaluation functions belong to a MultiCamMapper without a data set");
    if (owner->ensure_problem()) throw std::runtime_error(aar_last_error());
    if (owner->with_huber_ && aar_problem_set_huber_delta(owner->problem_, owner->hubberDelta)) throw std::runtime_error(aar_last_error());
    x_full = owner->problem_vector();
    return owner->problem_;
}
}  // namespace detail

bool MultiCamMapper::probed(int kind) {
    detail::EvalProbe &p = detail::eval_probe();
    if (!p.active) return false;
    p.id.owner = this;
    p.id.kind = kind;
    return true;
}

void MultiCamMapper::set_optmize_flag_cam_poses(bool f) { config_.optimize_cam_poses = f; drop_problem(); }
void MultiCamMapper::set_optmize_flag_marker_poses(bool f) { config_.optimize_marker_poses = f; drop_problem(); }
void MultiCamMapper::set_optmize_flag_object_poses(bool f) { config_.optimize_object_poses = f; drop_problem(); }
void MultiCamMapper::set_optmize_flag_cam_intrinsics(bool f) { config_.optimize_cam_intrinsics = f; drop_problem(); }
void MultiCamMapper::set_with_huber(bool wh) { with_huber_ = wh; drop_problem(); }
void MultiCamMapper::set_config(Config &conf) { config_ = conf; drop_problem(); }

size_t MultiCamMapper::get_num_vars(const Config &conf) {  // libs/multicam_mapper.cpp:239-250
    if (!data_) return 0;
    size_t n = 0;
    if (conf.optimize_cam_poses) n += (size_t)(data_->num_cams - 1) * 6;
    if (conf.optimize_marker_poses) n += (size_t)(data_->num_markers - 1) * 6;
    if (conf.optimize_object_poses) n += (size_t)data_->num_frames * 6;
    if (conf.optimize_cam_intrinsics) n += (size_t)data_->num_cams * 9;
    return n;
}

// mats2eVec (:445-461): the optimised groups of x_full, cameras | markers | frames | intrinsics
void MultiCamMapper::mats2eVec() {
    io_vec.assign(get_num_vars(config_), 0.0);
    if (!data_) return;
    PoseLayout L;
    L.C = data_->num_cams; L.M = data_->num_markers; L.F = data_->num_frames;
    size_t k = 0;
    auto copy = [&](int64_t off, int64_t n) { for (int64_t i = 0; i < n; i++) io_vec[k++] = data_->x_full[off + i]; };
    if (config_.optimize_cam_poses) copy(L.full_cam0(), 6LL * (L.C - 1));
    if (config_.optimize_marker_poses) copy(L.full_mk0(), 6LL * (L.M - 1));
    if (config_.optimize_object_poses) copy(L.full_fr0(), 6LL * L.F);
    if (config_.optimize_cam_intrinsics)
        for (int c = 0; c < L.C; c++) {  // fill_io_vec_cam_intrinsics, :488-498
            const double *K = data_->cam_mats + 9 * c;
            io_vec[k++] = K[0]; io_vec[k++] = K[2]; io_vec[k++] = K[4]; io_vec[k++] = K[5];
            for (int j = 0; j < 5; j++) io_vec[k++] = data_->dist_coeffs[5 * c + j];
        }
}

// eVec2Mats (:595-606)
void MultiCamMapper::eVec2Mats(const eVector &v) {
    PoseLayout L;
    L.C = data_->num_cams; L.M = data_->num_markers; L.F = data_->num_frames;
    size_t k = 0;
    auto copy = [&](int64_t off, int64_t n) { for (int64_t i = 0; i < n; i++) data_->x_full[off + i] = v[k++]; };
    if (config_.optimize_cam_poses) copy(L.full_cam0(), 6LL * (L.C - 1));
    if (config_.optimize_marker_poses) copy(L.full_mk0(), 6LL * (L.M - 1));
    if (config_.optimize_object_poses) copy(L.full_fr0(), 6LL * L.F);
    if (config_.optimize_cam_intrinsics)
        for (int c = 0; c < L.C; c++) {  // intrinsics_vec2mats, :580-593: cv::Mat::eye with fx, cx, fy, cy (a skew is gone), then d0..d4
            double *K = data_->cam_mats + 9 * c;
            K[0] = v[k++]; K[1] = 0; K[2] = v[k++]; K[3] = 0; K[4] = v[k++]; K[5] = v[k++]; K[6] = 0; K[7] = 0; K[8] = 1;
            for (int j = 0; j < 5; j++) data_->dist_coeffs[5 * c + j] = v[k++];
        }
}

// x_full of the device problem for the current Config: the pose vector, followed -- with optimize_cam_intrinsics -- by
// fx cx fy cy d0..d4 per camera (fill_io_vec_cam_intrinsics, :488-498), i.e. the `.solution` vector
std::vector<double> MultiCamMapper::problem_vector() {
    std::vector<double> x(data_->x_full, data_->x_full + aar_dataset_full_len(data_));
    if (config_.optimize_cam_intrinsics)
        for (int c = 0; c < data_->num_cams; c++) {
            const double *K = data_->cam_mats + 9 * c;
            x.push_back(K[0]); x.push_back(K[2]); x.push_back(K[4]); x.push_back(K[5]);
            for (int j = 0; j < 5; j++) x.push_back(data_->dist_coeffs[5 * c + j]);
        }
    return x;
}

int MultiCamMapper::ensure_problem() {
    if (problem_) return AAR_OK;
    aar_problem_desc d;
    aar_problem_desc_from_dataset(data_, &d);
    d.optimize_cam_poses = config_.optimize_cam_poses;
    d.optimize_marker_poses = config_.optimize_marker_poses;
    d.optimize_object_poses = config_.optimize_object_poses;
    d.optimize_cam_intrinsics = config_.optimize_cam_intrinsics;
    d.residual_mode = residual_mode;
    d.with_huber = with_huber_ ? 1 : 0;
    d.device_id = device_id;
    aar_solver_options so;
    aar_solver_default_options(&so);
    so.solver = solver_options_.solver;
    so.deterministic = solver_options_.deterministic ? 1 : 0;
    so.pcg_eta = solver_options_.pcg_eta;
    so.pcg_max_it = solver_options_.pcg_max_it;
    so.pcg_eta_loose = solver_options_.pcg_eta_loose;
    so.pcg_eta_switch = solver_options_.pcg_eta_switch;
    so.pcg_abs_tol = solver_options_.pcg_abs_tol;
    int rc = aar_problem_create_ex(&d, &so, &problem_);
    if (!rc && with_huber_) rc = aar_problem_set_huber_delta(problem_, hubberDelta);
    return rc;
}

void MultiCamMapper::set_solver_options(const SolverOptions &o) {
    solver_options_ = o;
    drop_problem();   // (the solver is a property of the device problem: the next solve() / track() builds one with these options)
}

aar_solver_stats MultiCamMapper::solver_stats() {
    if (!data_) throw std::runtime_error("MultiCamMapper::solver_stats: no data set");
    if (ensure_problem()) throw std::runtime_error(aar_last_error());
    aar_solver_stats st;
    st.struct_size = (uint32_t)sizeof st;
    if (aar_problem_get_solver_stats(problem_, &st)) throw std::runtime_error(aar_last_error());
    return st;
}

void MultiCamMapper::error_function(const eVector &input, eVector &error) {
    if (probed(detail::EVAL_ERROR_FUNCTION)) return;
    if (!data_) throw std::runtime_error("MultiCamMapper::error_function: no data set");
    if (input.size() != get_num_vars(config_)) throw std::runtime_error("MultiCamMapper::error_function: input has not the Config's number of variables");
    if (ensure_problem()) throw std::runtime_error(aar_last_error());
    if (with_huber_ && aar_problem_set_huber_delta(problem_, hubberDelta)) throw std::runtime_error(aar_last_error());
    std::vector<double> x = problem_vector();
    if (aar_problem_merge_z(problem_, input.data(), x.data())) throw std::runtime_error(aar_last_error());
    error.assign(8 * (size_t)data_->num_obs, 0.0);
    if (aar_eval_residuals(problem_, x.data(), error.data(), nullptr)) throw std::runtime_error(aar_last_error());
}

void MultiCamMapper::jacobian_function(const eVector &, SparseJacobian<double> &) {
    if (probed(detail::EVAL_JACOBIAN_FUNCTION)) return;
    throw std::logic_error("MultiCamMapper::jacobian_function: the Jacobian is analytic and lives on the device (k_passA / k_passB accumulate its blocks "
                           "into J^T J); hand this function to SparseLevMarq::solve / step instead of calling it");
}

void MultiCamMapper::error_function_tracking(const eVector &, eVector &) {
    if (probed(detail::EVAL_ERROR_FUNCTION_TRACKING)) return;
    throw std::logic_error("MultiCamMapper::error_function_tracking: the per-frame residuals of tracking are evaluated inside k_track; hand this "
                           "function to SparseLevMarq::solve(z, f) (or call track()) instead of calling it");
}

// optCallBack, libs/multicam_mapper.cpp:412-417: the Huber delta schedule, driven by the solver's step callback
void MultiCamMapper::optCallBack(const eVector &) {
    if (hubberDelta > 2.5) hubberDelta -= 7.5 / 500;
    if (with_huber_ && problem_ && aar_problem_set_huber_delta(problem_, hubberDelta)) throw std::runtime_error(aar_last_error());
}

void MultiCamMapper::solve() {   // libs/multicam_mapper.cpp:419-428, statement for statement
    if (!data_) throw std::runtime_error("MultiCamMapper::solve: no data set");
    using namespace std::placeholders;
    mats2eVec();
    solver.setParams(solver_params);   // (the reference installs them in init(), :326-330; a mirror's caller may edit solver_params until here)
    solver.setStepCallBackFunc(std::bind(&MultiCamMapper::optCallBack, this, _1), /*needs_z=*/false);
    {   // error_function(io_vec, error); cout << error.dot(error): the sum alone, without bringing 8N residuals to the host
        std::vector<double> x_start;
        aar_problem *pb = detail::bind_problem(this, x_start);
        double e0 = 0;
        if (aar_eval_residuals(pb, x_start.data(), nullptr, &e0)) throw std::runtime_error(aar_last_error());
        std::cout << "initial_error: " << e0 << "error size: " << 8 * data_->num_obs << std::endl;  // :424
    }
    hubberDelta = 10;
    solver.solve(io_vec, std::bind(&MultiCamMapper::error_function, this, _1, _2), std::bind(&MultiCamMapper::jacobian_function, this, _1, _2));
    last_report = solver.report;
    eVec2Mats(io_vec);
}

// track(), libs/multicam_mapper.cpp:430-443.  The reference refines ONE frame per call (apps/track.cpp:127-131 re-inits the
// mapper with the frame's detections each time); here every frame held by the data set is refined in one launch, each with
// its own LM (detail::solve_tracking -> aar_track).
void MultiCamMapper::track() {
    if (!data_) throw std::runtime_error("MultiCamMapper::track: no data set");
    using namespace std::placeholders;
    mats2eVec();
    hubberDelta = 10;  // :439
    solver.setParams(solver_params);
    solver.solve(io_vec, std::bind(&MultiCamMapper::error_function_tracking, this, _1, _2));
    eVec2Mats(io_vec);
}

namespace detail {
double solve_tracking(MultiCamMapper *m, std::vector<double> &z) {
    std::vector<double> x;
    aar_problem *pb = bind_problem(m, x);
    if (z.size() != m->get_num_vars(m->config_)) throw std::runtime_error("SparseLevMarq::solve: z has not the Config's number of variables");
    if (aar_problem_merge_z(pb, z.data(), x.data())) throw std::runtime_error(aar_last_error());
    aar_lm_params p;
    aar_lm_default_params(&p);
    const SparseLevMarq<double>::Params &sp = m->solver._params;
    p.max_iters = sp.maxIters;
    p.min_error = sp.minError;
    p.min_step_error_diff = sp.min_step_error_diff;
    p.min_average_step_error_diff = sp.min_average_step_error_diff;
    p.tau = sp.tau;
    m->track_iterations.assign(m->data_->num_frames, 0);
    m->track_errors.assign(m->data_->num_frames, 0.0);
    if (aar_track(pb, x.data(), &p, m->track_iterations.data(), m->track_errors.data())) throw std::runtime_error(aar_last_error());
    if (aar_problem_extract_z(pb, x.data(), z.data())) throw std::runtime_error(aar_last_error());   // (only the frame poses have moved)
    memcpy(m->data_->x_full, x.data(), sizeof(double) * aar_dataset_full_len(m->data_));   // ... also when the Config keeps them out of z
    double e = 0;
    for (double v : m->track_errors) e += v;
    return e;
}
}  // namespace detail

bool MultiCamMapper::write_solution_file(std::string path) {
    if (!data_) return false;
    aar_dataset tmp = *data_;
    tmp.optimize_cam_poses = config_.optimize_cam_poses;
    tmp.optimize_marker_poses = config_.optimize_marker_poses;
    tmp.optimize_object_poses = config_.optimize_object_poses;
    tmp.optimize_cam_intrinsics = config_.optimize_cam_intrinsics;
    if (aar_solution_write(path.c_str(), &tmp)) {
        std::cout << aar_last_error() << std::endl;
        return false;
    }
    return true;
}

bool MultiCamMapper::read_solution_file(std::string path) {
    aar_dataset *d = nullptr;
    if (aar_solution_read(path.c_str(), &d)) {
        std::cout << aar_last_error() << std::endl;
        return false;
    }
    drop_problem();
    aar_dataset_free(data_);
    data_ = d;
    cam_models_.clear();
    config_.optimize_cam_poses = d->optimize_cam_poses != 0;
    config_.optimize_marker_poses = d->optimize_marker_poses != 0;
    config_.optimize_object_poses = d->optimize_object_poses != 0;
    config_.optimize_cam_intrinsics = d->optimize_cam_intrinsics != 0;
    mats2eVec();
    return true;
}

void MultiCamMapper::write_text_solution_file(std::string text_path) {
    if (!data_ || aar_solution_write_yaml(text_path.c_str(), data_)) throw std::runtime_error(aar_last_error());
}

static Mat44 to44(const Rigid &T) {
    Mat44 m;
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) m[r * 4 + c] = T.R[r * 3 + c];
        m[r * 4 + 3] = T.t[r];
    }
    m[12] = m[13] = m[14] = 0;
    m[15] = 1;
    return m;
}

MultiCamMapper::MatArrays MultiCamMapper::get_mat_arrays() {
    MatArrays ma;
    if (!data_) return ma;
    PoseLayout L;
    L.C = data_->num_cams; L.M = data_->num_markers; L.F = data_->num_frames; L.rc = data_->root_cam; L.rm = data_->root_marker;
    for (int c = 0; c < L.C; c++)
        ma.transforms_to_root_cam[data_->cam_ids[c]] = to44(c == L.rc ? Rigid::identity() : pose_to_rigid(data_->x_full + L.full_cam0() + 6LL * L.cam_slot(c)));
    for (int m = 0; m < L.M; m++)
        ma.transforms_to_root_marker[data_->marker_ids[m]] = to44(m == L.rm ? Rigid::identity() : pose_to_rigid(data_->x_full + L.full_mk0() + 6LL * L.mk_slot(m)));
    for (int f = 0; f < L.F; f++) ma.object_to_global[data_->frame_ids[f]] = to44(pose_to_rigid(data_->x_full + L.full_fr0() + 6LL * f));
    return ma;
}

// ---- Initializer mirror (libs/initializer.h) ----
Initializer::Initializer(const aar_detections *dts, double marker_s, const std::vector<aar_cam_model> &cam_c,
                         const std::set<int> &excluded_cs, int device_id) {
    aar_init_params prm;
    aar_init_default_params(&prm);
    prm.marker_size = marker_s;
    prm.device_id = device_id;
    std::vector<int32_t> ex(excluded_cs.begin(), excluded_cs.end());
    prm.n_excluded = (int32_t)ex.size();
    prm.excluded_cams = ex.data();
    if (aar_initializer_run(dts, cam_c.data(), (int32_t)cam_c.size(), &prm, &data_)) throw std::runtime_error(aar_last_error());
}
Initializer::~Initializer() { aar_dataset_free(data_); }
aar_dataset *Initializer::release() {
    aar_dataset *d = data_;
    data_ = nullptr;
    return d;
}
aar_detections *Initializer::read_detections_file(std::string path, const std::vector<int> &subseqs) {
    aar_detections *d = nullptr;
    std::vector<int32_t> ss(subseqs.begin(), subseqs.end());
    if (aar_detections_read(path.c_str(), ss.data(), (int32_t)ss.size(), &d)) throw std::runtime_error(aar_last_error());
    return d;
}
std::set<int> Initializer::get_marker_ids() { return data_ ? std::set<int>(data_->marker_ids, data_->marker_ids + data_->num_markers) : std::set<int>(); }
std::set<int> Initializer::get_cam_ids() { return data_ ? std::set<int>(data_->cam_ids, data_->cam_ids + data_->num_cams) : std::set<int>(); }
int Initializer::get_root_cam() { return data_ ? data_->cam_ids[data_->root_cam] : -1; }
int Initializer::get_root_marker() { return data_ ? data_->marker_ids[data_->root_marker] : -1; }
double Initializer::get_marker_size() { return data_ ? data_->marker_size : 0; }
std::map<int, Mat44> Initializer::get_transforms_to_root_cam() {
    std::map<int, Mat44> r;
    if (!data_) return r;
    PoseLayout L;
    L.C = data_->num_cams; L.M = data_->num_markers; L.F = data_->num_frames; L.rc = data_->root_cam; L.rm = data_->root_marker;
    for (int c = 0; c < L.C; c++)
        r[data_->cam_ids[c]] = to44(c == L.rc ? Rigid::identity() : pose_to_rigid(data_->x_full + L.full_cam0() + 6LL * L.cam_slot(c)));
    return r;
}
std::map<int, Mat44> Initializer::get_transforms_to_root_marker() {
    std::map<int, Mat44> r;
    if (!data_) return r;
    PoseLayout L;
    L.C = data_->num_cams; L.M = data_->num_markers; L.F = data_->num_frames; L.rc = data_->root_cam; L.rm = data_->root_marker;
    for (int m = 0; m < L.M; m++)
        r[data_->marker_ids[m]] = to44(m == L.rm ? Rigid::identity() : pose_to_rigid(data_->x_full + L.full_mk0() + 6LL * L.mk_slot(m)));
    return r;
}
std::map<int, Mat44> Initializer::get_object_transforms() {
    std::map<int, Mat44> r;
    if (!data_) return r;
    PoseLayout L;
    L.C = data_->num_cams; L.M = data_->num_markers; L.F = data_->num_frames; L.rc = data_->root_cam; L.rm = data_->root_marker;
    for (int f = 0; f < L.F; f++) r[data_->frame_ids[f]] = to44(pose_to_rigid(data_->x_full + L.full_fr0() + 6LL * f));
    return r;
}

size_t MultiCamMapper::get_root_cam() { return data_ ? (size_t)data_->cam_ids[data_->root_cam] : 0; }
size_t MultiCamMapper::get_root_marker() { return data_ ? (size_t)data_->marker_ids[data_->root_marker] : 0; }
double MultiCamMapper::get_marker_size() { return data_ ? data_->marker_size : 0; }
void MultiCamMapper::remove_distortions() {
    if (!data_) return;
    aar_dataset *d = data_;
    for (int c = 0; c < d->num_cams; c++) {
        std::vector<int64_t> idx;
        for (int64_t o = 0; o < d->num_obs; o++)
            if (d->obs_cam[o] == c) idx.push_back(o);
        if (idx.empty()) continue;
        std::vector<float> pts(8 * idx.size());
        for (size_t k = 0; k < idx.size(); k++) memcpy(&pts[8 * k], d->obs_uv + 8 * idx[k], 8 * sizeof(float));
        const bool full = (int)cam_models_.size() == d->num_cams;   // built from calibrations: all of distortion_coefficients
        if (aar_undistort_points(d->cam_mats + 9 * c, full ? cam_models_[c].dist : d->dist_coeffs + 5 * c, full ? cam_models_[c].n_dist : 5,
                                 (int64_t)(4 * idx.size()), pts.data(), pts.data(), device_id))
            throw std::runtime_error(aar_last_error());
        for (size_t k = 0; k < idx.size(); k++) memcpy(d->obs_uv + 8 * idx[k], &pts[8 * k], 8 * sizeof(float));
    }
    drop_problem();   // the device copy of the observations is stale now
}

std::vector<std::array<int, 2>> MultiCamMapper::get_image_sizes() {
    std::vector<std::array<int, 2>> r;
    if (data_)
        for (int c = 0; c < data_->num_cams; c++) r.push_back({data_->image_sizes[2 * c], data_->image_sizes[2 * c + 1]});
    return r;
}

}  // namespace aar
