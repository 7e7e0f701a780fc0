// Host-side mirror of the reference's MultiCamMapper for the find_solution path
// (libs/multicam_mapper.h:14-83): same method names, argument meaning and error behaviour, with
// cv::Mat replaced by plain 4x4 row-major arrays and Eigen::VectorXd by std::vector<double>, because
// neither OpenCV nor Eigen exists on the GPU box.  Every numeric method forwards to the C ABI of
// include/aar.h (HIP kernels); there is no CPU arithmetic path behind this class.
#pragma once
#include <array>
#include <cstring>
#include <functional>
#include <map>
#include <stdexcept>
#include <type_traits>
#include <set>
#include <string>
#include <vector>

#include "../../include/aar.h"

namespace aar {

class MultiCamMapper;

// Stand-in for Eigen::SparseMatrix<T> in the signature of a Jacobian function (libs/sparselevmarq.h:63).  The Jacobian of the
// accelerated path never exists as a host matrix -- its blocks are accumulated into J^T J on the device -- so for
// MultiCamMapper's own functions the type only keeps reference-shaped code compiling.  A caller's OWN Jacobian function (the
// host loop of detail::HostLevMarq) fills it entry by entry: coordinate form, duplicates are added up, as Eigen's
// setFromTriplets does.
template <typename T>
struct Triplet {   // Eigen::Triplet's interface
    Triplet() {}
    Triplet(int64_t r, int64_t c, T v) : r_(r), c_(c), v_(v) {}
    int64_t row() const { return r_; }
    int64_t col() const { return c_; }
    T value() const { return v_; }
    int64_t r_ = 0, c_ = 0;
    T v_ = 0;
};
template <typename T>
struct SparseJacobian {
    int64_t rows = 0, cols = 0;
    std::vector<int64_t> row, col;
    std::vector<T> val;
    void resize(int64_t r, int64_t c) { rows = r; cols = c; setZero(); }
    void setZero() { row.clear(); col.clear(); val.clear(); }
    T &insert(int64_t r, int64_t c) { row.push_back(r); col.push_back(c); val.push_back(T(0)); return val.back(); }
    template <class It>
    void setFromTriplets(It begin, It end) { setZero(); for (It t = begin; t != end; ++t) insert(t->row(), t->col()) = t->value(); }
    int64_t nonZeros() const { return (int64_t)val.size(); }
};

namespace detail {
// How the solver mirror recognises the evaluation functions it can run.  The reference's callers hand the solver
// std::bind(&MultiCamMapper::error_function, this, _1, _2) and friends (libs/multicam_mapper.cpp:426,441), which reach the solver
// type-erased.  The mirror calls such a callable ONCE with the probe below raised; the evaluation members of aar::MultiCamMapper
// (error_function, jacobian_function, error_function_tracking) then answer with who they are instead of computing, and the
// solver dispatches to the device path of that mapper (aar_lm_solve / aar_lm_step / aar_track).  A callable that does not
// answer is a host function: the solver throws std::logic_error -- there is no CPU loop behind this class.
enum EvalKind { EVAL_NONE = 0, EVAL_ERROR_FUNCTION = 1, EVAL_JACOBIAN_FUNCTION = 2, EVAL_ERROR_FUNCTION_TRACKING = 3 };
struct EvalId {
    MultiCamMapper *owner = nullptr;
    int kind = EVAL_NONE;
};
struct EvalProbe {
    bool active = false;
    EvalId id;
};
EvalProbe &eval_probe();   // one per thread
// The LM loop of libs/sparselevmarq.h for evaluation functions that live on the host (host_levmarq.cpp): what SparseLevMarq<double> runs when its callables are
// not MultiCamMapper's own.  Dense normal equations, LDL^T without pivoting; same init / step / solve rules, exits and quirks as the reference.
class HostLevMarq {
   public:
    typedef std::vector<double> eVector;
    typedef std::function<void(const eVector &, eVector &)> F;
    typedef std::function<void(const eVector &, SparseJacobian<double> &)> FJ;
    struct Prm {
        int maxIters = 100;
        double minError = 1e-5, min_step_error_diff = 0, min_average_step_error_diff = 0.001, tau = 1, der_epsilon = 1e-3;
        bool verbose = false;
    };
    void init(const eVector &z, const F &f);                                     // libs/sparselevmarq.h:238-249
    bool step(const F &f, const FJ &fJ, const Prm &prm);                         // :349-430
    double solve(eVector &z, const F &f, const FJ &fJ, const Prm &prm, const std::function<void(const eVector &)> &step_cb,
                 const std::function<bool(const eVector &)> &stop_fn);            // :440-472
    static void central_differences(const eVector &z, SparseJacobian<double> &J, const F &f, double der_epsilon);   // :193-214
    eVector curr_z, x;            // current point; the residual vector of the LAST evaluation (the reference's x64)
    double currErr = 0, prevErr = 0, mu = -1, v = 2, last_gain = 0;
    int last_tries = 0, iterations = 0, exit_code = 0;   // exit_code: the reference's mustExit (1 minError, 2 small step / rejected, 3 error grew; 0: maxIters or a stop function)
    bool active = false;          // init() has run on the host path
    SparseJacobian<double> J;
};
// the mapper's device problem (created on demand) and its full parameter vector for the current Config; throws on failure
aar_problem *bind_problem(MultiCamMapper *owner, std::vector<double> &x_full);
aar_problem *current_problem(const MultiCamMapper *owner);   // what the mapper holds right now (nullptr after a Config change)
// MultiCamMapper::track() behind solve(z, error_function_tracking): every frame's own LM on the device; returns the summed final error
double solve_tracking(MultiCamMapper *owner, std::vector<double> &z);
}  // namespace detail

// ucoslam::SparseLevMarq<T> (libs/sparselevmarq.h:26-141) over the C ABI: Params with the reference's field names, setParams,
// solve / init / step with the reference's signatures (:80,88,95-96,118), getCurrentSolution, setStepCallBackFunc, setStopFunction.
// Evaluation functions that are aar::MultiCamMapper's own (see detail::EvalProbe) run on the GPU: error_function + jacobian_function as
// the HIP kernels of that mapper's device problem, error_function_tracking as aar_track; there solve(z, f) -- "automatic Jacobian" in the
// reference, central differences -- uses the same analytic device Jacobian.  Any OTHER callable is a host function: the solver then is the
// reference's general sparse LM on the host (detail::HostLevMarq, host_levmarq.cpp: same rules, exits and quirks; central differences for
// solve(z, f)), as SURVEY.md section 8b keeps it for API compatibility.  The probe costs a host function one extra evaluation per solve / init.
// z is the reference's parameter vector for the problem's Config (mats2eVec order).
template <typename T>
class SparseLevMarq {
    static_assert(std::is_same<T, double>::value, "the accelerated path computes in double, as the reference's instantiation does");

   public:
    struct Params {
        Params() {}
        // (sic) the reference's constructor stores _min_step_error_diff in BOTH step fields (libs/sparselevmarq.h:32-39)
        Params(int _maxIters, T _minError, T _min_step_error_diff = 0, T /*_min_average_step_error_diff*/ = 0.001, T _tau = 1, T _der_epsilon = 1e-3) {
            maxIters = _maxIters;
            minError = _minError;
            min_step_error_diff = _min_step_error_diff;
            min_average_step_error_diff = _min_step_error_diff;
            tau = _tau;
            der_epsilon = _der_epsilon;
        }
        int maxIters = 100;
        T minError = 1e-5;
        T min_step_error_diff = 0;
        T min_average_step_error_diff = 0.001;
        T tau = 1;
        T der_epsilon = 1e-3;  // the host loop's central differences (the device Jacobian is analytic)
        bool cal_dev_parallel = true;
        bool use_omp = true;
        bool verbose = false;
    };
    typedef std::vector<T> eVector;
    typedef std::function<void(const eVector &, eVector &)> F_z_x;
    typedef std::function<void(const eVector &, SparseJacobian<T> &)> F_z_J;

    // the device problem whose kernels stand for f and J; x_full supplies the groups the Config keeps fixed.  The overloads that
    // take the evaluation functions attach by themselves; this is for callers that hold an aar_problem of their own.
    void attach(aar_problem *problem, const double *x_full) {
        problem_ = problem;
        x_.assign(x_full, x_full + aar_problem_full_len(problem));
    }
    void detach() { problem_ = nullptr; }   // the problem is about to be destroyed
    void setParams(int maxIters, T minError, T min_step_error_diff = 0, T tau = 1, T der_epsilon = 1e-3) {   // :259-266
        _params.maxIters = maxIters;
        _params.minError = minError;
        _params.min_step_error_diff = min_step_error_diff;
        _params.tau = tau;
        _params.der_epsilon = der_epsilon;
    }
    void setParams(const Params &p) { _params = p; }

    T solve(eVector &z, F_z_x f_z_x, F_z_J f_J) {   // :80, :440-472
        if (bind_eval(z, f_z_x, &f_J) == detail::EVAL_NONE) return host_.solve(z, f_z_x, f_J, host_prm(), step_cb_, stop_fn_);
        return solve(z);
    }
    T solve(eVector &z, F_z_x f_z_x) {   // :118, :474-477
        const int kind = bind_eval(z, f_z_x, nullptr);
        if (kind == detail::EVAL_NONE) return host_.solve(z, f_z_x, numeric_jacobian(f_z_x), host_prm(), step_cb_, stop_fn_);
        if (kind == detail::EVAL_ERROR_FUNCTION_TRACKING) return detail::solve_tracking(owner_, z);
        return solve(z);
    }
    void init(eVector &z, F_z_x f_z_x) {   // :88, :238-249
        const int kind = bind_eval(z, f_z_x, nullptr);
        if (kind == detail::EVAL_NONE) { host_.init(z, f_z_x); return; }
        if (kind != detail::EVAL_ERROR_FUNCTION) throw std::logic_error("SparseLevMarq::init: the step-by-step mode runs MultiCamMapper::error_function or a host function");
        host_.active = false;
        init(z);
    }
    bool step(F_z_x f_z_x, F_z_J f_J) {   // :95, :349-430
        if (host_.active) return host_.step(f_z_x, f_J, host_prm());   // (the functions init() was given: a host loop cannot tell, the reference does not check either)
        check_eval(f_z_x, &f_J);
        return step();
    }
    bool step(F_z_x f_z_x) {   // :96, :250-256
        if (host_.active) return host_.step(f_z_x, numeric_jacobian(f_z_x), host_prm());
        check_eval(f_z_x, nullptr);
        return step();
    }

    // the same entry points for a problem attached with attach(): no evaluation function to name
    T solve(eVector &z) {   // :440-472
        need();
        install();
        if (aar_problem_merge_z(problem_, z.data(), x_.data())) fail();
        const aar_lm_params p = c_params();
        std::memset(&report, 0, sizeof report);
        if (aar_lm_solve(problem_, x_.data(), &p, &report)) fail();
        if (aar_problem_extract_z(problem_, x_.data(), z.data())) fail();
        return report.final_err;
    }
    void init(eVector &z) {   // :238-249
        need();
        if (aar_problem_merge_z(problem_, z.data(), x_.data())) fail();
        const aar_lm_params p = c_params();
        if (aar_lm_init(problem_, x_.data(), &p)) fail();
    }
    bool step() {   // :349-430; the step callback is solve()'s business in the reference too
        need();
        aar_lm_iter it;
        if (aar_lm_step(problem_, &it)) fail();
        last_iter = it;
        return it.accepted != 0;
    }
    T getCurrentSolution(eVector &z) {   // :432-437
        if (host_.active) { z = host_.curr_z; return host_.currErr; }
        need();
        double err = 0;
        if (aar_lm_get_solution(problem_, x_.data(), &err)) fail();
        z.resize((size_t)aar_problem_num_vars(problem_));
        if (aar_problem_extract_z(problem_, x_.data(), z.data())) fail();
        return err;
    }
    // needs_z = false spares the device -> host copy of curr_z for callbacks that ignore their argument (it is then empty)
    void setStepCallBackFunc(std::function<void(const eVector &)> callback, bool needs_z = true) { step_cb_ = callback; step_needs_z_ = needs_z; }
    void setStopFunction(std::function<bool(const eVector &)> stop_function) { stop_fn_ = stop_function; }

    Params _params;
    aar_lm_report report;     // of the last solve() on the device path
    aar_lm_iter last_iter;    // of the last step() on the device path
    const detail::HostLevMarq &host_state() const { return host_; }   // of the host path: mu, errors, iterations, the reference's exit code

   private:
    void need() const { if (!problem_) throw std::runtime_error("SparseLevMarq: no problem attached"); }
    [[noreturn]] static void fail() { throw std::runtime_error(aar_last_error()); }
    // call the callable once with the probe raised: who is it?
    template <class Fn, class Out>
    static detail::EvalId identify(const Fn &fn, const eVector &z) {
        if (!fn) throw std::logic_error("SparseLevMarq: empty evaluation function");
        detail::EvalProbe &p = detail::eval_probe();
        p.active = true;
        p.id = detail::EvalId();
        Out out;
        try { fn(z, out); } catch (...) { p.active = false; throw; }
        p.active = false;
        return p.id;   // kind EVAL_NONE: nobody answered -- a host function
    }
    static detail::EvalId identify_pair(const eVector &z, const F_z_x &f, const F_z_J *J) {
        const detail::EvalId idf = identify<F_z_x, eVector>(f, z);
        if (idf.kind == detail::EVAL_JACOBIAN_FUNCTION) throw std::logic_error("SparseLevMarq: a Jacobian function was passed as the error function");
        if (J) {
            const detail::EvalId idj = identify<F_z_J, SparseJacobian<T>>(*J, z);
            if (idf.kind == detail::EVAL_NONE && idj.kind == detail::EVAL_NONE) return idf;   // both on the host
            if (idj.kind != detail::EVAL_JACOBIAN_FUNCTION || idj.owner != idf.owner || idf.kind != detail::EVAL_ERROR_FUNCTION)
                throw std::logic_error("SparseLevMarq: error and Jacobian function must be error_function and jacobian_function of the same MultiCamMapper, or both host functions");
        }
        return idf;
    }
    // identify the pair and attach to their mapper's device problem; returns the kind of f
    int bind_eval(const eVector &z, const F_z_x &f, const F_z_J *J) {
        const detail::EvalId id = identify_pair(z, f, J);
        host_.active = false;
        if (id.kind == detail::EVAL_NONE) return id.kind;
        owner_ = id.owner;
        if (id.kind == detail::EVAL_ERROR_FUNCTION) problem_ = detail::bind_problem(owner_, x_);
        return id.kind;
    }
    // step(f, J): the functions must be the ones init() was given
    void check_eval(const F_z_x &f, const F_z_J *J) {
        need();
        const detail::EvalId id = identify_pair(eVector(), f, J);
        if (id.owner != owner_ || id.kind != detail::EVAL_ERROR_FUNCTION) throw std::logic_error("SparseLevMarq::step: not the evaluation functions init() was called with");
        if (detail::current_problem(owner_) != problem_) {   // a Config / data change of the mapper destroyed the problem init() ran on
            problem_ = nullptr;
            throw std::runtime_error("SparseLevMarq::step: the mapper's device problem has changed since init()");
        }
    }
    detail::HostLevMarq::Prm host_prm() const {
        detail::HostLevMarq::Prm p;
        p.maxIters = _params.maxIters; p.minError = _params.minError; p.min_step_error_diff = _params.min_step_error_diff;
        p.min_average_step_error_diff = _params.min_average_step_error_diff; p.tau = _params.tau; p.der_epsilon = _params.der_epsilon; p.verbose = _params.verbose;
        return p;
    }
    F_z_J numeric_jacobian(const F_z_x &f) const {   // libs/sparselevmarq.h:222-228: calcDerivates bound to f
        const double eps = _params.der_epsilon;
        return [f, eps](const eVector &z, SparseJacobian<T> &J) { detail::HostLevMarq::central_differences(z, J, f, eps); };
    }
    aar_lm_params c_params() const {
        aar_lm_params p;
        aar_lm_default_params(&p);
        p.max_iters = _params.maxIters;
        p.min_error = _params.minError;
        p.min_step_error_diff = _params.min_step_error_diff;
        p.min_average_step_error_diff = _params.min_average_step_error_diff;
        p.tau = _params.tau;
        p.verbose = _params.verbose ? 1 : 0;
        return p;
    }
    static void step_tramp(void *ctx, const double *z, int64_t n) {
        SparseLevMarq *self = static_cast<SparseLevMarq *>(ctx);
        if (z) self->zbuf_.assign(z, z + n); else self->zbuf_.clear();
        self->step_cb_(self->zbuf_);
    }
    static int stop_tramp(void *ctx, const double *z, int64_t n) {
        SparseLevMarq *self = static_cast<SparseLevMarq *>(ctx);
        self->zbuf_.assign(z, z + n);
        return self->stop_fn_(self->zbuf_) ? 1 : 0;
    }
    void install() {
        if (aar_lm_set_step_callback(problem_, step_cb_ ? &step_tramp : nullptr, this, step_needs_z_ ? 1 : 0)) fail();
        if (aar_lm_set_stop_function(problem_, stop_fn_ ? &stop_tramp : nullptr, this)) fail();
    }
    aar_problem *problem_ = nullptr;
    detail::HostLevMarq host_;          // the host loop and its state (evaluation functions that are not the mapper's)
    MultiCamMapper *owner_ = nullptr;   // whose evaluation functions the last solve / init named
    std::vector<double> x_;
    eVector zbuf_;
    std::function<void(const eVector &)> step_cb_;
    std::function<bool(const eVector &)> stop_fn_;
    bool step_needs_z_ = true;
};

typedef std::array<double, 16> Mat44;  // row-major 4x4, the reference's CV_64F cv::Mat transforms

// aruco::Marker as the path uses it (3rdparty/aruco/aruco/marker.h:46-58): an id and four image corners x0 y0 .. x3 y3
struct Marker {
    int id = 0;
    float corners[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};
typedef std::map<int, std::map<int, std::vector<Marker>>> FrameCamMarkers;   // frame id -> camera id -> detections, as the reference's fcm

// Mirror of the reference's Initializer (libs/initializer.h:9-43) over aar_initializer_run: the constructor runs
// obtain_pose_estimations + init_transforms (IPPE, candidate sets and votes on the device) and throws std::runtime_error
// on failure; the getters return what the reference's return (ids, id -> 4x4 transform maps).
class Initializer {
   public:
    Initializer(const aar_detections *dts, double marker_s, const std::vector<aar_cam_model> &cam_c,
                const std::set<int> &excluded_cs = std::set<int>(), int device_id = 0);
    ~Initializer();
    Initializer(const Initializer &) = delete;
    Initializer &operator=(const Initializer &) = delete;
    // throws std::runtime_error when the file cannot be opened, as the reference does (libs/initializer.cpp:319-320);
    // free the result with aar_detections_free
    static aar_detections *read_detections_file(std::string path, const std::vector<int> &subseqs = std::vector<int>());
    std::set<int> get_marker_ids();
    std::set<int> get_cam_ids();
    int get_root_cam();
    int get_root_marker();
    std::map<int, Mat44> get_transforms_to_root_cam();
    std::map<int, Mat44> get_transforms_to_root_marker();
    std::map<int, Mat44> get_object_transforms();
    double get_marker_size();
    const aar_dataset *dataset() const { return data_; }
    aar_dataset *release();   // hands the data set to MultiCamMapper(Initializer&)

   private:
    aar_dataset *data_ = nullptr;
};

class MultiCamMapper {
   public:
    typedef SparseLevMarq<double>::eVector eVector;

    struct Config {  // libs/multicam_mapper.h:75-81
        bool optimize_cam_poses = true;
        bool optimize_object_poses = true;
        bool optimize_marker_poses = true;
        bool optimize_cam_intrinsics = true;
    };

    struct MatArrays {  // id -> transform, ascending id (libs/multicam_mapper.h:141-179)
        std::map<int, Mat44> transforms_to_root_cam, transforms_to_root_marker, object_to_global;
    };

    MultiCamMapper();
    // Takes ownership of a data set produced by aar_solution_read / aar_synth_generate: the state the
    // reference's MultiCamMapper(Initializer&) constructor ends with (libs/multicam_mapper.cpp:252-335).
    explicit MultiCamMapper(aar_dataset *dataset);
    explicit MultiCamMapper(Initializer &initializer);   // libs/multicam_mapper.cpp:252-254; takes the initializer's data set
    // libs/multicam_mapper.h:17 / multicam_mapper.cpp:256-259: root ids, id -> 4x4 transform maps (camera -> root camera, marker ->
    // root marker, root marker -> root camera per frame), RAW detections per frame and camera, marker size, calibrations
    // indexed by camera id.  Runs init(): observation order and dropping of unknown cameras / markers as fill_iteration_arrays
    // (:345-377), remove_distortions on the device, "the very initial error" printed -- so it needs a GPU, like the rest.
    MultiCamMapper(size_t root_c, const std::map<int, Mat44> &T_to_root_cam, size_t root_m, const std::map<int, Mat44> &T_to_root_marker,
                   const std::map<int, Mat44> &obj_transforms, const FrameCamMarkers &fcm, float m_size, std::vector<aar_cam_model> &cam_confs);
    void init(size_t root_c, const std::map<int, Mat44> &T_to_root_cam, size_t root_m, const std::map<int, Mat44> &T_to_root_marker,
              const std::map<int, Mat44> &object_poses, const FrameCamMarkers &fcm, float m_size, const std::vector<aar_cam_model> &cam_confs);   // :281-335
    // :272-279, apps/track.cpp:127-131: new frames and RAW detections for a mapper that keeps its cameras, markers and intrinsics
    void init(const std::map<int, Mat44> &object_poses, const FrameCamMarkers &fcm);
    ~MultiCamMapper();
    MultiCamMapper(const MultiCamMapper &) = delete;
    MultiCamMapper &operator=(const MultiCamMapper &) = delete;

    void set_optmize_flag_cam_poses(bool);       // sic: the reference spells it "optmize"
    void set_optmize_flag_marker_poses(bool);
    void set_optmize_flag_object_poses(bool);
    void set_optmize_flag_cam_intrinsics(bool);  // fx cx fy cy (+ the five idle distortion entries) per camera join z (:488-498)
    void set_with_huber(bool);                   // Huber-weighted residual rows + optCallBack's delta schedule (:412-417)
    void set_config(Config &conf);
    size_t get_num_vars(const Config &conf);

    void solve();                                               // libs/multicam_mapper.cpp:419-428
    void track();                                               // :430-443, every frame of the data set at once
    void error_function(const eVector &input, eVector &error);  // :731-737
    // :739-801 (private in the reference; public here so that reference-shaped caller code outside the class can bind it).  The
    // Jacobian of the accelerated path is analytic and never leaves the device: called by the solver mirror's probe it names
    // itself (detail::EvalProbe), called directly it throws std::logic_error.
    void jacobian_function(const eVector &input, SparseJacobian<double> &J);
    // :1021-1051, the residual of track(): recognised by SparseLevMarq::solve(z, f) (-> aar_track); a direct call throws
    // std::logic_error (the per-frame residuals of tracking only exist inside k_track)
    void error_function_tracking(const eVector &input, eVector &error);
    void optCallBack(const eVector &v);                         // :412-417
    bool write_solution_file(std::string path);                 // :1053-1099
    bool read_solution_file(std::string path);                  // :1124-1205
    void write_text_solution_file(std::string text_path);       // :1233-1268
    // :554-578: cv::undistortPoints(corners, K, dist, noArray(), P = K) on every detection, camera by camera, on the device.
    // The reference runs it inside init() on raw detections; `.solution` files already hold undistorted corners, so here it is
    // an explicit call for data sets built from raw `aruco.detections` + calib files.  Throws std::runtime_error on failure.
    void remove_distortions();

    MatArrays get_mat_arrays();
    size_t get_root_cam();     // ids, as in the reference
    size_t get_root_marker();
    double get_marker_size();
    std::vector<std::array<int, 2>> get_image_sizes();

    eVector io_vec;           // packed parameters for the current Config (mats2eVec, :445-461)
    float hubberDelta = 2.5;
    SparseLevMarq<double> solver;                 // libs/multicam_mapper.h:198; solve() drives it as the reference does (:419-428)
    SparseLevMarq<double>::Params solver_params;  // what MultiCamMapper::init installs (:326-330)
    aar_lm_report last_report;                    // iterations, errors and timing of the last solve()
    std::vector<int32_t> track_iterations;        // per frame, after track()
    std::vector<double> track_errors;
    int device_id = 0;
    int residual_mode = AAR_RES_F32;
    // How the damped normal equations are solved (aar_solver_options, include/aar.h) -- the counterpart of configuring the reference's solver object
    // through SparseLevMarq::Params (libs/sparselevmarq.h:30-50), and kept beside them.  AUTO (the default) picks by size: the direct chain for one tile of
    // unknowns (<= 16 cameras + markers); CG on the explicit reduced system (SPCG, two-level preconditioned: block-Jacobi + the groups' rigid-motion modes)
    // wherever it fits (<= 224 cameras + markers); CG through the frame blocks (PCG) from 96 entities on when incidences x (incidences per frame - 30) >= 4e6,
    // and where SPCG does not fit -- inexact LM with ONE pose-grade forcing term (SPCG 3e-4, PCG 5e-3) and an absolute tolerance beside it; DIRECT is the
    // reference's Eigen::SimplicialLDLT step to rounding, for callers who want the reference's every step.  Takes effect when the device problem is (re)built: set it before solve() / track();
    // set_solver_options() drops a problem that exists already.
    struct SolverOptions {
        int solver = AAR_SOLVER_AUTO;     // AAR_SOLVER_AUTO | _DIRECT | _SPCG | _PCG
        bool deterministic = false;       // fixed-order sums: two runs give the same bits
        double pcg_eta = 0.0;             // forcing term of the late LM steps (0: the solver's default)
        int pcg_max_it = 0;               // iteration cap of an inner CG solve (0: the solver's default)
        double pcg_eta_loose = 0.0;       // forcing term of the early LM steps (0: default; <= pcg_eta: no sequence)
        double pcg_eta_switch = 0.0;      // "early" = the last accepted step took more than this share of the error away (0: default 0.01)
        double pcg_abs_tol = 0.0;         // absolute tolerance of an inner solve in pose units, beside the relative one (0: default, SPCG 2e-5 / PCG 5e-5)
    };
    void set_solver_options(const SolverOptions &o);
    const SolverOptions &get_solver_options() const { return solver_options_; }
    aar_solver_stats solver_stats();      // what the problem runs with (AUTO resolved) and what its inner solver has done; builds the problem if need be

    const aar_dataset *dataset() const { return data_; }

   private:
    friend aar_problem *detail::bind_problem(MultiCamMapper *, std::vector<double> &);
    friend aar_problem *detail::current_problem(const MultiCamMapper *);
    friend double detail::solve_tracking(MultiCamMapper *, std::vector<double> &);
    bool probed(int kind);   // true: the solver mirror asked who this function is (and has been told)
    std::vector<double> problem_vector();
    void mats2eVec();
    void eVec2Mats(const eVector &v);
    int ensure_problem();
    void drop_problem();

    void load_frames(const std::map<int, Mat44> &object_poses, const FrameCamMarkers &fcm);
    aar_dataset *data_ = nullptr;
    aar_problem *problem_ = nullptr;
    std::vector<aar_cam_model> cam_models_;   // per camera INDEX, when the mapper was built from calibrations (full distortion vectors)
    Config config_;
    SolverOptions solver_options_;
    bool with_huber_ = false;
};

}  // namespace aar
