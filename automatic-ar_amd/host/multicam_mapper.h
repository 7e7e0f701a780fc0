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

// ucoslam::SparseLevMarq<T> (libs/sparselevmarq.h:26-141) over the C ABI: Params with the reference's field names, setParams,
// init / step / getCurrentSolution, solve, setStepCallBackFunc, setStopFunction.  The evaluation functions f (error_function) and
// J (jacobian_function) are the MAPPER'S OWN -- the HIP kernels of the attached aar_problem; solve(z, f, J) with arbitrary host
// callbacks is not offered (a host-callback Jacobian cannot run on the device and a CPU loop would be a fallback path).
// z is the reference's parameter vector for the problem's Config (mats2eVec order).
template <typename T>
class SparseLevMarq {
    static_assert(std::is_same<T, double>::value, "the accelerated path computes in double, as the reference's instantiation does");

   public:
    struct Params {
        int maxIters = 100;
        T minError = 1e-5;
        T min_step_error_diff = 0;
        T min_average_step_error_diff = 0.001;
        T tau = 1;
        T der_epsilon = 1e-3;  // unused: the Jacobian is analytic here
        bool cal_dev_parallel = true;
        bool use_omp = true;
        bool verbose = false;
    };
    typedef std::vector<T> eVector;

    // the device problem whose kernels stand for f and J; x_full supplies the groups the Config keeps fixed
    void attach(aar_problem *problem, const double *x_full) {
        problem_ = problem;
        x_.assign(x_full, x_full + aar_problem_full_len(problem));
    }
    void setParams(const Params &p) { _params = p; }
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
    aar_lm_report report;     // of the last solve()
    aar_lm_iter last_iter;    // of the last step()

   private:
    void need() const { if (!problem_) throw std::runtime_error("SparseLevMarq: no problem attached"); }
    [[noreturn]] static void fail() { throw std::runtime_error(aar_last_error()); }
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

    const aar_dataset *dataset() const { return data_; }

   private:
    void optCallBack(const eVector &v);   // :412-417
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
    bool with_huber_ = false;
};

}  // namespace aar
