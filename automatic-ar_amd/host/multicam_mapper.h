// Host-side mirror of the reference's MultiCamMapper for the find_solution path
// (libs/multicam_mapper.h:14-83): same method names, argument meaning and error behaviour, with
// cv::Mat replaced by plain 4x4 row-major arrays and Eigen::VectorXd by std::vector<double>, because
// neither OpenCV nor Eigen exists on the GPU box.  Every numeric method forwards to the C ABI of
// include/aar.h (HIP kernels); there is no CPU arithmetic path behind this class.
#pragma once
#include <array>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "../../include/aar.h"

namespace aar {

// ucoslam::SparseLevMarq<T>::Params, libs/sparselevmarq.h:30-50 (field names kept)
template <typename T>
struct SparseLevMarq {
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
};

typedef std::array<double, 16> Mat44;  // row-major 4x4, the reference's CV_64F cv::Mat transforms

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
    ~MultiCamMapper();
    MultiCamMapper(const MultiCamMapper &) = delete;
    MultiCamMapper &operator=(const MultiCamMapper &) = delete;

    void set_optmize_flag_cam_poses(bool);       // sic: the reference spells it "optmize"
    void set_optmize_flag_marker_poses(bool);
    void set_optmize_flag_object_poses(bool);
    void set_optmize_flag_cam_intrinsics(bool);  // true is rejected at solve(): outside this path (apps/find_solution.cpp:140)
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
    SparseLevMarq<double>::Params solver_params;  // what MultiCamMapper::init installs (:326-330)
    aar_lm_report last_report;                    // iterations, errors and timing of the last solve()
    std::vector<int32_t> track_iterations;        // per frame, after track()
    std::vector<double> track_errors;
    int device_id = 0;
    int residual_mode = AAR_RES_F32;

    const aar_dataset *dataset() const { return data_; }

   private:
    void mats2eVec();
    void eVec2Mats(const eVector &v);
    int ensure_problem();
    void drop_problem();

    aar_dataset *data_ = nullptr;
    aar_problem *problem_ = nullptr;
    Config config_;
    bool with_huber_ = false;
};

}  // namespace aar
