// aar_find_solution: the find_solution driver for the accelerated path (apps/find_solution.cpp:28-181).
//
//   aar_find_solution <data_folder> <marker_size> [ignored] [-subseqs] [-exclude-cams ...] [-with-huber] [-thresh t] [-solver direct|spcg|pcg|auto]
//
// File contract kept from the reference (:45,99-100,146-147,162-163,175-177): reads <folder>/aruco.detections and
// <folder>/<cam>/calib.{xml,yml,yaml}, runs the Initializer (IPPE poses, votes, spanning trees -- on the GPU, aar_initializer_run),
// writes <folder>/initial<suffix>.solution(.yaml), solves, writes <folder>/final<suffix>.solution(.yaml) and prints
// "The algorithm took: ...".  With -from-initial, or when the folder holds no calibration, it starts from an existing
// initial<suffix>.solution instead (e.g. one the reference wrote).  A synthetic folder in the reference's formats:
//   aar_find_solution --synth <config 1..5> <out_folder>
#include <sys/stat.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "multicam_mapper.h"

using namespace std;

static int print_usage(const char *a0) {
    cout << "Usage: " << a0 << " <data_folder_path> <marker_size> [ignored] [-subseqs] [-exclude-cams <cam_id> ...] [-with-huber] [-thresh <t>] [-from-initial] [-solver direct|spcg|pcg|auto]" << endl;
    cout << "       " << a0 << " --synth <config 1..5> <out_folder>   (write a synthetic data set in the reference's file formats)" << endl;
    return -1;
}

static int synth(int cfg, const string &folder) {
    aar_synth_desc sd;
    aar_synth_default(&sd, cfg);
    aar_dataset *d = nullptr;
    if (aar_synth_generate(&sd, &d)) { cerr << aar_last_error() << endl; return 1; }
    mkdir(folder.c_str(), 0755);
    int rc = aar_detections_write((folder + "/aruco.detections").c_str(), d);
    for (int c = 0; c < d->num_cams && !rc; c++) {  // one calib.yml per camera slot, cv::FileStorage YAML dialect
        char dir[64];
        snprintf(dir, sizeof dir, "/cam_%03d", d->cam_ids[c]);
        mkdir((folder + dir).c_str(), 0755);
        FILE *f = fopen((folder + dir + "/calib.yml").c_str(), "w");
        if (!f) { rc = 1; break; }
        const double *K = d->cam_mats + 9 * c;
        fprintf(f, "%%YAML:1.0\n---\nimage_width: %d\nimage_height: %d\ncamera_matrix: !!opencv-matrix\n   rows: 3\n   cols: 3\n   dt: d\n   data: [ ",
                d->image_sizes[2 * c], d->image_sizes[2 * c + 1]);
        for (int i = 0; i < 9; i++) fprintf(f, "%.17g%s", K[i], i < 8 ? ", " : " ]\n");
        fprintf(f, "distortion_coefficients: !!opencv-matrix\n   rows: 1\n   cols: 5\n   dt: d\n   data: [ 0., 0., 0., 0., 0. ]\n");
        fclose(f);
    }
    rc |= aar_solution_write((folder + "/initial.solution").c_str(), d);
    rc |= aar_solution_write_yaml((folder + "/initial.solution.yaml").c_str(), d);
    if (rc) cerr << aar_last_error() << endl;
    cout << "wrote " << folder << ": cams=" << d->num_cams << " markers=" << d->num_markers << " frames=" << d->num_frames
         << " marker-observations=" << d->num_obs << endl;
    aar_dataset_free(d);
    return rc ? 1 : 0;
}

int main(int argc, char *argv[]) {
    if (argc >= 4 && string(argv[1]) == "--synth") return synth(atoi(argv[2]), argv[3]);
    if (argc < 3) return print_usage(argv[0]);
    const string folder_path = argv[1];
    const double marker_size = stod(argv[2]);
    bool use_subseqs = false, with_huber = false, set_threshold = false, from_initial = false, tracking_only = false;
    double threshold = 2.0;
    set<int> excluded_cams;
    int solver = AAR_SOLVER_AUTO;   // not an option of the reference: how the damped systems are solved (aar_solver_options); `-solver direct` = the reference's every step
    enum ArgFlag { NONE, ExcludeCams, Threshold, Solver } arg_flag = NONE;
    for (int i = 4; i < argc; i++) {  // sic: the reference starts at argv[4] (apps/find_solution.cpp:47)
        const string a = argv[i];
        if (a == "-subseqs") use_subseqs = true;
        else if (a == "-exclude-cams") arg_flag = ExcludeCams;
        else if (a == "-with-huber") { with_huber = true; arg_flag = NONE; }
        else if (a == "-from-initial") { from_initial = true; arg_flag = NONE; }
        else if (a == "-tracking-only") { tracking_only = true; arg_flag = NONE; }   // names the files only, as in the reference (:54-57,76-77)
        else if (a == "-thresh") { set_threshold = true; arg_flag = Threshold; }
        else if (a == "-solver") arg_flag = Solver;
        else if (arg_flag == Solver) {
            solver = a == "spcg" ? AAR_SOLVER_SPCG : a == "pcg" ? AAR_SOLVER_PCG : a == "auto" ? AAR_SOLVER_AUTO : a == "direct" ? AAR_SOLVER_DIRECT : -1;
            if (solver < 0) return print_usage(argv[0]);
            arg_flag = NONE;
        }
        else if (arg_flag == ExcludeCams) excluded_cams.insert(stoi(a));
        else if (arg_flag == Threshold) { threshold = stod(a); arg_flag = NONE; }
    }
    string name = "";
    if (tracking_only) name += "_tracking_only";
    if (use_subseqs) name += "_subseqs";
    if (with_huber) name += "_with_huber";
    if (!excluded_cams.empty()) {
        name += "_excluded_cams";
        for (int c : excluded_cams) name += "_" + to_string(c);
    }
    if (set_threshold) {
        char dbuf[32];
        snprintf(dbuf, sizeof dbuf, "%.1f", threshold);
        name += "_thresh_" + string(dbuf);
    }
    name += ".solution";
    const string initial_path = folder_path + "/initial" + name, final_path = folder_path + "/final" + name;

    // Initializer (apps/find_solution.cpp:101-113,142-147).  `-thresh` only names the files in the reference (the value never
    // reaches the Initializer, whose threshold stays 2.0, libs/initializer.h:52); kept that way.
    aar_cam_model *cams = nullptr;
    int32_t n_cams = 0;
    if (!from_initial && aar_cam_configs_read(folder_path.c_str(), &cams, &n_cams) != AAR_OK) n_cams = 0;
    aar_dataset *init = nullptr;
    if (!from_initial && n_cams > 0) {
        try {
            vector<int> subseqs;
            if (use_subseqs) {
                int32_t *ss = nullptr, n_sub = 0;
                if (aar_subseqs_read((folder_path + "/subseqs.txt").c_str(), &ss, &n_sub)) throw runtime_error(aar_last_error());
                subseqs.assign(ss, ss + n_sub);
                free(ss);
            }
            aar_detections *detections = aar::Initializer::read_detections_file(folder_path + "/aruco.detections", subseqs);
            const long long n_det = detections->num_det;
            const auto t0 = chrono::system_clock::now();
            try {
                aar::Initializer initializer(detections, marker_size, vector<aar_cam_model>(cams, cams + n_cams), excluded_cams);
                init = initializer.release();
            } catch (...) {
                aar_detections_free(detections);
                throw;
            }
            aar_detections_free(detections);
            const chrono::duration<double> di = chrono::system_clock::now() - t0;
            cout << "Initializer: " << n_det << " detections -> " << init->num_cams << " cameras, " << init->num_markers << " markers, "
                 << init->num_frames << " frames in " << di.count() << " s" << endl;
        } catch (const exception &e) {
            cerr << "Initializer failed: " << e.what() << endl;
            free(cams);
            return 1;
        }
    }
    free(cams);
    aar::MultiCamMapper mcm(init);
    if (init) {
        mcm.write_solution_file(initial_path);
        mcm.write_text_solution_file(initial_path + ".yaml");
    } else if (!mcm.read_solution_file(initial_path)) {
        cerr << "No calibration folders under " << folder_path << " and no " << initial_path << endl;
        return 1;
    }
    if (fabs(mcm.get_marker_size() - (double)(float)marker_size) > 1e-9)
        cerr << "warning: marker_size argument " << marker_size << " differs from the solution file's " << mcm.get_marker_size() << endl;
    mcm.solver_params.verbose = true;
    mcm.set_optmize_flag_cam_intrinsics(false);  // apps/find_solution.cpp:140
    if (with_huber) mcm.set_with_huber(true);
    {
        aar::MultiCamMapper::SolverOptions so;
        so.solver = solver;
        mcm.set_solver_options(so);
    }
    const auto start = chrono::system_clock::now();
    try {
        mcm.solve();
    } catch (const exception &e) {
        cerr << "solve failed: " << e.what() << endl;
        return 2;
    }
    const chrono::duration<double> d = chrono::system_clock::now() - start;
    mcm.write_solution_file(final_path);
    mcm.write_text_solution_file(final_path + ".yaml");
    const aar_lm_report &r = mcm.last_report;
    {
        const aar_solver_stats st = mcm.solver_stats();
        const char *names[] = {"direct", "pcg", "spcg", "auto"};
        cout << "solver: " << names[st.solver & 3] << ", " << st.total_iterations << " CG iterations in " << st.solves << " damped solves, " << st.fallbacks
             << " redone by the direct chain" << endl;
    }
    cout << "LM iterations: " << r.iterations << "  error " << r.initial_err << " -> " << r.final_err << "  (" << r.iterations / r.solve_seconds
         << " LM it/s in the solver loop)" << endl;
    const int minutes = (int)(d.count() / 60);
    const int seconds = (int)lround(d.count() - minutes * 60);
    cout << "The algorithm took: " << minutes << " minutes " << seconds << " seconds" << endl;
    return 0;
}
