// aar_find_solution: the find_solution driver for the accelerated path (apps/find_solution.cpp:28-181).
//
//   aar_find_solution <data_folder> <marker_size> [ignored] [-subseqs] [-exclude-cams ...] [-with-huber] [-thresh t]
//
// File contract kept from the reference: reads <folder>/initial<suffix>.solution, writes
// <folder>/final<suffix>.solution and .yaml, prints "The algorithm took: ..." (:45,99-100,162-163,175-177).
// The step that PRODUCES initial.solution in the reference -- Initializer (IPPE votes + MST,
// libs/initializer.cpp) -- is outside this path (SURVEY.md 8f "next" #1): run the reference's find_solution
// once (it writes initial.solution before solve(), :146) or generate a synthetic folder with
//   aar_find_solution --synth <config 1..5> <out_folder>
#include <sys/stat.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <set>
#include <string>

#include "multicam_mapper.h"

using namespace std;

static int print_usage(const char *a0) {
    cout << "Usage: " << a0 << " <data_folder_path> <marker_size> [-subseqs] [-exclude-cams <cam_id> ...] [-with-huber] [-thresh <t>]" << endl;
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
    bool use_subseqs = false, with_huber = false, set_threshold = false;
    double threshold = 2.0;
    set<int> excluded_cams;
    enum ArgFlag { NONE, ExcludeCams, Threshold } arg_flag = NONE;
    for (int i = 4; i < argc; i++) {  // sic: the reference starts at argv[4] (apps/find_solution.cpp:47)
        const string a = argv[i];
        if (a == "-subseqs") use_subseqs = true;
        else if (a == "-exclude-cams") arg_flag = ExcludeCams;
        else if (a == "-with-huber") { with_huber = true; arg_flag = NONE; }
        else if (a == "-thresh") { set_threshold = true; arg_flag = Threshold; }
        else if (arg_flag == ExcludeCams) excluded_cams.insert(stoi(a));
        else if (arg_flag == Threshold) { threshold = stod(a); arg_flag = NONE; }
    }
    string name = "";
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

    aar::MultiCamMapper mcm;
    if (!mcm.read_solution_file(initial_path)) {
        cerr << "No " << initial_path << ": this driver starts from the initial solution the reference's Initializer writes." << endl;
        return 1;
    }
    if (fabs(mcm.get_marker_size() - (double)(float)marker_size) > 1e-9)
        cerr << "warning: marker_size argument " << marker_size << " differs from the solution file's " << mcm.get_marker_size() << endl;
    mcm.solver_params.verbose = true;
    mcm.set_optmize_flag_cam_intrinsics(false);  // apps/find_solution.cpp:140
    if (with_huber) mcm.set_with_huber(true);
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
    cout << "LM iterations: " << r.iterations << "  error " << r.initial_err << " -> " << r.final_err << "  (" << r.iterations / r.solve_seconds
         << " LM it/s in the solver loop)" << endl;
    const int minutes = (int)(d.count() / 60);
    const int seconds = (int)lround(d.count() - minutes * 60);
    cout << "The algorithm took: " << minutes << " minutes " << seconds << " seconds" << endl;
    return 0;
}
