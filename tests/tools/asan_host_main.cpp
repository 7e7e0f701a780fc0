#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/aar.h"
#include "../../automatic-ar_amd/host/init_device.h"
// device entry points stubbed: this binary exercises the HOST code under ASan/UBSan only
namespace aar {
int initdev_create(int32_t, InitDevice **) { return set_error(AAR_ERR_NO_DEVICE, "stub"); }
void initdev_destroy(InitDevice *) {}
int initdev_ippe(InitDevice *, const aar_cam_model *, int32_t, float, int64_t, const float *, const int32_t *, float *, float *, float *) { return -2; }
int initdev_pair_vote(InitDevice *, int, int64_t, const int32_t *, const int32_t *, int64_t, const int64_t *, double, int64_t *, double *, double *) { return -2; }
int initdev_object_vote(InitDevice *, int64_t, const int32_t *, const int32_t *, const int32_t *, int32_t, const double *, int32_t, const double *, int64_t, const int64_t *, double, int64_t *, double *, double *) { return -2; }
}
extern "C" int aar_undistort_points(const double *, const double *, int32_t, int64_t, const float *, float *, int32_t) { return -2; }
int main() {
    aar_synth_desc sd; aar_synth_default(&sd, 3); sd.num_frames = 40;
    aar_dataset *d = nullptr;
    if (aar_synth_generate(&sd, &d)) { puts(aar_last_error()); return 1; }
    if (system("rm -rf /tmp/aar_asan_f && mkdir -p /tmp/aar_asan_f/cam_000 /tmp/aar_asan_f/cam_001")) return 1;
    aar_detections_write("/tmp/aar_asan_f/aruco.detections", d);
    aar_solution_write("/tmp/aar_asan_f/i.solution", d);
    aar_solution_write_yaml("/tmp/aar_asan_f/i.solution.yaml", d);
    aar_dataset *d2 = nullptr;
    if (aar_solution_read("/tmp/aar_asan_f/i.solution", &d2)) { puts(aar_last_error()); return 1; }
    printf("obs %lld %lld\n", (long long)d->num_obs, (long long)d2->num_obs);
    aar_detections *det = nullptr;
    int32_t ss[4] = {2, 5, 10, 12};
    if (aar_detections_read("/tmp/aar_asan_f/aruco.detections", ss, 4, &det)) { puts(aar_last_error()); return 1; }
    printf("det %lld frames %d cams %d\n", (long long)det->num_det, det->num_frames, det->num_cams);
    // truncated file
    { FILE *f = fopen("/tmp/aar_asan_f/aruco.detections", "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
      std::vector<char> b(n); fread(b.data(), 1, n, f); fclose(f);
      for (long cut : {0L, 3L, 8L, 9L, 20L, n / 2, n - 1}) { f = fopen("/tmp/aar_asan_f/t.det", "wb"); fwrite(b.data(), 1, cut, f); fclose(f);
        aar_detections *t = nullptr; int rc = aar_detections_read("/tmp/aar_asan_f/t.det", nullptr, 0, &t); if (!rc) aar_detections_free(t); } }
    const char *yml = "%YAML:1.0\n---\nimage_width: 640\nimage_height: 480\ncamera_matrix: !!opencv-matrix\n   rows: 3\n   cols: 3\n   dt: d\n   data: [ 500., 0., 320., 0., 500., 240., 0., 0., 1. ]\ndistortion_coefficients: !!opencv-matrix\n   rows: 1\n   cols: 5\n   dt: d\n   data: [ 0.1, 0., 0., 0., 0. ]\n";
    for (const char *p : {"/tmp/aar_asan_f/cam_000/calib.yml", "/tmp/aar_asan_f/cam_001/calib.yml"}) { FILE *f = fopen(p, "w"); fputs(yml, f); fclose(f); }
    { FILE *f = fopen("/tmp/aar_asan_f/cam_001/calib.xml", "w"); fputs("<?xml version=\"1.0\"?><opencv_storage><image_width>1</image_width>", f); fclose(f); }   // malformed
    aar_cam_model *cams = nullptr; int32_t nc = 0;
    aar_cam_configs_read("/tmp/aar_asan_f", &cams, &nc);
    printf("cams %d\n", nc);
    aar_init_params ip; aar_init_default_params(&ip);
    aar_dataset *o = nullptr;
    int rc = aar_initializer_run(det, cams, nc, &ip, &o);   // reaches the device stub: NO_DEVICE
    printf("init rc %d (%s)\n", rc, aar_last_error());
    int32_t begin[9]; std::vector<int64_t> per(d->num_frames, 3);
    aar_plan_shards(d->num_frames, per.data(), 8, begin);
    int32_t *sub = nullptr, nsub = 0; { FILE *f = fopen("/tmp/aar_asan_f/subseqs.txt", "w"); fputs("1 2\n7 9 x", f); fclose(f); }
    aar_subseqs_read("/tmp/aar_asan_f/subseqs.txt", &sub, &nsub); printf("subseqs %d\n", nsub); free(sub);
    free(cams); aar_detections_free(det); aar_dataset_free(d); aar_dataset_free(d2);
    return 0;
}
