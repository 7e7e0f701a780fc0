// File formats of the find_solution path.  LP64, little-endian, no padding.
//   .solution        MultiCamMapper::write_solution_file / read_solution_file, libs/multicam_mapper.cpp:1053-1099,1124-1205
//   .solution.yaml   MultiCamMapper::write_text_solution_file, libs/multicam_mapper.cpp:1233-1268
//   aruco.detections MultiCamMapper::write_detections_file, libs/multicam_mapper.cpp:216-237
// Marker record = ArucoSerdes::serialize_marker, libs/aruco_serdes.cpp:9-24: int32 id + 4 x (float x, float y).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "internal.h"
#include "se3.h"

namespace {

using namespace aar;

struct Writer {
    FILE *f;
    template <class T>
    void put(const T &v) { fwrite(&v, sizeof(T), 1, f); }
};

struct Reader {
    FILE *f;
    bool ok = true;
    template <class T>
    T get() {
        T v{};
        if (fread(&v, sizeof(T), 1, f) != 1) ok = false;
        return v;
    }
};

void write_marker(Writer &w, int32_t id, const float *uv) {
    w.put(id);
    for (int k = 0; k < 8; k++) w.put(uv[k]);
}

// cv::FileStorage number formatting: integral doubles as "%d.", everything else "%.16e"
std::string fs_double(double v) {
    char buf[64];
    if (std::isfinite(v) && v == std::floor(v) && std::fabs(v) < 1e9)
        snprintf(buf, sizeof buf, "%d.", (int)v);
    else
        snprintf(buf, sizeof buf, "%.16e", v);
    return buf;
}

void yaml_transform(FILE *f, const char *id_key, int id, const Rigid &T) {
    fprintf(f, "   - { %s:%d, transform: !!opencv-matrix { rows:4, cols:4, dt:d, data:[ ", id_key, id);
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            double v = r < 3 ? (c < 3 ? T.R[r * 3 + c] : T.t[r]) : (c == 3 ? 1.0 : 0.0);
            fprintf(f, "%s%s", fs_double(v).c_str(), (r == 3 && c == 3) ? " " : ", ");
            if ((r * 4 + c) % 4 == 3 && !(r == 3 && c == 3)) fprintf(f, "\n       ");
        }
    fprintf(f, "] } }\n");
}

}  // namespace

extern "C" {

int aar_solution_write(const char *path, const aar_dataset *d) {
    if (!path || !d) return set_error(AAR_ERR_INVALID, "aar_solution_write: null argument");
    FILE *f = fopen(path, "wb");
    if (!f) return set_error(AAR_ERR_IO, "Could not open a file in: %s for writing.", path);
    Writer w{f};
    const int C = d->num_cams, M = d->num_markers, F = d->num_frames;
    w.put<size_t>((size_t)C);
    for (int c = 0; c < C; c++) w.put<int32_t>(d->cam_ids[c]);
    w.put<size_t>((size_t)d->cam_ids[d->root_cam]);  // root_cam is an id in the file
    for (int c = 0; c < C; c++) { w.put<int32_t>(d->image_sizes[2 * c]); w.put<int32_t>(d->image_sizes[2 * c + 1]); }
    w.put<size_t>((size_t)M);
    for (int m = 0; m < M; m++) w.put<int32_t>(d->marker_ids[m]);
    w.put<size_t>((size_t)d->marker_ids[d->root_marker]);
    w.put<double>(d->marker_size);
    w.put<size_t>((size_t)F);
    for (int i = 0; i < F; i++) w.put<int32_t>(d->frame_ids[i]);
    // Always the full default-Config vector (:1085-1089).  The stored poses are Rodrigues(mat) of the
    // matrices (mats2eVec at :1054,1086), i.e. the vector after a vec -> mat -> vec round trip.
    const int64_t len = aar_dataset_full_len(d);
    for (int64_t i = 0; i < len; i += 6) {
        double v[6];
        rigid_to_pose(pose_to_rigid(d->x_full + i), v);
        for (int k = 0; k < 6; k++) w.put<double>(v[k]);
    }
    for (int c = 0; c < C; c++) {  // fill_io_vec_cam_intrinsics, :488-498
        const double *K = d->cam_mats + 9 * c;
        w.put<double>(K[0]); w.put<double>(K[2]); w.put<double>(K[4]); w.put<double>(K[5]);
        for (int j = 0; j < 5; j++) w.put<double>(d->dist_coeffs[5 * c + j]);
    }
    // serialize_frame_cam_markers, :1030-1051 (observations are frame-major, then camera, detection order)
    std::vector<int64_t> fstart(F + 1, 0);
    for (int64_t o = 0; o < d->num_obs; o++) fstart[d->obs_frame[o] + 1]++;
    for (int i = 0; i < F; i++) fstart[i + 1] += fstart[i];
    w.put<size_t>((size_t)F);
    for (int i = 0; i < F; i++) {
        w.put<int32_t>(d->frame_ids[i]);
        // one record per camera of the data set, empty ones included: that is what the reference writes after its Initializer
        // (read_detections_file gives every frame an entry per camera slot, libs/initializer.cpp:333-347), and what its reader
        // needs -- it takes the c-th record of a frame for camera id c (:1117-1119)
        // (a frame's observations of one camera need not be contiguous in the caller's arrays: gather them, in order)
        std::vector<std::vector<int64_t>> of_cam(C);
        for (int64_t o = fstart[i]; o < fstart[i + 1]; o++) of_cam[d->obs_cam[o]].push_back(o);
        w.put<size_t>((size_t)C);
        for (int c = 0; c < C; c++) {
            w.put<int32_t>(d->cam_ids[c]);
            w.put<size_t>(of_cam[c].size());
            for (int64_t o : of_cam[c]) write_marker(w, d->marker_ids[d->obs_marker[o]], d->obs_uv + 8 * o);
        }
    }
    w.put<bool>(d->optimize_cam_poses != 0);
    w.put<bool>(d->optimize_marker_poses != 0);
    w.put<bool>(d->optimize_object_poses != 0);
    w.put<bool>(d->optimize_cam_intrinsics != 0);
    const bool good = !ferror(f);
    fclose(f);
    return good ? AAR_OK : set_error(AAR_ERR_IO, "write error on %s", path);
}

int aar_solution_read(const char *path, aar_dataset **out) { return aar_solution_read_ex(path, 0, out); }

int aar_solution_read_ex(const char *path, int32_t read_flags, aar_dataset **out) {
    if (!path || !out) return set_error(AAR_ERR_INVALID, "aar_solution_read: null argument");
    const bool ref_index = (read_flags & AAR_SOLUTION_REFERENCE_INDEXING) != 0;
    FILE *f = fopen(path, "rb");
    if (!f) return set_error(AAR_ERR_IO, "Could not open a file in: %s for reading.", path);
    Reader r{f};
    const size_t C = r.get<size_t>();
    if (!r.ok || C == 0 || C > (1u << 20)) { fclose(f); return set_error(AAR_ERR_IO, "%s: bad camera count", path); }
    std::vector<int32_t> cam_ids(C);
    for (auto &v : cam_ids) v = r.get<int32_t>();
    const size_t root_cam_id = r.get<size_t>();
    std::vector<int32_t> sizes(2 * C);
    for (auto &v : sizes) v = r.get<int32_t>();
    const size_t M = r.get<size_t>();
    if (!r.ok || M == 0 || M > (1u << 24)) { fclose(f); return set_error(AAR_ERR_IO, "%s: bad marker count", path); }
    std::vector<int32_t> marker_ids(M);
    for (auto &v : marker_ids) v = r.get<int32_t>();
    const size_t root_marker_id = r.get<size_t>();
    const double marker_size = r.get<double>();
    const size_t F = r.get<size_t>();
    if (!r.ok || F > (1u << 28)) { fclose(f); return set_error(AAR_ERR_IO, "%s: bad frame count", path); }
    std::vector<int32_t> frame_ids(F);
    for (auto &v : frame_ids) v = r.get<int32_t>();
    const int64_t len = 6LL * (C - 1) + 6LL * (M - 1) + 6LL * F;
    std::vector<double> vec(len + 9 * C);
    for (auto &v : vec) v = r.get<double>();
    if (!r.ok) { fclose(f); return set_error(AAR_ERR_IO, "%s: truncated header", path); }

    std::map<int, int> cam_index, marker_index, frame_index;
    for (size_t i = 0; i < C; i++) cam_index[cam_ids[i]] = (int)i;
    for (size_t i = 0; i < M; i++) marker_index[marker_ids[i]] = (int)i;
    for (size_t i = 0; i < F; i++) frame_index[frame_ids[i]] = (int)i;

    // deserialize_frame_cam_markers (:1101-1122).  The reference re-indexes frames and cameras by loop
    // counter and ignores the stored ids (:1117-1119); by default this reader honours the ids,
    // which is identical whenever ids are 0..n-1 and every frame lists every camera, and correct otherwise.
    // AAR_SOLUTION_REFERENCE_INDEXING reproduces the reference: the f-th frame record is filed under frame id f and the
    // c-th camera record of a frame under camera id c, whatever ids the file stores; fill_iteration_arrays (:345-377) then
    // drops the observations of a camera id it does not know, and an unknown frame id is what the reference would hand to
    // MatArray::operator[] (libs/multicam_mapper.h:118-123: a silent std::map insertion / an out_of_range) -- an error here.
    struct O { int f, c, m; float uv[8]; };
    std::vector<O> obs;
    const size_t num_f = r.get<size_t>();
    for (size_t i = 0; i < num_f && r.ok; i++) {
        int32_t fid = r.get<int32_t>();
        if (ref_index) fid = (int32_t)i;
        const size_t num_c = r.get<size_t>();
        for (size_t j = 0; j < num_c && r.ok; j++) {
            int32_t cid = r.get<int32_t>();
            if (ref_index) cid = (int32_t)j;
            const size_t num_m = r.get<size_t>();
            for (size_t k = 0; k < num_m && r.ok; k++) {
                O o;
                const int32_t mid = r.get<int32_t>();
                for (int q = 0; q < 8; q++) o.uv[q] = r.get<float>();
                auto fi = frame_index.find(fid);
                auto ci = cam_index.find(cid);
                auto mi = marker_index.find(mid);
                if (ref_index && fi == frame_index.end()) {
                    fclose(f);
                    return set_error(AAR_ERR_INVALID, "%s: with the reference's re-indexing frame record %zu becomes frame id %d, which the file does not list", path, i, fid);
                }
                // fill_iteration_arrays drops observations of unknown cameras / markers (:356-367)
                if (fi == frame_index.end() || ci == cam_index.end() || mi == marker_index.end()) continue;
                o.f = fi->second; o.c = ci->second; o.m = mi->second;
                obs.push_back(o);
            }
        }
    }
    bool flags[4] = {true, true, true, false};
    for (int i = 0; i < 4; i++) flags[i] = r.get<bool>();
    const bool ok = r.ok;
    fclose(f);
    if (!ok) return set_error(AAR_ERR_IO, "%s: truncated body", path);
    if (!cam_index.count((int)root_cam_id) || !marker_index.count((int)root_marker_id))
        return set_error(AAR_ERR_IO, "%s: root camera / marker id not in the id lists", path);
    // the file's map order is ascending id; keep observations frame-major in index order
    for (size_t i = 1; i < obs.size(); i++)
        if (obs[i].f < obs[i - 1].f) return set_error(AAR_ERR_IO, "%s: frames are not in ascending id order", path);

    aar_dataset *d = dataset_alloc((int)C, (int)M, (int)F, (int64_t)obs.size(), false);
    memcpy(d->cam_ids, cam_ids.data(), sizeof(int32_t) * C);
    memcpy(d->marker_ids, marker_ids.data(), sizeof(int32_t) * M);
    if (F) memcpy(d->frame_ids, frame_ids.data(), sizeof(int32_t) * F);
    memcpy(d->image_sizes, sizes.data(), sizeof(int32_t) * 2 * C);
    d->root_cam = cam_index[(int)root_cam_id];
    d->root_marker = marker_index[(int)root_marker_id];
    d->marker_size = marker_size;
    memcpy(d->x_full, vec.data(), sizeof(double) * len);
    for (size_t c = 0; c < C; c++) {  // intrinsics_vec2mats, :580-593
        const double *q = vec.data() + len + 9 * c;
        double *K = d->cam_mats + 9 * c;
        K[0] = q[0]; K[1] = 0; K[2] = q[1]; K[3] = 0; K[4] = q[2]; K[5] = q[3]; K[6] = 0; K[7] = 0; K[8] = 1;
        for (int j = 0; j < 5; j++) d->dist_coeffs[5 * c + j] = q[4 + j];
    }
    for (size_t i = 0; i < obs.size(); i++) {
        d->obs_frame[i] = obs[i].f; d->obs_cam[i] = obs[i].c; d->obs_marker[i] = obs[i].m;
        memcpy(d->obs_uv + 8 * i, obs[i].uv, sizeof(float) * 8);
    }
    d->optimize_cam_poses = flags[0]; d->optimize_marker_poses = flags[1];
    d->optimize_object_poses = flags[2]; d->optimize_cam_intrinsics = flags[3];
    *out = d;
    return AAR_OK;
}

int aar_solution_write_yaml(const char *path, const aar_dataset *d) {
    if (!path || !d) return set_error(AAR_ERR_INVALID, "aar_solution_write_yaml: null argument");
    FILE *f = fopen(path, "w");
    if (!f) return set_error(AAR_ERR_IO, "Could not open a file in: %s for writing.", path);
    PoseLayout L;
    L.C = d->num_cams; L.M = d->num_markers; L.F = d->num_frames; L.rc = d->root_cam; L.rm = d->root_marker;
    fprintf(f, "%%YAML:1.0\n---\n");
    fprintf(f, "marker_size: %s\n", fs_double(d->marker_size).c_str());
    fprintf(f, "transforms_to_root_cam:\n");
    for (int c = 0; c < L.C; c++) {
        Rigid T = (c == L.rc) ? Rigid::identity() : pose_to_rigid(d->x_full + L.full_cam0() + 6LL * L.cam_slot(c));
        yaml_transform(f, "cam_id", d->cam_ids[c], T);
    }
    fprintf(f, "transforms_to_root_marker:\n");
    for (int m = 0; m < L.M; m++) {
        Rigid T = (m == L.rm) ? Rigid::identity() : pose_to_rigid(d->x_full + L.full_mk0() + 6LL * L.mk_slot(m));
        yaml_transform(f, "marker_id", d->marker_ids[m], T);
    }
    fprintf(f, "root_marker_to_root_cam:\n");
    for (int i = 0; i < L.F; i++) yaml_transform(f, "frame_id", d->frame_ids[i], pose_to_rigid(d->x_full + L.full_fr0() + 6LL * i));
    const bool good = !ferror(f);
    fclose(f);
    return good ? AAR_OK : set_error(AAR_ERR_IO, "write error on %s", path);
}

int aar_detections_write(const char *path, const aar_dataset *d) {
    if (!path || !d) return set_error(AAR_ERR_INVALID, "aar_detections_write: null argument");
    FILE *f = fopen(path, "wb");
    if (!f) return set_error(AAR_ERR_IO, "Could not open to write the detection file at: %s", path);
    Writer w{f};
    // camera slot = camera id (detect_markers writes one slot per calib folder, apps/detect_markers.cpp:97-101)
    int max_cam = 0;
    for (int c = 0; c < d->num_cams; c++) max_cam = d->cam_ids[c] > max_cam ? d->cam_ids[c] : max_cam;
    const size_t num_cams = (size_t)max_cam + 1;
    w.put<size_t>(num_cams);
    std::vector<int64_t> fstart(d->num_frames + 1, 0);
    for (int64_t o = 0; o < d->num_obs; o++) fstart[d->obs_frame[o] + 1]++;
    for (int i = 0; i < d->num_frames; i++) fstart[i + 1] += fstart[i];
    int next_frame_id = 0;
    for (int i = 0; i < d->num_frames; i++) {
        // frame index in the file = position: pad the frames that were dropped with empty records
        for (; next_frame_id < d->frame_ids[i]; next_frame_id++)
            for (size_t c = 0; c < num_cams; c++) w.put<size_t>(0);
        std::vector<std::vector<int64_t>> per_cam(num_cams);
        for (int64_t o = fstart[i]; o < fstart[i + 1]; o++) per_cam[d->cam_ids[d->obs_cam[o]]].push_back(o);
        for (size_t c = 0; c < num_cams; c++) {
            w.put<size_t>(per_cam[c].size());
            for (int64_t o : per_cam[c]) write_marker(w, d->marker_ids[d->obs_marker[o]], d->obs_uv + 8 * o);
        }
        next_frame_id = d->frame_ids[i] + 1;
    }
    const bool good = !ferror(f);
    fclose(f);
    return good ? AAR_OK : set_error(AAR_ERR_IO, "write error on %s", path);
}

}  // extern "C"
