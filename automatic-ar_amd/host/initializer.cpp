// Host logic of the Initializer (libs/initializer.cpp): file readers, candidate enumeration in the reference's container order,
// the two spanning trees, and the data set MultiCamMapper(Initializer&) starts from (libs/multicam_mapper.cpp:252-254,281-331).
// Everything per-detection or per-candidate runs on the device (csrc/init_kernels.hip); there is no CPU pose solver or vote here.
#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <fstream>
#include <limits>
#include <map>
#include <queue>
#include <set>
#include <string>
#include <vector>

#include "init_device.h"
#include "se3.h"

namespace aar {
namespace {

// 3x4 top of a 4x4 transform, row-major
struct Aff12 {
    double m[12];
};

Aff12 aff_identity() {
    Aff12 a;
    for (int i = 0; i < 12; i++) a.m[i] = (i % 5 == 0) ? 1.0 : 0.0;
    return a;
}

Aff12 aff_mul(const Aff12 &x, const Aff12 &y) {
    Aff12 r;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) r.m[i * 4 + j] = x.m[i * 4] * y.m[j] + x.m[i * 4 + 1] * y.m[4 + j] + x.m[i * 4 + 2] * y.m[8 + j];
        r.m[i * 4 + 3] = x.m[i * 4] * y.m[3] + x.m[i * 4 + 1] * y.m[7] + x.m[i * 4 + 2] * y.m[11] + x.m[i * 4 + 3];
    }
    return r;
}

// cv::Mat::inv() of the 4x4: general inverse (votes pick float-rounded poses whose rotation block is not exactly orthogonal)
Aff12 aff_inv(const Aff12 &x) {
    const double a = x.m[0], b = x.m[1], c = x.m[2], d = x.m[4], e = x.m[5], f = x.m[6], g = x.m[8], h = x.m[9], i = x.m[10];
    const double c00 = e * i - f * h, c01 = f * g - d * i, c02 = d * h - e * g;
    const double idet = 1.0 / (a * c00 + b * c01 + c * c02);
    Aff12 r;
    r.m[0] = c00 * idet; r.m[1] = (c * h - b * i) * idet; r.m[2] = (b * f - c * e) * idet;
    r.m[4] = c01 * idet; r.m[5] = (a * i - c * g) * idet; r.m[6] = (c * d - a * f) * idet;
    r.m[8] = c02 * idet; r.m[9] = (b * g - a * h) * idet; r.m[10] = (a * e - b * d) * idet;
    for (int k = 0; k < 3; k++) r.m[k * 4 + 3] = -(r.m[k * 4] * x.m[3] + r.m[k * 4 + 1] * x.m[7] + r.m[k * 4 + 2] * x.m[11]);
    return r;
}

void aff_to_pose(const Aff12 &a, double v[6]) {  // transformation_mat2vec, libs/multicam_mapper.cpp:475-486
    Rigid r;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) r.R[i * 3 + j] = a.m[i * 4 + j];
        r.t[i] = a.m[i * 4 + 3];
    }
    rigid_to_pose(r, v);
}

struct Cand {
    int32_t key, a, b;
};

// One pose-estimation entry of a frame: detection u of (outer, inner); `u` carries poses 2u and (has2) 2u+1.
struct Ent {
    int32_t outer, inner, u;
};

// fill_transformation_sets for ONE frame (libs/initializer.cpp:95-125): objects = distinct `inner` of one `outer`, candidates in
// the order it1 -> pose i -> it2 > it1 -> pose j
void enumerate_pairs(std::vector<Ent> &ents, const std::vector<char> &has2, int32_t n_inner, std::vector<Cand> &out) {
    std::stable_sort(ents.begin(), ents.end(), [](const Ent &x, const Ent &y) {
        return x.outer != y.outer ? x.outer < y.outer : x.inner < y.inner;
    });
    std::vector<std::pair<int32_t, std::vector<int32_t>>> objs;
    size_t g0 = 0;
    while (g0 < ents.size()) {
        size_t g1 = g0;
        objs.clear();
        while (g1 < ents.size() && ents[g1].outer == ents[g0].outer) {
            if (objs.empty() || objs.back().first != ents[g1].inner) objs.push_back({ents[g1].inner, {}});
            objs.back().second.push_back(2 * ents[g1].u);
            if (has2[ents[g1].u]) objs.back().second.push_back(2 * ents[g1].u + 1);
            g1++;
        }
        if (objs.size() > 1)
            for (size_t x = 0; x < objs.size(); x++)
                for (int32_t pi : objs[x].second)
                    for (size_t y = x + 1; y < objs.size(); y++)
                        for (int32_t pj : objs[y].second) out.push_back({objs[x].first * n_inner + objs[y].first, pi, pj});
        g0 = g1;
    }
}

struct Edge {
    double weight;
    Aff12 T;
};

// make_mst + find_transforms_to_root (libs/initializer.cpp:237-315) on ranks 0..n-1 (rank order = id order, root = rank 0).
// Returns false and the first unreachable rank when the co-visibility graph is not connected.
bool spanning_transforms(int n, const std::map<int64_t, Edge> &edges, std::vector<Aff12> &to_root, int &unreachable) {
    const double inf = std::numeric_limits<double>::max();
    std::vector<double> dist(n, inf);
    std::vector<int> parent(n, -1);
    std::vector<char> outside(n, 1);
    std::vector<std::set<int>> children(n);
    if (n > 0) dist[0] = 0;
    for (int round = 0; round < n; round++) {
        int mn = -1;
        for (int i = 0; i < n; i++)
            if (outside[i] && (mn < 0 || dist[i] < dist[mn])) mn = i;
        for (int i = 0; i < n; i++) {
            if (!outside[i] || i == mn) continue;
            const int lo = std::min(mn, i), hi = std::max(mn, i);
            auto e = edges.find((int64_t)lo * n + hi);
            if (e == edges.end()) continue;
            if (e->second.weight < dist[i]) {
                dist[i] = e->second.weight;
                if (parent[i] != -1) children[parent[i]].erase(i);
                children[mn].insert(i);
                parent[i] = mn;
            }
        }
        outside[mn] = 0;
    }
    std::vector<char> have(n, 0);
    to_root.assign(n, aff_identity());
    if (n == 0) return true;
    have[0] = 1;
    std::queue<int> q;
    q.push(0);
    while (!q.empty()) {
        const int p = q.front();
        for (int c : children[p]) {
            Aff12 T = c < p ? edges.at((int64_t)c * n + p).T : aff_inv(edges.at((int64_t)p * n + c).T);
            if (p != 0) T = aff_mul(to_root[p], T);
            to_root[c] = T;
            have[c] = 1;
            q.push(c);
        }
        q.pop();
    }
    for (int i = 0; i < n; i++)
        if (!have[i]) { unreachable = i; return false; }
    return true;
}

struct DevGuard {
    InitDevice *d = nullptr;
    ~DevGuard() { if (d) initdev_destroy(d); }
};

}  // namespace
}  // namespace aar

using namespace aar;

extern "C" {

void aar_init_default_params(aar_init_params *p) {
    memset(p, 0, sizeof *p);
    p->marker_size = 0.05;
    p->threshold = 2.0;       // libs/initializer.h:52
    p->min_detections = 2;    // libs/initializer.h:51
}

void aar_detections_free(aar_detections *d) {
    if (!d) return;
    free(d->det_frame); free(d->det_cam); free(d->det_id); free(d->det_uv);
    free(d);
}

int aar_detections_read(const char *path, const int32_t *subseqs, int32_t n_subseqs, aar_detections **out) {
    if (!path || !out || n_subseqs < 0 || (n_subseqs > 0 && !subseqs)) return set_error(AAR_ERR_INVALID, "aar_detections_read: bad argument");
    FILE *f = fopen(path, "rb");
    if (!f) return set_error(AAR_ERR_IO, "Could not open to read the detection file at: %s", path);
    std::vector<int32_t> fr, cm, id;
    std::vector<float> uv;
    size_t num_cams = 0;
    int num_frames = 0;
    bool bad = false;
    if (fread(&num_cams, sizeof num_cams, 1, f) == 1) {
        if (num_cams > (1u << 20)) bad = true;
        for (int frame = 0; !bad; frame++) {
            // a frame record that ends early is dropped as a whole (libs/initializer.cpp:331-346)
            const size_t keep = fr.size();
            bool end_of_data = false;
            for (size_t c = 0; c < num_cams && !end_of_data; c++) {
                size_t n = 0;
                if (fread(&n, sizeof n, 1, f) != 1) { end_of_data = true; break; }
                if (n > (1u << 24)) { bad = true; break; }
                for (size_t m = 0; m < n; m++) {
                    int32_t mid = 0;
                    float xy[8] = {0};
                    if (fread(&mid, sizeof mid, 1, f) != 1 || fread(xy, sizeof(float), 8, f) != 8) { end_of_data = true; break; }
                    fr.push_back(frame); cm.push_back((int32_t)c); id.push_back(mid);
                    uv.insert(uv.end(), xy, xy + 8);
                }
            }
            if (end_of_data || bad) {
                fr.resize(keep); cm.resize(keep); id.resize(keep); uv.resize(8 * keep);
                break;
            }
            num_frames = frame + 1;
        }
    }
    fclose(f);
    if (bad) return set_error(AAR_ERR_IO, "%s is not an aruco.detections file", path);
    // sub-sequences: frames before the first and between consecutive ranges are emptied (libs/initializer.cpp:350-359)
    if (n_subseqs > 0) {
        std::vector<char> drop((size_t)num_frames, 0);
        int prev_last = -1;
        for (int i = 0; i + 1 < n_subseqs; i += 2) {
            for (int g = prev_last + 1; g < subseqs[i]; g++) {
                if (g < 0 || g >= num_frames) return set_error(AAR_ERR_INVALID, "sub-sequence frame %d outside the %d frames of %s", g, num_frames, path);
                drop[g] = 1;
            }
            prev_last = subseqs[i + 1];
        }
        size_t w = 0;
        for (size_t i = 0; i < fr.size(); i++)
            if (!drop[fr[i]]) {
                fr[w] = fr[i]; cm[w] = cm[i]; id[w] = id[i];
                memmove(&uv[8 * w], &uv[8 * i], 8 * sizeof(float));
                w++;
            }
        fr.resize(w); cm.resize(w); id.resize(w); uv.resize(8 * w);
    }
    aar_detections *d = (aar_detections *)calloc(1, sizeof *d);
    const size_t n = fr.size();
    d->num_cams = (int32_t)num_cams; d->num_frames = num_frames; d->num_det = (int64_t)n;
    d->det_frame = (int32_t *)calloc(n ? n : 1, sizeof(int32_t));
    d->det_cam = (int32_t *)calloc(n ? n : 1, sizeof(int32_t));
    d->det_id = (int32_t *)calloc(n ? n : 1, sizeof(int32_t));
    d->det_uv = (float *)calloc(n ? 8 * n : 1, sizeof(float));
    if (n) {
        memcpy(d->det_frame, fr.data(), n * sizeof(int32_t)); memcpy(d->det_cam, cm.data(), n * sizeof(int32_t));
        memcpy(d->det_id, id.data(), n * sizeof(int32_t)); memcpy(d->det_uv, uv.data(), 8 * n * sizeof(float));
    }
    *out = d;
    return AAR_OK;
}

int aar_subseqs_read(const char *path, int32_t **out, int32_t *n) {
    if (!path || !out || !n) return set_error(AAR_ERR_INVALID, "aar_subseqs_read: null argument");
    std::ifstream in(path);
    if (!in.is_open()) return set_error(AAR_ERR_IO, "Could not open subsequences file at: %s", path);
    std::vector<int32_t> v;
    int num;
    while (in >> num) v.push_back(num);
    *out = (int32_t *)calloc(v.size() ? v.size() : 1, sizeof(int32_t));
    if (!v.empty()) memcpy(*out, v.data(), v.size() * sizeof(int32_t));
    *n = (int32_t)v.size();
    return AAR_OK;
}

int aar_cam_configs_read(const char *folder, aar_cam_model **out, int32_t *n_cams) {
    return aar_cam_configs_read_ex(folder, AAR_DIR_ORDER_NAME, out, n_cams);
}

int aar_cam_configs_read_ex(const char *folder, int32_t dir_order, aar_cam_model **out, int32_t *n_cams) {
    if (!folder || !out || !n_cams) return set_error(AAR_ERR_INVALID, "aar_cam_configs_read: null argument");
    if (dir_order != AAR_DIR_ORDER_NAME && dir_order != AAR_DIR_ORDER_READDIR) return set_error(AAR_ERR_INVALID, "aar_cam_configs_read_ex: bad dir_order");
    DIR *dir = opendir(folder);
    if (!dir) return set_error(AAR_ERR_IO, "could not open folder %s", folder);
    std::vector<std::string> dirs;
    for (dirent *e = readdir(dir); e; e = readdir(dir)) {
        const std::string name = e->d_name;
        // the reference's get_dirs_list keeps "." and ".." (libs/filesystem.cpp:9-12): a calib file in the data folder itself
        // or in its parent would become a camera slot there; kept out in name order, kept in in readdir order
        if (dir_order == AAR_DIR_ORDER_NAME && (name == "." || name == "..")) continue;
        struct stat st;
        if (stat((std::string(folder) + "/" + name).c_str(), &st) == 0 && S_ISDIR(st.st_mode)) dirs.push_back(name);
    }
    closedir(dir);
    if (dir_order == AAR_DIR_ORDER_NAME) std::sort(dirs.begin(), dirs.end());   // AAR_DIR_ORDER_READDIR: as the file system lists them (libs/cam_config.cpp:80-95)
    std::vector<aar_cam_model> cams;
    const char *exts[3] = {"xml", "yml", "yaml"};
    for (const std::string &dn : dirs)
        for (const char *ext : exts) {
            const std::string p = std::string(folder) + "/" + dn + "/calib." + ext;
            struct stat st;
            if (stat(p.c_str(), &st) != 0) continue;
            aar_cam_model m;
            memset(&m, 0, sizeof m);
            if (aar_cam_config_read(p.c_str(), m.K, m.dist, &m.n_dist, &m.width, &m.height) == AAR_OK) cams.push_back(m);
        }
    *out = (aar_cam_model *)calloc(cams.size() ? cams.size() : 1, sizeof(aar_cam_model));
    if (!cams.empty()) memcpy(*out, cams.data(), cams.size() * sizeof(aar_cam_model));
    *n_cams = (int32_t)cams.size();
    return AAR_OK;
}

// `fixed` = a solution whose camera and marker transforms are kept (apps/track.cpp:70-89,117-119): only obtain_pose_estimations
// and init_object_transforms run, on the detections of cameras / markers the solution knows.
static int run_initializer(const aar_detections *det, const aar_cam_model *cams, int32_t n_cams, const aar_init_params *prm,
                           const aar_dataset *fixed, aar_dataset **out) {
    if (!det || !prm || !out || n_cams < 0 || (n_cams > 0 && !cams)) return set_error(AAR_ERR_INVALID, "aar_initializer_run: null argument");
    if (!(prm->marker_size > 0)) return set_error(AAR_ERR_INVALID, "aar_initializer_run: marker_size must be positive");
    if (prm->n_excluded < 0 || (prm->n_excluded > 0 && !prm->excluded_cams)) return set_error(AAR_ERR_INVALID, "aar_initializer_run: bad excluded_cams");
    const int64_t nd = det->num_det;
    const int S = det->num_cams, NF = det->num_frames;
    const bool verbose = getenv("AAR_INIT_VERBOSE") != nullptr;   // stage sizes and times on stderr (scripts/init_bench.py)
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_mark = now();
    auto lap = [&](const char *what) {
        const double t = now();
        if (verbose) fprintf(stderr, "[aar init] %-28s %8.3f ms\n", what, 1e3 * (t - t_mark));
        t_mark = t;
    };
    std::set<int> excl(prm->excluded_cams, prm->excluded_cams + prm->n_excluded);
    for (int64_t i = 0; i < nd; i++) {
        if (det->det_frame[i] < 0 || det->det_frame[i] >= NF || det->det_cam[i] < 0 || det->det_cam[i] >= S)
            return set_error(AAR_ERR_INVALID, "detection %lld: frame / camera slot out of range", (long long)i);
        if (i > 0 && (det->det_frame[i] < det->det_frame[i - 1] ||
                      (det->det_frame[i] == det->det_frame[i - 1] && det->det_cam[i] < det->det_cam[i - 1])))
            return set_error(AAR_ERR_INVALID, "detections must be ordered by frame, then camera slot");
    }

    std::set<int> known_cams, known_markers;
    if (fixed) {
        known_cams.insert(fixed->cam_ids, fixed->cam_ids + fixed->num_cams);
        known_markers.insert(fixed->marker_ids, fixed->marker_ids + fixed->num_markers);
    }
    // with a fixed solution, a detection of a camera / marker it does not hold cannot be placed and is dropped (the reference
    // substitutes the identity for it in fill_transformation_set and then fails in MatArray::m.at)
    auto takes_part = [&](int64_t i) {
        if (excl.count(det->det_cam[i])) return false;
        return !fixed || (known_cams.count(det->det_cam[i]) && known_markers.count(det->det_id[i]));
    };
    // ---- frames with at least min_detections detections in the cameras that take part (libs/initializer.cpp:371-380) ----
    std::vector<int64_t> per_frame((size_t)NF, 0);
    for (int64_t i = 0; i < nd; i++)
        if (takes_part(i)) per_frame[det->det_frame[i]]++;
    std::vector<int64_t> used;          // detection indices, file order
    std::vector<int32_t> kept_frames;   // keys of frame_cam_markers
    for (int64_t i = 0; i < nd; i++) {
        const int f = det->det_frame[i];
        if (!takes_part(i) || !(per_frame[f] >= prm->min_detections)) continue;
        if (kept_frames.empty() || kept_frames.back() != f) kept_frames.push_back(f);
        used.push_back(i);
    }
    const int64_t U = (int64_t)used.size();
    if (U == 0) return set_error(AAR_ERR_INVALID, "no frame has %d or more detections", prm->min_detections);
    std::set<int> cam_set, mk_set;
    for (int64_t u = 0; u < U; u++) { cam_set.insert(det->det_cam[used[u]]); mk_set.insert(det->det_id[used[u]]); }
    int max_used_cam = *cam_set.rbegin();
    if (fixed) { cam_set = known_cams; mk_set = known_markers; }   // the solution's ids, seen or not
    const std::vector<int32_t> cam_ids(cam_set.begin(), cam_set.end()), marker_ids(mk_set.begin(), mk_set.end());
    const int C = (int)cam_ids.size(), M = (int)marker_ids.size(), F = (int)kept_frames.size();
    if (M > 46340 || C > 46340) return set_error(AAR_ERR_UNSUPPORTED, "more than 46340 cameras or markers");   // pair keys are int32
    if (max_used_cam >= n_cams)
        return set_error(AAR_ERR_INVALID, "camera slot %d has detections but only %d calibrations were given", max_used_cam, n_cams);
    std::map<int, int> cam_rank, mk_rank;
    for (int c = 0; c < C; c++) cam_rank[cam_ids[c]] = c;
    for (int m = 0; m < M; m++) mk_rank[marker_ids[m]] = m;

    // ---- obtain_pose_estimations: IPPE on every used detection (device) ----
    DevGuard G;
    if (int rc = initdev_create(prm->device_id, &G.d)) return rc;
    std::vector<float> uv(8 * (size_t)U), e1((size_t)U), e2((size_t)U), uvK(8 * (size_t)U);
    std::vector<int32_t> ucam((size_t)U), urank_c((size_t)U), urank_m((size_t)U), uframe((size_t)U);
    for (int64_t u = 0; u < U; u++) {
        memcpy(&uv[8 * u], det->det_uv + 8 * used[u], 8 * sizeof(float));
        ucam[u] = det->det_cam[used[u]];
        urank_c[u] = cam_rank[ucam[u]];
        urank_m[u] = mk_rank[det->det_id[used[u]]];
        uframe[u] = det->det_frame[used[u]];
    }
    if (int rc = initdev_ippe(G.d, cams, n_cams, (float)prm->marker_size, U, uv.data(), ucam.data(), e1.data(), e2.data(), uvK.data()))
        return rc;
    lap("ippe (device, incl. copies)");
    std::vector<char> has2((size_t)U);
    for (int64_t u = 0; u < U; u++) has2[u] = ((double)e2[u] / (double)e1[u] < prm->threshold) ? 1 : 0;  // :409

    // ---- init_transforms_cam / init_transforms_marker (libs/initializer.cpp:421-451) ----
    std::vector<Aff12> to_root[2];
    PoseLayout L;
    L.C = C; L.M = M; L.F = F; L.rc = fixed ? fixed->root_cam : 0; L.rm = fixed ? fixed->root_marker : 0;
    if (fixed) {
        auto from_pose = [](const double *v) {
            const Rigid r = pose_to_rigid(v);
            Aff12 a;
            for (int i = 0; i < 3; i++) {
                for (int j = 0; j < 3; j++) a.m[i * 4 + j] = r.R[i * 3 + j];
                a.m[i * 4 + 3] = r.t[i];
            }
            return a;
        };
        to_root[0].assign(C, aff_identity());
        to_root[1].assign(M, aff_identity());
        for (int c = 0; c < C; c++)
            if (c != L.rc) to_root[0][c] = from_pose(fixed->x_full + L.full_cam0() + 6LL * L.cam_slot(c));
        for (int m = 0; m < M; m++)
            if (m != L.rm) to_root[1][m] = from_pose(fixed->x_full + L.full_mk0() + 6LL * L.mk_slot(m));
    }
    for (int type = 0; type < 2 && !fixed; type++) {
        const int n_nodes = type == 0 ? C : M;
        std::vector<Cand> cands;
        std::vector<Ent> ents;
        for (int64_t u0 = 0; u0 < U;) {
            int64_t u1 = u0;
            ents.clear();
            while (u1 < U && uframe[u1] == uframe[u0]) {
                // cameras: poses of one marker in several cameras; markers: poses of several markers in one camera
                ents.push_back(type == 0 ? Ent{urank_m[u1], urank_c[u1], (int32_t)u1} : Ent{urank_c[u1], urank_m[u1], (int32_t)u1});
                u1++;
            }
            enumerate_pairs(ents, has2, n_nodes, cands);
            u0 = u1;
        }
        if ((int64_t)cands.size() >= (1LL << 31) - 64) return set_error(AAR_ERR_UNSUPPORTED, "more than 2^31 candidate transforms");
        // group by set, stable: inside a set the reference's push order (frame, outer id, i, j) is kept
        std::vector<int32_t> set_of((size_t)n_nodes * (size_t)n_nodes, -1);   // key = rank1 * n_nodes + rank2 -> set
        for (const Cand &c : cands) set_of[c.key] = 0;
        std::vector<int32_t> set_key;
        for (size_t k = 0; k < set_of.size(); k++)
            if (set_of[k] == 0) { set_of[k] = (int32_t)set_key.size(); set_key.push_back((int32_t)k); }
        const int32_t ns = (int32_t)set_key.size();
        std::vector<int64_t> begin((size_t)ns + 1, 0);
        for (const Cand &c : cands) begin[set_of[c.key] + 1]++;
        for (int s = 0; s < ns; s++) begin[s + 1] += begin[s];
        std::vector<int64_t> fill(begin.begin(), begin.end() - 1);
        std::vector<int32_t> ca(cands.size()), cb(cands.size());
        for (const Cand &c : cands) {
            const int64_t at = fill[set_of[c.key]]++;
            ca[at] = c.a; cb[at] = c.b;
        }
        lap(type == 0 ? "camera candidates (host)" : "marker candidates (host)");
        if (verbose) {
            double pairs = 0;
            for (int s = 0; s < ns; s++) pairs += (double)(begin[s + 1] - begin[s]) * (double)(begin[s + 1] - begin[s]);
            fprintf(stderr, "[aar init] %s sets: %d, candidates: %zu, vote pairs: %.4g\n", type == 0 ? "camera" : "marker", ns, cands.size(), pairs);
        }
        std::vector<int64_t> best((size_t)ns);
        std::vector<double> weight((size_t)ns), bestT(12 * (size_t)ns);
        if (int rc = initdev_pair_vote(G.d, type, (int64_t)cands.size(), ca.data(), cb.data(), ns, begin.data(), prm->marker_size,
                                       best.data(), weight.data(), bestT.data()))
            return rc;
        lap(type == 0 ? "camera votes (device)" : "marker votes (device)");
        std::map<int64_t, Edge> edges;
        for (int s = 0; s < ns; s++) {
            if (best[s] < 0) continue;  // every candidate of the set is NaN
            Edge e;
            e.weight = weight[s];
            memcpy(e.T.m, &bestT[12 * (size_t)s], sizeof e.T.m);
            edges[(int64_t)set_key[s]] = e;   // key = rank1 * n_nodes + rank2, rank1 < rank2
        }
        int miss = -1;
        if (!spanning_transforms(n_nodes, edges, to_root[type], miss))
            return set_error(AAR_ERR_INVALID, "%s %d is not connected to the root %s %d by co-visible detections",
                             type == 0 ? "camera" : "marker", type == 0 ? cam_ids[miss] : marker_ids[miss],
                             type == 0 ? "camera" : "marker", type == 0 ? cam_ids[0] : marker_ids[0]);
    }

    // ---- init_object_transforms (libs/initializer.cpp:453-465): one set per kept frame, marker id -> camera id -> pose ----
    std::vector<int32_t> cpose, ccam, cmk;
    std::vector<int64_t> fbegin((size_t)F + 1, 0);
    {
        int fi = 0;
        std::vector<Ent> ents;
        for (int64_t u0 = 0; u0 < U; fi++) {
            int64_t u1 = u0;
            ents.clear();
            while (u1 < U && uframe[u1] == uframe[u0]) { ents.push_back(Ent{urank_m[u1], urank_c[u1], (int32_t)u1}); u1++; }
            std::stable_sort(ents.begin(), ents.end(), [](const Ent &x, const Ent &y) {
                return x.outer != y.outer ? x.outer < y.outer : x.inner < y.inner;
            });
            for (const Ent &e : ents) {
                cpose.push_back(2 * e.u); ccam.push_back(e.inner); cmk.push_back(e.outer);
                if (has2[e.u]) { cpose.push_back(2 * e.u + 1); ccam.push_back(e.inner); cmk.push_back(e.outer); }
            }
            fbegin[fi + 1] = (int64_t)cpose.size();
            u0 = u1;
        }
    }
    if (verbose) {
        double pairs = 0;
        for (int f = 0; f < F; f++) pairs += (double)(fbegin[f + 1] - fbegin[f]) * (double)(fbegin[f + 1] - fbegin[f]);
        fprintf(stderr, "[aar init] frame sets: %d, candidates: %zu, vote pairs: %.4g\n", F, cpose.size(), pairs);
    }
    std::vector<int64_t> fbest((size_t)F);
    std::vector<double> fweight((size_t)F), fT(12 * (size_t)F);
    if (int rc = initdev_object_vote(G.d, (int64_t)cpose.size(), cpose.data(), ccam.data(), cmk.data(), C, to_root[0][0].m, M,
                                     to_root[1][0].m, F, fbegin.data(), prm->marker_size, fbest.data(), fweight.data(), fT.data()))
        return rc;
    lap("frame votes (host + device)");
    for (int f = 0; f < F; f++)
        if (fbest[f] < 0) return set_error(AAR_ERR_NUMERIC, "frame %d: no finite object pose candidate", kept_frames[f]);

    // ---- MultiCamMapper::init (libs/multicam_mapper.cpp:281-331): ids, intrinsics, undistorted corners, pose vector ----
    aar_dataset *d = dataset_alloc(C, M, F, U, false);
    d->root_cam = L.rc; d->root_marker = L.rm;                 // *cam_ids.begin(), *marker_ids.begin() -- or the solution's
    d->marker_size = (double)(float)prm->marker_size;          // float m_size parameter, libs/multicam_mapper.h:20
    for (int c = 0; c < C; c++) {
        d->cam_ids[c] = cam_ids[c];
        if (fixed) {   // MultiCamMapper keeps its own intrinsics across init(object_poses, fcm), libs/multicam_mapper.cpp:272-279
            d->image_sizes[2 * c] = fixed->image_sizes[2 * c]; d->image_sizes[2 * c + 1] = fixed->image_sizes[2 * c + 1];
            memcpy(d->cam_mats + 9 * c, fixed->cam_mats + 9 * c, 9 * sizeof(double));
            memcpy(d->dist_coeffs + 5 * c, fixed->dist_coeffs + 5 * c, 5 * sizeof(double));
            continue;
        }
        const aar_cam_model &cm = cams[cam_ids[c]];
        d->image_sizes[2 * c] = cm.width; d->image_sizes[2 * c + 1] = cm.height;
        memcpy(d->cam_mats + 9 * c, cm.K, 9 * sizeof(double));
        for (int k = 0; k < 5; k++) d->dist_coeffs[5 * c + k] = k < cm.n_dist ? cm.dist[k] : 0.0;
    }
    for (int m = 0; m < M; m++) d->marker_ids[m] = marker_ids[m];
    for (int f = 0; f < F; f++) d->frame_ids[f] = kept_frames[f];
    {
        int fi = -1, last = -1;
        for (int64_t u = 0; u < U; u++) {
            if (uframe[u] != last) { fi++; last = uframe[u]; }
            d->obs_frame[u] = fi; d->obs_cam[u] = urank_c[u]; d->obs_marker[u] = urank_m[u];
        }
        memcpy(d->obs_uv, uvK.data(), sizeof(float) * 8 * (size_t)U);
    }
    if (fixed) {
        memcpy(d->x_full, fixed->x_full, sizeof(double) * (size_t)L.full_fr0());   // cameras | markers, untouched
        d->optimize_cam_poses = d->optimize_marker_poses = 0;                      // apps/track.cpp:93-95
    } else {
        for (int c = 1; c < C; c++) aff_to_pose(to_root[0][c], d->x_full + L.full_cam0() + 6LL * L.cam_slot(c));
        for (int m = 1; m < M; m++) aff_to_pose(to_root[1][m], d->x_full + L.full_mk0() + 6LL * L.mk_slot(m));
    }
    for (int f = 0; f < F; f++) {
        Aff12 T;
        memcpy(T.m, &fT[12 * (size_t)f], sizeof T.m);
        aff_to_pose(T, d->x_full + L.full_fr0() + 6LL * f);
    }
    *out = d;
    return AAR_OK;
}

int aar_initializer_run(const aar_detections *det, const aar_cam_model *cams, int32_t n_cams, const aar_init_params *prm,
                        aar_dataset **out) {
    return run_initializer(det, cams, n_cams, prm, nullptr, out);
}

int aar_initializer_object_poses(const aar_dataset *solution, const aar_detections *det, const aar_cam_model *cams, int32_t n_cams,
                                 const aar_init_params *prm, aar_dataset **out) {
    if (!solution) return set_error(AAR_ERR_INVALID, "aar_initializer_object_poses: null solution");
    if (solution->num_cams < 1 || solution->num_markers < 1) return set_error(AAR_ERR_INVALID, "aar_initializer_object_poses: empty solution");
    return run_initializer(det, cams, n_cams, prm, solution, out);
}

}  // extern "C"
