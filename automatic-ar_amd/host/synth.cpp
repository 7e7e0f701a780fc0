// Deterministic synthetic multi-camera / multi-marker sequences.
// The reference ships no data sets (README.md:55-56 links external downloads), so every benchmark and
// parity input of this repository comes from here.  Portable by construction: splitmix64, explicit
// 53-bit doubles, Box-Muller; no std::*_distribution.
#include <cmath>
#include <cstring>
#include <vector>

#include "internal.h"
#include "se3.h"

namespace {

using namespace aar;

struct Rng {
    uint64_t s;
    bool have = false;
    double spare = 0;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        return z ^ (z >> 31);
    }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }  // [0,1)
    double normal() {
        if (have) { have = false; return spare; }
        const double u1 = ((double)(next() >> 11) + 1.0) * (1.0 / 9007199254740992.0);  // (0,1]
        const double u2 = uniform();
        const double r = std::sqrt(-2.0 * std::log(u1)), a = 6.283185307179586476925 * u2;
        spare = r * std::sin(a);
        have = true;
        return r * std::cos(a);
    }
};

void normalize3(double v[3]) {
    const double n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    v[0] /= n; v[1] /= n; v[2] /= n;
}
void cross3(const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
Rigid from_axes(const double x[3], const double y[3], const double z[3], const double pos[3]) {
    Rigid r;
    for (int i = 0; i < 3; i++) {
        r.R[i * 3 + 0] = x[i]; r.R[i * 3 + 1] = y[i]; r.R[i * 3 + 2] = z[i];
        r.t[i] = pos[i];
    }
    return r;
}

}  // namespace

extern "C" {

void aar_synth_default(aar_synth_desc *d, int32_t config_index) {
    memset(d, 0, sizeof *d);
    // BASELINE.json configs (index into the JSON array): 1: 4/12/100, 2: 8/40/500, 3: 8/40/2000, 4: 16/200/5000.
    // SURVEY.md numbers them 2..5; `config_index` here is the SURVEY number.
    switch (config_index) {
        case 1: d->num_cams = 3;  d->num_markers = 6;   d->num_frames = 180;  break;  // box-like plumbing case (one full turn; the bottom face is never seen)
        case 2: d->num_cams = 4;  d->num_markers = 12;  d->num_frames = 100;  break;
        case 3: d->num_cams = 8;  d->num_markers = 40;  d->num_frames = 500;  break;
        case 4: d->num_cams = 8;  d->num_markers = 40;  d->num_frames = 2000; break;
        case 5: d->num_cams = 16; d->num_markers = 200; d->num_frames = 5000; break;
        default: d->num_cams = 4; d->num_markers = 12; d->num_frames = 100; break;
    }
    d->seed = 20190219ULL + (uint64_t)config_index;
    d->marker_size = 0.05;
    d->noise_px = 0.3;
    d->init_rot_sigma = 0.02;
    d->init_trans_sigma = 0.01;
    d->init_scale = 1.0;
    if (config_index == 1) {   // the "box" case: three cameras side by side looking at one object, faces seen at a slant
        d->cam_arc_deg = 50.0;
        d->min_view_cos = 0.35;
        d->noise_px = 0.1;    // 25-pixel markers seen at up to 70 degrees: at 0.3 px the Initializer's marker votes (IPPE's two-fold
                              // ambiguity) pick flipped poses in this small scene and the LM ends in a 5 px local optimum
    }
}

int aar_synth_generate(const aar_synth_desc *sd, aar_dataset **out) {
    if (!sd || !out) return set_error(AAR_ERR_INVALID, "aar_synth_generate: null argument");
    const int C = sd->num_cams, M = sd->num_markers, F0 = sd->num_frames;
    if (C < 1 || M < 1 || F0 < 1) return set_error(AAR_ERR_INVALID, "aar_synth_generate: need C,M,F >= 1");
    Rng rng(sd->seed);
    const double PI = 3.14159265358979323846;
    const double centre[3] = {0, 0, 2.0};
    const double min_cos = sd->min_view_cos > 0 ? sd->min_view_cos : 0.8;

    // --- cameras: ring of radius 2 m around the scene centre in camera 0's x-z plane, looking at it ---
    std::vector<Rigid> Tc(C);  // camera i -> camera 0 (= world)
    for (int i = 0; i < C; i++) {
        double az = 2 * PI * i / C + (i > 0 ? 0.1 : 0.0);  // offset keeps the opposite camera off theta = pi
        if (sd->cam_arc_deg > 0 && C > 1) {   // cameras side by side on an arc centred on camera 0: 0, +s, -s, +2s, ...
            const double step = sd->cam_arc_deg * PI / 180.0 / (C - 1);
            az = ((i & 1) ? 1.0 : -1.0) * ((i + 1) / 2) * step;
        }
        const double pos[3] = {centre[0] + 2.0 * std::sin(az), 0, centre[2] - 2.0 * std::cos(az)};
        double z[3] = {centre[0] - pos[0], centre[1] - pos[1], centre[2] - pos[2]};
        normalize3(z);
        const double y[3] = {0, 1, 0};  // image y = world y (down)
        double x[3];
        cross3(y, z, x);
        normalize3(x);
        Tc[i] = from_axes(x, y, z, pos);
    }
    // --- markers: Fibonacci lattice on a 0.25 m sphere, z axis = outward normal ---
    std::vector<Rigid> O(M);  // marker i -> object frame
    const double golden = PI * (3.0 - std::sqrt(5.0));
    for (int i = 0; i < M; i++) {
        const double zf = 1.0 - (2.0 * i + 1.0) / M, rr = std::sqrt(std::fmax(0.0, 1.0 - zf * zf)), ph = golden * i;
        double n[3] = {rr * std::cos(ph), rr * std::sin(ph), zf};
        normalize3(n);
        double up[3] = {0, 0, 1};
        if (std::fabs(n[2]) > 0.9) { up[0] = 1; up[2] = 0; }
        double x[3], y[3];
        cross3(up, n, x);
        normalize3(x);
        cross3(n, x, y);
        const double pos[3] = {0.25 * n[0], 0.25 * n[1], 0.25 * n[2]};
        O[i] = from_axes(x, y, n, pos);
    }
    const Rigid O0inv = inverse(O[0]);
    std::vector<Rigid> Tm(M);  // marker i -> marker 0 (root)
    for (int i = 0; i < M; i++) Tm[i] = compose(O0inv, O[i]);

    const float hs = (float)sd->marker_size / 2.f;  // aruco Marker::get3DPoints, marker.cpp:358-367
    const double h = (double)hs;
    const double X[4][3] = {{-h, h, 0}, {h, h, 0}, {h, -h, 0}, {-h, -h, 0}};
    const double fx = 1000, fy = 1000, cx = 640, cy = 360;
    const int W = 1280, H = 720;

    // --- frames: object pose, visibility, noisy detections ---
    struct Obs { int f, c, m; float uv[8]; };
    std::vector<Obs> obs;
    std::vector<Rigid> Tf;       // root marker -> camera 0 for the kept frames
    std::vector<int> frame_ids;
    for (int f = 0; f < F0; f++) {
        const double ang = f * (2.0 * PI / 180.0);
        const double tilt = 0.35, pre = 0.011 * f;
        double axis[3] = {std::sin(tilt) * std::cos(pre), std::cos(tilt), std::sin(tilt) * std::sin(pre)};
        normalize3(axis);
        const double w[3] = {axis[0] * ang, axis[1] * ang, axis[2] * ang};
        Rigid P;  // object -> camera 0
        rodrigues_vec2mat(w, P.R);
        P.t[0] = centre[0] + 0.15 * std::sin(0.021 * f);
        P.t[1] = centre[1] + 0.15 * std::sin(0.034 * f + 0.5);
        P.t[2] = centre[2] + 0.15 * std::sin(0.013 * f + 1.1);
        std::vector<Obs> fobs;
        for (int c = 0; c < C; c++) {
            const Rigid Tci = inverse(Tc[c]);
            for (int m = 0; m < M; m++) {
                const Rigid Wm = compose(P, O[m]);  // marker -> world
                const double nrm[3] = {Wm.R[2], Wm.R[5], Wm.R[8]};
                double dir[3] = {Tc[c].t[0] - Wm.t[0], Tc[c].t[1] - Wm.t[1], Tc[c].t[2] - Wm.t[2]};
                normalize3(dir);
                if (nrm[0] * dir[0] + nrm[1] * dir[1] + nrm[2] * dir[2] < min_cos) continue;
                const Rigid Cm = compose(Tci, Wm);  // marker -> camera c
                double uv[8];
                bool inside = true;
                for (int k = 0; k < 4 && inside; k++) {
                    double p[3];
                    mat3_vec(Cm.R, X[k], p);
                    for (int i = 0; i < 3; i++) p[i] += Cm.t[i];
                    if (p[2] <= 0.05) { inside = false; break; }
                    uv[2 * k] = fx * p[0] / p[2] + cx;
                    uv[2 * k + 1] = fy * p[1] / p[2] + cy;
                    if (uv[2 * k] < 0 || uv[2 * k] >= W || uv[2 * k + 1] < 0 || uv[2 * k + 1] >= H) inside = false;
                }
                if (!inside) continue;
                Obs o;
                o.f = f; o.c = c; o.m = m;
                for (int k = 0; k < 8; k++) o.uv[k] = (float)(uv[k] + sd->noise_px * rng.normal());
                fobs.push_back(o);
            }
        }
        if (fobs.size() < 2) continue;  // libs/initializer.cpp:379
        const int fi = (int)Tf.size();
        for (auto &o : fobs) { o.f = fi; obs.push_back(o); }
        Tf.push_back(compose(P, O[0]));
        frame_ids.push_back(f);
    }
    const int F = (int)Tf.size();
    if (F == 0) return set_error(AAR_ERR_INVALID, "aar_synth_generate: no frame has >= 2 observations");

    aar_dataset *d = dataset_alloc(C, M, F, (int64_t)obs.size(), true);
    d->root_cam = 0; d->root_marker = 0;
    d->marker_size = (double)(float)sd->marker_size;  // float m_size parameter, libs/multicam_mapper.h:20
    for (int c = 0; c < C; c++) {
        d->cam_ids[c] = c;
        d->image_sizes[2 * c] = W; d->image_sizes[2 * c + 1] = H;
        double *K = d->cam_mats + 9 * c;
        K[0] = fx; K[1] = 0; K[2] = cx; K[3] = 0; K[4] = fy; K[5] = cy; K[6] = 0; K[7] = 0; K[8] = 1;
    }
    for (int m = 0; m < M; m++) d->marker_ids[m] = m;
    for (int f = 0; f < F; f++) d->frame_ids[f] = frame_ids[f];
    for (size_t i = 0; i < obs.size(); i++) {
        d->obs_frame[i] = obs[i].f; d->obs_cam[i] = obs[i].c; d->obs_marker[i] = obs[i].m;
        memcpy(d->obs_uv + 8 * i, obs[i].uv, sizeof(float) * 8);
    }
    // ground truth and perturbed initial guess, reference packing order
    PoseLayout L;
    L.C = C; L.M = M; L.F = F; L.rc = 0; L.rm = 0;
    const double sr = sd->init_rot_sigma * sd->init_scale, st = sd->init_trans_sigma * sd->init_scale;
    auto put = [&](int64_t off, const Rigid &T) {
        rigid_to_pose(T, d->x_truth + off);
        for (int i = 0; i < 3; i++) d->x_full[off + i] = d->x_truth[off + i] + sr * rng.normal();
        for (int i = 3; i < 6; i++) d->x_full[off + i] = d->x_truth[off + i] + st * rng.normal();
    };
    for (int c = 0; c < C; c++)
        if (c != L.rc) put(L.full_cam0() + 6LL * L.cam_slot(c), Tc[c]);
    for (int m = 0; m < M; m++)
        if (m != L.rm) put(L.full_mk0() + 6LL * L.mk_slot(m), Tm[m]);
    for (int f = 0; f < F; f++) put(L.full_fr0() + 6LL * f, Tf[f]);
    *out = d;
    return AAR_OK;
}

}  // extern "C"
