/*
 * oracle/init_oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See init_oracle.h (parity unpinned: OpenCV absent).
 *
 * Restates libs/initializer.cpp and the square-marker IPPE of 3rdparty/aruco/aruco/ippe.cpp with the reference's container
 * semantics.  Compiled with -ffp-contract=off: the float reprojection error of IPPE is a chain of individually rounded float
 * operations (ippe.cpp:289-321) and must not be fused.
 */
#include "init_oracle.h"

#include <cfloat>
#include <cmath>
#include <cstring>
#include <limits>
#include <map>
#include <queue>
#include <set>
#include <tuple>
#include <utility>
#include <vector>

namespace {

struct M4 {
    double a[16];
    bool empty = false;
};

M4 eye4() {
    M4 m;
    for (int i = 0; i < 16; i++) m.a[i] = (i % 5 == 0) ? 1.0 : 0.0;
    return m;
}

// cv::Mat * cv::Mat for 4x4 CV_64F: every element accumulated over k = 0..3 in order
M4 mul(const M4 &x, const M4 &y) {
    M4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0;
            for (int k = 0; k < 4; k++) s += x.a[i * 4 + k] * y.a[k * 4 + j];
            r.a[i * 4 + j] = s;
        }
    return r;
}

// cv::Mat::inv() (DECOMP_LU) on a 4x4: Gaussian elimination with partial pivoting on [A | I]
M4 inv(const M4 &x) {
    double A[4][8];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            A[i][j] = x.a[i * 4 + j];
            A[i][4 + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int c = 0; c < 4; c++) {
        int p = c;
        for (int r = c + 1; r < 4; r++)
            if (std::fabs(A[r][c]) > std::fabs(A[p][c])) p = r;
        if (p != c)
            for (int j = 0; j < 8; j++) std::swap(A[p][j], A[c][j]);
        const double d = 1.0 / A[c][c];
        for (int r = c + 1; r < 4; r++) {
            const double f = A[r][c] * d;
            for (int j = c; j < 8; j++) A[r][j] -= f * A[c][j];
        }
    }
    for (int c = 3; c >= 0; c--) {  // back substitution, one right-hand side per column of I
        for (int j = 4; j < 8; j++) {
            double s = A[c][j];
            for (int k = c + 1; k < 4; k++) s -= A[c][k] * A[k][j];
            A[c][j] = s / A[c][c];
        }
    }
    M4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) r.a[i * 4 + j] = A[i][4 + j];
    return r;
}

// ---------------------------------------------------------------------------------------------------------------------
// IPPE for a centred square (Collins & Bartoli, "Infinitesimal Plane-based Pose Estimation", IJCV 2014, Algorithm 1), as
// 3rdparty/aruco/aruco/ippe.cpp:141-223 applies it.
// ---------------------------------------------------------------------------------------------------------------------

// Homography of the square (-h,h),(h,h),(h,-h),(-h,-h) onto four points, h22 = 1 (ippe.cpp:538-578 solves the same 8x8
// system in expanded form; here: unit-square-to-quadrilateral mapping composed with the affine map of the marker frame).
void square_homography(const float *q, double h, double H[9]) {
    const double x0 = q[0], y0 = q[1], x1 = q[2], y1 = q[3], x2 = q[4], y2 = q[5], x3 = q[6], y3 = q[7];
    const double sx = x0 - x1 + x2 - x3, sy = y0 - y1 + y2 - y3;
    const double dx1 = x1 - x2, dx2 = x3 - x2, dy1 = y1 - y2, dy2 = y3 - y2;
    const double den = dx1 * dy2 - dy1 * dx2;
    const double g = (sx * dy2 - sy * dx2) / den, k = (dx1 * sy - dy1 * sx) / den;
    // unit square (u,v): (0,0)->q0, (1,0)->q1, (1,1)->q2, (0,1)->q3
    const double U[9] = {x1 - x0 + g * x1, x3 - x0 + k * x3, x0, y1 - y0 + g * y1, y3 - y0 + k * y3, y0, g, k, 1.0};
    // u = (X + h) / 2h, v = (h - Y) / 2h
    const double s = 1.0 / (2.0 * h);
    double Hn[9];
    for (int r = 0; r < 3; r++) {
        Hn[r * 3 + 0] = U[r * 3 + 0] * s;
        Hn[r * 3 + 1] = -U[r * 3 + 1] * s;
        Hn[r * 3 + 2] = 0.5 * (U[r * 3 + 0] + U[r * 3 + 1]) + U[r * 3 + 2];
    }
    for (int i = 0; i < 9; i++) H[i] = Hn[i] / Hn[8];
}

void mat3mul(const double *A, const double *B, double *C) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}

// ippe.cpp:427-536.  J = Jacobian of the homography at the origin, (p,q) = image of the origin.
void ippe_rotations(const double J[4], double p, double q, double Ra[9], double Rb[9]) {
    // Rv: rotation that takes the optical axis onto the ray through (p,q)
    const double s = std::sqrt(p * p + q * q + 1), t = std::sqrt(p * p + q * q);
    const double ct = 1 / s, st = std::sqrt(1 - 1 / (s * s));
    const double kx = p / t, ky = q / t;
    const double Rv[9] = {(ct - 1) * kx * kx + 1, kx * ky * (ct - 1),     kx * st,
                          kx * ky * (ct - 1),     (ct - 1) * ky * ky + 1, ky * st,
                          -kx * st,               -ky * st,               (ct - 1) * (kx * kx + ky * ky) + 1};
    // B = [I2 | -(p,q)] Rv(:,0:1);  A = B^-1 J
    const double b00 = Rv[0] - p * Rv[6], b01 = Rv[1] - p * Rv[7], b10 = Rv[3] - q * Rv[6], b11 = Rv[4] - q * Rv[7];
    const double di = 1.0 / (b00 * b11 - b01 * b10);
    const double i00 = di * b11, i01 = -di * b01, i10 = -di * b10, i11 = di * b00;
    const double a00 = i00 * J[0] + i01 * J[2], a01 = i00 * J[1] + i01 * J[3];
    const double a10 = i10 * J[0] + i11 * J[2], a11 = i10 * J[1] + i11 * J[3];
    // largest singular value of A
    const double n00 = a00 * a00 + a01 * a01, n01 = a00 * a10 + a01 * a11, n11 = a10 * a10 + a11 * a11;
    const double gamma = std::sqrt(0.5 * (n00 + n11 + std::sqrt((n00 - n11) * (n00 - n11) + 4.0 * n01 * n01)));
    const double r00 = a00 / gamma, r01 = a01 / gamma, r10 = a10 / gamma, r11 = a11 / gamma;
    // third row completes unit columns; the sign of b1 makes the columns orthogonal
    double b0 = std::sqrt(-r00 * r00 - r10 * r10 + 1), b1 = std::sqrt(-r01 * r01 - r11 * r11 + 1);
    if (-r00 * r01 - r10 * r11 < 0) b1 = -b1;
    for (int sol = 0; sol < 2; sol++) {
        const double c0 = sol ? -b0 : b0, c1 = sol ? -b1 : b1;
        // Q = [u v u x v] with u = (r00,r10,c0), v = (r01,r11,c1)
        const double Q[9] = {r00, r01, c1 * r10 - c0 * r11,
                             r10, r11, c0 * r01 - c1 * r00,
                             c0,  c1,  r00 * r11 - r01 * r10};
        mat3mul(Rv, Q, sol ? Rb : Ra);
    }
}

// ippe.cpp:347-425: least-squares translation for a fixed rotation (normal equations of x*(r_z + t_z) = r_x + t_x, ...)
void ippe_translation(const float model[4][3], const float *img, const double R[9], double t[3]) {
    double Sa = 0, Sb = 0, Sq = 0, B0 = 0, B1 = 0, B2 = 0;
    const double n = 4;
    for (int i = 0; i < 4; i++) {
        const double X = model[i][0], Y = model[i][1], Z = model[i][2];
        const double rx = R[0] * X + R[1] * Y + R[2] * Z, ry = R[3] * X + R[4] * Y + R[5] * Z, rz = R[6] * X + R[7] * Y + R[8] * Z;
        const double a = -(double)img[2 * i], b = -(double)img[2 * i + 1];
        Sa += a; Sb += b; Sq += a * a + b * b;
        const double bx = (double)img[2 * i] * rz - rx, by = (double)img[2 * i + 1] * rz - ry;
        B0 += bx; B1 += by; B2 += a * bx + b * by;
    }
    // [n 0 Sa; 0 n Sb; Sa Sb Sq] t = B, by the adjugate as ippe.cpp:407-424
    const double dinv = 1.0 / (n * n * Sq - n * Sb * Sb - Sa * n * Sa);
    t[0] = dinv * ((n * Sq - Sb * Sb) * B0 + (Sa * Sb) * B1 + (-Sa * n) * B2);
    t[1] = dinv * ((Sb * Sa) * B0 + (n * Sq - Sa * Sa) * B1 + (-n * Sb) * B2);
    t[2] = dinv * ((-n * Sa) * B0 + (-n * Sb) * B1 + (n * n) * B2);
}

// ippe.cpp:289-321: float arithmetic, each product rounded on its own
float ippe_reproj_error(const double R[9], const double t[3], const float model[4][3], const float *img) {
    float err = 0;
    for (int i = 0; i < 4; i++) {
        const float px = static_cast<float>(R[0] * model[i][0]) + static_cast<float>(R[1] * model[i][1]) +
                         static_cast<float>(R[2] * model[i][2] + t[0]);
        const float py = static_cast<float>(R[3] * model[i][0]) + static_cast<float>(R[4] * model[i][1]) +
                         static_cast<float>(R[5] * model[i][2] + t[1]);
        const float pz = static_cast<float>(R[6] * model[i][0]) + static_cast<float>(R[7] * model[i][1]) +
                         static_cast<float>(R[8] * model[i][2] + t[2]);
        const float dx = px / pz - img[2 * i], dy = py / pz - img[2 * i + 1];
        err = err + std::sqrt(dx * dx + dy * dy);
    }
    return err;
}

// ippe.cpp:323-345 then getRTMatrix's cv::Rodrigues + CV_32F conversion (ippe.cpp:40-93)
void rt_matrix_f32(const double R[9], const double t[3], double T[16]) {
    const double tr = R[0] + R[4] + R[8];
    const double w = std::acos((tr - 1.0) / 2.0);
    double rv[3] = {0, 0, 0};
    if (!(w < std::numeric_limits<double>::epsilon())) {
        const double d = 1 / (2 * std::sin(w)) * w;
        rv[0] = d * (R[7] - R[5]); rv[1] = d * (R[2] - R[6]); rv[2] = d * (R[3] - R[1]);
    }
    // cv::Rodrigues, vector -> matrix
    double M[9];
    const double th = std::sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
    if (th < DBL_EPSILON) {
        for (int i = 0; i < 9; i++) M[i] = (i % 4 == 0) ? 1.0 : 0.0;
    } else {
        const double c = std::cos(th), s = std::sin(th), c1 = 1. - c, ith = 1. / th;
        const double x = rv[0] * ith, y = rv[1] * ith, z = rv[2] * ith;
        M[0] = c + c1 * x * x;     M[1] = c1 * x * y - s * z; M[2] = c1 * x * z + s * y;
        M[3] = c1 * x * y + s * z; M[4] = c + c1 * y * y;     M[5] = c1 * y * z - s * x;
        M[6] = c1 * x * z - s * y; M[7] = c1 * y * z + s * x; M[8] = c + c1 * z * z;
    }
    for (int i = 0; i < 16; i++) T[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) T[i * 4 + j] = (double)(float)M[i * 3 + j];
        T[i * 4 + 3] = (double)(float)t[i];
    }
}

struct Pose {
    M4 T;
    double err;
};
typedef std::map<int, std::map<int, std::vector<Pose>>> PoseMap;        // [outer][inner] -> candidates
typedef std::tuple<M4, M4, M4, double> Cand;                             // (T, T1_inv, T2_inv, error)
typedef std::map<int, std::map<int, std::vector<Cand>>> CandSets;
typedef std::map<int, std::map<int, std::pair<M4, double>>> BestMap;

int64_t vote(double marker_size, const std::vector<Cand> &sol, double *cost, double &weight) {
    const double h = marker_size / 2;
    M4 P;  // columns = corners (libs/initializer.cpp:152-169)
    const double px[4] = {-h, h, h, -h}, py[4] = {h, h, -h, -h};
    for (int c = 0; c < 4; c++) {
        P.a[0 * 4 + c] = px[c]; P.a[1 * 4 + c] = py[c]; P.a[2 * 4 + c] = 0; P.a[3 * 4 + c] = 1;
    }
    double min_error = std::numeric_limits<double>::max();
    int64_t min_index = -1;
    for (size_t i = 0; i < sol.size(); i++) {
        const M4 &T = std::get<0>(sol[i]);
        double curr = 0;
        for (size_t j = 0; j < sol.size(); j++) {
            const M4 p2 = mul(mul(mul(std::get<2>(sol[j]), T), std::get<1>(sol[j])), P);
            double s = 0;
            for (int c = 0; c < 4; c++) {
                double q = 0;
                for (int r = 0; r < 3; r++) {
                    const double d = P.a[r * 4 + c] - p2.a[r * 4 + c];
                    q += d * d;
                }
                s += std::sqrt(q);
            }
            curr += s;
        }
        if (cost) cost[i] = curr;
        if (curr < min_error) {
            min_index = (int64_t)i;
            min_error = curr;
            weight = min_error;
        }
    }
    return min_index;
}

struct InitState {
    std::set<int> cam_ids, marker_ids;
    std::map<int, bool> kept_frames;  // keys of frame_cam_markers
    int root_cam = -1, root_marker = -1;
    std::map<int, M4> to_root_cam, to_root_marker, object_transforms;
};

// libs/initializer.cpp:95-125
void fill_sets(bool camera, const PoseMap &est, CandSets &sets) {
    for (auto it = est.begin(); it != est.end(); ++it) {
        const auto &objects = it->second;
        if (objects.size() > 1)
            for (auto it1 = objects.begin(); it1 != objects.end(); ++it1)
                for (size_t i = 0; i < it1->second.size(); i++)
                    for (auto it2 = std::next(it1); it2 != objects.end(); ++it2)
                        for (size_t j = 0; j < it2->second.size(); j++) {
                            const Pose &p1 = it1->second[i], &p2 = it2->second[j];
                            if (camera)
                                sets[it1->first][it2->first].push_back(
                                    std::make_tuple(mul(p2.T, inv(p1.T)), p1.T, inv(p2.T), p1.err * p2.err));
                            else
                                sets[it1->first][it2->first].push_back(
                                    std::make_tuple(mul(inv(p2.T), p1.T), inv(p1.T), p2.T, p1.err * p2.err));
                        }
    }
}

struct Node {
    int id;
    mutable double distance;
    mutable int parent;
    bool operator<(const Node &n) const { return id < n.id; }
};

// libs/initializer.cpp:237-289: Prim's tree grown from the root; edge weight = the vote's summed error
void make_mst(int start, const std::set<int> &ids, const BestMap &adj, std::map<int, std::set<int>> &children) {
    std::set<Node> outside;
    for (int id : ids) outside.insert(Node{id, id == start ? 0.0 : std::numeric_limits<double>::max(), -1});
    while (!outside.empty()) {
        auto mn = outside.begin();
        for (auto it = outside.begin(); it != outside.end(); ++it)
            if (it->distance < mn->distance) mn = it;
        for (auto it = outside.begin(); it != outside.end(); ++it) {
            const int lo = mn->id < it->id ? mn->id : it->id, hi = mn->id < it->id ? it->id : mn->id;
            auto a = adj.find(lo);
            if (a == adj.end()) continue;
            auto b = a->second.find(hi);
            if (b == a->second.end()) continue;
            const double error = b->second.second;
            if (error < it->distance) {
                it->distance = error;
                if (it->parent != -1) children[it->parent].erase(it->id);
                children[mn->id].insert(it->id);
                it->parent = mn->id;
            }
        }
        outside.erase(mn);
    }
}

// libs/initializer.cpp:291-315
void transforms_to_root(int root, const std::map<int, std::set<int>> &children, const BestMap &best, std::map<int, M4> &out) {
    out[root] = eye4();
    std::queue<int> q;
    q.push(root);
    while (!q.empty()) {
        const int parent = q.front();
        auto ch = children.find(parent);
        if (ch != children.end())
            for (int child : ch->second) {
                if (child < parent)
                    out[child] = best.at(child).at(parent).first;
                else
                    out[child] = inv(best.at(parent).at(child).first);
                if (parent != root) out[child] = mul(out[parent], out[child]);
                q.push(child);
            }
        q.pop();
    }
}

}  // namespace

extern "C" {

void orc_inv4(const double *A, double *Ainv) {
    M4 m;
    std::memcpy(m.a, A, sizeof m.a);
    const M4 r = inv(m);
    std::memcpy(Ainv, r.a, sizeof r.a);
}

void orc_undistort_normalized(const orc_cam_model *cam, int64_t n, const float *in, float *out) {
    const double *K = cam->K;
    double k[12];
    for (int i = 0; i < 12; i++) k[i] = (i < cam->n_dist) ? cam->dist[i] : 0.0;
    const double ifx = 1.0 / K[0], ify = 1.0 / K[4];
    for (int64_t i = 0; i < n; i++) {
        double x = ((double)in[2 * i] - K[2]) * ifx, y = ((double)in[2 * i + 1] - K[5]) * ify;
        const double x0 = x, y0 = y;
        for (int it = 0; it < 5; it++) {
            const double r2 = x * x + y * y;
            const double icdist = (1.0 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1.0 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            const double dx = 2.0 * k[2] * x * y + k[3] * (r2 + 2.0 * x * x) + k[8] * r2 + k[9] * r2 * r2;
            const double dy = k[2] * (r2 + 2.0 * y * y) + 2.0 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
            x = (x0 - dx) * icdist;
            y = (y0 - dy) * icdist;
        }
        out[2 * i] = (float)x;
        out[2 * i + 1] = (float)y;
    }
}

void orc_ippe_square(float marker_size, const orc_cam_model *cam, int64_t n, const float *uv, double *T1, double *e1,
                     double *T2, double *e2) {
    const float hf = marker_size / 2.0f;
    const float model[4][3] = {{-hf, hf, 0}, {hf, hf, 0}, {hf, -hf, 0}, {-hf, -hf, 0}};
    for (int64_t d = 0; d < n; d++) {
        float q[8];
        orc_undistort_normalized(cam, 4, uv + 8 * d, q);
        double H[9];
        square_homography(q, (double)hf, H);
        const double J[4] = {H[0] - H[6] * H[2], H[1] - H[7] * H[2], H[3] - H[6] * H[5], H[4] - H[7] * H[5]};
        double Ra[9], Rb[9], ta[3], tb[3];
        ippe_rotations(J, H[2], H[5], Ra, Rb);
        ippe_translation(model, q, Ra, ta);
        ippe_translation(model, q, Rb, tb);
        const float ea = ippe_reproj_error(Ra, ta, model, q), eb = ippe_reproj_error(Rb, tb, model, q);
        if (ea < eb) {
            rt_matrix_f32(Ra, ta, T1 + 16 * d); rt_matrix_f32(Rb, tb, T2 + 16 * d);
            e1[d] = ea; e2[d] = eb;
        } else {
            rt_matrix_f32(Rb, tb, T1 + 16 * d); rt_matrix_f32(Ra, ta, T2 + 16 * d);
            e1[d] = eb; e2[d] = ea;
        }
    }
}

int64_t orc_vote(double marker_size, int64_t n, const double *T, const double *T1inv, const double *T2inv, double *cost,
                 double *weight) {
    std::vector<Cand> sol((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        M4 a, b, c;
        std::memcpy(a.a, T + 16 * i, sizeof a.a);
        std::memcpy(b.a, T1inv + 16 * i, sizeof b.a);
        std::memcpy(c.a, T2inv + 16 * i, sizeof c.a);
        sol[i] = std::make_tuple(a, b, c, 0.0);
    }
    double w = 0;
    const int64_t best = vote(marker_size, sol, cost, w);
    if (weight) *weight = w;
    return best;
}

struct FixedTransforms {   // apps/track.cpp:70-89: set_transforms_to_root_cam / _marker instead of init_transforms
    std::map<int, M4> cam, marker;
};

static void *init_impl(int32_t num_cam_slots, int32_t num_frames, int64_t n_det, const int32_t *det_frame,
                       const int32_t *det_cam, const int32_t *det_id, const float *det_uv, double marker_size,
                       const orc_cam_model *cams, const int32_t *excluded, int32_t n_excluded, double threshold,
                       int32_t min_detections, const FixedTransforms *fixed) {
    InitState *st = new InitState;
    std::set<int> excl(excluded, excluded + n_excluded);
    // detections[frame][cam] = indices into the flat arrays, in file order
    std::vector<std::vector<std::vector<int64_t>>> det(num_frames, std::vector<std::vector<int64_t>>(num_cam_slots));
    for (int64_t i = 0; i < n_det; i++) det[det_frame[i]][det_cam[i]].push_back(i);

    // ---- obtain_pose_estimations (libs/initializer.cpp:364-418) ----
    std::map<int, PoseMap> frame_poses_cam, frame_poses_marker;
    for (int f = 0; f < num_frames; f++) {
        int num_detections = 0;
        for (int c = 0; c < num_cam_slots; c++)
            if (!excl.count(c)) num_detections += (int)det[f][c].size();
        if (!(num_detections >= min_detections)) continue;
        PoseMap est_marker, est_cam;
        for (int c = 0; c < num_cam_slots; c++) {
            if (excl.count(c) || det[f][c].empty()) continue;
            st->cam_ids.insert(c);
            st->kept_frames[f] = true;
            for (int64_t i : det[f][c]) {
                const int id = det_id[i];
                st->marker_ids.insert(id);
                Pose s0, s1;
                orc_ippe_square((float)marker_size, &cams[c], 1, det_uv + 8 * i, s0.T.a, &s0.err, s1.T.a, &s1.err);
                est_cam[id][c].push_back(s0);
                est_marker[c][id].push_back(s0);
                if (s1.err / s0.err < threshold) {
                    est_cam[id][c].push_back(s1);
                    est_marker[c][id].push_back(s1);
                }
            }
        }
        frame_poses_cam[f] = est_cam;
        frame_poses_marker[f] = est_marker;
    }

    if (fixed) {
        st->to_root_cam = fixed->cam;
        st->to_root_marker = fixed->marker;
    }
    // ---- init_transforms_cam / init_transforms_marker (libs/initializer.cpp:421-451) ----
    for (int pass = 0; pass < 2 && !fixed; pass++) {
        const bool camera = pass == 0;
        std::map<int, PoseMap> &poses = camera ? frame_poses_cam : frame_poses_marker;
        CandSets sets;
        for (int f = 0; f < num_frames; f++) fill_sets(camera, poses[f], sets);  // operator[]: empty frames appear
        BestMap best;
        for (auto &a : sets)
            for (auto &b : a.second) {
                double w = 0;
                const int64_t idx = vote(marker_size, b.second, nullptr, w);
                if (idx < 0) continue;  // all-NaN set: the reference indexes solutions[-1] here
                best[a.first][b.first] = std::make_pair(std::get<0>(b.second[idx]), w);
            }
        const std::set<int> &ids = camera ? st->cam_ids : st->marker_ids;
        if (ids.empty()) continue;
        const int root = *ids.begin();
        (camera ? st->root_cam : st->root_marker) = root;
        std::map<int, std::set<int>> tree;
        make_mst(root, ids, best, tree);
        transforms_to_root(root, tree, best, camera ? st->to_root_cam : st->to_root_marker);
    }

    // ---- init_object_transforms (libs/initializer.cpp:453-465) with fill_transformation_set (:73-93) ----
    for (auto &fr : frame_poses_cam) {
        std::vector<Cand> set;
        for (auto &mk : fr.second)
            for (auto &cm : mk.second) {
                M4 T_mr = eye4(), T_rm = eye4(), T_cr = eye4(), T_rc = eye4();
                auto im = st->to_root_marker.find(mk.first);
                if (im != st->to_root_marker.end()) { T_mr = im->second; T_rm = inv(T_mr); }
                auto ic = st->to_root_cam.find(cm.first);
                if (ic != st->to_root_cam.end()) { T_cr = ic->second; T_rc = inv(T_cr); }
                for (const Pose &p : cm.second) {
                    const M4 T_cm = inv(p.T);
                    set.push_back(std::make_tuple(mul(mul(T_cr, p.T), T_rm), mul(T_mr, T_cm), T_rc, p.err));
                }
            }
        double w = 0;
        const int64_t idx = vote(marker_size, set, nullptr, w);
        if (idx >= 0) st->object_transforms[fr.first] = std::get<0>(set[idx]);
    }
    return st;
}

void *orc_init_run(int32_t num_cam_slots, int32_t num_frames, int64_t n_det, const int32_t *det_frame,
                   const int32_t *det_cam, const int32_t *det_id, const float *det_uv, double marker_size,
                   const orc_cam_model *cams, const int32_t *excluded, int32_t n_excluded, double threshold,
                   int32_t min_detections) {
    return init_impl(num_cam_slots, num_frames, n_det, det_frame, det_cam, det_id, det_uv, marker_size, cams, excluded, n_excluded,
                     threshold, min_detections, nullptr);
}

void *orc_init_object_poses(int32_t num_cam_slots, int32_t num_frames, int64_t n_det, const int32_t *det_frame,
                            const int32_t *det_cam, const int32_t *det_id, const float *det_uv, double marker_size,
                            const orc_cam_model *cams, int32_t n_fixed_cams, const int32_t *cam_ids, const double *T_cam,
                            int32_t n_fixed_markers, const int32_t *marker_ids, const double *T_marker, double threshold,
                            int32_t min_detections) {
    FixedTransforms fx;
    for (int i = 0; i < n_fixed_cams; i++) std::memcpy(fx.cam[cam_ids[i]].a, T_cam + 16 * i, sizeof(double) * 16);
    for (int i = 0; i < n_fixed_markers; i++) std::memcpy(fx.marker[marker_ids[i]].a, T_marker + 16 * i, sizeof(double) * 16);
    return init_impl(num_cam_slots, num_frames, n_det, det_frame, det_cam, det_id, det_uv, marker_size, cams, nullptr, 0, threshold,
                     min_detections, &fx);
}

void orc_init_counts(const void *h, int32_t counts[6]) {
    const InitState *st = static_cast<const InitState *>(h);
    counts[0] = (int32_t)st->to_root_cam.size();
    counts[1] = (int32_t)st->to_root_marker.size();
    counts[2] = (int32_t)st->object_transforms.size();
    counts[3] = (int32_t)st->kept_frames.size();
    counts[4] = st->root_cam;
    counts[5] = st->root_marker;
}

void orc_init_get(const void *h, int32_t *cam_ids, double *T_cam, int32_t *marker_ids, double *T_marker, int32_t *frame_ids,
                  double *T_object, int32_t *kept_frame_ids) {
    const InitState *st = static_cast<const InitState *>(h);
    int i = 0;
    for (auto &e : st->to_root_cam) { cam_ids[i] = e.first; std::memcpy(T_cam + 16 * i, e.second.a, sizeof e.second.a); i++; }
    i = 0;
    for (auto &e : st->to_root_marker) { marker_ids[i] = e.first; std::memcpy(T_marker + 16 * i, e.second.a, sizeof e.second.a); i++; }
    i = 0;
    for (auto &e : st->object_transforms) { frame_ids[i] = e.first; std::memcpy(T_object + 16 * i, e.second.a, sizeof e.second.a); i++; }
    i = 0;
    if (kept_frame_ids)
        for (auto &e : st->kept_frames) kept_frame_ids[i++] = e.first;
}

void orc_init_free(void *h) { delete static_cast<InitState *>(h); }

}  // extern "C"
