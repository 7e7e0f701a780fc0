// Reader for the camera calibration files of the reference (<dir>/<cam>/calib.{xml,yml,yaml}): the four keys
// CamConfig::read_from_file pulls out of a cv::FileStorage (libs/cam_config.cpp:52-80), in the XML and YAML 1.0 dialects
// cv::FileStorage writes.  Not a general FileStorage parser: scalars and `opencv-matrix` nodes of doubles / floats only.
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "internal.h"

namespace {

bool read_file(const char *path, std::string &out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::ostringstream ss;
    ss << f.rdbuf();
    out = ss.str();
    return true;
}

// numbers in a cv::FileStorage data list: separated by blanks, commas, newlines; "1." and ".5" and "1e+03" all occur
std::vector<double> parse_numbers(const std::string &t) {
    std::vector<double> v;
    const char *p = t.c_str(), *end = p + t.size();
    while (p < end) {
        while (p < end && !(isdigit((unsigned char)*p) || *p == '-' || *p == '+' || *p == '.')) p++;
        if (p >= end) break;
        char *q = nullptr;
        const double x = strtod(p, &q);
        if (q == p) { p++; continue; }
        v.push_back(x);
        p = q;
    }
    return v;
}

// ---- XML dialect: <key>scalar</key>, <key type_id="opencv-matrix"><rows>..</rows><cols>..</cols><dt>..</dt><data>..</data></key>
bool xml_node(const std::string &doc, const std::string &key, std::string &body) {
    const std::string open = "<" + key;
    size_t a = 0;
    while ((a = doc.find(open, a)) != std::string::npos) {
        const char c = doc[a + open.size()];
        if (c == '>' || c == ' ' || c == '\t' || c == '\n' || c == '\r') break;
        a += open.size();
    }
    if (a == std::string::npos) return false;
    const size_t b = doc.find('>', a);
    const size_t e = doc.find("</" + key + ">", b);
    if (b == std::string::npos || e == std::string::npos) return false;
    body = doc.substr(b + 1, e - b - 1);
    return true;
}

bool xml_matrix(const std::string &doc, const std::string &key, int &rows, int &cols, std::vector<double> &data) {
    std::string node, t;
    if (!xml_node(doc, key, node)) return false;
    if (!xml_node(node, "rows", t)) return false;
    rows = atoi(t.c_str());
    if (!xml_node(node, "cols", t)) return false;
    cols = atoi(t.c_str());
    if (!xml_node(node, "data", t)) return false;
    data = parse_numbers(t);
    return (int)data.size() == rows * cols;
}

// ---- YAML 1.0 dialect: "key: scalar", "key: !!opencv-matrix\n   rows: r\n   cols: c\n   dt: d\n   data: [ .. ]" (data may wrap)
size_t yaml_key(const std::string &doc, const std::string &key) {
    size_t a = 0;
    const std::string k = key + ":";
    while ((a = doc.find(k, a)) != std::string::npos) {
        if (a == 0 || doc[a - 1] == '\n' || doc[a - 1] == ' ' || doc[a - 1] == '\t') return a + k.size();
        a += k.size();
    }
    return std::string::npos;
}

bool yaml_scalar(const std::string &doc, const std::string &key, double &v) {
    const size_t a = yaml_key(doc, key);
    if (a == std::string::npos) return false;
    const size_t e = doc.find('\n', a);
    const std::vector<double> n = parse_numbers(doc.substr(a, e == std::string::npos ? std::string::npos : e - a));
    if (n.empty()) return false;
    v = n[0];
    return true;
}

bool yaml_matrix(const std::string &doc, const std::string &key, int &rows, int &cols, std::vector<double> &data) {
    const size_t a = yaml_key(doc, key);
    if (a == std::string::npos) return false;
    const std::string rest = doc.substr(a);
    double r = 0, c = 0;
    if (!yaml_scalar(rest, "rows", r) || !yaml_scalar(rest, "cols", c)) return false;
    rows = (int)r;
    cols = (int)c;
    const size_t d = yaml_key(rest, "data");
    if (d == std::string::npos) return false;
    const size_t lb = rest.find('[', d), rb = rest.find(']', d);
    if (lb == std::string::npos || rb == std::string::npos || rb < lb) return false;
    data = parse_numbers(rest.substr(lb + 1, rb - lb - 1));
    return (int)data.size() == rows * cols;
}

}  // namespace

extern "C" int aar_cam_config_read(const char *path, double K[9], double dist[AAR_MAX_DIST], int32_t *n_dist, int32_t *width,
                                   int32_t *height) {
    if (!path || !K || !dist || !n_dist || !width || !height) return aar::set_error(AAR_ERR_INVALID, "aar_cam_config_read: null argument");
    std::string doc;
    if (!read_file(path, doc)) return aar::set_error(AAR_ERR_IO, "cannot open %s", path);
    const bool xml = doc.find("<opencv_storage") != std::string::npos;
    int r = 0, c = 0;
    std::vector<double> km, dm;
    double w = 0, h = 0;
    bool ok;
    if (xml) {
        std::string t;
        ok = xml_node(doc, "image_width", t) && !(parse_numbers(t).empty());
        if (ok) w = parse_numbers(t)[0];
        ok = ok && xml_node(doc, "image_height", t) && !(parse_numbers(t).empty());
        if (ok) h = parse_numbers(t)[0];
        ok = ok && xml_matrix(doc, "camera_matrix", r, c, km) && r == 3 && c == 3;
        int dr = 0, dc = 0;
        ok = ok && xml_matrix(doc, "distortion_coefficients", dr, dc, dm);
    } else {
        ok = yaml_scalar(doc, "image_width", w) && yaml_scalar(doc, "image_height", h);
        ok = ok && yaml_matrix(doc, "camera_matrix", r, c, km) && r == 3 && c == 3;
        int dr = 0, dc = 0;
        ok = ok && yaml_matrix(doc, "distortion_coefficients", dr, dc, dm);
    }
    // the reference refuses a file that lacks any of the four keys (libs/cam_config.cpp:57-77)
    if (!ok) return aar::set_error(AAR_ERR_IO, "%s: image_width, image_height, camera_matrix (3x3) and distortion_coefficients are all required", path);
    if ((int)dm.size() > AAR_MAX_DIST) return aar::set_error(AAR_ERR_UNSUPPORTED, "%s: %d distortion coefficients (at most %d: no tilt model)", path, (int)dm.size(), AAR_MAX_DIST);
    for (int i = 0; i < 9; i++) K[i] = km[i];
    for (int i = 0; i < AAR_MAX_DIST; i++) dist[i] = i < (int)dm.size() ? dm[i] : 0.0;
    *n_dist = (int32_t)dm.size();
    *width = (int32_t)w;
    *height = (int32_t)h;
    return AAR_OK;
}
