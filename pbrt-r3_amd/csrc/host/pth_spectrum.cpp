// pth_spectrum.cpp -- see pth_spectrum.h
#include "pth_spectrum.h"

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <sstream>

namespace pth {

namespace {
const int kSpectralSamples = 60;             // spectrum_config.rs
const float kLambdaStart = 400.0f, kLambdaEnd = 700.0f;

float lerp(float t, float a, float b) { return (1.0f - t) * a + t * b; }

bool samples_sorted(const std::vector<float>& l) {
    for (size_t i = 0; i + 1 < l.size(); i++)
        if (l[i] > l[i + 1]) return false;
    return true;
}
void sort_samples(std::vector<float>& l, std::vector<float>& v) {        // utils.rs:14-25 (stable)
    std::vector<std::pair<float, float>> p(l.size());
    for (size_t i = 0; i < l.size(); i++) p[i] = {l[i], v[i]};
    std::stable_sort(p.begin(), p.end(), [](const std::pair<float, float>& a, const std::pair<float, float>& b) { return a.first < b.first; });
    for (size_t i = 0; i < l.size(); i++) { l[i] = p[i].first; v[i] = p[i].second; }
}
// base/functions.rs:105-120 with the predicate v[index] <= l
size_t find_interval_le(const std::vector<float>& v, float l) {
    long first = 0, len = (long)v.size();
    while (len > 0) {
        long half = len >> 1, middle = first + half;
        if (v[(size_t)middle] <= l) { first = middle + 1; len -= half + 1; }
        else len = half;
    }
    long r = first - 1, hi = (long)v.size() - 2;
    return (size_t)(r < 0 ? 0 : (r > hi ? hi : r));
}
float interpolate_samples(const std::vector<float>& lambda, const std::vector<float>& vals, float l) {   // utils.rs:27-41
    size_t n = lambda.size();
    if (l <= lambda[0]) return vals[0];
    if (l >= lambda[n - 1]) return vals[n - 1];
    size_t o = find_interval_le(lambda, l);
    float t = (l - lambda[o]) / (lambda[o + 1] - lambda[o]);
    return lerp(t, vals[o], vals[o + 1]);
}
// build/spectrum/utils.rs:8-63
float average_samples(const std::vector<float>& lambda, const std::vector<float>& vals, float ls, float le) {
    size_t n = lambda.size();
    if (le <= lambda[0]) return vals[0];
    if (ls >= lambda[n - 1]) return vals[n - 1];
    if (n == 1) return vals[0];
    float sum = 0.0f;
    if (ls < lambda[0]) sum += vals[0] * (lambda[0] - ls);
    if (le > lambda[n - 1]) sum += vals[n - 1] * (le - lambda[n - 1]);
    size_t i = 0;
    while (ls > lambda[i + 1]) i++;
    auto interp = [&](float w, size_t k) { return lerp((w - lambda[k]) / (lambda[k + 1] - lambda[k]), vals[k], vals[k + 1]); };
    while (i + 1 < n && le >= lambda[i]) {
        float s0 = std::fmax(ls, lambda[i]), s1 = std::fmin(le, lambda[i + 1]);
        sum += 0.5f * (interp(s0, i) + interp(s1, i)) * (s1 - s0);
        i++;
    }
    return sum / (le - ls);
}
void sample_spectrum(const std::vector<float>& lambda, const std::vector<float>& vals, float out[kSpectralSamples]) {    // utils.rs:65-81
    for (int i = 0; i < kSpectralSamples; i++) {
        float wl0 = lerp((float)i / (float)kSpectralSamples, kLambdaStart, kLambdaEnd);
        float wl1 = lerp((float)(i + 1) / (float)kSpectralSamples, kLambdaStart, kLambdaEnd);
        out[i] = average_samples(lambda, vals, wl0, wl1);
    }
}
void xyz_to_rgb(const float xyz[3], float rgb[3]) {            // core/spectrum/convert.rs:3-9
    rgb[0] = 3.240479f * xyz[0] - 1.537150f * xyz[1] - 0.498535f * xyz[2];
    rgb[1] = -0.969256f * xyz[0] + 1.875991f * xyz[1] + 0.041556f * xyz[2];
    rgb[2] = 0.055648f * xyz[0] - 0.204043f * xyz[1] + 1.057311f * xyz[2];
}
// Rust's str::parse::<f32>: decimal floats, "inf"/"nan"; anything else is a parse error
bool parse_f32(const std::string& tok, float* out) {
    if (tok.empty()) return false;
    for (char c : tok)
        if (c == 'x' || c == 'X' || c == 'p' || c == 'P') return false;          // no hex floats
    char* end = nullptr;
    float v = std::strtof(tok.c_str(), &end);
    if (end == tok.c_str() || *end != '\0') return false;
    *out = v;
    return true;
}
}  // namespace

bool SpectrumTables::load(const std::string& dir, std::string* err) {
    std::string path = dir + "/spectrum_tables.bin";
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) { if (err) *err = "cannot read " + path; return false; }
    char magic[8];
    uint32_t n = 0, m = 0;
    auto rd = [&](std::vector<float>& v, uint32_t k) { v.resize(k); return std::fread(v.data(), 4, k, f) == k; };
    bool ok = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, "PTSPECT1", 8) == 0 && std::fread(&n, 4, 1, f) == 1 && n >= 2 && n < 100000 &&
              rd(cie_x, n) && rd(cie_y, n) && rd(cie_z, n) && rd(cie_lambda, n) && std::fread(&cie_y_integral, 4, 1, f) == 1 &&
              std::fread(&m, 4, 1, f) == 1 && m >= 2 && m < 100000 && rd(cu_lambda, m) && rd(cu_n, m) && rd(cu_k, m);
    std::fclose(f);
    if (!ok) { if (err) *err = "malformed " + path; return false; }
    // ARRAY_CIE_X/Y/Z of build/spectrum/build_xyz.rs:10-26
    sx.resize(kSpectralSamples); sy.resize(kSpectralSamples); sz.resize(kSpectralSamples);
    sample_spectrum(cie_lambda, cie_x, sx.data());
    sample_spectrum(cie_lambda, cie_y, sy.data());
    sample_spectrum(cie_lambda, cie_z, sz.data());
    return true;
}

const SpectrumTables* spectrum_tables(std::string* err) {
    static SpectrumTables tables;
    static bool loaded = false, failed = false;
    static std::string fail_msg;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (loaded) return &tables;
    if (failed) { if (err) *err = fail_msg; return nullptr; }
    std::string dir;
    if (const char* e = std::getenv("PBRTGPU_DATA_DIR")) dir = e;
    else {                                   // <directory of this shared library>/../data
        Dl_info di;
        if (dladdr((const void*)&spectrum_tables, &di) && di.dli_fname) {
            std::string so = di.dli_fname;
            size_t k = so.find_last_of('/');
            dir = (k == std::string::npos ? std::string(".") : so.substr(0, k)) + "/../data";
        }
    }
    if (!tables.load(dir, &fail_msg)) { failed = true; if (err) *err = fail_msg; return nullptr; }
    loaded = true;
    return &tables;
}

void rgb_from_sampled(const SpectrumTables& T, std::vector<float> lambda, std::vector<float> vals, float rgb[3]) {
    if (!samples_sorted(lambda)) sort_samples(lambda, vals);
    float xyz[3] = {0.0f, 0.0f, 0.0f};
    const size_t n = T.cie_lambda.size();
    for (size_t i = 0; i < n; i++) {
        float val = interpolate_samples(lambda, vals, T.cie_lambda[i]);
        xyz[0] += val * T.cie_x[i];
        xyz[1] += val * T.cie_y[i];
        xyz[2] += val * T.cie_z[i];
    }
    float scale = (T.cie_lambda[n - 1] - T.cie_lambda[0]) / (T.cie_y_integral * (float)n);
    xyz[0] *= scale; xyz[1] *= scale; xyz[2] *= scale;
    xyz_to_rgb(xyz, rgb);
}

namespace {
void sampled_to_rgb(const SpectrumTables& T, const float c[kSpectralSamples], float rgb[3]) {     // sampled.rs:53-67, :89-92
    float xyz[3] = {0.0f, 0.0f, 0.0f};
    for (int i = 0; i < kSpectralSamples; i++) {
        xyz[0] += T.sx[i] * c[i];
        xyz[1] += T.sy[i] * c[i];
        xyz[2] += T.sz[i] * c[i];
    }
    float scale = (kLambdaEnd - kLambdaStart) / (T.cie_y_integral * (float)kSpectralSamples);
    xyz[0] *= scale; xyz[1] *= scale; xyz[2] *= scale;
    xyz_to_rgb(xyz, rgb);
}
double planck(double lambda_nm, double t) {                  // blackbody.rs:5-22
    const double C = 299792458.0, H = 6.62606957e-34, KB = 1.3806488e-23;
    double l = lambda_nm * 1e-9;
    double lambda5 = (l * l) * (l * l) * l;
    return (2.0 * H * C * C) / (lambda5 * (std::exp((H * C) / (l * KB * t)) - 1.0));
}
}  // namespace
double planck_law(double lambda_nm, double t) { return planck(lambda_nm, t); }

void rgb_from_blackbody(const SpectrumTables& T, const std::vector<float>& values, float rgb[3]) {
    float s[kSpectralSamples];
    for (int i = 0; i < kSpectralSamples; i++) s[i] = 0.0f;
    const size_t n = T.cie_lambda.size();
    for (size_t k = 0; k + 1 < values.size(); k += 2) {
        double t = (double)values[k];
        std::vector<float> le(n, 0.0f);
        if (t > 0.0) {                                        // blackbody_normalized (blackbody.rs:24-36)
            double max_l = planck(2.8977721e-3 / t * 1e9, t);
            for (size_t i = 0; i < n; i++) le[i] = (float)(planck((double)T.cie_lambda[i], t) / max_l);
        }
        float c[kSpectralSamples];
        sample_spectrum(T.cie_lambda, le, c);
        for (int i = 0; i < kSpectralSamples; i++) s[i] += c[i] * values[k + 1];
    }
    sampled_to_rgb(T, s, rgb);
}

bool rgb_from_spd_file(const SpectrumTables& T, const std::string& path, float rgb[3], std::string* err) {
    std::ifstream in(path);
    if (!in) { if (err) *err = "Unable to open file \"" + path + "\"."; return false; }
    std::vector<float> values;
    std::string line;
    while (std::getline(in, line)) {
        if (line.find('#') != std::string::npos) continue;        // float_file.rs:15: a line with a '#' anywhere is dropped whole
        std::istringstream ls(line);
        std::string tok;
        while (ls >> tok) {
            float v = 0.0f;
            if (!parse_f32(tok, &v)) v = 0.0f;                     // "Unexpected text": the reference warns and takes 0
            values.push_back(v);
        }
    }
    std::vector<float> wl, v;
    for (size_t j = 0; j + 1 < values.size(); j += 2) { wl.push_back(values[j]); v.push_back(values[j + 1]); }
    if (wl.size() < 1) { if (err) *err = "spectrum file \"" + path + "\" holds no samples"; return false; }
    if (!samples_sorted(wl)) sort_samples(wl, v);
    float c[kSpectralSamples];
    sample_spectrum(wl, v, c);
    sampled_to_rgb(T, c, rgb);
    return true;
}

}  // namespace pth

extern "C" void pth_blackbody(const double* lambda_nm, int n, double t_kelvin, double* out) {      // blackbody.rs:3-22
    for (int i = 0; i < n; i++) out[i] = t_kelvin <= 0.0 ? 0.0 : pth::planck_law(lambda_nm[i], t_kelvin);
}
