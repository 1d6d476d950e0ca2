// pbrt_gpu -- command-line front end mirroring `pbrt-r3 -i scene.pbrt` (src/bin/pbrt.rs:44-132,
// :263-356): parse the scene description, hand the flattened scene to the MI355X library, write
// the image.  Options follow the reference's names where they exist.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <dlfcn.h>

#include "../../../include/pbrtgpu_host.h"

static bool write_floats(const std::string& path, const std::vector<float>& v) {
    FILE* f = std::fopen(path.c_str(), "wb");
    const bool ok = f && std::fwrite(v.data(), 4, v.size(), f) == v.size();
    if (f) std::fclose(f);
    if (!ok) std::fprintf(stderr, "pbrt_gpu: cannot write %s\n", path.c_str());
    return ok;
}

static void usage() {
    std::fprintf(stderr,
                 "usage: pbrt_gpu [-i] scene.pbrt [options]\n"
                 "  Renders the scene with the wavefront path tracer on a HIP device (no CPU fallback) and writes a\n"
                 "  the image: .exr (float, uncompressed), .png (8-bit sRGB) or .pfm by extension.  Film \"filename\" is used when\n"
                 "  --outfile is absent.\n"
                 "  -o, --outfile <file>      image file to write\n"
                 "  -s, --pixelsamples <n>    override the sampler's pixelsamples\n"
                 "      --quick               pixelsamples 1 and a quarter of the film resolution per axis\n"
                 "      --quick_full_resolution   --quick at the full resolution\n"
                 "  -c, --cat                 print the parsed directives, do not render\n"
                 "      --stats               print the ray counters\n"
                 "      --quiet               only error messages\n"
                 "  -j, --nthreads <n>        accepted for compatibility (the device schedules itself)\n"
                 "      --device <d>          HIP device index\n"
                 "      --display-server <host:port>   stream the image to a tev viewer while it renders (bands of tile rows)\n"
                 "      --gpus <n>            film tiles dealt round-robin to n GPUs (devices 0..n-1, one host thread and one\n"
                 "                            library context each), films summed with one RCCL reduce to device 0\n"
                 "      --devices <a,b,..>    the devices --gpus uses, one rank each (default 0..n-1).  A device named twice runs two\n"
                 "                            contexts on it and sums the films through the host: the rehearsal of the n-GPU flow on one GPU\n"
                 "      --xyzw <file>         also write the raw {X,Y,Z,weight} film (4 floats per pixel, native byte order)\n");
}

// --gpus N: the reference's tile loop (sampler.rs:266-301) sharded over GPUs inside one process.  Every device holds the whole
// scene and renders tiles r, r+N, r+2N, ...; the only exchange is the film sum.  No Python, no torch: RCCL's communicators come
// from ncclCommInitAll (librccl.so.1, resolved at run time like the library does) and the sum is pt_film_allreduce.  When the
// device list names one device twice (--devices 0,0: the rehearsal a one-GPU box allows -- RCCL refuses two ranks on one
// device) the films are summed through the host instead, rank by rank (pt_film_download_xyzw -> pt_film_add_xyzw on rank 0).
struct RankReport { int device = 0; size_t tiles = 0; double render_ms = 0, reduce_ms = 0; pt_counters cnt; };

// All ranks meet here before the exchange; if any of them has failed, nobody enters it (a collective entered by some ranks only
// never returns).
struct Rendezvous {
    std::mutex m;
    std::condition_variable cv;
    int waiting = 0, generation = 0, n = 0;
    bool any_failed = false;
    bool arrive(bool failed) {
        std::unique_lock<std::mutex> lk(m);
        any_failed = any_failed || failed;
        const int gen = generation;
        if (++waiting == n) { waiting = 0; generation++; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != gen; });
        return !any_failed;
    }
};

static int render_multi_gpu(const pt_scene_desc* dsc, const std::vector<int>& devs, std::vector<float>& rgb, pt_scene_info* info_out, pt_counters* total,
                            double* secs_out, std::vector<RankReport>* reports, std::vector<float>* xyzw_out) {
    const int n_gpus = (int)devs.size();
    bool shared_device = false;
    for (int i = 0; i < n_gpus; i++)
        for (int j = 0; j < i; j++) shared_device = shared_device || devs[(size_t)i] == devs[(size_t)j];
    std::vector<void*> comms((size_t)n_gpus, nullptr);
    int (*destroy)(void*) = nullptr;
    if (!shared_device) {
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        auto init_all = h ? reinterpret_cast<int (*)(void**, int, const int*)>(dlsym(h, "ncclCommInitAll")) : nullptr;
        destroy = h ? reinterpret_cast<int (*)(void*)>(dlsym(h, "ncclCommDestroy")) : nullptr;
        if (!init_all || !destroy) { std::fprintf(stderr, "pbrt_gpu: --gpus needs librccl.so.1 (ncclCommInitAll)\n"); return 1; }
        if (init_all(comms.data(), n_gpus, devs.data()) != 0) { std::fprintf(stderr, "pbrt_gpu: ncclCommInitAll failed for %d devices\n", n_gpus); return 1; }
    }
    std::vector<pt_context*> ctxs((size_t)n_gpus, nullptr);
    std::vector<int> rc((size_t)n_gpus, 0);
    std::vector<std::string> errs((size_t)n_gpus);
    reports->assign((size_t)n_gpus, RankReport());
    auto each = [&](auto body) {
        std::vector<std::thread> th;
        for (int r = 0; r < n_gpus; r++) th.emplace_back([&, r] { body(r); });
        for (auto& t : th) t.join();
        for (int r = 0; r < n_gpus; r++) if (rc[(size_t)r]) { std::fprintf(stderr, "pbrt_gpu: rank %d (device %d): %s\n", r, devs[(size_t)r], errs[(size_t)r].c_str()); return false; }
        return true;
    };
    auto check = [&](int r, pt_status st, const char* what) {
        if (st != PT_OK && rc[(size_t)r] == 0) { rc[(size_t)r] = 1; errs[(size_t)r] = std::string(what) + ": " + (ctxs[(size_t)r] ? pt_last_error(ctxs[(size_t)r]) : "no context"); }
        return st == PT_OK;
    };
    bool ok = each([&](int r) {
        if (!check(r, pt_context_create(devs[(size_t)r], &ctxs[(size_t)r]), "no usable HIP device")) return;
        check(r, pt_scene_upload(ctxs[(size_t)r], dsc), "scene upload");
    });
    double secs = 0;
    if (ok) {
        pt_scene_info_get(ctxs[0], info_out);
        const int32_t* sb = info_out->sample_bounds;
        std::vector<std::vector<pt_tile>> mine((size_t)n_gpus);
        int k = 0;
        for (int32_t y = sb[1]; y < sb[3]; y += 16)
            for (int32_t x = sb[0]; x < sb[2]; x += 16, k++) mine[(size_t)(k % n_gpus)].push_back({x, y, std::min(x + 16, sb[2]), std::min(y + 16, sb[3])});
        const size_t film_floats = (size_t)(info_out->cropped_bounds[2] - info_out->cropped_bounds[0]) * (size_t)(info_out->cropped_bounds[3] - info_out->cropped_bounds[1]) * 4;
        std::vector<std::vector<float>> staged((size_t)n_gpus);
        Rendezvous meet;
        meet.n = n_gpus;
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        auto t0 = std::chrono::steady_clock::now();
        ok = each([&](int r) {
            pt_context* c = ctxs[(size_t)r];
            RankReport& rep = (*reports)[(size_t)r];
            rep.device = devs[(size_t)r]; rep.tiles = mine[(size_t)r].size();
            auto ta = std::chrono::steady_clock::now();
            if (check(r, pt_film_clear(c), "film clear") && !mine[(size_t)r].empty())
                check(r, pt_render(c, mine[(size_t)r].data(), (uint32_t)mine[(size_t)r].size()), "render");
            auto tb = std::chrono::steady_clock::now();
            rep.render_ms = ms(ta, tb);
            if (shared_device && r > 0 && rc[(size_t)r] == 0) {        // host-staged exchange: hand the film over before the meeting point
                staged[(size_t)r].resize(film_floats);
                check(r, pt_film_download_xyzw(c, staged[(size_t)r].data()), "film download");
            }
            if (meet.arrive(rc[(size_t)r] != 0)) {                      // everybody rendered: the exchange
                if (!shared_device) check(r, pt_film_allreduce(c, comms[(size_t)r], 0), "film reduce");
                else if (r == 0)
                    for (int q = 1; q < n_gpus; q++)
                        if (!check(0, pt_film_add_xyzw(c, staged[(size_t)q].data()), "film add")) break;
            }
            rep.reduce_ms = ms(tb, std::chrono::steady_clock::now());   // the wait for the slowest rank + the exchange itself
            pt_get_counters(c, &rep.cnt);
        });
        if (ok) {
            int w = info_out->cropped_bounds[2] - info_out->cropped_bounds[0], hgt = info_out->cropped_bounds[3] - info_out->cropped_bounds[1];
            rgb.assign((size_t)w * hgt * 3, 0.0f);
            ok = pt_film_resolve_rgb(ctxs[0], rgb.data()) == PT_OK;
        }
        secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (ok && xyzw_out) { xyzw_out->resize(film_floats); ok = pt_film_download_xyzw(ctxs[0], xyzw_out->data()) == PT_OK; }
    }
    std::memset(total, 0, sizeof(*total));
    for (int r = 0; r < n_gpus; r++) {
        const pt_counters& c = (*reports)[(size_t)r].cnt;
        total->regular_rays += c.regular_rays; total->shadow_rays += c.shadow_rays;
        total->camera_rays += c.camera_rays; total->path_vertices += c.path_vertices;
        if (ctxs[(size_t)r]) pt_context_destroy(ctxs[(size_t)r]);
        if (comms[(size_t)r] && destroy) destroy(comms[(size_t)r]);
    }
    *secs_out = secs;
    return ok ? 0 : 1;
}

int main(int argc, char** argv) {
    std::string input, outfile, display_server;
    int spp = 0, device = 0, gpus = 0;
    std::vector<int> devices;
    std::string xyzw_file;
    bool quiet = false, cat = false, stats = false;
    pth_options opts;
    std::memset(&opts, 0, sizeof(opts));
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto need = [&](const char* name) -> const char* { if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", name); std::exit(2); } return argv[++i]; };
        if (a == "-i" || a == "--infile") input = need("-i");
        else if (a == "-o" || a == "--outfile") outfile = need("--outfile");
        else if (a == "--pixelsamples" || a == "-s") spp = std::max(1, std::atoi(need("--pixelsamples")));
        else if (a == "--quick") opts.quick = 1;
        else if (a == "--quick_full_resolution" || a == "--quick-full-resolution") opts.quick_full_resolution = 1;
        else if (a == "--cat" || a == "-c") cat = true;
        else if (a == "--stats") stats = true;
        else if (a == "--nthreads" || a == "-j") (void)need("--nthreads");
        else if (a == "--device") device = std::atoi(need("--device"));
        else if (a == "--gpus") gpus = std::max(1, std::atoi(need("--gpus")));
        else if (a == "--devices") { for (const char* q = need("--devices"); *q;) { devices.push_back(std::atoi(q)); while (*q && *q != ',') q++; if (*q) q++; } }
        else if (a == "--xyzw") xyzw_file = need("--xyzw");
        else if (a == "--display-server") display_server = need("--display-server");
        else if (a == "--quiet") quiet = true;
        else if (a == "-h" || a == "--help") { usage(); return 0; }
        else if (!a.empty() && a[0] != '-') input = a;
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); usage(); return 2; }
    }
    if (input.empty()) { usage(); return 2; }
    if (cat) {                                    // --cat (bin/pbrt.rs:211-224): the directives as parsed
        FILE* f = std::fopen(input.c_str(), "rb");
        if (!f) { std::fprintf(stderr, "pbrt_gpu: cannot open %s\n", input.c_str()); return 1; }
        std::string text;
        char buf[65536];
        size_t n;
        while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) text.append(buf, n);
        std::fclose(f);
        std::string dir = input.find_last_of('/') == std::string::npos ? std::string(".") : input.substr(0, input.find_last_of('/'));
        std::vector<char> out(text.size() * 8 + (1u << 20));
        pt_status cs = pth_parse_to_log(text.c_str(), dir.c_str(), out.data(), out.size());
        std::fputs(out.data(), cs == PT_OK ? stdout : stderr);
        return cs == PT_OK ? 0 : 1;
    }
    opts.pixelsamples = spp;
    char err[1024] = {0};
    pth_scene* scene = nullptr;
    pt_status st = pth_parse_file_opts(input.c_str(), &opts, &scene, err, sizeof(err));
    if (st != PT_OK) { std::fprintf(stderr, "pbrt_gpu: %s\n", err); return 1; }
    if (!quiet && pth_scene_warnings(scene)[0]) std::fprintf(stderr, "%s", pth_scene_warnings(scene));
    if (outfile.empty()) {
        outfile = pth_scene_output_filename(scene);
        size_t dot = outfile.find_last_of('.');
        std::string ext = dot == std::string::npos ? std::string() : outfile.substr(dot);
        if (ext != ".exr" && ext != ".png" && ext != ".pfm") outfile = (dot == std::string::npos ? outfile : outfile.substr(0, dot)) + ".pfm";
    }
    if (gpus > 0 || !devices.empty()) {
        if (devices.empty()) for (int i = 0; i < gpus; i++) devices.push_back(i);
        if (gpus > 0 && (int)devices.size() != gpus) { std::fprintf(stderr, "pbrt_gpu: --gpus %d but --devices lists %zu\n", gpus, devices.size()); pth_scene_free(scene); return 2; }
        gpus = (int)devices.size();
        std::vector<float> rgb;
        pt_scene_info info;
        pt_counters c;
        double secs = 0;
        std::vector<RankReport> reports;
        std::vector<float> xyzw;
        if (render_multi_gpu(pth_scene_get_desc(scene), devices, rgb, &info, &c, &secs, &reports, xyzw_file.empty() ? nullptr : &xyzw) != 0) { pth_scene_free(scene); return 1; }
        if (!xyzw_file.empty() && !write_floats(xyzw_file, xyzw)) return 1;
        int w = info.cropped_bounds[2] - info.cropped_bounds[0], h = info.cropped_bounds[3] - info.cropped_bounds[1];
        const pt_scene_desc* dsc = pth_scene_get_desc(scene);
        if (pth_write_image(outfile.c_str(), rgb.data(), w, h, info.cropped_bounds[0], info.cropped_bounds[1], dsc->xres, dsc->yres) != PT_OK) { std::fprintf(stderr, "pbrt_gpu: cannot write %s\n", outfile.c_str()); return 1; }
        if (!quiet)
            std::fprintf(stderr, "pbrt_gpu: %s  %dx%d, %d spp on %d rank(s)  rendered + reduced in %.3f s  %.1f Mrays/s  -> %s\n", input.c_str(), w, h, info.spp, gpus, secs,
                         (double)(c.regular_rays + c.shadow_rays) / secs / 1e6, outfile.c_str());
        if (stats) {
            std::fprintf(stderr, "  Regular ray intersection tests %llu\n  Shadow ray intersection tests  %llu\n  Camera rays %llu, path vertices %llu\n",
                         (unsigned long long)c.regular_rays, (unsigned long long)c.shadow_rays, (unsigned long long)c.camera_rays, (unsigned long long)c.path_vertices);
            for (size_t r = 0; r < reports.size(); r++)      // who rendered how long, who waited in the exchange
                std::fprintf(stderr, "  rank %zu device %d: tiles %zu  render_ms %.3f  reduce_ms %.3f  rays %llu\n", r, reports[r].device, reports[r].tiles, reports[r].render_ms,
                             reports[r].reduce_ms, (unsigned long long)(reports[r].cnt.regular_rays + reports[r].cnt.shadow_rays));
        }
        pth_scene_free(scene);
        return 0;
    }
    pt_context* ctx = nullptr;
    st = pt_context_create(device, &ctx);
    if (st != PT_OK) { std::fprintf(stderr, "pbrt_gpu: no usable HIP device %d (there is no CPU fallback)\n", device); return 1; }
    auto fail = [&](const char* what) { std::fprintf(stderr, "pbrt_gpu: %s: %s\n", what, pt_last_error(ctx)); pt_context_destroy(ctx); pth_scene_free(scene); return 1; };
    if (pt_scene_upload(ctx, pth_scene_get_desc(scene)) != PT_OK) return fail("scene upload");
    pt_scene_info info;
    pt_scene_info_get(ctx, &info);
    int w = info.cropped_bounds[2] - info.cropped_bounds[0], h = info.cropped_bounds[3] - info.cropped_bounds[1];
    auto t0 = std::chrono::steady_clock::now();
    std::vector<float> rgb((size_t)w * h * 3);
    if (!display_server.empty()) {
        // --display-server (bin/pbrt.rs:297-309): Film::render_start creates the image at the full resolution, every finished band of
        // tile rows is pushed as Film::update_display pushes a merged tile (film.rs:278-360)
        pth_display* disp = nullptr;
        char derr[256] = {0};
        if (pth_display_connect(display_server.c_str(), &disp, derr, sizeof(derr)) != PT_OK) { std::fprintf(stderr, "pbrt_gpu: %s\n", derr); pt_context_destroy(ctx); pth_scene_free(scene); return 1; }
        const pt_scene_desc* dd = pth_scene_get_desc(scene);
        pth_display_start(disp, pth_scene_output_filename(scene), (uint32_t)dd->xres, (uint32_t)dd->yres);
        if (pt_film_clear(ctx) != PT_OK) return fail("film clear");
        const int32_t* sb = info.sample_bounds;
        const int32_t band = 128;
        for (int32_t y0 = sb[1]; y0 < sb[3]; y0 += band) {
            std::vector<pt_tile> tiles;
            for (int32_t y = y0; y < std::min(y0 + band, sb[3]); y += 16)
                for (int32_t x = sb[0]; x < sb[2]; x += 16) tiles.push_back({x, y, std::min(x + 16, sb[2]), std::min(y + 16, std::min(y0 + band, sb[3]))});
            if (pt_render(ctx, tiles.data(), (uint32_t)tiles.size()) != PT_OK) return fail("render");
            if (pt_film_resolve_rgb(ctx, rgb.data()) != PT_OK) return fail("film resolve");
            const int32_t cy0 = std::max(y0, info.cropped_bounds[1]), cy1 = std::min(std::min(y0 + band, sb[3]), info.cropped_bounds[3]);
            if (cy1 > cy0)
                pth_display_update(disp, (uint32_t)info.cropped_bounds[0], (uint32_t)cy0, (uint32_t)w, (uint32_t)(cy1 - cy0), &rgb[(size_t)(cy0 - info.cropped_bounds[1]) * w * 3]);
        }
        pth_display_close(disp);
    } else if (pt_film_clear(ctx) != PT_OK || pt_render(ctx, nullptr, 0) != PT_OK) return fail("render");
    if (pt_film_resolve_rgb(ctx, rgb.data()) != PT_OK) return fail("film resolve");
    if (!xyzw_file.empty()) {
        std::vector<float> xyzw((size_t)w * h * 4);
        if (pt_film_download_xyzw(ctx, xyzw.data()) != PT_OK) return fail("film download");
        if (!write_floats(xyzw_file, xyzw)) return 1;
    }
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    pt_counters c;
    pt_get_counters(ctx, &c);
    const pt_scene_desc* dsc = pth_scene_get_desc(scene);
    if (pth_write_image(outfile.c_str(), rgb.data(), w, h, info.cropped_bounds[0], info.cropped_bounds[1], dsc->xres, dsc->yres) != PT_OK) { std::fprintf(stderr, "pbrt_gpu: cannot write %s\n", outfile.c_str()); return 1; }
    if (!quiet)
        std::fprintf(stderr, "pbrt_gpu: %s  %dx%d, %d spp, %u lights, BVH %u nodes (%.0f ms)  rendered in %.3f s  %.1f Mrays/s  -> %s\n", input.c_str(), w, h,
                     info.spp, info.n_lights, info.n_nodes, info.bvh_build_ms, secs, (double)(c.regular_rays + c.shadow_rays) / secs / 1e6, outfile.c_str());
    if (stats)                                    // the counters behind "Intersections/..." (core/scene/scene.rs:11-12)
        std::fprintf(stderr, "  Regular ray intersection tests %llu\n  Shadow ray intersection tests  %llu\n  Camera rays %llu, path vertices %llu\n",
                     (unsigned long long)c.regular_rays, (unsigned long long)c.shadow_rays, (unsigned long long)c.camera_rays, (unsigned long long)c.path_vertices);
    pt_context_destroy(ctx);
    pth_scene_free(scene);
    return 0;
}
