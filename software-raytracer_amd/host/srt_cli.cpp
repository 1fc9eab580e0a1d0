// srt_render — command-line front end of the host layer: load a scene in the reference's
// JSON format, path-trace it on one MI355X, write the framebuffer as a binary PPM (P6)
// top-down (the framebuffer's memory rows are already in blit order, Raytracer.cpp:64).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "renderer.hpp"

using namespace srt_host;

static void usage() {
    std::fprintf(stderr,
                 "usage: srt_render --scene FILE [--width 1280] [--height 720] [--spp 32] [--bounces 2]\n"
                 "                  [--fov 55] [--seed 0] [--device 0 | --devices 0,1,2,...] [--out frame.ppm] [--resave FILE]\n"
                 "  --devices: one frame over several GPUs of this node in one process (equal row bands, one gather;\n"
                 "             a device may be listed more than once); bands of equal estimated cost (default; --balance is accepted\n"
                 "             and means the same), --equal-bands: bands of equal height\n");
}

int main(int argc, char** argv) {
    std::string scene_path, out = "frame.ppm", resave;
    int W = 1280, H = 720, spp = 32, bounces = 2, fov = 55, device = 0;  // Raytracer.cpp:26-27,31-32
    unsigned seed = 0;
    std::vector<int> devices;
    bool equal_bands = false;
    for (int i = 1; i < argc; ++i) {
        auto need = [&](const char* n) -> const char* {
            if (i + 1 >= argc) {
                std::fprintf(stderr, "%s needs a value\n", n);
                std::exit(2);
            }
            return argv[++i];
        };
        if (!std::strcmp(argv[i], "--scene")) scene_path = need("--scene");
        else if (!std::strcmp(argv[i], "--width")) W = std::atoi(need("--width"));
        else if (!std::strcmp(argv[i], "--height")) H = std::atoi(need("--height"));
        else if (!std::strcmp(argv[i], "--spp")) spp = std::atoi(need("--spp"));
        else if (!std::strcmp(argv[i], "--bounces")) bounces = std::atoi(need("--bounces"));
        else if (!std::strcmp(argv[i], "--fov")) fov = std::atoi(need("--fov"));
        else if (!std::strcmp(argv[i], "--seed")) seed = (unsigned)std::strtoul(need("--seed"), nullptr, 10);
        else if (!std::strcmp(argv[i], "--device")) device = std::atoi(need("--device"));
        else if (!std::strcmp(argv[i], "--balance")) equal_bands = false;
        else if (!std::strcmp(argv[i], "--equal-bands")) equal_bands = true;
        else if (!std::strcmp(argv[i], "--devices")) {
            for (const char* p = need("--devices"); *p;) {
                devices.push_back(std::atoi(p));
                while (*p && *p != ',') ++p;
                if (*p == ',') ++p;
            }
        }
        else if (!std::strcmp(argv[i], "--out")) out = need("--out");
        else if (!std::strcmp(argv[i], "--resave")) resave = need("--resave");
        else {
            usage();
            return 2;
        }
    }
    if (scene_path.empty() || W <= 0 || H <= 0 || spp <= 0) {
        usage();
        return 2;
    }
    Scene scene(scene_path);
    scene.Load();
    if (!scene.lastError().empty()) std::fprintf(stderr, "scene: %s\n", scene.lastError().c_str());  // Scene.hpp:76
    std::fprintf(stderr, "scene %s: %zu objects\n", scene_path.c_str(), scene.GetObjects().size());
    if (!resave.empty()) scene.SaveAs(resave);
    auto write_ppm = [&](const std::vector<uint32_t>& fb) {
        FILE* f = std::fopen(out.c_str(), "wb");
        if (!f) {
            std::perror(out.c_str());
            return 1;
        }
        std::fprintf(f, "P6\n%d %d\n255\n", W, H);
        for (uint32_t px : fb) {
            unsigned char rgb[3] = {(unsigned char)(px >> 16), (unsigned char)(px >> 8), (unsigned char)px};
            std::fwrite(rgb, 1, 3, f);
        }
        std::fclose(f);
        return 0;
    };
    if (!devices.empty()) {
        try {
            MultiGpuRenderer m(devices, W, H);
            m.SetScene(scene);
            m.Configure(Transform(), fov, bounces, seed);
            if (equal_bands) m.UseEqualBands(true);
            auto t0 = std::chrono::steady_clock::now();
            m.RenderSamples((uint32_t)spp, true);
            std::vector<uint32_t> fb((size_t)W * H);
            m.ReadFramebuffer(fb.data(), (size_t)W * 4);
            double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::vector<srt_stats> st = m.Stats();
            double slowest = 0;
            for (size_t i = 0; i < st.size(); ++i) {
                int b, e;
                m.Band(i, &b, &e);
                std::fprintf(stderr, "  part %zu (device %d, memory rows %d-%d): kernel %.3f ms, %.2f rays/sample\n", i, devices[i], b, e, st[i].kernel_ms,
                             (double)st[i].rays / (double)st[i].path_samples);
                slowest = st[i].kernel_ms > slowest ? st[i].kernel_ms : slowest;
            }
            std::fprintf(stderr, "%dx%d spp=%d bounces=%d over %zu parts: slowest kernel %.3f ms, render + gather + read-back wall %.3f ms\n", W, H, spp, bounces,
                         st.size(), slowest, wall * 1e3);
            return write_ppm(fb);
        } catch (const std::exception& e) {
            std::fprintf(stderr, "error: %s\n", e.what());
            return 1;
        }
    }
    try {
        PathTraceRenderer r(device, W, H);
        r.FOV = fov;
        r.MAXBOUNCES = bounces;
        r.seed = seed;
        r.SetScene(scene);
        auto t0 = std::chrono::steady_clock::now();
        r.RenderSamples((uint32_t)spp, true);
        r.Wait();
        double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        srt_stats st = r.Stats();
        std::fprintf(stderr, "%dx%d spp=%d bounces=%d: kernel %.3f ms (wall %.3f ms), %.3e path-samples/s, %.2f rays/sample\n", W, H, spp,
                     bounces, st.kernel_ms, wall * 1e3, (double)st.path_samples / (st.kernel_ms * 1e-3),
                     (double)st.rays / (double)st.path_samples);
        std::vector<uint32_t> fb((size_t)W * H);
        r.ReadFramebuffer(fb.data(), (size_t)W * 4);
        if (write_ppm(fb)) return 1;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
