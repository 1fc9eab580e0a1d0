// srt_viewer — the interactive front end around the accelerated hot path.
//
// What the reference's main loop does around its workers (Raytracer/Raytracer.cpp:362-612): read input,
// move / turn the camera, pick, change settings, raise doSetFrame, run the accumulate state machine,
// release the workers for one frame, blit renderSurface.  Here the workers are `PathTraceRenderer`
// (host/renderer.hpp -> libsrt_pathtrace.so) and the blit is one `ReadFramebuffer(pixels, pitch)` into
// the window surface: the framebuffer is ARGB8888, rows in blit order, exactly what SetScreenPixel
// writes into renderSurface->pixels (:64,75), so the surface is used unchanged.
//
// Two back ends share ViewerCore:
//   * headless (always built, tested): input comes from a script, frames go to PPM files;
//   * SDL2 window (compiled only when the Makefile finds SDL2 through pkg-config, -DSRT_WITH_SDL2 —
//     this image has no SDL2, so that part has only been syntax-checked against hand-written
//     declarations, never linked or run; it makes the same ViewerCore calls the headless back end
//     exercises).
// The reference's ImGui inspector (vendored Dear ImGui) is not reproduced; its settings are on keys.
//
// Input semantics mirrored from the reference (citations: Raytracer.cpp):
//   right mouse held + motion  camera.RotateAboutAxis(dx * mouseSpeed * 0.03, WORLDUP), then
//                              RotateAboutAxis(dy * mouseSpeed * 0.03, camera.right)      :390-396
//   W A S D E Q                position +/- forward / right / up * speed,
//                              speed = moveSpeed * delta, LSHIFT: 2 * delta                :499-521
//   left click                 deselect, or pick at (x, SCREEN_HEIGHT - y)                 :525-541
//   DELETE                     remove the selected object from the scene                   :490-497
//   P                          pause flag                                                  :386-388
// and, standing in for the inspector's "Settings" (:461-483):
//   M  switch render mode (SIMPLEDRAW)     F / G  FOV -/+ 1 (15..103)     B / N  light bounces -/+ 1
//   1..4  render scale 0.25 / 0.5 / 0.75 / 1.0 (clamped to 0.5 in preview mode, :481-483)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "renderer.hpp"

#ifdef SRT_WITH_SDL2
#include <SDL.h>
#endif

using namespace srt_host;

namespace {

struct InputState {  // one frame's worth, what SDLInputManager hands the loop
    bool w = false, a = false, s = false, d = false, e = false, q = false, lshift = false;  // held
    bool right_held = false;
    int mouse_dx = 0, mouse_dy = 0;  // relative motion of this frame
    bool left_down = false;          // edge
    int mouse_x = 0, mouse_y = 0;    // window coordinates of the click, y down
    std::string pressed;             // key-down edges of this frame: any of "PMFGBN1234X" (X = DELETE)
};

class ViewerCore {
   public:
    ViewerCore(int device, int width, int height, const std::string& scene_file) : scene_(scene_file), r_(device, width, height) {
        scene_.Load();
        if (!scene_.lastError().empty()) std::fprintf(stderr, "scene: %s\n", scene_.lastError().c_str());  // Scene.hpp:76
        r_.SetScene(scene_);
    }
    PathTraceRenderer& renderer() { return r_; }
    bool paused() const { return pause_; }
    int selected() const { return r_.selectedObject; }

    // one pass of the loop body between "thread safe after this point" (:385) and the release of the
    // workers (:592-595); delta in seconds (:558-560)
    void Frame(const InputState& in, float delta) {
        const float mouseSpeed = .08f, moveSpeed = 1;  // :353-354
        for (char k : in.pressed) Key(k);
        if (in.right_held) {  // :390-396
            r_.Invalidate();
            r_.camera.RotateAboutAxis((float)(in.mouse_dx * mouseSpeed * 0.03), Vec3(0, 1, 0));
            r_.camera.RotateAboutAxis((float)(in.mouse_dy * mouseSpeed * 0.03), r_.camera.right);
        }
        float speed = moveSpeed * delta;  // :499
        if (in.lshift) speed = 2 * delta;
        const Vec3 before = r_.camera.position;
        Vec3& p = r_.camera.position;
        if (in.w) p = p + r_.camera.forward * speed;
        if (in.d) p = p + r_.camera.right * speed;
        if (in.a) p = p - r_.camera.right * speed;
        if (in.s) p = p - r_.camera.forward * speed;
        if (in.e) p = p + r_.camera.up * speed;
        if (in.q) p = p - r_.camera.up * speed;
        if (p.x != before.x || p.y != before.y || p.z != before.z) r_.Invalidate();  // :519-521
        if (in.left_down) {  // :525-541
            if (r_.selectedObject >= 0) r_.selectedObject = -1;
            else r_.selectedObject = r_.Pick(in.mouse_x, in.mouse_y);
        }
        r_.RenderFrame();  // :572-595
    }

   private:
    void Key(char k) {
        switch (k) {
            case 'P': pause_ = !pause_; break;                                     // :386-388
            case 'M': r_.SIMPLEDRAW = !r_.SIMPLEDRAW; r_.Invalidate(); break;      // :462-465
            case 'F': case 'G': {                                                  // :468-473
                int f = r_.FOV + (k == 'G' ? 1 : -1);
                f = f < 15 ? 15 : (f > 103 ? 103 : f);
                if (f != r_.FOV) r_.FOV = f, r_.Invalidate();
                break;
            }
            case 'B': case 'N': {                                                  // :474-480
                int b = r_.MAXBOUNCES + (k == 'N' ? 1 : -1);
                b = b < 0 ? 0 : b;
                if (b != r_.MAXBOUNCES) r_.MAXBOUNCES = b, r_.Invalidate();
                break;
            }
            case '1': case '2': case '3': case '4': r_.SCREEN_SCALE = 0.25f * (float)(k - '0'); break;  // :481
            case 'X':                                                              // :490-497
                if (r_.selectedObject >= 0) {
                    scene_.RemoveObject((size_t)r_.selectedObject);
                    r_.selectedObject = -1;
                    r_.SetScene(scene_);  // ObjectsToRender follows the scene; raises doSetFrame
                }
                break;
            default: break;
        }
        if (r_.SIMPLEDRAW) r_.SCREEN_SCALE = r_.SCREEN_SCALE < 0.25f ? 0.25f : (r_.SCREEN_SCALE > 0.5f ? 0.5f : r_.SCREEN_SCALE);  // :481-483
    }
    Scene scene_;
    PathTraceRenderer r_;
    bool pause_ = false;
};

void write_ppm(PathTraceRenderer& r, const std::string& path) {
    const int W = r.width(), H = r.height();
    std::vector<uint32_t> fb((size_t)W * H);
    r.Wait();
    r.ReadFramebuffer(fb.data(), (size_t)W * 4);
    std::vector<unsigned char> rgb((size_t)W * H * 3);
    for (size_t i = 0; i < fb.size(); ++i) rgb[3 * i] = (fb[i] >> 16) & 255, rgb[3 * i + 1] = (fb[i] >> 8) & 255, rgb[3 * i + 2] = fb[i] & 255;
    std::ofstream f(path, std::ios::binary);
    f << "P6\n" << W << " " << H << "\n255\n";
    f.write((const char*)rgb.data(), (std::streamsize)rgb.size());
}

// Script of the headless back end, one command per line ('#' starts a comment):
//   delta SECONDS | hold KEYS | release KEYS   (KEYS out of W A S D E Q and L for LSHIFT)
//   press KEYS (P M F G B N 1 2 3 4 X, applied to the next frame only)
//   rmb down|up | move DX DY (relative mouse motion of the next frame) | click X Y
//   frames N | save FILE.ppm | print
int run_script(ViewerCore& core, std::istream& script) {
    InputState in;
    float delta = 1.0f / 60.0f;
    std::string line;
    int frames_run = 0;
    auto one_frame = [&] {
        core.Frame(in, delta);
        in.pressed.clear();
        in.mouse_dx = in.mouse_dy = 0;
        in.left_down = false;
        ++frames_run;
    };
    auto set_keys = [&](const std::string& keys, bool v) {
        for (char k : keys) switch (k) {
            case 'W': in.w = v; break; case 'A': in.a = v; break; case 'S': in.s = v; break; case 'D': in.d = v; break;
            case 'E': in.e = v; break; case 'Q': in.q = v; break; case 'L': in.lshift = v; break; default: break;
        }
    };
    while (std::getline(script, line)) {
        std::istringstream ss(line);
        std::string cmd;
        if (!(ss >> cmd) || cmd[0] == '#') continue;
        if (cmd == "delta") ss >> delta;
        else if (cmd == "hold" || cmd == "release") { std::string k; ss >> k; set_keys(k, cmd == "hold"); }
        else if (cmd == "press") { std::string k; ss >> k; in.pressed += k; }
        else if (cmd == "rmb") { std::string v; ss >> v; in.right_held = v == "down"; }
        else if (cmd == "move") ss >> in.mouse_dx >> in.mouse_dy;
        else if (cmd == "click") { ss >> in.mouse_x >> in.mouse_y; in.left_down = true; one_frame(); }
        else if (cmd == "frames") { int n = 0; ss >> n; for (int i = 0; i < n; ++i) one_frame(); }
        else if (cmd == "save") { std::string p; ss >> p; write_ppm(core.renderer(), p); }
        else if (cmd == "print") {
            PathTraceRenderer& r = core.renderer();
            std::printf("frames %d acc %d simpledraw %d fov %d bounces %d scale %.2f selected %d pos %.9g %.9g %.9g fwd %.9g %.9g %.9g paused %d\n", frames_run,
                        r.ACCUMULATIONFRAMES, (int)r.SIMPLEDRAW, r.FOV, r.MAXBOUNCES, r.SCREEN_SCALE, core.selected(), r.camera.position.x, r.camera.position.y,
                        r.camera.position.z, r.camera.forward.x, r.camera.forward.y, r.camera.forward.z, (int)core.paused());
        } else {
            std::fprintf(stderr, "script: unknown command '%s'\n", cmd.c_str());
            return 2;
        }
    }
    return 0;
}

#ifdef SRT_WITH_SDL2
// The window back end: SDL2 only provides the window surface and the events; the pixels come straight
// from the library (no conversion: SDL_PIXELFORMAT_ARGB8888 is the framebuffer's layout).
int run_window(ViewerCore& core) {
    if (SDL_Init(SDL_INIT_VIDEO) != 0) {
        std::fprintf(stderr, "SDL_Init: %s\n", SDL_GetError());
        return 1;
    }
    PathTraceRenderer& r = core.renderer();
    SDL_Window* window = SDL_CreateWindow("srt_viewer", SDL_WINDOWPOS_CENTERED, SDL_WINDOWPOS_CENTERED, r.width(), r.height(), SDL_WINDOW_SHOWN);
    if (!window) {
        std::fprintf(stderr, "SDL_CreateWindow: %s\n", SDL_GetError());
        SDL_Quit();
        return 1;
    }
    SDL_Surface* screen = SDL_GetWindowSurface(window);
    SDL_Surface* frame = SDL_CreateRGBSurfaceWithFormat(0, r.width(), r.height(), 32, SDL_PIXELFORMAT_ARGB8888);
    InputState in;
    auto t1 = std::chrono::steady_clock::now();
    float delta = 1.0f / 60.0f;
    bool quit = false;
    while (!quit) {
        in.pressed.clear();
        in.left_down = false;
        SDL_Event ev;
        while (SDL_PollEvent(&ev)) {
            if (ev.type == SDL_QUIT) quit = true;
            else if (ev.type == SDL_MOUSEBUTTONDOWN && ev.button.button == SDL_BUTTON_LEFT) in.left_down = true, in.mouse_x = ev.button.x, in.mouse_y = ev.button.y;
            else if (ev.type == SDL_KEYDOWN && !ev.key.repeat) {
                switch (ev.key.keysym.scancode) {
                    case SDL_SCANCODE_P: in.pressed += 'P'; break; case SDL_SCANCODE_M: in.pressed += 'M'; break;
                    case SDL_SCANCODE_F: in.pressed += 'F'; break; case SDL_SCANCODE_G: in.pressed += 'G'; break;
                    case SDL_SCANCODE_B: in.pressed += 'B'; break; case SDL_SCANCODE_N: in.pressed += 'N'; break;
                    case SDL_SCANCODE_1: in.pressed += '1'; break; case SDL_SCANCODE_2: in.pressed += '2'; break;
                    case SDL_SCANCODE_3: in.pressed += '3'; break; case SDL_SCANCODE_4: in.pressed += '4'; break;
                    case SDL_SCANCODE_DELETE: in.pressed += 'X'; break; case SDL_SCANCODE_ESCAPE: quit = true; break;
                    default: break;
                }
            }
        }
        const Uint8* keys = SDL_GetKeyboardState(nullptr);
        in.w = keys[SDL_SCANCODE_W], in.a = keys[SDL_SCANCODE_A], in.s = keys[SDL_SCANCODE_S], in.d = keys[SDL_SCANCODE_D];
        in.e = keys[SDL_SCANCODE_E], in.q = keys[SDL_SCANCODE_Q], in.lshift = keys[SDL_SCANCODE_LSHIFT];
        const Uint32 buttons = SDL_GetRelativeMouseState(&in.mouse_dx, &in.mouse_dy);
        in.right_held = (buttons & SDL_BUTTON(SDL_BUTTON_RIGHT)) != 0;
        core.Frame(in, delta);
        r.Wait();
        SDL_LockSurface(frame);
        r.ReadFramebuffer(frame->pixels, (size_t)frame->pitch);  // the blit (:64,75): ARGB8888, top row first
        SDL_UnlockSurface(frame);
        SDL_BlitSurface(frame, nullptr, screen, nullptr);
        SDL_UpdateWindowSurface(window);
        const auto t2 = std::chrono::steady_clock::now();
        delta = (float)std::chrono::duration_cast<std::chrono::milliseconds>(t2 - t1).count() / 1000.0f;  // :558-560
        t1 = t2;
    }
    SDL_FreeSurface(frame);
    SDL_DestroyWindow(window);
    SDL_Quit();
    return 0;
}
#endif

}  // namespace

int main(int argc, char** argv) {
    std::string scene_path, script_path;
    int W = 1280, H = 720, device = 0;  // Raytracer.cpp:26-27
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
        else if (!std::strcmp(argv[i], "--device")) device = std::atoi(need("--device"));
        else if (!std::strcmp(argv[i], "--script")) script_path = need("--script");
        else scene_path.clear(), i = argc;
    }
    if (scene_path.empty() || W <= 0 || H <= 0) {
        std::fprintf(stderr, "usage: srt_viewer --scene FILE [--width 1280] [--height 720] [--device 0] [--script FILE|-]\n"
                             "       without --script an SDL2 window is opened (only in builds with SDL2)\n");
        return 2;
    }
    try {
        ViewerCore core(device, W, H, scene_path);
        if (!script_path.empty()) {
            if (script_path == "-") return run_script(core, std::cin);
            std::ifstream f(script_path);
            if (!f.good()) {
                std::fprintf(stderr, "cannot open %s\n", script_path.c_str());
                return 2;
            }
            return run_script(core, f);
        }
#ifdef SRT_WITH_SDL2
        return run_window(core);
#else
        std::fprintf(stderr, "srt_viewer: built without SDL2 (not in this image); use --script for the headless back end\n");
        return 3;
#endif
    } catch (const std::exception& e) {
        std::fprintf(stderr, "srt_viewer: %s\n", e.what());
        return 1;
    }
}
