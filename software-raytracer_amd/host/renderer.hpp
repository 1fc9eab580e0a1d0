// renderer.hpp — host C++ that owns camera, settings and the progressive-sample state and
// drives the C-ABI (include/srt_pathtrace.h).  It stands where the reference's main loop
// stands relative to its workers (Raytracer/Raytracer.cpp:329-342, 373-384, 572-595).
#pragma once

#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "scene.hpp"
#include "srt_pathtrace.h"

namespace srt_host {

struct Vec3 {  // the members of Common.hpp's float3 that Transform uses
    float x = 0, y = 0, z = 0;
    Vec3() = default;
    Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    Vec3 operator+(const Vec3& o) const { return {x + o.x, y + o.y, z + o.z}; }
    Vec3 operator-(const Vec3& o) const { return {x - o.x, y - o.y, z - o.z}; }
    Vec3 operator*(float s) const { return {x * s, y * s, z * s}; }  // float3 * float3(s)
    static float Dot(const Vec3& l, const Vec3& r) { return (l.x * r.x + l.y * r.y + l.z * r.z); }
    static Vec3 Cross(const Vec3& l, const Vec3& r) {  // Common.hpp:94-96
        return {l.y * r.z - r.y * l.z, r.x * l.z - l.x * r.z, l.x * r.y - r.x * l.y};
    }
};

// Transform (Common.hpp:281-292)
struct Transform {
    Vec3 right{1, 0, 0};
    Vec3 up{0, 1, 0};
    Vec3 forward{0, 0, 1};
    Vec3 position{0, 0, 0};
    Vec3 scale{1, 1, 1};
    // Rodrigues rotation of the three basis vectors (Common.hpp:287-291)
    void RotateAboutAxis(float angle, Vec3 axis) {
        forward = forward * cosf(angle) + Vec3::Cross(axis, forward) * sinf(angle) + axis * Vec3::Dot(axis, forward) * (1 - cosf(angle));
        up = up * cosf(angle) + Vec3::Cross(axis, up) * sinf(angle) + axis * Vec3::Dot(axis, up) * (1 - cosf(angle));
        right = right * cosf(angle) + Vec3::Cross(axis, right) * sinf(angle) + axis * Vec3::Dot(axis, right) * (1 - cosf(angle));
    }
};

class RendererError : public std::runtime_error {
   public:
    RendererError(int code, const std::string& m) : std::runtime_error(m), code_(code) {}
    int code() const { return code_; }

   private:
    int code_;
};

// One GPU, one image (or one row band of it).
class PathTraceRenderer {
   public:
    // the reference's mutable globals (Raytracer.cpp:30-35,47-48,53) as members, same defaults
    float SCREEN_SCALE = .5f;
    int FOV = 55;
    int MAXBOUNCES = 2;
    int TARGETFRAMES = 4096;
    int ACCUMULATIONFRAMES = 1;
    bool SIMPLEDRAW = true;
    float progressiveResolutionScaler = 1;  // :47, set to 1 at :271
    bool setFrame = false;                  // :48
    int selectedObject = -1;                // :53 as an index into ObjectsToRender, -1 = NULL
    static constexpr int THREADS = 16;      // :28 — only used for the block anchoring of :235,330
    uint32_t seed = 0;  // the reference only ever names srand(0) (:263)
    Transform camera;   // :295-297

    PathTraceRenderer(int device, int width, int height);
    ~PathTraceRenderer();
    PathTraceRenderer(const PathTraceRenderer&) = delete;
    PathTraceRenderer& operator=(const PathTraceRenderer&) = delete;

    int width() const { return width_; }
    int height() const { return height_; }
    srt_context* handle() { return ctx_; }

    // ObjectsToRender = scene.GetObjects(); doSetFrame = true   (:293, :421-423)
    void SetScene(const Scene& scene);
    void SetEnvironment(const srt_environment& env);
    // restrict rendering to memory rows [begin,end) (multi-GPU row stripes)
    void SetRowBand(int begin, int end);
    void RowBand(int* begin, int* end) const { *begin = row_begin_, *end = row_end_; }
    // doSetFrame = true: any camera / object / setting edit (:391,453,461,469,476,497,522)
    void Invalidate() { doSetFrame_ = true; }

    // One iteration of the reference's frame loop as far as rendering is concerned: the
    // accumulate state machine (:572-590) followed by releasing the workers for ONE frame
    // (:592-595) with the globals as they then stand — including SIMPLEDRAW (preview shader),
    // the steps x steps progressive blocks (steps = ceil(1/(SCREEN_SCALE*scaler)), :233) anchored
    // at the 16 worker stripes (:330-340), and the reference's quirk that the first full
    // resolution frame after an edit has setFrame == true AND ACCUMULATIONFRAMES == 2.
    // The very first call renders with the start-up globals, as the workers do before the
    // loop's first pass.  Returns false when nothing was launched (ACC == TARGETFRAMES, :572).
    bool RenderFrame();
    // Picking (:525-541): x, y in window coordinates (y down, as the mouse reports it)
    int Pick(int mouse_x, int mouse_y);

    // Clean sequence used by benchmarks and fixtures: `count` further samples in ONE
    // launch; the first call after Invalidate() starts at sample 1 with reset.
    void RenderSamples(uint32_t count, bool count_rays = false);

    void Wait();
    bool Done();
    srt_stats Stats();
    // blit: copy the band into an SDL-surface-like buffer (renderSurface->pixels, :64)
    void ReadFramebuffer(void* pixels, size_t pitch_bytes);
    std::vector<float> ReadAccumulator();

    void PushCamera() { push_camera(); }  // srt_set_camera with the members as they stand (used by MultiGpuRenderer)

   private:
    void check(int rc, const char* what);
    void push_camera();
    srt_context* ctx_ = nullptr;
    int width_, height_;
    int row_begin_, row_end_;
    bool doSetFrame_ = false;
    bool first_frame_ = true;
    bool clean_reset_ = true;  // RenderSamples: next call starts at sample 1 with reset
    uint32_t next_clean_sample_ = 1;
};

// One frame over N GPUs of one node, in ONE process: the reference's split of a frame over 16 worker threads
// (disjoint column stripes of one surface, Raytracer.cpp:330-342) becomes disjoint bands of MEMORY rows, one
// PathTraceRenderer (= one srt_context, one stream) per device.  Pixels share nothing — the random stream is keyed
// by the absolute pixel — so the bands render with no exchange, and ONE gather joins them: every band is copied,
// device to device, into the first device's framebuffer (srt_gather_band: peer copies over xGMI, each peer on its
// own link to the root).  Rank-order concatenation of memory-row bands IS the final image.  The same device may be
// listed several times (N contexts on N streams of one GPU): that is how the class is tested on a one-GPU box.
class MultiGpuRenderer {
   public:
    MultiGpuRenderer(const std::vector<int>& devices, int width, int height);
    ~MultiGpuRenderer();
    MultiGpuRenderer(const MultiGpuRenderer&) = delete;
    MultiGpuRenderer& operator=(const MultiGpuRenderer&) = delete;

    size_t size() const { return parts_.size(); }
    PathTraceRenderer& part(size_t i) { return *parts_[i]; }
    // memory-row band of part i.  Equal bands (the first height % N parts take one more row) until the first
    // RenderSamples; from then on, by default, bands of equal ESTIMATED cost (below).
    void Band(size_t i, int* begin, int* end) const;
    // Bands of equal ESTIMATED cost for the current scene, camera and bounce count (srt_estimate_row_costs on part 0: a
    // device-side probe that runs the kernel's path pool over a quarter of the pixels for the frame's first 32 samples and counts
    // its loop trips — the work of about 8 sample-frames on ONE device, synchronous, deterministic), boundaries on multiples of 2
    // rows (8 and 16 were tried: too coarse for the 48..64-row floor bands of an 8-way 1080p split).  Equal bands
    // leave the GPUs that own sky idle: on Scene1 the slowest of 8 equal bands takes 2.2x the average.
    //
    // Who decides the bands (round 4; the round-3 class re-split before EVERY restarted accumulation, whatever it cost):
    //   * automatic (default): RenderSamples(count) makes the balanced split when the accumulation (re)starts — after SetScene,
    //     Configure, Invalidate: every band starts from sample 1 then anyway — AND the request is worth the probe:
    //     count >= AutoBalanceMinSamples() x devices (default 32 per device: the probe's 8 sample-frames on one device are then
    //     at most a quarter of what the devices are about to do).  A smaller request — a preview frame, a camera move with a few
    //     samples per frame — keeps the bands it has (equal ones at first).  A split made for the same scene, camera, field of
    //     view and bounce count is reused: a new seed or an Invalidate() alone never probes again.
    //   * BalanceBands(): the balanced split now, kept until scene, camera, field of view or bounces change.
    //   * UseEqualBands(true): north_star's literal equal bands, never re-split.
    //   * UseManualBands(true): the bands the caller gave the parts through part(i).SetRowBand() are left alone (the caller
    //     answers for their covering the frame); without it the automatic split REPLACES such bands.
    void BalanceBands();
    void UseEqualBands(bool equal);
    void UseManualBands(bool manual) { manual_bands_ = manual; }
    void SetAutoBalanceMinSamples(uint32_t per_device) { auto_min_samples_ = per_device; }
    uint32_t AutoBalanceMinSamples() const { return auto_min_samples_; }

    void SetScene(const Scene& scene);  // replicated: every device gets its own copy (a few KB; meshes: a few MB)
    void SetEnvironment(const srt_environment& env);
    // camera and settings of every part (the reference's globals; see PathTraceRenderer)
    void Configure(const Transform& camera, int fov, int max_bounces, uint32_t seed);
    void Invalidate();
    // `count` further samples on every band (all launches are enqueued before anything is waited for), then the
    // gather into part 0; after Wait() part 0 holds the whole frame
    void RenderSamples(uint32_t count, bool count_rays = false);
    void Wait();
    void ReadFramebuffer(void* pixels, size_t pitch_bytes);  // the whole frame, from part 0
    // per part: kernel time of its last launch and its ray count (imbalance of static bands, SURVEY §8e caveat)
    std::vector<srt_stats> Stats();

   private:
    std::vector<PathTraceRenderer*> parts_;
    std::vector<int> bounds_;  // band i = memory rows [bounds_[i], bounds_[i + 1])
    int width_, height_;
    bool equal_bands_ = false;    // UseEqualBands(true)
    bool manual_bands_ = false;   // UseManualBands(true)
    bool split_pending_ = true;   // the accumulation restarts with the next RenderSamples: the moment rows may change owners
    uint32_t auto_min_samples_ = 32;
    // what the current balanced split was made for (valid while balanced_): a split for the same inputs is reused
    bool balanced_ = false;
    unsigned long long scene_generation_ = 0, split_scene_generation_ = 0;
    Transform split_camera_{};
    int split_fov_ = 0, split_bounces_ = 0;
    bool SplitIsCurrent() const;
    void EqualBands();
};

}  // namespace srt_host
