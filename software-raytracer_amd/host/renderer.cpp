// renderer.cpp — see renderer.hpp.
#include "renderer.hpp"

namespace srt_host {

void PathTraceRenderer::check(int rc, const char* what) {
    if (rc != SRT_OK) {
        const char* msg = srt_last_error(ctx_);
        throw RendererError(rc, std::string(what) + ": " + (msg ? msg : "?"));
    }
}

PathTraceRenderer::PathTraceRenderer(int device, int width, int height)
    : width_(width), height_(height), row_begin_(0), row_end_(height) {
    int rc = srt_create(device, width, height, &ctx_);
    if (rc != SRT_OK) {
        const char* msg = srt_last_error(nullptr);
        throw RendererError(rc, std::string("srt_create: ") + (msg ? msg : "?"));
    }
}

PathTraceRenderer::~PathTraceRenderer() { srt_destroy(ctx_); }

void PathTraceRenderer::SetScene(const Scene& scene) {
    std::vector<srt_object> flat = scene.Flatten();
    std::vector<srt_mesh> meshes = scene.MeshViews();  // EXTENSION: geometry of "Mesh" renderers
    check(srt_set_meshes(ctx_, meshes.data(), meshes.size()), "srt_set_meshes");
    check(srt_set_scene(ctx_, flat.data(), flat.size()), "srt_set_scene");
    doSetFrame_ = true;
}

void PathTraceRenderer::SetEnvironment(const srt_environment& env) {
    check(srt_set_environment(ctx_, &env), "srt_set_environment");
    doSetFrame_ = true;
}

void PathTraceRenderer::SetRowBand(int begin, int end) {
    if (begin < 0 || end > height_ || begin >= end) throw RendererError(SRT_ERR_INVALID_ARG, "SetRowBand: bad band");
    row_begin_ = begin;
    row_end_ = end;
    doSetFrame_ = true;
}

void PathTraceRenderer::push_camera() {
    srt_camera c{};
    const Vec3* src[4] = {&camera.position, &camera.right, &camera.up, &camera.forward};
    float* dst[4] = {c.position, c.right, c.up, c.forward};
    for (int i = 0; i < 4; ++i) {
        dst[i][0] = src[i]->x;
        dst[i][1] = src[i]->y;
        dst[i][2] = src[i]->z;
    }
    c.fov_degrees = FOV;
    check(srt_set_camera(ctx_, &c), "srt_set_camera");
}

bool PathTraceRenderer::RenderFrame() {
    if (first_frame_) {
        first_frame_ = false;  // workers start with the initial globals (:30-35,47-48,271)
    } else {
        if (ACCUMULATIONFRAMES == TARGETFRAMES) return false;  // :572-574
        if (doSetFrame_) {                                     // :576-582
            setFrame = true;
            ACCUMULATIONFRAMES = 1;
            progressiveResolutionScaler = 1.0f / 4.0f;
            doSetFrame_ = false;
        } else {                                               // :583-590
            setFrame = false;
            if (progressiveResolutionScaler != 1) setFrame = true;
            progressiveResolutionScaler = 1;
            ACCUMULATIONFRAMES += SIMPLEDRAW ? 0 : 1;
        }
    }
    push_camera();
    srt_render_params p{};
    p.row_begin = row_begin_;
    p.row_end = row_end_;
    p.first_sample = (uint32_t)ACCUMULATIONFRAMES;
    p.sample_count = 1;
    p.max_bounces = MAXBOUNCES < 0 ? 0 : MAXBOUNCES;  // :475
    p.seed = seed;
    p.flags = (setFrame ? SRT_RENDER_RESET : 0u) | (SIMPLEDRAW ? SRT_RENDER_PREVIEW : 0u);
    p.steps = (int)ceil(1 / ((float)SCREEN_SCALE * progressiveResolutionScaler));  // :233
    p.stripe_width = (int)ceil(width_ / (THREADS)) + 1;                             // :330 (integer divide first)
    p.selected_object = selectedObject;
    check(srt_render(ctx_, &p), "srt_render");
    clean_reset_ = true;
    return true;
}

int PathTraceRenderer::Pick(int mouse_x, int mouse_y) {
    push_camera();
    int idx = -1;
    check(srt_pick(ctx_, mouse_x, height_ - mouse_y, &idx), "srt_pick");  // :532 y = SCREEN_HEIGHT - y
    return idx;
}

void PathTraceRenderer::RenderSamples(uint32_t count, bool count_rays) {
    if (count == 0) return;
    bool reset = doSetFrame_ || clean_reset_;
    if (reset) next_clean_sample_ = 1;
    doSetFrame_ = false;
    clean_reset_ = false;
    push_camera();
    srt_render_params p{};
    p.row_begin = row_begin_;
    p.row_end = row_end_;
    p.first_sample = next_clean_sample_;
    p.sample_count = count;
    p.max_bounces = MAXBOUNCES < 0 ? 0 : MAXBOUNCES;
    p.seed = seed;
    p.flags = (reset ? SRT_RENDER_RESET : 0) | (count_rays ? SRT_RENDER_COUNT_RAYS : 0);
    p.steps = 1;
    p.selected_object = -1;
    check(srt_render(ctx_, &p), "srt_render");
    next_clean_sample_ += count;
    ACCUMULATIONFRAMES = (int)(next_clean_sample_ - 1);
    first_frame_ = false;
    setFrame = false;
    progressiveResolutionScaler = 1;
}

void PathTraceRenderer::Wait() { check(srt_wait(ctx_), "srt_wait"); }

bool PathTraceRenderer::Done() {
    int d = 0;
    check(srt_poll(ctx_, &d), "srt_poll");
    return d != 0;
}

srt_stats PathTraceRenderer::Stats() {
    srt_stats s{};
    check(srt_get_stats(ctx_, &s), "srt_get_stats");
    return s;
}

void PathTraceRenderer::ReadFramebuffer(void* pixels, size_t pitch_bytes) {
    check(srt_read_framebuffer(ctx_, pixels, pitch_bytes, row_begin_, row_end_), "srt_read_framebuffer");
}

std::vector<float> PathTraceRenderer::ReadAccumulator() {
    std::vector<float> out((size_t)width_ * height_ * 4);
    check(srt_read_accumulator(ctx_, out.data()), "srt_read_accumulator");
    return out;
}

// ---- MultiGpuRenderer -------------------------------------------------------------------
MultiGpuRenderer::MultiGpuRenderer(const std::vector<int>& devices, int width, int height) : width_(width), height_(height) {
    if (devices.empty() || (int)devices.size() > height) throw RendererError(SRT_ERR_INVALID_ARG, "MultiGpuRenderer: need 1 <= devices <= height");
    bounds_.assign(devices.size() + 1, 0);
    try {
        for (int d : devices) parts_.push_back(new PathTraceRenderer(d, width, height));
        EqualBands();
        split_pending_ = true;  // (the first RenderSamples makes the default split)
    } catch (...) {
        for (PathTraceRenderer* p : parts_) delete p;
        throw;
    }
}

MultiGpuRenderer::~MultiGpuRenderer() {
    for (PathTraceRenderer* p : parts_) delete p;
}

void MultiGpuRenderer::Band(size_t i, int* begin, int* end) const {
    if (manual_bands_) {  // the caller's own bands (part(i).SetRowBand)
        parts_[i]->RowBand(begin, end);
        return;
    }
    *begin = bounds_[i];
    *end = bounds_[i + 1];
}

void MultiGpuRenderer::EqualBands() {
    const int n = (int)parts_.size(), q = height_ / n, r = height_ % n;
    for (int k = 0; k <= n; ++k) bounds_[(size_t)k] = k * q + (k < r ? k : r);
    for (int k = 0; k < n; ++k) parts_[(size_t)k]->SetRowBand(bounds_[(size_t)k], bounds_[(size_t)k + 1]);
    split_pending_ = false;
    balanced_ = false;
}

static bool same_vec(const Vec3& a, const Vec3& b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

// the balanced split in force was made for this scene, camera, field of view and bounce count (the probe's inputs that matter;
// its seed moves the estimate by noise only)
bool MultiGpuRenderer::SplitIsCurrent() const {
    if (!balanced_ || parts_.empty()) return false;
    const PathTraceRenderer& p0 = *parts_[0];
    return split_scene_generation_ == scene_generation_ && split_fov_ == p0.FOV && split_bounces_ == p0.MAXBOUNCES &&
           same_vec(split_camera_.position, p0.camera.position) && same_vec(split_camera_.right, p0.camera.right) &&
           same_vec(split_camera_.up, p0.camera.up) && same_vec(split_camera_.forward, p0.camera.forward);
}

void MultiGpuRenderer::UseEqualBands(bool equal) {
    equal_bands_ = equal;
    Invalidate();  // rows change owners: every band starts over
}

void MultiGpuRenderer::BalanceBands() {
    const int n = (int)parts_.size();
    if (n < 2) return;
    std::vector<float> cost((size_t)height_);
    PathTraceRenderer& p0 = *parts_[0];
    p0.Wait();
    p0.PushCamera();
    int rc = srt_estimate_row_costs(p0.handle(), p0.MAXBOUNCES < 0 ? 0 : p0.MAXBOUNCES, p0.seed, cost.data());
    if (rc != SRT_OK) {
        const char* msg = srt_last_error(p0.handle());
        throw RendererError(rc, std::string("srt_estimate_row_costs: ") + (msg ? msg : "?"));
    }
    std::vector<double> prefix((size_t)height_ + 1, 0.0);
    for (int i = 0; i < height_; ++i) prefix[(size_t)i + 1] = prefix[(size_t)i] + (cost[(size_t)i] > 0 ? cost[(size_t)i] : 0.0);
    const double total = prefix[(size_t)height_];
    bounds_[0] = 0;
    for (int k = 1; k < n; ++k) {  // first row whose prefix cost reaches k/n of the total, rounded to 2 rows, every band >= 1 row
        const int lo = bounds_[(size_t)k - 1] + 1, hi = height_ - (n - k);
        int i = lo;
        while (i < hi && prefix[(size_t)i] < total * k / n) ++i;
        const int j = ((i + 1) / 2) * 2;
        if (j >= lo && j <= hi) i = j;
        bounds_[(size_t)k] = i < lo ? lo : (i > hi ? hi : i);
    }
    bounds_[(size_t)n] = height_;
    for (int k = 0; k < n; ++k) parts_[(size_t)k]->SetRowBand(bounds_[(size_t)k], bounds_[(size_t)k + 1]);
    split_pending_ = false;
    balanced_ = true;
    split_scene_generation_ = scene_generation_;
    split_camera_ = p0.camera, split_fov_ = p0.FOV, split_bounces_ = p0.MAXBOUNCES;
}

void MultiGpuRenderer::SetScene(const Scene& scene) {
    for (PathTraceRenderer* p : parts_) p->SetScene(scene);
    ++scene_generation_;
    split_pending_ = true;
}

void MultiGpuRenderer::SetEnvironment(const srt_environment& env) {
    for (PathTraceRenderer* p : parts_) p->SetEnvironment(env);
}

void MultiGpuRenderer::Configure(const Transform& camera, int fov, int max_bounces, uint32_t seed) {
    for (PathTraceRenderer* p : parts_) {
        p->camera = camera;
        p->FOV = fov;
        p->MAXBOUNCES = max_bounces;
        p->seed = seed;
        p->Invalidate();
    }
    split_pending_ = true;
}

void MultiGpuRenderer::Invalidate() {
    for (PathTraceRenderer* p : parts_) p->Invalidate();
    split_pending_ = true;
}

void MultiGpuRenderer::RenderSamples(uint32_t count, bool count_rays) {
    if (split_pending_ && !manual_bands_) {  // the accumulation starts over: the moment rows may change owners
        // The balance probe is the work of about 8 sample-frames on ONE device, synchronous: worth it only for a request that is a
        // multiple of that per device, and not again for inputs the split in force was made for (renderer.hpp).
        if (equal_bands_) {
            EqualBands();
        } else if (SplitIsCurrent()) {
            // (same scene, camera, field of view and bounces: a new seed or an Invalidate() alone does not move the costs)
        } else if ((unsigned long long)count >= (unsigned long long)auto_min_samples_ * parts_.size()) {
            BalanceBands();
        } else if (balanced_) {
            EqualBands();  // a split made for other inputs is worth no more than equal bands, and the request is too small to probe for
        }
    }
    split_pending_ = false;
    for (PathTraceRenderer* p : parts_) p->RenderSamples(count, count_rays);  // asynchronous: one stream per part
    for (size_t i = 1; i < parts_.size(); ++i) {                              // the one gather
        int b, e;
        Band(i, &b, &e);
        int rc = srt_gather_band(parts_[0]->handle(), parts_[i]->handle(), b, e);
        if (rc != SRT_OK) {
            const char* msg = srt_last_error(parts_[0]->handle());
            throw RendererError(rc, std::string("srt_gather_band: ") + (msg ? msg : "?"));
        }
    }
}

void MultiGpuRenderer::Wait() {
    for (size_t i = parts_.size(); i-- > 0;) parts_[i]->Wait();  // part 0 last: its stream waits for every band
}

void MultiGpuRenderer::ReadFramebuffer(void* pixels, size_t pitch_bytes) {
    Wait();
    int rc = srt_read_framebuffer(parts_[0]->handle(), pixels, pitch_bytes, 0, height_);
    if (rc != SRT_OK) {
        const char* msg = srt_last_error(parts_[0]->handle());
        throw RendererError(rc, std::string("srt_read_framebuffer: ") + (msg ? msg : "?"));
    }
}

std::vector<srt_stats> MultiGpuRenderer::Stats() {
    std::vector<srt_stats> out;
    for (PathTraceRenderer* p : parts_) out.push_back(p->Stats());
    return out;
}

}  // namespace srt_host
