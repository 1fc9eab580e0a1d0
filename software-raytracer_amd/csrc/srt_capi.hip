// srt_capi.hip — implementation of include/srt_pathtrace.h on HIP (gfx950).
//
// Host duties only: flatten ObjectsToRender into the device scene image, fold the
// frame-constant camera terms of GetRayDirection (Raytracer.cpp:111-115: tanf, aspect),
// launch srt::pathtrace_kernel on the handle's stream, move buffers.  No CPU fallback:
// without a usable HIP device srt_create fails.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <new>
#include <vector>

#include "srt_kernel.hip.h"
#include "srt_pathtrace.h"

namespace {

thread_local char g_create_error[512] = "";

struct HostCamera {
    srt_camera cam;
    bool set = false;
};

}  // namespace

struct srt_context {
    int device = 0;
    int width = 0, height = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;  // launch stream (own or caller's)
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    bool launched = false;

    // device buffers
    float4* d_scene = nullptr;
    size_t scene_capacity_vec4 = 0;
    int scene_vec4 = 0, ns = 0, nb = 0;
    bool scene_set = false;
    std::vector<float4> h_scene;  // staging for the async upload

    uint32_t* d_fb_own = nullptr;
    float4* d_acc_own = nullptr;
    uint32_t* d_fb = nullptr;
    float4* d_acc = nullptr;
    unsigned long long* d_rays = nullptr;

    srt_environment env;
    HostCamera camera;
    srt_stats stats{};
    bool stats_pending = false;
    uint64_t pending_samples = 0;
    bool count_rays = false;
    int lds_limit_bytes = 64 * 1024;

    char error[512] = "";
};

namespace {

int fail(srt_context* ctx, int code, const char* fmt, ...) {
    char* dst = ctx ? ctx->error : g_create_error;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
    return code;
}

#define SRT_HIP(ctx, call)                                                                       \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(ctx, e_ == hipErrorOutOfMemory ? SRT_ERR_OOM : SRT_ERR_HIP, "%s: %s", #call, \
                        hipGetErrorString(e_));                                                  \
    } while (0)

float clamp0h(float v) { return v < 0 ? 0.0f : v; }

}  // namespace

extern "C" {

int srt_abi_version(void) { return SRT_ABI_VERSION; }

int srt_device_count(int* count) {
    if (!count) return SRT_ERR_INVALID_ARG;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        *count = 0;
        fail(nullptr, SRT_ERR_NO_DEVICE, "no HIP device (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
        return SRT_ERR_NO_DEVICE;
    }
    *count = n;
    return SRT_OK;
}

const char* srt_last_error(const srt_context* ctx) { return ctx ? ctx->error : g_create_error; }

int srt_environment_default(srt_environment* env) {
    if (!env) return SRT_ERR_INVALID_ARG;
    // SunDirection = float3(1,-1,-1).Normalized()   (Raytracer.cpp:55,264)
    float len = sqrtf(1.0f * 1.0f + -1.0f * -1.0f + -1.0f * -1.0f);
    env->sun_direction[0] = 1.0f / len;
    env->sun_direction[1] = -1.0f / len;
    env->sun_direction[2] = -1.0f / len;
    // SkyColor = Color(.2,.35,1.0f)*10.0f; HorizonColor = Color(1.0,0.9f,0.5f)*5.0f  (:56-57)
    env->sky_color[0] = clamp0h((float).2 * 10.0f);
    env->sky_color[1] = clamp0h((float).35 * 10.0f);
    env->sky_color[2] = clamp0h(1.0f * 10.0f);
    env->horizon_color[0] = clamp0h((float)1.0 * 5.0f);
    env->horizon_color[1] = clamp0h(0.9f * 5.0f);
    env->horizon_color[2] = clamp0h(0.5f * 5.0f);
    env->ground_color[0] = .08f;  // :58
    env->ground_color[1] = .06f;
    env->ground_color[2] = .03f;
    env->sun_color[0] = env->sun_color[1] = env->sun_color[2] = 500.0f;  // :59
    return SRT_OK;
}

int srt_create(int device, int width, int height, srt_context** out) {
    if (!out) return fail(nullptr, SRT_ERR_INVALID_ARG, "srt_create: out is NULL");
    *out = nullptr;
    if (width <= 0 || height <= 0 || (long long)width * height > 0x7fffffffLL)
        return fail(nullptr, SRT_ERR_INVALID_ARG, "srt_create: bad size %dx%d", width, height);
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, SRT_ERR_NO_DEVICE, "srt_create: no HIP device (%s); this library has no CPU fallback",
                    e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(nullptr, SRT_ERR_INVALID_ARG, "srt_create: device %d of %d", device, n);
    srt_context* ctx = new (std::nothrow) srt_context();
    if (!ctx) return fail(nullptr, SRT_ERR_OOM, "srt_create: host allocation failed");
    ctx->device = device;
    ctx->width = width;
    ctx->height = height;
    srt_environment_default(&ctx->env);
    auto bail = [&](hipError_t err, const char* what) {
        int code = fail(nullptr, err == hipErrorOutOfMemory ? SRT_ERR_OOM : SRT_ERR_HIP, "srt_create: %s: %s", what,
                        hipGetErrorString(err));
        srt_destroy(ctx);
        return code;
    };
    if ((e = hipSetDevice(device)) != hipSuccess) return bail(e, "hipSetDevice");
    if ((e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    ctx->stream = ctx->own_stream;
    if ((e = hipEventCreate(&ctx->ev_begin)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipEventCreate(&ctx->ev_end)) != hipSuccess) return bail(e, "hipEventCreate");
    const size_t px = (size_t)width * height;
    if ((e = hipMalloc((void**)&ctx->d_fb_own, px * sizeof(uint32_t))) != hipSuccess) return bail(e, "hipMalloc framebuffer");
    if ((e = hipMalloc((void**)&ctx->d_acc_own, px * sizeof(float4))) != hipSuccess) return bail(e, "hipMalloc accumulator");
    if ((e = hipMalloc((void**)&ctx->d_rays, sizeof(unsigned long long))) != hipSuccess) return bail(e, "hipMalloc counter");
    if ((e = hipMemsetAsync(ctx->d_fb_own, 0, px * sizeof(uint32_t), ctx->stream)) != hipSuccess) return bail(e, "hipMemset");
    if ((e = hipMemsetAsync(ctx->d_acc_own, 0, px * sizeof(float4), ctx->stream)) != hipSuccess) return bail(e, "hipMemset");
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return bail(e, "hipStreamSynchronize");
    ctx->d_fb = ctx->d_fb_own;
    ctx->d_acc = ctx->d_acc_own;
    int lds = 0;
    if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && lds > 0)
        ctx->lds_limit_bytes = lds;
    *out = ctx;
    return SRT_OK;
}

int srt_destroy(srt_context* ctx) {
    if (!ctx) return SRT_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->d_scene) (void)hipFree(ctx->d_scene);
    if (ctx->d_fb_own) (void)hipFree(ctx->d_fb_own);
    if (ctx->d_acc_own) (void)hipFree(ctx->d_acc_own);
    if (ctx->d_rays) (void)hipFree(ctx->d_rays);
    if (ctx->ev_begin) (void)hipEventDestroy(ctx->ev_begin);
    if (ctx->ev_end) (void)hipEventDestroy(ctx->ev_end);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return SRT_OK;
}

int srt_set_scene(srt_context* ctx, const srt_object* objects, size_t count) {
    if (!ctx) return SRT_ERR_INVALID_ARG;
    if (count && !objects) return fail(ctx, SRT_ERR_INVALID_ARG, "srt_set_scene: objects is NULL");
    if (count > 0x3fffffff) return fail(ctx, SRT_ERR_INVALID_ARG, "srt_set_scene: too many objects");
    SRT_HIP(ctx, hipSetDevice(ctx->device));
    int ns = 0, nb = 0;
    for (size_t i = 0; i < count; ++i) {
        if (objects[i].type == SRT_OBJ_SPHERE)
            ++ns;
        else if (objects[i].type == SRT_OBJ_BOX)
            ++nb;
        else if (objects[i].type != SRT_OBJ_NONE)
            return fail(ctx, SRT_ERR_INVALID_ARG, "srt_set_scene: object %zu has unknown type %d", i, objects[i].type);
    }
    const int ns4 = (ns + 3) & ~3;  // sphere block padded for the 4-wide scan
    const int nvec = ns4 + 2 * nb + 3 * (ns + nb);
    if ((size_t)nvec * sizeof(float4) > (size_t)ctx->lds_limit_bytes)
        return fail(ctx, SRT_ERR_INVALID_ARG,
                    "srt_set_scene: %d spheres + %d boxes need %zu B of LDS, limit %d B (tile streaming not built yet)", ns, nb,
                    (size_t)nvec * sizeof(float4), ctx->lds_limit_bytes);
    // the previous upload may still be in flight from h_scene
    SRT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->h_scene.assign((size_t)(nvec > 0 ? nvec : 1), make_float4(0, 0, 0, 0));
    float4* sph = ctx->h_scene.data();
    for (int i = ns; i < ns4; ++i) sph[i] = make_float4(0, 0, 0, -1.0f);  // d2 > r*r always: never a candidate
    float4* box = sph + ns4;
    float4* mat = box + 2 * nb;
    int is = 0, ib = 0;
    for (size_t i = 0; i < count; ++i) {
        const srt_object& o = objects[i];
        int p;
        if (o.type == SRT_OBJ_SPHERE) {
            // squaredRadius = sphereRadius * sphereRadius   (Object.hpp:122)
            sph[is] = make_float4(o.position[0], o.position[1], o.position[2], o.radius * o.radius);
            p = is++;
        } else if (o.type == SRT_OBJ_BOX) {
            box[2 * ib] = make_float4(o.position[0], o.position[1], o.position[2], 0.0f);
            box[2 * ib + 1] = make_float4(o.half_size[0], o.half_size[1], o.half_size[2], 0.0f);
            p = ns + ib++;
        } else {
            continue;  // inert Object: Raytrace() is never valid (Object.hpp:21-23)
        }
        const srt_material& m = o.material;
        float ord;
        int32_t idx = (int32_t)i;
        memcpy(&ord, &idx, 4);
        mat[3 * p + 0] = make_float4(m.smoothness, m.specular_amount, m.base_color[0], m.base_color[1]);
        mat[3 * p + 1] = make_float4(m.base_color[2], m.emissive_color[0], m.emissive_color[1], m.emissive_color[2]);
        mat[3 * p + 2] = make_float4(m.specular_color[0], m.specular_color[1], m.specular_color[2], ord);
    }
    if ((size_t)nvec > ctx->scene_capacity_vec4 || !ctx->d_scene) {
        if (ctx->d_scene) SRT_HIP(ctx, hipFree(ctx->d_scene));
        ctx->d_scene = nullptr;
        size_t cap = (size_t)(nvec > 0 ? nvec : 1);
        SRT_HIP(ctx, hipMalloc((void**)&ctx->d_scene, cap * sizeof(float4)));
        ctx->scene_capacity_vec4 = cap;
    }
    SRT_HIP(ctx, hipMemcpyAsync(ctx->d_scene, ctx->h_scene.data(), ctx->h_scene.size() * sizeof(float4), hipMemcpyHostToDevice,
                                ctx->stream));
    ctx->scene_vec4 = nvec;
    ctx->ns = ns;
    ctx->nb = nb;
    ctx->scene_set = true;
    return SRT_OK;
}

int srt_set_environment(srt_context* ctx, const srt_environment* env) {
    if (!ctx || !env) return SRT_ERR_INVALID_ARG;
    ctx->env = *env;
    return SRT_OK;
}

int srt_set_camera(srt_context* ctx, const srt_camera* camera) {
    if (!ctx || !camera) return SRT_ERR_INVALID_ARG;
    ctx->camera.cam = *camera;
    ctx->camera.set = true;
    return SRT_OK;
}

int srt_set_stream(srt_context* ctx, void* hip_stream) {
    if (!ctx) return SRT_ERR_INVALID_ARG;
    SRT_HIP(ctx, hipSetDevice(ctx->device));
    SRT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return SRT_OK;
}

int srt_bind_output(srt_context* ctx, void* d_framebuffer, void* d_accumulator) {
    if (!ctx) return SRT_ERR_INVALID_ARG;
    SRT_HIP(ctx, hipSetDevice(ctx->device));
    SRT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->d_fb = d_framebuffer ? (uint32_t*)d_framebuffer : ctx->d_fb_own;
    ctx->d_acc = d_accumulator ? (float4*)d_accumulator : ctx->d_acc_own;
    return SRT_OK;
}

int srt_device_framebuffer(srt_context* ctx, void** d_ptr) {
    if (!ctx || !d_ptr) return SRT_ERR_INVALID_ARG;
    *d_ptr = ctx->d_fb;
    return SRT_OK;
}

int srt_device_accumulator(srt_context* ctx, void** d_ptr) {
    if (!ctx || !d_ptr) return SRT_ERR_INVALID_ARG;
    *d_ptr = ctx->d_acc;
    return SRT_OK;
}

int srt_render(srt_context* ctx, const srt_render_params* p) {
    if (!ctx || !p) return SRT_ERR_INVALID_ARG;
    if (!ctx->scene_set) return fail(ctx, SRT_ERR_STATE, "srt_render: srt_set_scene has not been called");
    if (!ctx->camera.set) return fail(ctx, SRT_ERR_STATE, "srt_render: srt_set_camera has not been called");
    const int W = ctx->width, H = ctx->height;
    if (p->row_begin < 0 || p->row_end > H || p->row_begin >= p->row_end)
        return fail(ctx, SRT_ERR_INVALID_ARG, "srt_render: bad row band [%d,%d) for height %d", p->row_begin, p->row_end, H);
    if (p->first_sample < 1 || p->sample_count < 1 || p->max_bounces < 0)
        return fail(ctx, SRT_ERR_INVALID_ARG, "srt_render: first_sample/sample_count must be >= 1 and max_bounces >= 0");
    if ((uint64_t)p->first_sample + p->sample_count > 0x7fffffffull)
        return fail(ctx, SRT_ERR_INVALID_ARG, "srt_render: sample index overflows int (ACCUMULATIONFRAMES is an int)");
    SRT_HIP(ctx, hipSetDevice(ctx->device));

    srt::KernelParams K;
    memset(&K, 0, sizeof K);
    const srt_camera& c = ctx->camera.cam;
    // frame-constant part of GetRayDirection (Raytracer.cpp:107-115)
    const float clipDistance = .01f;
    float aspecRatio = (float)W / (float)H;
    float hFov = (float)(c.fov_degrees * 3.14159265358979323846 / 180.0f);
    float rd = (clipDistance * tanf(hFov / 2.0f)) * aspecRatio;
    float ld = (clipDistance * tanf(hFov / 2.0f));
    for (int i = 0; i < 3; ++i) {
        K.cam_pos[i] = c.position[i];
        K.right_rd[i] = c.right[i] * rd;
        K.up_ld[i] = c.up[i] * ld;
        K.fwd_clip[i] = c.forward[i] * clipDistance;
        K.sun_dir[i] = ctx->env.sun_direction[i];
        K.sky[i] = ctx->env.sky_color[i];
        K.horizon[i] = ctx->env.horizon_color[i];
        K.ground[i] = ctx->env.ground_color[i];
        K.sun[i] = ctx->env.sun_color[i];
    }
    K.width = W;
    K.height = H;
    K.y0 = H - p->row_end;  // memory rows [rb,re) = scene rows [H-re, H-rb)
    K.rows = p->row_end - p->row_begin;
    K.first_sample = p->first_sample;
    K.sample_count = p->sample_count;
    K.max_bounces = p->max_bounces;
    K.seed = p->seed;
    K.flags = p->flags & (SRT_RENDER_RESET | SRT_RENDER_COUNT_RAYS);
    K.ns = ctx->ns;
    K.nb = ctx->nb;
    K.scene_vec4 = ctx->scene_vec4;
    K.scene = ctx->d_scene;
    K.accumulator = ctx->d_acc;
    K.framebuffer = ctx->d_fb;
    K.ray_counter = ctx->d_rays;

    ctx->count_rays = (p->flags & SRT_RENDER_COUNT_RAYS) != 0;
    if (ctx->count_rays) SRT_HIP(ctx, hipMemsetAsync(ctx->d_rays, 0, sizeof(unsigned long long), ctx->stream));

    dim3 grid((unsigned)((W + srt::WG_W - 1) / srt::WG_W), (unsigned)((K.rows + srt::WG_H - 1) / srt::WG_H));
    dim3 block(srt::WG_THREADS);
    size_t lds_bytes = (size_t)(ctx->scene_vec4 > 0 ? ctx->scene_vec4 : 1) * sizeof(float4);
    SRT_HIP(ctx, hipEventRecord(ctx->ev_begin, ctx->stream));
    hipLaunchKernelGGL(srt::pathtrace_kernel, grid, block, lds_bytes, ctx->stream, K);
    SRT_HIP(ctx, hipGetLastError());
    SRT_HIP(ctx, hipEventRecord(ctx->ev_end, ctx->stream));
    ctx->launched = true;
    ctx->stats_pending = true;
    ctx->pending_samples = (uint64_t)W * (uint64_t)K.rows * p->sample_count;
    return SRT_OK;
}

int srt_wait(srt_context* ctx) {
    if (!ctx) return SRT_ERR_INVALID_ARG;
    SRT_HIP(ctx, hipSetDevice(ctx->device));
    SRT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SRT_OK;
}

int srt_poll(srt_context* ctx, int* done) {
    if (!ctx || !done) return SRT_ERR_INVALID_ARG;
    SRT_HIP(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipStreamQuery(ctx->stream);
    if (e == hipSuccess) {
        *done = 1;
        return SRT_OK;
    }
    if (e == hipErrorNotReady) {
        *done = 0;
        return SRT_OK;
    }
    return fail(ctx, SRT_ERR_HIP, "hipStreamQuery: %s", hipGetErrorString(e));
}

int srt_get_stats(srt_context* ctx, srt_stats* out) {
    if (!ctx || !out) return SRT_ERR_INVALID_ARG;
    if (!ctx->launched) return fail(ctx, SRT_ERR_STATE, "srt_get_stats: nothing rendered yet");
    SRT_HIP(ctx, hipSetDevice(ctx->device));
    SRT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->stats_pending) {
        float ms = 0.0f;
        SRT_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end));
        ctx->stats.kernel_ms = ms;
        ctx->stats.path_samples = ctx->pending_samples;
        ctx->stats.rays = 0;
        if (ctx->count_rays) {
            unsigned long long r = 0;
            SRT_HIP(ctx, hipMemcpy(&r, ctx->d_rays, sizeof r, hipMemcpyDeviceToHost));
            ctx->stats.rays = r;
        }
        ctx->stats_pending = false;
    }
    *out = ctx->stats;
    return SRT_OK;
}

int srt_read_framebuffer(srt_context* ctx, void* dst, size_t pitch_bytes, int row_begin, int row_end) {
    if (!ctx || !dst) return SRT_ERR_INVALID_ARG;
    const size_t rowb = (size_t)ctx->width * 4;
    if (row_begin < 0 || row_end > ctx->height || row_begin >= row_end || pitch_bytes < rowb)
        return fail(ctx, SRT_ERR_INVALID_ARG, "srt_read_framebuffer: bad rows [%d,%d) or pitch %zu", row_begin, row_end, pitch_bytes);
    SRT_HIP(ctx, hipSetDevice(ctx->device));
    SRT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SRT_HIP(ctx, hipMemcpy2D(dst, pitch_bytes, (const char*)ctx->d_fb + (size_t)row_begin * rowb, rowb, rowb,
                             (size_t)(row_end - row_begin), hipMemcpyDeviceToHost));
    return SRT_OK;
}

int srt_read_accumulator(srt_context* ctx, float* dst_rgba) {
    if (!ctx || !dst_rgba) return SRT_ERR_INVALID_ARG;
    SRT_HIP(ctx, hipSetDevice(ctx->device));
    SRT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SRT_HIP(ctx, hipMemcpy(dst_rgba, ctx->d_acc, (size_t)ctx->width * ctx->height * sizeof(float4), hipMemcpyDeviceToHost));
    return SRT_OK;
}

int srt_write_accumulator(srt_context* ctx, const float* src_rgba) {
    if (!ctx || !src_rgba) return SRT_ERR_INVALID_ARG;
    SRT_HIP(ctx, hipSetDevice(ctx->device));
    SRT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SRT_HIP(ctx, hipMemcpy(ctx->d_acc, src_rgba, (size_t)ctx->width * ctx->height * sizeof(float4), hipMemcpyHostToDevice));
    return SRT_OK;
}

}  // extern "C"
