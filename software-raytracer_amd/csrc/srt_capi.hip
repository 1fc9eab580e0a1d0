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
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <exception>
#include <new>
#include <vector>

#include "srt_kernel.hip.h"
#include "srt_launch_shape.h"
#include "srt_scene_image.h"
#include "srt_mesh_bvh.h"
#include "srt_pathtrace.h"

namespace {

thread_local char g_create_error[512] = "";

// Tuning switches.  The shipped library has none: every value below is a constant.  A development build
// (make -C software-raytracer_amd/csrc dev -> libsrt_pathtrace_dev.so, -DSRT_DEV) reads them from the
// environment for in-process A/B timing (tests/ab_bench.py); all settings produce identical bits.
#ifndef SRT_FILL_MIN
#define SRT_FILL_MIN 0.85  // mesh launches: simulated fill of the chip's workgroup slots below which a launch of >= 256 spp is cut into four sample chunks
#endif
#ifndef SRT_ORDER_MIN_WG
#define SRT_ORDER_MIN_WG 256  // fewest blocks of tiles for which a launch records costs and is dispatched in cost order (round 3: 512 -> 256,
                              // the 48..64-row bands of a cost-balanced 8-rank 1080p frame: -4..-5 %)
#endif
struct DevSwitches {
    int kernel = 0;        // SRT_KERNEL: tuning variant
    bool no_cluster = false;  // SRT_NO_CLUSTER
    int tile_h = 0;        // SRT_TILE_H: 8/4/2/1 forces the tile height
    int defer = -1;        // SRT_DEFER: 0 never chunk samples, n > 0 force n samples per chunk
    bool lpt = true;       // SRT_LPT=0: natural dispatch order
    int lpt_buckets = 128; // SRT_LPT_BUCKETS
    int chunk_beta = 30;   // SRT_CHUNK_BETA (percent): sample chunks from recorded block costs, see srt_render
    bool host_order = true;  // SRT_HOST_ORDER=0: no host-derived initial dispatch order
    int kernel_flags = 0;    // SRT_KFLAGS: extra KernelParams.flags bits of timing experiments
    bool chain = true;       // SRT_CHAIN=0: every chunk of a sample-chunked launch goes through the sample buffer (round 3's form)
};
#ifdef SRT_DEV
const DevSwitches& dev_switches() {
    static const DevSwitches sw = [] {
        DevSwitches d;
        auto geti = [](const char* name, int dflt) {
            const char* v = getenv(name);
            return v ? atoi(v) : dflt;
        };
        d.kernel = geti("SRT_KERNEL", 0);
        d.no_cluster = getenv("SRT_NO_CLUSTER") != nullptr;
        d.tile_h = geti("SRT_TILE_H", 0);
        d.defer = geti("SRT_DEFER", -1);
        d.lpt = geti("SRT_LPT", 1) != 0;
        int b = geti("SRT_LPT_BUCKETS", 128);
        d.lpt_buckets = b < 2 ? 2 : (b > 4096 ? 4096 : b);
        d.chunk_beta = geti("SRT_CHUNK_BETA", 30);
        d.host_order = geti("SRT_HOST_ORDER", 1) != 0;
        d.kernel_flags = geti("SRT_KFLAGS", 0);
        d.chain = geti("SRT_CHAIN", 1) != 0;
        return d;
    }();
    return sw;
}
#else
constexpr DevSwitches k_dev_switches{};
constexpr const DevSwitches& dev_switches() { return k_dev_switches; }
#endif

constexpr size_t REC_WORDS = 1 + 4 * srt::TALLY_N;  // per block in the cost record: wave time, then TALLY_N counts for each of the four waves
// srt_launch_shape.h is host-only C++ (unit-tested on the CPU) and repeats the kernel's tile geometry:
static_assert(srt::SHAPE_TILE_H == srt::TILE_H && srt::SHAPE_WG_W == srt::WG_W && srt::SHAPE_WG_H == srt::WG_H && srt::SHAPE_WG_TILES_Y == srt::WG_TILES_Y &&
              srt::SHAPE_WAVES_PER_WG == srt::WG_TILES_X * srt::WG_TILES_Y && srt::SHAPE_TALLY_N == srt::TALLY_N, "srt_launch_shape.h and srt_kernel.hip.h disagree");

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
    // two images of the same scene: [0] clustered (default), [1] plain brute force (A/B aid)
    float4* d_scene[2] = {nullptr, nullptr};
    size_t scene_capacity_vec4[2] = {0, 0};
    srt::SceneLayout layout[2];
    bool scene_set = false;
    std::vector<float4> h_scene[2];  // staging for the async upload

    uint32_t* d_fb_own = nullptr;
    float4* d_acc_own = nullptr;
    uint32_t* d_fb = nullptr;
    float4* d_acc = nullptr;
    unsigned long long* d_rays = nullptr;
    unsigned long long* d_work = nullptr;  // SRT_RENDER_COUNT_WORK: the launch's loop counts (srt::TALLY_ALL words)
    int* d_pick = nullptr;
    int last_pick[4] = {0, 0, 0, 0};  // list index, distance bits, primitive id, normal.z bits (debug)

    // EXTENSION: triangle meshes
    std::vector<srt::HostMesh> meshes;
    srt::MeshImage mesh_image;
    float4* d_bvh_nodes = nullptr;
    float4* d_bvh_tris = nullptr;
    int32_t* d_bvh_gidpos = nullptr;

    // sample-chunked launches (narrow row bands at high sample counts): sample colours + per-tile masks
    float4* d_samples = nullptr;
    size_t samples_capacity = 0;  // bytes
    unsigned long long* d_tile_masks = nullptr;
    size_t tile_masks_capacity = 0;  // entries
    uint32_t* d_tile_chain = nullptr;  // sample-chunked launches: chunks of a tile folded in order so far (KernelParams.tile_chain), same capacity

    // cost-ordered dispatch: a launch may record the ray count of every block of tiles; once that copy has
    // arrived (polled, never waited for) later launches of the same grid start the expensive blocks first
    uint32_t* d_wg_cost = nullptr;
    uint32_t* d_wg_est = nullptr;    // estimated block costs (block_cost_kernel), input of order_sort_kernel
    uint32_t* d_wg_order = nullptr;
    uint32_t* h_wg_cost = nullptr;   // pinned
    uint32_t* h_wg_order = nullptr;  // pinned
    size_t wg_capacity = 0;
    unsigned order_gx = 0, order_gy = 0;  // grid the order in d_wg_order was made for (0 = none)
    unsigned rec_gx = 0, rec_gy = 0;      // grid of the recording in flight
    bool recording = false;               // a cost copy is in flight (ev_cost)
    bool rec_has_work = false;            // ... and behind the times it holds the waves' loop counts (the recording launch was a TALLY instantiation)
    double rec_step_w = 0.0;              // what a pool step of the recorded launch weighs (probe_step_weight: depends on the scene's layout)
    int work_layout[4] = {0, 0, 0, 0};    // of the last launch: uniform spheres, clusters, spheres per cluster, boxes (srt_get_work_counts)
    bool order_stale = true;              // scene / camera changed since the costs were recorded
    // the launch-shape record (round 4): every block's WORK as the recording launch counted it — loop trips under the balance
    // probe's weights, not times — so the sample-chunk rule is a deterministic function of scene, camera, band and call history
    srt::WorkRecord work;                 // the launch-shape record of the current band (srt_launch_shape.h)
    int band_y0 = -1, band_rows = -1;     // the row band the order, the recording and the cost figures above belong to
    bool estimate_stale = true;           // the scene changed since the order was last estimated on the device
    bool order_disabled = false;          // buffers for the feedback could not be allocated
    hipEvent_t ev_cost = nullptr, ev_order = nullptr, ev_gather = nullptr, ev_read = nullptr;
    unsigned long long peer_asked = 0;    // srt_gather_band: destination devices this context has asked hipDeviceCanAccessPeer about (once per pair)
    unsigned long long peer_direct = 0;   // ... and those it may reach directly (peer access enabled)
    char gather_path[160] = "no gather yet";  // which way this context's last srt_gather_band went (srt_gather_path)

    srt_environment env;
    HostCamera camera;
    srt_stats stats{};
    bool stats_pending = false;
    uint64_t pending_samples = 0;
    uint32_t pending_chunks = 1;
    bool count_rays = false;
    bool count_work = false, count_work_valid = false;
    bool pending_timed = true;  // the last render was bracketed by ev_begin / ev_end (not SRT_RENDER_NO_TIMING)
    uint32_t pending_tile_rows = 8, pending_chunk_samples = 0, pending_shape_source = 0;
    srt_work_counts work_counts{};
    int lds_limit_bytes = 64 * 1024;
    int cu_count = 256;
    bool scene_in_lds[2] = {true, true};  // per scene image: does it fit into LDS next to the scratch?
    bool pick_in_lds[2] = {true, true};
    int variant = -1;  // >= 0 overrides SRT_KERNEL (set through srt_debug_set_variant)

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
    if ((e = hipMalloc((void**)&ctx->d_pick, 4 * sizeof(int))) != hipSuccess) return bail(e, "hipMalloc pick");
    if ((e = hipMalloc((void**)&ctx->d_work, srt::TALLY_ALL * sizeof(unsigned long long))) != hipSuccess) return bail(e, "hipMalloc work counters");
    if ((e = hipMemsetAsync(ctx->d_fb_own, 0, px * sizeof(uint32_t), ctx->stream)) != hipSuccess) return bail(e, "hipMemset");
    if ((e = hipMemsetAsync(ctx->d_acc_own, 0, px * sizeof(float4), ctx->stream)) != hipSuccess) return bail(e, "hipMemset");
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return bail(e, "hipStreamSynchronize");
    ctx->d_fb = ctx->d_fb_own;
    ctx->d_acc = ctx->d_acc_own;
    int lds = 0;
    if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && lds > 0)
        ctx->lds_limit_bytes = lds;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) ctx->cu_count = cus;
    *out = ctx;
    return SRT_OK;
}

int srt_destroy(srt_context* ctx) {
    if (!ctx) return SRT_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < 2; ++i)
        if (ctx->d_scene[i]) (void)hipFree(ctx->d_scene[i]);
    if (ctx->d_fb_own) (void)hipFree(ctx->d_fb_own);
    if (ctx->d_acc_own) (void)hipFree(ctx->d_acc_own);
    if (ctx->d_rays) (void)hipFree(ctx->d_rays);
    if (ctx->d_work) (void)hipFree(ctx->d_work);
    if (ctx->d_bvh_nodes) (void)hipFree(ctx->d_bvh_nodes);
    if (ctx->d_bvh_tris) (void)hipFree(ctx->d_bvh_tris);
    if (ctx->d_bvh_gidpos) (void)hipFree(ctx->d_bvh_gidpos);
    if (ctx->d_pick) (void)hipFree(ctx->d_pick);
    if (ctx->d_samples) (void)hipFree(ctx->d_samples);
    if (ctx->d_wg_cost) (void)hipFree(ctx->d_wg_cost);
    if (ctx->d_wg_est) (void)hipFree(ctx->d_wg_est);
    if (ctx->d_wg_order) (void)hipFree(ctx->d_wg_order);
    if (ctx->h_wg_cost) (void)hipHostFree(ctx->h_wg_cost);
    if (ctx->h_wg_order) (void)hipHostFree(ctx->h_wg_order);
    if (ctx->ev_cost) (void)hipEventDestroy(ctx->ev_cost);
    if (ctx->ev_order) (void)hipEventDestroy(ctx->ev_order);
    if (ctx->ev_gather) (void)hipEventDestroy(ctx->ev_gather);
    if (ctx->ev_read) (void)hipEventDestroy(ctx->ev_read);
    if (ctx->d_tile_masks) (void)hipFree(ctx->d_tile_masks);
    if (ctx->d_tile_chain) (void)hipFree(ctx->d_tile_chain);
    if (ctx->ev_begin) (void)hipEventDestroy(ctx->ev_begin);
    if (ctx->ev_end) (void)hipEventDestroy(ctx->ev_end);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return SRT_OK;
}

static int set_scene_impl(srt_context* ctx, const srt_object* objects, size_t count) {
    if (!ctx) return SRT_ERR_INVALID_ARG;
    if (count && !objects) return fail(ctx, SRT_ERR_INVALID_ARG, "srt_set_scene: objects is NULL");
    if (count > 0x3fffffff) return fail(ctx, SRT_ERR_INVALID_ARG, "srt_set_scene: too many objects");
    SRT_HIP(ctx, hipSetDevice(ctx->device));
    for (size_t i = 0; i < count; ++i) {
        int t = objects[i].type;
        if (t != SRT_OBJ_SPHERE && t != SRT_OBJ_BOX && t != SRT_OBJ_NONE && t != SRT_OBJ_MESH)
            return fail(ctx, SRT_ERR_INVALID_ARG, "srt_set_scene: object %zu has unknown type %d", i, t);
        if (t == SRT_OBJ_MESH && (objects[i].mesh < 0 || (size_t)objects[i].mesh >= ctx->meshes.size()))
            return fail(ctx, SRT_ERR_INVALID_ARG, "srt_set_scene: object %zu refers to mesh %d but %zu meshes are set (call srt_set_meshes first)",
                        i, objects[i].mesh, ctx->meshes.size());
    }
    // the previous upload may still be in flight from h_scene
    SRT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // from here on the context holds no scene until this call completes: a failure below must not leave
    // a half-replaced one (new image, freed BVH) for srt_render to launch on
    ctx->scene_set = false;
    const bool no_cluster = dev_switches().no_cluster;
    bool has_mesh = false;
    for (size_t i = 0; i < count; ++i) has_mesh = has_mesh || objects[i].type == SRT_OBJ_MESH;
    for (int v = 0; v < 2; ++v) {
        // with meshes both images are the same one, so that primitive ids agree with the BVH
        srt::SceneLayout L = srt::build_scene_image(objects, count, (v == 0 || has_mesh) && !no_cluster, ctx->h_scene[v]);
        srt::environment_rows(ctx->env, ctx->h_scene[v].data() + srt::SRT_CONST_ENV_ROW);
        // hit_key packs the list index in 15 bits and the primitive id in 16
        if (count >= 32768 || L.nsT + L.nb + L.nm >= 65536)
            return fail(ctx, SRT_ERR_INVALID_ARG, "srt_set_scene: %zu objects (%d sphere slots + %d boxes + %d meshes) exceed the 32767-object limit",
                        count, L.nsT, L.nb, L.nm);
        // an image that does not fit into LDS next to the per-wave scratch stays in HBM (slower kernel
        // instantiation, same bits); the pick kernel runs one wave, so it is judged separately.  The LDS instantiations also
        // take the square root of the sphere test in its short form (sqrt_window, srt_kernel.hip.h), which needs every r*r in
        // [2^-72, FLT_MAX]: a scene with a sphere outside that window runs the instantiations that keep the library sqrtf.
        const size_t image_bytes = (size_t)L.total_vec4 * sizeof(float4);
        const size_t mesh_scratch = has_mesh ? (size_t)(srt::WG_MESH_SCRATCH_BYTES) : 0;
        ctx->scene_in_lds[v] = L.radii_in_sqrt_window && image_bytes + (size_t)srt::WG_SCRATCH_BYTES + mesh_scratch <= (size_t)ctx->lds_limit_bytes;
        ctx->pick_in_lds[v] = L.radii_in_sqrt_window && image_bytes + srt::WAVE_SCRATCH_BYTES + srt::MESH_WAVE_BYTES <= (size_t)ctx->lds_limit_bytes;
        if (ctx->h_scene[v].size() > ctx->scene_capacity_vec4[v] || !ctx->d_scene[v]) {
            if (ctx->d_scene[v]) SRT_HIP(ctx, hipFree(ctx->d_scene[v]));
            ctx->d_scene[v] = nullptr;
            SRT_HIP(ctx, hipMalloc((void**)&ctx->d_scene[v], ctx->h_scene[v].size() * sizeof(float4)));
            ctx->scene_capacity_vec4[v] = ctx->h_scene[v].size();
        }
        SRT_HIP(ctx, hipMemcpyAsync(ctx->d_scene[v], ctx->h_scene[v].data(), ctx->h_scene[v].size() * sizeof(float4),
                                    hipMemcpyHostToDevice, ctx->stream));
        ctx->layout[v] = L;
    }
    // EXTENSION: flatten mesh objects into a world-space triangle list + BVH (HBM resident).  The size limits
    // of the device encoding (24-bit triangle ids, 26-bit node ids) are checked BEFORE the build and the uploads.
    {
        unsigned long long total_tris = 0;
        for (size_t i = 0; i < count; ++i)
            if (objects[i].type == SRT_OBJ_MESH) total_tris += ctx->meshes[(size_t)objects[i].mesh].indices.size() / 3;
        if (total_tris >= (1ull << 24))
            return fail(ctx, SRT_ERR_INVALID_ARG, "srt_set_scene: %llu mesh triangles exceed the limit of 2^24 - 1 per scene", total_tris);
    }
    srt::build_mesh_image(objects, count, ctx->meshes, ctx->layout[0].nsT + ctx->layout[0].nb, ctx->mesh_image);
    if (ctx->d_bvh_nodes) SRT_HIP(ctx, hipFree(ctx->d_bvh_nodes));
    if (ctx->d_bvh_tris) SRT_HIP(ctx, hipFree(ctx->d_bvh_tris));
    if (ctx->d_bvh_gidpos) SRT_HIP(ctx, hipFree(ctx->d_bvh_gidpos));
    ctx->d_bvh_nodes = ctx->d_bvh_tris = nullptr;
    ctx->d_bvh_gidpos = nullptr;
    if (ctx->mesh_image.n_tris > 0) {
        // strict depth-first traversal (the kernel's last resort) keeps at most 7 entries per level
        if (7 * ctx->mesh_image.max_depth + 80 > srt::MESH_Q)
            return fail(ctx, SRT_ERR_INVALID_ARG, "srt_set_scene: BVH too deep (%d levels)", ctx->mesh_image.max_depth);
        if (ctx->mesh_image.n_nodes >= (1 << 26) || ctx->mesh_image.n_tris >= (1 << 24))  // (item encoding of the traversal queues)
            return fail(ctx, SRT_ERR_INVALID_ARG, "srt_set_scene: mesh too large (%d triangles)", ctx->mesh_image.n_tris);
        for (int ax = 0; ax < 3; ++ax)  // keeps cell * slope finite in the kernel's plane distances
            if (!(fabsf(ctx->mesh_image.center[ax]) + ctx->mesh_image.half[ax] <= 1e9f))
                return fail(ctx, SRT_ERR_INVALID_ARG, "srt_set_scene: mesh coordinates beyond 1e9 are not supported");
        SRT_HIP(ctx, hipMalloc((void**)&ctx->d_bvh_nodes, ctx->mesh_image.nodes.size() * sizeof(float4)));
        SRT_HIP(ctx, hipMalloc((void**)&ctx->d_bvh_tris, ctx->mesh_image.tris.size() * sizeof(float4)));
        SRT_HIP(ctx, hipMemcpyAsync(ctx->d_bvh_nodes, ctx->mesh_image.nodes.data(), ctx->mesh_image.nodes.size() * sizeof(float4),
                                    hipMemcpyHostToDevice, ctx->stream));
        SRT_HIP(ctx, hipMemcpyAsync(ctx->d_bvh_tris, ctx->mesh_image.tris.data(), ctx->mesh_image.tris.size() * sizeof(float4),
                                    hipMemcpyHostToDevice, ctx->stream));
        SRT_HIP(ctx, hipMalloc((void**)&ctx->d_bvh_gidpos, ctx->mesh_image.gidpos.size() * sizeof(int32_t)));
        SRT_HIP(ctx, hipMemcpyAsync(ctx->d_bvh_gidpos, ctx->mesh_image.gidpos.data(), ctx->mesh_image.gidpos.size() * sizeof(int32_t),
                                    hipMemcpyHostToDevice, ctx->stream));
    }
    ctx->scene_set = true;
    ctx->order_stale = true;
    ctx->estimate_stale = true;
    ctx->work.clear();  // the recorded block work describes another scene
    // ... and so does a cost copy that may still be in flight, and the dispatch order made from the old scene's costs: both are
    // dropped (the stream was synchronised above, so nothing still writes h_wg_cost), the next launch estimates afresh
    ctx->recording = false;
    ctx->order_gx = ctx->order_gy = 0;
    return SRT_OK;
}

// Host-side allocation failures (std::bad_alloc from the image / BVH builders) must not cross the C boundary.
int srt_set_scene(srt_context* ctx, const srt_object* objects, size_t count) {
    try {
        return set_scene_impl(ctx, objects, count);
    } catch (const std::bad_alloc&) {
        if (ctx) ctx->scene_set = false;
        return fail(ctx, SRT_ERR_OOM, "srt_set_scene: host allocation failed");
    } catch (const std::exception& e) {
        if (ctx) ctx->scene_set = false;
        return fail(ctx, SRT_ERR_INVALID_ARG, "srt_set_scene: %s", e.what());
    }
}

static int set_meshes_impl(srt_context* ctx, const srt_mesh* meshes, size_t count) {
    if (!ctx) return SRT_ERR_INVALID_ARG;
    if (count && !meshes) return fail(ctx, SRT_ERR_INVALID_ARG, "srt_set_meshes: meshes is NULL");
    std::vector<srt::HostMesh> copy(count);
    for (size_t i = 0; i < count; ++i) {
        const srt_mesh& m = meshes[i];
        if ((m.vertex_count && !m.vertices) || (m.triangle_count && !m.indices))
            return fail(ctx, SRT_ERR_INVALID_ARG, "srt_set_meshes: mesh %zu has NULL arrays", i);
        if (m.triangle_count >= ((size_t)1 << 24))  // the same limit srt_set_scene applies to the scene's total
            return fail(ctx, SRT_ERR_INVALID_ARG, "srt_set_meshes: mesh %zu has %zu triangles, the limit is 2^24 - 1", i, m.triangle_count);
        copy[i].vertices.assign(m.vertices, m.vertices + 3 * m.vertex_count);
        copy[i].indices.assign(m.indices, m.indices + 3 * m.triangle_count);
    }
    ctx->meshes.swap(copy);
    return SRT_OK;
}

int srt_set_meshes(srt_context* ctx, const srt_mesh* meshes, size_t count) {
    try {
        return set_meshes_impl(ctx, meshes, count);
    } catch (const std::bad_alloc&) {
        return fail(ctx, SRT_ERR_OOM, "srt_set_meshes: host allocation failed");
    } catch (const std::exception& e) {
        return fail(ctx, SRT_ERR_INVALID_ARG, "srt_set_meshes: %s", e.what());
    }
}

int srt_set_environment(srt_context* ctx, const srt_environment* env) {
    if (!ctx || !env) return SRT_ERR_INVALID_ARG;
    ctx->env = *env;
    // the environment lives in the scene image's constants block (srt_scene_image.h): patch the uploaded images in place
    if (ctx->scene_set) {
        SRT_HIP(ctx, hipSetDevice(ctx->device));
        SRT_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (h_scene may still be the source of an upload in flight)
        for (int v = 0; v < 2; ++v) {
            float4* rows = ctx->h_scene[v].data() + srt::SRT_CONST_ENV_ROW;
            srt::environment_rows(ctx->env, rows);
            SRT_HIP(ctx, hipMemcpyAsync(ctx->d_scene[v] + srt::SRT_CONST_ENV_ROW, rows, 4 * sizeof(float4), hipMemcpyHostToDevice, ctx->stream));
        }
    }
    return SRT_OK;
}

int srt_set_camera(srt_context* ctx, const srt_camera* camera) {
    if (!ctx || !camera) return SRT_ERR_INVALID_ARG;
    ctx->camera.cam = *camera;
    ctx->camera.set = true;
    // block costs change with the view: the order learned for the previous view stays in use (for a moving camera it is
    // a better guess than a fresh probe estimate) until the next launch has recorded new costs
    ctx->order_stale = true;
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
    // (no synchronisation, round 4: a launch takes its buffer addresses when it is enqueued, so renders in flight keep writing
    // the buffers they were given and only later calls see the new ones — which is what lets a caller render frame k + 1 into a
    // second framebuffer while frame k is still being copied to the host, bench.py's overlapped read-back)
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

}  // extern "C"

static int fill_kernel_params(srt_context* ctx, const srt_render_params* p, srt::KernelParams& K, size_t& lds_bytes, int& use, int& img) {
    const int W = ctx->width, H = ctx->height;
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
    }
    K.width = W;
    K.height = H;
    K.y0 = H - p->row_end;  // memory rows [rb,re) = scene rows [H-re, H-rb)
    K.rows = p->row_end - p->row_begin;
    K.first_sample = p->first_sample;
    K.sample_count = p->sample_count;
    K.max_bounces = p->max_bounces;
    K.seed = p->seed;
    K.flags = p->flags & (SRT_RENDER_RESET | SRT_RENDER_COUNT_RAYS | SRT_RENDER_PREVIEW);
#ifdef SRT_DEV
    K.flags |= (uint32_t)dev_switches().kernel_flags;  // SRT_KFLAGS: timing experiments, see the kernel
#endif
    K.steps = p->steps > 1 ? p->steps : 1;
    K.stripe_width = p->stripe_width > 0 ? p->stripe_width : 0;
    K.selected = p->selected_object;
    use = ctx->variant >= 0 ? ctx->variant : dev_switches().kernel;
    img = (use == 2) ? 1 : 0;  // variant 2: plain brute-force image
    K.mesh_defer = use >= 100 && use < 200 ? use - 100 : 12;  // variants 100 + n: mesh phases wait for n rays
    K.mesh_wait = use >= 200 && use < 300 ? use - 200 : 3;   // variants 200 + w: ... for at most w steps
    if (use >= 300 && use < 500) K.mesh_defer = (use - 300) / 10, K.mesh_wait = (use - 300) % 10;  // variants 300 + 10 n + w: both
#ifdef SRT_DEV
    if (use >= 1000 && use < 2000) K.flags |= (uint32_t)(use - 1000) << 8;  // variants 1000 + f: kernel experiment flags f << 8
#endif
    const srt::SceneLayout& SL = ctx->layout[img];
    K.nu4 = SL.nu4;
    K.nu = SL.nu;
    K.nc = SL.nc;
    K.K = SL.K;
    K.nsT = SL.nsT;
    K.nb = SL.nb;
    if (SL.boxes_finite) K.flags |= srt::KF_BOXES_FINITE;
    K.off_bounds = SL.off_bounds;
    K.off_box = SL.off_box;
    K.off_mat = SL.off_mat;
    K.scene_vec4 = SL.total_vec4;
    K.scene = ctx->d_scene[img];
    K.bvh_nodes = ctx->d_bvh_nodes;
    K.bvh_tris = ctx->d_bvh_tris;
    K.bvh_gidpos = ctx->d_bvh_gidpos;
    K.n_tris = ctx->mesh_image.n_tris;
    for (int i = 0; i < 3; ++i) K.mesh_center[i] = ctx->mesh_image.center[i], K.mesh_half[i] = ctx->mesh_image.half[i];
    K.mesh_r1 = (ctx->mesh_image.half[0] + ctx->mesh_image.half[1] + ctx->mesh_image.half[2]) +
                (fabsf(ctx->mesh_image.center[0]) + fabsf(ctx->mesh_image.center[1]) + fabsf(ctx->mesh_image.center[2]));
    K.mesh_bs_radius = ctx->mesh_image.bs_radius;
    K.accumulator = ctx->d_acc;
    K.framebuffer = ctx->d_fb;
    K.ray_counter = ctx->d_rays;

    lds_bytes = (ctx->scene_in_lds[img] ? (size_t)(SL.total_vec4 > 0 ? SL.total_vec4 : 1) * sizeof(float4) : 0) +
                srt::WG_SCRATCH_BYTES + (ctx->mesh_image.n_tris > 0 ? srt::WG_MESH_SCRATCH_BYTES : 0);
    return SRT_OK;
}

// What a block costs, from its counts: the weights are wave instructions per trip of the loop counted (a pool step costs its fixed
// part plus the uniform-sphere groups, cluster bounds and boxes every step runs through), fitted on measured band times of
// configs 3 and 5, Scene3, Scene_indirect and config 4's scene (tools/band_fit.py, profiles/r03/band_fit*.txt).  Valid for a probe
// of PROBE_SAMPLES = 32 samples (shorter pools take more steps per sample) and, as relative weights, for a launch's own counts.
struct ProbeWeights {
    // a pool step: its fixed part + what every step runs through per group of four uniform spheres / cluster bound / box / mesh root test
    double step = 700.0, step_ugroup = 70.0, step_cluster = 12.0, step_box = 45.0, step_mesh = 60.0;
    // Round 4: refitted JOINTLY on the 2- / 4- / 8-way splits of configs 3 and 5 AND of Scene3 at 1080p / 512 spp — the held-out
    // scene round 3's weights failed on (mean / slowest 0.80 at N = 8: they had been fitted in a closed loop on configs 3 and 5 only,
    // where the group weight of 760 and the per-tile 5830 stood in for each other; per block a group of exact tests costs a
    // seventh of a step, profiles/r04/shape_fit.txt, and Scene3's rows differ in exactly that).  Scene2, Scene_indirect and config
    // 4's scene at 4K stay held out (profiles/emulated_ranks.json; tools/band_fit.py, profiles/r04/band_fit.txt).
    // Refitted once more at the end of round 4 on the final kernels (a pool step had become ~ 9 % cheaper, the exact rounds cheaper than
    // that; same three workloads, same tool; before: group 134, node_round 27, leaf_trip 200, mesh_phase 64, node_test 62, wave 80,
    // untraced_wave 74): mean / slowest at N = 8 0.941 / 0.934 / 0.933 on the fitted workloads, 0.962 / 0.991 / 0.927 on the held-out ones.
    double group = 220.0;          // four clustered spheres through the exact test for 64 items (with the scatter, shuffles and merge of its round)
    double node_round = 23.0, leaf_trip = 346.0, mesh_phase = 58.0, node_test = 63.0;  // BVH traversal: rounds, triangle trips, phases, child boxes per lane and round
    double wave = 114.0;           // per tile: staging, primary rays, ring — what every sample chunk of a real launch repeats
    double untraced_wave = 94.0;   // a tile with sample-independent pixels folds their colour sample by sample
};
static double probe_step_weight(const srt::KernelParams& K, const ProbeWeights& w) {
    return w.step + w.step_ugroup * ((K.nu + 3) / 4) + w.step_cluster * K.nc + w.step_box * K.nb + (K.n_tris > 0 ? w.step_mesh : 0.0);
}
static double probe_block_cost(const uint32_t* c, const srt::KernelParams& K, const ProbeWeights& w) {
    return probe_step_weight(K, w) * c[srt::TALLY_STEPS] + w.group * c[srt::TALLY_GROUPS] + w.node_round * c[srt::TALLY_NODE_ROUNDS] + w.leaf_trip * c[srt::TALLY_LEAF_TRIPS] +
           w.mesh_phase * c[srt::TALLY_MESH_PHASES] + w.wave * c[srt::TALLY_WAVES] + w.untraced_wave * c[srt::TALLY_UNTRACED_WAVES] + w.node_test * c[srt::TALLY_NODE_TESTS];
}

// A recording launch's cost copy has completed: make the dispatch order of the following launches from the blocks' wave TIMES
// (any order gives the same image) and the launch-shape record from the blocks' WORK (counts: the same in every run).
static int consume_record(srt_context* ctx) {
    ctx->recording = false;
    const size_t n = (size_t)ctx->rec_gx * ctx->rec_gy;
    if (ctx->order_gx) (void)hipEventSynchronize(ctx->ev_order);  // (long done) the previous upload read h_wg_order
    // linear buckets between the cheapest and the dearest block, expensive first; the counting sort
    // keeps the spatial order inside a bucket
    const int NB = dev_switches().lpt_buckets;
    // a launch whose blocks all cost about the same (5th..95th percentile within 1.5x) keeps the
    // natural order: nothing to gain, and neighbouring blocks stay together
    bool uniform = false;
    {
        std::vector<uint32_t> tmp(ctx->h_wg_cost, ctx->h_wg_cost + n);
        std::nth_element(tmp.begin(), tmp.begin() + n / 20, tmp.end());
        const double p05 = (double)tmp[n / 20];
        std::nth_element(tmp.begin(), tmp.begin() + (n - 1 - n / 20), tmp.end());
        const double p95 = (double)tmp[n - 1 - n / 20];
        uniform = p95 <= 1.5 * p05;
    }
    uint32_t lo = 0xFFFFFFFFu, hi = 0;
    for (size_t i = 0; i < n; ++i) lo = ctx->h_wg_cost[i] < lo ? ctx->h_wg_cost[i] : lo, hi = ctx->h_wg_cost[i] > hi ? ctx->h_wg_cost[i] : hi;
    // the launch-shape record: behind the times, the loop counts of every wave (present when the recording launch kept them)
    ctx->work.clear();
    if (ctx->rec_has_work) srt::weigh_record(ctx->h_wg_cost + n, n, ctx->rec_gx, ctx->rec_gy, ctx->rec_step_w, ctx->mesh_image.n_tris > 0, ctx->work);
#ifdef SRT_DEV
    if (getenv("SRT_DEBUG_CHUNKS")) {  // the time-based figures of round 3 next to the counted ones, for calibration
        double tsum = 0.0;
        for (size_t i = 0; i < n; ++i) tsum += (double)ctx->h_wg_cost[i];
        float ms = 0.0f;
        double tfill = 0.0;
        if (ctx->pending_timed && hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end) == hipSuccess && ms > 0.0f)
            tfill = tsum * 1e-5 / ((double)ms * ctx->cu_count * (ctx->mesh_image.n_tris > 0 ? 16.0 : 20.0));
        else
            (void)hipGetLastError();
        const double slots = (double)ctx->cu_count * (ctx->mesh_image.n_tris > 0 ? 3.0 : 4.0);
        fprintf(stderr, "record: %zu blocks grid %u x %u | TIME dearest %u sum %.0f ratio %.3f fill %.3f (%.3f ms) | WORK dearest %.0f sum %.0f ratio %.3f", n, ctx->rec_gx, ctx->rec_gy,
                hi, tsum, tsum > 0 ? hi * slots / tsum : 0.0, tfill, ms, ctx->work.max, ctx->work.sum, ctx->work.sum > 0 ? ctx->work.max * slots / ctx->work.sum : 0.0);
        const int wslots = ctx->cu_count * (ctx->mesh_image.n_tris > 0 ? 4 : 5);
        for (int c : {1, 2, 3, 4, 6, 8}) fprintf(stderr, " fill(%d)=%.3f", c, srt::simulate_fill(ctx->work.blocks, c, wslots));
        fprintf(stderr, "\n");
    }
    if (const char* dump = getenv("SRT_DUMP_RECORD")) {  // development aid (tools/shape_fit.py): the raw record, appended as one binary blob
        if (FILE* f = fopen(dump, "ab")) {
            float ms = 0.0f;
            if (!ctx->pending_timed || hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end) != hipSuccess) ms = 0.0f, (void)hipGetLastError();
            uint32_t ms_bits, sw_bits;
            const float sw = (float)ctx->rec_step_w;
            memcpy(&ms_bits, &ms, 4), memcpy(&sw_bits, &sw, 4);
            const uint32_t head[10] = {0x53525452u, ctx->rec_gx, ctx->rec_gy, (uint32_t)ctx->band_y0, (uint32_t)ctx->band_rows, ms_bits, sw_bits,
                                       (uint32_t)ctx->cu_count, ctx->mesh_image.n_tris > 0 ? 1u : 0u, ctx->rec_has_work ? 1u : 0u};
            fwrite(head, 4, 10, f);
            fwrite(ctx->h_wg_cost, 4, n * REC_WORDS, f);
            fclose(f);
        }
    }
#endif
    const double scale = hi > lo ? (double)(NB - 1) / (double)(hi - lo) : 0.0;
    auto bucket = [&](uint32_t c) { return (NB - 1) - (int)((double)(c - lo) * scale); };
    std::vector<size_t> start((size_t)NB + 1, 0);
    for (size_t i = 0; i < n; ++i) ++start[(size_t)bucket(ctx->h_wg_cost[i]) + 1];
    for (int k = 0; k < NB; ++k) start[(size_t)k + 1] += start[(size_t)k];
    for (size_t i = 0; i < n; ++i) ctx->h_wg_order[start[(size_t)bucket(ctx->h_wg_cost[i])]++] = (uint32_t)i;
    if (uniform)
        for (size_t i = 0; i < n; ++i) ctx->h_wg_order[i] = (uint32_t)i;
    SRT_HIP(ctx, hipMemcpyAsync(ctx->d_wg_order, ctx->h_wg_order, n * 4, hipMemcpyHostToDevice, ctx->stream));
    SRT_HIP(ctx, hipEventRecord(ctx->ev_order, ctx->stream));
    ctx->order_gx = ctx->rec_gx, ctx->order_gy = ctx->rec_gy;
    return SRT_OK;
}

extern "C" {

int srt_render(srt_context* ctx, const srt_render_params* p) {
    if (!ctx || !p) return SRT_ERR_INVALID_ARG;
    if (!ctx->scene_set) return fail(ctx, SRT_ERR_STATE, "srt_render: srt_set_scene has not been called");
    if (!ctx->camera.set) return fail(ctx, SRT_ERR_STATE, "srt_render: srt_set_camera has not been called");
    const int W = ctx->width, H = ctx->height;
    if (p->row_begin < 0 || p->row_end > H || p->row_begin >= p->row_end)
        return fail(ctx, SRT_ERR_INVALID_ARG, "srt_render: bad row band [%d,%d) for height %d", p->row_begin, p->row_end, H);
    if (p->first_sample < 1 || p->sample_count < 1 || p->max_bounces < 0)
        return fail(ctx, SRT_ERR_INVALID_ARG, "srt_render: first_sample/sample_count must be >= 1 and max_bounces >= 0");
    if (p->steps < 0 || p->stripe_width < 0)
        return fail(ctx, SRT_ERR_INVALID_ARG, "srt_render: steps and stripe_width must be >= 0");
    if (p->sample_count > (1u << 20))
        return fail(ctx, SRT_ERR_INVALID_ARG, "srt_render: sample_count is limited to 2^20 per call (render in several calls)");
    if ((uint64_t)p->first_sample + p->sample_count > 0x7fffffffull)
        return fail(ctx, SRT_ERR_INVALID_ARG, "srt_render: sample index overflows int (ACCUMULATIONFRAMES is an int)");
    SRT_HIP(ctx, hipSetDevice(ctx->device));

    srt::KernelParams K;
    size_t lds_bytes = 0;
    int use = 0;
    int img = 0;
    const int fill_rc = fill_kernel_params(ctx, p, K, lds_bytes, use, img);
    if (fill_rc != SRT_OK) return fill_rc;
    ctx->count_rays = (p->flags & SRT_RENDER_COUNT_RAYS) != 0;
    if (ctx->count_rays) SRT_HIP(ctx, hipMemsetAsync(ctx->d_rays, 0, sizeof(unsigned long long), ctx->stream));
    // The learned dispatch order, a cost copy in flight and the work record of the chunk rule describe ONE row band.  Another
    // band of the same height has the same grid but other blocks behind every index: it starts from a fresh estimate, like a new
    // scene.  (Found the hard way: a 270-row band of config 5 launched after its neighbour inherited "no sample chunks" and
    // ran 46 ms instead of 12.)
    if (K.y0 != ctx->band_y0 || K.rows != ctx->band_rows) {
        if (ctx->recording) (void)hipEventSynchronize(ctx->ev_cost);  // (nothing may still write h_wg_cost when the next record starts)
        ctx->band_y0 = K.y0, ctx->band_rows = K.rows;
        ctx->order_stale = ctx->estimate_stale = true;
        ctx->work.clear();
        ctx->recording = false;
        ctx->order_gx = ctx->order_gy = 0;
    }
    // A record in flight is WAITED for (round 3 polled it): the launch after a recording launch always sees the record, whatever the
    // host's timing — so which launch of a sequence changes to the recorded shape and order does not vary from run to run.  The
    // wait ends when the recording launch does; it happens once per scene / camera / band change (a launch records only then),
    // and costs the one enqueue that could have overlapped that launch's tail.
    if (ctx->recording) {
        SRT_HIP(ctx, hipEventSynchronize(ctx->ev_cost));
        const int rc = consume_record(ctx);
        if (rc != SRT_OK) return rc;
    }

    // Tile height: with few rows and many samples per pixel (a narrow stripe of a multi-GPU frame) 8-row
    // tiles give too few workgroups to fill 256 CUs x 4 resident workgroups and leave nothing to balance
    // the tail with; halve the tile (twice the workgroups, same lanes at work in each wave's path pool)
    // until there are about four rounds of workgroups.  Results do not depend on the tiling.
    // Progressive blocks (steps > 1): when every pixel of a block ends up with a value that one lane can produce — the
    // launch starts the frame (all pixels of a block then hold the same running mean) or adds ONE sample (each pixel folds
    // the block's colour into its own mean) — the launch's lanes are blocks, not pixels: 1 / steps^2 of the lanes, one
    // ray per block, steps^2 pixel stores per lane (Raytracer.cpp:235-248).  Anything else (several samples onto an
    // accumulated frame) keeps one lane per pixel with the block's ray traced once per wave tile.
    const bool bgrid = K.steps > 1 && ((K.flags & SRT_RENDER_RESET) || p->sample_count == 1);
    long long grid_w = W, grid_h = K.rows;
    if (bgrid) {
        const long long sw = K.stripe_width > 0 ? K.stripe_width : W;
        grid_w = ((W + sw - 1) / sw) * ((sw + K.steps - 1) / K.steps);
        grid_h = (K.y0 + K.rows - 1) / K.steps - K.y0 / K.steps + 1;  // block rows that meet the band (scene rows)
        K.flags |= srt::KF_BLOCK_GRID;
    }
    K.bgrid_w = (int32_t)grid_w, K.bgrid_h = (int32_t)grid_h;
    // Tile height, sample chunks and the taper of the last chunks: srt_launch_shape.h — a pure function of the request, the grid, the
    // CU count and the band's work record (counts, not times; round 3 read the blocks' wave times and the launch's event time, and
    // config 5's rank-4 band flipped between one piece and nine layers).  Same inputs and call history, same shape; unit-tested on
    // the CPU (tests/native/shape_check.cpp).  The one thing outside it: whether the sample buffer can be had.
    srt::ShapeRequest req;
    req.grid_w = grid_w, req.grid_h = grid_h, req.rows = K.rows, req.sample_count = p->sample_count, req.steps = K.steps;
    req.block_grid = bgrid, req.mesh = K.n_tris > 0, req.cu_count = ctx->cu_count;
    srt::ShapeOverrides ov;
    ov.tile_h = dev_switches().tile_h, ov.defer = dev_switches().defer, ov.chunk_beta = dev_switches().chunk_beta;
    ov.no_taper = (dev_switches().kernel_flags & 0x400) != 0, ov.fill_min = SRT_FILL_MIN;
    srt::LaunchShape shape = srt::plan_launch_shape(req, &ctx->work, ov);
#ifdef SRT_DEV
    if (getenv("SRT_DEBUG_CHUNKS") && shape.source)
        fprintf(stderr, "chunks: dearest %.0f sum %.0f blocks %lld ratio %.3f fill %.3f -> %d layer(s) of %d\n", ctx->work.max, ctx->work.sum, shape.wg8, shape.ratio, shape.fill, shape.chunks, shape.chunk);
#endif
    const long long wg_x = shape.wg_x, wg8 = shape.wg8;
    if (shape.chunks >= 2) {
        // Costs 1 KiB of HBM per tile and sample; falls back to small tiles when that is not available
        const size_t tiles = (size_t)wg8 * srt::WG_TILES_X * srt::WG_TILES_Y;
        const size_t need = tiles * (size_t)p->sample_count * 64 * sizeof(float4);
        bool ok = need <= ((size_t)96 << 30);  // (a third of the 288 GB; 24 GB until round 3 — the sky half of config 5 at 4K x 1024 spp needs 67)
        if (ok && need > ctx->samples_capacity) {  // growing: take at most a quarter of what is free
            size_t free_b = 0, total_b = 0;
            ok = hipMemGetInfo(&free_b, &total_b) == hipSuccess && need <= (free_b + ctx->samples_capacity) / 4;
        }
        if (ok && need > ctx->samples_capacity) {
            if (ctx->d_samples) (void)hipFree(ctx->d_samples);
            ctx->d_samples = nullptr;
            ctx->samples_capacity = 0;
            if (hipMalloc((void**)&ctx->d_samples, need) == hipSuccess) ctx->samples_capacity = need;
            else ok = false, (void)hipGetLastError();
        }
        if (ok && tiles > ctx->tile_masks_capacity) {
            if (ctx->d_tile_masks) (void)hipFree(ctx->d_tile_masks);
            if (ctx->d_tile_chain) (void)hipFree(ctx->d_tile_chain);
            ctx->d_tile_masks = nullptr;
            ctx->d_tile_chain = nullptr;
            ctx->tile_masks_capacity = 0;
            if (hipMalloc((void**)&ctx->d_tile_masks, tiles * sizeof(unsigned long long)) == hipSuccess &&
                hipMalloc((void**)&ctx->d_tile_chain, tiles * sizeof(uint32_t)) == hipSuccess)
                ctx->tile_masks_capacity = tiles;
            else
                ok = false, (void)hipGetLastError();
        }
        if (!ok) srt::shape_without_sample_buffer(shape);  // no room for the sample buffer: small tiles instead
    }
    srt::finish_launch_shape(shape, p->sample_count, ov);
    const bool defer = shape.chunks >= 2;
    const int tile_h = shape.tile_h, chunk = shape.chunk, chunks = shape.chunks;
    const uint32_t shape_source = shape.source;
    K.tile_h = tile_h;
    K.chunk = defer ? chunk : 0;
    K.chunk_full = shape.chunk_full;
    K.sample_rows = ctx->d_samples;
    K.tile_masks = ctx->d_tile_masks;
    // chained chunks: a chunk that finds its tile's running mean at its own first sample folds its samples itself (srt_kernel.hip.h,
    // KernelParams.tile_chain); the counts start at zero
    K.tile_chain = defer && dev_switches().chain ? ctx->d_tile_chain : nullptr;
    K.chunk_layers = chunks;
    if (K.tile_chain) SRT_HIP(ctx, hipMemsetAsync(K.tile_chain, 0, (size_t)wg8 * srt::WG_TILES_X * srt::WG_TILES_Y * sizeof(uint32_t), ctx->stream));
    dim3 grid((unsigned)wg_x, (unsigned)((grid_h + tile_h * srt::WG_TILES_Y - 1) / (tile_h * srt::WG_TILES_Y)), (unsigned)chunks);
    dim3 block(srt::WG_THREADS);
    // Cost-ordered dispatch.  The hardware starts workgroups in linear order; with the natural order the
    // last ones to start are whatever lies at the top of the band, and the chip idles while a few expensive
    // blocks finish.  Starting blocks in order of decreasing cost (coarse buckets, so that neighbours stay
    // together) removes most of that tail: Scene1 3.39 -> 3.25 ms, config 4 19.7 -> 17.5 ms.  Costs are
    // the blocks' wave-cycles in the band's recording launch (the first after a scene, camera or band change), copied back
    // asynchronously; the next srt_render of the handle waits for that copy (see above).  Any order gives the same image.
    const bool order_env = dev_switches().lpt;
    bool record = false;
    const size_t nwg = (size_t)grid.x * grid.y;
    const bool in_lds = ctx->scene_in_lds[img];
    if (order_env && !ctx->order_disabled && nwg >= SRT_ORDER_MIN_WG && p->sample_count >= 4 && !(p->flags & SRT_RENDER_PREVIEW) && !bgrid) {
        if (nwg > ctx->wg_capacity) {
            SRT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->d_wg_cost) (void)hipFree(ctx->d_wg_cost);
            if (ctx->d_wg_est) (void)hipFree(ctx->d_wg_est);
            if (ctx->d_wg_order) (void)hipFree(ctx->d_wg_order);
            if (ctx->h_wg_cost) (void)hipHostFree(ctx->h_wg_cost);
            if (ctx->h_wg_order) (void)hipHostFree(ctx->h_wg_order);
            ctx->d_wg_cost = ctx->d_wg_est = ctx->d_wg_order = ctx->h_wg_cost = ctx->h_wg_order = nullptr;
            ctx->wg_capacity = 0;
            ctx->order_gx = ctx->order_gy = 0;
            ctx->recording = false;
            // an optimisation must not be able to fail a render: if anything here cannot be had (pinned host
            // memory, for one), the handle simply keeps the natural order from now on
            // (cost buffers: the blocks' wave times, then the loop counts of every wave of every block)
            const bool ok = hipMalloc((void**)&ctx->d_wg_cost, REC_WORDS * nwg * 4) == hipSuccess && hipMalloc((void**)&ctx->d_wg_order, nwg * 4) == hipSuccess &&
                            hipMalloc((void**)&ctx->d_wg_est, 2 * nwg * 4) == hipSuccess &&  // raw + smoothed
                            hipHostMalloc((void**)&ctx->h_wg_cost, REC_WORDS * nwg * 4, hipHostMallocDefault) == hipSuccess &&
                            hipHostMalloc((void**)&ctx->h_wg_order, nwg * 4, hipHostMallocDefault) == hipSuccess &&
                            (ctx->ev_cost || hipEventCreateWithFlags(&ctx->ev_cost, hipEventDisableTiming) == hipSuccess) &&
                            (ctx->ev_order || hipEventCreateWithFlags(&ctx->ev_order, hipEventDisableTiming) == hipSuccess);
            if (ok) {
                ctx->wg_capacity = nwg;
            } else {
                (void)hipGetLastError();
                ctx->order_disabled = true;
            }
        }
    }
    if (order_env && !ctx->order_disabled && ctx->wg_capacity >= nwg && nwg >= SRT_ORDER_MIN_WG && p->sample_count >= 4 && !(p->flags & SRT_RENDER_PREVIEW) && !bgrid) {
        // No recorded costs for this frame yet (first launch, or the scene has changed): estimate the blocks'
        // costs on the device — 16 one-sample probe paths per block, block_cost_kernel — and sort them there (order_sort_kernel);
        // both run on the launch stream ahead of the frame, nothing comes back to the host.  Measured with warm clocks:
        // first launch of a frame vs the learned order: config 4 +13 % -> see DESIGN.md, Scene1 +3 %.
        if (dev_switches().host_order && (ctx->estimate_stale || ctx->order_gx != grid.x || ctx->order_gy != grid.y)) {
            const size_t image_bytes = (size_t)(K.scene_vec4 > 0 ? K.scene_vec4 : 1) * sizeof(float4);
            const size_t est_lds = (ctx->pick_in_lds[img] ? image_bytes : 0) + srt::WAVE_SCRATCH_BYTES + srt::MESH_WAVE_BYTES;
            if (ctx->pick_in_lds[img])
                hipLaunchKernelGGL(srt::block_cost_kernel<true>, dim3((unsigned)((nwg + 3) / 4)), dim3(64), est_lds, ctx->stream, K, ctx->d_wg_est, (int)grid.x, (int)nwg);
            else
                hipLaunchKernelGGL(srt::block_cost_kernel<false>, dim3((unsigned)((nwg + 3) / 4)), dim3(64), est_lds, ctx->stream, K, ctx->d_wg_est, (int)grid.x, (int)nwg);
            hipLaunchKernelGGL(srt::smooth_cost_kernel, dim3((unsigned)((nwg + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_wg_est, ctx->d_wg_est + nwg, (int)nwg, (int)grid.x);
            hipLaunchKernelGGL(srt::order_sort_kernel, dim3(1), dim3(srt::ORDER_SORT_THREADS), 0, ctx->stream, ctx->d_wg_est + nwg, ctx->d_wg_order, (int)nwg);
            if (hipGetLastError() == hipSuccess) {
                ctx->order_gx = grid.x, ctx->order_gy = grid.y;
                ctx->estimate_stale = false;
            } else {
                ctx->order_gx = ctx->order_gy = 0;  // an optimisation must not fail the render: natural order
            }
        }
        if (ctx->order_gx == grid.x && ctx->order_gy == grid.y) K.wg_order = ctx->d_wg_order;
        if (ctx->order_stale || ctx->order_gx != grid.x || ctx->order_gy != grid.y) {  // (no record is in flight here: it was waited for above)
            SRT_HIP(ctx, hipMemsetAsync(ctx->d_wg_cost, 0, REC_WORDS * nwg * 4, ctx->stream));
            K.wg_cost = ctx->d_wg_cost;
            K.wg_blocks = (uint32_t)nwg;
            record = true;
        }
    }
    // The TALLY instantiations keep the wave-uniform loop counts (srt_kernel.hip.h, Tally): the recording launch of a band (its
    // blocks' work is the launch-shape record) and launches with SRT_RENDER_COUNT_WORK.  Scene images that live in HBM have none
    // (a correctness fallback): such launches keep the static shape rule and report no work counts.
    // (A recording launch keeps the counts only where the sample-chunk rule could ever read them — launches of the sample counts the
    // rule applies to.  A 32-sample analytic frame, config 2, records its blocks' times with the plain instantiation: the counting
    // one is 0..2 % slower, bench.py's kernel_ms_counting_launch, and that would be the frame's FIRST launch.)
    const bool want_work = (p->flags & SRT_RENDER_COUNT_WORK) != 0;
    const bool rule_applies = (p->sample_count >= 64 || K.n_tris > 0) && p->sample_count >= 32 && K.steps <= 1;
    const bool tally = ((record && rule_applies) || want_work) && in_lds;
    ctx->count_work = want_work;
    ctx->count_work_valid = want_work && tally;
    if (tally && want_work) {
        SRT_HIP(ctx, hipMemsetAsync(ctx->d_work, 0, srt::TALLY_ALL * sizeof(unsigned long long), ctx->stream));
        K.work_counter = ctx->d_work;
    }
    const bool timing = !(p->flags & SRT_RENDER_NO_TIMING);
    if (timing) SRT_HIP(ctx, hipEventRecord(ctx->ev_begin, ctx->stream));
    // instantiation: mesh or not, scene image in LDS or HBM, full tiles / small tiles (multi-sample
    // hand-out) / sample chunks, with or without the loop counts; variants 1 / 3 are a development aid for in-process A/B
    // timing.  All are bit-identical.
    const bool multi = tile_h < srt::TILE_H || (K.steps > 1 && !(K.flags & SRT_RENDER_PREVIEW)) || bgrid;
    auto launch = [&](auto k_lds, auto k_lds_multi, auto k_lds_defer, auto k_hbm, auto k_hbm_multi, auto k_hbm_defer, auto t_lds, auto t_lds_multi, auto t_lds_defer) {
        if (tally && defer) hipLaunchKernelGGL(t_lds_defer, grid, block, lds_bytes, ctx->stream, K);
        else if (tally && multi) hipLaunchKernelGGL(t_lds_multi, grid, block, lds_bytes, ctx->stream, K);
        else if (tally) hipLaunchKernelGGL(t_lds, grid, block, lds_bytes, ctx->stream, K);
        else if (in_lds && defer) hipLaunchKernelGGL(k_lds_defer, grid, block, lds_bytes, ctx->stream, K);
        else if (in_lds && multi) hipLaunchKernelGGL(k_lds_multi, grid, block, lds_bytes, ctx->stream, K);
        else if (in_lds) hipLaunchKernelGGL(k_lds, grid, block, lds_bytes, ctx->stream, K);
        else if (defer) hipLaunchKernelGGL(k_hbm_defer, grid, block, lds_bytes, ctx->stream, K);
        else if (multi) hipLaunchKernelGGL(k_hbm_multi, grid, block, lds_bytes, ctx->stream, K);
        else hipLaunchKernelGGL(k_hbm, grid, block, lds_bytes, ctx->stream, K);
    };
    if (K.n_tris > 0)  // EXTENSION: scenes with triangle meshes use the BVH-enabled instantiation
        launch(srt::pathtrace_kernel<4, true, true, false, false>, srt::pathtrace_kernel<4, true, true, true, false>,
               srt::pathtrace_kernel<4, true, true, false, true>, srt::pathtrace_kernel<4, true, false, false, false>,
               srt::pathtrace_kernel<4, true, false, true, false>, srt::pathtrace_kernel<4, true, false, false, true>,
               srt::pathtrace_kernel<4, true, true, false, false, false, true>, srt::pathtrace_kernel<4, true, true, true, false, false, true>,
               srt::pathtrace_kernel<4, true, true, false, true, false, true>);
#ifdef SRT_DEV  // occupancy variants for A/B timing; never in the shipped library
    else if (use == 1 && in_lds && !multi && !defer && !tally)
        hipLaunchKernelGGL((srt::pathtrace_kernel<4, false>), grid, block, lds_bytes, ctx->stream, K);
    else if (use == 3 && in_lds && !multi && !defer && !tally)
        hipLaunchKernelGGL((srt::pathtrace_kernel<3, false>), grid, block, lds_bytes, ctx->stream, K);
#endif
    else
        // (five waves per SIMD, 96 VGPRs.  Since srt_powf's coefficients come from the LDS constants block — the 64-bit literals had
        // been living in hoisted register pairs — the kernels need 85..95 registers, the multi-sample hand-out of small tiles /
        // progressive blocks included (it stayed at four waves before: 111), and the mesh kernels 119..125: four waves, no spill)
        launch(srt::pathtrace_kernel<5, false, true, false, false>, srt::pathtrace_kernel<5, false, true, true, false>,
               srt::pathtrace_kernel<5, false, true, false, true>, srt::pathtrace_kernel<5, false, false, false, false>,
               srt::pathtrace_kernel<5, false, false, true, false>, srt::pathtrace_kernel<5, false, false, false, true>,
               srt::pathtrace_kernel<5, false, true, false, false, false, true>, srt::pathtrace_kernel<5, false, true, true, false, false, true>,
               srt::pathtrace_kernel<5, false, true, false, true, false, true>);
    if (defer) {
        SRT_HIP(ctx, hipGetLastError());
        hipLaunchKernelGGL(srt::fold_kernel, dim3((unsigned)wg8), dim3(256), 0, ctx->stream, K, (int)wg_x);
    }
    SRT_HIP(ctx, hipGetLastError());
    if (timing) SRT_HIP(ctx, hipEventRecord(ctx->ev_end, ctx->stream));
    ctx->pending_timed = timing;
    if (record) {
        SRT_HIP(ctx, hipMemcpyAsync(ctx->h_wg_cost, ctx->d_wg_cost, REC_WORDS * nwg * 4, hipMemcpyDeviceToHost, ctx->stream));
        SRT_HIP(ctx, hipEventRecord(ctx->ev_cost, ctx->stream));
        ctx->recording = true;
        ctx->rec_gx = grid.x, ctx->rec_gy = grid.y;
        ctx->rec_has_work = tally;  // (a counting launch that also records does keep them)
        ctx->rec_step_w = probe_step_weight(K, ProbeWeights());
        ctx->order_stale = false;
    }
    ctx->launched = true;
    ctx->stats_pending = true;
    ctx->pending_samples = (uint64_t)W * (uint64_t)K.rows * p->sample_count;
    ctx->pending_chunks = (uint32_t)chunks;
    ctx->pending_tile_rows = (uint32_t)tile_h;
    ctx->pending_chunk_samples = defer ? (uint32_t)chunk : 0u;
    ctx->pending_shape_source = shape_source;
    ctx->work_layout[0] = K.nu, ctx->work_layout[1] = K.nc, ctx->work_layout[2] = K.K, ctx->work_layout[3] = K.nb;
    return SRT_OK;
}

#ifdef SRT_DEV
// Development aids (libsrt_pathtrace_dev.so only, not part of the public header): the last pick's raw
// record, and a kernel tuning variant for A/B timing inside one process.  All variants produce identical bits.
int srt_debug_last_pick(srt_context* ctx, int* out4) {
    if (!ctx || !out4) return SRT_ERR_INVALID_ARG;
    memcpy(out4, ctx->last_pick, sizeof ctx->last_pick);
    return SRT_OK;
}

int srt_debug_set_variant(srt_context* ctx, int variant) {
    if (!ctx) return SRT_ERR_INVALID_ARG;
    ctx->variant = variant;
    return SRT_OK;
}
#endif  // SRT_DEV

#ifdef SRT_STATS
int srt_debug_read_stats(unsigned long long* out8) {
    (void)hipDeviceSynchronize();
#if SRT_STATS == 7 || SRT_STATS == 8  // per-wave rows, summed here
    std::vector<unsigned long long> rows(8 * (size_t)srt::SEG_ROWS);
    hipError_t e = hipMemcpyFromSymbol(rows.data(), HIP_SYMBOL(srt::g_seg), rows.size() * sizeof(unsigned long long));
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    for (size_t r = 0; r < (size_t)srt::SEG_ROWS; ++r)
        for (int i = 0; i < 8; ++i) out8[i] += rows[8 * r + (size_t)i];
    std::fill(rows.begin(), rows.end(), 0ull);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(srt::g_seg), rows.data(), rows.size() * sizeof(unsigned long long));
    return e == hipSuccess ? 0 : 3;
#else
    hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(srt::g_stats), 8 * sizeof(unsigned long long));
    unsigned long long z[8] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(srt::g_stats), z, sizeof z);
    return e == hipSuccess ? 0 : 3;
#endif
}
#endif

#if defined(SRT_STATS) && SRT_STATS == 6
int srt_debug_read_wave_log(unsigned long long* out, size_t waves) {
    (void)hipDeviceSynchronize();
    if (waves > (size_t)srt::WAVE_LOG_MAX) waves = srt::WAVE_LOG_MAX;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(srt::g_wave_log), waves * 6 * sizeof(unsigned long long)) == hipSuccess ? 0 : 3;
}
#endif

int srt_pick(srt_context* ctx, int x, int y, int* object_index) {
    if (!ctx || !object_index) return SRT_ERR_INVALID_ARG;
    if (!ctx->scene_set) return fail(ctx, SRT_ERR_STATE, "srt_pick: srt_set_scene has not been called");
    if (!ctx->camera.set) return fail(ctx, SRT_ERR_STATE, "srt_pick: srt_set_camera has not been called");
    SRT_HIP(ctx, hipSetDevice(ctx->device));
    srt_render_params p{};
    p.row_begin = 0;
    p.row_end = ctx->height;
    p.first_sample = 1;
    p.sample_count = 1;
    srt::KernelParams K;
    size_t lds_bytes = 0;
    int use = 0;
    int img = 0;
    if (const int frc = fill_kernel_params(ctx, &p, K, lds_bytes, use, img)) return frc;
    int* d_out = ctx->d_pick;
    // one wave: image (if it fits) + one wave's scratch + one wave's mesh queues
    const size_t image_bytes = (size_t)(K.scene_vec4 > 0 ? K.scene_vec4 : 1) * sizeof(float4);
    if (ctx->pick_in_lds[img])
        hipLaunchKernelGGL(srt::pick_kernel<true>, dim3(1), dim3(64), image_bytes + srt::WAVE_SCRATCH_BYTES + srt::MESH_WAVE_BYTES, ctx->stream, K, x, y, d_out);
    else
        hipLaunchKernelGGL(srt::pick_kernel<false>, dim3(1), dim3(64), (size_t)(srt::WAVE_SCRATCH_BYTES + srt::MESH_WAVE_BYTES), ctx->stream, K, x, y, d_out);
    SRT_HIP(ctx, hipGetLastError());
    int idx[4] = {-1, 0, 0, 0};
    SRT_HIP(ctx, hipMemcpyAsync(idx, d_out, sizeof idx, hipMemcpyDeviceToHost, ctx->stream));
    SRT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *object_index = idx[0];
    memcpy(ctx->last_pick, idx, sizeof idx);
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
        (void)hipGetLastError();  // not an error: keep it from surfacing in a later hipGetLastError()
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
        if (ctx->pending_timed) SRT_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end));
        ctx->stats.kernel_ms = ms;
        ctx->stats.path_samples = ctx->pending_samples;
        ctx->stats.sample_chunks = ctx->pending_chunks;
        ctx->stats.tile_rows = ctx->pending_tile_rows;
        ctx->stats.chunk_samples = ctx->pending_chunk_samples;
        ctx->stats.shape_source = ctx->pending_shape_source;
        ctx->stats.rays = 0;
        memset(&ctx->work_counts, 0, sizeof ctx->work_counts);
        if (ctx->count_work && ctx->count_work_valid) {
            unsigned long long t[srt::TALLY_ALL];
            SRT_HIP(ctx, hipMemcpy(t, ctx->d_work, sizeof t, hipMemcpyDeviceToHost));
            srt_work_counts& wc = ctx->work_counts;
            const uint64_t nu = (uint64_t)ctx->work_layout[0], nc = (uint64_t)ctx->work_layout[1], nb = (uint64_t)ctx->work_layout[3];
            wc.valid = 1;
            wc.waves = t[srt::TALLY_WAVES];
            wc.pool_steps = t[srt::TALLY_STEPS];
            wc.closest_hit_calls = t[srt::TALLY_CALLS];
            // a wave runs every trip of these loops for all of its 64 lanes, whatever they carry: executed tests = trips x 64
            wc.uniform_sphere_tests = t[srt::TALLY_CALLS] * nu * 64u;
            wc.box_tests = t[srt::TALLY_CALLS] * nb * 64u;
            wc.cluster_bound_tests = t[srt::TALLY_BOUND_CALLS] * nc * 64u;
            wc.cluster_sphere_tests = t[srt::TALLY_SPHERE_TESTS] * 64u;
            wc.cluster_items = t[srt::TALLY_ITEMS];
            wc.bvh_child_tests = t[srt::TALLY_NODE_TESTS] * 64u;
            wc.triangle_tests = t[srt::TALLY_LEAF_TRIPS] * 64u;
            wc.bvh_node_rounds = t[srt::TALLY_NODE_ROUNDS];
            wc.mesh_phases = t[srt::TALLY_MESH_PHASES];
        }
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

int srt_get_work_counts(srt_context* ctx, srt_work_counts* out) {
    if (!ctx || !out) return SRT_ERR_INVALID_ARG;
    srt_stats st;
    const int rc = srt_get_stats(ctx, &st);  // waits, and reads the counters of the last render back
    if (rc != SRT_OK) return rc;
    if (!ctx->count_work) return fail(ctx, SRT_ERR_STATE, "srt_get_work_counts: the last srt_render did not ask for SRT_RENDER_COUNT_WORK");
    *out = ctx->work_counts;
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

int srt_read_framebuffer_async(srt_context* ctx, void* dst, size_t pitch_bytes, int row_begin, int row_end, void* copy_stream) {
    if (!ctx || !dst) return SRT_ERR_INVALID_ARG;
    const size_t rowb = (size_t)ctx->width * 4;
    if (row_begin < 0 || row_end > ctx->height || row_begin >= row_end || pitch_bytes < rowb)
        return fail(ctx, SRT_ERR_INVALID_ARG, "srt_read_framebuffer_async: bad rows [%d,%d) or pitch %zu", row_begin, row_end, pitch_bytes);
    SRT_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t cs = copy_stream ? (hipStream_t)copy_stream : ctx->stream;
    if (cs != ctx->stream) {  // the copy starts when the renders enqueued so far have finished, not before
        if (!ctx->ev_read) SRT_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_read, hipEventDisableTiming));
        SRT_HIP(ctx, hipEventRecord(ctx->ev_read, ctx->stream));
        SRT_HIP(ctx, hipStreamWaitEvent(cs, ctx->ev_read, 0));
    }
    SRT_HIP(ctx, hipMemcpy2DAsync(dst, pitch_bytes, (const char*)ctx->d_fb + (size_t)row_begin * rowb, rowb, rowb, (size_t)(row_end - row_begin),
                                  hipMemcpyDeviceToHost, cs));
    return SRT_OK;
}

int srt_read_accumulator(srt_context* ctx, float* dst_rgba) {
    if (!ctx || !dst_rgba) return SRT_ERR_INVALID_ARG;
    SRT_HIP(ctx, hipSetDevice(ctx->device));
    SRT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SRT_HIP(ctx, hipMemcpy(dst_rgba, ctx->d_acc, (size_t)ctx->width * ctx->height * sizeof(float4), hipMemcpyDeviceToHost));
    return SRT_OK;
}

// The balance probe: the PROBE instantiation of pathtrace_kernel runs the real path pool over the WHOLE frame for the frame's
// first PROBE_SAMPLES = 32 samples on ONE of every workgroup's four waves (a quarter of the pixels) — nothing of the frame is read
// or written — and every wave adds what its loops did (srt::TALLY_*: pool steps, groups of exactly tested spheres, BVH rounds,
// triangle trips, phases, child-box tests, tiles with and without untraced pixels) to its 16 x 16 block's counters.  Counts, not
// times: the same on every GPU and in every run.  Cost: the work of 32 x 1/4 = 8 sample-frames on one device — 1.6 % of a 512-spp
// launch of the frame, 0.8 % of a 1024-spp one, a quarter of a 32-spp one — plus a host round trip (the call is synchronous).
constexpr int PROBE_SAMPLES = 32;
static_assert(PROBE_SAMPLES == 32, "ProbeWeights were fitted on probes of 32 samples (shorter pools take 12..37 % more steps per sample, unevenly over a frame): refit them (tools/band_fit.py) when this changes");
static int run_pool_probe(srt_context* ctx, int max_bounces, uint32_t seed, std::vector<uint32_t>& counts, int& bx, int& by) {
    SRT_HIP(ctx, hipSetDevice(ctx->device));
    const int W = ctx->width, H = ctx->height;
    srt_render_params p{};
    p.row_begin = 0, p.row_end = H, p.first_sample = 1, p.sample_count = PROBE_SAMPLES, p.max_bounces = max_bounces, p.seed = seed;
    p.flags = SRT_RENDER_RESET;
    srt::KernelParams K;
    size_t lds_bytes = 0;
    int use = 0, img = 0;
    if (const int frc = fill_kernel_params(ctx, &p, K, lds_bytes, use, img)) return frc;
    K.flags = (K.flags & srt::KF_BOXES_FINITE) | SRT_RENDER_RESET;
    K.tile_h = srt::TILE_H;
    K.accumulator = nullptr, K.framebuffer = nullptr, K.ray_counter = nullptr;  // the probe touches none of them
    bx = (W + srt::WG_W - 1) / srt::WG_W, by = (H + srt::WG_H - 1) / srt::WG_H;
    const size_t n = (size_t)bx * by, words = n * srt::TALLY_N;
    uint32_t* d = nullptr;
    SRT_HIP(ctx, hipMalloc((void**)&d, words * sizeof(uint32_t)));
    hipError_t e = hipMemsetAsync(d, 0, words * sizeof(uint32_t), ctx->stream);
    K.wg_cost = d;
    const dim3 grid((unsigned)bx, (unsigned)by, 1), block(srt::WG_THREADS);
    if (e == hipSuccess) {
        const bool in_lds = ctx->scene_in_lds[img];
        if (K.n_tris > 0 && in_lds) hipLaunchKernelGGL((srt::pathtrace_kernel<4, true, true, false, false, true>), grid, block, lds_bytes, ctx->stream, K);
        else if (K.n_tris > 0) hipLaunchKernelGGL((srt::pathtrace_kernel<4, true, false, false, false, true>), grid, block, lds_bytes, ctx->stream, K);
        else if (in_lds) hipLaunchKernelGGL((srt::pathtrace_kernel<4, false, true, false, false, true>), grid, block, lds_bytes, ctx->stream, K);
        else hipLaunchKernelGGL((srt::pathtrace_kernel<4, false, false, false, false, true>), grid, block, lds_bytes, ctx->stream, K);
        e = hipGetLastError();
    }
    counts.resize(words);
    if (e == hipSuccess) e = hipMemcpyAsync(counts.data(), d, words * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(ctx, SRT_ERR_HIP, "balance probe: %s", hipGetErrorString(e));
    return SRT_OK;
}

int srt_estimate_row_costs(srt_context* ctx, int max_bounces, uint32_t seed, float* row_costs) {
    if (!ctx || !row_costs) return SRT_ERR_INVALID_ARG;
    if (!ctx->scene_set) return fail(ctx, SRT_ERR_STATE, "srt_estimate_row_costs: srt_set_scene has not been called");
    if (!ctx->camera.set) return fail(ctx, SRT_ERR_STATE, "srt_estimate_row_costs: srt_set_camera has not been called");
    if (max_bounces < 0) return fail(ctx, SRT_ERR_INVALID_ARG, "srt_estimate_row_costs: max_bounces must be >= 0");
    const int H = ctx->height;
    std::vector<uint32_t> counts;
    int bx = 0, by = 0;
    const int rc = run_pool_probe(ctx, max_bounces, seed, counts, bx, by);
    if (rc != SRT_OK) return rc;
    srt_render_params p{};
    p.row_begin = 0, p.row_end = H, p.first_sample = 1, p.sample_count = 1, p.max_bounces = max_bounces;
    srt::KernelParams K;
    size_t lds_bytes = 0;
    int use = 0, img = 0;
    if (const int frc = fill_kernel_params(ctx, &p, K, lds_bytes, use, img)) return frc;
    const ProbeWeights w;
    // a block covers WG_H scene rows; its cost is spread evenly over them; memory row m = scene row H - 1 - m
    for (int m = 0; m < H; ++m) row_costs[m] = 0.0f;
    for (int j = 0; j < by; ++j) {
        double sum = 0;
        for (int i = 0; i < bx; ++i) sum += probe_block_cost(&counts[((size_t)j * bx + i) * srt::TALLY_N], K, w);
        const int y0 = j * srt::WG_H, y1 = y0 + srt::WG_H < H ? y0 + srt::WG_H : H;
        for (int y = y0; y < y1; ++y) row_costs[H - 1 - y] = (float)(sum / (double)(y1 - y0));
    }
    return SRT_OK;
}

#ifdef SRT_DEV
// development aid (tools/band_fit.py): the balance probe's raw counts — out[TALLY_N * (bx * by)], block (i, j) covers scene rows
// [16 j, 16 j + 16) — and the scene constants a step's cost depends on: consts = {uniform sphere groups, clusters, boxes, triangles}
int srt_debug_probe_counts(srt_context* ctx, int max_bounces, uint32_t seed, uint32_t* out, int* blocks_x, int* blocks_y, int* consts4) {
    if (!ctx || !out || !blocks_x || !blocks_y || !consts4) return SRT_ERR_INVALID_ARG;
    std::vector<uint32_t> counts;
    const int rc = run_pool_probe(ctx, max_bounces, seed, counts, *blocks_x, *blocks_y);
    if (rc != SRT_OK) return rc;
    memcpy(out, counts.data(), counts.size() * sizeof(uint32_t));
    const srt::SceneLayout& SL = ctx->layout[0];
    consts4[0] = (SL.nu + 3) / 4, consts4[1] = SL.nc, consts4[2] = SL.nb, consts4[3] = ctx->mesh_image.n_tris;
    return SRT_OK;
}
#endif

int srt_selftest_arith(int device, uint32_t seed, uint64_t vectors, uint64_t* mismatches) {
    if (!mismatches || vectors == 0 || vectors > (1ull << 36)) return SRT_ERR_INVALID_ARG;
    int prev = 0;
    if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(device) != hipSuccess) {
        (void)hipGetLastError();
        return SRT_ERR_NO_DEVICE;
    }
    unsigned long long* d = nullptr;
    hipError_t e = hipMalloc((void**)&d, sizeof *d);
    if (e == hipSuccess) e = hipMemset(d, 0, sizeof *d);
    for (uint64_t done = 0; e == hipSuccess && done < vectors; done += 1ull << 28) {  // grids of at most 2^20 blocks
        const uint64_t part = vectors - done < (1ull << 28) ? vectors - done : (1ull << 28);
        hipLaunchKernelGGL(srt::selftest_normalize_kernel, dim3((unsigned)((part + 255) / 256)), dim3(256), 0, 0, seed + (uint32_t)(done >> 28) * 0x85EBCA6Bu, (unsigned long long)done, part, d);
        e = hipGetLastError();
    }
    unsigned long long h = 0;
    if (e == hipSuccess) e = hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost);
    if (d) (void)hipFree(d);
    (void)hipSetDevice(prev);
    if (e != hipSuccess) return SRT_ERR_HIP;
    *mismatches = h;
    return SRT_OK;
}

int srt_gather_band(srt_context* dst, srt_context* src, int row_begin, int row_end) {
    if (!dst || !src) return SRT_ERR_INVALID_ARG;
    if (dst->width != src->width || dst->height != src->height)
        return fail(dst, SRT_ERR_INVALID_ARG, "srt_gather_band: %dx%d into %dx%d", src->width, src->height, dst->width, dst->height);
    if (row_begin < 0 || row_end > src->height || row_begin >= row_end)
        return fail(dst, SRT_ERR_INVALID_ARG, "srt_gather_band: bad rows [%d,%d)", row_begin, row_end);
    if (dst == src) return SRT_OK;
    const size_t rowb = (size_t)src->width * 4, off = (size_t)row_begin * rowb, bytes = (size_t)(row_end - row_begin) * rowb;
    // every HIP call below runs with the SOURCE's device current; whatever happens, the caller gets the destination's
    // device back (a failure must not leave the thread on another GPU)
    hipError_t e = hipSetDevice(src->device);
    const char* what = "hipSetDevice";
    if (e == hipSuccess && src->device != dst->device) {
        // direct xGMI path where the topology offers it: asked once per pair of devices and source context
        // (hipDeviceCanAccessPeer), enabled once; without peer access the runtime stages the copy through the host.
        // NOTE: this branch needs two GPUs and has never executed on this project's one-GPU boxes — it is
        // unverified code; srt_gather_path() says which way a gather went, so the first multi-GPU run tells.
        const unsigned long long bit = dst->device < 64 ? 1ull << dst->device : 0ull;
        if (bit && !(src->peer_asked & bit)) {
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, src->device, dst->device) != hipSuccess) can = 0, (void)hipGetLastError();
            if (can) {
                const hipError_t pe = hipDeviceEnablePeerAccess(dst->device, 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) can = 0;
                if (pe != hipSuccess) (void)hipGetLastError();
            }
            src->peer_asked |= bit;
            if (can) src->peer_direct |= bit;
        }
        const bool direct = bit && (src->peer_direct & bit);
        snprintf(src->gather_path, sizeof src->gather_path, direct ? "device %d -> device %d: hipMemcpyPeerAsync with peer access enabled (direct link)"
                                                                   : "device %d -> device %d: hipMemcpyPeerAsync WITHOUT peer access (staged by the runtime)", src->device, dst->device);
        what = "hipMemcpyPeerAsync";
        e = hipMemcpyPeerAsync((char*)dst->d_fb + off, dst->device, (const char*)src->d_fb + off, src->device, bytes, src->stream);
    } else if (e == hipSuccess) {
        snprintf(src->gather_path, sizeof src->gather_path, "device %d -> device %d: same device, hipMemcpyAsync device to device", src->device, dst->device);
        what = "hipMemcpyAsync";
        e = hipMemcpyAsync((char*)dst->d_fb + off, (const char*)src->d_fb + off, bytes, hipMemcpyDeviceToDevice, src->stream);
    }
    if (e == hipSuccess && !src->ev_gather) what = "hipEventCreate", e = hipEventCreateWithFlags(&src->ev_gather, hipEventDisableTiming);
    if (e == hipSuccess) what = "hipEventRecord", e = hipEventRecord(src->ev_gather, src->stream);
    const hipError_t back = hipSetDevice(dst->device);
    if (e == hipSuccess) what = "hipSetDevice", e = back;
    if (e == hipSuccess) what = "hipStreamWaitEvent", e = hipStreamWaitEvent(dst->stream, src->ev_gather, 0);
    if (e != hipSuccess) return fail(dst, e == hipErrorOutOfMemory ? SRT_ERR_OOM : SRT_ERR_HIP, "srt_gather_band: %s: %s", what, hipGetErrorString(e));
    return SRT_OK;
}

const char* srt_gather_path(const srt_context* src) { return src ? src->gather_path : "(null context)"; }

int srt_write_accumulator(srt_context* ctx, const float* src_rgba) {
    if (!ctx || !src_rgba) return SRT_ERR_INVALID_ARG;
    SRT_HIP(ctx, hipSetDevice(ctx->device));
    SRT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SRT_HIP(ctx, hipMemcpy(ctx->d_acc, src_rgba, (size_t)ctx->width * ctx->height * sizeof(float4), hipMemcpyHostToDevice));
    return SRT_OK;
}

}  // extern "C"
