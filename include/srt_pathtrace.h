/*
 * srt_pathtrace.h — C-ABI of the MI355X path-trace library (libsrt_pathtrace.so).
 *
 * (ABI 6.)  This is the drop-in boundary for ONE hot path of JoshuaLim007/Software-Raytracer:
 * the per-pixel trace / shade / accumulate loop.  The reference has no plugin or
 * FFI interface; the seam this ABI replaces is the tile worker
 *
 *     void renderArea(unsigned index, unsigned minX, unsigned maxX,
 *                     unsigned minY, unsigned maxY, const Transform* camera)
 *                                                  (Raytracer/Raytracer.cpp:223-257)
 *
 * together with the globals that worker reads (Raytracer.cpp:30-35,46-48,55-61) and
 * the two buffers it writes (colorBuffer :60, renderSurface->pixels :50,64).
 * Every entry point below names the reference state it stands for.
 *
 * Conventions
 *   - plain C, POD structs, caller-owned host memory; no pointer is retained after a
 *     call returns (srt_set_scene copies).
 *   - every function returns an srt_status (0 = ok); srt_last_error() gives text.
 *     Nothing throws or aborts across the boundary.
 *   - a handle is single-owner and not re-entrant; independent handles (one per GPU)
 *     may be used from independent threads.
 *   - srt_render is asynchronous on the handle's HIP stream; srt_wait / srt_poll
 *     stand for the reference's threadGroupStatus[] scan (Raytracer.cpp:373-384).
 *   - there is NO CPU fallback: without a HIP device srt_create fails with
 *     SRT_ERR_NO_DEVICE.
 *
 * Framebuffer contract (Raytracer.cpp:64, Common.hpp:189-208):
 *   uint32 per pixel = A<<24 | R<<16 | G<<8 | B  (A is always 0), rows bottom-up:
 *   scene pixel (x, y) lives in memory row H-1-y.  "Memory rows" below always mean
 *   rows of that flipped image; the float4 accumulator is indexed x + y*W with the
 *   scene row y (Raytracer.cpp:67), exactly as colorBuffer is.
 */
#ifndef SRT_PATHTRACE_H
#define SRT_PATHTRACE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRT_ABI_VERSION 6

typedef enum srt_status {
    SRT_OK = 0,
    SRT_ERR_INVALID_ARG = 1,
    SRT_ERR_NO_DEVICE = 2,   /* no HIP device / HIP runtime unusable: there is no CPU fallback */
    SRT_ERR_HIP = 3,         /* a HIP call failed; text in srt_last_error */
    SRT_ERR_STATE = 4,       /* call order violated (e.g. render before set_scene / set_camera) */
    SRT_ERR_OOM = 5
} srt_status;

/* Object kinds the reference loader knows (Raytracer/Scene.hpp:43-55). */
typedef enum srt_object_type {
    SRT_OBJ_NONE = 0,    /* inert Object: occupies a list slot, never hit (Object.hpp:21-23) */
    SRT_OBJ_SPHERE = 1,  /* Sphere (Object.hpp:86-168): uses position + radius          */
    SRT_OBJ_BOX = 2,     /* Box    (Object.hpp:170-234): axis-aligned, half_size = Box::size */
    SRT_OBJ_MESH = 3     /* EXTENSION (not in the reference, which has no triangle primitive):
                            an indexed triangle mesh, see srt_mesh / srt_set_meshes          */
} srt_object_type;

/* Material (Raytracer/Common.hpp:293-319), 11 floats. Colours are the r,g,b of the
 * reference's Color (its a is always 0 on this path). */
typedef struct srt_material {
    float smoothness;
    float specular_amount;
    float base_color[3];
    float emissive_color[3];
    float specular_color[3];
} srt_material;

/* One element of ObjectsToRender (Raytracer.cpp:61), flattened. List order is kept:
 * the closest-hit scan keeps the lower index on equal distances (Raytracer.cpp:132). */
typedef struct srt_object {
    int32_t type;        /* srt_object_type */
    float position[3];   /* transform.position */
    float radius;        /* Sphere::radius (transform.scale is NOT used by the intersector) */
    float half_size[3];  /* Box::size, half extents */
    srt_material material;
    int32_t mesh;        /* SRT_OBJ_MESH: index into the array given to srt_set_meshes; else ignored */
} srt_object;

/* EXTENSION — triangle meshes (BASELINE.json configs 4-5).  The reference defines no triangle
 * arithmetic, so this project does: world vertex = vertex + object position
 * (binary32 add); Moller-Trumbore in binary32 without FMA in a fixed operation order; a hit is
 * valid for 0.01 <= t <= 10000 (the Box bounds, Object.hpp:226); the normal is the unit geometric
 * normal turned against the ray.  Among equal distances the earlier object in ObjectsToRender
 * wins, then the lower triangle index.  Vertices: 3 floats each; indices: 3 uint32 per triangle.
 * Limits (srt_set_scene fails with SRT_ERR_INVALID_ARG beyond them): 20 M triangles per mesh, 16 M in a
 * scene, world coordinates of magnitude <= 1e9, BVH depth <= 72 levels.  Triangles with an index out of
 * range or a non-finite vertex are ignored (they cannot produce a valid hit). */
typedef struct srt_mesh {
    const float* vertices;
    size_t vertex_count;
    const uint32_t* indices;
    size_t triangle_count;
} srt_mesh;

/* Environment globals (Raytracer.cpp:55-59). sun_direction is the already normalised
 * vector (the reference normalises once at start-up, :264). */
typedef struct srt_environment {
    float sun_direction[3];
    float sky_color[3];
    float horizon_color[3];
    float ground_color[3];
    float sun_color[3];
} srt_environment;

/* Camera = the reference's Transform (Common.hpp:281-292) + the FOV global (:31).
 * FOV is an integer number of degrees, vertical, exactly as in Raytracer.cpp:112. */
typedef struct srt_camera {
    float position[3];
    float right[3];
    float up[3];
    float forward[3];
    int32_t fov_degrees;
} srt_camera;

#define SRT_RENDER_RESET 1u       /* first sample of this call overwrites (setFrame, :69-71) */
#define SRT_RENDER_COUNT_RAYS 2u  /* fill srt_stats.rays (costs one atomic per wave)       */
#define SRT_RENDER_PREVIEW 4u     /* SIMPLEDRAW == true: the one-ray preview shader (:147-160)
                                     instead of the path-traced branch (:162-185)           */
#define SRT_RENDER_COUNT_WORK 8u  /* count what the launch's loops execute (srt_get_work_counts); same image, a little slower */
#define SRT_RENDER_NO_TIMING 16u  /* do not bracket this render's kernels with timing events: srt_stats.kernel_ms reads 0 for
                                     it.  Two stream markers less per launch (about 1 % of a 2 ms launch) for a caller that
                                     queues render after render and times them itself, or not at all */

/* One render call = sample_count successive "frames" of the reference's loop over a
 * band of memory rows, all on the device, accumulator kept in registers in between
 * (or, for launches with many samples per pixel, the sample colours kept in device memory and folded
 * in order by a second kernel — INTEGRATION.md §6; the result bits are the same).
 *   sample f (1-based, = ACCUMULATIONFRAMES) keys the RNG and sets the running-mean
 *   weight w = (float)(1.0 / f)               (Raytracer.cpp:66-67).
 *   Clean sequence: first_sample = 1 with SRT_RENDER_RESET, later calls continue with
 *   first_sample = previous + count and no reset. */
typedef struct srt_render_params {
    int32_t row_begin;      /* first memory row (inclusive), 0 = top line of the blitted image */
    int32_t row_end;        /* one past the last memory row */
    uint32_t first_sample;  /* >= 1 */
    uint32_t sample_count;  /* >= 1 */
    int32_t max_bounces;    /* MAXBOUNCES (Raytracer.cpp:32), >= 0 */
    uint32_t seed;          /* RNG seed; the reference names only srand(0) (:263) */
    uint32_t flags;         /* SRT_RENDER_* */
    /* progressive-resolution blocks of renderArea (:233-248): one ray per steps x steps block,
     * its colour replicated into every pixel of the block.  0 or 1 = one ray per pixel.  The
     * reference anchors blocks at the start of each worker's column stripe (:235, :338-340):
     * stripe_width = that stripe width (`div`), 0 = a single stripe (anchors at x = 0). */
    int32_t steps;
    int32_t stripe_width;
    int32_t selected_object; /* list index of selectedObject (:53) for the preview highlight; -1 = none */
} srt_render_params;

typedef struct srt_stats {
    uint64_t rays;          /* GetClosestObject calls (primary counted once per sample; steps > 1: per block, as renderArea traces) */
    uint64_t path_samples;  /* W_band * H_band * sample_count of the last render */
    float kernel_ms;        /* HIP-event time of the last render's kernel(s) on its stream; 0 for a render with SRT_RENDER_NO_TIMING */
    uint32_t sample_chunks; /* 1: one kernel traced and folded every sample; n > 1: the samples of a tile were split
                               over n workgroups (grid layers) that stored the colours, and a second, streaming kernel folded
                               them in order (16 B written + 16 B read per traced sample on top of the 20 B/pixel) */
    /* The launch shape srt_render chose (ABI 6).  It is a function of the request, the grid, the device's CU count and — from the
     * second launch of a band on — the loop counts the band's first launch recorded; never of a clock: the same calls give the
     * same shape in every run. */
    uint32_t tile_rows;     /* rows of a wavefront's pixel tile: 8, or 4 / 2 / 1 for launches of few workgroups */
    uint32_t chunk_samples; /* samples per full-size sample chunk (the last layers may hold half as many); 0: not chunked */
    uint32_t shape_source;  /* 0: the static rule (request and grid only); 1: the band's recorded block work as well */
} srt_stats;

/* What the kernels of ONE render executed (SRT_RENDER_COUNT_WORK), counted by the launch itself in wave-uniform loop counters.
 * "tests" are lane-level and EXECUTED: a wavefront runs every trip of these loops for all 64 lanes whatever they carry, so a
 * trip counts 64 tests; cluster_items counts the useful ones of the exact rounds.  Not part of the reference's interface; it is
 * what bench.py prices its roofline line from. */
typedef struct srt_work_counts {
    uint32_t valid;                /* 0: this launch could not count (scene image too large for LDS); all fields 0 */
    uint32_t reserved;
    uint64_t waves;                /* wavefronts that ran a tile (each stages the scene and traces 64 primary rays) */
    uint64_t pool_steps;           /* wave-level steps of the path pool: one bounce for up to 64 paths */
    uint64_t closest_hit_calls;    /* GetClosestObject at wave level: pool steps + the tiles' primary rays */
    uint64_t uniform_sphere_tests; /* Sphere::Raytrace, spheres every ray is tested against            (Object.hpp:104-141) */
    uint64_t cluster_bound_tests;  /* conservative cluster-bound tests (no counterpart in the reference: the culling filter) */
    uint64_t cluster_sphere_tests; /* Sphere::Raytrace, clustered spheres, in the exact rounds (a round: 64 lanes x 4, 2 or 1 tests) */
    uint64_t cluster_items;        /* (ray, cluster) pairs that passed the bounds: the rounds' useful lanes */
    uint64_t box_tests;            /* Box::Raytrace                                                    (Object.hpp:173-233) */
    uint64_t bvh_child_tests;      /* EXTENSION: quantized child boxes of BVH nodes */
    uint64_t triangle_tests;       /* EXTENSION: Moller-Trumbore */
    uint64_t bvh_node_rounds;      /* EXTENSION: wave-level node rounds of the traversal */
    uint64_t mesh_phases;          /* EXTENSION: wave-level traversal phases */
} srt_work_counts;

typedef struct srt_context srt_context;

/* ---- lifetime ------------------------------------------------------------------ */
int srt_abi_version(void);
/* Number of HIP devices visible; *count = 0 and SRT_ERR_NO_DEVICE when none. */
int srt_device_count(int* count);
/* width/height replace SCREEN_WIDTH / SCREEN_HEIGHT (Raytracer.cpp:26-27) at run time. */
int srt_create(int device, int width, int height, srt_context** out);
int srt_destroy(srt_context* ctx);
/* Text of the last failure on ctx (ctx may be NULL: last srt_create failure of this thread). */
const char* srt_last_error(const srt_context* ctx);

/* ---- state the worker reads ---------------------------------------------------- */
/* Replaces ObjectsToRender (Raytracer.cpp:61,293). Copies; count may be 0 and at most 32767
 * (SRT_ERR_INVALID_ARG beyond).  Scenes of up to ~2000 primitives are staged in LDS; larger ones are
 * read from HBM by a slower instantiation of the same kernel (same results). */
int srt_set_scene(srt_context* ctx, const srt_object* objects, size_t count);
/* EXTENSION: mesh geometry referenced by SRT_OBJ_MESH objects.  Call BEFORE srt_set_scene; copies. */
int srt_set_meshes(srt_context* ctx, const srt_mesh* meshes, size_t count);
/* Replaces SunDirection/SkyColor/HorizonColor/GroundColor/SunColor (:55-59). */
int srt_set_environment(srt_context* ctx, const srt_environment* env);
/* Fills env with the reference's start-up values, computed the way :55-59,264 do. */
int srt_environment_default(srt_environment* env);
/* Replaces the `camera` Transform handed to renderArea (:227,295-297) and FOV (:31). */
int srt_set_camera(srt_context* ctx, const srt_camera* camera);

/* ---- optional device-side plumbing --------------------------------------------- */
/* Launch on this hipStream_t instead of the handle's own stream (NULL = own stream). */
int srt_set_stream(srt_context* ctx, void* hip_stream);
/* Render into caller-provided DEVICE buffers (e.g. a torch tensor's data_ptr) instead of
 * the handle's own: framebuffer = W*H uint32, accumulator = W*H float4. NULL = own.  Takes effect for the
 * calls that follow and does not wait: renders already enqueued keep the buffers they were given (the caller
 * keeps those alive until they finish) — so frame k + 1 can render into a second framebuffer while frame k
 * is copied to the host. */
int srt_bind_output(srt_context* ctx, void* d_framebuffer, void* d_accumulator);
int srt_device_framebuffer(srt_context* ctx, void** d_ptr);
int srt_device_accumulator(srt_context* ctx, void** d_ptr);

/* ---- the hot path -------------------------------------------------------------- */
/* Replaces one release of the workers (threadGroupStatus[i] = false, :592-595) for
 * sample_count frames. Asynchronous. */
int srt_render(srt_context* ctx, const srt_render_params* params);
int srt_wait(srt_context* ctx);             /* block until the last render finished */
int srt_poll(srt_context* ctx, int* done);  /* *done = 1 when finished              */
int srt_get_stats(srt_context* ctx, srt_stats* out);  /* waits for the last render */
/* The loop counts of the last render, which must have had SRT_RENDER_COUNT_WORK set (else SRT_ERR_STATE).  Waits. */
int srt_get_work_counts(srt_context* ctx, srt_work_counts* out);

/* Picking (Raytracer.cpp:525-541): GetClosestObject(camera.position, GetRayDirection(camera, x, y))
 * with y in SCENE rows (the reference flips the mouse y first, :532).  *object_index = list
 * index of the hit object or -1.  Synchronous. */
int srt_pick(srt_context* ctx, int x, int y, int* object_index);

/* ---- buffers the worker writes ------------------------------------------------- */
/* Copies memory rows [row_begin,row_end) into dst (dst points at row_begin's first
 * pixel), pitch_bytes per row (>= 4*W) — the renderSurface->pixels layout (:64). Waits. */
int srt_read_framebuffer(srt_context* ctx, void* dst, size_t pitch_bytes, int row_begin, int row_end);
/* The same copy without the wait: enqueued on `copy_stream` (a hipStream_t of the caller's; NULL = the handle's launch stream)
 * behind every render enqueued on the handle so far, then the call returns.  dst must stay valid — and should be pinned host memory,
 * or the copy is not asynchronous — until the caller has synchronised copy_stream.  With two device framebuffers bound in turn
 * (srt_bind_output, which does not wait) frame k travels to the host while frame k + 1 renders: the blit of Raytracer.cpp:549-556
 * off the critical path (bench.py `readback`: 0.155 ms per 1080p frame hidden, +1.5 % per step instead of +6.9 %). */
int srt_read_framebuffer_async(srt_context* ctx, void* dst, size_t pitch_bytes, int row_begin, int row_end, void* copy_stream);
/* colorBuffer (:60): W*H float4 (r,g,b,a), index x + y*W, scene rows. Wait + copy. */
int srt_read_accumulator(srt_context* ctx, float* dst_rgba);
int srt_write_accumulator(srt_context* ctx, const float* src_rgba);

/* ---- multi-GPU row stripes in ONE process -----------------------------
 * The reference splits a frame over 16 worker threads that write disjoint columns of one
 * surface (Raytracer.cpp:330-342); across GPUs the split is disjoint bands of MEMORY rows,
 * one context per GPU, and this call is the gather: it copies memory rows
 * [row_begin,row_end) of `src`'s framebuffer into the same rows of `dst`'s framebuffer,
 * device to device (a peer copy over xGMI when the two contexts live on different GPUs),
 * enqueued on src's stream behind its render; dst's stream is made to wait for it, so a
 * following srt_read_framebuffer(dst) sees the band.  Both contexts must have the same
 * width and height.  (Across PROCESSES the same gather is one RCCL collective:
 * software-raytracer_amd/stripes.py, bench.py --gpus N.) */
int srt_gather_band(srt_context* dst, srt_context* src, int row_begin, int row_end);
/* Which way `src`'s last srt_gather_band went, as text: the same device, a peer copy with peer access enabled (hipDeviceCanAccessPeer
 * is asked once per pair of devices), or a copy the runtime had to stage.  The cross-device ways have never run on this project's
 * one-GPU boxes; this is how the first multi-GPU run reports what it did.  The pointer stays valid until src is destroyed. */
const char* srt_gather_path(const srt_context* src);
/* Relative cost of every MEMORY row of the frame for the current scene and camera (row_costs[height], arbitrary
 * units), from a device-side probe: the path-trace kernel's own path pool, run over a quarter of the pixels for the
 * frame's first 32 samples, COUNTING what its loops do (pool steps, exactly tested sphere groups, BVH rounds, per-tile
 * work) instead of writing the frame; the counts are weighed into a cost.  Nothing of the frame is read or written.
 * Costs the work of 8 sample-frames on this device (32 samples on a quarter of the pixels: 1.6 % of a 512-spp launch of the
 * frame, a quarter of a 32-spp one) and a host round trip.  Counts, not times: deterministic — every process of a multi-GPU job computes the
 * same numbers, so the ranks can agree on cost-balanced row bands without talking to each other.  The reference's
 * static split into 16 equal column stripes (Raytracer.cpp:330-342) leaves its workers idle behind the slowest one;
 * equal ROW bands are worse (sky rows cost a tenth of floor rows).  Synchronous. */
int srt_estimate_row_costs(srt_context* ctx, int max_bounces, uint32_t seed, float* row_costs);

/* ---- diagnostics ----------------------------------------------------------------- */
/* Self-test of the kernel's shortened arithmetic (csrc/srt_kernel.hip.h: float3::Normalized, the box slab slopes and the
 * accumulation weight 1 / frame without the rescaling and fix-up steps of the library sqrt / divide where those are
 * identities): `vectors` pseudo-random vectors — all magnitudes, zero and denormal components, infinities, NaNs — through
 * the short and the library path on the device, and 1 / frame for every frame up to 2^24 (vectors >= 2^24 covers them
 * all); the random direction's normalization without its window test on triples of draws (the 64 extreme combinations, then random
 * ones); the sphere test's short square root against sqrtf for the bit pattern of every vector index below 2^31 that lies in
 * its window (vectors >= 2^31 covers every float there); *mismatches = how many differ in any bit (must be 0).  Not part of the
 * reference's interface. */
int srt_selftest_arith(int device, uint32_t seed, uint64_t vectors, uint64_t* mismatches);

#ifdef __cplusplus
}
#endif
#endif /* SRT_PATHTRACE_H */
