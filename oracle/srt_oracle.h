/*
 * srt_oracle.h — CPU oracle for the path-trace hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product (libsrt_pathtrace.so, the C++ host) never links, loads or
 * calls it, and has no CPU fallback.
 *
 * PARITY UNPINNED: the reference ships no tests, golden images or known-answer
 * vectors, and it cannot be built in this image (needs SDL2, <Windows.h> and MSVC
 * dialect).  This oracle is therefore a line-by-line restatement of
 * the reference's arithmetic, pinned only by hand-derived analytic vectors and by
 * the scene generator recorded in Raytracer.cpp:299-325 — not by reference output.
 * (The one part of the reference that does build here, its vendored JSON library, pins
 * the scene wire format instead: ref_json_harness.cpp, tests/test_json_vs_reference.py.)
 */
#ifndef SRT_ORACLE_H
#define SRT_ORACLE_H

#include "srt_pathtrace.h" /* POD structs only (srt_object, srt_camera, ...) */

#ifdef __cplusplus
extern "C" {
#endif

/* which powf GetEnvironmentColor uses */
#define SRT_ORACLE_POW_LIBM 0   /* the host libm's powf — what the reference source says */
#define SRT_ORACLE_POW_SHARED 1 /* srt_powf (include/srt_defs.h) — bit-comparable with the GPU */

/* how the image is split over CPU threads (results are identical; timing differs) */
#define SRT_ORACLE_SPLIT_ROWS 0    /* rows dealt round-robin to the threads (balanced) */
#define SRT_ORACLE_SPLIT_REF_COLS 1 /* reference-faithful: `threads` column stripes of
                                       ceil(W/threads)+1, x-outer/y-inner walk
                                       (Raytracer.cpp:235-237,330-342) */

typedef struct srt_oracle_job {
    const srt_object* objects;
    size_t object_count;
    const srt_mesh* meshes;   /* EXTENSION (srt_pathtrace.h): geometry of SRT_OBJ_MESH objects, may be NULL */
    size_t mesh_count;
    const srt_environment* env;
    const srt_camera* camera;
    int32_t width, height;
    srt_render_params params; /* same meaning as for srt_render */
    float* accumulator;       /* W*H*4 floats, index (x + y*W)*4, scene rows; read unless RESET */
    uint32_t* framebuffer;    /* W*H uint32, bottom-up (memory row H-1-y); may be NULL */
    int32_t pow_mode;         /* SRT_ORACLE_POW_* */
    int32_t threads;          /* >= 1 */
    int32_t split;            /* SRT_ORACLE_SPLIT_* */
    uint64_t rays_out;        /* GetClosestObject calls, same counting rule as srt_stats.rays */
    int32_t col_begin, col_end; /* test aid: render only columns [col_begin, col_end) of the band; 0,0 = all */
} srt_oracle_job;

/* Render params.sample_count frames of the reference loop for the memory-row band. */
int srt_oracle_render(srt_oracle_job* job);

/* --- per-function probes (unit vectors for tests) -------------------------------- */
/* GetRayDirection (Raytracer.cpp:106-122) */
void srt_oracle_ray_direction(const srt_camera* cam, int32_t width, int32_t height, int32_t px,
                              int32_t py, float out_dir[3]);
/* Object::Raytrace for one object (Object.hpp:153-167 / 224-233): returns valid flag */
int srt_oracle_intersect(const srt_object* obj, const float origin[3], const float dir[3],
                         float out_normal[3], float out_point[3], float* out_distance);
/* EXTENSION: this project's triangle test for one triangle given by its three
 * WORLD vertices; returns valid flag */
int srt_oracle_triangle(const float v0[3], const float v1[3], const float v2[3], const float origin[3],
                        const float dir[3], float out_normal[3], float out_point[3], float* out_distance);
/* GetClosestObject (Raytracer.cpp:123-140): returns object index or -1 */
int srt_oracle_closest(const srt_object* objects, size_t count, const float origin[3],
                       const float dir[3], float out_normal[3], float out_point[3],
                       float* out_distance);
/* GetEnvironmentColor (Raytracer.cpp:77-89) */
void srt_oracle_environment(const srt_environment* env, const float dir[3], int32_t pow_mode,
                            float out_rgb[3]);
/* RaytraceScene, path-traced branch (Raytracer.cpp:141-146,162-185,212) for one sample */
void srt_oracle_trace_sample(const srt_object* objects, size_t count, const srt_environment* env,
                             const srt_camera* cam, int32_t width, int32_t height, int32_t px,
                             int32_t py, uint32_t sample, int32_t max_bounces, uint32_t seed,
                             int32_t pow_mode, float out_rgba[4], uint32_t* rays,
                             uint32_t* draws);
/* tone-map + pack of SetScreenPixel (Raytracer.cpp:73-75, Common.hpp:189-208) */
uint32_t srt_oracle_tonemap_pack(const float rgba[4]);
/* srt_powf as compiled into the oracle (to compare with the device build) */
float srt_oracle_powf_shared(float x, float y);
/* the reference's start-up environment (Raytracer.cpp:55-59,264) */
void srt_oracle_environment_default(srt_environment* env);

#ifdef __cplusplus
}
#endif
#endif
