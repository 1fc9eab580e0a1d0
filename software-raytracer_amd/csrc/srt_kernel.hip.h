// srt_kernel.hip.h — gfx950 (CDNA4) path-trace kernel.
//
// What it computes is, bit for bit, the reference's per-pixel loop
//   GetRayDirection -> RaytraceScene -> SetScreenPixel   (Raytracer/Raytracer.cpp:63-213)
// with Object.hpp's Sphere / Box intersectors and Common.hpp's float3 / Color value
// semantics, for `sample_count` successive frames in one launch (plus the preview shader,
// progressive blocks and picking of Raytracer.cpp:147-160, 233-248, 525-541).  HOW it computes
// it is MI355X-first and shares nothing with the reference's structure:
//
//   * one work-item = one pixel for set-up and output; wavefront (64 lanes) = 8x8 pixel tile;
//     workgroup = 4 waves = 16x16 pixels (dealt to the waves pixel by pixel); 85..95 VGPRs -> 5 waves per SIMD (mesh: 120, 4).
//   * the flattened scene image (srt_scene_image.h) is staged ONCE per workgroup into LDS.
//   * primary rays do not depend on the sample (no jitter, Raytracer.cpp:109-110): the primary
//     hit is found once per pixel; pixels whose colour is sample-invariant finish immediately.
//   * wave-level path pool: traced pixels are compacted into slots (ballot + mbcnt); free lanes
//     pull (slot, sample) tasks, finished sample colours go through an LDS ring and are folded
//     into the order-dependent running mean (Raytracer.cpp:65-71) strictly in sample order by the
//     slot's owner lane; one 16-byte accumulator store + one 4-byte ARGB store per pixel.
//   * closest_hit runs in wave-uniform control flow (idle lanes help):
//       - few large "uniform" spheres: broadcast ds_read_b128, exact arithmetic, sqrt only under
//         a ballot-uniform branch;
//       - small spheres in clusters of K: per-lane conservative bound test (FMA allowed, proven
//         conservative), then all (ray, cluster) pairs of the wave are compacted into an LDS work
//         list, lanes pull items, fetch the ray with __shfl, run the EXACT arithmetic and merge
//         through a 64-bit LDS atomicMin on (ordered distance, list index, primitive);
//       - boxes: exact iq slab test, per-ray part hoisted;
//       - EXTENSION, triangle meshes: wave-cooperative traversal of an 8-wide quantized BVH (LDS
//         LIFO of (ray, node) items, 8 lanes per item when few wait), srt_mesh_bvh.h.
//   * counter-based RNG keyed (seed, absolute pixel, sample, draw#): include/srt_defs.h.
//   * launches with few rows and many samples per pixel (a stripe of a multi-GPU frame) would have too few,
//     too long workgroups: from 16 spp the tile of a wave shrinks (P.tile_h rows, MULTI hand-out), from 64
//     spp the samples of a tile are split over several workgroups (DEFER) that store the colours as rows of
//     a sample buffer, and fold_kernel folds them in sample order (HBM-bound, 16 B per traced sample); the last chunks of a
//     launch are half as long (the launch's tail).
//   * the PROBE instantiation runs the same pool without touching the frame and counts its loop trips: the cost estimate behind
//     the multi-GPU row split (srt_estimate_row_costs).
//
// Bit-exactness rules (the file is compiled with -ffp-contract=off; hipcc's default
// correctly-rounded fp32 divide/sqrt stays on; fp32 denormals are not flushed):
//   every expression of the reference keeps its association; Color results go through clamp0()
//   like Color's constructor (Common.hpp:253-262); comparisons keep their NaN behaviour
//   (a > b ? a : b — fmaxf / fminf only in the box slab test, and only when the wave has seen that no operand can be a NaN).
//   Explicit FMAs appear only (a) in conservative filters, where
//   any rounding is covered by the inflation proofs, (b) in srt_powf / rand_unit, where host and
//   device execute the same fused operations or the result is verified exhaustively, (c) in the short forms of the library's
//   own sqrt / divide expansions (normalized(), div_window(): the library's operations minus its rescaling and fix-up steps
//   where those are identities; srt_selftest_arith compares the two on the device).
//   Winner updates are written as branch-free selects (see the note in closest_hit).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "srt_defs.h"

namespace srt {

// ---- device scene image (identical bytes in HBM and in LDS): see srt_scene_image.h ----
// [0,nu4) uniform spheres | [nu4,nsT) clustered spheres (nc clusters of K) | nc cluster bounds |
// 2*nb box rows | 3 material rows per primitive id (spheres: index in [0,nsT); boxes: nsT + j).
struct KernelParams {
    float cam_pos[3];
    float right_rd[3];  // cameraTransform.right * rd          (Raytracer.cpp:114,116)
    float up_ld[3];     // cameraTransform.up * ld             (:115,117)
    float fwd_clip[3];  // cameraTransform.forward * clipDist  (:113)
    // (the environment of Raytracer.cpp:55-59 travels in the scene image's constants block, srt_scene_image.h)
    int32_t width, height;
    int32_t y0, rows;  // scene rows [y0, y0+rows)
    uint32_t first_sample, sample_count;
    int32_t max_bounces;
    uint32_t seed, flags;
    int32_t steps, stripe_width, selected;  // progressive blocks (:233-248), selectedObject (:53)
    int32_t bgrid_w, bgrid_h;               // flags & KF_BLOCK_GRID: the launch's lanes are BLOCKS, this many columns / rows of them
    // scene image layout (srt_scene_image.h)
    int32_t nu4, nc, K, nsT, nb;
    int32_t nu;  // uniform spheres that are not padding
    int32_t off_bounds, off_box, off_mat;
    int32_t scene_vec4;  // number of float4 in the scene image
    const float4* scene;
    // EXTENSION: triangle meshes (srt_mesh_bvh.h); n_tris == 0 -> none
    const float4* bvh_nodes;
    const float4* bvh_tris;
    const int32_t* bvh_gidpos;  // global triangle id -> position in bvh_tris
    int32_t n_tris;
    float mesh_center[3];
    float mesh_half[3];  // root box half extents
    float mesh_r1;       // their sum + |centre|_1
    float mesh_bs_radius;  // radius of a sphere around mesh_center that contains every triangle
    int32_t mesh_defer;  // path pool: fewest rays that start a mesh phase (closest_hit)
    int32_t mesh_wait;   // path pool: ... unless some ray has been waiting for this many steps
    int32_t tile_h;      // rows of a wave's pixel tile: 8, 4, 2 or 1 (pathtrace_kernel)
    // sample-chunked launches (DEFER instantiation): workgroup z traces samples [z*chunk, (z+1)*chunk) of
    // its tiles and stores the colours in sample order; fold_kernel then folds them into the running mean
    int32_t chunk;           // samples per workgroup, 0 = everything in one workgroup (no sample buffer)
    int32_t chunk_full;      // grid layers z < chunk_full trace `chunk` samples each, the layers behind them half as many (the launch's tail)
    float4* sample_rows;     // [tile][sample][64 slots] sample colours
    unsigned long long* tile_masks;  // [tile] which pixels of the tile are traced (slot k = k-th set bit)
    // chained chunks (round 4): [tile] how many of the tile's chunks have been folded IN ORDER straight into the accumulator by
    // the workgroups that traced them (zeroed by the host before the launch; NULL: every chunk goes through the sample buffer).
    // Chunk z of a tile folds its own samples if it finds z here when it starts — chunk z - 1 has stored the running mean — and
    // raises the count when it is done; otherwise it stores its colours as rows of the sample buffer, like every chunk behind it,
    // and fold_kernel picks the tile up from the count.  No workgroup ever waits for another.
    uint32_t* tile_chain;
    int32_t chunk_layers;    // grid layers of the launch (fold_kernel: where a tile's chain may stand)
    // cost-ordered dispatch: workgroup i of the launch works on tile block wg_order[i] (NULL: i); every
    // wave adds its run time (10 ns ticks of the constant clock) to wg_cost[block] (NULL: not recorded) for the order of the next launch
    const uint32_t* wg_order;
    uint32_t* wg_cost;       // [wg_blocks] the blocks' wave time, then (TALLY instantiations) [wg_blocks][4 waves][TALLY_N] the waves' loop counts
    uint32_t wg_blocks;
    unsigned long long* work_counter;  // SRT_RENDER_COUNT_WORK: [TALLY_ALL] totals of the launch (NULL: not counted)
    float4* accumulator;
    uint32_t* framebuffer;
    unsigned long long* ray_counter;
};

// internal KernelParams.flags bit (srt_render sets it): progressive-block launch whose lanes stand for steps x steps
// blocks instead of pixels — a lane traces its block's ray and writes all of the block's pixels
constexpr uint32_t KF_BLOCK_GRID = 0x10u;
constexpr uint32_t KF_BOXES_FINITE = 0x20u;  // (srt_set_scene found every box centre and half size below 1e29 in size: closest_hit's NaN-free slab test)
constexpr uint32_t SRT_MESH_ORDER_W = 6u;  // block_cost_kernel: ordering cost of a ray that ends on a mesh, in analytic rays (as the balance cost)
constexpr int TILE_W = 8, TILE_H = 8;       // per wavefront
constexpr int WG_TILES_X = 2, WG_TILES_Y = 2;  // waves per workgroup
constexpr int WG_THREADS = 64 * WG_TILES_X * WG_TILES_Y;
constexpr int WG_W = TILE_W * WG_TILES_X, WG_H = TILE_H * WG_TILES_Y;
// per-wave LDS scratch behind the scene image: work list of (ray lane, cluster) items + one
// 64-bit result slot per lane (balanced phase 2 of closest_hit)
constexpr int WORK_MAX = 512;
// ring entries per slot when all 64 pixels of the tile are traced (a power of two).  Round 3: 2 (4 before).  The analytic kernel
// runs FIVE waves per SIMD (96 VGPRs) and the mesh kernel FOUR (128): that many workgroups' LDS fit into a CU's 160 KiB only
// with the shorter ring.  With unchanged occupancy the shorter ring costs 3..7 % (a slot whose sample runs long can have only one
// more in flight); the extra wave wins that back and more: Scene_indirect -4.4 %, Scene3 -4.5 %, Scene1 -1.5 %, config 4 -5 %.
constexpr int RING_DEPTH = 2;
// per-wave: 64 result slots (8 B) | work list (2 B) | 64 pixel records (48 B) | ring (16 B)
constexpr int WAVE_SCRATCH_BYTES = 64 * 8 + WORK_MAX * 2 + 64 * 12 * 4 + 64 * RING_DEPTH * 16;
constexpr int WG_SCRATCH_BYTES = WAVE_SCRATCH_BYTES * WG_TILES_X * WG_TILES_Y;
// extra per-wave LDS of the mesh kernel: node LIFO + leaf queue of the cooperative BVH traversal
// (sized so that three workgroups of the mesh kernel still fit into a CU's 160 KiB next to the Scene1-sized image)
constexpr int MESH_Q = 512;  // entries of the traversal buffer: the node LIFO grows up from its bottom, the leaf queue down from its top
constexpr float MESH_T_MIN_CULL = 0.0099f;  // just below the smallest valid triangle distance, (float)0.01
#ifdef SRT_STATS  // development build only (make STATS=1|2): traversal counters read by srt_debug_read_stats
__device__ unsigned long long g_stats[8];
#define SRT_STAT(i, v) do { if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0) atomicAdd(&g_stats[i], (unsigned long long)(v)); } while (0)
#else
#define SRT_STAT(i, v) do { } while (0)
#endif
#if defined(SRT_STATS) && SRT_STATS == 3  // make STATS=3: wave-cycles per section of the pool loop (tests/section_profile.py)
struct Prof {
    long long last, acc[8];
};
#define SRT_PROF_PARAM , Prof& prof
#define SRT_PROF_ARG , prof
#define SRT_TICK(i) do { const long long now_ = (long long)__builtin_readcyclecounter(); prof.acc[i] += now_ - prof.last; prof.last = now_; } while (0)
#else
#define SRT_PROF_PARAM
#define SRT_PROF_ARG
#define SRT_TICK(i) do { } while (0)
#endif
#if defined(SRT_STATS) && (SRT_STATS == 7 || SRT_STATS == 8)  // make dev STATS=7 / 8: wave-cycles per segment of a BVH node round / leaf round (tests/mesh_stats.py, SRT_STATS_MODE=7 / 8)
constexpr int SEG_ROWS = 1 << 16;  // a row of eight sums per wave (by block and wave, modulo): no contended atomics inside the timed code
__device__ unsigned long long g_seg[8 * SEG_ROWS];
// everything issued so far has completed (loads, LDS) and `dep` has been computed when the clock is read
#define SRT_SEG_(i, dep) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" :: "v"(dep) : "memory"); const long long now_ = (long long)__builtin_readcyclecounter(); seg_acc[i] += now_ - seg_t; seg_t = now_; } while (0)
#if SRT_STATS == 7
#define SRT_SEG(i, dep) SRT_SEG_(i, dep)
#define SRT_SEGL(i, dep) do { } while (0)
#else
#define SRT_SEG(i, dep) do { } while (0)
#define SRT_SEGL(i, dep) SRT_SEG_(i, dep)
#endif
#else
#define SRT_SEG(i, dep) do { } while (0)
#define SRT_SEGL(i, dep) do { } while (0)
#endif
#if defined(SRT_STATS) && SRT_STATS == 6  // make dev STATS=6: per-wave clock records for tools/timer_probe.py (the block-cost timer anomaly)
constexpr int WAVE_LOG_MAX = 1 << 17;
__device__ unsigned long long g_wave_log[6 * WAVE_LOG_MAX];
#endif
constexpr int MESH_WAVE_BYTES = MESH_Q * 4;
constexpr int WG_MESH_SCRATCH_BYTES = MESH_WAVE_BYTES * WG_TILES_X * WG_TILES_Y;

// inclusive prefix sum over the 64 lanes of a wave (all lanes active): four shifts inside the rows of 16, then the row totals
// are passed on with the row-broadcast controls — six v_add_u32_dpp
__device__ __forceinline__ unsigned wave_inclusive_scan(unsigned v) {
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ float clamp0(float v) { return v < 0 ? 0.0f : v; }  // Common.hpp:254-257
// Where a clamp0 of the reference is NOT spelled out below ("NN"): clamp0 changes strictly negative values only (-0 and NaN pass),
// and a sum or product of values that are never strictly negative is never strictly negative either.  The material colours and
// the environment colours arrive clamped in the scene image (srt_scene_image.h), spec is 0 or 1, the weights of the running mean
// lie in [0, 1]; so L, T, sample colours and every accumulator after its first fold are NN and the reference's clamp of such a
// sum or product is the identity — two instructions and a VCC hazard each, about a third of the shading code before.

// (float)rand() / RAND_MAX (Raytracer.cpp:93-95,165,182) for r in [0, 32767]: the correctly rounded
// quotient via one Newton correction (q0 = r*y, e = fma(-q0, b, r), q = fma(e, y, q0)) instead of
// the generic IEEE divide expansion.  Equal to r / 32767.0f for ALL 32768 inputs — checked
// exhaustively on the CPU (tests/test_defs.py) — so the bits are the reference's.
__device__ __forceinline__ float rand_unit(uint32_t r) {
    const float b = 32767.0f, y = 0x1.0002p-15f;  // y = RN(1/b)
    const float a = (float)r;
    const float q0 = a * y;
    return __builtin_fmaf(__builtin_fmaf(-q0, b, a), y, q0);
}

// n / d, correctly rounded, for operands INSIDE the window in which the library division's rescaling (v_div_scale) and fix-up
// (v_div_fixup) are identities: d normal with 2^-60 <= |d| <= 2^60, and n zero, NaN or within 2^60 of d either way.  The call
// sites guarantee the window by construction (see there); normalized() below, whose operands are arbitrary, tests it.  What
// is left of the expansion: the reciprocal, its two-step refinement, the two-step refinement of the quotient — the same
// operations on the same values as the library path, seven instructions instead of eleven.
__device__ __forceinline__ float div_window(float n, float d) {
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float r1 = __builtin_fmaf(__builtin_fmaf(-d, r0, 1.0f), r0, r0);
    const float q0 = n * r1;
    const float q1 = __builtin_fmaf(__builtin_fmaf(-d, q0, n), r1, q0);
    return __builtin_fmaf(__builtin_fmaf(-d, q1, n), r1, q1);
}

// sqrtf(x), correctly rounded, for x INSIDE the window in which the library expansion's rescaling and class test are identities:
// x = +0, x NaN, or 2^-96 <= x < inf (the library scales arguments below 2^-96 by 2^32 and the root by 2^-16, and hands +-0 and
// +inf back through v_cmp_class).  What is left: v_sqrt_f32 and the two one-ulp corrections — each neighbour's residual with one
// FMA — ten instructions instead of seventeen and five hazard no-ops.  For x = +0 the corrections keep 0 (the lower neighbour of 0
// is a NaN bit pattern, its residual a NaN, the compare false; the upper neighbour's residual is 0, not above 0); a NaN goes
// through unchanged in both forms.  The one call site is the sphere test's  tc - sqrt(r*r - d2)  for a candidate (d2 <= r*r): if
// r*r >= 2^-72 the difference of the two floats is 0 or at least 2^-96 (b <= a/2: a - b >= a/2; else Sterbenz: exact, a
// multiple of ulp(b) >= 2^-96) — srt_set_scene checks every sphere's r*r against [2^-72, FLT_MAX] and sends a scene with a
// sphere outside it to the instantiations that read the scene from memory, which keep the library sqrtf (SCENE_LDS == false).
// srt_selftest_arith compares the two forms on the device.
__device__ __forceinline__ float sqrt_window(float x) {
    float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
    const float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
    s = (0.0f >= rm) ? sm : s;
    s = (0.0f < rp) ? sp : s;
    return s;
}

struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
// float3::Normalized (Common.hpp:159-162): sqrtf and three IEEE divisions by the same length.
// The compiler expands sqrtf into v_sqrt_f32 + the two one-ulp corrections, wrapped in a rescaling for arguments below 2^-96 and a
// class test for 0 / inf / NaN, and EACH division into v_div_scale x 2, v_rcp_f32, the two-step refinement of the reciprocal, the
// two-step refinement of the quotient (v_div_fmas) and v_div_fixup: 20 + 3 x 11 instructions, four of them quarter-rate.  With
// the squared length in [2^-96, 2^96] and every component at least 2^-60 of the length (so: not zero, quotient normal) the
// rescalings and fix-ups are identities — v_div_scale hands its operands back, v_div_fmas is a plain FMA — and what remains is the
// chain below, with the reciprocal's refinement done once for all three quotients: the same operations on the same values, the
// same bits.  One wave-uniform test decides; any lane outside the window (an axis-parallel ray has zero components) sends the
// wave down the library path.  tests/test_gpu_fast_arith.py compares the two paths bit for bit on 2^28 random and edge vectors.
__device__ __forceinline__ V3 normalized_ieee(V3 a) {
    float length = sqrtf((a.x * a.x + a.y * a.y) + a.z * a.z);
    return v3(a.x / length, a.y / length, a.z / length);
}
__device__ __forceinline__ V3 normalized(V3 a) {
    const float x = (a.x * a.x + a.y * a.y) + a.z * a.z;
    const float s0 = __builtin_amdgcn_sqrtf(x);
    const float sm = __uint_as_float(__float_as_uint(s0) - 1u), sp = __uint_as_float(__float_as_uint(s0) + 1u);
    float len = (0.0f >= __builtin_fmaf(-sm, s0, x)) ? sm : s0;  // the expansion's corrections: one ulp down, one ulp up
    len = (0.0f < __builtin_fmaf(-sp, s0, x)) ? sp : len;
    const float tiny = len * 0x1p-60f;
    const bool ok = x >= 0x1p-96f && x <= 0x1p96f && fabsf(a.x) >= tiny && fabsf(a.y) >= tiny && fabsf(a.z) >= tiny;  // (NaN: false)
    if (__builtin_amdgcn_ballot_w64(!ok) != 0ull) return normalized_ieee(a);
    const float r0 = __builtin_amdgcn_rcpf(len);
    const float r1 = __builtin_fmaf(__builtin_fmaf(-len, r0, 1.0f), r0, r0);
    auto quot = [&](float n) {
        const float q0 = n * r1;
        const float q1 = __builtin_fmaf(__builtin_fmaf(-len, q0, n), r1, q0);
        return __builtin_fmaf(__builtin_fmaf(-len, q1, n), r1, q1);
    };
    return v3(quot(a.x), quot(a.y), quot(a.z));
}
// normalized() for a vector that is inside the window BY CONSTRUCTION: no test, no library path.  The one caller is the random
// direction of GetRandomDirection (Raytracer.cpp:90-97): its components are (r / 32767 - 0.5) * 2 for an integer r in [0, 32767] —
// never zero (r / 32767 = 0.5 has no integer solution; the nearest, r = 16383 and 16384, give -+3.05e-5 after an exact
// subtraction and an exact doubling) and at most 1 in size: squared length in [2.8e-9, 3], every component at least 1.7e-5 of the
// length.  The same chain as normalized()'s short path, i.e. the library's operations on the same values.
__device__ __forceinline__ V3 normalized_in_window(V3 a) {
    const float x = (a.x * a.x + a.y * a.y) + a.z * a.z;
    const float len = sqrt_window(x);
    const float r0 = __builtin_amdgcn_rcpf(len);
    const float r1 = __builtin_fmaf(__builtin_fmaf(-len, r0, 1.0f), r0, r0);
    auto quot = [&](float n) {
        const float q0 = n * r1;
        const float q1 = __builtin_fmaf(__builtin_fmaf(-len, q0, n), r1, q0);
        return __builtin_fmaf(__builtin_fmaf(-len, q1, n), r1, q1);
    };
    return v3(quot(a.x), quot(a.y), quot(a.z));
}
__device__ __forceinline__ float dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }


constexpr int CONST_ROWS = 46, CONST_ENV_ROW = 32, CONST_COEF_ROW = 36;  // srt_scene_image.h: SRT_CONST_ROWS / SRT_CONST_ENV_ROW / SRT_CONST_COEF_ROW
struct Lds {
    const float4* c;  // the image's constants block: srt_powf's table (32 rows), the environment (4 rows)
    const float4* v;  // the primitives behind it (all offsets below count from here)
    int nu4, nu, nc, K, nsT, nb, off_bounds, off_box, off_mat;
    unsigned long long* res;  // this wave's 64 result slots
    unsigned short* work;     // this wave's work list
    float* pix;               // this wave's 64 pixel records (12 floats each)
    float4* ring;             // this wave's sample-colour ring
    unsigned* meshq;          // mesh kernel only: [MESH_Q] node LIFO from the bottom, leaf queue from the top
    __device__ __forceinline__ float4 sphere(int p) const { return v[p]; }
    __device__ __forceinline__ float4 bound(int k) const { return v[off_bounds + k]; }
    __device__ __forceinline__ float4 box_c(int j) const { return v[off_box + 2 * j]; }
    __device__ __forceinline__ float4 box_h(int j) const { return v[off_box + 2 * j + 1]; }
    // (the compiler makes one v_mad_u64_u32 of `base + 48 * p`; spelled with 24-bit multiplies it was 0..1 % slower: profiles/r04/ab_notes.txt)
    __device__ __forceinline__ float4 mat(int p, int row) const { return v[off_mat + 3 * p + row]; }
    __device__ __forceinline__ int order(int p) const { return __float_as_int(v[off_mat + 3 * p + 2].w); }
};

// sign(float3) component (Common.hpp:328-333): t != 0 ? t / abs(t) : 0.  For finite t != 0 the quotient is exactly +-1, for
// +-inf and NaN it is NaN: copysign(1, t) + (t - t) has the same value in every case (t - t is +0 or NaN) without the divide.
__device__ __forceinline__ float sign1(float t) { return t != 0 ? __builtin_copysignf(1.0f, t) + (t - t) : 0.0f; }
// template max / min (Common.hpp:344-351)
__device__ __forceinline__ float tmax(float a, float b) { return a > b ? a : b; }
__device__ __forceinline__ float tmin(float a, float b) { return a < b ? a : b; }

struct BoxRay {  // per-ray part of Box::iBox (Object.hpp:175): depends on rd only
    V3 sgn, m, am;
};
__device__ __forceinline__ BoxRay box_ray_setup(V3 rd) {
    BoxRay b;
    b.sgn = v3(sign1(rd.x), sign1(rd.y), sign1(rd.z));
    const float eps = (float)1e-8;
    // (every direction that gets here comes out of normalized(): components in [-1, 1] or NaN, so the denominators lie in
    // [1e-8, 1] — tmax turns a NaN into eps — and the numerators are +-1, +0 or NaN: inside div_window's window)
    b.m = v3(div_window(b.sgn.x, tmax(fabsf(rd.x), eps)), div_window(b.sgn.y, tmax(fabsf(rd.y), eps)), div_window(b.sgn.z, tmax(fabsf(rd.z), eps)));
    b.am = v3(fabsf(b.m.x), fabsf(b.m.y), fabsf(b.m.z));
    return b;
}
// Box::iBox (Object.hpp:173-200) distance part; t1 out for the normal.
// NO_NAN: the caller has seen that the direction of every lane that counts is finite and that its origin and the scene's boxes are
// below 1e29 in size; then no slab distance is a NaN or an infinity on the way (slopes are +-1e8 at most in size, or +0 for a zero
// component; products below 2e37, sums below 4e37) and `a > b ? a : b` differs from the
// hardware's max / min only in which zero comes out of (+0, -0) — and tN, tF go nowhere but into comparisons and, from 0.01 up,
// into the result: one v_max3_f32 / v_min3_f32 instead of four compare-and-select pairs.
template <bool NO_NAN>
__device__ __forceinline__ float ibox_dist(const BoxRay& br, V3 ro, V3 size, V3& t1) {
    V3 n = v3(br.m.x * ro.x, br.m.y * ro.y, br.m.z * ro.z);
    V3 k = v3(br.am.x * size.x, br.am.y * size.y, br.am.z * size.z);
    t1 = v3(n.x * -1 - k.x, n.y * -1 - k.y, n.z * -1 - k.z);
    V3 t2 = v3(n.x * -1 + k.x, n.y * -1 + k.y, n.z * -1 + k.z);
    float tN = NO_NAN ? __builtin_fmaxf(__builtin_fmaxf(t1.x, t1.y), t1.z) : tmax(tmax(t1.x, t1.y), t1.z);
    float tF = NO_NAN ? __builtin_fminf(__builtin_fminf(t2.x, t2.y), t2.z) : tmin(tmin(t2.x, t2.y), t2.z);
    const float FMAX = 3.402823466e+38f;
    if (tN > tF || tF <= 0.0f) return FMAX;
    if (tN >= (float)0.01 && tN <= 10000.0f) return tN;
    if (tF >= (float)0.01 && tF <= 10000.0f) return tF;
    return FMAX;
}
// normal = -sign(rd) * step(t1.yzx, t1) * step(t1.zxy, t1)   (Object.hpp:189,193)
__device__ __forceinline__ V3 ibox_normal(const BoxRay& br, V3 t1) {
    V3 s1 = v3(t1.y <= t1.x ? 1.0f : 0.0f, t1.z <= t1.y ? 1.0f : 0.0f, t1.x <= t1.z ? 1.0f : 0.0f);
    V3 s2 = v3(t1.z <= t1.x ? 1.0f : 0.0f, t1.x <= t1.y ? 1.0f : 0.0f, t1.y <= t1.z ? 1.0f : 0.0f);
    return v3((br.sgn.x * -1) * s1.x * s2.x, (br.sgn.y * -1) * s1.y * s2.y, (br.sgn.z * -1) * s1.z * s2.z);
}

// Wave-uniform counts of what the loops of the path pool and of closest_hit do for a tile — the trip counts the kernel's
// instruction count (and, for meshes, its memory round trips) follow from.  Kept by the TALLY instantiations only (scalar adds on
// wave-uniform values; the steady-state kernels get the empty Tally: no code, no scalar registers).  Three things are made from them:
//   * the balance probe (PROBE instantiation, srt_estimate_row_costs): the first TALLY_N words per 16 x 16 block, weighed by the
//     host (srt_capi.hip, ProbeWeights);
//   * the launch-shape record (round 4): the launch that records block costs for the dispatch order — the first launch of a band
//     after the scene or the camera changed — also records the counts of every wave of every block; the host weighs them like the
//     probe's (ProbeWeights) and the sample-chunk rule of srt_render reads only that: counts, not times, so the same inputs give
//     the same launch shape in every run;
//   * SRT_RENDER_COUNT_WORK: the launch's totals (srt_get_work_counts), from which bench.py prices the EXECUTED lane-operations of
//     its roofline line.
// (Also counted in round 3 and dropped, their fitted weights came out at nothing: second halves of the sphere test, trips of the
// scatter loop, fold iterations, boxes with a valid hit, steps with an environment lookup, traced and untraced pixels; rays per
// mesh phase.)
enum {
    TALLY_STEPS = 0,       // pool steps (one closest_hit call of the whole wave each)
    TALLY_GROUPS,          // groups of four clustered spheres put through the exact test (a round of up to 64 items: K / 4, however many lanes share an item)
    TALLY_NODE_ROUNDS,     // mesh traversal: node rounds
    TALLY_LEAF_TRIPS,      // mesh traversal: triangle trips of the leaf rounds (one triangle test per lane each)
    TALLY_MESH_PHASES,     // mesh traversal: phases started
    TALLY_WAVES,           // waves (each stages the scene and traces its pixels' primary rays — once per sample chunk in a real launch)
    TALLY_UNTRACED_WAVES,  // waves with an untraced pixel (the sample-independent colour is folded sample by sample, once per wave)
    TALLY_NODE_TESTS,      // mesh traversal: child boxes tested per lane, summed over the node rounds (8, 4, 2 or 1 a round)
    TALLY_N = 8,           // words per block of the balance probe's record (tools/band_fit.py, tools/emulate_ranks.py)
    TALLY_CALLS = 8,       // closest_hit calls (pool steps + the primary ray's): every one runs all uniform spheres and all boxes
    TALLY_BOUND_CALLS,     // ... of which went through the cluster bounds (all of them, unless a lane's direction was not unit length)
    TALLY_ITEMS,           // (ray, cluster) pairs that survived the bounds: the USEFUL lanes of the exact rounds
    TALLY_SPHERE_TESTS,    // candidate tests of clustered spheres per lane (a round: K with one lane per item, K / 2 or K / 4 when two or four lanes share one)
    TALLY_ALL = 12
};
template <bool ON>
struct Tally {
    __device__ __forceinline__ void add(int, unsigned) {}
    __device__ __forceinline__ unsigned get(int) const { return 0u; }
};
// The counters live in ONE vector register: lane i of the wave holds counter i (TALLY_ALL <= 64).  An add is a compare, a select and
// an add on a wave-uniform amount — three VALU instructions, a handful of adds per pool step — instead of a scalar add on one of
// twelve scalar registers: the kernels have no scalar register to spare (the first version kept twelve SGPR counters and spilled
// 13..25 more of them into vector lanes inside the hot loops: the counting instantiation ran 4..5 % behind the plain one).
// add() must be called with all 64 lanes active (wave-uniform control flow), with a wave-uniform amount.
template <>
struct Tally<true> {
    unsigned v = 0u;
    unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    __device__ __forceinline__ void add(int i, unsigned x) { v += lane == (unsigned)i ? x : 0u; }
    __device__ __forceinline__ unsigned get(int i) const { return (unsigned)__builtin_amdgcn_readlane((int)v, i); }
};

// Loads and stores that are coherent across the GPU's XCDs without a fence (relaxed atomics at agent scope: gfx950 sets sc1 on
// them, they are served by the memory side of the L2s).  The chained chunks order them with the counters' own completion — the
// producer waits for its stores (s_waitcnt vmcnt(0)) before it raises the tile's count, the consumer's loads depend on the count
// it read — so neither side needs the whole-L2 write-back / invalidate an acquire or release fence at agent scope would bring.
__device__ __forceinline__ uint32_t coherent_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void coherent_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float4 coherent_load(const float4* p) {
    const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
    return make_float4(__uint_as_float(coherent_load(q)), __uint_as_float(coherent_load(q + 1)), __uint_as_float(coherent_load(q + 2)), __uint_as_float(coherent_load(q + 3)));
}
__device__ __forceinline__ void coherent_store(float4* p, const float4 v) {
    uint32_t* q = reinterpret_cast<uint32_t*>(p);
    coherent_store(q, __float_as_uint(v.x)), coherent_store(q + 1, __float_as_uint(v.y)), coherent_store(q + 2, __float_as_uint(v.z)), coherent_store(q + 3, __float_as_uint(v.w));
}

struct Hit {
    float t;   // distance of the recorded hit
    int prim;  // primitive id, -1 = miss (rayHit.valid == false)
    V3 n, p;   // normal, point (valid when prim >= 0)
};

// ---- (distance, list index, primitive) packed so that unsigned 64-bit order == the
// reference's rule "smaller distance wins, equal distance keeps the earlier list entry"
// (Raytracer.cpp:132).  -0 and +0 compare equal there, so the key uses +0 and keeps the
// original sign in bit 0.  Bits: [63:32] ordered distance | [31:17] list index | [16:1] prim.
__device__ __forceinline__ unsigned long long hit_key(float t, int ord, int p) {
    unsigned u = __float_as_uint(t + 0.0f);
    u ^= (u & 0x80000000u) ? 0xFFFFFFFFu : 0x80000000u;
    return ((unsigned long long)u << 32) | ((unsigned)ord << 17) | ((unsigned)p << 1) | (__float_as_uint(t) >> 31);
}
__device__ __forceinline__ void hit_unkey(unsigned long long k, float& t, int& p) {
    unsigned u = (unsigned)(k >> 32);
    u = (u & 0x80000000u) ? (u ^ 0x80000000u) : ~u;
    t = __uint_as_float(u);
    if (t == 0.0f) t = __uint_as_float((unsigned)(k & 1ull) << 31);
    p = (int)((k >> 1) & 0xFFFFu);
}

// GetClosestObject (Raytracer.cpp:123-140) over the LDS scene.  Must be called from
// wave-uniform control flow; lanes with active == false take part in the cooperative phases
// (they help test other lanes' rays) but trace nothing themselves.
//
// The reference scans ObjectsToRender in list order and keeps a hit only if its distance is
// strictly smaller (:132), i.e. the result is the lexicographic minimum of (distance, list
// index) over all valid hits.  We visit primitives in a different order, so an exact
// distance tie is resolved by comparing list indices.
//   1. uniform spheres: every lane, broadcast LDS reads, exact arithmetic.
//   2. clustered spheres: each lane marks the clusters its ray can possibly touch
//      (conservative bound, srt_scene_image.h); the wave then compacts all
//      (ray, cluster) pairs into a work list with ballot/mbcnt prefix sums, lanes pull items,
//      fetch the ray with __shfl, run the EXACT sphere arithmetic and merge through a 64-bit
//      LDS atomicMin on hit_key — so the wave does sum(pairs)/64 rounds, not max-per-lane.
//   3. boxes: every lane, exact arithmetic.
template <bool MESH, bool TALLY = false, bool SHORT_SQRT = false>
__device__ __forceinline__ Hit closest_hit(const Lds& S, const KernelParams& P, V3 o, V3 d, bool active, int defer_min, bool& deferred, Tally<TALLY>& tally SRT_PROF_PARAM) {
    tally.add(TALLY_CALLS, 1u);
    float best = __builtin_inff();
    int bp = -1;
    // exact sphere test of ray (ro, rd) against four spheres; updates (tb, pb) with the tie rule.
    // Part 1 (always): the cheap candidate test d2 <= r*r.  The second half (sqrt, compare) runs per lane
    // over the lane's own candidates (`candidates` below): a group nobody can hit costs a single branch.
    struct Cand {
        float tc, x;  // tc and r*r - d2
        bool c;
    };
    auto part1 = [&](const float4 s, V3 ro, V3 rd, bool on) {
        // Sphere::line_sphere_intersection (Object.hpp:104-141)
        float Lx = s.x - ro.x, Ly = s.y - ro.y, Lz = s.z - ro.z;                    // :115
        float tc = fabsf((Lx * rd.x + Ly * rd.y) + Lz * rd.z);                      // :118-119
        float qx = rd.x * tc + ro.x, qy = rd.y * tc + ro.y, qz = rd.z * tc + ro.z;  // :121
        float ex = qx - s.x, ey = qy - s.y, ez = qz - s.z;                          // :124
        float d2 = (ex * ex + ey * ey) + ez * ez;                                   // :125
        return Cand{tc, s.w - d2, on && !(d2 > s.w)};                               // :127
    };
    // Second half of the sphere test (sqrt, Object.hpp:131-133; the closest test and its tie rule, Raytracer.cpp:130-132) for a
    // lane's candidates among four spheres p .. p + 3 (bit i of m: sphere p + i is a candidate), PER LANE: every lane walks its own
    // candidates, lowest first, so the loop makes as many trips as the lane with the most candidates has — one, seldom two.  (Until
    // round 4 the second half ran once per sphere SLOT with a candidate in any lane: in an exact round every lane looks at another
    // cluster, so nearly every slot had one somewhere — about 3.3 second halves a round, each for a handful of lanes; and the
    // uniform spheres ran one per sphere.)  The result is the lexicographic minimum of (distance, list index) in whatever order the
    // candidates are looked at: same bits.  Interleaved A/B on one box (profiles/r04/ab_notes.txt): exact rounds: Scene1 1080p 32
    // spp 2.265 -> 2.200 ms, Scene_indirect -0.7 %, config 3's bands -0.7 %, config 4 -0.5 %, Scene3 +0.3 %; uniform spheres on top:
    // Scene1 -1.9 %, Scene1_reflection -1.8 %, config 3's floor band -3.1 %, config 4 -1.2 %, Scene3 -0.6 %, Scene_indirect +0.4 %.
    // Branch-free winner update on purpose (see the note in the triangle phase); a library sqrtf (its short form behind a
    // wave-uniform window test, as in normalized(), doubled the code of the many inlined copies for no gain: round 3).
    auto candidates = [&](const Cand& k0, const Cand& k1, const Cand& k2, const Cand& k3, unsigned m, int p, float& tb, int& pb) {
        while (__builtin_amdgcn_ballot_w64(m != 0u) != 0ull) {
            const bool c = m != 0u;
            const bool b1 = (m & 1u) == 0u, b2 = (m & 3u) == 0u, b3 = (m & 7u) == 0u;  // the lowest candidate is not slot 0 / not 0..1 / not 0..2
            const float tc = b3 ? k3.tc : b2 ? k2.tc : b1 ? k1.tc : k0.tc;
            const float x = b3 ? k3.x : b2 ? k2.x : b1 ? k1.x : k0.x;
            const int pj = p + (b3 ? 3 : b2 ? 2 : b1 ? 1 : 0);
            const float t1 = tc - (SHORT_SQRT ? sqrt_window(x) : sqrtf(x));  // :131-133 (lanes without a candidate compute on a dead value)
            // Raytracer.cpp:130-132; on an exact tie the earlier entry of ObjectsToRender wins
            const bool tie = c & (t1 == tb) & (pb >= 0);
            bool win = c & (t1 < tb);
            // rare: only then are the list indices needed (two LDS reads).  (The ballot of the bare compare — a superset of `tie` — is the
            // compare's own scalar result; the ballot of a combined predicate costs a select and a compare more.)
            if (__builtin_amdgcn_ballot_w64(t1 == tb) != 0ull) {
                const int op = S.order(pj), ob = S.order(tie ? pb : pj);
                win = win | (tie & (op < ob));
            }
            tb = win ? t1 : tb;
            pb = win ? pj : pb;
            m &= m - 1u;
        }
    };
    auto test4c = [&](const float4 s0, const float4 s1, const float4 s2, const float4 s3, int p, V3 ro, V3 rd, bool on, float& tb, int& pb) {
        const Cand k0 = part1(s0, ro, rd, on), k1 = part1(s1, ro, rd, on), k2 = part1(s2, ro, rd, on), k3 = part1(s3, ro, rd, on);
        candidates(k0, k1, k2, k3, (k0.c ? 1u : 0u) | (k1.c ? 2u : 0u) | (k2.c ? 4u : 0u) | (k3.c ? 8u : 0u), p, tb, pb);
    };
    SRT_TICK(2);
    // ---- 1. uniform spheres: broadcast ds_read_b128, 4 per trip
    for (int j = 0; j + 4 <= S.nu; j += 4) {
        const float4 s0 = S.v[j], s1 = S.v[j + 1], s2 = S.v[j + 2], s3 = S.v[j + 3];
        test4c(s0, s1, s2, s3, j, o, d, active, best, bp);
    }
    if (S.nu & 3) {  // the last, partial group without its padding (Scene1 has 3 uniform spheres, Scene3 / Scene_indirect 2)
        const int j = S.nu & ~3, rem = S.nu & 3;
        const float4 s0 = S.v[j], s1 = S.v[j + 1], s2 = S.v[j + 2];
        const Cand k0 = part1(s0, o, d, active);
        Cand k1 = k0, k2 = k0;
        unsigned m = k0.c ? 1u : 0u;
        if (rem >= 2) k1 = part1(s1, o, d, active), m |= k1.c ? 2u : 0u;
        if (rem == 3) k2 = part1(s2, o, d, active), m |= k2.c ? 4u : 0u;
        candidates(k0, k1, k2, k0, m, j, best, bp);
    }
    SRT_TICK(3);
    // ---- 2. clustered spheres
    if (S.nc > 0) {
        const float dd = __builtin_fmaf(d.z, d.z, __builtin_fmaf(d.y, d.y, d.x * d.x));
        const bool unit = fabsf(dd - 1.0f) <= 1e-6f;  // false for NaN
        if (__builtin_amdgcn_ballot_w64(active && !unit) == 0ull) {
            const float o1 = (fabsf(o.x) + fabsf(o.y)) + fabsf(o.z);
            unsigned mlo = 0u, mhi = 0u;  // this lane's clusters: bit k of (mhi : mlo) = the ray may touch cluster k
            // (Tried in round 4 and dropped: ONE bound around all clustered spheres first — the same test, the same proof — so that a wave
            // none of whose rays passes it skips the nc cluster bounds: Scene3 -1.1 %, config 3's middle band -1.4 %, but Scene1 +1.9 %,
            // Scene_indirect +2.0 %, config 3's floor band +1.7 %: where the rays are, some lane nearly always points at the grid.)
            tally.add(TALLY_BOUND_CALLS, 1u);
            // The lane's mask is shifted in bit by bit — a select and a share of a shift-or per bound, where `mask |= 1ull << k` cost two
            // moves of the scalar bit into vector registers, two selects and an or — and turned round once at the end; it stays in two
            // 32-bit words (the 64-bit find-first-set and clear-lowest of the scatter loop were 12 instructions a trip, now 7; scenes
            // with more than 32 clusters take a second loop).  Round 4; same bits; interleaved A/B (profiles/r04/ab_notes.txt): config 2
            // 2.165 -> 2.085 ms, config 3's middle band -2.7 %, one-sample launches -2.3 %, Scene3 -1.8 %, config 4 -1.6 %, the others -0.5 %.
            auto passes = [&](int k) {
                const float4 b = S.bound(k);
                float Lx = b.x - o.x, Ly = b.y - o.y, Lz = b.z - o.z;
                float LL = __builtin_fmaf(Lz, Lz, __builtin_fmaf(Ly, Ly, Lx * Lx));
                float sd = __builtin_fmaf(Lz, d.z, __builtin_fmaf(Ly, d.y, Lx * d.x));
                float Rinf = __builtin_fmaf(8e-6f, o1, b.w);
                float lhs = __builtin_fmaf(-sd, sd, LL);
                float rhs = __builtin_fmaf(4e-6f, LL, Rinf * Rinf);
                return lhs <= rhs ? 1u : 0u;
            };
            {
                const int n_lo = S.nc < 32 ? S.nc : 32;
                for (int k = 0; k < n_lo; ++k) mlo = (mlo << 1) | passes(k);  // phase 1: conservative cluster bounds, uniform reads
                mlo = active ? __builtin_bitreverse32(mlo) >> (32 - n_lo) : 0u;  // (nc > 0 here)
                if (S.nc > 32) {
                    for (int k = 32; k < S.nc; ++k) mhi = (mhi << 1) | passes(k);
                    mhi = active ? __builtin_bitreverse32(mhi) >> (64 - S.nc) : 0u;
                }
            }
            const int K4 = S.K >> 2;
            // exclusive prefix sum of the lanes' cluster counts over the wave: one DPP scan (six adds; seven ballot slices before)
            const int cnt = __builtin_popcount(mlo) + __builtin_popcount(mhi);
            const unsigned incl = wave_inclusive_scan((unsigned)cnt);
            const int prefix = (int)incl - cnt;
            const int total = __builtin_amdgcn_readlane((int)incl, 63);
            SRT_TICK(4);
#if defined(SRT_STATS) && SRT_STATS == 4
            {
                const int st_active = __builtin_popcountll(__builtin_amdgcn_ballot_w64(active));
                SRT_STAT(2, total);
                SRT_STAT(3, (total + 63) / 64);
                SRT_STAT(7, st_active);
            }
#endif
            tally.add(TALLY_ITEMS, (unsigned)total);
            if (total > 0 && total <= WORK_MAX) {
                const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
                // the lane's own best so far enters the merge slot
                S.res[lane] = bp >= 0 ? hit_key(best, S.order(bp), bp) : ~0ull;
                {  // scatter this lane's (lane, cluster) items behind its prefix
                    unsigned m = mlo;
                    int w = prefix;
                    while (__builtin_amdgcn_ballot_w64(m != 0u) != 0ull) {
                        if (m != 0u) {
                            S.work[w++] = (unsigned short)((lane << 8) | __builtin_ctz(m));
                            m &= m - 1u;
                        }
                    }
                    if (S.nc > 32) {
                        m = mhi;
                        while (__builtin_amdgcn_ballot_w64(m != 0u) != 0ull) {
                            if (m != 0u) {
                                S.work[w++] = (unsigned short)((lane << 8) | (32 + __builtin_ctz(m)));
                                m &= m - 1u;
                            }
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
                // A round takes up to 64 items with one lane each — or, when no more than 32 / 16 are left, two / four lanes per item, each
                // with two / one of a group's four spheres (round 4): the last round of a step is seldom full — Scene1 has 79 items in
                // the average step, the second round used to run all four candidate tests in 64 lanes for 15 of them.  The merge is an
                // atomicMin, so it does not matter how many lanes report for a ray.  Same bits; interleaved A/B
                // (profiles/r04/ab_notes.txt): config 2 2.179 -> 2.121 ms, one-sample launches -1.9 %, config 3's middle band -1.2 %,
                // Scene_indirect -1.1 %, Scene1_reflection -0.9 %, Scene3 -0.5 %, config 3's floor band +-0, config 4 +0.7 %.
                for (int base = 0; base < total;) {
                    const int rem = total - base;
                    const int logS = rem <= 16 ? 2 : rem <= 32 ? 1 : 0, n = 4 >> logS;  // (wave-uniform)
                    const int take = rem < (64 >> logS) ? rem : (64 >> logS);
                    const int slot = lane >> logS, sub = lane & ((1 << logS) - 1);
                    const bool on = slot < take;
                    const unsigned item = on ? S.work[base + slot] : (unsigned)(lane << 8);
                    base += take;
                    const int src = (int)(item >> 8), k = (int)(item & 63u);
                    V3 ro = v3(__shfl(o.x, src), __shfl(o.y, src), __shfl(o.z, src));
                    V3 rd = v3(__shfl(d.x, src), __shfl(d.y, src), __shfl(d.z, src));
                    float tb = __builtin_inff();
                    int pb = -1;
                    tally.add(TALLY_GROUPS, (unsigned)K4);
                    tally.add(TALLY_SPHERE_TESTS, (unsigned)(K4 * n));
                    for (int i = 0; i < K4; ++i) {
                        const int p = S.nu4 + (k * K4 + i) * 4 + sub * n;  // per-lane LDS gather
                        const Cand k0 = part1(S.v[p], ro, rd, on);
                        Cand k1 = k0, k2 = k0, k3 = k0;
                        unsigned m = k0.c ? 1u : 0u;
                        if (n >= 2) {
                            k1 = part1(S.v[p + 1], ro, rd, on), m |= k1.c ? 2u : 0u;
                            if (n == 4) {
                                k2 = part1(S.v[p + 2], ro, rd, on), m |= k2.c ? 4u : 0u;
                                k3 = part1(S.v[p + 3], ro, rd, on), m |= k3.c ? 8u : 0u;
                            }
                        }
                        candidates(k0, k1, k2, k3, m, p, tb, pb);
                    }
                    if (pb >= 0) atomicMin(&S.res[src], hit_key(tb, S.order(pb), pb));
                }
                __builtin_amdgcn_wave_barrier();
                const unsigned long long r = S.res[lane];
                if (r != ~0ull) hit_unkey(r, best, bp);
            } else if (total > WORK_MAX) {  // pathological: per-lane loop over own candidates
                unsigned long long mask = ((unsigned long long)mhi << 32) | mlo;
                while (__builtin_amdgcn_ballot_w64(mask != 0ull) != 0ull) {
                    const bool on = mask != 0ull;
                    const int k = on ? __builtin_ctzll(mask) : 0;
                    mask &= mask - 1ull;
                    tally.add(TALLY_GROUPS, (unsigned)K4);
                    tally.add(TALLY_SPHERE_TESTS, (unsigned)(K4 * 4));
                    for (int i = 0; i < K4; ++i) {
                        const int p = S.nu4 + (k * K4 + i) * 4;
                        const float4 s0 = S.v[p], s1 = S.v[p + 1], s2 = S.v[p + 2], s3 = S.v[p + 3];
                        test4c(s0, s1, s2, s3, p, o, d, on, best, bp);
                    }
                }
            }
        } else {  // some lane's direction is not unit length (degenerate lerp): brute force
            tally.add(TALLY_GROUPS, (unsigned)((S.nsT - S.nu4) >> 2));
            tally.add(TALLY_SPHERE_TESTS, (unsigned)(S.nsT - S.nu4));
            for (int j = S.nu4; j < S.nsT; j += 4) {
                const float4 s0 = S.v[j], s1 = S.v[j + 1], s2 = S.v[j + 2], s3 = S.v[j + 3];
                test4c(s0, s1, s2, s3, j, o, d, active, best, bp);
            }
        }
    }
    SRT_TICK(5);
    // ---- 3. boxes
    // (Tried in round 3 and dropped: boxes BEFORE the clusters plus a cull of clusters whose nearest possible hit, |s| - R - slack,
    // lies beyond the best distance so far — valid, all tests bit-exact, but the two extra instructions per bound cost more than
    // the culled exact tests saved: Scene_indirect +1.4 %, Scene1 +2.3 %, config 4 +3.4 %.)
    Hit h;
    V3 bt1 = v3(0, 0, 0);
    BoxRay br;
    const int nsT = S.nsT, nb = S.nb;
    if (nb > 0) {
        br = box_ray_setup(d);
        // (a NaN or infinity anywhere in a ray that counts, or an origin so far out that a slab product could overflow — |o|_1 at or
        // beyond 1e29, the bound srt_set_scene holds the boxes to as well, srt_scene_image.h — sends the wave through the
        // comparisons as the reference writes them)
        const float omag = (fabsf(o.x) + fabsf(o.y)) + fabsf(o.z), dsum = (d.x + d.y) + d.z;
        const bool no_nan = (P.flags & KF_BOXES_FINITE) != 0 && __builtin_amdgcn_ballot_w64(active && !(omag < 1e29f && fabsf(dsum) < __builtin_inff())) == 0ull;
        auto boxes = [&](auto tag) {
            for (int j = 0; j < nb; ++j) {
                const float4 c = S.box_c(j), hs = S.box_h(j);
                V3 t1;
                float dist = ibox_dist<decltype(tag)::value>(br, v3(o.x - c.x, o.y - c.y, o.z - c.z), v3(hs.x, hs.y, hs.z), t1);
                bool valid = active && dist != 3.402823466e+38f;  // Object.hpp:231
                if (__builtin_amdgcn_ballot_w64(dist != 3.402823466e+38f) != 0ull) {  // (bare compares: see `candidates`; supersets of valid / tie)
                    const bool tie = valid & (dist == best) & (bp >= 0);
                    bool win = valid & (dist < best);
                    if (__builtin_amdgcn_ballot_w64(dist == best) != 0ull) {  // rare: list indices only on an exact tie
                        const int ob = S.order(tie ? bp : nsT + j);
                        win = win | (tie & (S.order(nsT + j) < ob));
                    }
                    best = win ? dist : best;
                    bp = win ? nsT + j : bp;
                    // component-wise: a whole-struct select is lowered to a pointer select + copies through scratch
                    bt1 = v3(win ? t1.x : bt1.x, win ? t1.y : bt1.y, win ? t1.z : bt1.z);
                }
            }
        };
        if (no_nan) boxes(std::true_type{});
        else boxes(std::false_type{});
    }
    // ---- 4. EXTENSION: triangle meshes — traversal of the host-built 8-wide BVH (HBM/L2).
    // The triangle arithmetic is this project's definition (srt_pathtrace.h); the box filter is
    // conservative: boxes are padded per ray by 1e-5 * (|o - centre|_1 + |centre|_1 + mesh size), a
    // bound on every coordinate and distance involved, far above the rounding of the
    // Moller-Trumbore test (~1e-6 * |o - v0|) and of the plane distances below (~4e-7 of the same
    // magnitudes); the slab comparison itself has slack for the approximate reciprocals.
    int btri = -1;               // winner's position in the triangle array (-1: not a triangle)
    V3 tri_n = v3(0, 0, 0);      // its e1 x e2
    int bord = 0x7fffffff;       // list index of the best analytic hit's object
    if constexpr (MESH) {
        if (P.n_tris > 0) {
            if (bp >= 0) bord = S.order(bp);
            const float pad = 1e-5f * (((fabsf(o.x - P.mesh_center[0]) + fabsf(o.y - P.mesh_center[1])) + fabsf(o.z - P.mesh_center[2])) + P.mesh_r1) + 1e-7f;
            const V3 inv = v3(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
            // slopes clamped to 2^96 for the traversal: an axis the ray (almost) does not move along keeps its
            // sign-correct, still astronomically large plane distances without inf * 0
            const V3 rinv_own = v3(fminf(fmaxf(inv.x, -0x1p96f), 0x1p96f), fminf(fmaxf(inv.y, -0x1p96f), 0x1p96f), fminf(fmaxf(inv.z, -0x1p96f), 0x1p96f));
            // root box (kernel argument, no memory access): most rays never come near the mesh, and
            // when no lane of the wave does, the whole phase is skipped
            bool go;
            {
                float t1x = ((P.mesh_center[0] - P.mesh_half[0] - pad) - o.x) * inv.x, t2x = ((P.mesh_center[0] + P.mesh_half[0] + pad) - o.x) * inv.x;
                float t1y = ((P.mesh_center[1] - P.mesh_half[1] - pad) - o.y) * inv.y, t2y = ((P.mesh_center[1] + P.mesh_half[1] + pad) - o.y) * inv.y;
                float t1z = ((P.mesh_center[2] - P.mesh_half[2] - pad) - o.z) * inv.z, t2z = ((P.mesh_center[2] + P.mesh_half[2] + pad) - o.z) * inv.z;
                float tmin = fmaxf(fmaxf(fminf(t1x, t2x), fminf(t1y, t2y)), fminf(t1z, t2z));
                float tmax = fminf(fminf(fmaxf(t1x, t2x), fmaxf(t1y, t2y)), fmaxf(t1z, t2z));
                // a triangle hit is only valid for t >= 0.01 (the Box distance bounds, Object.hpp:226) and lies inside the padded box, so
                // a box the ray has left before t = 0.0099 holds nothing for it — this is what lets a bounce ray that starts
                // ON the mesh and points away from it skip the traversal instead of descending to the leaf it came from
                go = active && tmin <= tmax * 1.00001f + 1e-6f && tmax * 1.00001f + 1e-6f >= MESH_T_MIN_CULL && tmin <= 10001.0f && tmin * 0.9999f - 1e-5f <= best;
            }
            // bounding sphere of all triangles (centre C = the root box's, radius R): the ray can only hit something while it
            // is inside it.  s = (C - o).d, D^2 = |C - o|^2 - s^2; inside for t in [s - q, s + q], q = sqrt(R'^2 - D^2), with
            // R' = R + 2 * pad (pad bounds every rounding error of the coordinates involved, see above) and the same slack on
            // D^2 as the cluster bound of srt_scene_image.h.  Needs a unit-length direction; otherwise the box test stands alone.
            // What this buys: a bounce ray that starts on a round mesh and points away from it leaves the sphere before
            // t = 0.0099 and skips the traversal, where the box hierarchy would be descended down to the leaf it came from.
            {
                const float Lx = P.mesh_center[0] - o.x, Ly = P.mesh_center[1] - o.y, Lz = P.mesh_center[2] - o.z;
                const float LL = __builtin_fmaf(Lz, Lz, __builtin_fmaf(Ly, Ly, Lx * Lx));
                const float sd = __builtin_fmaf(Lz, d.z, __builtin_fmaf(Ly, d.y, Lx * d.x));
                const float dd = __builtin_fmaf(d.z, d.z, __builtin_fmaf(d.y, d.y, d.x * d.x));
                const float Rp = P.mesh_bs_radius + 2.0f * pad;
                const float disc = __builtin_fmaf(4e-6f, LL, Rp * Rp) - __builtin_fmaf(-sd, sd, LL);
                const float q = __builtin_amdgcn_sqrtf(fmaxf(disc, 0.0f)) * 1.00001f;
                const bool unit = fabsf(dd - 1.0f) <= 1e-6f;
                const bool inside = disc >= 0.0f && (sd + q) + pad >= MESH_T_MIN_CULL && ((sd - q) - pad) * 0.9999f - 1e-5f <= best;
                go = go && (inside || !unit);
            }
            // ---- wave-cooperative traversal of the 8-wide BVH.  Rays that come near the mesh put
            // (ray lane, node) items on a LIFO in LDS.  A node round pops items and tests the 8 quantized
            // child boxes of each, spread over 8 / 4 / 2 / 1 lanes by the number of items waiting (a lone ray
            // still uses the wave and descends a level per round).  Surviving children are pushed behind the
            // offsets of one wave scan, leaves to a second queue that a leaf round drains — a leaf's <= 4
            // triangles over 4 / 2 / 1 lanes.
            // A ray's best triangle is merged through a 64-bit LDS atomicMin on (ordered t || global
            // triangle id) — the id order is (list order of the object, triangle index), i.e. the tie
            // rule — and doubles as the culling distance.  The merge is idempotent, so when a queue would
            // overflow the batch is simply abandoned and redone with fewer rays at a time; one ray
            // popping one item per round (strict depth-first) is bounded by 7 * depth + 8 entries, which
            // the host checks against MESH_Q.
            //
            // Deferral (path pool only): a mesh phase costs the wave some seven rounds of ~350 instructions however
            // few rays take part, and most phases would be started by two or three stray bounce rays.  So
            // unless at least `defer_min` rays want the mesh — or no lane has anything else to do — the
            // rays are reported back as deferred: the pool parks them and offers them again next round.
#ifdef SRT_DEV  // timing experiments (results are wrong): 0x100 = no mesh phases at all, 0x200 = no triangle tests
            if (P.flags & 0x100u) go = false;
#endif
            unsigned long long pend = __builtin_amdgcn_ballot_w64(go);
            const int n_go = __builtin_popcountll(pend);
            if (n_go < defer_min && n_go != __builtin_popcountll(__builtin_amdgcn_ballot_w64(active))) {
                deferred = go;
                pend = 0ull;
            } else {
                deferred = false;
            }
            if (pend != 0ull) {
                tally.add(TALLY_MESH_PHASES, 1u);
                const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
                unsigned* qn = S.meshq;               // node item i at qn[i]
                unsigned* qlt = S.meshq + MESH_Q - 1;  // leaf item i at qlt[-i]: both queues share the buffer's free middle
                auto okey = [](float t) {  // order-preserving float -> uint
                    unsigned u = __float_as_uint(t + 0.0f);
                    return u ^ ((u & 0x80000000u) ? 0xFFFFFFFFu : 0x80000000u);
                };
                auto unkey = [](unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u ^ 0x80000000u) : ~u); };
                bool overflow = false;
                auto mbcnt64 = [](unsigned long long m) { return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)); };
                auto push = [&](bool pred, unsigned item, unsigned* q, int& n, int cap) {
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(pred);
                    const int cnt = __builtin_popcountll(m);
                    if (n + cnt > cap) {
                        overflow = true;
                    } else {
                        if (pred) q[n + mbcnt64(m)] = item;
                        n += cnt;
                    }
                };
                S.res[lane] = ((unsigned long long)okey(best) << 32) | 0xFFFFFFFFull;  // no triangle yet
                int batch = 64;
                bool strict = false;
#if defined(SRT_STATS) && (SRT_STATS == 7 || SRT_STATS == 8)
                long long seg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
#if defined(SRT_STATS) && SRT_STATS == 5  // make dev STATS=5: wave-cycles per mesh phase and per wave life (tests/mesh_stats.py)
                const long long st_t0 = (long long)__builtin_readcyclecounter();
#endif
#ifdef SRT_STATS
                const int st_cls = n_go <= 2 ? 0 : n_go <= 8 ? 1 : n_go <= 32 ? 2 : 3;
                int st_rounds = 0;
                if (SRT_STATS == 1) {
                    SRT_STAT(1, n_go);
                    SRT_STAT(0, 1);
                }
#endif
                while (pend != 0ull) {
                    // the next (up to `batch`) waiting rays enter at the root
                    const bool mine = (pend >> lane) & 1ull;
                    const int rank = mbcnt64(pend);
                    const bool sel = mine && rank < batch;
                    const unsigned long long selmask = __builtin_amdgcn_ballot_w64(sel);
                    int nN = 0, nL = 0;
                    overflow = false;
                    push(sel, (unsigned)lane << 26, qn, nN, MESH_Q);
                    __builtin_amdgcn_wave_barrier();
                    while (!overflow && (nN > 0 || nL > 0)) {
                        // ---- one round pops items of ONE queue.  (Tried and measured slower on config 4: serving both queues in
                        // every round — rounds per phase 11.0 -> 9.9, but every round then runs both code paths, +8 % time; a
                        // leaf queue of single triangles, one per lane — fewer leaf rounds, but more queue traffic and the
                        // node rounds get throttled by the leaf queue's room, +7 %.)
                        // Nodes first, until 64 leaves wait.  The two kinds of round are separate branches — each pops, loads and
                        // fetches its rays on its own — so that neither pays for selects between node and triangle data.
                        const bool node_round = nN > 0 && nL < 64;
#ifdef SRT_STATS
                        st_rounds += 1;
#endif
#if defined(SRT_STATS) && (SRT_STATS == 7 || SRT_STATS == 8)
                        long long seg_t = (long long)__builtin_readcyclecounter();
#endif
                        if (node_round) {
                            tally.add(TALLY_NODE_ROUNDS, 1u);
                            int logP = 3, takeN = 1;
                            if (!strict) {
                                logP = nN <= 8 ? 3 : nN <= 16 ? 2 : nN <= 32 ? 1 : 0;
                                takeN = nN < (64 >> logP) ? nN : (64 >> logP);
                                // leave room for the expected pushes (about 3 per item)
                                int room = (MESH_Q - nN - nL) / 6;  // (about 3 pushes per item to either queue)
                                room = room < 8 ? 8 : room;
                                takeN = takeN < room ? takeN : room;
                            }
                            nN -= takeN;  // the items [nN, nN + takeN) are popped
                            tally.add(TALLY_NODE_TESTS, 8u >> logP);
#ifdef SRT_STATS
                            if (SRT_STATS == 1) {
                                SRT_STAT(2, 1);
                                SRT_STAT(3, takeN);
                            }
#endif
                            const int slotN = lane >> logP, sub = lane & ((1 << logP) - 1);
                            const bool onN = slotN < takeN;
                            const unsigned item = onN ? qn[nN + slotN] : 0u;
                            SRT_SEG(0, item);  // the pop
                            const int src = (int)(item >> 26), code = (int)(item & 0x3FFFFFFu);
                            // the five rows of the node are requested before the ray is fetched, so that the memory round trip overlaps
                            // the shuffles.  (Every lane loads — idle lanes the root: a load under a lane mask would make the compiler wait
                            // for it right here, to merge the registers with those of the lanes that do not load.  The top levels of the
                            // tree need no copy in LDS: they stay in the vector L1, a copy measured no faster.)
                            const float4* rowp = P.bvh_nodes + 5 * (size_t)code;
                            const float4 h0 = rowp[0], h1 = rowp[1], q0 = rowp[2], q1 = rowp[3], q2 = rowp[4];
                            const V3 ro = v3(__shfl(o.x, src), __shfl(o.y, src), __shfl(o.z, src));
                            const V3 rinv = v3(__shfl(rinv_own.x, src), __shfl(rinv_own.y, src), __shfl(rinv_own.z, src));
                            const float rpad = __shfl(pad, src);
                            const float thr = unkey((unsigned)(S.res[src] >> 32));
                            // a child is entered only if a triangle inside could still beat the ray's best hit: entry distance
                            // <= (thr + 1e-5) / 0.9999 (written as an upper bound of it), and <= 10001
                            const float thr2 = fminf(__builtin_fmaf(fabsf(thr), 2e-4f, thr + 1e-5f), 10001.0f);
                            SRT_SEG(1, ((((h0.x + h1.x) + q0.x) + (q1.x + q2.w)) + ((ro.x + ro.y) + ro.z)) + (((rinv.x + rinv.y) + rinv.z) + (rpad + thr2)));  // node rows, ray
                            const unsigned ex = __float_as_uint(h0.w);
                            const unsigned innermask = ex >> 24, lw = __float_as_uint(h1.z), leafmask = lw & 255u, counts = lw >> 8;
                            // plane distance = q * (cell * rinv) + ((origin -/+ pad) - ro) * rinv, one FMA per plane
                            const float sx = __uint_as_float((ex & 255u) << 23) * rinv.x, sy = __uint_as_float(((ex >> 8) & 255u) << 23) * rinv.y,
                                        sz = __uint_as_float(((ex >> 16) & 255u) << 23) * rinv.z;
                            const float lx0 = ((h0.x - rpad) - ro.x) * rinv.x, hx0 = ((h0.x + rpad) - ro.x) * rinv.x;
                            const float ly0 = ((h0.y - rpad) - ro.y) * rinv.y, hy0 = ((h0.y + rpad) - ro.y) * rinv.y;
                            const float lz0 = ((h0.z - rpad) - ro.z) * rinv.z, hz0 = ((h0.z + rpad) - ro.z) * rinv.z;
                            // the ray's direction picks the entry and the exit plane of every axis once per node (lo <= hi and a
                            // monotone FMA, so this equals the min / max of the two plane distances); an absent child has an
                            // inverted box and sits in neither mask
                            const bool px = rinv.x >= 0.0f, py = rinv.y >= 0.0f, pz = rinv.z >= 0.0f;
                            const float bnx = px ? lx0 : hx0, bfx = px ? hx0 : lx0, bny = py ? ly0 : hy0, bfy = py ? hy0 : ly0, bnz = pz ? lz0 : hz0,
                                        bfz = pz ? hz0 : lz0;
                            // byte planes: children 0..3 in the first word of a plane, 4..7 in the second
                            const unsigned lox0 = __float_as_uint(q0.x), lox1 = __float_as_uint(q0.y), loy0 = __float_as_uint(q0.z), loy1 = __float_as_uint(q0.w);
                            const unsigned loz0 = __float_as_uint(q1.x), loz1 = __float_as_uint(q1.y), hix0 = __float_as_uint(q1.z), hix1 = __float_as_uint(q1.w);
                            const unsigned hiy0 = __float_as_uint(q2.x), hiy1 = __float_as_uint(q2.y), hiz0 = __float_as_uint(q2.z), hiz1 = __float_as_uint(q2.w);
                            auto byte_f = [](unsigned w, int b) { return (float)((w >> (8 * b)) & 255u); };
                            // child whose six plane bytes are byte b of the given words: is its (padded) box entered?
                            auto enters = [&](unsigned nx, unsigned ny, unsigned nz, unsigned fx, unsigned fy, unsigned fz, int b) {
                                const float tnx = __builtin_fmaf(byte_f(nx, b), sx, bnx), tfx = __builtin_fmaf(byte_f(fx, b), sx, bfx);
                                const float tny = __builtin_fmaf(byte_f(ny, b), sy, bny), tfy = __builtin_fmaf(byte_f(fy, b), sy, bfy);
                                const float tnz = __builtin_fmaf(byte_f(nz, b), sz, bnz), tfz = __builtin_fmaf(byte_f(fz, b), sz, bfz);
                                const float tn = fmaxf(fmaxf(tnx, tny), tnz), tf = fminf(fminf(tfx, tfy), tfz);
                                // exit distance >= the smallest valid t, folded into the entry side; a NaN entry stays NaN and rejects
                                const float lo = (MESH_T_MIN_CULL > tn) ? MESH_T_MIN_CULL : tn;
                                const float hi = fminf(__builtin_fmaf(tf, 1.00001f, 1e-6f), thr2);
                                return lo <= hi;
                            };
                            unsigned mask = 0u;
                            if (logP == 0) {  // one lane per item: all 8 children
                                const unsigned nx0 = px ? lox0 : hix0, fx0 = px ? hix0 : lox0, ny0 = py ? loy0 : hiy0, fy0 = py ? hiy0 : loy0, nz0 = pz ? loz0 : hiz0,
                                               fz0 = pz ? hiz0 : loz0;
                                const unsigned nx1 = px ? lox1 : hix1, fx1 = px ? hix1 : lox1, ny1 = py ? loy1 : hiy1, fy1 = py ? hiy1 : loy1, nz1 = pz ? loz1 : hiz1,
                                               fz1 = pz ? hiz1 : loz1;
#pragma unroll
                                for (int b = 0; b < 4; ++b) {
                                    mask |= enters(nx0, ny0, nz0, fx0, fy0, fz0, b) ? (1u << b) : 0u;
                                    mask |= enters(nx1, ny1, nz1, fx1, fy1, fz1, b) ? (16u << b) : 0u;
                                }
                            } else {  // 2 / 4 / 8 lanes per item: this lane takes 4 / 2 / 1 consecutive children
                                const int nb = 8 >> logP;                 // children of this lane
                                const int first = sub * nb;                // its first child
                                const bool second = first >= 4;            // which word of the planes
                                const unsigned sh = (unsigned)(first & 3) * 8u;
                                const unsigned wlx = (second ? lox1 : lox0) >> sh, wly = (second ? loy1 : loy0) >> sh, wlz = (second ? loz1 : loz0) >> sh;
                                const unsigned whx = (second ? hix1 : hix0) >> sh, why = (second ? hiy1 : hiy0) >> sh, whz = (second ? hiz1 : hiz0) >> sh;
                                const unsigned nx = px ? wlx : whx, fx = px ? whx : wlx, ny = py ? wly : why, fy = py ? why : wly, nz = pz ? wlz : whz, fz = pz ? whz : wlz;
                                unsigned m4 = enters(nx, ny, nz, fx, fy, fz, 0) ? 1u : 0u;
                                if (nb >= 2) {
                                    m4 |= enters(nx, ny, nz, fx, fy, fz, 1) ? 2u : 0u;
                                    if (nb == 4) {
                                        m4 |= enters(nx, ny, nz, fx, fy, fz, 2) ? 4u : 0u;
                                        m4 |= enters(nx, ny, nz, fx, fy, fz, 3) ? 8u : 0u;
                                    }
                                }
                                mask = m4 << first;
                            }
                            if (!onN) mask = 0u;
                            SRT_SEG(2, mask);  // child tests
                            const unsigned tag = (unsigned)src << 26;
                            // exclusive prefix sums of both survivor counts in ONE wave scan (inner count in the low half, leaf count in
                            // the high half; six DPP adds instead of eight ballot rounds)
                            unsigned mi = mask & innermask, ml = mask & leafmask;
                            const unsigned both = (unsigned)__builtin_popcount(mi) | ((unsigned)__builtin_popcount(ml) << 16);
                            const unsigned incl = wave_inclusive_scan(both);
                            const unsigned total = (unsigned)__builtin_amdgcn_readlane((int)incl, 63), excl = incl - both;
                            const int totN = (int)(total & 0xFFFFu), totL = (int)(total >> 16);
                            SRT_SEG(3, excl + total);  // scan
                            if (nN + totN + nL + totL > MESH_Q) {
                                overflow = true;
                            } else {
                                if (totN > 0) {  // surviving inner children -> node LIFO
                                    const unsigned first_inner = __float_as_uint(h1.x);
                                    int w = nN + (int)(excl & 0xFFFFu);
                                    while (__builtin_amdgcn_ballot_w64(mi != 0u) != 0ull) {
                                        if (mi != 0u) {
                                            const unsigned below = (mi & (0u - mi)) - 1u;  // mask of the children before the lowest survivor
                                            qn[w++] = tag | (first_inner + (unsigned)__builtin_popcount(innermask & below));
                                            mi &= mi - 1u;
                                        }
                                    }
                                    nN += totN;
                                }
                                if (totL > 0) {  // surviving leaf children -> leaf queue, item = (first triangle) * 4 + (count - 1)
                                    const unsigned first_tri = __float_as_uint(h1.y);
                                    int w = nL + (int)(excl >> 16);  // (over this round's popped items: their reads have long completed)
                                    while (__builtin_amdgcn_ballot_w64(ml != 0u) != 0ull) {
                                        if (ml != 0u) {
                                            const unsigned bit = ml & (0u - ml), below = bit - 1u;
                                            const unsigned below2 = bit * bit - 1u;  // the count fields (2 bits each) of the children before
                                            const unsigned cf = counts & below2;
                                            const unsigned first = first_tri + (unsigned)__builtin_popcount(leafmask & below) + (unsigned)__builtin_popcount(cf & 0x5555u) +
                                                                   2u * (unsigned)__builtin_popcount(cf & 0xAAAAu);
                                            const unsigned c2 = 2u * (unsigned)__builtin_ctz(bit);
                                            qlt[-(w++)] = tag | (first * 4u + ((counts >> c2) & 3u));
                                            ml &= ml - 1u;
                                        }
                                    }
                                    nL += totL;
                                }
                            }
                            SRT_SEG(4, mi + ml);  // pushes
#if defined(SRT_STATS) && SRT_STATS == 7
                            seg_acc[6] += 1;
#endif
                        }
                        else {
                            // a leaf round spreads a leaf's <= 4 triangles over 4 / 2 / 1 lanes by the number of leaves waiting: up to 16
                            // leaves take one trip, up to 32 two, up to 64 four — one round instead of up to four
                            const int logL = nL <= 16 ? 2 : nL <= 32 ? 1 : 0;
                            const int takeL = nL < (64 >> logL) ? nL : (64 >> logL);
                            const int trips = 4 >> logL;
                            tally.add(TALLY_LEAF_TRIPS, (unsigned)trips);
                            nL -= takeL;  // the items [nL, nL + takeL) are popped
#ifdef SRT_STATS
                            if (SRT_STATS == 1) {
                                SRT_STAT(4, 1);
                                SRT_STAT(5, takeL);
                            }
#endif
                            const int slotL = lane >> logL, subL = lane & ((1 << logL) - 1);
                            const bool onL = slotL < takeL;
                            const unsigned item = onL ? qlt[-(nL + slotL)] : 0u;
                            SRT_SEGL(0, item);  // the pop
                            const int src = (int)(item >> 26), code = (int)(item & 0x3FFFFFFu);
                            const int lcnt = (code & 3) + 1;  // leaf items: (first triangle) * 4 + (count - 1)
                            const float4* rowp = P.bvh_tris + 3 * (size_t)((code >> 2) + (subL < lcnt ? subL : 0));  // (idle lanes: triangle 0)
                            float4 a = rowp[0], b = rowp[1], c = rowp[2];
                            const V3 ro = v3(__shfl(o.x, src), __shfl(o.y, src), __shfl(o.z, src));
                            const V3 rd = v3(__shfl(d.x, src), __shfl(d.y, src), __shfl(d.z, src));
                            SRT_SEGL(1, ((a.x + b.x) + c.x) + (((ro.x + ro.y) + ro.z) + ((rd.x + rd.y) + rd.z)));  // the first trip's rows, the ray
                            for (int j = 0;; ++j) {
                                const int k = subL + (j << logL);  // this trip's triangle of the leaf
                                const bool more = j + 1 < trips;   // (wave-uniform)
                                float4 na = a, nb = b, nc = c;
                                if (more) {  // the next trip's rows are requested before this trip's arithmetic
                                    const int kn = subL + ((j + 1) << logL);
                                    const float4* np = P.bvh_tris + 3 * (size_t)((code >> 2) + (kn < lcnt ? kn : 0));
                                    na = np[0], nb = np[1], nc = np[2];
                                }
                                // Moller-Trumbore, binary32, no FMA, fixed order (the oracle's triangle_raytrace)
                                V3 pv = v3(rd.y * c.z - rd.z * c.y, rd.z * c.x - rd.x * c.z, rd.x * c.y - rd.y * c.x);
                                float det = (b.x * pv.x + b.y * pv.y) + b.z * pv.z;
                                float idet = 1.0f / det;
                                V3 tv = v3(ro.x - a.x, ro.y - a.y, ro.z - a.z);
                                float u = ((tv.x * pv.x + tv.y * pv.y) + tv.z * pv.z) * idet;
                                V3 qv = v3(tv.y * b.z - tv.z * b.y, tv.z * b.x - tv.x * b.z, tv.x * b.y - tv.y * b.x);
                                float vv = ((rd.x * qv.x + rd.y * qv.y) + rd.z * qv.z) * idet;
                                float t = ((c.x * qv.x + c.y * qv.y) + c.z * qv.z) * idet;
                                bool ok = onL & (k < lcnt) & (fabsf(det) >= 1e-12f) & (u >= 0.0f) & (u <= 1.0f) & (vv >= 0.0f) & (u + vv <= 1.0f) &
                                          (t >= (float)0.01) & (t <= 10000.0f);
#ifdef SRT_DEV
                                if (P.flags & 0x200u) ok = false;
#endif
                                // the merge key is (ordered t || global triangle id): the id order is (list order of the object,
                                // triangle index), i.e. the tie rule; atomicMin makes the merge order-independent
                                if (ok) atomicMin(&S.res[src], ((unsigned long long)okey(t) << 32) | (unsigned)__float_as_int(b.w));
#if defined(SRT_STATS) && SRT_STATS == 8
                                if (j == 0) SRT_SEGL(2, t);  // the first trip's arithmetic and merge
#endif
                                if (!more) break;
                                a = na, b = nb, c = nc;
                            }
                            SRT_SEG(5, a.x);  // a whole leaf round
                            SRT_SEGL(3, a.x);  // the later trips
#if defined(SRT_STATS) && (SRT_STATS == 7 || SRT_STATS == 8)
                            seg_acc[7] += 1;
#endif
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                    if (!overflow) {
                        pend &= ~selmask;
                    } else if (batch > 1) {
                        batch = batch > 4 ? batch >> 2 : 1;
                    } else {
                        strict = true;  // cannot overflow: the host bounds 7 * depth + 80 (the stack + the leaves that may wait) by MESH_Q
#ifdef SRT_STATS
                        if (SRT_STATS == 1) SRT_STAT(7, 1);
#endif
                    }
#ifdef SRT_STATS
                    if (SRT_STATS == 1 && overflow) SRT_STAT(6, 1);
#endif
                }
#if defined(SRT_STATS) && (SRT_STATS == 7 || SRT_STATS == 8)
                {
                    const int lane_ = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
                    const unsigned row_ = ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4u + (threadIdx.x >> 6);
                    const long long v_ = lane_ == 0 ? seg_acc[0] : lane_ == 1 ? seg_acc[1] : lane_ == 2 ? seg_acc[2] : lane_ == 3 ? seg_acc[3] : lane_ == 4 ? seg_acc[4] :
                                         lane_ == 5 ? seg_acc[5] : lane_ == 6 ? seg_acc[6] : seg_acc[7];
                    if (lane_ < 8) atomicAdd(&g_seg[8u * (row_ & (unsigned)(SEG_ROWS - 1)) + (unsigned)lane_], (unsigned long long)v_);
                }
#endif
#if defined(SRT_STATS) && SRT_STATS == 5
                SRT_STAT(0, (long long)__builtin_readcyclecounter() - st_t0);
                SRT_STAT(4, st_rounds);
                SRT_STAT(6, 1);
#endif
#ifdef SRT_STATS
                if (SRT_STATS == 2) {  // histogram by the number of rays entering the phase: phases / rounds per class
                    SRT_STAT(st_cls, 1);
                    SRT_STAT(4 + st_cls, st_rounds);
                }
#endif
                {  // the ray's best triangle, if any, against the best analytic hit
                    const unsigned long long r = S.res[lane];
                    const unsigned gid = (unsigned)r;
                    if (go && gid != 0xFFFFFFFFu) {
                        const float tm = unkey((unsigned)(r >> 32));
                        const int pos = P.bvh_gidpos[gid];
                        // the triangle's primitive id, edges (for the normal) and list index in ONE round trip
                        const float pw = P.bvh_tris[3 * pos].w;
                        const float4 e1 = P.bvh_tris[3 * pos + 1], e2 = P.bvh_tris[3 * pos + 2];
                        const bool win = (tm < best) | ((tm == best) & (__float_as_int(e2.w) < bord));
                        best = win ? tm : best;
                        bp = win ? __float_as_int(pw) : bp;
                        btri = win ? pos : btri;
                        if (win) tri_n = v3(e1.y * e2.z - e1.z * e2.y, e1.z * e2.x - e1.x * e2.z, e1.x * e2.y - e1.y * e2.x);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    h.t = best;
    h.prim = bp;
    h.p = v3(o.x + d.x * best, o.y + d.y * best, o.z + d.z * best);  // Object.hpp:136 / :229
    if (MESH && btri >= 0) {
        V3 n = normalized(tri_n);  // unit geometric normal, e1 x e2
        if (dot3(n, d) > 0) n = v3(n.x * -1, n.y * -1, n.z * -1);                                      // turned against the ray
        h.n = n;
    } else if (bp >= nsT) {
        h.n = ibox_normal(br, bt1);
    } else if (bp >= 0) {
        const float4 s = S.sphere(bp);
        h.n = normalized(v3(h.p.x - s.x, h.p.y - s.y, h.p.z - s.z));  // Object.hpp:137
    } else {
        h.n = v3(0, 0, 0);
    }
    SRT_TICK(6);
    return h;
}

struct RGB {
    float r, g, b;
};
// Color::Lerp (Common.hpp:275-279), rgb part
__device__ __forceinline__ RGB color_lerp(RGB a, RGB b, float t) {
    return RGB{clamp0(a.r * (1 - t) + b.r * t), clamp0(a.g * (1 - t) + b.g * t), clamp0(a.b * (1 - t) + b.b * t)};
}

// GetEnvironmentColor (Raytracer.cpp:77-89).  The four colours arrive clamped (Color's constructor, Common.hpp:253-262) in
// the image's constants block, next to srt_powf's table: LDS reads, no kernel-argument registers, no constant-memory trips.
__device__ __forceinline__ RGB environment(const Lds& S, V3 d) {
    const float4 e0 = S.c[CONST_ENV_ROW], e1 = S.c[CONST_ENV_ROW + 1], e2 = S.c[CONST_ENV_ROW + 2], e3 = S.c[CONST_ENV_ROW + 3];
    RGB Sky{e0.x, e0.y, e0.z};
    RGB Horizon{e1.x, e1.y, e1.z};
    RGB Ground{e2.x, e2.y, e2.z};
    float upd = (d.x * 0.0f + d.y * 1.0f) + d.z * 0.0f;  // Dot(rayDirection, WORLDUP) :78
    float sd = (d.x * (e3.x * -1) + d.y * (e3.y * -1)) + d.z * (e3.z * -1);
    // :79 compares the float with the DOUBLE literal 0.99, which lies between the floats 0x3F7D70A3 (0.98999995) and 0x3F7D70A4
    // (0.99000001): (double)sd > 0.99 exactly when sd >= 0x3F7D70A4 (a NaN fails both) — one float compare, no conversion
    bool sunny = sd >= __uint_as_float(0x3F7D70A4u);
    RGB Sun{sunny ? e0.w : 0.0f, sunny ? e1.w : 0.0f, sunny ? e2.w : 0.0f};
    // one srt_powf call site for both branches (:81 powf(upd, 0.1f) / :87 powf(|upd|, .05f)):
    // the arguments are selected per lane, so up- and down-going lanes do not serialize.
    const bool up = upd > 0;
    const float pw = srt_powf_tab(up ? upd : fabsf(upd), up ? 0.1f : .05f, reinterpret_cast<const double*>(S.c), reinterpret_cast<const double*>(S.c + CONST_COEF_ROW));
    RGB t;
    if (up) {
        t = color_lerp(Horizon, Sky, pw);  // :81
        RGB sky01{Sky.r * 0.1f, Sky.g * 0.1f, Sky.b * 0.1f};  // (NN)
        t = color_lerp(t, sky01, upd);     // :82
    } else {
        t = color_lerp(Horizon, Ground, pw);  // :86-87
    }
    return RGB{t.r + Sun.r, t.g + Sun.g, t.b + Sun.b};  // :83 / :87 (NN: t comes out of Color::Lerp's clamp)
}

// (int)f with x86 cvttss2si semantics (reference platform), see oracle cvtt_x86
__device__ __forceinline__ int cvtt_x86(float f) {
    if (f != f || f >= 2147483648.0f || f < -2147483648.0f) return (int)0x80000000;
    return (int)f;
}
__device__ __forceinline__ uint32_t pack_channel(float v) {  // Common.hpp:190-203
    int s = cvtt_x86(v * 255);
    if (s > 255) s = 255;
    return (uint32_t)(uint8_t)s;
}

// SetScreenPixel accumulate half (Raytracer.cpp:65-71) for sample index sidx (0-based in
// this launch); the colour's alpha is always +0
__device__ __forceinline__ void accumulate_sample(const KernelParams& P, float4& acc, RGB c, uint32_t sidx) {
    const uint32_t frame = P.first_sample + sidx;
    if (sidx == 0 && (P.flags & 1u) != 0) {
        acc = make_float4(c.r, c.g, c.b, 0.0f);
    } else {
        // :66  float weight = 1.0 / ACCUMULATIONFRAMES (double divide, rounded once).  For
        // frame <= 2^24 the float divide gives the same bits (1/f cannot sit within 2^-53
        // of a binary32 rounding boundary unless it is exact; checked exhaustively in tests).
        // (1 / an integer in [1, 2^24]: inside div_window's window; checked against the division for every one of them)
        float weight = frame <= 16777216u ? div_window(1.0f, (float)(int)frame) : (float)(1.0 / (double)(int)frame);
        float om = 1 - weight;
        // :67; the accumulator may come from the caller (srt_write_accumulator): its product keeps the clamp, the rest is NN
        acc.x = clamp0(acc.x * om) + c.r * weight;
        acc.y = clamp0(acc.y * om) + c.g * weight;
        acc.z = clamp0(acc.z * om) + c.b * weight;
        acc.w = clamp0(acc.w * om) + 0.0f * weight;
    }
}
// SetScreenPixel tone-map + pack + the two stores (Raytracer.cpp:64,73-75)
__device__ __forceinline__ uint32_t tone_map(const float4 acc) {
    // (every accumulator that gets here has been through accumulate_sample or is a sample colour: NN)
    float r = acc.x / (1.0f + acc.x);
    float g = acc.y / (1.0f + acc.y);
    float b = acc.z / (1.0f + acc.z);
    float a = acc.w / (0.0f + acc.w);
    return pack_channel(a) << 24 | pack_channel(r) << 16 | pack_channel(g) << 8 | pack_channel(b);
}
__device__ __forceinline__ void store_pixel(const KernelParams& P, uint32_t pix, const float4 acc) {
    P.accumulator[pix] = acc;
    const uint32_t px = tone_map(acc);
    const uint32_t py = pix / (uint32_t)P.width, pxx = pix - py * (uint32_t)P.width;
    P.framebuffer[(size_t)(P.height - 1 - (int)py) * P.width + pxx] = px;
}

// Per-wave view of the workgroup's LDS: [scene image (SCENE_LDS only)] [waves x WAVE_SCRATCH_BYTES]
// [waves x MESH_WAVE_BYTES (mesh kernel only)].  SCENE_LDS == false is the fallback for scene images
// that do not fit next to the scratch (thousands of analytic primitives): the same image is then read
// from HBM/L2 through the same accessors — slower per test, same arithmetic, same bits.
template <bool SCENE_LDS>
__device__ __forceinline__ Lds make_lds(const KernelParams& P, float4* lds, int waves, int wave) {
    constexpr int stride = WAVE_SCRATCH_BYTES;
    char* wg = reinterpret_cast<char*>(lds + (SCENE_LDS ? P.scene_vec4 : 0));
    char* scratch = wg + wave * stride;
    const float4* image;
    if constexpr (SCENE_LDS)
        image = lds;
    else
        image = P.scene;
    return Lds{image, image + CONST_ROWS, P.nu4, P.nu, P.nc, P.K, P.nsT, P.nb, P.off_bounds, P.off_box, P.off_mat,
               reinterpret_cast<unsigned long long*>(scratch), reinterpret_cast<unsigned short*>(scratch + 64 * 8),
               reinterpret_cast<float*>(scratch + 64 * 8 + WORK_MAX * 2),
               reinterpret_cast<float4*>(scratch + 64 * 8 + WORK_MAX * 2 + 64 * 12 * 4),
               reinterpret_cast<unsigned*>(wg + waves * stride + wave * MESH_WAVE_BYTES)};
}

// MULTI: the path pool may hand several samples of one pixel out at once (used with small tiles, where
// there are fewer pixels than lanes).  A separate instantiation because the extra live values cost the
// full-tile kernel, which sits exactly at its 128-VGPR budget, four spilled registers and 3 % of its speed.
// DEFER: sample-chunked launch (P.chunk > 0, grid z = chunk index) for narrow row bands at high sample
// counts, where one workgroup per tile for ALL samples would leave too few, too long workgroups.  The
// running mean is order-dependent (Raytracer.cpp:67), so chunks cannot fold on their own: the owner lanes
// store the colours, in sample order, as coalesced rows of P.sample_rows and fold_kernel does the fold.
// PROBE: the balance probe of srt_estimate_row_costs — the same pool over the frame's first few samples, nothing read from or
// written to the frame; instead every wave adds its Tally to the TALLY_N words of its block in P.wg_cost.
// TALLY: the instantiation that keeps the wave-uniform loop counts (see Tally): the recording launch of a band, and launches with
// SRT_RENDER_COUNT_WORK.  Same results; a few scalar adds per pool step and some scalar-register pressure the steady-state
// instantiations do not pay.
template <int MIN_WAVES, bool MESH, bool SCENE_LDS = true, bool MULTI = false, bool DEFER = false, bool PROBE = false, bool TALLY_ = false>
__global__ void __launch_bounds__(WG_THREADS, MIN_WAVES) pathtrace_kernel(const KernelParams P) {
    static_assert(!PROBE || (!MULTI && !DEFER), "the probe runs the full-tile pool");
    constexpr bool TALLY = TALLY_ || PROBE;
    extern __shared__ float4 lds_scene[];
    Tally<TALLY> tally{};
#if defined(SRT_STATS) && SRT_STATS == 3
    Prof prof;
    prof.last = (long long)__builtin_readcyclecounter();
    for (int i = 0; i < 8; ++i) prof.acc[i] = 0;
#endif
    if constexpr (SCENE_LDS) {  // stage the scene image into LDS (coalesced 16-byte loads)
        // All of a thread's rows are requested before the first one is stored: written as a plain copy loop this compiled to
        // load - wait - store per row, five dependent L2 round trips (3..4 us) before a workgroup of Scene1 could start —
        // 2 % of a 32-sample tile, 4 % of a 24-sample chunk, most of what an all-sky tile costs.  Eight rows per thread cover
        // 32 KB; larger images finish in the loop.
        constexpr int STAGE = 8;
        const int n = P.scene_vec4;
        float4 row[STAGE];
#pragma unroll
        for (int k = 0; k < STAGE; ++k) {
            const int i = (int)threadIdx.x + k * WG_THREADS;
            row[k] = i < n ? P.scene[i] : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < STAGE; ++k) {
            const int i = (int)threadIdx.x + k * WG_THREADS;
            if (i < n) lds_scene[i] = row[k];
        }
        for (int i = (int)threadIdx.x + STAGE * WG_THREADS; i < n; i += WG_THREADS) lds_scene[i] = P.scene[i];
        __syncthreads();
    }
    const Lds S = make_lds<SCENE_LDS>(P, lds_scene, WG_TILES_X * WG_TILES_Y, threadIdx.x >> 6);
    // the probe: ONE of the workgroup's four waves — the four hold the same mix of pixels (dealt pixel by pixel, see below), so a
    // quarter of the pixels stands for the block, and the samples go into depth instead: a pool that runs 8 samples per pixel has
    // 12..37 % more steps per sample than a launch's chunks of 32 and more (the tail of a tile), unevenly over the frame
    if (PROBE && (threadIdx.x >> 6) != 0) return;

    // ---- pixel of this lane ----------------------------------------------------------
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // a wave holds TILE_W x P.tile_h pixels (tile_h = 8, or less when the launch would otherwise have
    // too few workgroups to fill the chip: narrow row bands at high sample counts).  Lanes without a pixel
    // still work in the path pool.
    const int tile_h = P.tile_h;
    // which block of tiles: the hardware starts workgroups in linear order, wg_order maps that order to
    // blocks sorted by decreasing cost so that the expensive ones do not end up in the tail
#if defined(SRT_STATS) && SRT_STATS == 6
    const long long t_start = (long long)__builtin_readcyclecounter();
    const unsigned long long r_start = __builtin_amdgcn_s_memrealtime();
    const unsigned hw_start = __builtin_amdgcn_s_getreg(0xF804), xcc_start = __builtin_amdgcn_s_getreg(0xF814);  // HW_ID, XCC_ID
#elif defined(SRT_STATS) && SRT_STATS == 5
    const long long t_start = (long long)__builtin_readcyclecounter();
#else
    // block costs are timed with s_memrealtime, the constant 100 MHz clock all XCDs share — NOT with s_memtime
    // (__builtin_readcyclecounter): that one is a per-XCD shader-clock counter (tools/timer_probe.py: the eight XCDs' values lie
    // 4e10 .. 1.8e11 ticks apart), and a difference of two of its readings was seen to come out negative on some boxes
    const long long t_start = !PROBE && P.wg_cost ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
#endif
    uint32_t block_id = blockIdx.y * gridDim.x + blockIdx.x;
    if (P.wg_order) block_id = P.wg_order[block_id];
    const int bx = (int)(block_id % gridDim.x), by = (int)(block_id / gridDim.x);
    // The workgroup's 16 x (2 * tile_h) pixels are dealt to its four waves pixel by pixel (wave = (x & 1) + 2 * (y & 1)), not
    // as four 8 x tile_h quadrants: every wave then holds the same mix of sky, floor and objects, the four finish together,
    // and no wave slot idles while the workgroup's slowest quadrant is still running (the workgroup's LDS is held until
    // then).  Measured against quadrants on one box: Scene1 2.95 -> 2.73 ms, config 4 10.75 -> 10.36 ms, Scene3 -2 %,
    // Scene_indirect -1.2 %; same bits.
    const int tx = bx * WG_W + (lane & (TILE_W - 1)) * WG_TILES_X + (wave % WG_TILES_X);
    const int ty = by * (tile_h * WG_TILES_Y) + (lane / TILE_W) * WG_TILES_Y + (wave / WG_TILES_X);
    const int W = P.width, H = P.height;
    // ---- progressive blocks (Raytracer.cpp:235-248): the ray of a steps x steps block goes
    // through the block's anchor pixel; blocks start at the worker stripe's first column.
    // Block grid (MULTI instantiation, KF_BLOCK_GRID): the launch's lanes are blocks — column tx of the grid is block
    // tx % n of stripe tx / n (n = blocks per stripe), row ty the ty-th block row that meets the band — so the launch does
    // work in proportion to the number of blocks, like renderArea: one RaytraceScene per block, steps^2 SetScreenPixels.
    const bool bgrid = MULTI && (P.flags & KF_BLOCK_GRID) != 0;
    bool in_range_;
    int x_, y_, ax, ay;
    if (bgrid) {
        const int sw = P.stripe_width > 0 ? P.stripe_width : W;
        const int per_stripe = (sw + P.steps - 1) / P.steps;
        const int stripe = tx / per_stripe;
        ax = stripe * sw + (tx - stripe * per_stripe) * P.steps;
        ay = (P.y0 / P.steps + ty) * P.steps;
        in_range_ = tx < P.bgrid_w && ty < P.bgrid_h && ax < W && ay < P.y0 + P.rows && (lane / TILE_W) < tile_h;
        if (!in_range_) ax = 0, ay = (P.y0 / P.steps) * P.steps;
        x_ = ax, y_ = ay;
    } else {
        in_range_ = tx < W && ty < P.rows && (lane / TILE_W) < tile_h;
        x_ = in_range_ ? tx : 0, y_ = in_range_ ? (P.y0 + ty) : P.y0;
        ax = x_, ay = y_;
        if (P.steps > 1) {
            const int s0 = P.stripe_width > 0 ? (x_ / P.stripe_width) * P.stripe_width : 0;
            ax = s0 + ((x_ - s0) / P.steps) * P.steps;
            ay = (y_ / P.steps) * P.steps;
        }
    }
    const bool in_range = in_range_;
    const int x = x_, y = y_;
    const uint32_t pixel = (uint32_t)(x + y * W);  // (block grid: the anchor; it may lie above the band)
    const uint32_t rng_pixel = (uint32_t)(ax + ay * W);
    // Block grid: the pixels of the wave's finished blocks are written by ALL lanes together — lane groups of steps^2 lanes
    // take a block each (steps = 8: a block per pass, eight rows of eight neighbouring pixels) — instead of every block's
    // lane looping over its own steps^2 pixels.  `val` is the block's running mean (the launch starts the frame: the same
    // for all its pixels, tone-mapped once) or its ONE sample colour (`keep`: every pixel folds it into its own mean).
    // A block's pixels: inside its stripe, the image and the band.
    auto write_blocks = [&](bool have, int bx0, int by0, float4 val, bool keep) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(have);
        if (m == 0ull) return;
        const int n = __builtin_popcountll(m);
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        float4* sv = S.ring;                                      // [64] values
        unsigned* sa = reinterpret_cast<unsigned*>(S.work);       // [64] anchor x | [64] anchor y | [64] tone-mapped pixel
        __builtin_amdgcn_wave_barrier();
        if (have) {
            sv[rank] = val;
            sa[rank] = (unsigned)bx0, sa[64 + rank] = (unsigned)by0;
            sa[128 + rank] = keep ? 0u : tone_map(val);
        }
        __builtin_amdgcn_wave_barrier();
        const int steps = P.steps, per = steps * steps;
        const int L = per < 64 ? per : 64, G = 64 / L;  // lanes per block, blocks per pass
        const int g = lane / L, ql = lane - g * L;
        for (int b0 = 0; b0 < n; b0 += G) {
            const int b = b0 + g;
            const bool on = g < G && b < n;
            const int bi = on ? b : 0;
            const int x0 = (int)sa[bi], y0b = (int)sa[64 + bi];
            const float4 v = sv[bi];
            const uint32_t px = sa[128 + bi];
            int x_end = x0 + steps;
            if (P.stripe_width > 0) {
                const int s_end = (x0 / P.stripe_width + 1) * P.stripe_width;
                x_end = x_end < s_end ? x_end : s_end;
            }
            x_end = x_end < W ? x_end : W;
            int y_hi = y0b + steps;
            y_hi = y_hi < P.y0 + P.rows ? y_hi : P.y0 + P.rows;
            for (int q = ql; q < per; q += 64) {  // (one trip for steps <= 8)
                const int j = q / steps, i = q - j * steps;
                const int xx = x0 + i, yy = y0b + j;
                if (on && xx < x_end && yy >= P.y0 && yy < y_hi) {
                    const uint32_t pp = (uint32_t)(xx + yy * W);
                    if (keep) {
                        float4 a = P.accumulator[pp];
                        accumulate_sample(P, a, RGB{v.x, v.y, v.z}, 0);
                        P.accumulator[pp] = a;
                        P.framebuffer[(size_t)(H - 1 - yy) * W + xx] = tone_map(a);
                    } else {
                        P.accumulator[pp] = v;
                        P.framebuffer[(size_t)(H - 1 - yy) * W + xx] = px;
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
    // Blocks are traced ONCE per wave tile, by the first lane of the tile that lies in the block (its leader), and
    // every pixel of the block folds the leader's sample colours into its own running mean (renderArea :239-248 does
    // exactly that: one RaytraceScene per block, one SetScreenPixel per pixel).  A block that straddles tiles is traced
    // once in each.  Launches with steps > 1 use the MULTI instantiation (few slots per tile, several samples each).
    int lead_lane = lane;
    if constexpr (MULTI) {
        if (P.steps > 1 && !(P.flags & 4u) && !bgrid) {  // (the preview shader traces nothing: every pixel shades for itself)
            const uint32_t key = in_range ? rng_pixel : 0xFFFFFFFFu - (uint32_t)lane;  // lanes without a pixel lead themselves
            unsigned long long same = 0ull;
            for (int i = 0; i < 64; ++i)
                if ((uint32_t)__builtin_amdgcn_readlane((int)key, i) == key) same |= 1ull << i;
            lead_lane = __builtin_ctzll(same);
        }
    }
    const bool is_leader = lead_lane == lane;

    // ---- GetRayDirection (Raytracer.cpp:106-122) ----------------------------------
    float nX = ((float)ax / (float)W) * 2 - 1;
    float nY = ((float)ay / (float)H) * 2 - 1;
    V3 u = v3(P.right_rd[0] * nX, P.right_rd[1] * nX, P.right_rd[2] * nX);
    V3 vv = v3(P.up_ld[0] * nY, P.up_ld[1] * nY, P.up_ld[2] * nY);
    const V3 dir0 = normalized(v3((u.x + vv.x) + P.fwd_clip[0], (u.y + vv.y) + P.fwd_clip[1], (u.z + vv.z) + P.fwd_clip[2]));
    const V3 cam = v3(P.cam_pos[0], P.cam_pos[1], P.cam_pos[2]);

    // ---- primary hit: identical for every sample ------------------------------------
    bool parked = false;  // path pool: this lane's ray waits for a mesh phase (see closest_hit)
    const Hit h0 = closest_hit<MESH, TALLY, SCENE_LDS>(S, P, cam, dir0, true, 1, parked, tally SRT_PROF_ARG);

    const bool reset = (P.flags & 1u) != 0;
    // samples of this workgroup: all of them, or chunk blockIdx.z of the launch
    // (the last layers of the grid — the last workgroups to start — trace half chunks: the launch's tail is as long as its
    // last workgroups run)
    uint32_t s_base = 0u, count = P.sample_count;
    if constexpr (DEFER) {
        const uint32_t z = blockIdx.z, zf = (uint32_t)P.chunk_full, S = (uint32_t)P.chunk, half = S >> 1;
        s_base = z < zf ? z * S : zf * S + (z - zf) * half;
        const uint32_t want = z < zf ? S : half;
        count = P.sample_count - s_base < want ? P.sample_count - s_base : want;
    }
    const int B = P.max_bounces;
    unsigned rays = 0;

    auto accumulate = [&](float4& acc, RGB c, uint32_t sidx) { accumulate_sample(P, acc, c, sidx); };
    auto write_pixel = [&](uint32_t pix, const float4 acc) { store_pixel(P, pix, acc); };

    // ---- pixels whose colour does not depend on the sample: finish them right here --------
    //  primary miss -> env(dir0) every frame (:143-145); MAXBOUNCES == 0 -> EmissiveColor (:162,212)
    //  SIMPLEDRAW -> the one-ray preview shader (:147-160), no random draws
    const bool preview = (P.flags & 4u) != 0;
    const bool pix_traced = in_range && h0.prim >= 0 && B > 0 && !preview;  // (the same for every pixel of a block)
    const bool traced = pix_traced && is_leader;
    float4 untraced_val = make_float4(0, 0, 0, 0);
    tally.add(TALLY_UNTRACED_WAVES, __builtin_amdgcn_ballot_w64(in_range && !pix_traced) != 0ull ? 1u : 0u);
    tally.add(TALLY_WAVES, 1u);
    if (!PROBE && in_range && !pix_traced && (!DEFER || blockIdx.z == 0)) {  // (chunked: once, by the first chunk, for all samples)
        RGB c;
        if (h0.prim < 0) {
            c = environment(S, dir0);
        } else if (preview) {
            float4 m0 = S.mat(h0.prim, 0), m1 = S.mat(h0.prim, 1);
            float k2 = 2 * dot3(dir0, h0.n);  // rayDirection.Reflect(normal), Common.hpp:163-165
            RGB refl = environment(S, v3(dir0.x - h0.n.x * k2, dir0.y - h0.n.y * k2, dir0.z - h0.n.z * k2));  // :148
            const float k = m0.y, sm = m0.x;  // :149-150
            float fresnal = 0;
            if (S.order(h0.prim) == P.selected) {  // :153
                fresnal = 1 - dot3(v3(h0.n.x * -1, h0.n.y * -1, h0.n.z * -1), dir0);  // :154
                fresnal = tmax(fresnal, 0.0f);                                       // :155
                if (fresnal < 0.0f) {                                                // smoothstep, Common.hpp:352-365
                    fresnal = 0;
                } else if (fresnal >= 0.5f) {
                    fresnal = 1;
                } else {
                    fresnal = (fresnal - 0.0f) / (0.5f - 0.0f);
                    fresnal = fresnal * fresnal * (3 - 2 * fresnal);
                }
            }
            RGB base{clamp0(m0.z), clamp0(m0.w), clamp0(m1.x)}, emis{clamp0(m1.y), clamp0(m1.z), clamp0(m1.w)};
            const float omk = 1 - k;
            // BaseColor * (1-k) + reflectedColor * k * s + EmissiveColor   (:159)
            RGB a{clamp0(clamp0(clamp0(base.r * omk) + clamp0(clamp0(refl.r * k) * sm)) + emis.r),
                  clamp0(clamp0(clamp0(base.g * omk) + clamp0(clamp0(refl.g * k) * sm)) + emis.g),
                  clamp0(clamp0(clamp0(base.b * omk) + clamp0(clamp0(refl.b * k) * sm)) + emis.b)};
            c = color_lerp(a, RGB{3.0f, 3.0f, 0.0f}, fresnal);
        } else {
            float4 m1 = S.mat(h0.prim, 1);
            c = RGB{clamp0(m1.y), clamp0(m1.z), clamp0(m1.w)};
        }
        if (bgrid) {  // (written below, by the whole wave)
            untraced_val = make_float4(c.r, c.g, c.b, 0.0f);
            if (reset) {
                untraced_val = make_float4(0, 0, 0, 0);
                for (uint32_t i = 0; i < P.sample_count; ++i) accumulate(untraced_val, c, i);
            }
        } else {
            float4 acc = reset ? make_float4(0, 0, 0, 0) : P.accumulator[pixel];
            for (uint32_t i = 0; i < P.sample_count; ++i) accumulate(acc, c, i);
            write_pixel(pixel, acc);
        }
        if (is_leader) rays += P.sample_count;  // one GetClosestObject call per block and frame
    }
    if constexpr (MULTI) {
        if (bgrid) write_blocks(in_range && !pix_traced, ax, ay, untraced_val, !reset);
    }

    // ---- wave-level path pool -----------------------------------------------------------------
    // The pixels that need tracing are compacted into slots 0..n_hit-1 (ballot + mbcnt).  A task
    // is (slot, sample); tasks are handed out in sample-major order to whichever lane is free,
    // so every lane stays busy regardless of how hit pixels and path lengths are distributed
    // over the tile.  A finished sample colour goes into a small LDS ring; lane k owns slot k
    // and folds the ring into the running mean strictly in sample order (the mean is
    // order-dependent, Raytracer.cpp:67), then stores the pixel at the end.
    const unsigned long long hitmask = __builtin_amdgcn_ballot_w64(traced);
    const size_t tile_id = (size_t)block_id * (WG_TILES_X * WG_TILES_Y) + wave;
    if (DEFER && blockIdx.z == 0 && lane == 0) P.tile_masks[tile_id] = hitmask;
    const int n_hit = __builtin_popcountll(hitmask);
    if (n_hit > 0) {
        float* rec = S.pix;  // [64][12]: dir0, n0, p0, prim0, rng pixel (block anchor), output pixel
        if (traced) {
            const int slot = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(hitmask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hitmask, 0u));
            float* r = rec + slot * 12;
            r[0] = dir0.x, r[1] = dir0.y, r[2] = dir0.z;
            r[3] = h0.n.x, r[4] = h0.n.y, r[5] = h0.n.z;
            r[6] = h0.p.x, r[7] = h0.p.y, r[8] = h0.p.z;
            r[9] = __int_as_float(h0.prim);
            // srt_rng_key's first two rounds depend on seed and pixel only: done here once, the sample's round at the hand-out
            r[10] = __uint_as_float(srt_mix32(srt_mix32(P.seed ^ 0xA511E9B3U) + rng_pixel));
            r[11] = __uint_as_float(pixel);
        }
        constexpr int ring_depth = RING_DEPTH;
        const int depth = (64 * ring_depth) / n_hit;  // ring entries per slot (>= ring_depth)
        // sample -> ring row: a mask when depth is a power of two (full tiles: 4), else a real modulo (~20 instructions)
        const bool depth_pow2 = (depth & (depth - 1)) == 0;
        auto ring_row = [&](uint32_t s) { return depth_pow2 ? (s & (uint32_t)(depth - 1)) : (s % (uint32_t)depth); };
        float4* ring = S.ring;                        // [depth][n_hit] of (r, g, b, tag)
        for (int i = lane; i < 64 * ring_depth; i += 64) ring[i] = make_float4(0, 0, 0, __uint_as_float(0xFFFFFFFFu));
        __builtin_amdgcn_wave_barrier();

        // owner state (lane k owns slot k)
        const bool owner = lane < n_hit;
        uint32_t own_pixel = 0, own_done = owner ? 0u : count;  // samples folded so far
        float4 acc = make_float4(0, 0, 0, 0);
        // progressive blocks: a slot is a block, and every PIXEL lane of the block folds the slot's colours into the
        // running mean of its own pixel, in step with the slot's owner (same ring entries, same test, same iteration)
        const bool blocks = MULTI && P.steps > 1 && !(P.flags & 4u) && !bgrid;
        // block grid: the owner folds ONE running mean for the whole block when the launch starts the frame (all pixels then
        // hold the same value); when it continues a frame the launch has one sample (srt_render), the owner keeps its
        // colour and every pixel of the block folds it into its own accumulator at the end
        const bool bgrid_keep = bgrid && !reset;
        const bool fold_px = blocks && pix_traced;
        const int fslot = __builtin_popcountll(hitmask & ((1ull << lead_lane) - 1ull));  // the leader's slot
        uint32_t fdone = fold_px ? 0u : count;
        if (blocks) {
            if (fold_px && !reset) acc = P.accumulator[pixel];
        } else if (owner) {
            own_pixel = __float_as_uint(rec[lane * 12 + 11]);
            if (!DEFER && !reset && !bgrid) acc = P.accumulator[own_pixel];
        }
        // chained chunks (KernelParams.tile_chain): does this tile's running mean stand at this chunk's first sample?
        bool chained = false;  // (wave-uniform)
        if constexpr (DEFER) {
            if (P.tile_chain) {
                const uint32_t at = blockIdx.z == 0 ? 0u : (uint32_t)__builtin_amdgcn_readfirstlane((int)coherent_load(P.tile_chain + tile_id));
                chained = at == blockIdx.z;
                if (chained && owner && (blockIdx.z > 0 || !reset)) acc = coherent_load(P.accumulator + own_pixel);
            }
        }

        // path state of the task this lane is running
        bool busy = false;
        uint32_t task = 0;  // sample << 6 | slot (one register: the kernel sits right at its VGPR budget)
        RGB L{0, 0, 0}, T{0, 0, 0};
        V3 sray = v3(0, 0, 1), hn = v3(0, 0, 0), hp = v3(0, 0, 0);
        int hprim = 0, bounce = 0;
        float spec = 0.0f;
        uint32_t rng = 0;

        // per-slot hand-out state, kept by the owner lane: samples [0, own_next) have been
        // handed out, [0, own_done) folded.  A slot may run `depth` samples ahead of its own
        // fold point (ring capacity); slots do not wait for each other.
        uint32_t own_next = own_done;
        int rot = 0;  // wave-uniform rotation of the slot priority
        unsigned handed = 0;  // wave-uniform: samples handed out so far (n_hit x the mean own_next)
        const int fold_pace = (63 + n_hit) / n_hit;
        int park_steps = 0;  // (mesh kernel) consecutive steps in which some ray waited for a mesh phase

        while (true) {
            SRT_TICK(7);
            // ---- fold finished samples, in order, into the owners' running means.  A slot finishes at most as many samples per
            // step as lanes work for it (64 / n_hit on average): that many iterations keep pace, and the few lanes that have a
            // second sample ready (finished out of order) wait for the next step, where they share the iteration with many
            // others — a further iteration for their sake costs the whole wave a fold.  Once the hand-out has ended, or nothing
            // runs, everything ready is folded.
            const bool drain = __builtin_amdgcn_ballot_w64(busy) == 0ull || __builtin_amdgcn_ballot_w64(own_next < count) == 0ull;
            // (Round 3, with the ring of two entries: folding eagerly — every step, up to `depth` iterations, no waiting for a
            // fifth of the slots — frees ring capacity sooner but costs more folds than it gains: +2..+7 %.)
            const int fold_its = drain || fold_pace > depth ? depth : fold_pace;
            for (int it = 0; it < fold_its; ++it) {
                bool ready = false;
                float4 e = make_float4(0, 0, 0, 0);
                if (own_done < count) {
                    e = ring[ring_row(own_done) * n_hit + lane];
                    ready = __float_as_uint(e.w) == own_done;
                }
                if constexpr (MULTI) {
                    if (blocks) {  // the pixels of a block fold; the owner only advances the slot's fold point
                        bool ready_px = false;
                        float4 f = make_float4(0, 0, 0, 0);
                        if (fdone < count) {
                            f = ring[ring_row(fdone) * n_hit + fslot];
                            ready_px = __float_as_uint(f.w) == fdone;
                        }
                        if (__builtin_amdgcn_ballot_w64(ready | ready_px) == 0ull) break;
                        if (ready_px) {
                            accumulate(acc, RGB{f.x, f.y, f.z}, fdone);
                            ++fdone;
                        }
                        if (ready) ++own_done;
                        continue;
                    }
                }
                const unsigned long long readym = __builtin_amdgcn_ballot_w64(ready);
                if (readym == 0ull) break;
#if defined(SRT_STATS) && SRT_STATS == 4
                {
                    const int st_ready = __builtin_popcountll(readym);
                    SRT_STAT(4, 1);
                    SRT_STAT(5, st_ready);
                }
#endif
                // long paths (Scene_indirect: 8 rays per sample) finish a few samples per step: a fold for fewer than a fifth of
                // the slots waits for the next step's, unless a slot is out of ring capacity or the hand-out has ended
                if (it == 0 && !drain && __builtin_popcountll(readym) * 5 < n_hit &&
                    __builtin_amdgcn_ballot_w64(own_next < count && own_next >= own_done + (uint32_t)depth) == 0ull)
                    break;
                if (ready) {
                    if (DEFER && chained)  // this chunk is the tile's next one: straight into the running mean
                        accumulate(acc, RGB{e.x, e.y, e.z}, s_base + own_done);
                    else if constexpr (DEFER)  // row (tile, sample) of the sample buffer: 64 slots of 16 B, coalesced
                        P.sample_rows[(tile_id * P.sample_count + s_base + own_done) * 64 + lane] = e;
                    else if (MULTI && bgrid_keep)
                        acc = e;
                    else
                        accumulate(acc, RGB{e.x, e.y, e.z}, own_done);
                    ++own_done;
                }
            }
            if (__builtin_amdgcn_ballot_w64(own_done < count) == 0ull) break;  // every slot finished

            SRT_TICK(0);
            // ---- hand out tasks: the i-th free lane takes the next sample of the i-th slot that has ring
            // capacity (ballot ranks on both sides, matched through S.work).  MULTI: when there are more
            // free lanes than such slots, the lanes are dealt round-robin to the slots and a slot hands out
            // as many consecutive samples as it is dealt lanes, up to its capacity — so a small tile
            // (P.tile_h) still keeps all 64 lanes busy
            // (Two passes.  Tried in round 3: up to 8 passes for tiles with fewer than 16 traced pixels, 4 below 32 — every pass
            // starts at most one sample per slot, so such a tile keeps few lanes busy — but on whole frames and on the 135-row
            // bands of config 3 it was 0.4..2 % slower: the sparse tiles are too few, the longer loop costs everyone.)
            bool fresh = false;
            for (int pass = 0; pass < 2; ++pass) {
                const unsigned long long freem = __builtin_amdgcn_ballot_w64(!busy);
                const uint32_t lim = count < own_done + (uint32_t)depth ? count : own_done + (uint32_t)depth;
                const int avail = own_next < lim ? (int)(lim - own_next) : 0;
                const bool can = avail > 0;
                const unsigned long long canm = __builtin_amdgcn_ballot_w64(can);
                if (freem == 0ull || canm == 0ull) break;
                const int nfree = __builtin_popcountll(freem), ncan = __builtin_popcountll(canm);
                // slot priority rotates every round so that all slots advance at the same pace
                // (a fixed order would starve the high slots and leave them for a thin tail)
                // — and, of the slots that can, those at or behind the wave's mean hand-out point go first: with a ring of two
                // entries a slot that fell behind stalls the tail of the tile (lanes idle while its last samples run one after
                // the other); serving the laggards first keeps the slots level, ~5 % fewer pool steps (tests/pool_stats.py)
                rot = (rot + 23) & 63;
                const unsigned long long lagm = __builtin_amdgcn_ballot_w64(can && (unsigned)__umul24(own_next, (unsigned)n_hit) <= handed);
                const unsigned long long restm = canm & ~lagm;
                // rank inside a tier, counted cyclically from lane `rot`: bits below the lane, minus the bits below `rot`,
                // plus the whole tier for the lanes that wrapped (all 32-bit: no per-lane 64-bit mask)
                const unsigned long long lowrot = (1ull << rot) - 1ull;
                const int nlag = __builtin_popcountll(lagm);
                const bool lag = (lagm >> lane) & 1ull;
                const unsigned long long tierm = lag ? lagm : restm;
                const int tier_below = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(tierm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)tierm, 0u));
                const int tier_off = lag ? -__builtin_popcountll(lagm & lowrot) : nlag - __builtin_popcountll(restm & lowrot);
                const int tier_wrap = lane < rot ? (lag ? nlag : ncan - nlag) : 0;
                const int crank = tier_below + tier_off + tier_wrap;
                handed += (unsigned)(nfree < ncan ? nfree : ncan);
                const int frank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(freem >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)freem, 0u));
                unsigned* match = reinterpret_cast<unsigned*>(S.work);  // [64] of (sample << 6 | slot), ~0 = nothing
                unsigned m = 0xFFFFFFFFu;
                if (!MULTI || ncan >= nfree) {  // one sample from each of the first nfree slots
                    if (can && crank < nfree) {
                        match[crank] = (own_next << 6) | (unsigned)lane;
                        ++own_next;
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (!busy && frank < ncan) m = match[frank];
                } else {  // more free lanes than slots: slots hand out several samples, entries may stay empty
                    if (!busy) match[frank] = 0xFFFFFFFFu;
                    __builtin_amdgcn_wave_barrier();
                    if (can) {
                        int given = 0;
                        for (int j = crank; j < nfree && given < avail; j += ncan) {
                            match[j] = ((own_next + (uint32_t)given) << 6) | (unsigned)lane;
                            ++given;
                        }
                        own_next += (uint32_t)given;
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (!busy) m = match[frank];
                }
                if (m != 0xFFFFFFFFu) {  // (the sample is set up below, once for the lanes of both passes)
                    task = m;
                    busy = true;
                    fresh = true;
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (fresh) {  // a lane that has just taken a task: Raytracer.cpp:162-166 for sample sidx of that pixel
                const uint32_t sidx = task >> 6;
                const float* r = rec + (int)(task & 63u) * 12;
                const int prim0 = __float_as_int(r[9]);
                rng = srt_mix32(__float_as_uint(r[10]) + (P.first_sample + s_base + sidx));  // = srt_rng_key(seed, pixel, sample), see the record
                uint32_t rr = srt_mix32(rng) >> 17;
                rng += 0x9E3779B9U;
                float4 m0 = S.mat(prim0, 0), m1 = S.mat(prim0, 1);
                spec = (m0.y >= rand_unit(rr)) ? 1.0f : 0.0f;  // :165
                L = RGB{m1.y, m1.z, m1.w};                                         // :162 (clamped in the image)
                T = RGB{m0.z, m0.w, m1.x};                                         // :163
                sray = v3(r[0], r[1], r[2]);                                       // :164
                hn = v3(r[3], r[4], r[5]);
                hp = v3(r[6], r[7], r[8]);
                hprim = prim0;
                bounce = 0;
                ++rays;  // the sample's primary GetClosestObject call (:142)
            }

            SRT_TICK(1);
#if defined(SRT_STATS) && SRT_STATS == 4  // make dev STATS=4: lane occupancy of the pool (tests/pool_stats.py)
            {
                const int st_busy = __builtin_popcountll(__builtin_amdgcn_ballot_w64(busy));  // (outside the macro: it runs under a lane-0 mask)
                SRT_STAT(0, 1);
                SRT_STAT(1, st_busy);
                SRT_STAT(6, n_hit);
            }
#endif
#if defined(SRT_STATS) && SRT_STATS == 9  // make dev STATS=9: pool steps by the number of lanes on a live path (tests/pool_stats.py --histogram)
            {
                const int nb_ = __builtin_popcountll(__builtin_amdgcn_ballot_w64(busy));
                SRT_STAT(nb_ <= 8 ? 0 : nb_ <= 16 ? 1 : nb_ <= 32 ? 2 : nb_ <= 48 ? 3 : nb_ <= 56 ? 4 : nb_ <= 60 ? 5 : nb_ <= 63 ? 6 : 7, 1);
            }
#endif
            tally.add(TALLY_STEPS, 1u);
            // ---- one bounce for every busy lane
            V3 o = v3(0, 0, 0);
            const float ofs = .00001f;
            if (MESH && busy && parked) {  // the ray made earlier, offered to the mesh phase again
                o = v3(hp.x + hn.x * ofs, hp.y + hn.y * ofs, hp.z + hn.z * ofs);
            } else if (busy) {
                if (bounce != 0) {  // :169-171
                    T = RGB{T.r * 0.8f, T.g * 0.8f, T.b * 0.8f};  // (NN)
                }
                // reflectedRay = sray.Reflect(normal)  (:172, Common.hpp:163-165)
                float k2 = 2 * dot3(sray, hn);
                V3 refl = v3(sray.x - hn.x * k2, sray.y - hn.y * k2, sray.z - hn.z * k2);
                // GetRandomNormalOrientedHemisphere (:90-105): exactly three draws, x then y then z
                uint32_t r0 = srt_mix32(rng) >> 17;
                uint32_t r1 = srt_mix32(rng + 0x9E3779B9U) >> 17;
                uint32_t r2 = srt_mix32(rng + 2u * 0x9E3779B9U) >> 17;
                rng += 3u * 0x9E3779B9U;
                V3 sr = v3((rand_unit(r0) - 0.5f) * 2, (rand_unit(r1) - 0.5f) * 2,
                           (rand_unit(r2) - 0.5f) * 2);
                sr = normalized_in_window(sr);  // (inside normalized()'s window by construction: see there)
                if (dot3(sr, hn) < 0) sr = v3(sr.x * -1, sr.y * -1, sr.z * -1);
                // float3::Lerp(sray, reflectedRay, Smoothness * specularProb)  (:175)
                float tt = S.mat(hprim, 0).x * spec;
                V3 l = v3(sr.x * (1 - tt) + refl.x * tt, sr.y * (1 - tt) + refl.y * tt, sr.z * (1 - tt) + refl.z * tt);
                sray = normalized(l);  // :176
                o = v3(hp.x + hn.x * ofs, hp.y + hn.y * ofs, hp.z + hn.z * ofs);  // :177
            }
            // the scan runs in wave-uniform control flow: idle lanes help with other lanes' rays
            // (a mesh phase waits for P.mesh_defer rays (12) — but not for long (P.mesh_wait = 3 steps): where mesh rays are rare a parked ray would hold its lane
            // and, through the ring, its slot for many steps; after P.mesh_wait steps with someone parked the phase runs for whoever is there)
            const Hit h = closest_hit<MESH, TALLY, SCENE_LDS>(S, P, o, sray, busy, MESH && park_steps >= P.mesh_wait ? 1 : P.mesh_defer, parked, tally SRT_PROF_ARG);
            if constexpr (MESH) park_steps = __builtin_amdgcn_ballot_w64(busy && parked) != 0ull ? park_steps + 1 : 0;
            if (busy && !(MESH && parked)) {
                ++rays;
                bool end_path;
                if (h.prim < 0) {  // :178-181
                    RGB e = environment(S, sray);
                    L = RGB{L.r + e.r * T.r, L.g + e.g * T.g, L.b + e.b * T.b};  // (NN)
                    end_path = true;
                } else {
                    float4 m0 = S.mat(h.prim, 0), m1 = S.mat(h.prim, 1), m2 = S.mat(h.prim, 2);
                    uint32_t r = srt_mix32(rng) >> 17;
                    rng += 0x9E3779B9U;
                    spec = (m0.y >= rand_unit(r)) ? 1.0f : 0.0f;  // :182
                    RGB Em{m1.y, m1.z, m1.w};
                    L = RGB{L.r + Em.r * T.r, L.g + Em.g * T.g, L.b + Em.b * T.b};  // :183 (NN)
                    RGB Bc{m0.z, m0.w, m1.x}, Sc{m2.x, m2.y, m2.z};
                    const float ns = 1 - spec;                                          // :184, Color::Lerp with t = 0 or 1 (NN)
                    RGB f{Bc.r * ns + Sc.r * spec, Bc.g * ns + Sc.g * spec, Bc.b * ns + Sc.b * spec};
                    T = RGB{T.r * f.r, T.g * f.g, T.b * f.b};
                    hn = h.n;
                    hp = h.p;
                    hprim = h.prim;
                    ++bounce;
                    end_path = bounce >= B;
                }
                if (end_path) {  // hand the sample colour to the slot's owner
                    const uint32_t sidx = task >> 6;
                    ring[ring_row(sidx) * n_hit + (int)(task & 63u)] = make_float4(L.r, L.g, L.b, __uint_as_float(sidx));
                    busy = false;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (blocks) {
            if (fold_px) write_pixel(pixel, acc);
        } else if (MULTI && bgrid) {
            const int oy = (int)(own_pixel / (uint32_t)W), ox = (int)(own_pixel - (uint32_t)oy * (uint32_t)W);
            write_blocks(owner, ox, oy, acc, bgrid_keep);
        } else if (!DEFER && !PROBE && owner) {
            write_pixel(own_pixel, acc);
        }
        if constexpr (DEFER) {
            if (chained) {  // the running mean goes back — with the pixel, if this was the tile's last chunk — and the tile's count goes up
                if (owner) {
                    if (blockIdx.z + 1 == gridDim.z) write_pixel(own_pixel, acc);
                    else coherent_store(P.accumulator + own_pixel, acc);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every lane's stores have completed
                if (lane == 0) coherent_store(P.tile_chain + tile_id, (uint32_t)blockIdx.z + 1u);
            }
        }
    }

#if defined(SRT_STATS) && SRT_STATS == 3
    SRT_TICK(7);
    for (int i = 0; i < 8; ++i) SRT_STAT(i, prof.acc[i] > 0 ? prof.acc[i] : 0);
#endif
#if defined(SRT_STATS) && SRT_STATS == 5
    SRT_STAT(7, (long long)__builtin_readcyclecounter() - t_start);  // the wave's whole life
#endif
#if defined(SRT_STATS) && SRT_STATS == 6
    {
        const long long t_end = (long long)__builtin_readcyclecounter();
        const unsigned long long r_end = __builtin_amdgcn_s_memrealtime();
        const unsigned hw_end = __builtin_amdgcn_s_getreg(0xF804), xcc_end = __builtin_amdgcn_s_getreg(0xF814);
        const size_t w = ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (WG_TILES_X * WG_TILES_Y) + wave;
        if (lane == 0 && w < (size_t)WAVE_LOG_MAX) {
            unsigned long long* e = g_wave_log + 6 * w;
            e[0] = (unsigned long long)t_start, e[1] = (unsigned long long)t_end, e[2] = r_start, e[3] = r_end;
            e[4] = (unsigned long long)hw_start | ((unsigned long long)xcc_start << 32), e[5] = (unsigned long long)hw_end | ((unsigned long long)xcc_end << 32);
        }
    }
#endif
    if constexpr (PROBE) {
        if (lane < TALLY_N && tally.v) atomicAdd(&P.wg_cost[(size_t)block_id * TALLY_N + lane], tally.v);  // (lane i holds counter i)
        return;
    }
    // SRT_RENDER_COUNT_WORK: what this wave's loops did, added to the launch's totals (srt_get_work_counts)
    if constexpr (TALLY) {
        if (P.work_counter && lane < TALLY_ALL && tally.v) atomicAdd(&P.work_counter[lane], (unsigned long long)tally.v);  // (lane i holds counter i)
    }
    if ((P.flags & 2u) || P.wg_cost) {  // SRT_RENDER_COUNT_RAYS / cost feedback for the dispatch order
        unsigned long long tot = rays;
        for (int off = 32; off > 0; off >>= 1) tot += __shfl_down(tot, off);
        if (lane == 0 && tot) {
            if (P.flags & 2u) atomicAdd(P.ray_counter, tot);
            if (P.wg_cost) {
                // the wave's run time in 10 ns ticks; should the two readings ever not belong together, the wave's ray count
                // stands in (about 16 ticks per ray at the usual rates) instead of a zero that would push the block to the tail
                const long long dt = (long long)__builtin_amdgcn_s_memrealtime() - t_start;
                atomicAdd(&P.wg_cost[block_id], (dt > 0 && dt < (1ll << 28)) ? (uint32_t)dt : (uint32_t)(tot < (1ull << 24) ? tot * 16ull : (1ull << 28)));
            }
        }
        // ... and what the wave's loops did, for the launch-shape rule (srt_render): the first TALLY_N counts of every wave of the
        // block (the four waves of a workgroup apart: a workgroup holds its slot until its slowest wave ends; the layers of a
        // sample-chunked launch add up).  Counts, not times: the same in every run.  The host weighs them (ProbeWeights).
        if constexpr (TALLY) if (P.wg_cost && lane < TALLY_N && tally.v) {
            uint32_t* rec = P.wg_cost + P.wg_blocks + ((size_t)block_id * (WG_TILES_X * WG_TILES_Y) + wave) * TALLY_N;
            atomicAdd(&rec[lane], tally.v);  // (lane i holds counter i)
        }
    }
}

// ---- one-sample launches (the reference's own frame loop adds ONE sample per frame, Raytracer.cpp:572-595) run through
// pathtrace_kernel like every other launch.  Two dedicated kernels were built, tested bit-exact and measured in round 3, and
// both were dropped: (1) a streaming pool — resident waves pull pixels from a device counter, a free lane starts
// the next pixel from its PRIMARY ray in the same closest_hit calls as the others' bounce rays: every step then pays ray
// generation, running mean, tone map and two scattered stores for a handful of lanes; Scene1 1080p 0.325 ms vs 0.286 ms,
// Scene_indirect 0.81 vs 0.60; (2) a 256-pixel pool — four primary rounds per wave, 28-byte records in LDS, the pool over the
// records, one dense fold/store pass: Scene1 0.293 vs 0.289 ms, Scene_indirect 0.73 vs 0.61, config 4's scene 0.73 vs 0.56.
// A 1080p frame is 8 pixels per lane of the chip: whatever a wave saves by not thinning out it loses to the coarser grain of
// the launch (a wave's second strip of 256 pixels is the whole launch's tail) and to the records' round trip.

// Second half of a sample-chunked launch: one wave per tile, lane k folds slot k's colours, in sample
// order, into the running mean and stores the pixel — the very operations the owner lanes of
// pathtrace_kernel perform when a workgroup traces all samples itself.  Streams the sample buffer once.
__global__ void __launch_bounds__(256) fold_kernel(const KernelParams P, int tiles_x_wg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t wg = blockIdx.x;
    const size_t tile_id = wg * (WG_TILES_X * WG_TILES_Y) + wave;
    const unsigned long long mask = P.tile_masks[tile_id];
    const int n_hit = __builtin_popcountll(mask);
    if (lane >= n_hit) return;
    // the lane-th set bit of the mask is this slot's lane (pixel) in the tile
    unsigned long long m = mask;
    for (int i = 0; i < lane; ++i) m &= m - 1ull;
    const int bit = __builtin_ctzll(m);
    const int bx = (int)(wg % (size_t)tiles_x_wg), by = (int)(wg / (size_t)tiles_x_wg);
    const int x = bx * WG_W + (bit & (TILE_W - 1)) * WG_TILES_X + (wave % WG_TILES_X);  // (pathtrace_kernel's pixel-to-wave dealing)
    const int y = P.y0 + by * WG_H + (bit / TILE_W) * WG_TILES_Y + (wave / WG_TILES_X);
    const uint32_t pixel = (uint32_t)(x + y * P.width);
    // chained chunks: the tile's first `at` chunks were folded by the workgroups that traced them (KernelParams.tile_chain)
    uint32_t first = 0u;
    if (P.tile_chain) {
        const uint32_t at = P.tile_chain[tile_id];
        if (at >= (uint32_t)P.chunk_layers) return;  // all of them: the last one stored the pixel
        const uint32_t zf = (uint32_t)P.chunk_full, S = (uint32_t)P.chunk, half = S >> 1;
        first = at < zf ? at * S : zf * S + (at - zf) * half;  // (pathtrace_kernel's s_base)
    }
    float4 acc = ((P.flags & 1u) && first == 0u) ? make_float4(0, 0, 0, 0) : P.accumulator[pixel];
    // (Round 3 tried the layout [tile][group of 8 samples][slot][8 samples] — a 128-byte line = 8 consecutive samples of ONE slot,
    // written by one lane — to get the sample buffer's lines to memory in one piece: WRITE_SIZE of config 3's floor band stayed at
    // 2.6x the bytes stored (5.5 GB for 2.1 GB: a line's eight 16-byte writes are many steps apart in either layout, longer than
    // 512 waves' open lines stay in an XCD's 4 MB L2), and this kernel's strided reads made the middle band 5 % slower.)
    const float4* row = P.sample_rows + tile_id * P.sample_count * 64 + lane;
    const uint32_t n = P.sample_count;
    uint32_t s = first;
    for (; s + 8 <= n; s += 8) {  // eight loads in flight per lane
        float4 c[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) c[k] = row[(size_t)(s + k) * 64];
#pragma unroll
        for (int k = 0; k < 8; ++k) accumulate_sample(P, acc, RGB{c[k].x, c[k].y, c[k].z}, s + k);
    }
    for (; s < n; ++s) {
        const float4 c = row[(size_t)s * 64];
        accumulate_sample(P, acc, RGB{c.x, c.y, c.z}, s);
    }
    store_pixel(P, pixel, acc);
}

// srt_selftest_arith: vector i of the test set, normalized() against normalized_ieee(), bit for bit.  Waves 0 mod 4 draw all
// three components from one moderate range (the short path runs), the others mix in every class of float (mostly the library
// path, and the decision between the two is itself under test).
__global__ void __launch_bounds__(256) selftest_normalize_kernel(uint32_t seed, unsigned long long first, unsigned long long n, unsigned long long* mismatches) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256ull + threadIdx.x;
    const unsigned wave_kind = (unsigned)((i >> 6) & 3ull);
    uint32_t h = srt_mix32(seed ^ (uint32_t)i) + (uint32_t)(i >> 32) * 0x9E3779B9U;
    auto next = [&]() { h = srt_mix32(h + 0x9E3779B9U); return h; };
    auto any_float = [&](unsigned kind) {
        const uint32_t r = next();
        switch (kind) {
            case 0: return __uint_as_float((r & 0x807FFFFFu) | ((100u + (next() % 56u)) << 23));   // |v| in [2^-27, 2^29)
            case 1: return __uint_as_float(r);                                                      // any bit pattern
            case 2: return __uint_as_float((r & 0x807FFFFFu) | ((next() % 255u) << 23));            // any finite exponent, denormals
            default: {
                const uint32_t pick = next() % 8u;
                const float special[8] = {0.0f, -0.0f, 1.0f, -1.0f, __uint_as_float(0x7f800000u), __uint_as_float(0x00000001u), __uint_as_float(0x7fc00000u), 0x1p-63f};
                return special[pick];
            }
        }
    };
    V3 a;
    if (wave_kind == 0) {
        const unsigned base = 40u + next() % 170u;  // one exponent per vector, components within 2^8 of each other
        auto comp = [&]() { const uint32_t r = next(); return __uint_as_float((r & 0x807FFFFFu) | ((base + (r >> 8) % 8u) << 23)); };
        a = v3(comp(), comp(), comp());
    } else {
        a = v3(any_float(next() % 4u), any_float(next() % 4u), any_float(wave_kind));
    }
    const V3 f = normalized(a), g = normalized_ieee(a);
    bool same = __float_as_uint(f.x) == __float_as_uint(g.x) && __float_as_uint(f.y) == __float_as_uint(g.y) && __float_as_uint(f.z) == __float_as_uint(g.z);
    // div_window at its two call sites: 1 / frame for every frame in [1, 2^24] (vector i checks frame i + 1), and the slab
    // slopes of box_ray_setup for the components of a normalized direction (g: in [-1, 1] or NaN, zeros included)
    if (i < 16777216ull) {
        const float fr = (float)(int)(i + 1ull);
        same = same && __float_as_uint(div_window(1.0f, fr)) == __float_as_uint(1.0f / fr);
    }
    const float eps = (float)1e-8;
    const float comps[3] = {g.x, g.y, g.z};
    for (int k = 0; k < 3; ++k) {
        const float sg = sign1(comps[k]), dn = tmax(fabsf(comps[k]), eps);
        const float u = div_window(sg, dn), w = sg / dn;
        same = same && (__float_as_uint(u) == __float_as_uint(w) || (u != u && w != w));  // (a NaN slope rejects every box whatever its payload)
    }
    // normalized_in_window on what its one caller feeds it — three draws of [0, 32767] through rand_unit, (r - 0.5) * 2 — against the
    // library path: the 64 combinations of the extreme and the two middle draws first, random triples after that
    {
        const unsigned long long g = first + i;
        const uint32_t edge[4] = {0u, 16383u, 16384u, 32767u};
        const uint32_t hh = next();
        const uint32_t r0 = g < 64ull ? edge[g & 3ull] : (hh & 32767u), r1 = g < 64ull ? edge[(g >> 2) & 3ull] : ((hh >> 15) & 32767u),
                       r2 = g < 64ull ? edge[(g >> 4) & 3ull] : (next() & 32767u);
        const V3 sr = v3((rand_unit(r0) - 0.5f) * 2, (rand_unit(r1) - 0.5f) * 2, (rand_unit(r2) - 0.5f) * 2);
        const V3 u = normalized_in_window(sr), w = normalized_ieee(sr);
        same = same && __float_as_uint(u.x) == __float_as_uint(w.x) && __float_as_uint(u.y) == __float_as_uint(w.y) && __float_as_uint(u.z) == __float_as_uint(w.z);
    }
    // sqrt_window against sqrtf inside its window: +0, NaN (any payload, either sign), every x in [2^-96, inf) — the bit pattern
    // of vector first + i for the call's first 2^31 vectors (so 2^31 vectors cover every non-negative float), a drawn one after that
    {
        const unsigned long long g = first + i;  // (the launch's first vector: a call runs in grids of 2^28)
        const uint32_t bits = g < 0x80000000ull ? (uint32_t)g : (next() & 0x7FFFFFFFu);
        const float x = __uint_as_float(bits);
        const bool in_window = bits == 0u || (bits >= 0x0F800000u && bits < 0x7F800000u) || bits > 0x7F800000u;
        if (in_window) {
            const float u = sqrt_window(x), w = sqrtf(x);
            same = same && (__float_as_uint(u) == __float_as_uint(w) || (u != u && w != w));
            const float xn = __uint_as_float(bits | 0x80000000u);  // negative NaNs (x is never a negative number: d2 <= r*r)
            if (bits > 0x7F800000u) {
                const float un = sqrt_window(xn), wn = sqrtf(xn);
                same = same && (un != un && wn != wn);
            }
        }
    }
    if (i < n && !same) atomicAdd(mismatches, 1ull);
}

// Picking (Raytracer.cpp:525-541): one wave, every lane traces the same ray.
template <bool SCENE_LDS>
__global__ void __launch_bounds__(64) pick_kernel(const KernelParams P, int px, int py, int* out_index) {
    extern __shared__ float4 lds_scene[];
    if constexpr (SCENE_LDS) {
        for (int i = threadIdx.x; i < P.scene_vec4; i += 64) lds_scene[i] = P.scene[i];
        __syncthreads();
    }
    const Lds S = make_lds<SCENE_LDS>(P, lds_scene, 1, 0);
    float nX = ((float)px / (float)P.width) * 2 - 1;
    float nY = ((float)py / (float)P.height) * 2 - 1;
    V3 u = v3(P.right_rd[0] * nX, P.right_rd[1] * nX, P.right_rd[2] * nX);
    V3 vv = v3(P.up_ld[0] * nY, P.up_ld[1] * nY, P.up_ld[2] * nY);
    const V3 dir = normalized(v3((u.x + vv.x) + P.fwd_clip[0], (u.y + vv.y) + P.fwd_clip[1], (u.z + vv.z) + P.fwd_clip[2]));
    bool deferred = false;
#if defined(SRT_STATS) && SRT_STATS == 3
    Prof prof{};
#endif
    Tally<false> no_tally;
    const Hit h = closest_hit<true, false, SCENE_LDS>(S, P, v3(P.cam_pos[0], P.cam_pos[1], P.cam_pos[2]), dir, true, 1, deferred, no_tally SRT_PROF_ARG);
    if (threadIdx.x == 0) {
        out_index[0] = h.prim >= 0 ? S.order(h.prim) : -1;
        out_index[1] = __float_as_int(h.t);
        out_index[2] = h.prim;
        out_index[3] = __float_as_int(h.n.z);
    }
}

// ---- initial dispatch order, before any launch of a frame has recorded block costs -----------------------
// A wave probes four blocks of tiles, 16 lanes each: every lane traces ONE path (the first sample of a pixel of
// a 4 x 4 grid over the block; same ray generation, materials and random stream as the real thing, no colours)
// and the block's cost estimate is the number of rays of its 16 paths, rays that end on a mesh counted four
// times.  That is 1/16 of the pixels at one sample — well under 1 % of a 32-spp frame — and it is what the order
// needs: where the long paths are (reflections, rooms, meshes), which the primary hit alone does not tell
// (measured: ordering by primary hits left the first launch of config 4 13 % and of Scene3 10 % behind the
// learned order).  Any order gives the same image.
constexpr int ORDER_BUCKETS = 16, ORDER_SORT_THREADS = 512;
// (Rounds 2-3 also derived the multi-GPU balance cost here, from these 16 paths per block; since round 3 that is the job of the
// PROBE instantiation of pathtrace_kernel, which runs the real pool — see Tally.)
template <bool SCENE_LDS>
__global__ void __launch_bounds__(64) block_cost_kernel(const KernelParams P, uint32_t* cost, int blocks_x, int n_blocks) {
    extern __shared__ float4 lds_scene[];
    if constexpr (SCENE_LDS) {
        for (int i = threadIdx.x; i < P.scene_vec4; i += 64) lds_scene[i] = P.scene[i];
        __syncthreads();
    }
    const Lds S = make_lds<SCENE_LDS>(P, lds_scene, 1, 0);
    const int lane = threadIdx.x, block = (int)blockIdx.x * 4 + (lane >> 4), k = lane & 15;
    const int bx = block % blocks_x, by = block / blocks_x;
    const int block_h = P.tile_h * WG_TILES_Y;
    const int tx = bx * WG_W + 4 * (k & 3) + 2, ty = by * block_h + ((k >> 2) * block_h) / 4 + block_h / 8;
    const bool in_range = block < n_blocks && tx < P.width && ty < P.rows;
    const int x = in_range ? tx : 0, y = in_range ? P.y0 + ty : P.y0;
    float nX = ((float)x / (float)P.width) * 2 - 1;
    float nY = ((float)y / (float)P.height) * 2 - 1;
    V3 u = v3(P.right_rd[0] * nX, P.right_rd[1] * nX, P.right_rd[2] * nX);
    V3 vv = v3(P.up_ld[0] * nY, P.up_ld[1] * nY, P.up_ld[2] * nY);
    V3 sray = normalized(v3((u.x + vv.x) + P.fwd_clip[0], (u.y + vv.y) + P.fwd_clip[1], (u.z + vv.z) + P.fwd_clip[2]));
    bool deferred = false;
#if defined(SRT_STATS) && SRT_STATS == 3
    Prof prof{};
#endif
    Tally<false> no_tally;
    Hit h = closest_hit<true, false, SCENE_LDS>(S, P, v3(P.cam_pos[0], P.cam_pos[1], P.cam_pos[2]), sray, in_range, 1, deferred, no_tally SRT_PROF_ARG);
    const int first_mesh_prim = S.nsT + S.nb;
    unsigned c = in_range ? 1u : 0u;
    bool alive = in_range && h.prim >= 0 && P.max_bounces > 0 && !(P.flags & 4u);
    uint32_t rng = srt_rng_key(P.seed, (uint32_t)(x + y * P.width), P.first_sample);
    float spec = 0.0f;
    if (alive) {
        spec = (S.mat(h.prim, 0).y >= rand_unit(srt_mix32(rng) >> 17)) ? 1.0f : 0.0f;
        rng += 0x9E3779B9U;
        c += h.prim >= first_mesh_prim ? (SRT_MESH_ORDER_W - 1u) : 0u;
    }
    for (int bounce = 0; bounce < P.max_bounces && __builtin_amdgcn_ballot_w64(alive) != 0ull; ++bounce) {
        V3 o = v3(0, 0, 0);
        if (alive) {  // the bounce step of the path pool (Raytracer.cpp:172-177), directions only
            const float k2 = 2 * dot3(sray, h.n);
            const V3 refl = v3(sray.x - h.n.x * k2, sray.y - h.n.y * k2, sray.z - h.n.z * k2);
            V3 sr = v3((rand_unit(srt_mix32(rng) >> 17) - 0.5f) * 2, (rand_unit(srt_mix32(rng + 0x9E3779B9U) >> 17) - 0.5f) * 2,
                       (rand_unit(srt_mix32(rng + 2u * 0x9E3779B9U) >> 17) - 0.5f) * 2);
            rng += 3u * 0x9E3779B9U;
            sr = normalized(sr);
            if (dot3(sr, h.n) < 0) sr = v3(sr.x * -1, sr.y * -1, sr.z * -1);
            const float tt = S.mat(h.prim, 0).x * spec;
            sray = normalized(v3(sr.x * (1 - tt) + refl.x * tt, sr.y * (1 - tt) + refl.y * tt, sr.z * (1 - tt) + refl.z * tt));
            o = v3(h.p.x + h.n.x * .00001f, h.p.y + h.n.y * .00001f, h.p.z + h.n.z * .00001f);
        }
        const Hit g = closest_hit<true, false, SCENE_LDS>(S, P, o, sray, alive, 1, deferred, no_tally SRT_PROF_ARG);
        if (alive) {
            c += g.prim >= first_mesh_prim ? SRT_MESH_ORDER_W : 1u;
            if (g.prim < 0) {
                alive = false;
            } else {
                spec = (S.mat(g.prim, 0).y >= rand_unit(srt_mix32(rng) >> 17)) ? 1.0f : 0.0f;
                rng += 0x9E3779B9U;
                h = g;
            }
        }
    }
    for (int off = 8; off > 0; off >>= 1) c += __shfl_down(c, off, 16);
    if (k == 0 && block < n_blocks) cost[block] = c;
}

// The probe estimate is 16 one-sample paths per block: noisy.  Costs vary smoothly over the image except at object
// edges, so every block gets the sum over its 3 x 3 neighbourhood (out[b], blocks_x blocks per row) ...
__global__ void __launch_bounds__(256) smooth_cost_kernel(const uint32_t* cost, uint32_t* out, int n, int blocks_x) {
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i >= n) return;
    const int bx = i % blocks_x, by = i / blocks_x, rows = (n + blocks_x - 1) / blocks_x;
    uint32_t sum = 0u;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int x = bx + dx < 0 ? 0 : bx + dx >= blocks_x ? blocks_x - 1 : bx + dx, y = by + dy < 0 ? 0 : by + dy >= rows ? rows - 1 : by + dy;
            const int j = y * blocks_x + x;
            sum += cost[j < n ? j : i];
        }
    out[i] = sum;
}

// ... and the blocks are sorted into only ORDER_BUCKETS linear cost buckets, dearest first, by a STABLE counting sort
// (one workgroup; n is a few thousand to a few hundred thousand): neighbours of similar cost stay in their natural
// order.  (The costs RECORDED by a launch are exact; the host sorts those into 128 buckets, srt_capi.hip.)
__global__ void __launch_bounds__(ORDER_SORT_THREADS) order_sort_kernel(const uint32_t* cost, uint32_t* order, int n) {
    __shared__ uint32_t hist[ORDER_BUCKETS * ORDER_SORT_THREADS];  // [bucket][thread]
    __shared__ uint32_t part[ORDER_SORT_THREADS];
    __shared__ uint32_t red[2 * ORDER_SORT_THREADS / 64];
    const int t = threadIdx.x;
    const int chunk = (n + ORDER_SORT_THREADS - 1) / ORDER_SORT_THREADS, b0 = t * chunk < n ? t * chunk : n, b1 = b0 + chunk < n ? b0 + chunk : n;
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    for (int i = b0; i < b1; ++i) {
        const uint32_t c = cost[i];
        lo = c < lo ? c : lo, hi = c > hi ? c : hi;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t l2 = __shfl_down(lo, off), h2 = __shfl_down(hi, off);
        lo = l2 < lo ? l2 : lo, hi = h2 > hi ? h2 : hi;
    }
    if ((t & 63) == 0) red[t >> 6] = lo, red[ORDER_SORT_THREADS / 64 + (t >> 6)] = hi;
    for (int i = t; i < ORDER_BUCKETS * ORDER_SORT_THREADS; i += ORDER_SORT_THREADS) hist[i] = 0u;
    __syncthreads();
    lo = 0xFFFFFFFFu, hi = 0u;
    for (int i = 0; i < ORDER_SORT_THREADS / 64; ++i) {
        lo = red[i] < lo ? red[i] : lo;
        hi = red[ORDER_SORT_THREADS / 64 + i] > hi ? red[ORDER_SORT_THREADS / 64 + i] : hi;
    }
    const float scale = hi > lo ? (float)ORDER_BUCKETS / ((float)(hi - lo) + 1.0f) : 0.0f;
    auto bucket = [&](uint32_t c) {
        int b = (ORDER_BUCKETS - 1) - (int)((float)(c - lo) * scale);
        return b < 0 ? 0 : b > ORDER_BUCKETS - 1 ? ORDER_BUCKETS - 1 : b;
    };
    for (int i = b0; i < b1; ++i) hist[bucket(cost[i]) * ORDER_SORT_THREADS + t] += 1u;
    __syncthreads();
    // exclusive prefix sum over the flattened [bucket][thread] table: thread t owns ORDER_BUCKETS consecutive entries
    uint32_t sum = 0u;
    for (int i = 0; i < ORDER_BUCKETS; ++i) sum += hist[t * ORDER_BUCKETS + i];
    part[t] = sum;
    __syncthreads();
    for (int off = 1; off < ORDER_SORT_THREADS; off <<= 1) {
        const uint32_t v = t >= off ? part[t - off] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t base = part[t] - sum;
    for (int i = 0; i < ORDER_BUCKETS; ++i) {
        const uint32_t v = hist[t * ORDER_BUCKETS + i];
        hist[t * ORDER_BUCKETS + i] = base;
        base += v;
    }
    __syncthreads();
    for (int i = b0; i < b1; ++i) {
        const int b = bucket(cost[i]);
        order[hist[b * ORDER_SORT_THREADS + t]++] = (uint32_t)i;
    }
}

}  // namespace srt
