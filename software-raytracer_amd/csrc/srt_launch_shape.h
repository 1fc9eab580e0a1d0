// srt_launch_shape.h — the launch shape srt_render chooses (tile height, sample chunks, the taper of the last chunks), as a pure
// function of its inputs: the request, the grid, the device's CU count and, once a row band's recording launch has run, what that
// launch's loops DID (counts, never times).  Host-only C++ without any HIP type, so that the rule is unit-tested on the CPU
// (tests/native/shape_check.cpp under ASan + UBSan) and the same calls give the same shape in every run.
#pragma once

#include <math.h>
#include <stddef.h>
#include <stdint.h>

#include <algorithm>
#include <vector>

namespace srt {

// geometry of a workgroup's tile block (srt_kernel.hip.h; srt_capi.hip static_asserts that they agree)
constexpr int SHAPE_TILE_H = 8, SHAPE_WG_W = 16, SHAPE_WG_H = 16, SHAPE_WG_TILES_Y = 2, SHAPE_WAVES_PER_WG = 4;
constexpr int SHAPE_TALLY_N = 8;  // loop counts per wave in the record: steps, groups, node rounds, leaf trips, mesh phases, waves, untraced waves, node tests

// the launch-shape record of one block of tiles: what its four waves' loops did, weighed — `sum` over the waves (the wave slots
// the block occupies over time), `longest` the heaviest wave (how long the workgroup holds its slot)
struct BlockWork {
    float sum, longest;
};

// What the waves of a REAL launch cost, from the counts its recording launch kept: fitted per BLOCK — non-negative least squares of
// every block's recorded wave time against its counts, over the records of configs 3, 4 and 5, config 4's scene at 4K, Scene3 and
// Scene_indirect (61 bands, 0.4 M blocks; tools/shape_fit.py on the dumps of a development build, profiles/r04/shape_fit.txt) — so
// that the rule's figures (dearest block over an even share, simulated fill) come out as they did from round 3's wave TIMES, within
// 10 %, and its thresholds carry over.  In units in which a pool step weighs `step_w` (the balance probe's per-step weight for the
// scene's layout).  Not the balance probe's weights: those are fitted on band totals of a 32-sample probe of a quarter of the pixels.
struct RecordWeights {
    double group, node_test, mesh_phase, wave;
};
constexpr RecordWeights k_record_weights_analytic{110.0, 0.0, 0.0, 2500.0};
constexpr RecordWeights k_record_weights_mesh{127.0, 34.0, 33.0, 6900.0};  // node_test: a child-box test per lane and round, carrying its round's share of pops, shuffles and pushes

struct WorkRecord {
    std::vector<BlockWork> blocks;  // of the recorded grid, longest-running first (what the fill simulation walks)
    double sum = 0.0;               // of the record (0: none)
    double max = 0.0;               // the dearest block: how uneven the blocks are decides the number of sample chunks
    unsigned gx = 0, gy = 0;        // the grid of 8-row tile blocks it was recorded on
    void clear() { blocks.clear(), sum = 0.0, max = 0.0, gx = gy = 0; }
};

// counts: [n blocks][4 waves][SHAPE_TALLY_N] as the recording launch wrote them -> the record
inline void weigh_record(const uint32_t* counts, size_t n, unsigned gx, unsigned gy, double step_w, bool mesh, WorkRecord& rec) {
    const RecordWeights& rw = mesh ? k_record_weights_mesh : k_record_weights_analytic;
    // (order of srt::TALLY_*: steps, groups, node rounds, leaf trips, mesh phases, waves, untraced waves, node tests)
    const double w[SHAPE_TALLY_N] = {step_w, rw.group, 0.0, 0.0, rw.mesh_phase, rw.wave, 0.0, rw.node_test};
    rec.blocks.resize(n);
    rec.sum = 0.0, rec.max = 0.0, rec.gx = gx, rec.gy = gy;
    for (size_t i = 0; i < n; ++i) {
        double sum = 0.0, longest = 0.0;
        for (int v = 0; v < SHAPE_WAVES_PER_WG; ++v) {
            const uint32_t* c = counts + (i * SHAPE_WAVES_PER_WG + (size_t)v) * SHAPE_TALLY_N;
            double x = 0.0;
            for (int k = 0; k < SHAPE_TALLY_N; ++k) x += w[k] * (double)c[k];
            sum += x;
            longest = x > longest ? x : longest;
        }
        rec.blocks[i] = BlockWork{(float)sum, (float)longest};
        rec.sum += sum;
        rec.max = sum > rec.max ? sum : rec.max;
    }
    // (ties in `longest` keep the block order: std::stable_sort, so the record is the same vector in every run)
    std::stable_sort(rec.blocks.begin(), rec.blocks.end(), [](const BlockWork& a, const BlockWork& b) { return a.longest > b.longest; });
}

// How full a launch of `layers` sample chunks keeps the wave slots of `slots` resident workgroups, from the recorded work alone:
// the workgroups are started layer by layer, longest block first (the cost order of the real dispatch), each on the slot that
// frees first; a workgroup holds its slot for its heaviest wave's work / layers, and its four waves occupy their wave slots for
// their own work / layers; the result is occupied wave-slot time / (wave slots x the time the last workgroup ends).  A
// deterministic stand-in for what round 3 read off the recorded launch's event time (wave time / launch time x resident waves):
// that figure moved with the clock and flipped launch shapes near its threshold.
inline double simulate_fill(const std::vector<BlockWork>& by_length, int layers, int slots) {
    if (by_length.empty() || slots < 1 || layers < 1) return 1.0;
    std::vector<double> heap((size_t)slots, 0.0);  // min-heap of the slots' finish times
    auto cmp = [](double a, double b) { return a > b; };
    double occupied = 0.0, end = 0.0;
    for (int z = 0; z < layers; ++z)
        for (const BlockWork& b : by_length) {
            std::pop_heap(heap.begin(), heap.end(), cmp);
            heap.back() += (double)b.longest / (double)layers;
            end = heap.back() > end ? heap.back() : end;
            std::push_heap(heap.begin(), heap.end(), cmp);
            occupied += (double)b.sum / (double)layers;
        }
    return end > 0.0 ? occupied / (end * (double)slots * SHAPE_WAVES_PER_WG) : 1.0;
}

struct ShapeRequest {
    long long grid_w = 0, grid_h = 0;  // lanes of the launch: pixels of the band, or blocks of a block grid
    int rows = 0;                      // scene rows of the band (the sample-chunk grid is always 8-row tiles over the band's pixels)
    uint32_t sample_count = 1;
    int steps = 1;                     // progressive blocks (> 1: no sample chunks)
    bool block_grid = false;           // the launch's lanes are blocks
    bool mesh = false;                 // the scene has triangle meshes (the BVH-enabled kernel: four waves per SIMD)
    int cu_count = 256;
};
struct ShapeOverrides {  // development switches (the shipped library passes the defaults)
    int tile_h = 0;      // 8 / 4 / 2 / 1 forces the tile height (and switches sample chunks off)
    int defer = -1;      // 0: never chunk samples, n > 0: force n samples per chunk
    int chunk_beta = 30; // percent: a chunk's share of an even share, see below
    bool no_taper = false;
    double fill_min = 0.85;
};
struct LaunchShape {
    int tile_h = SHAPE_TILE_H;  // rows of a wave's tile
    int chunk = 0;              // samples per full-size chunk; 0: not chunked
    int chunks = 1;             // grid layers (with the tapered ones)
    int chunk_full = 1;         // layers that trace `chunk` samples; the ones behind them half as many
    uint32_t source = 0;        // 0: the static rule (request and grid only), 1: the band's work record as well
    long long wg_x = 0, wg_y8 = 0, wg8 = 0;  // blocks of 16 x 16 pixels across, down (8-row tiles) and in all
    // figures behind the decision (development output)
    double ratio = 0.0, fill = 0.0;
};

// Tile height and sample chunks.  `rec` may be null or empty (no record for this band yet): the static rule.
inline LaunchShape plan_launch_shape(const ShapeRequest& q, const WorkRecord* rec, const ShapeOverrides& ov = ShapeOverrides()) {
    LaunchShape s;
    // Tile height: with few rows and many samples per pixel (a narrow stripe of a multi-GPU frame) 8-row tiles give too few
    // workgroups to fill 256 CUs x 4 resident workgroups and leave nothing to balance the tail with; halve the tile (twice the
    // workgroups, same lanes at work in each wave's path pool) until there are about four rounds of workgroups.
    s.wg_x = (q.grid_w + SHAPE_WG_W - 1) / SHAPE_WG_W;
    auto groups_at = [&](int th) { return s.wg_x * ((q.grid_h + th * SHAPE_WG_TILES_Y - 1) / (th * SHAPE_WG_TILES_Y)); };
    const long long want = 15LL * q.cu_count;  // ~4 rounds of the 4 workgroups a CU holds; measured on bands of 30..400 rows
    int tile_h = SHAPE_TILE_H;
    while (tile_h > 1 && q.sample_count >= 16 && groups_at(tile_h) < want) tile_h >>= 1;
    // a block grid is 1 / steps^2 of the pixel grid: keep at least two workgroups per CU (the waves' run time is latency)
    while (q.block_grid && tile_h > 1 && groups_at(tile_h) < 2LL * q.cu_count) tile_h >>= 1;
    if (ov.tile_h == 8 || ov.tile_h == 4 || ov.tile_h == 2 || ov.tile_h == 1) tile_h = ov.tile_h;
    // Sample-chunked launch, for 64 samples per pixel and more (32 with meshes): keep the full 8x8 tiles but give every tile to
    // several workgroups, each tracing one chunk of the samples and storing the colours; a second, streaming kernel folds them in
    // order (the running mean is order-dependent).  A narrow stripe of a multi-GPU frame then runs like the full single-GPU frame —
    // many short workgroups — instead of few long ones whose tail idles the chip.
    s.wg_y8 = ((long long)q.rows + SHAPE_WG_H - 1) / SHAPE_WG_H;
    s.wg8 = s.wg_x * s.wg_y8;
    int chunk = 0, chunks = 1;
    if (ov.tile_h == 0 && ov.defer != 0 && (q.sample_count >= 64 || ov.defer > 0 || q.mesh) && q.sample_count >= 32 && q.steps <= 1 && s.wg8 > 0) {
        long long c = (96LL * q.cu_count + s.wg8 - 1) / s.wg8;  // about 24 k workgroups in flight over the launch
        // With block works recorded for this grid (the band's recording launch) the number of chunks follows from how uneven the
        // blocks are: ratio = the dearest block over an even share of the whole launch per resident workgroup.  Well below 1 the
        // cost order alone fills the chip — no chunks for meshes (every chunk repeats the primary hits, mesh phases included, and
        // the colours make a round trip through the sample buffer), two for analytic scenes (finer grains at the tail: -2..-7 %).
        // From 0.85 on the dearest block is brought down to 0.3 of a share; mesh bands that are clearly uneven — through the ball's
        // edge — do better with a finer cut (round 4's sweeps of config 5's bands: rows 1350-1620, ratio 1.5: 155.6 ms in 6
        // layers, 151.2 in 10; 1388-1492, 3.7: 70.4 in 10, 67.7 in 18).
        // (no record yet: a mesh launch of >= 6000 blocks — a whole 1080p frame — starts unchunked, which is what the record of
        // such a frame asks for; smaller ones, the bands of a multi-GPU frame, start with the workgroup-count rule above)
        if (q.mesh && s.wg8 >= 6000) c = 1;
        const bool have_record = rec && rec->sum > 0.0 && (long long)rec->gx == s.wg_x && (long long)rec->gy == s.wg_y8;
        if (have_record) {
            const double slots = (double)q.cu_count * (q.mesh ? 3.0 : 4.0);
            s.ratio = rec->max * slots / rec->sum;
            c = s.ratio < 0.85 ? (q.mesh ? 1 : 2) : (long long)ceil(s.ratio * 100.0 / (double)ov.chunk_beta);
            if (q.mesh && s.ratio >= 1.2) c = (long long)ceil(s.ratio / 0.2);
            s.source = 1;
        }
        // Analytic scenes (five resident workgroups per CU, ring of two): at least ten rounds of workgroups, whatever the record
        // says — a band of evenly dear blocks has a ratio below 1 and got 2..4 chunks, i.e. 3.4 rounds with the last one 40 % full.
        // Chunks of fewer than 24 samples cost more than they balance (every chunk stages the scene and repeats the primary hits).
        const long long min_chunk = q.mesh ? 16 : 24;
        if (!q.mesh) {
            const long long c_fill = (10LL * 5 * q.cu_count + s.wg8 - 1) / s.wg8;
            if (c < c_fill) c = c_fill;
        } else {
            // Mesh launches: four rounds of their four workgroups per CU (a narrow band of evenly dear blocks left in one piece is
            // 1.9 rounds: config 5's floor band 1812-1938 76.9 ms, 74.6 with two chunks, 72.9 with eight) ...
            const long long c_fill = (4LL * 4 * q.cu_count + s.wg8 / 2) / s.wg8;  // (to the nearest: 4080 blocks are four rounds)
            if (c < c_fill) c = c_fill;
            // ... and a launch that would leave a good part of the chip's wave slots empty is cut into four: the upper 1066 rows of
            // config 5 (sky, far spheres, mirror balls: sparse tiles whose one busy wave holds the workgroup's slot) fill 0.63 in one
            // piece, 77.5 ms; 64 in six layers.  Its neighbours fill 0.86..0.93 and lose 1..5 % to any chunking.  (Only where a
            // chunk still has 64 samples and more: config 4's frame, 64 spp, ran 8.0 ms in four chunks of 16 instead of 7.2.)
            if (have_record && c < 4 && q.sample_count >= 256) {
                s.fill = simulate_fill(rec->blocks, (int)c, q.cu_count * 4);
                if (s.fill < ov.fill_min) c = 4;
            }
        }
        if (c > (long long)q.sample_count / min_chunk) c = (long long)q.sample_count / min_chunk;
        // A mesh launch that the rule leaves in ONE piece keeps full 8 x 8 tiles: the small tiles chosen above for launches of few
        // blocks (tuned on analytic scenes at 16..63 spp) cost the mesh kernel more than they balance (config 5's floor bands of a
        // cost-balanced 8-rank split: 107 -> 92 ms, 90 -> 78 ms).
        if (c < 2 && ov.defer <= 0 && q.mesh && ov.tile_h == 0) tile_h = SHAPE_TILE_H;
        if (c >= 2 || ov.defer > 0) {
            chunk = (int)((q.sample_count + c - 1) / c);
            if (ov.defer > 0) chunk = ov.defer;
            chunks = (int)((q.sample_count + chunk - 1) / chunk);
        }
    }
    s.tile_h = tile_h;
    s.chunk = chunks >= 2 ? chunk : 0;
    s.chunks = chunks >= 2 ? chunks : 1;
    s.chunk_full = s.chunks;
    return s;
}

// The sample buffer could not be had: everything in one workgroup per tile, with the small tiles chosen for the grid.
inline void shape_without_sample_buffer(LaunchShape& s) {
    s.chunk = 0, s.chunks = 1, s.chunk_full = 1;
}

// Sample-chunked launches keep full tiles, and their last two chunks (the last one, when there are only two or three) run as twice
// as many of half the size: the grid's last layers are the last workgroups to start, and a launch ends when its last workgroups do
// — in a band of evenly dear blocks that tail is one workgroup's run time (config 3's bands of 60..200 rows -2..-8 %, config 5's
// chunked bands -6..-9 %; three or four tapered chunks: the same as two).
inline void finish_launch_shape(LaunchShape& s, uint32_t sample_count, const ShapeOverrides& ov = ShapeOverrides()) {
    if (s.chunks < 2) return;
    s.tile_h = SHAPE_TILE_H;
    s.chunk_full = s.chunks;
    const int taper = s.chunks >= 4 ? 2 : 1;
    if (s.chunk >= 24 && !ov.no_taper) {
        const int half = s.chunk >> 1, full = s.chunks - taper;                       // layers that keep the full size
        const long long rest = (long long)sample_count - (long long)full * s.chunk;  // samples behind them (the last chunk may be short)
        s.chunk_full = full;
        s.chunks = full + (int)((rest + half - 1) / half);
    }
}

}  // namespace srt
