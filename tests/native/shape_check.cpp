// Unit test of csrc/srt_launch_shape.h (host-only): the launch shape srt_render chooses is a pure function of its inputs.  The
// expected shapes are the ones observed on the GPU for BASELINE's configs (profiles/r04/bench_*.json, chunk_rule_c5.txt).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "srt_launch_shape.h"

static int checks = 0;
#define CHECK(cond)                                                                  \
    do {                                                                             \
        ++checks;                                                                    \
        if (!(cond)) {                                                               \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);            \
            return 1;                                                                \
        }                                                                            \
    } while (0)

static srt::ShapeRequest request(long long w, int rows, uint32_t spp, bool mesh) {
    srt::ShapeRequest q;
    q.grid_w = w, q.grid_h = rows, q.rows = rows, q.sample_count = spp, q.mesh = mesh, q.cu_count = 256;
    return q;
}
static srt::LaunchShape shape_of(const srt::ShapeRequest& q, const srt::WorkRecord* rec) {
    srt::LaunchShape s = srt::plan_launch_shape(q, rec);
    srt::finish_launch_shape(s, q.sample_count);
    return s;
}
// a record of n blocks (grid gx x gy) from per-wave step counts given by f(block, wave)
template <class F>
static srt::WorkRecord record(unsigned gx, unsigned gy, bool mesh, F steps_of) {
    std::vector<uint32_t> counts((size_t)gx * gy * 4 * srt::SHAPE_TALLY_N, 0u);
    for (size_t i = 0; i < (size_t)gx * gy; ++i)
        for (int v = 0; v < 4; ++v) counts[(i * 4 + (size_t)v) * srt::SHAPE_TALLY_N + 0] = steps_of(i, v);
    srt::WorkRecord r;
    srt::weigh_record(counts.data(), (size_t)gx * gy, gx, gy, 1000.0, mesh, r);
    return r;
}

int main() {
    // ---- static rule (no record) ----
    {   // config 2: 1080p, 32 spp: full tiles, one piece
        const srt::LaunchShape s = shape_of(request(1920, 1080, 32, false), nullptr);
        CHECK(s.tile_h == 8 && s.chunks == 1 && s.chunk == 0 && s.source == 0 && s.wg8 == 120 * 68);
    }
    {   // config 3, one rank's 135-row band at 512 spp: 21 chunks of 25, the last two tapered -> 23 layers
        const srt::LaunchShape s = shape_of(request(1920, 135, 512, false), nullptr);
        CHECK(s.tile_h == 8 && s.chunk == 25 && s.chunks == 23 && s.chunk_full == 19 && s.wg8 == 1080);
        long long covered = (long long)s.chunk_full * s.chunk + (long long)(s.chunks - s.chunk_full) * (s.chunk / 2);
        CHECK(covered >= 512);  // the layers cover every sample
    }
    {   // config 4: whole 1080p frame, 64 spp, mesh: one piece, full tiles
        const srt::LaunchShape s = shape_of(request(1920, 1080, 64, true), nullptr);
        CHECK(s.tile_h == 8 && s.chunks == 1);
    }
    {   // config 5, one rank's 270-row band of 4K at 1024 spp, mesh: 7 chunks of 147 -> 9 layers
        const srt::LaunchShape s = shape_of(request(3840, 270, 1024, true), nullptr);
        CHECK(s.chunk == 147 && s.chunks == 9 && s.wg8 == 4080);
    }
    {   // few rows at 16..63 spp: small tiles, no chunks
        const srt::LaunchShape s = shape_of(request(640, 64, 24, false), nullptr);
        CHECK(s.tile_h == 1 && s.chunks == 1);
        const srt::LaunchShape t = shape_of(request(640, 64, 8, false), nullptr);  // below 16 spp the tile stays
        CHECK(t.tile_h == 8);
    }
    {   // a block grid (steps = 8 over 1080p: 240 x 135 blocks): at least two workgroups per CU
        srt::ShapeRequest q = request(240, 135, 1, false);
        q.rows = 1080, q.steps = 8, q.block_grid = true;
        const srt::LaunchShape s = shape_of(q, nullptr);
        CHECK(s.chunks == 1 && s.tile_h < 8 && (15LL * ((135 + 2 * s.tile_h - 1) / (2 * s.tile_h))) >= 512);
    }
    // ---- with a record ----
    {   // analytic band of evenly dear blocks: ratio 1024 / 1080 = 0.95 -> 4, but at least ten rounds of workgroups -> 12 chunks of 43 -> 14 layers
        const srt::WorkRecord r = record(120, 9, false, [](size_t, int) { return 100u; });
        const srt::LaunchShape s = shape_of(request(1920, 135, 512, false), &r);
        CHECK(s.source == 1 && s.chunk == 43 && s.chunks == 14 && s.ratio > 0.94 && s.ratio < 0.96);
        // a record of another grid is not this band's
        const srt::LaunchShape o = shape_of(request(1920, 151, 512, false), &r);
        CHECK(o.source == 0);
    }
    {   // mesh band, clearly uneven (ratio 1.5): the finer cut, 8 chunks of 128 -> 10 layers (config 5's dearest band)
        const srt::WorkRecord r = record(240, 17, true, [](size_t i, int) { return i == 0 ? 1500u * 4080u / 768u : 1000u; });
        const srt::LaunchShape s = shape_of(request(3840, 270, 1024, true), &r);
        CHECK(s.ratio > 1.45 && s.ratio < 1.55 && s.chunk == 128 && s.chunks == 10);
    }
    {   // mesh band just above the threshold (ratio 0.87): 3 chunks of 342 -> 4 layers (config 5's rank 4 of 8)
        const srt::WorkRecord r = record(240, 17, true, [](size_t i, int) { return i == 0 ? 870u * 4080u / 768u : 1000u; });
        const srt::LaunchShape s = shape_of(request(3840, 270, 1024, true), &r);
        CHECK(s.ratio > 0.85 && s.ratio < 0.89 && s.chunk == 342 && s.chunks == 4);
    }
    {   // mesh band of even blocks, every wave busy: one piece, full tiles (fill near 1)
        const srt::WorkRecord r = record(240, 17, true, [](size_t, int) { return 1000u; });
        const srt::LaunchShape s = shape_of(request(3840, 270, 1024, true), &r);
        CHECK(s.source == 1 && s.chunks == 1 && s.tile_h == 8 && s.fill > 0.9);
    }
    {   // the same band with SPARSE tiles — one busy wave per workgroup holds the slot for three idle ones: fill 0.25 -> four pieces -> 6 layers of 256
        const srt::WorkRecord r = record(240, 67, true, [](size_t, int v) { return v == 0 ? 1000u : 0u; });
        const srt::LaunchShape s = shape_of(request(3840, 1068, 1024, true), &r);
        CHECK(s.ratio < 0.85 && s.fill < 0.3 && s.chunk == 256 && s.chunks == 6);
        // ... but not for a launch whose chunks would fall below 64 samples (config 4's 64 spp)
        const srt::WorkRecord r4 = record(120, 68, true, [](size_t, int v) { return v == 0 ? 1000u : 0u; });
        CHECK(shape_of(request(1920, 1080, 64, true), &r4).chunks == 1);
    }
    // ---- simulate_fill ----
    {
        std::vector<srt::BlockWork> even(2048, srt::BlockWork{4.0f, 1.0f});
        CHECK(srt::simulate_fill(even, 1, 1024) > 0.999 && srt::simulate_fill(even, 3, 1024) > 0.999);
        std::vector<srt::BlockWork> one_more(1025, srt::BlockWork{4.0f, 1.0f});
        const double f = srt::simulate_fill(one_more, 1, 1024);
        CHECK(f > 0.49 && f < 0.51);                                   // the last block runs alone for a whole round
        CHECK(srt::simulate_fill(one_more, 8, 1024) > f);              // finer grains fill better
        std::vector<srt::BlockWork> sparse(2048, srt::BlockWork{1.0f, 1.0f});
        const double g = srt::simulate_fill(sparse, 1, 1024);
        CHECK(g > 0.249 && g < 0.251);
        CHECK(srt::simulate_fill(std::vector<srt::BlockWork>(), 1, 1024) == 1.0);
    }
    // ---- the record is the same vector whatever order equal blocks arrive in, and the decision the same on every call ----
    {
        const srt::WorkRecord a = record(120, 9, false, [](size_t i, int v) { return (uint32_t)(100 + (i * 7 + (size_t)v) % 13); });
        const srt::WorkRecord b = record(120, 9, false, [](size_t i, int v) { return (uint32_t)(100 + (i * 7 + (size_t)v) % 13); });
        CHECK(a.blocks.size() == b.blocks.size() && a.sum == b.sum && a.max == b.max);
        for (size_t i = 0; i < a.blocks.size(); ++i) CHECK(a.blocks[i].sum == b.blocks[i].sum && a.blocks[i].longest == b.blocks[i].longest);
        for (size_t i = 1; i < a.blocks.size(); ++i) CHECK(a.blocks[i - 1].longest >= a.blocks[i].longest);
        const srt::LaunchShape s1 = shape_of(request(1920, 135, 512, false), &a), s2 = shape_of(request(1920, 135, 512, false), &b);
        CHECK(s1.chunk == s2.chunk && s1.chunks == s2.chunks && s1.tile_h == s2.tile_h && s1.chunk_full == s2.chunk_full);
    }
    // ---- development overrides ----
    {
        srt::ShapeOverrides ov;
        ov.defer = 64;
        srt::LaunchShape s = srt::plan_launch_shape(request(1920, 135, 512, false), nullptr, ov);
        srt::finish_launch_shape(s, 512, ov);
        CHECK(s.chunk == 64 && s.chunks == 6 + 2 + 2);  // 8 chunks of 64, the last two as four of 32
        ov.no_taper = true;
        s = srt::plan_launch_shape(request(1920, 135, 512, false), nullptr, ov);
        srt::finish_launch_shape(s, 512, ov);
        CHECK(s.chunks == 8 && s.chunk_full == 8);
        // no sample buffer: back to one workgroup per tile with the small tiles of the grid
        srt::LaunchShape t = srt::plan_launch_shape(request(1920, 135, 512, false), nullptr);
        const int small = t.tile_h;
        srt::shape_without_sample_buffer(t);
        srt::finish_launch_shape(t, 512);
        CHECK(t.chunks == 1 && t.chunk == 0 && t.tile_h == small && small < 8);
    }
    std::printf("ok %d checks\n", checks);
    return 0;
}
