// srt_scene_image.h — host-side builder of the device scene image (flattened, LDS-shaped).
//
// ObjectsToRender (Raytracer.cpp:61) arrives as srt_object[] in list order.  The image is an
// array of float4 that the kernel copies verbatim into LDS:
//
//   SRT_CONST_ROWS rows of constants come FIRST (the kernel's primitive offsets below count from behind them):
//     rows 0..31   srt_powf's table, row i = the bytes of the doubles (invc_i, lnc_i) — read from LDS instead of from
//                  constant memory (two dependent memory round trips per environment colour otherwise)
//     rows 32..35  the environment (Raytracer.cpp:55-59), colours clamped like Color's constructor (Common.hpp:253-262):
//                  (sky.rgb, sun.r) (horizon.rgb, sun.g) (ground.rgb, sun.b) (sunDirection.xyz, 0) — patched in place by
//                  srt_set_environment, out of the kernel's argument registers
//     rows 36..45  srt_pow_coef: the coefficients of srt_powf's two polynomials and its range-reduction constants (20 doubles) —
//                  one broadcast LDS read per term instead of two moves that build the 64-bit literal in front of every FMA
//   [0, nu4)                 "uniform" spheres (cx,cy,cz,r*r), list order, padded to a
//                            multiple of 4 with never-hit dummies (0,0,0,-1).  Every lane
//                            tests all of them (wave-uniform broadcast reads).
//   [nu4, nsT)               clustered spheres: nc clusters of K (multiple of 4) spheres,
//                            spatially grouped, padded with dummies.  nsT = nu4 + nc*K.
//   [nsT, nsT+nc)            cluster bounds (Cx,Cy,Cz,Rg): a bounding sphere of the members,
//                            inflated so that the kernel's cheap test is CONSERVATIVE with
//                            respect to the reference's float arithmetic (proof below).
//   [.., +2nb)               boxes (cx,cy,cz,_)(hx,hy,hz,_), list order.
//   [off_mat, +3(nsT+nb+nm)) three material rows per primitive id p (spheres: p = index in
//                            [0,nsT); boxes: p = nsT + j; mesh objects: p = nsT + nb + m):
//                              (smoothness, specular_amount, base.r, base.g)
//                              (base.b, emissive.r, emissive.g, emissive.b)
//                              (specular.r, specular.g, specular.b, bits(list index))
//
// The list index restores the reference's tie rule (strict '<' in list order keeps the
// lower index, Raytracer.cpp:132) after the spheres have been permuted.
//
// ---- why the cluster test is conservative -------------------------------------------------
// Reference test for sphere j (Object.hpp:115-133), evaluated in binary32 without FMA:
//   L = c-o; tc = |L.d|; e = (d*tc + o) - c; d2 = e.e; candidate iff !(d2 > r*r);
//   recorded only if additionally t1 = tc - sqrt(r*r - d2) < best (so NaN never records).
// Let D be the exact distance from c to the LINE through o along d, eps = 2^-24, and assume
// finite inputs with |d.d - 1| <= 1e-6 (the kernel checks this and otherwise falls back to
// the brute-force scan).  Bounding every rounding step:
//   d2_float >= D^2 - [ 4.2e-7*D*(|L|+|o|) + 4e-7*D^2 + 2e-12*|L|^2 ]
// (the |o| term appears because q = d*tc + o is rounded at the magnitude of absolute
// coordinates).  For a sphere BEHIND the ray (L.d < 0) the mirrored point gives
// |e|^2 = |L|^2 + (3+eta) s^2 >= |L|^2 >= D^2, so the same bound holds.  Hence a sphere can be
// a candidate only if  D <= r + rho,  rho = 4e-6*(|o|+|c|+|L|) + 1e-5*r   (2x margin).
// For a cluster with centre C and R_geo >= |c_j - C| + r_j for all members:
//   D_C <= D_j + |c_j - C| <= R_geo*(1+1e-5) + 8e-6*(|o|_1 + cmax),   cmax = max |c_j|.
// The kernel evaluates  lhs = fma(-s,s,LL)  (LL = |C-o|^2, s = (C-o).d, FMA chains) whose error
// against D_C^2 is below 1.6e-6*LL, and passes the cluster iff
//   lhs <= (Rg + 8e-6*|o|_1)^2 + 4e-6*LL,   Rg = R_geo*(1+1e-5) + 8e-6*cmax  rounded UP.
// So "cluster fails" implies no member can be a candidate: culling never changes the result.
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "srt_defs.h"
#include "srt_pathtrace.h"

namespace srt {

// Magnitude bound behind the kernel's NaN-free box slab test (closest_hit, KF_BOXES_FINITE): with every box centre and half size
// and the ray origin's |.|_1 below it, o - c is below 2e29 in size, the slab slopes are at most 1e8 (or exactly 0), so every
// product is below 2e37 and every sum of two below 4e37 — finite, hence never inf - inf or 0 * inf: no slab distance is a NaN.
// (Round 3 asked only for FINITE numbers, which does not exclude an overflow to infinity on the way: 3.4e30-sized coordinates
// under a slope of 1e8.  Such scenes now take the comparisons as the reference writes them.)
constexpr float SRT_BOX_NO_NAN_BOUND = 1e29f;

constexpr int SRT_CONST_ROWS = 46, SRT_CONST_ENV_ROW = 32, SRT_CONST_COEF_ROW = 36;  // (rows 36-45: srt_pow_coef, 20 doubles)

struct SceneLayout {
    int nu4 = 0;      // uniform sphere slots (multiple of 4)
    int nu = 0;       // uniform spheres actually there (the rest of the slots are dummies)
    int nc = 0;       // clusters (<= 64)
    int K = 4;        // sphere slots per cluster (multiple of 4)
    int nsT = 0;      // nu4 + nc*K
    int nb = 0;       // boxes
    int nm = 0;       // mesh objects (EXTENSION): primitive ids nsT + nb + m, geometry lives in the BVH
    int off_bounds = 0, off_box = 0, off_mat = 0;
    int total_vec4 = 0;
    int n_spheres = 0;  // real spheres (for statistics)
    // every box centre and half size is a number of magnitude below SRT_BOX_NO_NAN_BOUND (the kernel's NaN-free slab test relies on
    // it, together with the same bound on the ray's origin: see closest_hit)
    bool boxes_finite = true;
    // every sphere's r*r is a number in [2^-72, FLT_MAX] (the kernel's short square root relies on it: sqrt_window in
    // srt_kernel.hip.h; a radius below 1.5e-11, zero, infinite or NaN sends the scene to the instantiations that keep the library sqrtf)
    bool radii_in_sqrt_window = true;
};

// the four environment rows of the constants block (colours through Color's clamping constructor, Common.hpp:253-262)
inline void environment_rows(const srt_environment& e, float4 rows[4]) {
    auto c0 = [](float v) { return v < 0 ? 0.0f : v; };
    rows[0] = make_float4(c0(e.sky_color[0]), c0(e.sky_color[1]), c0(e.sky_color[2]), c0(e.sun_color[0]));
    rows[1] = make_float4(c0(e.horizon_color[0]), c0(e.horizon_color[1]), c0(e.horizon_color[2]), c0(e.sun_color[1]));
    rows[2] = make_float4(c0(e.ground_color[0]), c0(e.ground_color[1]), c0(e.ground_color[2]), c0(e.sun_color[2]));
    rows[3] = make_float4(e.sun_direction[0], e.sun_direction[1], e.sun_direction[2], 0.0f);
}

inline uint32_t morton3(uint32_t x, uint32_t y, uint32_t z) {  // 10 bits per axis
    auto spread = [](uint32_t v) {
        v &= 0x3ff;
        v = (v | (v << 16)) & 0x030000FF;
        v = (v | (v << 8)) & 0x0300F00F;
        v = (v | (v << 4)) & 0x030C30C3;
        v = (v | (v << 2)) & 0x09249249;
        return v;
    };
    return spread(x) | (spread(y) << 1) | (spread(z) << 2);
}

// Builds the image. `cluster` = false puts every sphere in the uniform block.
inline SceneLayout build_scene_image(const srt_object* objects, size_t count, bool cluster, std::vector<float4>& img) {
    std::vector<int> spheres, boxes, meshobjs;
    for (size_t i = 0; i < count; ++i) {
        if (objects[i].type == SRT_OBJ_SPHERE) spheres.push_back((int)i);
        if (objects[i].type == SRT_OBJ_BOX) boxes.push_back((int)i);
        if (objects[i].type == SRT_OBJ_MESH) meshobjs.push_back((int)i);
    }
    auto finite_sphere = [&](int i) {
        const srt_object& o = objects[i];
        return std::isfinite(o.position[0]) && std::isfinite(o.position[1]) && std::isfinite(o.position[2]) && std::isfinite(o.radius) &&
               fabs((double)o.position[0]) < 1e15 && fabs((double)o.position[1]) < 1e15 && fabs((double)o.position[2]) < 1e15 &&
               fabs((double)o.radius) < 1e15;
    };
    std::vector<int> uni, small;
    if (cluster && spheres.size() >= 16) {
        std::vector<double> radii;
        for (int i : spheres)
            if (finite_sphere(i)) radii.push_back(fabs((double)objects[i].radius));
        double median = 0;
        if (!radii.empty()) {
            std::nth_element(radii.begin(), radii.begin() + radii.size() / 2, radii.end());
            median = radii[radii.size() / 2];
        }
        for (int i : spheres) {
            bool big = !finite_sphere(i) || fabs((double)objects[i].radius) > 4.0 * median;
            (big ? uni : small).push_back(i);
        }
        if (small.size() < 8) {  // not worth a second level
            uni = spheres;
            small.clear();
        }
    } else {
        uni = spheres;
    }

    SceneLayout L;
    L.n_spheres = (int)spheres.size();
    L.nb = (int)boxes.size();
    L.nm = (int)meshobjs.size();
    L.nu = (int)uni.size();
    L.nu4 = (L.nu + 3) & ~3;
    // order the small spheres along a Morton curve of their centres, then cut into clusters
    if (!small.empty()) {
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (int i : small)
            for (int a = 0; a < 3; ++a) {
                lo[a] = std::min(lo[a], (double)objects[i].position[a]);
                hi[a] = std::max(hi[a], (double)objects[i].position[a]);
            }
        std::vector<std::pair<uint32_t, int>> keyed;
        for (int i : small) {
            uint32_t q[3];
            for (int a = 0; a < 3; ++a) {
                double ext = hi[a] - lo[a];
                double t = ext > 0 ? ((double)objects[i].position[a] - lo[a]) / ext : 0.0;
                q[a] = (uint32_t)std::min(1023.0, std::max(0.0, t * 1023.0));
            }
            keyed.push_back({morton3(q[0], q[1], q[2]), i});
        }
        std::stable_sort(keyed.begin(), keyed.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
        for (size_t k = 0; k < keyed.size(); ++k) small[k] = keyed[k].second;
        int K = 4;
        while ((int)((small.size() + K - 1) / K) > 64) K += 4;
        L.K = K;
        L.nc = (int)((small.size() + K - 1) / K);
        // The Morton cut is only a starting point: where the spheres do not fill a regular grid it makes clusters that
        // straddle gaps (Scene3 / Scene_indirect: bounding radii 0.54 .. 2.77 for spheres of radius 0.2), and the number of
        // (ray, cluster) pairs that reach the exact tests grows with the clusters' cross-sections.  Exchange members between
        // clusters while that lowers the sum of the squared bounding radii (deterministic sweep order; sizes stay as they
        // are; any grouping is equally valid for the culling proof).  Those two scenes: sum of R^2 16.9 -> 4.9.
        if (K == 4 && L.nc >= 2) {
            auto radius2 = [&](const int* m, int n) {
                double c[3] = {0, 0, 0};
                for (int k = 0; k < n; ++k)
                    for (int a = 0; a < 3; ++a) c[a] += (double)objects[m[k]].position[a];
                for (int a = 0; a < 3; ++a) c[a] /= (double)n;
                double r = 0;
                for (int k = 0; k < n; ++k) {
                    const srt_object& o = objects[m[k]];
                    const double dx = (double)o.position[0] - c[0], dy = (double)o.position[1] - c[1], dz = (double)o.position[2] - c[2];
                    r = std::max(r, sqrt(dx * dx + dy * dy + dz * dz) + fabs((double)o.radius));
                }
                return r * r;
            };
            auto size_of = [&](int c) { return (int)std::min(small.size(), (size_t)(c + 1) * (size_t)K) - c * K; };
            std::vector<double> cost((size_t)L.nc);
            for (int c = 0; c < L.nc; ++c) cost[(size_t)c] = radius2(&small[(size_t)c * K], size_of(c));
            for (int pass = 0; pass < 64; ++pass) {
                bool improved = false;
                for (int a = 0; a < L.nc; ++a)
                    for (int b = a + 1; b < L.nc; ++b)
                        for (int i = 0; i < size_of(a); ++i)
                            for (int j = 0; j < size_of(b); ++j) {
                                int &x = small[(size_t)a * K + i], &y = small[(size_t)b * K + j];
                                std::swap(x, y);
                                const double ca = radius2(&small[(size_t)a * K], size_of(a)), cb = radius2(&small[(size_t)b * K], size_of(b));
                                if (ca + cb < (cost[(size_t)a] + cost[(size_t)b]) * (1.0 - 1e-9)) {
                                    cost[(size_t)a] = ca, cost[(size_t)b] = cb;
                                    improved = true;
                                } else {
                                    std::swap(x, y);
                                }
                            }
                if (!improved) break;
            }
        }
    }
    L.nsT = L.nu4 + L.nc * L.K;
    L.off_bounds = L.nsT;
    L.off_box = L.off_bounds + L.nc;
    L.off_mat = L.off_box + 2 * L.nb;
    L.total_vec4 = SRT_CONST_ROWS + L.off_mat + 3 * (L.nsT + L.nb + L.nm);

    std::vector<float4> full((size_t)L.total_vec4, make_float4(0, 0, 0, 0));
    memcpy(full.data(), srt_pow_table_host, 32 * sizeof(float4));  // 64 doubles = 32 rows; the environment rows are patched by the caller
    memcpy(full.data() + SRT_CONST_COEF_ROW, srt_pow_coef_host, 10 * sizeof(float4));
    img.assign((size_t)std::max(L.total_vec4 - SRT_CONST_ROWS, 1), make_float4(0, 0, 0, 0));
    const float4 dummy = make_float4(0, 0, 0, -1.0f);  // d2 > r*r always: never a candidate
    for (int p = 0; p < L.nsT; ++p) img[p] = dummy;

    auto put_material = [&](int p, int list_index) {
        const srt_material& m = objects[list_index].material;
        float ord;
        int32_t idx = list_index;
        memcpy(&ord, &idx, 4);
        // colours go through Color's clamping constructor (Common.hpp:253-262) HERE, once, instead of at every use in the kernel
        auto c0 = [](float v) { return v < 0 ? 0.0f : v; };
        img[L.off_mat + 3 * p + 0] = make_float4(m.smoothness, m.specular_amount, c0(m.base_color[0]), c0(m.base_color[1]));
        img[L.off_mat + 3 * p + 1] = make_float4(c0(m.base_color[2]), c0(m.emissive_color[0]), c0(m.emissive_color[1]), c0(m.emissive_color[2]));
        img[L.off_mat + 3 * p + 2] = make_float4(c0(m.specular_color[0]), c0(m.specular_color[1]), c0(m.specular_color[2]), ord);
    };
    auto put_sphere = [&](int p, int i) {
        const srt_object& o = objects[i];
        const float r2 = o.radius * o.radius;
        img[p] = make_float4(o.position[0], o.position[1], o.position[2], r2);  // Object.hpp:122
        L.radii_in_sqrt_window = L.radii_in_sqrt_window && r2 >= 0x1p-72f && r2 <= 3.402823466e+38f;  // (NaN: false)
        put_material(p, i);
    };
    for (size_t k = 0; k < uni.size(); ++k) put_sphere((int)k, uni[k]);
    for (int c = 0; c < L.nc; ++c) {
        const size_t b = (size_t)c * L.K, e = std::min(small.size(), b + L.K);
        double C[3] = {0, 0, 0};
        for (size_t k = b; k < e; ++k)
            for (int a = 0; a < 3; ++a) C[a] += (double)objects[small[k]].position[a];
        for (int a = 0; a < 3; ++a) C[a] /= (double)(e - b);
        {  // a centre with a smaller enclosing radius than the centroid's, if the Badoiu-Clarkson walk finds one
            auto radius_at = [&](const double* c) {
                double r = 0;
                for (size_t k = b; k < e; ++k) {
                    const srt_object& o = objects[small[k]];
                    const double dx = (double)o.position[0] - c[0], dy = (double)o.position[1] - c[1], dz = (double)o.position[2] - c[2];
                    r = std::max(r, sqrt(dx * dx + dy * dy + dz * dz) + fabs((double)o.radius));
                }
                return r;
            };
            double c[3] = {C[0], C[1], C[2]}, best = radius_at(C);
            for (int t = 1; t <= 256; ++t) {
                size_t far = b;
                double dfar = -1;
                for (size_t k = b; k < e; ++k) {
                    const srt_object& o = objects[small[k]];
                    const double dx = (double)o.position[0] - c[0], dy = (double)o.position[1] - c[1], dz = (double)o.position[2] - c[2];
                    const double dd = sqrt(dx * dx + dy * dy + dz * dz) + fabs((double)o.radius);
                    if (dd > dfar) dfar = dd, far = k;
                }
                for (int a = 0; a < 3; ++a) c[a] += ((double)objects[small[far]].position[a] - c[a]) / (double)(t + 1);
                const double r = radius_at(c);
                if (r < best) best = r, C[0] = c[0], C[1] = c[1], C[2] = c[2];
            }
        }
        float Cf[3];
        for (int a = 0; a < 3; ++a) Cf[a] = (float)C[a];
        double Rgeo = 0, cmax = 0;
        for (size_t k = b; k < e; ++k) {
            const srt_object& o = objects[small[k]];
            double dx = (double)o.position[0] - Cf[0], dy = (double)o.position[1] - Cf[1], dz = (double)o.position[2] - Cf[2];
            Rgeo = std::max(Rgeo, sqrt(dx * dx + dy * dy + dz * dz) + fabs((double)o.radius));
            cmax = std::max(cmax, fabs((double)o.position[0]) + fabs((double)o.position[1]) + fabs((double)o.position[2]));
            put_sphere(L.nu4 + (int)k, small[k]);
        }
        double Rg = Rgeo * (1.0 + 1e-5) + 8e-6 * cmax + 1e-30;
        float Rgf = (float)Rg;
        if ((double)Rgf < Rg) Rgf = nextafterf(Rgf, INFINITY);
        img[L.off_bounds + c] = make_float4(Cf[0], Cf[1], Cf[2], Rgf);
    }
    for (size_t j = 0; j < boxes.size(); ++j) {
        const srt_object& o = objects[boxes[j]];
        img[L.off_box + 2 * j] = make_float4(o.position[0], o.position[1], o.position[2], 0.0f);
        img[L.off_box + 2 * j + 1] = make_float4(o.half_size[0], o.half_size[1], o.half_size[2], 0.0f);
        for (int a = 0; a < 3; ++a) L.boxes_finite = L.boxes_finite && std::fabs(o.position[a]) < SRT_BOX_NO_NAN_BOUND && std::fabs(o.half_size[a]) < SRT_BOX_NO_NAN_BOUND;  // (NaN: false)
        put_material(L.nsT + (int)j, boxes[j]);
    }
    for (size_t m = 0; m < meshobjs.size(); ++m) put_material(L.nsT + L.nb + (int)m, meshobjs[m]);
    memcpy(full.data() + SRT_CONST_ROWS, img.data(), (size_t)(L.total_vec4 - SRT_CONST_ROWS) * sizeof(float4));
    img.swap(full);
    return L;
}

}  // namespace srt
