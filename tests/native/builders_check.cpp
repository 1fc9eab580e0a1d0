#include <cstdio>
#include <random>
#include "srt_scene_image.h"
#include "srt_mesh_bvh.h"
int main() {
    std::mt19937 rng(5);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    size_t nodes = 0, vec4 = 0;
    for (int it = 0; it < 300; ++it) {
        int nobj = rng() % 200;
        std::vector<srt_object> objs(nobj);
        std::vector<srt::HostMesh> meshes(1 + rng() % 2);
        for (auto& m : meshes) {
            int nt = it % 7 == 0 ? 0 : (int)(rng() % (it % 11 == 0 ? 20000 : 300));
            int mode = rng() % 5;
            for (int k = 0; k < nt; ++k) {
                float b[3] = {U(rng) * 3, U(rng) * 3, U(rng) * 3};
                for (int q = 0; q < 3; ++q)
                    for (int a = 0; a < 3; ++a) {
                        float v = mode == 0 ? b[a] : b[a] + U(rng) * (mode == 1 ? 0.0f : 0.3f);
                        if (mode == 2 && k % 17 == 0) v *= 1e6f;
                        if (mode == 3) v = (float)(int)(v * 2);
                        m.vertices.push_back(v);
                    }
                uint32_t base = (uint32_t)(3 * k);
                m.indices.push_back(base); m.indices.push_back(base + 1); m.indices.push_back(mode == 4 && k % 5 == 0 ? base + 100000 : base + 2);
            }
        }
        for (auto& o : objs) {
            memset(&o, 0, sizeof o);
            int t = rng() % 10;
            o.type = t < 6 ? SRT_OBJ_SPHERE : t < 8 ? SRT_OBJ_BOX : t < 9 ? SRT_OBJ_MESH : SRT_OBJ_NONE;
            for (int a = 0; a < 3; ++a) o.position[a] = U(rng) * (it % 5 == 0 ? 1e5f : 8.f), o.half_size[a] = U(rng) + 1.f;
            o.radius = it % 13 == 0 && rng() % 20 == 0 ? INFINITY : U(rng) * (rng() % 30 == 0 ? 500.f : 0.5f);
            if (rng() % 50 == 0) o.position[0] = NAN;
            o.mesh = (int)(rng() % meshes.size());
        }
        std::vector<float4> img;
        srt::SceneLayout L = srt::build_scene_image(objs.data(), objs.size(), it % 3 != 0, img);
        vec4 += (size_t)L.total_vec4;
        srt::MeshImage mi;
        srt::build_mesh_image(objs.data(), objs.size(), meshes, L.nsT + L.nb, mi);
        nodes += (size_t)mi.n_nodes;
        // structural checks of the wide BVH: every triangle is referenced by exactly one leaf
        if (mi.n_tris > 0) {
            std::vector<int> seen((size_t)mi.n_tris, 0);
            std::vector<int> stack{0};
            while (!stack.empty()) {
                int nd = stack.back(); stack.pop_back();
                const float4 h0 = mi.nodes[srt::NODE_VEC4 * (size_t)nd], h1 = mi.nodes[srt::NODE_VEC4 * (size_t)nd + 1];
                uint32_t w0, first_inner, first_tri, lw;
                memcpy(&w0, &h0.w, 4); memcpy(&first_inner, &h1.x, 4); memcpy(&first_tri, &h1.y, 4); memcpy(&lw, &h1.z, 4);
                const uint32_t innermask = w0 >> 24, leafmask = lw & 255u, counts = lw >> 8;
                if (innermask & leafmask) { std::printf("child both inner and leaf\n"); return 1; }
                int ni = 0; uint32_t tri = first_tri;
                for (int c = 0; c < 8; ++c) {
                    if (innermask >> c & 1u) { int ref = (int)first_inner + ni++; if (ref <= nd || ref >= mi.n_nodes) { std::printf("bad node ref\n"); return 1; } stack.push_back(ref); }
                    else if (leafmask >> c & 1u) { int cnt = 1 + (int)((counts >> (2 * c)) & 3u); for (int k = 0; k < cnt; ++k) { if ((int)tri >= mi.n_tris) { std::printf("bad leaf\n"); return 1; } seen[(size_t)tri++]++; } }
                }
            }
            for (int k = 0; k < mi.n_tris; ++k) if (seen[(size_t)k] != 1) { std::printf("triangle %d seen %d times (it %d)\n", k, seen[(size_t)k], it); return 1; }
        }
    }
    {  // adversarial for a binned SAH: geometrically spaced centroids peel one triangle off per level.  The
       // builder must neither recurse once per triangle nor produce a tree deeper than its median-split bound.
        srt::HostMesh m;
        const int nt = 60000;
        double x = 1e-30;
        for (int k = 0; k < nt; ++k) {
            x *= 1.0012;
            const float f = (float)x;
            const float v[9] = {f, 0, 0, f * 1.0001f, f * 1e-3f, 0, f, 0, f * 1e-3f};
            for (float q : v) m.vertices.push_back(q);
            m.indices.push_back(3 * k), m.indices.push_back(3 * k + 1), m.indices.push_back(3 * k + 2);
        }
        srt_object o;
        memset(&o, 0, sizeof o);
        o.type = SRT_OBJ_MESH;
        srt::MeshImage mi;
        srt::build_mesh_image(&o, 1, std::vector<srt::HostMesh>{m}, 0, mi);
        if (mi.n_tris != nt || mi.max_depth > 40 + 17) { std::printf("adversarial mesh: %d tris depth %d\n", mi.n_tris, mi.max_depth); return 1; }
    }
    std::printf("ok nodes %zu vec4 %zu\n", nodes, vec4);
}
