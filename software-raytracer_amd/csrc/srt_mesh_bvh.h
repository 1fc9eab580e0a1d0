// srt_mesh_bvh.h — host-side flattening of SRT_OBJ_MESH objects into one world-space
// triangle list + a binary BVH (EXTENSION: the reference has no triangle primitive).
//
// Device layout (HBM, read through L2; too large for LDS at 100k triangles):
//   tris : 3 float4 per triangle, in BVH leaf order
//            (v0.xyz, bits(primitive id p))      p indexes the LDS material table
//            (e1.xyz, bits(global triangle id))  id = position in (object list order, triangle
//            (e2.xyz, bits(list index))               index) — restores the tie rule
//   nodes: 2 float4 per node, depth-first order (left child = node + 1)
//            (lo.xyz, bits(right child | first triangle))
//            (hi.xyz, bits(-(split axis + 1) for an inner node | triangle count of a leaf))
// Build: median split of the centroids along the longest axis, leaves of <= 4 triangles —
// deterministic and O(n log n).  Bounds are exact (float min/max of the float vertices); the
// kernel pads them per ray (see closest_hit) so that the box filter is conservative with
// respect to the rounding of the triangle test.
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "srt_pathtrace.h"

namespace srt {

struct HostMesh {
    std::vector<float> vertices;
    std::vector<uint32_t> indices;
};

struct MeshImage {
    std::vector<float4> tris, nodes;
    int n_tris = 0, n_nodes = 0, n_mesh_objects = 0, max_depth = 0;
    float center[3] = {0, 0, 0}, half[3] = {0, 0, 0};  // root box
};

inline float bits_of(int32_t v) {
    float f;
    memcpy(&f, &v, 4);
    return f;
}

// prim_base = primitive id of the first mesh object (spheres and boxes come before)
inline void build_mesh_image(const srt_object* objects, size_t count, const std::vector<HostMesh>& meshes, int prim_base,
                             MeshImage& out) {
    struct Tri {
        float v0[3], e1[3], e2[3], lo[3], hi[3], c[3];
        int32_t prim, gid, ord;
    };
    std::vector<Tri> tris;
    int mesh_obj = 0;
    for (size_t i = 0; i < count; ++i) {
        const srt_object& o = objects[i];
        if (o.type != SRT_OBJ_MESH) continue;
        const int prim = prim_base + mesh_obj++;
        if (o.mesh < 0 || (size_t)o.mesh >= meshes.size()) continue;
        const HostMesh& m = meshes[(size_t)o.mesh];
        const size_t nv = m.vertices.size() / 3, nt = m.indices.size() / 3;
        for (size_t k = 0; k < nt; ++k) {
            const uint32_t a = m.indices[3 * k], b = m.indices[3 * k + 1], c = m.indices[3 * k + 2];
            if (a >= nv || b >= nv || c >= nv) continue;
            Tri t;
            float v[3][3];
            const uint32_t ix[3] = {a, b, c};
            for (int q = 0; q < 3; ++q)
                for (int ax = 0; ax < 3; ++ax) v[q][ax] = m.vertices[3 * (size_t)ix[q] + ax] + o.position[ax];  // world = vertex + position
            for (int ax = 0; ax < 3; ++ax) {
                t.v0[ax] = v[0][ax];
                t.e1[ax] = v[1][ax] - v[0][ax];
                t.e2[ax] = v[2][ax] - v[0][ax];
                t.lo[ax] = std::min(v[0][ax], std::min(v[1][ax], v[2][ax]));
                t.hi[ax] = std::max(v[0][ax], std::max(v[1][ax], v[2][ax]));
                t.c[ax] = (t.lo[ax] + t.hi[ax]) * 0.5f;
            }
            t.prim = prim;
            t.gid = (int32_t)tris.size();
            t.ord = (int32_t)i;
            bool finite = true;
            for (int ax = 0; ax < 3; ++ax) finite = finite && std::isfinite(t.lo[ax]) && std::isfinite(t.hi[ax]);
            if (finite) tris.push_back(t);  // a non-finite triangle can never produce a valid hit
        }
    }
    out = MeshImage();
    out.n_mesh_objects = mesh_obj;
    out.n_tris = (int)tris.size();
    if (tris.empty()) return;

    struct Node {
        float lo[3], hi[3];
        int32_t a, b;  // inner: a = right child, b = -(split axis + 1); leaf: a = first triangle, b = count
    };
    std::vector<Node> nodes;
    nodes.reserve(tris.size() / 2 + 16);
    struct Rec {
        // depth-first: the left child of node `me` is me + 1, the right child's index is returned
        static int build(std::vector<Node>& nodes, std::vector<Tri>& tris, int b, int e, int depth, int& max_depth) {
            const int me = (int)nodes.size();
            nodes.push_back(Node());
            Node nd;
            float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (int ax = 0; ax < 3; ++ax) {
                nd.lo[ax] = INFINITY;
                nd.hi[ax] = -INFINITY;
            }
            for (int k = b; k < e; ++k)
                for (int ax = 0; ax < 3; ++ax) {
                    nd.lo[ax] = std::min(nd.lo[ax], tris[k].lo[ax]);
                    nd.hi[ax] = std::max(nd.hi[ax], tris[k].hi[ax]);
                    clo[ax] = std::min(clo[ax], tris[k].c[ax]);
                    chi[ax] = std::max(chi[ax], tris[k].c[ax]);
                }
            max_depth = std::max(max_depth, depth);
            if (e - b <= 4) {
                nd.a = b;
                nd.b = e - b;
            } else {
                int axis = 0;
                if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
                if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
                const int mid = (b + e) / 2;
                std::nth_element(tris.begin() + b, tris.begin() + mid, tris.begin() + e, [axis](const Tri& x, const Tri& y) {
                    return x.c[axis] < y.c[axis] || (x.c[axis] == y.c[axis] && x.gid < y.gid);
                });
                build(nodes, tris, b, mid, depth + 1, max_depth);  // left = me + 1
                nd.a = build(nodes, tris, mid, e, depth + 1, max_depth);
                nd.b = -(axis + 1);  // inner node: split axis, so the kernel can visit the near child first
            }
            nodes[me] = nd;
            return me;
        }
    };
    out.max_depth = 0;
    Rec::build(nodes, tris, 0, (int)tris.size(), 1, out.max_depth);

    out.n_nodes = (int)nodes.size();
    out.nodes.resize(nodes.size() * 2);
    for (size_t k = 0; k < nodes.size(); ++k) {
        out.nodes[2 * k] = make_float4(nodes[k].lo[0], nodes[k].lo[1], nodes[k].lo[2], bits_of(nodes[k].a));
        out.nodes[2 * k + 1] = make_float4(nodes[k].hi[0], nodes[k].hi[1], nodes[k].hi[2], bits_of(nodes[k].b));
    }
    out.tris.resize(tris.size() * 3);
    for (size_t k = 0; k < tris.size(); ++k) {
        out.tris[3 * k] = make_float4(tris[k].v0[0], tris[k].v0[1], tris[k].v0[2], bits_of(tris[k].prim));
        out.tris[3 * k + 1] = make_float4(tris[k].e1[0], tris[k].e1[1], tris[k].e1[2], bits_of(tris[k].gid));
        out.tris[3 * k + 2] = make_float4(tris[k].e2[0], tris[k].e2[1], tris[k].e2[2], bits_of(tris[k].ord));
    }
    for (int ax = 0; ax < 3; ++ax) {
        out.center[ax] = 0.5f * (nodes[0].lo[ax] + nodes[0].hi[ax]);
        out.half[ax] = 0.5f * (nodes[0].hi[ax] - nodes[0].lo[ax]);
    }
}

}  // namespace srt
