// srt_mesh_bvh.h — host-side flattening of SRT_OBJ_MESH objects into one world-space
// triangle list + an 8-wide BVH (EXTENSION: the reference has no triangle primitive).
//
// Device layout (HBM, read through L2; too large for LDS at 100k triangles):
//   tris : 3 float4 per triangle, in BVH leaf order
//            (v0.xyz, bits(primitive id p))      p indexes the LDS material table
//            (e1.xyz, bits(global triangle id))  id = position in (object list order, triangle
//            (e2.xyz, bits(list index))               index) — restores the tie rule
//   nodes: 5 float4 (80 B) per 8-WIDE inner node, child boxes quantized to 8 bits on the node's own box:
//            (origin.xyz, bits(ex | ey<<8 | ez<<16 | innermask<<24))   cell size per axis = 2^(e-127)
//            (bits(first inner child), bits(first leaf triangle), bits(leafmask | counts<<8), 0)
//            (lo.x[0..3], lo.x[4..7], lo.y[0..3], lo.y[4..7])   one byte per child
//            (lo.z[0..3], lo.z[4..7], hi.x[0..3], hi.x[4..7])
//            (hi.y[0..3], hi.y[4..7], hi.z[0..3], hi.z[4..7])
//          child box = origin + q*cell, lo rounded down / hi up, so it encloses the exact box.
//          Children are IMPLICIT: nodes are numbered breadth-first, so the inner children of a node are
//          consecutive — child c (bit c of innermask) is node  first_inner + popcount(innermask & ((1<<c)-1));
//          triangles are stored in the same breadth-first order, so the leaf children of a node are
//          consecutive too — child c (bit c of leafmask) holds  1 + ((counts >> 2c) & 3)  triangles starting at
//          first_leaf_triangle + the sum of the counts of the leaf children before it.  A child in neither
//          mask is absent (its box is inverted).  Node 0 is the root; a mesh of <= 4 triangles is a root
//          with one leaf child.
// Build: binary binned surface-area heuristic (16 bins per axis, median fallback), leaves of <= 4
// triangles, then collapsed to 8 children per node — deterministic.  Bounds are exact (float min/max of the float vertices); the
// kernel pads them per ray (see closest_hit) so that the box filter is conservative with
// respect to the rounding of the triangle test.
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <stdexcept>
#include <vector>

#include "srt_pathtrace.h"

namespace srt {

struct HostMesh {
    std::vector<float> vertices;
    std::vector<uint32_t> indices;
};

constexpr int NODE_VEC4 = 5;  // float4 per node (80 B)

struct MeshImage {
    std::vector<float4> tris, nodes;
    std::vector<int32_t> gidpos;  // global triangle id -> position in `tris` (leaf order)
    int n_tris = 0, n_nodes = 0, n_mesh_objects = 0, max_depth = 0;
    float center[3] = {0, 0, 0}, half[3] = {0, 0, 0};  // root box
    float bs_radius = 0;  // radius of a sphere around `center` that contains every triangle (rounded up)
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
    // Depth-first with an explicit stack (an adversarial mesh — geometrically spaced centroids — lets the
    // binned SAH peel one triangle off per level, so recursion depth would be O(triangles)): the left child
    // of node `me` is me + 1, the right child is numbered when the left subtree is complete.  Beyond
    // SAH_DEPTH levels only median splits are made, which bounds the depth by SAH_DEPTH + log2(n).
    constexpr int SAH_DEPTH = 40;
    struct Frame {
        int b, e, depth, parent;  // parent >= 0: this is that node's right child
    };
    std::vector<Frame> stack;
    stack.push_back(Frame{0, (int)tris.size(), 1, -1});
    int max_depth = 0;
    while (!stack.empty()) {
        const Frame f = stack.back();
        stack.pop_back();
        const int b = f.b, e = f.e, depth = f.depth;
        const int me = (int)nodes.size();
        if (f.parent >= 0) nodes[(size_t)f.parent].a = me;
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
            nodes[(size_t)me] = nd;
            continue;
        }
        // binned surface-area heuristic (16 bins per axis); falls back to the median of the
        // longest axis when no split is cheaper (leaves must not exceed 4 triangles)
        auto area = [](const float* lo, const float* hi) {
            double dx = (double)hi[0] - lo[0], dy = (double)hi[1] - lo[1], dz = (double)hi[2] - lo[2];
            return dx < 0 ? 0.0 : 2.0 * (dx * dy + dy * dz + dz * dx);
        };
        constexpr int NB = 16;
        int best_axis = -1, best_bin = -1;
        double best_cost = 1e300;
        for (int ax = 0; ax < 3 && depth <= SAH_DEPTH; ++ax) {
            const double ext = (double)chi[ax] - clo[ax];
            if (!(ext > 0)) continue;
            float blo[NB][3], bhi[NB][3];
            int bcnt[NB];
            for (int q = 0; q < NB; ++q) {
                bcnt[q] = 0;
                for (int a2 = 0; a2 < 3; ++a2) blo[q][a2] = INFINITY, bhi[q][a2] = -INFINITY;
            }
            for (int k = b; k < e; ++k) {
                int q = (int)(((double)tris[k].c[ax] - clo[ax]) / ext * NB);
                q = q < 0 ? 0 : (q >= NB ? NB - 1 : q);
                bcnt[q]++;
                for (int a2 = 0; a2 < 3; ++a2) {
                    blo[q][a2] = std::min(blo[q][a2], tris[k].lo[a2]);
                    bhi[q][a2] = std::max(bhi[q][a2], tris[k].hi[a2]);
                }
            }
            double rarea[NB];
            int rcnt[NB];
            {
                float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
                int c = 0;
                for (int q = NB - 1; q >= 1; --q) {
                    for (int a2 = 0; a2 < 3; ++a2) lo[a2] = std::min(lo[a2], blo[q][a2]), hi[a2] = std::max(hi[a2], bhi[q][a2]);
                    c += bcnt[q];
                    rarea[q] = area(lo, hi);
                    rcnt[q] = c;
                }
            }
            float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
            int c = 0;
            for (int q = 0; q < NB - 1; ++q) {  // split between bin q and q + 1
                for (int a2 = 0; a2 < 3; ++a2) lo[a2] = std::min(lo[a2], blo[q][a2]), hi[a2] = std::max(hi[a2], bhi[q][a2]);
                c += bcnt[q];
                if (c == 0 || rcnt[q + 1] == 0) continue;
                const double cost = area(lo, hi) * c + rarea[q + 1] * rcnt[q + 1];
                if (cost < best_cost) best_cost = cost, best_axis = ax, best_bin = q;
            }
        }
        int mid = b;
        if (best_axis >= 0) {
            const int ax = best_axis;
            const double ext = (double)chi[ax] - clo[ax], lo0 = clo[ax];
            const int bin = best_bin;
            auto it = std::stable_partition(tris.begin() + b, tris.begin() + e, [=](const Tri& x) {
                int q = (int)(((double)x.c[ax] - lo0) / ext * NB);
                q = q < 0 ? 0 : (q >= NB ? NB - 1 : q);
                return q <= bin;
            });
            mid = (int)(it - tris.begin());
        }
        if (best_axis < 0 || mid == b || mid == e) {  // degenerate (or very deep): median of the longest axis
            int axis = 0;
            if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
            if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
            mid = (b + e) / 2;
            std::nth_element(tris.begin() + b, tris.begin() + mid, tris.begin() + e, [axis](const Tri& x, const Tri& y) {
                return x.c[axis] < y.c[axis] || (x.c[axis] == y.c[axis] && x.gid < y.gid);
            });
        }
        nd.a = -1;  // set when the right child is numbered
        nd.b = 0;
        nodes[(size_t)me] = nd;
        stack.push_back(Frame{mid, e, depth + 1, me});  // right: after the whole left subtree
        stack.push_back(Frame{b, mid, depth + 1, -1});  // left = me + 1
    }
    out.max_depth = max_depth;

    // second pass: collapse the binary tree into 8-wide nodes so that the expected number of wide nodes a random
    // ray visits — the sum of the wide nodes' surface areas — is smallest (the dynamic programme of Ylitie et al.,
    // "Efficient incoherent ray traversal on GPUs through compressed wide BVHs", 2017, with our fixed leaves):
    //   F[n][i] = cheapest way to hang the subtree of binary node n below a wide node using <= i of its child slots
    //           = 0                                              n a leaf (its triangles cost the same in every collapse)
    //           = min(area(n) + D[n][8],  D[n][i])               n inner: its own wide node, or dissolved (i >= 2)
    //   D[n][i] = min over k of F[left][k] + F[right][i - k]     the slots split between n's two children
    // Nodes are in depth-first preorder (children after their parent), so one backward sweep fills the tables.  The greedy
    // collapse used before (keep expanding the child with the largest area) left the 224 x 224 sphere of BASELINE configs
    // 4-5 with 8499 wide nodes of 4.5 children on average, 8 levels deep.
    struct Wide {
        int child[8];  // binary node indices
        int n;
    };
    std::vector<Wide> wide;
    auto area_of = [&](int k) {
        const double dx = (double)nodes[k].hi[0] - nodes[k].lo[0], dy = (double)nodes[k].hi[1] - nodes[k].lo[1],
                     dz = (double)nodes[k].hi[2] - nodes[k].lo[2];
        return 2.0 * (dx * dy + dy * dz + dz * dx);
    };
    struct Plan {
        double F[9];  // F[1..8]
        double D[9];  // D[2..8]
    };
    std::vector<Plan> plan(nodes.size());
    for (size_t kk = nodes.size(); kk-- > 0;) {
        Plan& pl = plan[kk];
        if (nodes[kk].b > 0) {
            for (int i = 0; i <= 8; ++i) pl.F[i] = 0.0, pl.D[i] = 0.0;
            continue;
        }
        const Plan &pa = plan[kk + 1], &pb = plan[(size_t)nodes[kk].a];
        pl.D[0] = pl.D[1] = 1e300;
        for (int i = 2; i <= 8; ++i) {
            double best = 1e300;
            for (int k = 1; k < i; ++k) best = std::min(best, pa.F[k] + pb.F[i - k]);
            pl.D[i] = best;
        }
        const double own = area_of((int)kk) + pl.D[8];
        pl.F[0] = 1e300;
        pl.F[1] = own;
        for (int i = 2; i <= 8; ++i) pl.F[i] = std::min(own, pl.D[i]);
    }
    // children of the wide node that stands for binary inner node k: k's subtree dissolved into <= 8 slots, in spatial
    // (left to right) order.  Explicit stack; a slot count of 1 or a cheaper own node keeps an inner node whole.
    auto children_of = [&](int k, Wide& w) {
        struct Item {
            int node, slots;
            bool dissolve;
        };
        Item st[32];
        int sp = 0;
        w.n = 0;
        st[sp++] = Item{k, 8, true};
        while (sp > 0) {
            const Item it = st[--sp];
            const Node& nd = nodes[(size_t)it.node];
            const Plan& pl = plan[(size_t)it.node];
            if (nd.b > 0 || (!it.dissolve && (it.slots == 1 || pl.F[1] < pl.D[it.slots]))) {
                w.child[w.n++] = it.node;
                continue;
            }
            const Plan &pa = plan[(size_t)it.node + 1], &pb = plan[(size_t)nd.a];
            int bk = 1;
            double best = 1e300;
            for (int q = 1; q < it.slots; ++q)
                if (pa.F[q] + pb.F[it.slots - q] < best) best = pa.F[q] + pb.F[it.slots - q], bk = q;
            st[sp++] = Item{nd.a, it.slots - bk, false};  // right: after the left one (LIFO)
            st[sp++] = Item{it.node + 1, bk, false};
        }
    };
    std::vector<int> wide_of(nodes.size(), -1);  // binary inner node -> wide node that expands it
    std::vector<std::pair<int, int>> todo;       // (binary node, depth), breadth-first so that siblings are neighbours
    if (nodes[0].b > 0) {                        // the whole mesh is one leaf
        Wide w;
        w.child[0] = 0;
        w.n = 1;
        wide.push_back(w);
        out.max_depth = 1;
    } else {
        wide_of[0] = 0;
        wide.push_back(Wide());
        todo.push_back({0, 1});
        out.max_depth = 1;
        for (size_t q = 0; q < todo.size(); ++q) {
            const int k = todo[q].first, depth = todo[q].second;
            Wide w;
            children_of(k, w);
            for (int c = 0; c < w.n; ++c)
                if (nodes[w.child[c]].b <= 0) {
                    wide_of[w.child[c]] = (int)wide.size();
                    wide.push_back(Wide());
                    todo.push_back({w.child[c], depth + 1});
                    out.max_depth = std::max(out.max_depth, depth + 1);
                }
            wide[(size_t)wide_of[k]] = w;
        }
    }
    // triangle storage order: breadth-first like the nodes — for every wide node, the triangles of its leaf
    // children in child order — so that a node needs one "first leaf triangle" instead of a reference per child
    std::vector<int32_t> newpos(tris.size(), -1);  // position in `tris` (build order) -> position in the output
    std::vector<int32_t> first_tri(wide.size(), 0), first_inner(wide.size(), 0);
    {
        int32_t next = 0;
        for (size_t k = 0; k < wide.size(); ++k) {
            first_tri[k] = next;
            first_inner[k] = 0;
            for (int c = 0; c < wide[k].n; ++c) {
                const Node& ch = nodes[wide[k].child[c]];
                if (ch.b > 0)
                    for (int t = 0; t < ch.b; ++t) newpos[(size_t)(ch.a + t)] = next++;
                else if (first_inner[k] == 0)
                    first_inner[k] = wide_of[wide[k].child[c]];
            }
        }
    }
    // quantize: child boxes on a 256^3 grid spanned by the node's own box.  origin = node.lo (float),
    // cell = 2^e per axis (the smallest power of two with 255 cells >= extent); lo is rounded down,
    // hi up, in exact double arithmetic, so  origin + q*cell  (as real numbers) encloses the child.
    out.n_nodes = (int)wide.size();
    out.nodes.assign(wide.size() * NODE_VEC4, make_float4(0, 0, 0, 0));
    for (size_t k = 0; k < wide.size(); ++k) {
        const Wide& w = wide[k];
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int c = 0; c < w.n; ++c)
            for (int ax = 0; ax < 3; ++ax) {
                lo[ax] = std::min(lo[ax], nodes[w.child[c]].lo[ax]);
                hi[ax] = std::max(hi[ax], nodes[w.child[c]].hi[ax]);
            }
        uint32_t expo[3];
        double cell[3];
        for (int ax = 0; ax < 3; ++ax) {
            const double ext = (double)hi[ax] - (double)lo[ax];
            int e = -126;
            while (e < 127 && ldexp(255.0, e) < ext) ++e;
            expo[ax] = (uint32_t)(e + 127);
            cell[ax] = ldexp(1.0, e);
        }
        uint8_t q[6][8];
        uint32_t innermask = 0, leafmask = 0, counts = 0;
        int expect_inner = first_inner[k];
        for (int c = 0; c < 8; ++c) {
            if (c >= w.n) {  // absent child: inverted box, in neither mask
                for (int ax = 0; ax < 3; ++ax) q[ax][c] = 255, q[3 + ax][c] = 0;
                continue;
            }
            const Node& ch = nodes[w.child[c]];
            for (int ax = 0; ax < 3; ++ax) {
                double ql = floor(((double)ch.lo[ax] - (double)lo[ax]) / cell[ax]);
                double qh = ceil(((double)ch.hi[ax] - (double)lo[ax]) / cell[ax]);
                ql = ql < 0 ? 0 : (ql > 255 ? 255 : ql);
                qh = qh < 0 ? 0 : (qh > 255 ? 255 : qh);
                q[ax][c] = (uint8_t)ql;
                q[3 + ax][c] = (uint8_t)qh;
            }
            if (ch.b > 0) {
                leafmask |= 1u << c;
                counts |= (uint32_t)(ch.b - 1) << (2 * c);
            } else {
                innermask |= 1u << c;
                if (wide_of[w.child[c]] != expect_inner++) throw std::logic_error("BVH: inner children are not consecutive");
            }
        }
        auto pack4 = [&](int row, int first) {
            return bits_of((int32_t)((uint32_t)q[row][first] | ((uint32_t)q[row][first + 1] << 8) | ((uint32_t)q[row][first + 2] << 16) |
                                     ((uint32_t)q[row][first + 3] << 24)));
        };
        float4* nd = &out.nodes[NODE_VEC4 * k];
        nd[0] = make_float4(lo[0], lo[1], lo[2], bits_of((int32_t)(expo[0] | (expo[1] << 8) | (expo[2] << 16) | (innermask << 24))));
        nd[1] = make_float4(bits_of(first_inner[k]), bits_of(first_tri[k]), bits_of((int32_t)(leafmask | (counts << 8))), 0.0f);
        nd[2] = make_float4(pack4(0, 0), pack4(0, 4), pack4(1, 0), pack4(1, 4));  // lo.x[0..7], lo.y[0..7]
        nd[3] = make_float4(pack4(2, 0), pack4(2, 4), pack4(3, 0), pack4(3, 4));  // lo.z, hi.x
        nd[4] = make_float4(pack4(4, 0), pack4(4, 4), pack4(5, 0), pack4(5, 4));  // hi.y, hi.z
    }
    out.tris.resize(tris.size() * 3);
    out.gidpos.assign(tris.size(), 0);
    for (size_t k = 0; k < tris.size(); ++k) {
        const size_t pos = (size_t)newpos[k];
        out.gidpos[(size_t)tris[k].gid] = (int32_t)pos;
        out.tris[3 * pos] = make_float4(tris[k].v0[0], tris[k].v0[1], tris[k].v0[2], bits_of(tris[k].prim));
        out.tris[3 * pos + 1] = make_float4(tris[k].e1[0], tris[k].e1[1], tris[k].e1[2], bits_of(tris[k].gid));
        out.tris[3 * pos + 2] = make_float4(tris[k].e2[0], tris[k].e2[1], tris[k].e2[2], bits_of(tris[k].ord));
    }
    for (int ax = 0; ax < 3; ++ax) out.center[ax] = 0.5f * (nodes[0].lo[ax] + nodes[0].hi[ax]);
    {  // bounding sphere around the root box's centre: max vertex distance in double, inflated and rounded up
        double r2 = 0;
        for (const Tri& t : tris)
            for (int q = 0; q < 3; ++q) {
                double d2 = 0;
                for (int ax = 0; ax < 3; ++ax) {
                    const double v = (double)t.v0[ax] + (q == 1 ? (double)t.e1[ax] : q == 2 ? (double)t.e2[ax] : 0.0);
                    d2 += (v - (double)out.center[ax]) * (v - (double)out.center[ax]);
                }
                r2 = std::max(r2, d2);
            }
        // (v0 + e1 in double differs from the float vertex by <= 1 ulp of the coordinate: covered by the 1e-5 inflation)
        const double r = sqrt(r2) * (1.0 + 1e-5) + 1e-30;
        out.bs_radius = (float)r;
        if ((double)out.bs_radius < r) out.bs_radius = nextafterf(out.bs_radius, INFINITY);
    }
    for (int ax = 0; ax < 3; ++ax) {
        // half extent rounded up so that [center - half, center + half] contains the root box
        float h = std::max(nodes[0].hi[ax] - out.center[ax], out.center[ax] - nodes[0].lo[ax]);
        out.half[ax] = nextafterf(h * 1.000001f, INFINITY);
    }
}

}  // namespace srt
