// bvh_build.cpp -- see bvh_build.h.  Host code, f64, runs once per scene.

#include "bvh_build.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace rt {

namespace {

constexpr size_t kMaxShapes = 4;  // BVHNode::MAX_SHAPES, geometry/bvh.rs:34

void combine(const double amin[3], const double amax[3], const double bmin[3], const double bmax[3],
             double omin[3], double omax[3]) {  // BBox::combine, geometry/bbox.rs:23-36
    for (int k = 0; k < 3; ++k) {
        omin[k] = std::fmin(amin[k], bmin[k]);
        omax[k] = std::fmax(amax[k], bmax[k]);
    }
}

struct Builder {
    std::vector<BuildItem> &items;
    Tree &tree;

    // Builds the subtree over items[lo, hi) and returns its node index.
    int32_t build(size_t lo, size_t hi, int axis, int inner_above) {
        const int32_t me = (int32_t)tree.nodes.size();
        tree.nodes.emplace_back();
        if (hi - lo <= kMaxShapes) {
            // leaf: shapes stay in incoming order; box = left fold of combine (bvh.rs:38-57)
            double bmin[3], bmax[3];
            std::memcpy(bmin, items[lo].bmin, sizeof(bmin));
            std::memcpy(bmax, items[lo].bmax, sizeof(bmax));
            for (size_t i = lo + 1; i < hi; ++i) combine(bmin, bmax, items[i].bmin, items[i].bmax, bmin, bmax);
            TreeNode &n = tree.nodes[(size_t)me];
            std::memcpy(n.bmin, bmin, sizeof(bmin));
            std::memcpy(n.bmax, bmax, sizeof(bmax));
            n.first = (uint32_t)tree.order.size();
            n.count = (uint32_t)(hi - lo);
            for (size_t i = lo; i < hi; ++i) tree.order.push_back(items[i].id);
            tree.inner_depth = std::max(tree.inner_depth, inner_above);
            return me;
        }
        // stable sort on the centroid coordinate of the current axis (bvh.rs:60-68)
        std::stable_sort(items.begin() + (long)lo, items.begin() + (long)hi,
                         [axis](const BuildItem &p, const BuildItem &q) { return p.centroid[axis] < q.centroid[axis]; });
        const size_t mid = lo + (hi - lo) / 2;  // bvh.rs:71-72
        const int next = (axis + 1) % 3;        // bvh.rs:15-21
        const int32_t l = build(lo, mid, next, inner_above + 1);
        const int32_t r = build(mid, hi, next, inner_above + 1);
        TreeNode &n = tree.nodes[(size_t)me];
        n.left = l;
        n.right = r;
        n.axis = axis;
        combine(tree.nodes[(size_t)l].bmin, tree.nodes[(size_t)l].bmax, tree.nodes[(size_t)r].bmin,
                tree.nodes[(size_t)r].bmax, n.bmin, n.bmax);
        return me;
    }
};

// Binned surface-area-heuristic builder for the fast mode's trees.  Any tree over the same
// items gives the same closest hit; this one is built to be cheap to traverse, not to match the
// reference's traversal order.
struct SahBuilder {
    std::vector<BuildItem> &items;
    Tree &tree;
    size_t max_leaf;
    double node_cost = 1.0;

    static double half_area(const double mn[3], const double mx[3]) {
        const double dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        return dx * dy + dy * dz + dz * dx;
    }

    void leaf(int32_t me, size_t lo, size_t hi, int inner_above) {
        TreeNode &n = tree.nodes[(size_t)me];
        n.first = (uint32_t)tree.order.size();
        n.count = (uint32_t)(hi - lo);
        for (size_t i = lo; i < hi; ++i) tree.order.push_back(items[i].id);
        tree.inner_depth = std::max(tree.inner_depth, inner_above);
    }

    int32_t build(size_t lo, size_t hi, int inner_above) {
        const int32_t me = (int32_t)tree.nodes.size();
        tree.nodes.emplace_back();
        double bmin[3], bmax[3], cmin[3], cmax[3];
        for (int k = 0; k < 3; ++k) { bmin[k] = cmin[k] = INFINITY; bmax[k] = cmax[k] = -INFINITY; }
        for (size_t i = lo; i < hi; ++i)
            for (int k = 0; k < 3; ++k) {
                bmin[k] = std::fmin(bmin[k], items[i].bmin[k]);
                bmax[k] = std::fmax(bmax[k], items[i].bmax[k]);
                cmin[k] = std::fmin(cmin[k], items[i].centroid[k]);
                cmax[k] = std::fmax(cmax[k], items[i].centroid[k]);
            }
        std::memcpy(tree.nodes[(size_t)me].bmin, bmin, sizeof(bmin));
        std::memcpy(tree.nodes[(size_t)me].bmax, bmax, sizeof(bmax));
        const size_t n = hi - lo;
        if (n <= 1) { leaf(me, lo, hi, inner_above); return me; }

        constexpr int kBins = 32;
        int best_axis = -1, best_bin = -1;
        double best_cost = INFINITY;
        for (int axis = 0; axis < 3; ++axis) {
            const double ext = cmax[axis] - cmin[axis];
            if (!(ext > 0.0) || !std::isfinite(ext)) continue;
            struct Bin { double mn[3], mx[3]; size_t count; } bins[kBins];
            for (auto &b : bins) { for (int k = 0; k < 3; ++k) { b.mn[k] = INFINITY; b.mx[k] = -INFINITY; } b.count = 0; }
            const double scale = kBins / ext;
            for (size_t i = lo; i < hi; ++i) {
                int b = (int)((items[i].centroid[axis] - cmin[axis]) * scale);
                b = std::min(std::max(b, 0), kBins - 1);
                bins[b].count++;
                for (int k = 0; k < 3; ++k) {
                    bins[b].mn[k] = std::fmin(bins[b].mn[k], items[i].bmin[k]);
                    bins[b].mx[k] = std::fmax(bins[b].mx[k], items[i].bmax[k]);
                }
            }
            double right_area[kBins];
            size_t right_count[kBins];
            double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
            size_t cnt = 0;
            for (int b = kBins - 1; b > 0; --b) {
                for (int k = 0; k < 3; ++k) { mn[k] = std::fmin(mn[k], bins[b].mn[k]); mx[k] = std::fmax(mx[k], bins[b].mx[k]); }
                cnt += bins[b].count;
                right_area[b] = cnt ? half_area(mn, mx) : 0.0;
                right_count[b] = cnt;
            }
            for (int k = 0; k < 3; ++k) { mn[k] = INFINITY; mx[k] = -INFINITY; }
            cnt = 0;
            for (int b = 0; b + 1 < kBins; ++b) {  // split between bin b and b + 1
                for (int k = 0; k < 3; ++k) { mn[k] = std::fmin(mn[k], bins[b].mn[k]); mx[k] = std::fmax(mx[k], bins[b].mx[k]); }
                cnt += bins[b].count;
                if (cnt == 0 || right_count[b + 1] == 0) continue;
                const double cost = half_area(mn, mx) * (double)cnt + right_area[b + 1] * (double)right_count[b + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
            }
        }
        // leaf if allowed and no split is cheaper than testing the items here (node test = 1 item test)
        if (n <= max_leaf) {
            const double here = half_area(bmin, bmax) * (double)n;
            if (best_axis < 0 || best_cost + node_cost * half_area(bmin, bmax) >= here) { leaf(me, lo, hi, inner_above); return me; }
        }
        size_t mid;
        int axis;
        if (best_axis >= 0) {
            axis = best_axis;
            const double scale = kBins / (cmax[axis] - cmin[axis]), c0 = cmin[axis];
            const int bb = best_bin;
            auto it = std::stable_partition(items.begin() + (long)lo, items.begin() + (long)hi, [=](const BuildItem &q) {
                int b = (int)((q.centroid[axis] - c0) * scale);
                b = std::min(std::max(b, 0), kBins - 1);
                return b <= bb;
            });
            mid = (size_t)(it - items.begin());
        } else {  // all centroids coincide: halve in the given order
            axis = 0;
            mid = lo + n / 2;
        }
        const int32_t l = build(lo, mid, inner_above + 1);
        const int32_t r = build(mid, hi, inner_above + 1);
        TreeNode &nd = tree.nodes[(size_t)me];
        nd.left = l;
        nd.right = r;
        nd.axis = axis;
        return me;
    }
};

// f64 box -> f32 box.  A strictly positive extent is rounded outward so the
// fp32 box never excludes a point the f64 box contains; a zero extent stays
// zero, because the reference never hits a flat box (bbox.rs:89,104,119) and a
// padded one would become visible.
void box_to_f32(const double bmin[3], const double bmax[3], float omin[3], float omax[3]) {
    for (int k = 0; k < 3; ++k) {
        float lo = (float)bmin[k], hi = (float)bmax[k];
        if (bmin[k] < bmax[k]) {
            if ((double)lo > bmin[k]) lo = std::nextafterf(lo, -INFINITY);
            if ((double)hi < bmax[k]) hi = std::nextafterf(hi, INFINITY);
        }
        omin[k] = lo;
        omax[k] = hi;
    }
}

// `flat_pad` > 0 widens a zero extent by that much on both sides (the fast mode's own scene-level
// tree, where a flat box must stay enterable; see flatten_scene).
void set_child_box(RtNode &rec, bool right, const double bmin[3], const double bmax[3], double flat_pad = 0.0) {
    float lo[3], hi[3];
    double pmin[3], pmax[3];
    for (int k = 0; k < 3; ++k) {
        const bool flat = flat_pad > 0.0 && !(bmin[k] < bmax[k]);
        pmin[k] = flat ? bmin[k] - flat_pad : bmin[k];
        pmax[k] = flat ? bmax[k] + flat_pad : bmax[k];
    }
    box_to_f32(pmin, pmax, lo, hi);
    float *x = right ? rec.rx : rec.lx, *y = right ? rec.ry : rec.ly, *z = right ? rec.rz : rec.lz;
    x[0] = lo[0]; x[1] = hi[0];
    y[0] = lo[1]; y[1] = hi[1];
    z[0] = lo[2]; z[1] = hi[2];
}

// The affine map of a triangle (RtTriXf), from its f64 vertices.
void triangle_map(const double *v, RtTriXf &out) {
    const double a[3] = {v[0], v[1], v[2]};
    const double e1[3] = {v[3] - a[0], v[4] - a[1], v[5] - a[2]};
    const double e2[3] = {v[6] - a[0], v[7] - a[1], v[8] - a[2]};
    const double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    const double nn = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
    if (!(nn > 0.0) || !std::isfinite(nn)) {
        for (int k = 0; k < 8; ++k) out.uv[k] = NAN;
        for (int k = 0; k < 4; ++k) out.n[k] = NAN;
        return;
    }
    // rows of [e1 e2 n]^-1:  (e2 x n)/nn,  (n x e1)/nn,  n/nn
    const double ru[3] = {(e2[1] * n[2] - e2[2] * n[1]) / nn, (e2[2] * n[0] - e2[0] * n[2]) / nn, (e2[0] * n[1] - e2[1] * n[0]) / nn};
    const double rv[3] = {(n[1] * e1[2] - n[2] * e1[1]) / nn, (n[2] * e1[0] - n[0] * e1[2]) / nn, (n[0] * e1[1] - n[1] * e1[0]) / nn};
    const double rn[3] = {n[0] / nn, n[1] / nn, n[2] / nn};
    float u[4], w[4];
    for (int k = 0; k < 3; ++k) {
        u[k] = (float)ru[k];
        w[k] = (float)rv[k];
        out.n[k] = (float)rn[k];
    }
    // offsets from the ROUNDED rows, so that the map is exact at vertex a for the fp32 rows
    u[3] = (float)-((double)u[0] * a[0] + (double)u[1] * a[1] + (double)u[2] * a[2]);
    w[3] = (float)-((double)w[0] * a[0] + (double)w[1] * a[1] + (double)w[2] * a[2]);
    out.n[3] = (float)-((double)out.n[0] * a[0] + (double)out.n[1] * a[1] + (double)out.n[2] * a[2]);
    for (int k = 0; k < 4; ++k) {
        out.uv[2 * k] = u[k];
        out.uv[2 * k + 1] = w[k];
    }
}

// Items (shapes or triangles) below node `t`.
uint32_t subtree_items(const Tree &tree, int32_t t) {
    const TreeNode &n = tree.nodes[(size_t)t];
    return n.count > 0 ? n.count : subtree_items(tree, n.left) + subtree_items(tree, n.right);
}

// Emits the subtree `t` of `tree` into out.nodes and returns the reference a
// parent stores for it.  `prim_base` is added to leaf offsets.
//
// `larger_left` (the fast mode's trees only; the reference's trees keep the reference's child order): the child
// that holds more items goes into the left slot.  Any-hit packets visit the left child first whatever the
// direction of their rays, and a shadow ray that is blocked at all is more likely to be blocked where there
// is more to be blocked by (measured on the bunny frame's shadow kernel: more items first 0.532 ms, larger
// surface first 0.544, nearer first 0.576).  Closest-hit traversal is unaffected: it goes by
// right_first_mask, which is mirrored with the children.
int32_t emit(const Tree &tree, int32_t t, uint32_t prim_base, std::vector<RtNode> &nodes, double flat_pad = 0.0,
             bool larger_left = false) {
    const TreeNode &n = tree.nodes[(size_t)t];
    if (n.count > 0) return rt_make_leaf_ref(prim_base + n.first, n.count);
    const int32_t me = (int32_t)nodes.size();
    nodes.emplace_back();
    const bool swap = larger_left && subtree_items(tree, n.right) > subtree_items(tree, n.left);
    const int32_t li = swap ? n.right : n.left, ri = swap ? n.left : n.right;
    const TreeNode &l = tree.nodes[(size_t)li];
    const TreeNode &r = tree.nodes[(size_t)ri];
    RtNode rec{};
    set_child_box(rec, false, l.bmin, l.bmax, flat_pad);
    set_child_box(rec, true, r.bmin, r.bmax, flat_pad);
    nodes[(size_t)me] = rec;
    const int32_t lref = emit(tree, li, prim_base, nodes, flat_pad, larger_left);
    const int32_t rref = emit(tree, ri, prim_base, nodes, flat_pad, larger_left);
    nodes[(size_t)me].left = lref;
    nodes[(size_t)me].right = rref;
    nodes[(size_t)me].axis = n.axis;
    // the builders put the smaller centroids into n.left: a ray that runs towards smaller coordinates on the
    // split axis reaches n.right first
    const int32_t right_first = n.axis == 0 ? 0xAA : (n.axis == 1 ? 0xCC : 0xF0);
    nodes[(size_t)me].right_first_mask = swap ? (right_first ^ 0xFF) : right_first;
    return me * (int32_t)sizeof(RtNode);
}

// Header node: its left child is the tree's root (with the root's own box, which
// the reference tests on entry, bvh.rs:95), its right child is absent.
int32_t emit_with_header(const Tree &tree, uint32_t prim_base, std::vector<RtNode> &nodes, double flat_pad = 0.0,
                         bool larger_left = false) {
    const int32_t header = (int32_t)nodes.size();
    nodes.emplace_back();
    RtNode rec{};
    set_child_box(rec, false, tree.nodes[0].bmin, tree.nodes[0].bmax, flat_pad);
    // the absent right child is a box of zero extent: entry == exit on every axis, so no ray enters
    // it under any form of the slab test (an inverted +-inf box would pass a min/max test)
    rec.rx[0] = rec.rx[1] = rec.ry[0] = rec.ry[1] = rec.rz[0] = rec.rz[1] = 0.0f;
    rec.right = RT_REF_NONE;
    nodes[(size_t)header] = rec;
    const int32_t root_ref = emit(tree, 0, prim_base, nodes, flat_pad, larger_left);
    nodes[(size_t)header].left = root_ref;
    return header * (int32_t)sizeof(RtNode);
}

void set3(double dst[3], const RtVec3 &v) {
    dst[0] = v.x;
    dst[1] = v.y;
    dst[2] = v.z;
}

// Bounding box and centroid of a shape as the reference's constructors compute them.
bool shape_bounds(const RtScene &scene, const RtShape &s, const std::vector<Tree> &blas, BuildItem &it) {
    switch (s.type) {
    case RT_SHAPE_SPHERE:  // sphere.rs:23-26, 50-52
        it.bmin[0] = s.p0.x - s.radius; it.bmin[1] = s.p0.y - s.radius; it.bmin[2] = s.p0.z - s.radius;
        it.bmax[0] = s.p0.x + s.radius; it.bmax[1] = s.p0.y + s.radius; it.bmax[2] = s.p0.z + s.radius;
        set3(it.centroid, s.p0);
        return true;
    case RT_SHAPE_BOX:  // cuboid.rs:18, 55-57; bbox.rs:18
        set3(it.bmin, s.p0);
        set3(it.bmax, s.p1);
        for (int k = 0; k < 3; ++k) it.centroid[k] = (it.bmin[k] + it.bmax[k]) * 0.5;
        return true;
    case RT_SHAPE_TRIANGLE: {  // triangle.rs:23-32, 58-60
        const double a[3] = {s.p0.x, s.p0.y, s.p0.z}, b[3] = {s.p1.x, s.p1.y, s.p1.z}, c[3] = {s.p2.x, s.p2.y, s.p2.z};
        for (int k = 0; k < 3; ++k) {
            it.bmin[k] = std::fmin(std::fmin(a[k], b[k]), c[k]);
            it.bmax[k] = std::fmax(std::fmax(a[k], b[k]), c[k]);
            it.centroid[k] = (a[k] + b[k] + c[k]) / 3.0;
        }
        return true;
    }
    case RT_SHAPE_MESH: {  // mesh.rs:59-60, 83-85
        if (s.mesh < 0 || (uint32_t)s.mesh >= scene.n_meshes) return false;
        const TreeNode &root = blas[(size_t)s.mesh].nodes[0];
        std::memcpy(it.bmin, root.bmin, sizeof(it.bmin));
        std::memcpy(it.bmax, root.bmax, sizeof(it.bmax));
        for (int k = 0; k < 3; ++k) it.centroid[k] = (it.bmin[k] + it.bmax[k]) * 0.5;
        return true;
    }
    case RT_SHAPE_INSTANCE:  // instance.rs:42, 67-69
        set3(it.bmin, s.bbox_min);
        set3(it.bmax, s.bbox_max);
        for (int k = 0; k < 3; ++k) it.centroid[k] = (it.bmin[k] + it.bmax[k]) * 0.5;
        return true;
    default: return false;
    }
}

// fp32 payload of a sphere / box / triangle.
void fill_payload(const RtShape &s, float f[12]) {
    std::memset(f, 0, 12 * sizeof(float));
    switch (s.type) {
    case RT_SHAPE_SPHERE:
        f[0] = (float)s.p0.x; f[1] = (float)s.p0.y; f[2] = (float)s.p0.z; f[3] = (float)s.radius;
        break;
    case RT_SHAPE_BOX:
        f[0] = (float)s.p0.x; f[1] = (float)s.p0.y; f[2] = (float)s.p0.z;
        f[3] = (float)s.p1.x; f[4] = (float)s.p1.y; f[5] = (float)s.p1.z;
        break;
    case RT_SHAPE_TRIANGLE:
        f[0] = (float)s.p0.x; f[1] = (float)s.p0.y; f[2] = (float)s.p0.z;
        f[3] = (float)(s.p0.x - s.p1.x); f[4] = (float)(s.p0.y - s.p1.y); f[5] = (float)(s.p0.z - s.p1.z);
        f[6] = (float)(s.p0.x - s.p2.x); f[7] = (float)(s.p0.y - s.p2.y); f[8] = (float)(s.p0.z - s.p2.z);
        break;
    default: break;
    }
}

}  // namespace

// By how much a box of zero thickness is widened on each side in the fast mode's trees: 1e-5 of
// the largest extent or coordinate of the bounds it lives in.
static double flat_pad_for(const double bmin[3], const double bmax[3]) {
    double pad = 0.0;
    for (int c = 0; c < 3; ++c)
        if (bmin[c] <= bmax[c] && std::isfinite(bmin[c]) && std::isfinite(bmax[c]))
            pad = std::fmax(pad, std::fmax(bmax[c] - bmin[c], std::fmax(std::fabs(bmin[c]), std::fabs(bmax[c]))));
    return pad > 0.0 ? pad * 1e-5 : 1e-5;
}

// 2 * 24 levels hold 2^24 triangles even when every split is 1 : 3; beyond that the mesh keeps the reference's tree
static const int kMaxFastMeshDepth = 48;

Tree build_tree_sah(std::vector<BuildItem> items, size_t max_leaf, double node_cost) {
    Tree tree;
    if (items.empty()) return tree;
    tree.nodes.reserve(items.size() * 2);
    tree.order.reserve(items.size());
    SahBuilder b{items, tree, max_leaf};
    b.node_cost = node_cost;
    b.build(0, items.size(), 0);
    return tree;
}

Tree build_tree(std::vector<BuildItem> items) {
    Tree tree;
    if (items.empty()) return tree;
    tree.nodes.reserve(items.size());
    tree.order.reserve(items.size());
    Builder b{items, tree};
    b.build(0, items.size(), 0, 0);
    return tree;
}

std::string flatten_scene(const RtScene &scene, FlatScene &out, bool reference_mesh_trees) {
    out = FlatScene();
    if (scene.n_shapes == 0) return "scene has no shapes";

    // ---- one tree per mesh (geometry/mesh.rs:39-59)
    uint64_t tri_total = 0;
    std::vector<double> mesh_pad;  // by how much a flat box of the fast mode's mesh tree is widened
    for (uint32_t m = 0; m < scene.n_meshes; ++m) {
        const RtMesh &mesh = scene.meshes[m];
        if (mesh.n_triangles == 0 || !mesh.vertices) return "mesh without triangles";
        out.mesh_tri_base.push_back(tri_total);
        tri_total += mesh.n_triangles;
        if (tri_total >= (1ull << 26)) return "too many mesh triangles (their records are addressed by 32-bit byte offsets)";
        std::vector<BuildItem> items((size_t)mesh.n_triangles);
        for (uint64_t i = 0; i < mesh.n_triangles; ++i) {
            const double *v = mesh.vertices + i * 9;
            BuildItem &it = items[(size_t)i];
            for (int k = 0; k < 3; ++k) {
                it.bmin[k] = std::fmin(std::fmin(v[k], v[3 + k]), v[6 + k]);
                it.bmax[k] = std::fmax(std::fmax(v[k], v[3 + k]), v[6 + k]);
                it.centroid[k] = (v[k] + v[3 + k] + v[6 + k]) / 3.0;
            }
            it.id = (int32_t)i;
        }
        out.blas.push_back(build_tree(items));
        // The fast mode's tree of the same mesh, split by surface area (leaves of up to 4).  The
        // median split leaves a few regions where a ray that runs along the surface passes through
        // hundreds of overlapping boxes (bunny frame: the worst wave visits 304 nodes against 16 on
        // average); those rays set the run time of small launches.  As for the scene-level tree, it
        // holds exactly the triangles the reference can reach -- those of reference leaves whose box
        // is not flat (bbox.rs:89,104,119) -- and its own flat boxes are widened.  `reference_mesh_trees`
        // keeps the reference's tree in fast mode too, for comparison; a tree too deep for the traversal
        // stack's budget (a degenerate mesh) is dropped the same way.
        const bool own = !reference_mesh_trees;
        Tree fast;
        double pad = 0.0;
        if (own) {
            const Tree &ref = out.blas.back();
            std::vector<BuildItem> reachable;
            for (const TreeNode &n : ref.nodes) {
                if (n.count == 0) continue;
                if (!(n.bmin[0] < n.bmax[0]) || !(n.bmin[1] < n.bmax[1]) || !(n.bmin[2] < n.bmax[2])) continue;
                for (uint32_t k = 0; k < n.count; ++k) reachable.push_back(items[(size_t)ref.order[n.first + k]]);
            }
            std::sort(reachable.begin(), reachable.end(), [](const BuildItem &a, const BuildItem &b) { return a.id < b.id; });
            pad = flat_pad_for(ref.nodes[0].bmin, ref.nodes[0].bmax);
            // node cost 2: a node step of the packet walk (one 64-byte record, two box tests, a ballot
            // and a branch) costs about two triangle tests; measured on the bunny frame, 2 to 8 give
            // the same 1.186 ms where 1 gives 1.213 and 0.5 gives 1.286
            if (!reachable.empty()) fast = build_tree_sah(std::move(reachable), 4, 2.0);
            if (fast.inner_depth > kMaxFastMeshDepth) fast = Tree();
        }
        out.mesh_tri_base_fast.push_back(out.mesh_tri_base.back());  // bases of the copies are set below
        out.blas_fast.push_back(std::move(fast));
        mesh_pad.push_back(pad);
    }
    for (uint32_t m = 0; m < scene.n_meshes; ++m) {
        if (out.blas_fast[m].nodes.empty()) continue;
        const uint64_t count = out.blas_fast[m].order.size();
        if (tri_total + count >= (1ull << 26)) {  // no room for a second copy: keep the reference's tree
            out.blas_fast[m] = Tree();
            continue;
        }
        out.mesh_tri_base_fast[m] = tri_total;
        tri_total += count;
    }

    // ---- scene-level tree (scene/mod.rs:699-700)
    std::vector<BuildItem> top((size_t)scene.n_shapes);
    for (uint32_t i = 0; i < scene.n_shapes; ++i) {
        const RtShape &s = scene.shapes[i];
        if (s.type == RT_SHAPE_INSTANCE) {
            if (s.proto < 0 || (uint32_t)s.proto >= scene.n_protos) return "instance refers to a missing prototype";
            const RtShape &p = scene.protos[s.proto];
            if (p.type == RT_SHAPE_INSTANCE) return "an instanced shape can not be type instance";
            if (p.type == RT_SHAPE_MESH && (p.mesh < 0 || (uint32_t)p.mesh >= scene.n_meshes))
                return "prototype refers to a missing mesh";
        }
        if (s.shader < 0 || (uint32_t)s.shader >= scene.n_shaders) return "shape refers to a missing shader";
        if (!shape_bounds(scene, s, out.blas, top[i])) return "malformed shape";
        top[i].id = (int32_t)i;
    }
    out.tlas = build_tree(top);

    // ---- the fast mode's own scene-level tree.  The closest hit does not depend on the tree, with
    // one exception that the reference's tree does decide: a box of zero thickness is never entered
    // (bbox.rs:89,104,119), so the shapes of a reference LEAF whose box is flat are invisible (a lone
    // axis-aligned triangle, say), while the same shape sharing a leaf with others is visible.  The
    // fast tree therefore holds exactly the shapes the reference can reach, its own flat boxes are
    // widened so that they can be entered, and within that it is free to split by surface area,
    // which suits shapes of very different sizes better than the median split does.
    std::vector<BuildItem> reachable;
    double scene_min[3] = {INFINITY, INFINITY, INFINITY}, scene_max[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (const TreeNode &n : out.tlas.nodes) {
        if (n.count == 0) continue;
        if (!(n.bmin[0] < n.bmax[0]) || !(n.bmin[1] < n.bmax[1]) || !(n.bmin[2] < n.bmax[2])) continue;
        for (uint32_t k = 0; k < n.count; ++k) {
            const BuildItem &it = top[(size_t)out.tlas.order[n.first + k]];
            reachable.push_back(it);
            for (int c = 0; c < 3; ++c) {
                if (std::isfinite(it.bmin[c])) scene_min[c] = std::fmin(scene_min[c], it.bmin[c]);
                if (std::isfinite(it.bmax[c])) scene_max[c] = std::fmax(scene_max[c], it.bmax[c]);
            }
        }
    }
    const double flat_pad = flat_pad_for(scene_min, scene_max);
    std::sort(reachable.begin(), reachable.end(), [](const BuildItem &a, const BuildItem &b) { return a.id < b.id; });
    const bool own_fast_tree = !reachable.empty();
    // node cost 1: spheres, boxes and instances cost more to test than a node (spheres_1K: 0.84 ms with 1, 0.88 with 2)
    if (own_fast_tree) out.tlas_fast = build_tree_sah(std::move(reachable), 2, 1.0);

    // ---- node array: scene-level tree first, then the meshes
    out.tlas_root = emit_with_header(out.tlas, 0, out.nodes);
    const uint32_t fast_base = (uint32_t)out.tlas.order.size();  // the fast tree's primitives follow the reference tree's
    out.tlas_root_fast = own_fast_tree ? emit_with_header(out.tlas_fast, fast_base, out.nodes, flat_pad, true) : out.tlas_root;
    std::vector<int32_t> blas_root(scene.n_meshes), blas_root_fast(scene.n_meshes);
    int blas_depth = 0, blas_depth_fast = 0;
    for (uint32_t m = 0; m < scene.n_meshes; ++m) {
        blas_root[m] = emit_with_header(out.blas[m], (uint32_t)out.mesh_tri_base[m], out.nodes);
        blas_depth = std::max(blas_depth, out.blas[m].inner_depth);
    }
    for (uint32_t m = 0; m < scene.n_meshes; ++m) {
        blas_root_fast[m] = blas_root[m];
        blas_depth_fast = std::max(blas_depth_fast, out.blas[m].inner_depth);
        if (!own_fast_tree || out.blas_fast[m].nodes.empty()) continue;  // only the fast scene-level tree's primitives point here
        blas_root_fast[m] = emit_with_header(out.blas_fast[m], (uint32_t)out.mesh_tri_base_fast[m], out.nodes, mesh_pad[m], true);
        blas_depth_fast = std::max(blas_depth_fast, out.blas_fast[m].inner_depth);
    }
    // pushes on one path: one per inner node with two live children, plus, when a
    // mesh or instance is entered from a leaf, the rest of that leaf and the exit marker
    if (out.nodes.size() >= (1u << 25)) return "too many tree nodes for a 31-bit byte offset";
    out.stack_depth = out.tlas.inner_depth + 2 + blas_depth + 1;
    out.stack_depth_fast = std::max(out.tlas.inner_depth, out.tlas_fast.inner_depth) + 2 + blas_depth_fast + 1;

    // ---- triangles in leaf order
    out.tris.resize((size_t)tri_total);
    out.tri_xf.resize((size_t)tri_total);
    out.verts64.assign((size_t)tri_total * 12, 0.0);
    for (uint32_t copy = 0; copy < 2; ++copy)
    for (uint32_t m = 0; m < scene.n_meshes; ++m) {
        const RtMesh &mesh = scene.meshes[m];
        const Tree &t = copy == 0 ? out.blas[m] : out.blas_fast[m];
        const size_t base = (size_t)(copy == 0 ? out.mesh_tri_base[m] : out.mesh_tri_base_fast[m]);
        if (copy == 1 && t.nodes.empty()) continue;
        for (size_t k = 0; k < t.order.size(); ++k) {
            const int32_t orig = t.order[k];
            const double *v = mesh.vertices + (size_t)orig * 9;
            RtTri &tri = out.tris[base + k];
            for (int c = 0; c < 3; ++c) {
                tri.a[c] = (float)v[c];
                tri.ab[c] = (float)(v[c] - v[3 + c]);
                tri.ac[c] = (float)(v[c] - v[6 + c]);
            }
            tri.orig = orig;
            tri.mesh = (int32_t)m;
            tri._pad = 0;
            triangle_map(v, out.tri_xf[base + k]);
            // f64 vertices and the unit normal normalize((b - a) x (c - a)) of triangle.rs:22, evaluated
            // with the operation order of the device's D3 helpers (no contraction on either side)
            double *w = &out.verts64[(base + k) * 12];
            std::memcpy(w, v, 9 * sizeof(double));
            const double e1[3] = {v[3] - v[0], v[4] - v[1], v[5] - v[2]}, e2[3] = {v[6] - v[0], v[7] - v[1], v[8] - v[2]};
            const double cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
            const double len = std::sqrt((cx * cx + cy * cy) + cz * cz);
            w[9] = cx / len; w[10] = cy / len; w[11] = cz / len;
        }
    }

    // ---- top-level primitives in leaf order: the reference tree's, then the fast tree's
    const size_t n_tprims = out.tlas.order.size() + out.tlas_fast.order.size();
    out.tprims.resize(n_tprims);
    out.tprim_xf.assign(n_tprims * 12, 0.0f);
    for (size_t k = 0; k < n_tprims; ++k) {
        const int32_t id = k < out.tlas.order.size() ? out.tlas.order[k] : out.tlas_fast.order[k - out.tlas.order.size()];
        const RtShape &s = scene.shapes[id];
        const std::vector<int32_t> &mesh_root = k < out.tlas.order.size() ? blas_root : blas_root_fast;
        RtTPrim &p = out.tprims[k];
        std::memset(&p, 0, sizeof(p));
        p.type = s.type;
        p.shape_id = id;
        p.blas_root = -1;
        p.proto_type = -1;
        if (s.type == RT_SHAPE_MESH) {
            p.blas_root = mesh_root[(size_t)s.mesh];
        } else if (s.type == RT_SHAPE_INSTANCE) {
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 4; ++c) out.tprim_xf[k * 12 + (size_t)r * 4 + c] = (float)s.inv_transform[r * 4 + c];
            const RtShape &proto = scene.protos[s.proto];
            p.proto_type = proto.type;
            if (proto.type == RT_SHAPE_MESH) {
                p.blas_root = mesh_root[(size_t)proto.mesh];
                // world box of the instance (instance.rs:42), for the fast mode's pre-test
                const double bmin[3] = {s.bbox_min.x, s.bbox_min.y, s.bbox_min.z};
                const double bmax[3] = {s.bbox_max.x, s.bbox_max.y, s.bbox_max.z};
                box_to_f32(bmin, bmax, p.f, p.f + 3);
                // the same box once more as per-axis (min, max) pairs, the operand shape of the packet walk's test
                for (int c = 0; c < 3; ++c) { p.f[6 + 2 * c] = p.f[c]; p.f[7 + 2 * c] = p.f[3 + c]; }
            } else {
                fill_payload(proto, p.f);
            }
        } else {
            fill_payload(s, p.f);
        }
    }
    return "";
}

}  // namespace rt
