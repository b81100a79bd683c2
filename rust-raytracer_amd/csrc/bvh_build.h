// bvh_build.h -- host-side BVH construction with the reference's split rule and
// flattening of a whole scene into the fp32 records of rt_device_types.h.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rt_scene.h"
#include "rt_device_types.h"

namespace rt {

struct BuildItem {
    double bmin[3], bmax[3], centroid[3];
    int32_t id;
};

struct TreeNode {
    double bmin[3], bmax[3];
    int32_t left = -1, right = -1;  // indices into Tree::nodes, -1 in a leaf
    uint32_t first = 0, count = 0;  // leaf: range of Tree::order
    int32_t axis = 0;               // inner node: the axis its items were sorted on (0 x, 1 y, 2 z)
};

struct Tree {
    std::vector<TreeNode> nodes;  // pre-order, root at 0
    std::vector<int32_t> order;   // item ids in leaf order
    int inner_depth = 0;          // most inner nodes on any root-to-leaf path
};

// BVHNode::new (geometry/bvh.rs:36-87) starting on the X axis (bvh.rs:136).
Tree build_tree(std::vector<BuildItem> items);

// Binned surface-area-heuristic tree with at most `max_leaf` items per leaf (not the reference's
// tree: only for traversal that does not depend on the reference's visiting order).  `node_cost` =
// cost of a node step in units of one item test: a group of <= max_leaf items stays a leaf unless
// splitting it saves more than that.
Tree build_tree_sah(std::vector<BuildItem> items, size_t max_leaf, double node_cost);

struct FlatScene {
    std::vector<RtNode> nodes;
    std::vector<RtTPrim> tprims;
    std::vector<float> tprim_xf;          // 12 floats per tprim: fp32 rows of an instance's inverse transform
    std::vector<RtTri> tris;
    std::vector<RtTriXf> tri_xf;          // same order as `tris`
    std::vector<uint64_t> mesh_tri_base;  // first entry of `tris` / `verts64` triangle per mesh
    std::vector<uint64_t> mesh_tri_base_fast;  // the same for the fast mode's copy of the mesh (leaf order of
                                          // `blas_fast`); equal to mesh_tri_base where the mesh has no such copy
    std::vector<double> verts64;          // 12 doubles per triangle, same order as `tris`: a, b, c and the
                                          // unit normal Triangle::new stores (triangle.rs:22)
    int32_t tlas_root = 0;                // header node of the scene-level tree
    int32_t tlas_root_fast = 0;           // header node of the fast mode's scene-level tree (its leaves index
                                          // the second part of `tprims`)
    int stack_depth = 0;                  // traversal-stack entries a ray can need at most in the reference's trees
    int stack_depth_fast = 0;             // the same in the fast mode's trees
    Tree tlas;                            // kept for structural tests
    Tree tlas_fast;
    std::vector<Tree> blas;
    std::vector<Tree> blas_fast;          // the fast mode's mesh trees; empty tree = the mesh keeps the reference's
};

// Returns "" on success, else a message.  `reference_mesh_trees`: the fast mode walks the
// reference's mesh trees instead of surface-area trees of its own (RT_UPLOAD_REFERENCE_MESH_TREES).
std::string flatten_scene(const RtScene &scene, FlatScene &out, bool reference_mesh_trees = false);

}  // namespace rt
