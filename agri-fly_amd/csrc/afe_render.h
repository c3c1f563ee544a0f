// afe_render.h -- structures shared by the depth-camera kernels and their host side.
#pragma once
#include <stdint.h>

#include "../../include/agrifly_engine.h"

namespace afe {

// The builder's binary tree (host only).
struct BvhNode {
  float lo[3];
  int32_t a;   // leaf: first triangle (leaf order); inner: left child, right child = a + 1
  float hi[3];
  int32_t b;   // leaf: triangle count (> 0); inner: minus the node's depth (root = -1)
};

// What the kernel walks: one record per INNER node of the builder's tree, carrying the boxes of both
// children, so that a visit is one 64-byte scalar load (s_load_dwordx16) and decides about both
// children at once; leaves have no record of their own -- the parent names their triangles.
struct PairNode {
  float box_l[6];           // left child's box (inflated, see Builder::set_bounds): {lo, hi} per axis
  float box_r[6];           // right child's box
  uint32_t left, right;     // inner child: byte offset of its PairNode in the array; leaf child: first triangle (leaf order)
  uint32_t meta;            // bits 0-1 split axis (left = lower side), 8-15 / 16-23 triangle count of the
                            // left / right child (0 = inner), 24-31 depth of this node (root = 1)
  uint32_t pad;
};
static_assert(sizeof(PairNode) == 64, "one visit = one 64-byte scalar load");

// implemented in afe_engine.cpp: the stream the engine launches on and its device
void engine_stream_device(afe_engine *e, void **stream, int *device);
// afe_get_device_view for the library's own consumers (the slabs do not leave the engine: afe_sync keeps its short form)
int engine_device_view(afe_engine *e, struct afe_device_view *out);

}  // namespace afe
