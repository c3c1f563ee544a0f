// afe_render.h -- structures shared by the depth-camera kernels and their host side.
#pragma once
#include <stdint.h>

#include "../../include/agrifly_engine.h"

namespace afe {

// 32 bytes; sibling pairs are adjacent, so one inner-node visit reads one 64-byte line.
struct BvhNode {
  float lo[3];
  int32_t a;   // leaf: first triangle (leaf order); inner: left child, right child = a + 1
  float hi[3];
  int32_t b;   // leaf: triangle count (> 0); inner: minus the node's depth (root = -1)
};

// implemented in afe_engine.cpp: the stream the engine launches on and its device
void engine_stream_device(afe_engine *e, void **stream, int *device);

}  // namespace afe
