// afe_render.hip -- depth camera for the whole ensemble.
//
// Replaces the AirSim DepthVis RPC of Simulator/Rappids_Simulator/main.cpp:332-354 (and
// its image contract, :120-125,360): one ray per pixel, closest hit over a static
// triangle mesh, z-depth quantised to counts.  The reference has no renderer of its own;
// the test suite's CPU checker is a brute-force statement of the same contract and the
// kernel below performs the same fp64 operations per ray / triangle pair in the same order
// (no FMA contraction), so images are compared bit for bit.
//
// MI355X mapping
//   * one wave (64 lanes) per 8x8 pixel tile of one view, and the 64 rays walk the tree together:
//     one node index and one stack per wave (the stack in vector registers, one entry per lane);
//     a visit is ONE 64-byte scalar load carrying the boxes of both children (PairNode), triangles
//     (96 bytes: double edges + their own box) arrive by scalar loads too; box tests in fp32 against
//     conservatively inflated boxes, a slab = one packed FMA; triangle tests in fp64;
//   * eight copies of the tree, mirrored into each direction octant: a tile whose rays agree on
//     the direction signs (all but those on a coordinate plane through the camera) walks the copy in
//     which every component is positive -- near face = lo, nearer child = the one stored first,
//     9 vector instructions per box;
//   * the BVH and the triangles stay in HBM and are served from L2 / Infinity Cache; the block
//     index is remapped so that each XCD (own L2) renders a contiguous range of views;
//   * poses are prepared by a small kernel straight from the engine's state slabs, so a
//     closed loop never leaves the device: step -> render -> plan.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <list>
#include <mutex>
#include <vector>

#include "afe_render.h"
#include "afe_host.h"   // afe_dev_env

namespace afe {

// ---------------------------------------------------------------------------------------
// host: BVH construction (binned SAH, median fallback)
// ---------------------------------------------------------------------------------------
namespace {

struct Box {
  float lo[3], hi[3];
  void reset() {
    for (int k = 0; k < 3; k++) { lo[k] = std::numeric_limits<float>::infinity(); hi[k] = -lo[k]; }
  }
  void grow(const float *p) {
    for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); }
  }
  void grow(const Box &b) {
    for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); }
  }
  float half_area() const {
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
  }
};

struct Builder {
  const float *tri;              // n x 9
  std::vector<Box> tri_box;
  std::vector<float> centroid;   // n x 3
  std::vector<int32_t> order;    // permutation being partitioned
  std::vector<BvhNode> nodes;
  std::vector<uint8_t> node_axis;   // inner nodes: the axis the children were separated along (left = lower side)
  int max_depth = 0;
  bool median_only = false;
  int split_axis = 0;               // set by split()

  static constexpr int kBins = 16;
#ifndef AFE_BVH_LEAF
#define AFE_BVH_LEAF 4
#endif
  static constexpr int kLeaf = AFE_BVH_LEAF;

  Box range_box(int64_t first, int64_t count) const {
    Box b; b.reset();
    for (int64_t i = first; i < first + count; i++) b.grow(tri_box[order[i]]);
    return b;
  }

  void set_bounds(BvhNode &n, const Box &b) const {
    // inflate outward: the traversal's slab test may then round either way without ever
    // rejecting a box whose triangles the ray touches
    for (int k = 0; k < 3; k++) {
      const float eps = 3e-4f + 4e-6f * std::max(std::fabs(b.lo[k]), std::fabs(b.hi[k]));
      n.lo[k] = std::nextafterf(b.lo[k] - eps, -std::numeric_limits<float>::infinity());
      n.hi[k] = std::nextafterf(b.hi[k] + eps, std::numeric_limits<float>::infinity());
    }
  }

  // returns the split position (first index of the right part) or -1 for "make a leaf"
  int64_t split(int64_t first, int64_t count) {
    Box cb; cb.reset();
    for (int64_t i = first; i < first + count; i++) cb.grow(&centroid[3 * (size_t)order[i]]);
    int axis = 0;
    float ext = -1;
    for (int k = 0; k < 3; k++) if (cb.hi[k] - cb.lo[k] > ext) { ext = cb.hi[k] - cb.lo[k]; axis = k; }
    auto median = [&]() -> int64_t {
      split_axis = axis;
      const int64_t mid = first + count / 2;
      std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count,
                       [&](int32_t a, int32_t b) {
                         const float ca = centroid[3 * (size_t)a + axis], cb2 = centroid[3 * (size_t)b + axis];
                         return ca < cb2 || (ca == cb2 && a < b);
                       });
      return mid;
    };
    if (count <= kLeaf) return -1;
    if (median_only || !(ext > 0)) return median();

    float best_cost = std::numeric_limits<float>::infinity();
    int best_axis = -1, best_bin = -1;
    for (int k = 0; k < 3; k++) {
      const float e = cb.hi[k] - cb.lo[k];
      if (!(e > 0)) continue;
      Box bin_box[kBins];
      int64_t bin_n[kBins];
      for (int b = 0; b < kBins; b++) { bin_box[b].reset(); bin_n[b] = 0; }
      const float scale = kBins / e;
      for (int64_t i = first; i < first + count; i++) {
        const int32_t t = order[i];
        int b = (int)((centroid[3 * (size_t)t + k] - cb.lo[k]) * scale);
        b = std::min(std::max(b, 0), kBins - 1);
        bin_box[b].grow(tri_box[t]);
        bin_n[b]++;
      }
      float right_area[kBins];
      Box acc; acc.reset();
      for (int b = kBins - 1; b > 0; b--) { acc.grow(bin_box[b]); right_area[b] = acc.half_area(); }
      acc.reset();
      int64_t nl = 0;
      for (int b = 0; b < kBins - 1; b++) {
        acc.grow(bin_box[b]);
        nl += bin_n[b];
        const int64_t nr = count - nl;
        if (nl == 0 || nr == 0) continue;
        const float cost = acc.half_area() * (float)nl + right_area[b + 1] * (float)nr;
        if (cost < best_cost) { best_cost = cost; best_axis = k; best_bin = b; }
      }
    }
    if (best_axis < 0) return median();
    const float e = cb.hi[best_axis] - cb.lo[best_axis];
    const float scale = kBins / e;
    const float lo = cb.lo[best_axis];
    auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](int32_t t) {
      int b = (int)((centroid[3 * (size_t)t + best_axis] - lo) * scale);
      b = std::min(std::max(b, 0), kBins - 1);
      return b <= best_bin;
    });
    const int64_t mid = it - order.begin();
    if (mid == first || mid == first + count) return median();
    split_axis = best_axis;
    return mid;
  }

  void build(int64_t n) {
    tri_box.resize((size_t)n);
    centroid.resize(3 * (size_t)n);
    order.resize((size_t)n);
    for (int64_t i = 0; i < n; i++) {
      Box b; b.reset();
      for (int v = 0; v < 3; v++) b.grow(tri + 9 * i + 3 * v);
      tri_box[(size_t)i] = b;
      for (int k = 0; k < 3; k++) centroid[3 * (size_t)i + k] = 0.5f * (b.lo[k] + b.hi[k]);
      order[(size_t)i] = (int32_t)i;
    }
    nodes.clear();
    nodes.reserve((size_t)(2 * n + 2));
    nodes.push_back(BvhNode());
    node_axis.assign(1, 0);
    struct Work { int32_t node; int64_t first, count; int depth; };
    std::vector<Work> todo;
    todo.push_back({0, 0, n, 1});
    max_depth = 1;
    while (!todo.empty()) {
      const Work w = todo.back();
      todo.pop_back();
      max_depth = std::max(max_depth, w.depth);
      const Box b = range_box(w.first, w.count);
      set_bounds(nodes[(size_t)w.node], b);
      const int64_t mid = w.count > 1 ? split(w.first, w.count) : -1;
      if (mid < 0) {
        nodes[(size_t)w.node].a = (int32_t)w.first;
        nodes[(size_t)w.node].b = (int32_t)w.count;
        continue;
      }
      const int32_t left = (int32_t)nodes.size();
      nodes.push_back(BvhNode());
      nodes.push_back(BvhNode());
      node_axis.push_back(0);
      node_axis.push_back(0);
      node_axis[(size_t)w.node] = (uint8_t)split_axis;
      nodes[(size_t)w.node].a = left;
      nodes[(size_t)w.node].b = -w.depth;   // inner node: b <= 0; the depth rides along for the counting build
      todo.push_back({left, w.first, mid - w.first, w.depth + 1});
      todo.push_back({left + 1, mid, w.first + w.count - mid, w.depth + 1});
    }
  }

  // The kernel's form of the tree: one PairNode per inner node (afe_render.h), child references as byte
  // offsets into the array.  A mesh small enough to be a single leaf gets one record naming that leaf
  // twice (testing a triangle twice changes no minimum).
  //
  // `octant` (bit k = axis k mirrored): the same tree in coordinates mirrored about the origin along
  // those axes -- box {lo, hi} becomes {-hi, -lo} -- with the children of a node split along a mirrored
  // axis exchanged.  For rays whose direction is negative exactly along those axes, mirrored, every
  // component is positive: the near face of every slab is `lo`, and the child holding the lower
  // centroids (stored first) is the one the rays meet first.  (-hi)(-inv) is hi * inv bit for bit, so
  // the slab distances are the ones the unmirrored test computes.
  std::vector<PairNode> pairs(unsigned octant = 0) const {
    std::vector<int32_t> pair_of(nodes.size(), -1);
    int32_t n_pairs = 0;
    for (size_t k = 0; k < nodes.size(); k++) if (nodes[k].b <= 0) pair_of[k] = n_pairs++;
    std::vector<PairNode> out((size_t)std::max(n_pairs, 1));
    auto child = [&](const BvhNode &c, size_t index, float box[6], uint32_t &ref, uint32_t &count) {
      for (int k = 0; k < 3; k++) {
        const bool mirrored = (octant >> k) & 1u;
        box[2 * k] = mirrored ? -c.hi[k] : c.lo[k];
        box[2 * k + 1] = mirrored ? -c.lo[k] : c.hi[k];
      }
      if (c.b > 0) { ref = (uint32_t)c.a; count = (uint32_t)c.b; }
      else { ref = (uint32_t)pair_of[index] * (uint32_t)sizeof(PairNode); count = 0; }
    };
    if (n_pairs == 0) {
      PairNode &p = out[0];
      uint32_t cl = 0, cr = 0;
      child(nodes[0], 0, p.box_l, p.left, cl);
      child(nodes[0], 0, p.box_r, p.right, cr);
      p.meta = (cl << 8) | (cr << 16) | (1u << 24);
      p.pad = 0;
      return out;
    }
    for (size_t k = 0; k < nodes.size(); k++) {
      if (nodes[k].b > 0) continue;
      PairNode &p = out[(size_t)pair_of[k]];
      const unsigned axis = node_axis[k];
      const bool exchange = (octant >> axis) & 1u;
      const size_t l = (size_t)nodes[k].a + (exchange ? 1 : 0), r = (size_t)nodes[k].a + (exchange ? 0 : 1);
      uint32_t cl = 0, cr = 0;
      child(nodes[l], l, p.box_l, p.left, cl);
      child(nodes[r], r, p.box_r, p.right, cr);
      p.meta = axis | (cl << 8) | (cr << 16) | ((uint32_t)std::min(-nodes[k].b, 255) << 24);
      p.pad = 0;
    }
    return out;
  }
};

}  // namespace

// ---------------------------------------------------------------------------------------
// device
// ---------------------------------------------------------------------------------------
#ifndef AFE_TILE_W
#define AFE_TILE_W 8
#define AFE_TILE_H 8
#endif
constexpr int kTileW = AFE_TILE_W, kTileH = AFE_TILE_H, kStack = 32;
#ifndef AFE_ENTRY_GROUP
#define AFE_ENTRY_GROUP 2
#endif
constexpr int kEntryGroup = AFE_ENTRY_GROUP;
constexpr uint32_t kNoEntry = 0xffffffffu;     // afe_tile_entry_kernel: no node of the tree is within range of this tile group

struct PoseArgs {
  const void *pos, *att;      // planar, `stride` elements between components
  int64_t stride, first, count;
  int elem_size;              // 4 or 8
  double mount[4];
  double *poses;              // [count][12] = origin, row-major camera-to-world matrix
  const double *anchor_xy;    // engine state: x, y are relative to this set point (afe_device_view::pos_anchor_xy); NULL: absolute
};

__global__ void afe_camera_pose_kernel(PoseArgs a) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.count) return;
  const int64_t v = a.first + i;
  double p[3], q[4];
  if (a.elem_size == 8) {
    const double *P = (const double *)a.pos, *Q = (const double *)a.att;
    for (int k = 0; k < 3; k++) p[k] = P[k * a.stride + v];
    for (int k = 0; k < 4; k++) q[k] = Q[k * a.stride + v];
  } else {
    const float *P = (const float *)a.pos, *Q = (const float *)a.att;
    for (int k = 0; k < 3; k++) p[k] = (double)P[k * a.stride + v];
    for (int k = 0; k < 4; k++) q[k] = (double)Q[k * a.stride + v];
  }
  if (a.anchor_xy) { p[0] = a.anchor_xy[v] + p[0]; p[1] = a.anchor_xy[a.stride + v] + p[1]; }
  const double *m = a.mount;
  // att * mount, Rotation.hpp:124-131
  const double c0 = m[0] * q[0] - m[1] * q[1] - m[2] * q[2] - m[3] * q[3];
  const double c1 = m[1] * q[0] + m[0] * q[1] + m[3] * q[2] - m[2] * q[3];
  const double c2 = m[2] * q[0] - m[3] * q[1] + m[0] * q[2] + m[1] * q[3];
  const double c3 = m[3] * q[0] + m[2] * q[1] - m[1] * q[2] + m[0] * q[3];
  const double r0 = c0 * c0, r1 = c1 * c1, r2 = c2 * c2, r3 = c3 * c3;
  double *o = a.poses + 12 * i;
  o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
  // Rotation.hpp:196-220
  o[3] = r0 + r1 - r2 - r3;
  o[4] = 2 * c1 * c2 - 2 * c0 * c3;
  o[5] = 2 * c1 * c3 + 2 * c0 * c2;
  o[6] = 2 * c1 * c2 + 2 * c0 * c3;
  o[7] = r0 - r1 + r2 - r3;
  o[8] = 2 * c2 * c3 - 2 * c0 * c1;
  o[9] = 2 * c1 * c3 - 2 * c0 * c2;
  o[10] = 2 * c2 * c3 + 2 * c0 * c1;
  o[11] = r0 - r1 - r2 + r3;
}

// One triangle as the kernel wants it (96 B, fetched by scalar loads -- the leaf being visited is the
// same for the whole wave): vertex 0 and the two edges ALREADY in double, e = double(v_k) - double(v_0)
// exactly as the ray / triangle test forms them (so every lane is spared 9 conversions and 6
// subtractions per test), and the triangle's own box in fp32, inflated like the node boxes.
struct TriRec {
  double v0[3], e1[3], e2[3];
  float box[6];   // {lo, hi} per axis
#ifdef AFE_RENDER_FP32_CEILING
  float v0f[3], e1f[3], e2f[3];   // measurement build only, see ray_triangle
#endif
};

struct RenderArgs {
  const PairNode *pairs;    // eight copies of n_pairs records: [0] as built, [c] mirrored along the axes in c
  int64_t n_pairs;
  const TriRec *tris;       // leaf order
  const float *tribox;      // eight copies of n_tri x {lo, hi} x 3 (+ 2 pad): the triangles' boxes mirrored like the node boxes
  int64_t n_tri;
  // triangles whose box spans a good part of the scene (a ground plane's two) are kept OUT of the tree -- every tile
  // tests them first, which also gives every ray that looks down its pruning distance before the walk -- so that the
  // tree's top levels separate space: records [big_first, big_first + n_big) of tris
  uint32_t big_first, n_big;
  int groups_x, groups_per_view;   // tiles in groups of kEntryGroup x kEntryGroup for the entry table
  // per tile group of this launch (afe_tile_entry_kernel), ONE 64-bit word: low half = byte offset of the PairNode the walk
  // starts at (kNoEntry: nothing in range), bits 32-47 = which of the n_big out-of-tree triangles any ray of the group can
  // hit at all (all ones when the scene has none to cull).  NULL: from the root, every out-of-tree triangle.
  const uint64_t *entry;
  const double *poses;
  uint16_t *out;
  unsigned long long *counters;   // counting build only: see afe_render_depth_stats
  int64_t n_views, n_blocks, blocks_per_xcd;
  int width, height, tiles_x, tiles_per_view;
  double focal, cx, cy, depth_scale;
  int max_count;
  int plain_walk_only;      // afe_scene_set_walk(1)
  const double *uv;         // pixel-ray table: u(px) = (px - cx) / focal for px < tiles_x * 8, then v(py) likewise
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));

// Slab test against one box in fp32: can the ray be inside it anywhere in [0, best]?
// The builder inflates every box by 3e-4 m + 4e-6 |coordinate| -- two orders of magnitude more than
// what rounding the ray to fp32 can move it -- and the comparison below carries a relative slack of
// 1e-5, so the test only ever errs towards visiting a box: which triangles a ray reaches, and
// therefore the fp64 hit distance, cannot depend on it.
// oi = o * inv is precomputed per ray and a box arrives as three {lo, hi} pairs in scalar registers, so a
// slab is ONE packed FMA (v_pk_fma_f32: both faces at once).  |inv| is capped at 1e18 (kernel), so every
// distance here is finite.
__device__ __forceinline__ bool box_reached(f32x2 bx, f32x2 by, f32x2 bz, const float oi[3], const float inv[3],
                                            float best) {
  const f32x2 tx = __builtin_elementwise_fma(bx, (f32x2){inv[0], inv[0]}, (f32x2){-oi[0], -oi[0]});
  const f32x2 ty = __builtin_elementwise_fma(by, (f32x2){inv[1], inv[1]}, (f32x2){-oi[1], -oi[1]});
  const f32x2 tz = __builtin_elementwise_fma(bz, (f32x2){inv[2], inv[2]}, (f32x2){-oi[2], -oi[2]});
  float tmin = fmaxf(0.0f, fminf(tx.x, tx.y)), tmax = fminf(best, fmaxf(tx.x, tx.y));
  tmin = fmaxf(tmin, fminf(ty.x, ty.y)); tmax = fminf(tmax, fmaxf(ty.x, ty.y));
  tmin = fmaxf(tmin, fminf(tz.x, tz.y)); tmax = fminf(tmax, fmaxf(tz.x, tz.y));
  return tmin * 0.99999f <= tmax * 1.00001f;
}

__device__ __forceinline__ double ray_triangle(const double o[3], const double d[3], const TriRec &T) {
#pragma clang fp contract(off)
#ifdef AFE_RENDER_FP32_CEILING
  // MEASUREMENT BUILD ONLY (-DAFE_RENDER_FP32_CEILING): the whole test in fp32 with the hardware reciprocal and no
  // double-precision re-test anywhere.  Images are no longer the checker's; the point is the time, which bounds
  // from above what an fp32 filter with an fp64 re-test near decision boundaries could ever gain.
  {
    const float of[3] = {(float)o[0], (float)o[1], (float)o[2]}, df[3] = {(float)d[0], (float)d[1], (float)d[2]};
    const float *v0 = T.v0f, *e1 = T.e1f, *e2 = T.e2f;
    const float p[3] = {df[1] * e2[2] - df[2] * e2[1], df[2] * e2[0] - df[0] * e2[2], df[0] * e2[1] - df[1] * e2[0]};
    const float det = e1[0] * p[0] + e1[1] * p[1] + e1[2] * p[2];
    if (fabsf(det) < 1e-12f) return INFINITY;
    const float inv = __builtin_amdgcn_rcpf(det);
    const float tv[3] = {of[0] - v0[0], of[1] - v0[1], of[2] - v0[2]};
    const float u = (tv[0] * p[0] + tv[1] * p[1] + tv[2] * p[2]) * inv;
    if (u < 0.0f || u > 1.0f) return INFINITY;
    const float q[3] = {tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0]};
    const float v = (df[0] * q[0] + df[1] * q[1] + df[2] * q[2]) * inv;
    if (v < 0.0f || u + v > 1.0f) return INFINITY;
    const float t = (e2[0] * q[0] + e2[1] * q[1] + e2[2] * q[2]) * inv;
    return t > 0.0f ? (double)t : INFINITY;
  }
#endif
  // Moeller-Trumbore, two-sided; operation order is part of the contract (see file header).  v0, e1, e2
  // are the doubles the checker forms from the float vertices (afe_scene_create computes them once).
  const double *v0 = T.v0, *e1 = T.e1, *e2 = T.e2;
  const double p[3] = {d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0]};
  const double det = e1[0] * p[0] + e1[1] * p[1] + e1[2] * p[2];
  if (fabs(det) < 1e-12) return INFINITY;
  const double inv = 1.0 / det;
  const double tv[3] = {o[0] - v0[0], o[1] - v0[1], o[2] - v0[2]};
  const double u = (tv[0] * p[0] + tv[1] * p[1] + tv[2] * p[2]) * inv;
  if (u < 0.0 || u > 1.0) return INFINITY;
  const double q[3] = {tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0]};
  const double v = (d[0] * q[0] + d[1] * q[1] + d[2] * q[2]) * inv;
  if (v < 0.0 || u + v > 1.0) return INFINITY;
  const double t = (e2[0] * q[0] + e2[1] * q[1] + e2[2] * q[2]) * inv;
  return t > 0.0 ? t : INFINITY;
}

struct RayState {
  double o[3], d[3];
  float inv[3], oi[3];
  double best;
  float best_f;
};

struct WalkCounters { unsigned winner = 0xffffffffu, nodes = 0, tri_wave_box = 0, tri_wave_mt = 0, tri_lane_box = 0, tri_lane_mt = 0; };

__device__ __forceinline__ float as_float(uint32_t u) { return __builtin_bit_cast(float, u); }
// v_writelane_b32 (this compiler has the read side as a builtin, the write side only as the intrinsic)
extern "C" __device__ int afe_writelane(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

// What a lane keeps of its ray for the box tests.  ORDERED (all 64 rays of the tile agree on the sign of
// every direction component; the wave then walks the copy of the tree mirrored into the all-positive
// octant): per axis {|inv| (1 - 1e-5), |inv| (1 + 1e-5)} and the same two factors on -o * inv, so that one
// packed FMA on a {lo, hi} pair yields the near distance already slackened downwards and the far
// distance upwards, and the test is max3 / min3 and one comparison.  Otherwise {inv, inv}, {-oi, -oi}.
struct BoxRay { f32x2 scale[3], shift[3]; };

template <bool ORDERED>
__device__ __forceinline__ bool node_box_reached(f32x2 bx, f32x2 by, f32x2 bz, const BoxRay &r, float best) {
  const f32x2 tx = __builtin_elementwise_fma(bx, r.scale[0], r.shift[0]);
  const f32x2 ty = __builtin_elementwise_fma(by, r.scale[1], r.shift[1]);
  const f32x2 tz = __builtin_elementwise_fma(bz, r.scale[2], r.shift[2]);
  if (ORDERED) {
    // the four instructions themselves: written as fmaxf / fminf the compiler first quiets a possible signalling NaN in
    // `best` (one more vector instruction per visit and per triangle; nothing here can be a NaN, see the kernel)
    float tmin, tmax;
    asm("v_max_f32 %0, 0, %1" : "=v"(tmin) : "v"(tx.x));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmin) : "v"(tmin), "v"(ty.x), "v"(tz.x));
    asm("v_min_f32 %0, %1, %2" : "=v"(tmax) : "v"(best), "v"(tx.y));
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(tmax) : "v"(tmax), "v"(ty.y), "v"(tz.y));
    return tmin <= tmax;
  }
  float tmin = fmaxf(0.0f, fminf(tx.x, tx.y)), tmax = fminf(best, fmaxf(tx.x, tx.y));
  tmin = fmaxf(tmin, fminf(ty.x, ty.y)); tmax = fminf(tmax, fmaxf(ty.x, ty.y));
  tmin = fmaxf(tmin, fminf(tz.x, tz.y)); tmax = fminf(tmax, fmaxf(tz.x, tz.y));
  return tmin * 0.99999f <= tmax * 1.00001f;
}

// the triangles of one leaf, for the lanes in mask m (wave-uniform): the triangle's own (inflated) box
// first, in fp32 -- a ray that misses it, or enters it no nearer than its best hit so far, cannot gain
// anything from this triangle, and when that holds for all 64 rays of the tile the double-precision
// test is skipped altogether
template <bool COUNT, bool ORDERED>
__device__ __forceinline__ void leaf_triangles(const RenderArgs &a, RayState &ray, const BoxRay &br, const float *__restrict__ tribox,
                                               uint32_t first, unsigned count, uint64_t m, WalkCounters &cnt) {
#pragma clang fp contract(off)
  for (unsigned k = 0; k < count; k++) {
    const TriRec &T = a.tris[first + k];
    bool in_box;
    if (ORDERED) {
      // the mirrored copy of the triangle's box (32 B, one scalar load): near faces are `lo`, the 9-instruction test of the nodes
      const float *B = tribox + (size_t)(first + k) * 8;
      in_box = node_box_reached<true>((f32x2){B[0], B[1]}, (f32x2){B[2], B[3]}, (f32x2){B[4], B[5]}, br, ray.best_f);
    } else {
      in_box = box_reached((f32x2){T.box[0], T.box[1]}, (f32x2){T.box[2], T.box[3]}, (f32x2){T.box[4], T.box[5]}, ray.oi, ray.inv, ray.best_f);
    }
    const uint64_t reach = __ballot(in_box) & m;
    if (COUNT) { cnt.tri_lane_box += __builtin_amdgcn_inverse_ballot_w64(m) ? 1 : 0; cnt.tri_wave_box += 1; }
    if (reach) {
      const bool me = __builtin_amdgcn_inverse_ballot_w64(reach);
      if (COUNT) { cnt.tri_wave_mt += 1; cnt.tri_lane_mt += me ? 1 : 0; }
      if (me) {
        const double th = ray_triangle(ray.o, ray.d, T);
        if (th < ray.best) { ray.best = th; ray.best_f = __double2float_ru(th); if (COUNT) cnt.winner = first + k; }
      }
    }
  }
}

// The 64 rays of a tile walk the tree TOGETHER: one node index and one stack for the wave (so node
// records and triangles arrive by scalar loads, once per wave instead of once per lane).  A visit
// is one 64-byte scalar load (a PairNode: the boxes of both children); every lane tests its own ray
// against both, and a child is entered if the ray of any lane still in the node reaches it.  The child
// on the side the rays come from goes first -- in the mirrored copies (ORDERED) that is simply the one
// stored first, otherwise split axis x direction sign, a scalar decision; a leaf child's triangles are
// tested there and then, an inner child is descended into, and when both are inner the farther one
// is pushed together with the mask of the lanes that reached it.  Lane sets are 64-bit masks in
// scalar registers throughout (ballots), the stack lives in three vector registers (entry k in lane
// k: v_writelane / v_readlane, no LDS, nothing to wait for).  The mask of a popped entry may be stale
// (the ray's best hit may have come nearer since) -- that only costs box tests.  The closest hit is a
// minimum over the triangles each ray reaches, each value computed in double exactly as the checker
// does, so the result does not depend on the order.
template <bool COUNT, bool ORDERED>
__device__ __forceinline__ void walk(const RenderArgs &a, const PairNode *tree, const float *__restrict__ tribox, RayState &ray, const BoxRay &br,
                                     bool in_image, unsigned neg, WalkCounters &cnt, uint32_t start) {
  int sp = 0;
  int st_node = 0, st_lo = 0, st_hi = 0;
  uint32_t cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)start);   // byte offset of the PairNode (wave-uniform: one tile, one entry)
  uint64_t act = __ballot(in_image);
  for (;;) {
    const u32x16 R = *reinterpret_cast<const u32x16 *>(reinterpret_cast<const char *>(tree) + cur);
    const uint32_t meta = R[14];
    if (COUNT) cnt.nodes += 1;
    const bool bl = node_box_reached<ORDERED>((f32x2){as_float(R[0]), as_float(R[1])}, (f32x2){as_float(R[2]), as_float(R[3])},
                                              (f32x2){as_float(R[4]), as_float(R[5])}, br, ray.best_f);
    const bool bR = node_box_reached<ORDERED>((f32x2){as_float(R[6]), as_float(R[7])}, (f32x2){as_float(R[8]), as_float(R[9])},
                                              (f32x2){as_float(R[10]), as_float(R[11])}, br, ray.best_f);
    const uint64_t ml = __ballot(bl) & act, mr = __ballot(bR) & act;
    // nearer side first: the left child holds the lower centroids along the split axis
    const bool right_first = !ORDERED && ((neg >> (meta & 3u)) & 1u);
    const uint64_t m1 = right_first ? mr : ml, m2 = right_first ? ml : mr;
    const uint32_t ref1 = right_first ? R[13] : R[12], ref2 = right_first ? R[12] : R[13];
    const unsigned c1 = (right_first ? (meta >> 16) : (meta >> 8)) & 255u;
    const unsigned c2 = (right_first ? (meta >> 8) : (meta >> 16)) & 255u;
    // the far child goes into slot sp at EVERY visit and sp advances only when it is really pushed: a conditional
    // write makes the compiler carry the three stack registers through copies on both paths (six v_mov per visit,
    // an eighth of the kernel's vector instructions); what an unadvanced slot holds is never read
    // (nested tests on the masks and counts themselves: every one is a scalar compare and branch; the same decisions
    // written as booleans that are combined and reused cost twice the scalar instructions)
    uint32_t nxt = 0xffffffffu;
    uint64_t nact = 0;
    if (m1 != 0) {
      if (c1 != 0) leaf_triangles<COUNT, ORDERED>(a, ray, br, tribox, ref1, c1, m1, cnt);
      else { nxt = ref1; nact = m1; }
    }
    if (m2 != 0) {
      if (c2 != 0) leaf_triangles<COUNT, ORDERED>(a, ray, br, tribox, ref2, c2, m2, cnt);
      else if (nxt == 0xffffffffu) { nxt = ref2; nact = m2; }
      else {                         // both inner: the nearer one next, the farther one onto the stack
        st_node = afe_writelane((int)ref2, sp, st_node);
        st_lo = afe_writelane((int)(uint32_t)m2, sp, st_lo);
        st_hi = afe_writelane((int)(uint32_t)(m2 >> 32), sp, st_hi);
        sp++;
      }
    }
    if (nxt != 0xffffffffu) { cur = nxt; act = nact; continue; }
    if (sp == 0) break;
    --sp;
    cur = (uint32_t)__builtin_amdgcn_readlane(st_node, sp);
    act = (uint64_t)(uint32_t)__builtin_amdgcn_readlane(st_lo, sp) |
          ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(st_hi, sp) << 32);
  }
}

__global__ void __launch_bounds__(256) afe_pixel_ray_table_kernel(double *uv, int nx, int ny, double cx, double cy, double focal) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < nx) uv[i] = (i - cx) / focal;
  else if (i < nx + ny) uv[i] = ((i - nx) - cy) / focal;
}

// Where a tile's walk may start.  One lane per tile: the four corner rays of the tile bound its 64 (the direction is
// linear in the pixel, so every component of every ray lies between the corners' and, where the corners agree on its
// sign, so does its reciprocal); a child box that the bundle cannot reach within the camera's range -- per axis the
// earliest any ray can enter and the latest any can leave, the slab test on intervals -- holds nothing any ray of the
// tile can hit.  From the root, as long as exactly ONE child is reachable and it is an inner node, go there: the walk
// of the tile starts at the node where that stops.  In double, with a margin; an axis on which the corners do not
// agree (or a component is nearly zero) constrains nothing.  Conservative, so the images cannot change; what it saves
// is the visits of the top levels, whose other child is a part of the scene the tile's 10 m cannot reach.
__global__ void __launch_bounds__(64) afe_tile_entry_kernel(RenderArgs a, uint64_t *entry, int cull_big) {
#pragma clang fp contract(off)
  // (one lane per GROUP of kEntryGroup x kEntryGroup tiles: a quarter of the lanes, entries half a level higher)
  const int64_t logical = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (logical >= a.n_views * a.groups_per_view) return;
  const int64_t view = logical / a.groups_per_view;
  const int group = (int)(logical - view * a.groups_per_view);
  const int px0 = (group % a.groups_x) * kTileW * kEntryGroup, py0 = (group / a.groups_x) * kTileH * kEntryGroup;
  const int px1 = min(px0 + kTileW * kEntryGroup - 1, a.width - 1), py1 = min(py0 + kTileH * kEntryGroup - 1, a.height - 1);
  const double *pose = a.poses + 12 * view;
  const double us[2] = {a.uv[px0], a.uv[px1]}, vs[2] = {a.uv[a.tiles_x * kTileW + py0], a.uv[a.tiles_x * kTileW + py1]};
  // (single precision with margins a hundred times its rounding: the boxes are inflated by 3e-4 m and more, the test
  // only has to be conservative, and this kernel's time is not the images')
  float inv[3][4];
  int sign[3];          // +1 / -1: every corner agrees; 0: no constraint from this axis
  for (int k = 0; k < 3; k++) {
    int pos = 0, neg = 0;
    for (int c = 0; c < 4; c++) {
      const double d = pose[3 + 3 * k] * us[c & 1] + pose[4 + 3 * k] * vs[c >> 1] + pose[5 + 3 * k];
      pos += d > 1e-6; neg += d < -1e-6;
      inv[k][c] = __builtin_amdgcn_rcpf((float)d);
    }
    sign[k] = pos == 4 ? 1 : (neg == 4 ? -1 : 0);
  }
  const float o[3] = {(float)pose[0], (float)pose[1], (float)pose[2]};
  const float t_limit = (float)((double)a.max_count * a.depth_scale * 1.001);
  auto reachable = [&](const float *box) {
    float tmin = 0.0f, tmax = t_limit;
    for (int k = 0; k < 3; k++) {
      if (sign[k] == 0) continue;
      const float near = (sign[k] > 0 ? box[2 * k] : box[2 * k + 1]) - o[k];
      const float far = (sign[k] > 0 ? box[2 * k + 1] : box[2 * k]) - o[k];
      float tn = near * inv[k][0], tf = far * inv[k][0];
      for (int c = 1; c < 4; c++) { tn = fminf(tn, near * inv[k][c]); tf = fmaxf(tf, far * inv[k][c]); }
      tmin = fmaxf(tmin, tn - 1e-4f * (1.0f + fabsf(tn)));
      tmax = fminf(tmax, tf + 1e-4f * (1.0f + fabsf(tf)));
    }
    return tmin <= tmax;
  };
  uint32_t cur = 0;
  for (int level = 0; level < kStack; level++) {
    const PairNode &P = *reinterpret_cast<const PairNode *>(reinterpret_cast<const char *>(a.pairs) + cur);
    const bool hl = reachable(P.box_l), hr = reachable(P.box_r);
    const unsigned cl = (P.meta >> 8) & 255u, cr = (P.meta >> 16) & 255u;
    if (hl && !hr && cl == 0) cur = P.left;
    else if (hr && !hl && cr == 0) cur = P.right;
    else {
      // neither child within the camera's range for any ray of the group (sky, or an orchard that ends before the far
      // plane): there is nothing to walk -- the tiles go straight to their answer (round 6)
      if (!hl && !hr) cur = kNoEntry;
      break;
    }
  }
  unsigned mask = 0xffffu;
  // The triangles kept out of the tree (a ground plane's two): every tile used to test all of them -- two double-precision
  // tests where at most one can hit.  A ray's barycentric numerators N_u = tv.(d x e2), N_v = d.(tv x e1) and the
  // determinant det = e1.(d x e2) are LINEAR in the direction, the direction is linear in the pixel, so over the group's
  // pixel rectangle each of them -- and N_u + N_v - det -- takes its extremes at the four corner rays.  Where the corners
  // agree on the determinant's sign and put one of the hit conditions (u >= 0, v >= 0, u + v <= 1, t > 0, t within range)
  // out of reach for all four, with a margin a million times the rounding of the per-ray test, NO ray of the group hits
  // the triangle: its bit stays clear and the tiles skip its box and its double-precision test.  Conservative, in double.
  if (cull_big) {
    mask = 0;
    double dc[4][3];
    for (int c = 0; c < 4; c++)
      for (int k = 0; k < 3; k++) dc[c][k] = pose[3 + 3 * k] * us[c & 1] + pose[4 + 3 * k] * vs[c >> 1] + pose[5 + 3 * k];
    const double t_far = (double)a.max_count * a.depth_scale * 1.001;
    for (unsigned b = 0; b < a.n_big && b < 16u; b++) {
      const TriRec &T = a.tris[a.big_first + b];
      const double tv[3] = {pose[0] - T.v0[0], pose[1] - T.v0[1], pose[2] - T.v0[2]};
      const double q[3] = {tv[1] * T.e1[2] - tv[2] * T.e1[1], tv[2] * T.e1[0] - tv[0] * T.e1[2], tv[0] * T.e1[1] - tv[1] * T.e1[0]};
      const double nt = T.e2[0] * q[0] + T.e2[1] * q[1] + T.e2[2] * q[2];
      double det[4], nu[4], nv[4], mag = 0;
      for (int c = 0; c < 4; c++) {
        const double *d = dc[c];
        const double pv[3] = {d[1] * T.e2[2] - d[2] * T.e2[1], d[2] * T.e2[0] - d[0] * T.e2[2], d[0] * T.e2[1] - d[1] * T.e2[0]};
        det[c] = T.e1[0] * pv[0] + T.e1[1] * pv[1] + T.e1[2] * pv[2];
        nu[c] = tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2];
        nv[c] = d[0] * q[0] + d[1] * q[1] + d[2] * q[2];
        mag = fmax(mag, fabs(T.e1[0] * pv[0]) + fabs(T.e1[1] * pv[1]) + fabs(T.e1[2] * pv[2]) + fabs(tv[0] * pv[0]) + fabs(tv[1] * pv[1]) +
                            fabs(tv[2] * pv[2]) + fabs(d[0] * q[0]) + fabs(d[1] * q[1]) + fabs(d[2] * q[2]));
      }
      const double eps = 1e-9 * mag + 1e-300;
      bool keep = true;
      const bool pos = det[0] > eps && det[1] > eps && det[2] > eps && det[3] > eps;
      const bool neg = det[0] < -eps && det[1] < -eps && det[2] < -eps && det[3] < -eps;
      if (pos || neg) {
        const double sg = pos ? 1.0 : -1.0;
        double u_max = -1e300, v_max = -1e300, w_min = 1e300, det_max = 0;
        for (int c = 0; c < 4; c++) {
          u_max = fmax(u_max, sg * nu[c]);
          v_max = fmax(v_max, sg * nv[c]);
          w_min = fmin(w_min, sg * (nu[c] + nv[c] - det[c]));
          det_max = fmax(det_max, sg * det[c]);
        }
        // t = t_num / (sg det): behind the camera when negative, nearest where the determinant is largest
        const double t_num = sg * nt, eps_t = 1e-9 * (fabs(T.e2[0] * q[0]) + fabs(T.e2[1] * q[1]) + fabs(T.e2[2] * q[2])) + 1e-300;
        if (u_max < -eps || v_max < -eps || w_min > eps || t_num < -eps_t || (t_num > eps_t && t_num > t_far * det_max * (1.0 + 1e-9)))
          keep = false;
      }
      if (keep) mask |= 1u << b;
    }
  }
  // (one store: what the walk starts at and what it tests first travel together -- round 6 first kept the masks in a second
  // array behind the entries in the same stream-ordered allocation, and under a second process's load one render in two
  // thousand read masks that were not this launch's: tools/experiments/flight_repro.py)
  entry[logical] = (uint64_t)cur | ((uint64_t)mask << 32);
}

template <bool COUNT>
__global__ __launch_bounds__(kTileW *kTileH) void afe_render_depth_kernel(RenderArgs a) {
#pragma clang fp contract(off)
  // XCD-aware order: hardware block b runs on XCD b % 8; give each XCD a contiguous run
  // of logical blocks (= consecutive tiles of consecutive views) so that its L2 keeps the
  // part of the tree those views look at
  const int64_t hw = blockIdx.x;
  const int64_t logical = (hw % 8) * a.blocks_per_xcd + hw / 8;
  if (logical >= a.n_blocks) return;
  const int64_t view = logical / a.tiles_per_view;
  const int tile = (int)(logical - view * a.tiles_per_view);
  const int lane = threadIdx.x;
  const int px = (tile % a.tiles_x) * kTileW + (lane % kTileW);
  const int py = (tile / a.tiles_x) * kTileH + (lane / kTileW);
  const bool in_image = px < a.width && py < a.height;   // lanes outside still take part in the wave's votes

  const double *pose = a.poses + 12 * view;
  RayState ray;
  for (int k = 0; k < 3; k++) ray.o[k] = pose[k];
  // (px - cx) / focal and (py - cy) / focal: the contract's own double-precision divisions, done once per camera
  // by afe_pixel_ray_table_kernel instead of twice per ray (55 of the ~1 000 vector instructions of a wave)
  const double u = a.uv[px];
  const double v = a.uv[a.tiles_x * kTileW + py];
  for (int k = 0; k < 3; k++) ray.d[k] = pose[3 + 3 * k] * u + pose[4 + 3 * k] * v + pose[5 + 3 * k];
  for (int k = 0; k < 3; k++) {
    // |1/d| is capped: a direction component of exactly zero (a pixel on the principal axis of an
    // axis-aligned camera) would make it infinite and the slab distances inf - inf = NaN -- and NaNs are NOT
    // harmless in the min / max chains below (fmaxf(-inf, NaN) = -inf shuts a box the ray is inside of).
    // 1e18 instead stands for a component of 1e-18: over the 10 m range that moves the ray by 1e-17 m, and
    // every slab distance stays finite for coordinates up to 1e20 m.
    // (the hardware reciprocal: one instruction instead of the ten of an exact division; its last-place error is two
    // orders of magnitude inside the 1e-5 slack of the box tests, the only consumers)
    ray.inv[k] = fminf(fmaxf(__builtin_amdgcn_rcpf((float)ray.d[k]), -1e18f), 1e18f);
    ray.oi[k] = (float)ray.o[k] * ray.inv[k];   // o * inv, see box_reached
  }
  ray.best = INFINITY;
  // pruning bound of the fp32 box tests: the best hit so far, rounded up.  A hit at or beyond
  // max_count * depth_scale saturates to max_count exactly like a miss, so nothing farther than
  // that needs to be found at all.
  ray.best_f = __double2float_ru((double)a.max_count * a.depth_scale * 1.000001);

  // bit k of `neg`: direction component k is negative; `ordered`: all 64 rays of the tile agree on all
  // three signs (every tile that does not straddle a coordinate plane through the camera) -- the wave
  // then walks the copy of the tree mirrored into the octant where all of them are positive
  unsigned neg = 0;
  bool ordered = true;
  for (int k = 0; k < 3; k++) {
    const uint64_t n = __ballot(ray.inv[k] < 0.0f);
    neg |= n ? (1u << k) : 0u;
    ordered = ordered && (n == 0 || n == ~0ull);
  }
  ordered = ordered && !a.plain_walk_only;
  WalkCounters cnt;
  BoxRay br;
  // A camera at a non-finite position (a vehicle that has diverged) shows nothing, by the contract's own arithmetic: tv = o - v0
  // is not finite, so neither is any t, and no t that is not finite becomes a hit.  Without this its NaN passes every box test
  // (max / min drop a NaN operand) and each of its tiles walks the WHOLE tree: one such view cost 1 150 ordinary ones
  // (tools/experiments/nan_pose_probe.py).  A non-finite attitude needs no care: the entry pass already finds nothing to walk.
  const bool somewhere = __builtin_isfinite(ray.o[0]) && __builtin_isfinite(ray.o[1]) && __builtin_isfinite(ray.o[2]);
  if (!somewhere) {
  } else if (ordered) {
    for (int k = 0; k < 3; k++) {
      const float ai = fabsf(ray.inv[k]);      // mirrored axis: -inv, and (-o)(-inv) = o * inv
      br.scale[k] = (f32x2){ai * 0.99999f, ai * 1.00001f};
      br.shift[k] = (f32x2){-ray.oi[k] * 0.99999f, -ray.oi[k] * 1.00001f};
    }
    const float *tb = a.tribox + (int64_t)neg * a.n_tri * 8;
    const int64_t group = view * a.groups_per_view + (int64_t)((tile / a.tiles_x) / kEntryGroup) * a.groups_x + (tile % a.tiles_x) / kEntryGroup;
    const uint64_t word = a.entry ? a.entry[group] : (0xffffull << 32);
    const uint32_t start = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)word);
    // only the out-of-tree triangles some ray of this tile group can hit (afe_tile_entry_kernel)
    for (unsigned bm = (unsigned)__builtin_amdgcn_readfirstlane((int)(uint32_t)(word >> 32)) & ((1u << a.n_big) - 1u); bm; bm &= bm - 1u)
      leaf_triangles<COUNT, true>(a, ray, br, tb, a.big_first + (unsigned)(__builtin_ffs((int)bm) - 1), 1u, __ballot(in_image), cnt);
    if (start != kNoEntry) walk<COUNT, true>(a, a.pairs + (int64_t)neg * a.n_pairs, tb, ray, br, in_image, neg, cnt, start);
  } else {
    for (int k = 0; k < 3; k++) {
      br.scale[k] = (f32x2){ray.inv[k], ray.inv[k]};
      br.shift[k] = (f32x2){-ray.oi[k], -ray.oi[k]};
    }
    if (a.n_big) leaf_triangles<COUNT, false>(a, ray, br, a.tribox, a.big_first, a.n_big, __ballot(in_image), cnt);
    walk<COUNT, false>(a, a.pairs, a.tribox, ray, br, in_image, neg, cnt, 0u);
  }
  const double best = ray.best;
  const unsigned c_winner = cnt.winner, c_nodes = cnt.nodes, c_tri_wave_box = cnt.tri_wave_box, c_tri_wave_mt = cnt.tri_wave_mt,
                 c_tri_lane_box = cnt.tri_lane_box, c_tri_lane_mt = cnt.tri_lane_mt;

  uint16_t count = (uint16_t)a.max_count;
  if (best < INFINITY) {
    const double c = floor(best / a.depth_scale);
    if (c < (double)a.max_count) count = (uint16_t)c;
  }
  if (in_image) a.out[(view * a.height + py) * (int64_t)a.width + px] = count;
  if (COUNT) {
    // per wave: nodes visited, triangle box tests, triangle double-precision tests; per lane: the same two
    // for the rays that took part, and the rays themselves
    unsigned lane_box = c_tri_lane_box, lane_mt = c_tri_lane_mt, rays = in_image ? 1u : 0u;
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) {
      lane_box += __shfl_xor(lane_box, sft); lane_mt += __shfl_xor(lane_mt, sft); rays += __shfl_xor(rays, sft);
    }
    // the triangles this tile really shows: distinct winners among its rays (what no traversal could avoid testing)
    unsigned visible = 0;
    for (uint64_t todo = __ballot(in_image && best < INFINITY && c_winner != 0xffffffffu); todo;) {
      const unsigned w0 = (unsigned)__builtin_amdgcn_readlane((int)c_winner, (int)__ffsll((unsigned long long)todo) - 1);
      todo &= ~__ballot(c_winner == w0);
      visible++;
    }
    if (lane == 0) {
      atomicAdd(&a.counters[0], (unsigned long long)c_nodes);
      atomicAdd(&a.counters[1], (unsigned long long)c_tri_wave_box);
      atomicAdd(&a.counters[2], (unsigned long long)c_tri_wave_mt);
      atomicAdd(&a.counters[3], (unsigned long long)lane_box);
      atomicAdd(&a.counters[4], (unsigned long long)lane_mt);
      atomicAdd(&a.counters[5], (unsigned long long)rays);
      atomicAdd(&a.counters[6], 1ull);
      atomicAdd(&a.counters[7], (unsigned long long)visible);
    }
  }
}

}  // namespace afe

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
using namespace afe;

struct afe_scene {
  int device = 0;
  int64_t n_tri = 0, n_nodes = 0;
  int depth = 0;
  double bounds[6] = {0, 0, 0, 0, 0, 0};
  int plain_walk_only = 0;
  PairNode *pairs = nullptr;   // 8 x n_pairs (Builder::pairs)
  int64_t n_pairs = 0;
  TriRec *tris = nullptr;
  float *tribox = nullptr;     // 8 x n_tri x 8 floats
  uint32_t big_first = 0, n_big = 0;   // the triangles kept out of the tree (RenderArgs)
  // pixel-ray tables, one per camera geometry this scene has been rendered with (kept until the scene goes:
  // a launch still in flight on some stream may be reading one)
  struct RayTable { int width, height; double cx, cy, focal; double *uv; };
  std::vector<RayTable> ray_tables;
  // Tile-entry tables (one 64-bit word per tile group of a launch), kept and reused: a launch takes a table no launch in
  // flight is using -- `done` is recorded behind the render kernel that reads it -- or makes one.  (Round 6: the table used
  // to come from hipMallocAsync / hipFreeAsync around every launch.  With the entry pass doing more work per group, renders
  // from the headless loop -- system HIP runtime, another process loading the GPU -- then read tables that were not their
  // launch's, one image in two thousand: tools/experiments/flight_repro.py.  Not with a synchronous allocation, not with
  // tables that live as long as the scene.)
  // (`taken`: a launch is being queued on it -- nobody else may have it, whatever its event says; `used`: `done` has been
  //  recorded behind a render kernel that reads it, free again once that event has completed)
  struct EntryTable { void *p = nullptr; size_t bytes = 0; hipEvent_t done = nullptr; bool used = false, taken = false; };
  std::list<EntryTable> entry_tables;       // (a list: launches hold pointers to their table outside the lock)
  std::mutex entry_mutex;
};

namespace {
struct DevBuf {
  void *p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  bool alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1) == hipSuccess; }
  bool upload(const void *src, size_t bytes) {
    return alloc(bytes) && hipMemcpy(p, src, bytes, hipMemcpyHostToDevice) == hipSuccess;
  }
};

int pick_device(int device, int *out) {
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return AFE_ERR_NO_DEVICE;
  if (device < 0 && hipGetDevice(&device) != hipSuccess) return AFE_ERR_NO_DEVICE;
  if (device >= n_dev) return AFE_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess || std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return AFE_ERR_NO_DEVICE;
  if (hipSetDevice(device) != hipSuccess) return AFE_ERR_HIP;
  *out = device;
  return AFE_OK;
}

bool camera_ok(const afe_camera *c) {
  return c && c->width > 0 && c->height > 0 && c->focal_length > 0 && c->depth_scale > 0 && c->max_count > 0 &&
         c->max_count <= 65535;
}

// poses (device, [count][12]) -> images; out_dev: device buffer of count*h*w uint16
int launch_render(afe_scene *s, const afe_camera *cam, int64_t count, const double *poses, uint16_t *out_dev,
                  hipStream_t stream, float *kernel_ms, unsigned long long *dev_counters = nullptr) {
  RenderArgs r;
  {
    const int nx = ((cam->width + kTileW - 1) / kTileW) * kTileW, ny = ((cam->height + kTileH - 1) / kTileH) * kTileH;
    const double *uv = nullptr;
    for (const afe_scene::RayTable &t : s->ray_tables)
      if (t.width == cam->width && t.height == cam->height && t.cx == cam->cx && t.cy == cam->cy && t.focal == cam->focal_length) uv = t.uv;
    if (!uv) {
      double *fresh = nullptr;
      if (hipMalloc((void **)&fresh, (size_t)(nx + ny) * sizeof(double)) != hipSuccess) return AFE_ERR_HIP;
      hipLaunchKernelGGL(afe_pixel_ray_table_kernel, dim3((unsigned)((nx + ny + 255) / 256)), dim3(256), 0, stream, fresh, nx, ny,
                         cam->cx, cam->cy, cam->focal_length);
      // other streams may use the table from now on: make it complete before it is published
      if (hipStreamSynchronize(stream) != hipSuccess) { (void)hipFree(fresh); return AFE_ERR_HIP; }
      s->ray_tables.push_back({cam->width, cam->height, cam->cx, cam->cy, cam->focal_length, fresh});
      uv = fresh;
    }
    r.uv = uv;
  }
  r.pairs = s->pairs; r.n_pairs = s->n_pairs; r.tris = s->tris; r.tribox = s->tribox; r.n_tri = s->n_tri; r.counters = dev_counters;
  r.big_first = s->big_first; r.n_big = s->n_big; r.entry = nullptr;
  r.width = cam->width; r.height = cam->height;
  r.tiles_x = (cam->width + kTileW - 1) / kTileW;
  r.tiles_per_view = r.tiles_x * ((cam->height + kTileH - 1) / kTileH);
  r.groups_x = (r.tiles_x + kEntryGroup - 1) / kEntryGroup;
  r.groups_per_view = r.groups_x * (((cam->height + kTileH - 1) / kTileH + kEntryGroup - 1) / kEntryGroup);
  r.focal = cam->focal_length; r.cx = cam->cx; r.cy = cam->cy; r.depth_scale = cam->depth_scale;
  r.max_count = cam->max_count;
  r.plain_walk_only = s->plain_walk_only;
  // A launch may not exceed 2^32 threads in all (HIP truncates the product silently): 65 536 views of
  // 320 x 240 are 78.6 M tiles x 64 lanes = 5.0e9.  Views go out in runs that stay below 2^31 threads.
  const int64_t max_blocks = (int64_t(1) << 31) / (kTileW * kTileH);
  if (r.tiles_per_view > max_blocks) return AFE_ERR_OUT_OF_RANGE;
  const int64_t views_per_launch = max_blocks / r.tiles_per_view;
  const size_t px = (size_t)cam->width * cam->height;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (kernel_ms && (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)) return AFE_ERR_HIP;
  if (kernel_ms) (void)hipEventRecord(e0, stream);
  int rc = AFE_OK;
  for (int64_t v0 = 0; v0 < count && rc == AFE_OK; v0 += views_per_launch) {
    const int64_t nv = (count - v0) < views_per_launch ? (count - v0) : views_per_launch;
    r.poses = poses + 12 * v0;
    r.out = out_dev + (size_t)v0 * px;
    r.n_views = nv;
    r.n_blocks = nv * r.tiles_per_view;
    r.blocks_per_xcd = (r.n_blocks + 7) / 8;
    const int64_t grid = r.blocks_per_xcd * 8;
    uint64_t *entry = nullptr;
    afe_scene::EntryTable *table = nullptr;
    if (!s->plain_walk_only) {     // (the plain walk stays the independent formulation: from the root)
      // a table of the scene's that no launch in flight is reading (several streams may render one scene)
      const int64_t n_groups = nv * r.groups_per_view;
      const int cull_big = (s->n_big > 0 && s->n_big <= 16 && !afe_dev_env("AFE_RENDER_NO_BIG_MASK")) ? 1 : 0;      // (lab variable: every tile tests every out-of-tree triangle)
      {
        const size_t need = (size_t)n_groups * sizeof(uint64_t);
        std::lock_guard<std::mutex> lock(s->entry_mutex);
        for (afe_scene::EntryTable &t : s->entry_tables)
          if (t.bytes >= need && !t.taken && (!t.used || hipEventQuery(t.done) == hipSuccess)) { table = &t; break; }
        if (!table) {
          afe_scene::EntryTable t;
          if (hipMalloc(&t.p, need + need / 4) != hipSuccess || hipEventCreateWithFlags(&t.done, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            if (t.p) (void)hipFree(t.p);
            rc = AFE_ERR_HIP;
            break;
          }
          t.bytes = need + need / 4;
          s->entry_tables.push_back(t);
          table = &s->entry_tables.back();
        }
        table->taken = true;                             // ours until the event behind our render kernel is recorded (below)
        entry = (uint64_t *)table->p;
      }
      r.entry = nullptr;
      hipLaunchKernelGGL(afe_tile_entry_kernel, dim3((unsigned)((n_groups + 63) / 64)), dim3(64), 0, stream, r, entry, cull_big);
      r.entry = entry;
    }
    if (dev_counters) hipLaunchKernelGGL(afe_render_depth_kernel<true>, dim3((unsigned)grid), dim3(kTileW * kTileH), 0, stream, r);
    else hipLaunchKernelGGL(afe_render_depth_kernel<false>, dim3((unsigned)grid), dim3(kTileW * kTileH), 0, stream, r);
    rc = hipGetLastError() == hipSuccess ? AFE_OK : AFE_ERR_HIP;
    if (table) {      // free for the next launch once this render kernel has read it
      std::lock_guard<std::mutex> lock(s->entry_mutex);
      (void)hipEventRecord(table->done, stream);
      table->used = true;
      table->taken = false;
    }
  }
  if (kernel_ms) {
    (void)hipEventRecord(e1, stream);
    if (rc == AFE_OK && hipEventSynchronize(e1) != hipSuccess) rc = AFE_ERR_HIP;
    if (rc == AFE_OK) (void)hipEventElapsedTime(kernel_ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  }
  return rc;
}

int launch_poses(const void *pos, const double *anchor_xy, const void *att, int64_t stride, int64_t first, int64_t count, int elem_size,
                 const double mount[4], double *poses, hipStream_t stream) {
  PoseArgs p;
  p.anchor_xy = anchor_xy;
  p.pos = pos; p.att = att; p.stride = stride; p.first = first; p.count = count; p.elem_size = elem_size;
  static const double identity[4] = {1, 0, 0, 0};
  const double *m = mount ? mount : identity;
  for (int k = 0; k < 4; k++) p.mount[k] = m[k];
  p.poses = poses;
  hipLaunchKernelGGL(afe_camera_pose_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, p);
  return hipGetLastError() == hipSuccess ? AFE_OK : AFE_ERR_HIP;
}
}  // namespace

extern "C" int afe_camera_default(afe_camera *cam, int width, int height) {
  if (!cam || width <= 0 || height <= 0) return AFE_ERR_INVALID_ARG;
  std::memset(cam, 0, sizeof(*cam));
  cam->width = width; cam->height = height;
  cam->focal_length = width / 2.0;               // main.cpp:360
  cam->cx = width / 2.0; cam->cy = height / 2.0; // main.cpp:485-486
  cam->depth_scale = 10.0 / 256.0;               // main.cpp:121-122
  cam->max_count = 255;                          // 8-bit DepthVis
  return AFE_OK;
}

extern "C" int afe_camera_default_mount(double mount[4]) {
  if (!mount) return AFE_ERR_INVALID_ARG;
  // Rotationd::FromEulerYPR(-90 deg, 0, -90 deg), main.cpp:123-125 with Rotation.hpp:99-110
  const double y = -90.0 * M_PI / 180.0, p = 0.0, r = -90.0 * M_PI / 180.0;
  const double cy = std::cos(0.5 * y), sy = std::sin(0.5 * y), cp = std::cos(0.5 * p), sp = std::sin(0.5 * p);
  const double cr = std::cos(0.5 * r), sr = std::sin(0.5 * r);
  mount[0] = cy * cp * cr + sy * sp * sr;
  mount[1] = cy * cp * sr - sy * sp * cr;
  mount[2] = cy * sp * cr + sy * cp * sr;
  mount[3] = sy * cp * cr - cy * sp * sr;
  return AFE_OK;
}

extern "C" int afe_scene_create(int device, const float *triangles, int64_t n_tri, afe_scene **out) {
  if (!triangles || n_tri <= 0 || n_tri > 0x3fffffff || !out) return AFE_ERR_INVALID_ARG;
  for (int64_t i = 0; i < 9 * n_tri; i++) if (!std::isfinite(triangles[i])) return AFE_ERR_INVALID_ARG;
  int dev = 0;
  const int rc = pick_device(device, &dev);
  if (rc != AFE_OK) return rc;

  // Triangles whose own box is a good part of the scene's (a ground plane's two) stay out of the tree: inside it they
  // sit at the bottom of a chain of scene-wide nodes, and every one of those is a visit for every tile.  At most 16 of
  // them, and only if something is left for the tree; the kernel tests them before the walk.
  std::vector<int64_t> small_ix, big_ix;
  {
    Box all; all.reset();
    for (int64_t i = 0; i < 3 * n_tri; i++) all.grow(triangles + 3 * i);
    const float scene_area = all.half_area();
    for (int64_t i = 0; i < n_tri; i++) {
      Box tb; tb.reset();
      for (int v = 0; v < 3; v++) tb.grow(triangles + 9 * i + 3 * v);
      (tb.half_area() > 0.25f * scene_area ? big_ix : small_ix).push_back(i);
    }
    if (big_ix.size() > 16 || small_ix.empty()) {
      small_ix.resize((size_t)n_tri);
      for (int64_t i = 0; i < n_tri; i++) small_ix[(size_t)i] = i;
      big_ix.clear();
    }
  }
  const int64_t n_small = (int64_t)small_ix.size();
  std::vector<float> small_tris;
  if (!big_ix.empty()) {
    small_tris.resize((size_t)9 * n_small);
    for (int64_t i = 0; i < n_small; i++) std::memcpy(small_tris.data() + 9 * i, triangles + 9 * small_ix[(size_t)i], 9 * sizeof(float));
  }
  Builder b;
  b.tri = big_ix.empty() ? triangles : small_tris.data();
  b.build(n_small);
  if (b.max_depth > kStack) {   // degenerate input (e.g. many coincident centroids): balanced tree instead
    b.median_only = true;
    b.build(n_small);
    if (b.max_depth > kStack) return AFE_ERR_OUT_OF_RANGE;
  }
  // triangles in leaf order (then the ones kept out of the tree): vertex 0 and the edges in double (the checker's own
  // expressions, oracle: e = (double)v_k - (double)v_0), and the triangle's box inflated like a node's
  std::vector<TriRec> packed((size_t)n_tri);
  for (int64_t i = 0; i < n_tri; i++) {
    const float *src = triangles + 9 * (i < n_small ? small_ix[(size_t)b.order[(size_t)i]] : big_ix[(size_t)(i - n_small)]);
    TriRec &T = packed[(size_t)i];
    Box tb; tb.reset();
    for (int v = 0; v < 3; v++) tb.grow(src + 3 * v);
    BvhNode inflated;
    b.set_bounds(inflated, tb);
    for (int k = 0; k < 3; k++) {
      T.v0[k] = (double)src[k];
      T.e1[k] = (double)src[3 + k] - T.v0[k];
      T.e2[k] = (double)src[6 + k] - T.v0[k];
      T.box[2 * k] = inflated.lo[k];
      T.box[2 * k + 1] = inflated.hi[k];
#ifdef AFE_RENDER_FP32_CEILING
      T.v0f[k] = (float)T.v0[k]; T.e1f[k] = (float)T.e1[k]; T.e2f[k] = (float)T.e2[k];
#endif
    }
  }
  afe_scene *s = new afe_scene();
  s->device = dev;
  s->n_tri = n_tri;
  s->n_nodes = (int64_t)b.nodes.size();
  s->depth = b.max_depth;
  s->big_first = (uint32_t)n_small;
  s->n_big = (uint32_t)big_ix.size();
  Box all; all.reset();
  for (int64_t i = 0; i < 3 * n_tri; i++) all.grow(triangles + 3 * i);
  for (int k = 0; k < 3; k++) { s->bounds[k] = all.lo[k]; s->bounds[3 + k] = all.hi[k]; }
  std::vector<PairNode> pairs = b.pairs(0);
  s->n_pairs = (int64_t)pairs.size();
  if (pairs.size() * sizeof(PairNode) > 0xffffffffull) { delete s; return AFE_ERR_OUT_OF_RANGE; }   // 32-bit child offsets
  for (unsigned octant = 1; octant < 8; octant++) {
    const std::vector<PairNode> mirrored = b.pairs(octant);
    pairs.insert(pairs.end(), mirrored.begin(), mirrored.end());
  }
  // the triangles' boxes once more, mirrored like the node boxes of each octant copy ({lo, hi} -> {-hi, -lo})
  std::vector<float> tribox((size_t)8 * n_tri * 8, 0.0f);
  for (unsigned octant = 0; octant < 8; octant++)
    for (int64_t i = 0; i < n_tri; i++) {
      float *B = tribox.data() + ((size_t)octant * n_tri + i) * 8;
      const float *src = packed[(size_t)i].box;
      for (int k = 0; k < 3; k++) {
        const bool mirrored = (octant >> k) & 1u;
        B[2 * k] = mirrored ? -src[2 * k + 1] : src[2 * k];
        B[2 * k + 1] = mirrored ? -src[2 * k] : src[2 * k + 1];
      }
    }
  if (hipMalloc((void **)&s->pairs, pairs.size() * sizeof(PairNode)) != hipSuccess ||
      hipMalloc((void **)&s->tribox, tribox.size() * sizeof(float)) != hipSuccess ||
      hipMemcpy(s->tribox, tribox.data(), tribox.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
      hipMalloc((void **)&s->tris, packed.size() * sizeof(TriRec)) != hipSuccess ||
      hipMemcpy(s->pairs, pairs.data(), pairs.size() * sizeof(PairNode), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(s->tris, packed.data(), packed.size() * sizeof(TriRec), hipMemcpyHostToDevice) != hipSuccess) {
    afe_scene_destroy(s);
    return AFE_ERR_HIP;
  }
  *out = s;
  return AFE_OK;
}

// Pure host: build the hierarchy for a mesh and verify it (every triangle in exactly one leaf, every
// leaf's and inner node's box containing what hangs below it; then the eight mirrored PairNode arrays the
// kernel walks); no GPU needed.
extern "C" int afe_scene_check_hierarchy(const float *triangles, int64_t n_tri, int64_t *n_nodes, int *depth,
                                         int *max_leaf) {
  if (!triangles || n_tri <= 0 || n_tri > 0x3fffffff) return AFE_ERR_INVALID_ARG;
  for (int64_t i = 0; i < 9 * n_tri; i++) if (!std::isfinite(triangles[i])) return AFE_ERR_INVALID_ARG;
  Builder b;
  b.tri = triangles;
  b.build(n_tri);
  if (b.max_depth > kStack) { b.median_only = true; b.build(n_tri); }
  if (b.max_depth > kStack) return AFE_ERR_OUT_OF_RANGE;
  std::vector<int> seen((size_t)n_tri, 0);
  int worst_leaf = 0;
  // children's boxes inside the parent's, leaves cover their triangles
  for (size_t k = 0; k < b.nodes.size(); k++) {
    const BvhNode &nd = b.nodes[k];
    if (nd.b > 0) {
      worst_leaf = std::max(worst_leaf, (int)nd.b);
      for (int q = 0; q < nd.b; q++) {
        const int32_t t = b.order[(size_t)nd.a + q];
        seen[(size_t)t]++;
        for (int v = 0; v < 3; v++)
          for (int a = 0; a < 3; a++) {
            const float c = triangles[9 * (int64_t)t + 3 * v + a];
            if (!(c > nd.lo[a] && c < nd.hi[a])) return AFE_ERR_OUT_OF_RANGE;   // strictly inside the inflated box
          }
      }
    } else {
      if (nd.a <= (int32_t)k || (size_t)nd.a + 1 >= b.nodes.size()) return AFE_ERR_OUT_OF_RANGE;
      for (int c = 0; c < 2; c++)
        for (int a = 0; a < 3; a++) {
          const BvhNode &ch = b.nodes[(size_t)nd.a + c];
          // a child's inflation is computed from a smaller magnitude, so allow it the same slack again
          const float slack = 3e-4f + 4e-6f * std::max(std::fabs(nd.lo[a]), std::fabs(nd.hi[a]));
          if (ch.lo[a] < nd.lo[a] - slack || ch.hi[a] > nd.hi[a] + slack) return AFE_ERR_OUT_OF_RANGE;
        }
    }
  }
  for (int64_t i = 0; i < n_tri; i++) if (seen[(size_t)i] != 1) return AFE_ERR_OUT_OF_RANGE;
  // the kernel's form, all eight mirrored copies: walked from the root every copy reaches every triangle
  // exactly once (a single-leaf mesh: twice, by construction), child references stay inside the array and
  // point forward, every child box is the builder's box mirrored, and across a split along a mirrored axis
  // the children are exchanged (so that the first child is always the one on the lower mirrored side)
  const std::vector<PairNode> plain = b.pairs(0);
  const bool single_leaf = b.nodes[0].b > 0;
  for (unsigned octant = 0; octant < 8; octant++) {
    const std::vector<PairNode> pr = b.pairs(octant);
    if (pr.size() != plain.size()) return AFE_ERR_OUT_OF_RANGE;
    std::vector<int> hits((size_t)n_tri, 0);
    std::vector<uint32_t> todo(1, 0u);
    size_t visited = 0;
    while (!todo.empty()) {
      const uint32_t off = todo.back();
      todo.pop_back();
      if (off % sizeof(PairNode) || off / sizeof(PairNode) >= pr.size() || ++visited > pr.size()) return AFE_ERR_OUT_OF_RANGE;
      const PairNode &p = pr[off / sizeof(PairNode)], &q = plain[off / sizeof(PairNode)];
      const unsigned axis = p.meta & 3u;
      if (axis > 2 || axis != (q.meta & 3u) || (p.meta >> 24) != (q.meta >> 24)) return AFE_ERR_OUT_OF_RANGE;
      const bool exchanged = (octant >> axis) & 1u;
      const unsigned pl = (p.meta >> 8) & 255u, prc = (p.meta >> 16) & 255u, ql = (q.meta >> 8) & 255u, qr = (q.meta >> 16) & 255u;
      if (pl != (exchanged ? qr : ql) || prc != (exchanged ? ql : qr)) return AFE_ERR_OUT_OF_RANGE;
      for (int c = 0; c < 2; c++) {
        const float *box = c ? p.box_r : p.box_l;
        const float *orig = (c != 0) != exchanged ? q.box_r : q.box_l;   // the unmirrored copy's same child
        for (int k = 0; k < 3; k++) {
          const bool m = (octant >> k) & 1u;
          if (box[2 * k] != (m ? -orig[2 * k + 1] : orig[2 * k]) || box[2 * k + 1] != (m ? -orig[2 * k] : orig[2 * k + 1]) ||
              !(box[2 * k] < box[2 * k + 1]))
            return AFE_ERR_OUT_OF_RANGE;
        }
        const uint32_t ref = c ? p.right : p.left, count = (p.meta >> (c ? 16 : 8)) & 255u;
        if (count == 0) {
          if (ref <= off) return AFE_ERR_OUT_OF_RANGE;
          todo.push_back(ref);
        } else {
          if ((int64_t)ref + count > n_tri) return AFE_ERR_OUT_OF_RANGE;
          for (uint32_t t = 0; t < count; t++) {
            hits[(size_t)ref + t]++;
            const float *tri = triangles + 9 * (int64_t)b.order[(size_t)ref + t];
            for (int v = 0; v < 3; v++)
              for (int k = 0; k < 3; k++) {
                const float cm = ((octant >> k) & 1u) ? -tri[3 * v + k] : tri[3 * v + k];
                if (!(cm > box[2 * k] && cm < box[2 * k + 1])) return AFE_ERR_OUT_OF_RANGE;
              }
          }
        }
      }
    }
    for (int64_t i = 0; i < n_tri; i++) if (hits[(size_t)i] != (single_leaf ? 2 : 1)) return AFE_ERR_OUT_OF_RANGE;
  }
  if (n_nodes) *n_nodes = (int64_t)b.nodes.size();
  if (depth) *depth = b.max_depth;
  if (max_leaf) *max_leaf = worst_leaf;
  return AFE_OK;
}

extern "C" void afe_scene_destroy(afe_scene *s) {
  if (!s) return;
  for (const afe_scene::RayTable &t : s->ray_tables) (void)hipFree(t.uv);
  for (afe_scene::EntryTable &t : s->entry_tables) { if (t.done) { (void)hipEventSynchronize(t.done); (void)hipEventDestroy(t.done); } if (t.p) (void)hipFree(t.p); }
  if (s->pairs) (void)hipFree(s->pairs);
  if (s->tris) (void)hipFree(s->tris);
  if (s->tribox) (void)hipFree(s->tribox);
  delete s;
}

extern "C" int afe_scene_set_walk(afe_scene *s, int mode) {
  if (!s || (mode != 0 && mode != 1)) return AFE_ERR_INVALID_ARG;
  s->plain_walk_only = mode;
  return AFE_OK;
}

extern "C" int afe_scene_info(const afe_scene *s, int64_t *n_tri, int64_t *n_nodes, int *depth, double bounds[6]) {
  if (!s) return AFE_ERR_INVALID_ARG;
  if (n_tri) *n_tri = s->n_tri;
  if (n_nodes) *n_nodes = s->n_nodes;
  if (depth) *depth = s->depth;
  if (bounds) for (int k = 0; k < 6; k++) bounds[k] = s->bounds[k];
  return AFE_OK;
}

extern "C" int afe_render_depth(afe_scene *s, const afe_camera *cam, int64_t n_views, const double *pos,
                                const double *att, const double mount[4], uint16_t *depth_out, float *kernel_ms) {
  if (!s || !camera_ok(cam) || n_views < 0 || !pos || !att || !depth_out) return AFE_ERR_INVALID_ARG;
  if (n_views == 0) return AFE_OK;     // (an empty request is answered, like a getter of no vehicles)
  if (hipSetDevice(s->device) != hipSuccess) return AFE_ERR_HIP;
  const size_t px = (size_t)cam->width * cam->height;
  DevBuf d_pos, d_att, d_pose, d_out;
  if (!d_pos.upload(pos, (size_t)n_views * 24) || !d_att.upload(att, (size_t)n_views * 32) ||
      !d_pose.alloc((size_t)n_views * 96) || !d_out.alloc((size_t)n_views * px * 2))
    return AFE_ERR_HIP;
  int rc = launch_poses(d_pos.p, nullptr, d_att.p, n_views, 0, n_views, 8, mount, (double *)d_pose.p, nullptr);
  if (rc != AFE_OK) return rc;
  float ms = 0;
  rc = launch_render(s, cam, n_views, (const double *)d_pose.p, (uint16_t *)d_out.p, nullptr, &ms);
  if (rc != AFE_OK) return rc;
  if (kernel_ms) *kernel_ms = ms;
  if (hipMemcpy(depth_out, d_out.p, (size_t)n_views * px * 2, hipMemcpyDeviceToHost) != hipSuccess) return AFE_ERR_HIP;
  return AFE_OK;
}

// The counting build of the same kernel over explicit poses: what the traversal did, for the roofline
// accounting of bench.py.  stats: [0] nodes visited (per wave), [1] triangle box tests (per wave),
// [2] triangle double-precision tests actually executed (per wave), [3] triangle box tests (per ray),
// [4] double-precision tests (per ray), [5] rays, [6] waves (8 x 8 tiles), [7] distinct triangles that are some ray's closest hit, summed over the waves.
extern "C" int afe_render_depth_stats(afe_scene *s, const afe_camera *cam, int64_t n_views, const double *pos,
                                      const double *att, const double mount[4], uint64_t stats[8], float *kernel_ms) {
  if (!s || !camera_ok(cam) || n_views <= 0 || !pos || !att || !stats) return AFE_ERR_INVALID_ARG;
  if (hipSetDevice(s->device) != hipSuccess) return AFE_ERR_HIP;
  const size_t px = (size_t)cam->width * cam->height;
  DevBuf d_pos, d_att, d_pose, d_out, d_cnt;
  if (!d_pos.upload(pos, (size_t)n_views * 24) || !d_att.upload(att, (size_t)n_views * 32) ||
      !d_pose.alloc((size_t)n_views * 96) || !d_out.alloc((size_t)n_views * px * 2) || !d_cnt.alloc(64))
    return AFE_ERR_HIP;
  if (hipMemset(d_cnt.p, 0, 64) != hipSuccess) return AFE_ERR_HIP;
  int rc = launch_poses(d_pos.p, nullptr, d_att.p, n_views, 0, n_views, 8, mount, (double *)d_pose.p, nullptr);
  if (rc != AFE_OK) return rc;
  float ms = 0;
  rc = launch_render(s, cam, n_views, (const double *)d_pose.p, (uint16_t *)d_out.p, nullptr, &ms,
                     (unsigned long long *)d_cnt.p);
  if (rc != AFE_OK) return rc;
  if (kernel_ms) *kernel_ms = ms;
  unsigned long long host[8];
  if (hipMemcpy(host, d_cnt.p, 64, hipMemcpyDeviceToHost) != hipSuccess) return AFE_ERR_HIP;
  for (int k = 0; k < 8; k++) stats[k] = host[k];
  return AFE_OK;
}

extern "C" int afe_render_depth_engine(afe_engine *e, afe_scene *s, const afe_camera *cam, int64_t first,
                                       int64_t count, const double mount[4], void *depth_out, int out_is_device,
                                       float *kernel_ms) {
  if (!e || !s || !camera_ok(cam) || count < 0 || first < 0 || !depth_out) return AFE_ERR_INVALID_ARG;
  afe_device_view view;
  view.struct_bytes = sizeof(view);
  int rc = engine_device_view(e, &view);
  if (rc != AFE_OK) return rc;
  if (first > view.n_vehicles || count > view.n_vehicles - first) return AFE_ERR_OUT_OF_RANGE;   // (no sum: it can wrap)
  if (count == 0) return AFE_OK;
  hipStream_t stream = nullptr;
  int device = 0;
  engine_stream_device(e, (void **)&stream, &device);
  if (device != s->device) return AFE_ERR_INVALID_ARG;
  if (hipSetDevice(device) != hipSuccess) return AFE_ERR_HIP;
  const size_t px = (size_t)cam->width * cam->height;
  DevBuf d_pose, d_out;
  if (!d_pose.alloc((size_t)count * 96)) return AFE_ERR_HIP;
  uint16_t *out_dev = (uint16_t *)depth_out;
  if (!out_is_device) {
    if (!d_out.alloc((size_t)count * px * 2)) return AFE_ERR_HIP;
    out_dev = (uint16_t *)d_out.p;
  }
  rc = launch_poses(view.pos, view.pos_anchor_xy, view.att, view.stride, first, count, view.state_elem_size, mount, (double *)d_pose.p,
                    stream);
  if (rc != AFE_OK) return rc;
  float ms = 0;
  rc = launch_render(s, cam, count, (const double *)d_pose.p, out_dev, stream, &ms);  // synchronises (timing)
  if (rc != AFE_OK) return rc;
  if (kernel_ms) *kernel_ms = ms;
  if (!out_is_device && hipMemcpy(depth_out, out_dev, (size_t)count * px * 2, hipMemcpyDeviceToHost) != hipSuccess)
    return AFE_ERR_HIP;
  return AFE_OK;
}

extern "C" int afe_device_alloc(int device, uint64_t bytes, void **out) {
  if (!out || bytes == 0) return AFE_ERR_INVALID_ARG;
  int dev = 0;
  const int rc = pick_device(device, &dev);
  if (rc != AFE_OK) return rc;
  return hipMalloc(out, bytes) == hipSuccess ? AFE_OK : AFE_ERR_HIP;
}

extern "C" int afe_device_free(void *p) {
  if (!p) return AFE_OK;
  return hipFree(p) == hipSuccess ? AFE_OK : AFE_ERR_HIP;
}

extern "C" int afe_device_download(void *host_dst, const void *dev_src, uint64_t bytes) {
  if (!host_dst || !dev_src) return AFE_ERR_INVALID_ARG;
  if (hipDeviceSynchronize() != hipSuccess) return AFE_ERR_HIP;
  return hipMemcpy(host_dst, dev_src, bytes, hipMemcpyDeviceToHost) == hipSuccess ? AFE_OK : AFE_ERR_HIP;
}
