// afe_world.hip -- shared-world queries on the gathered position buffer:
// the consumers of the one inter-vehicle exchange the path has.
//
//   * nearest neighbour of every local vehicle among the whole ensemble, by a uniform grid
//     (cell list built by a counting sort on the device, 3x3x3 cell search, exact: rings are
//     added until no unvisited cell can hold a closer point; a shard asking for its block of a larger
//     gathered ensemble sorts only its own surroundings, see GridDesc::filtered);
//   * UWB-style ranging between requester / responder pairs, the batched form of
//     Simulation::UWBNetwork::Run (Components/Components/Simulation/UWBNetwork.cpp:22-89).
//
// Both read the planar fp32 xyz buffer afe_pack_positions / afe_gather_positions produce
// (12 B per vehicle, the all-gather payload of SURVEY 8e).  HBM / cache-latency bound integer and
// fp32 work; no contraction anywhere, hence no MFMA.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "afe_render.h"   // engine_stream_device
#include "afe_world.h"

namespace afe {
namespace {

struct GridDesc {
  float min[3];
  float inv_h, h;
  int n[3];          // cells per axis
  int64_t n_cells;   // n[0]*n[1]*n[2]; bin n_cells collects non-finite positions (never searched)
  // shard-local grids (this shard queries for a block of the gathered ensemble): the grid is shaped on the
  // shard's own vehicles, and vehicles of other shards farther than `keep` outside it never enter the sort
  int filtered;
  float keep_lo[3], keep_hi[3];
  int64_t self_first, self_count;
};
#define AFE_WORLD_DROPPED 0xffffffffu
#define AFE_WORLD_BRUTE_CHUNK 32768
#define AFE_WORLD_KEY_NONE 0x7f7fc99effffffffull   // (3.4e38f, no index): what a brute-force scan that finds nobody keeps

__device__ __forceinline__ int ordered(float f) {
  const int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7fffffff;
}
__host__ __device__ __forceinline__ float unordered(int i) {
  const int j = i >= 0 ? i : i ^ 0x7fffffff;
  float f;
  memcpy(&f, &j, 4);
  return f;
}
__device__ __forceinline__ bool finite3(float x, float y, float z) {
  return fabsf(x) < 3.0e38f && fabsf(y) < 3.0e38f && fabsf(z) < 3.0e38f;   // false for NaN and inf
}

// min / max of the finite positions (lo[3], hi[3] as order-preserving ints) and their first two
// moments (sum x y z, sum xx yy zz, count; double): one partial record per workgroup, reduced on the
// host, which reads them back anyway to decide the grid -- no atomics (thousands of waves adding into
// the same dozen words serialise at the memory side: measured 350 us, against 6 us this way)
#define AFE_WORLD_BOUNDS_BLOCKS 256
struct BoundsPartial {
  int lo[3], hi[3];
  double m[7];
};

__global__ void __launch_bounds__(256) world_bounds_kernel(const float *__restrict__ xyz_all, int64_t stride, int64_t first, int64_t n,
                                                           BoundsPartial *__restrict__ part) {
  const float *__restrict__ xyz = xyz_all + first;   // planar [3][stride], the block [first, first + n) of it
  __shared__ BoundsPartial wave_part[4];
  int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
  double m[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float x = xyz[i], y = xyz[stride + i], z = xyz[2 * stride + i];
    if (finite3(x, y, z)) {
      const int ox = ordered(x), oy = ordered(y), oz = ordered(z);
      lo[0] = min(lo[0], ox); hi[0] = max(hi[0], ox);
      lo[1] = min(lo[1], oy); hi[1] = max(hi[1], oy);
      lo[2] = min(lo[2], oz); hi[2] = max(hi[2], oz);
      m[0] += x; m[1] += y; m[2] += z;
      m[3] += (double)x * x; m[4] += (double)y * y; m[5] += (double)z * z;
      m[6] += 1.0;
    }
  }
#pragma unroll
  for (int c = 0; c < 3; c++) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
      lo[c] = min(lo[c], __shfl_xor(lo[c], s));
      hi[c] = max(hi[c], __shfl_xor(hi[c], s));
    }
  }
#pragma unroll
  for (int c = 0; c < 7; c++) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) m[c] += __shfl_xor(m[c], s);
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int c = 0; c < 3; c++) { wave_part[wave].lo[c] = lo[c]; wave_part[wave].hi[c] = hi[c]; }
#pragma unroll
    for (int c = 0; c < 7; c++) wave_part[wave].m[c] = m[c];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    BoundsPartial out = wave_part[0];
    for (int w = 1; w < 4; w++) {
      for (int c = 0; c < 3; c++) { out.lo[c] = min(out.lo[c], wave_part[w].lo[c]); out.hi[c] = max(out.hi[c], wave_part[w].hi[c]); }
      for (int c = 0; c < 7; c++) out.m[c] += wave_part[w].m[c];
    }
    part[blockIdx.x] = out;
  }
}

__device__ __forceinline__ void cell_of(const GridDesc &g, float x, float y, float z, int &cx, int &cy, int &cz) {
#pragma clang fp contract(off)
  cx = (int)((x - g.min[0]) * g.inv_h);
  cy = (int)((y - g.min[1]) * g.inv_h);
  cz = (int)((z - g.min[2]) * g.inv_h);
  cx = cx < 0 ? 0 : (cx >= g.n[0] ? g.n[0] - 1 : cx);
  cy = cy < 0 ? 0 : (cy >= g.n[1] ? g.n[1] - 1 : cy);
  cz = cz < 0 ? 0 : (cz >= g.n[2] ? g.n[2] - 1 : cz);
}

// counting sort, pass 1: cell of every point, histogram, slot of the point inside its cell
__global__ void __launch_bounds__(256) world_count_kernel(const float *__restrict__ xyz, int64_t n, GridDesc g,
                                                          uint32_t *__restrict__ counts, uint32_t *__restrict__ cell,
                                                          uint32_t *__restrict__ slot, uint32_t *__restrict__ left_count, uint32_t *__restrict__ words) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i == 0) { left_count[0] = 0; left_count[1] = 0; words[0] = 0; }   // this query's leftover counter; a fresh sort: nobody has moved
  if (i >= n) return;
  const float x = xyz[i], y = xyz[n + i], z = xyz[2 * n + i];
  uint32_t c = (uint32_t)g.n_cells;
  const bool fin = finite3(x, y, z);
  if (g.filtered && (uint64_t)(i - g.self_first) >= (uint64_t)g.self_count) {
    // somebody else's vehicle: it matters only if it can be the nearest neighbour of a query the rings settle,
    // i.e. if it lies within `keep` of the grid (world_query_kernel has the argument); NaN compares false: dropped
    const bool near = x >= g.keep_lo[0] && x <= g.keep_hi[0] && y >= g.keep_lo[1] && y <= g.keep_hi[1] &&
                      z >= g.keep_lo[2] && z <= g.keep_hi[2];
    if (!(fin && near)) { cell[i] = AFE_WORLD_DROPPED; return; }
  }
  if (fin) {
    int cx, cy, cz;
    cell_of(g, x, y, z, cx, cy, cz);
    c = (uint32_t)(((int64_t)cz * g.n[1] + cy) * g.n[0] + cx);
  }
  cell[i] = c;
  slot[i] = atomicAdd(&counts[c], 1u);
}

// exclusive scan of m counts in three launches (1024 per block)
__global__ void __launch_bounds__(256) world_scan_local_kernel(const uint32_t *in, uint32_t *out,   // in == out: each thread rewrites only what it read
                                                               uint32_t *__restrict__ block_sums, int64_t m) {
  __shared__ uint32_t wave_tot[4];
  const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x * 4;
  uint32_t v[4];
#pragma unroll
  for (int k = 0; k < 4; k++) v[k] = (base + k < m) ? in[base + k] : 0u;
  const uint32_t mine = v[0] + v[1] + v[2] + v[3];
  uint32_t incl = mine;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) {
    const uint32_t t = __shfl_up(incl, s);
    if (lane >= s) incl += t;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  uint32_t off = 0;
  for (int w = 0; w < wave; w++) off += wave_tot[w];
  uint32_t run = off + incl - mine;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (base + k < m) out[base + k] = run;
    run += v[k];
  }
  if (threadIdx.x == 255) block_sums[blockIdx.x] = off + incl;
}
// scan of the block sums by one workgroup (sequential over chunks of 256)
__global__ void __launch_bounds__(256) world_scan_blocks_kernel(uint32_t *__restrict__ block_sums, int64_t nb) {
  __shared__ uint32_t wave_tot[4];
  __shared__ uint32_t carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t b0 = 0; b0 < nb; b0 += 256) {
    const int64_t i = b0 + threadIdx.x;
    const uint32_t mine = i < nb ? block_sums[i] : 0u;
    uint32_t incl = mine;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
      const uint32_t t = __shfl_up(incl, s);
      if (lane >= s) incl += t;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t off = carry_s;
    for (int w = 0; w < wave; w++) off += wave_tot[w];
    if (i < nb) block_sums[i] = off + incl - mine;
    __syncthreads();
    if (threadIdx.x == 255) carry_s = off + incl;
    __syncthreads();
  }
}
__global__ void __launch_bounds__(256) world_scan_add_kernel(uint32_t *__restrict__ out, const uint32_t *__restrict__ block_sums, int64_t m) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < m) out[i] += block_sums[i >> 10];
}

// counting sort, pass 2: points into cell order as (x, y, z, global index)
// (slot[i] becomes the point's place in the cell order and ref keeps where it was: a later query may keep the order
// and only refresh the positions, world_regather_kernel)
__global__ void __launch_bounds__(256) world_scatter_kernel(const float *__restrict__ xyz, int64_t n, const uint32_t *__restrict__ starts,
                                                            const uint32_t *__restrict__ cell, uint32_t *__restrict__ slot,
                                                            uint4 *__restrict__ sorted, float *__restrict__ ref, int64_t ref_stride) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float x = xyz[i], y = xyz[n + i], z = xyz[2 * n + i];
  ref[i] = x; ref[ref_stride + i] = y; ref[2 * ref_stride + i] = z;
  const uint32_t c = cell[i];
  if (c == AFE_WORLD_DROPPED) return;
  const uint32_t at = starts[c] + slot[i];
  slot[i] = at;
  sorted[at] = make_uint4(__float_as_uint(x), __float_as_uint(y), __float_as_uint(z), (uint32_t)i);
}

// "Which workgroup of this launch finishes last?" -- thread 0 of every workgroup takes a ticket when its workgroup's
// stores are acknowledged; exactly one call per launch returns true.  Two levels (64 shard counters, a line each, then one
// top counter): thousands of tickets on ONE word would queue up at the memory side (~12 ns each).  No cache write-back is
// involved (an agent-scope fence per workgroup costs microseconds and made the fused launches slower than the separate
// ones they replaced: measured): what the last workgroup reads of the others' work was stored with agent-scope atomics.
#define AFE_WORLD_TICKET_WORDS (16 * 65)
__device__ __forceinline__ bool last_workgroup(uint32_t *tickets) {
  const unsigned nb = gridDim.x, shard = blockIdx.x & 63u;
  const unsigned in_shard = (nb - shard + 63u) / 64u, shards = nb < 64u ? nb : 64u;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (__hip_atomic_fetch_add(&tickets[16 * shard], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 != in_shard) return false;
  __hip_atomic_store(&tickets[16 * shard], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // for the next launch
  if (__hip_atomic_fetch_add(&tickets[16 * 64], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 != shards) return false;
  __hip_atomic_store(&tickets[16 * 64], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
}

// A query that KEEPS the cell order of an earlier one (afe_set_neighbour_sort_reuse): every point's current position
// goes to the place the sort gave it, and words[0] becomes an upper bound of how far any point -- sorted or dropped --
// has moved since the sort (float bits; non-negative floats order like their bit patterns).  The query -- which looks
// around the cell its position of TODAY falls into -- then trusts a ring only up to (r - 0.05) h less that bound: a point
// listed outside the rings was at least r h away from anywhere in the query's cell when it was sorted.  A point that was non-finite at the sort (dead bin, never scanned) and
// is finite now would be missed: the bound becomes infinite -- no ring settles anything, the brute force answers --
// and the host is told to sort again (host_flag, pinned memory).
__global__ void __launch_bounds__(256) world_regather_kernel(const float *__restrict__ xyz, int64_t n, const uint32_t *__restrict__ cell,
                                                             const uint32_t *__restrict__ slot, const float *__restrict__ ref, int64_t ref_stride,
                                                             uint4 *__restrict__ sorted, float *partial, uint32_t *__restrict__ left_count,
                                                             uint32_t *words, uint32_t *host_flag, uint32_t *tickets) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i == 0) { left_count[0] = 0; left_count[1] = 0; }   // this query's leftover counter
  float moved = 0.0f;
  if (i < n) {
    const float x = xyz[i], y = xyz[n + i], z = xyz[2 * n + i];
    const float rx = ref[i], ry = ref[ref_stride + i], rz = ref[2 * ref_stride + i];
    const bool fin = finite3(x, y, z), was = finite3(rx, ry, rz);
    if (fin && was) {
      const float dx = x - rx, dy = y - ry, dz = z - rz;
      moved = sqrtf((dx * dx + dy * dy) + dz * dz) * 1.000001f + 1e-30f;   // rounded up
      if (!(moved < 3.0e38f)) moved = __builtin_inff();
    } else if (fin && !was) {
      moved = __builtin_inff();
    }
    const uint32_t c = cell[i];
    if (c != AFE_WORLD_DROPPED) sorted[slot[i]] = make_uint4(__float_as_uint(x), __float_as_uint(y), __float_as_uint(z), (uint32_t)i);
  }
  // one partial maximum per workgroup (thousands of waves taking an atomic maximum on one word serialise at the memory
  // side: measured, that alone made the kept order slower than the sort); the workgroup that finishes LAST -- a ticket --
  // reduces the partials, so the bound is in words[0] when the launch ends and no second launch is needed
  __shared__ float wave_max[4];
  __shared__ bool last_block;
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) moved = fmaxf(moved, __shfl_xor(moved, s));
  if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = moved;
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_store(&partial[blockIdx.x], fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_block = last_workgroup(tickets);
  }
  __syncthreads();
  if (!last_block) return;
  float all = 0.0f;
  for (unsigned k = threadIdx.x; k < gridDim.x; k += 256) all = fmaxf(all, __hip_atomic_load(&partial[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) all = fmaxf(all, __shfl_xor(all, s));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = all;
  __syncthreads();
  if (threadIdx.x == 0) {
    all = fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]));
    words[0] = __float_as_uint(all);
    if (all > 3.0e38f) __hip_atomic_store(host_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// squared distance exactly as the brute-force definition rounds it (three products, two sums, fp32)
__device__ __forceinline__ float dist2(float ax, float ay, float az, float x, float y, float z) {
#pragma clang fp contract(off)
  const float dx = ax - x, dy = ay - y, dz = az - z;
  return (dx * dx + dy * dy) + dz * dz;
}
// the closest point wins; among equally close ones the lowest global index (what a 0..n-1 scan with `<` keeps)
__device__ __forceinline__ void consider(float d, int j, int me, float &best, int &best_j) {
  if (j != me && (d < best || (d == best && j < best_j))) { best = d; best_j = j; }
}

// candidates sorted[a .. b): four loads in flight per trip (the tail re-reads the last entry, which changes
// no minimum) -- one dependent cache round trip per four candidates instead of one each.  (Eight per trip, and the
// nine rows' start offsets of ring 1 fetched together ahead of the scans, were measured and are slower: 0.45
// against 0.40 ms on the smeared 2^20 world.  So are two cooperative forms in which a workgroup, or a single wave,
// stages the nine cell stretches around a run of 32 cells in LDS with coalesced loads and the vehicles scan from
// there: 0.57 and 0.42 ms -- lanes in cell order already share most of their cache lines.)
__device__ __forceinline__ void scan_range(const uint4 *__restrict__ sorted, uint32_t a, uint32_t b, float x, float y, float z, int me,
                                           float &best, int &best_j) {
  for (uint32_t k = a; k < b; k += 4) {
    const uint32_t last = b - 1;
    const uint4 p0 = sorted[k], p1 = sorted[min(k + 1, last)], p2 = sorted[min(k + 2, last)], p3 = sorted[min(k + 3, last)];
    consider(dist2(__uint_as_float(p0.x), __uint_as_float(p0.y), __uint_as_float(p0.z), x, y, z), (int)p0.w, me, best, best_j);
    consider(dist2(__uint_as_float(p1.x), __uint_as_float(p1.y), __uint_as_float(p1.z), x, y, z), (int)p1.w, me, best, best_j);
    consider(dist2(__uint_as_float(p2.x), __uint_as_float(p2.y), __uint_as_float(p2.z), x, y, z), (int)p2.w, me, best, best_j);
    consider(dist2(__uint_as_float(p3.x), __uint_as_float(p3.y), __uint_as_float(p3.z), x, y, z), (int)p3.w, me, best, best_j);
  }
}

#define AFE_WORLD_MAX_RING 6

// Filtered grids: vehicles of other shards outside the keep box (the grid and AFE_WORLD_MAX_RING + 1 cells around
// it) were never sorted.  Every one of them is at least `clear` -- the query's distance to the faces of the keep
// box -- away, so the rings may settle a query only with best < clear^2 (then no dropped vehicle can be the
// answer, nor tie with it).  Inside the grid clear >= (AFE_WORLD_MAX_RING + 1) h, more than any ring reaches, so
// nothing changes there; a fly-away clamped into a boundary cell settles only if its neighbour is nearer than
// the keep box's faces; and "the rings cover the whole grid" proves nothing about what was dropped.  Whatever
// the rings cannot settle goes to the brute force, which reads the whole gathered buffer.
__device__ __forceinline__ float clearance(const GridDesc &g, float x, float y, float z) {
#pragma clang fp contract(off)
  if (!g.filtered) return 3.4e38f;
  const float c = fminf(fminf(fminf(x - g.keep_lo[0], g.keep_hi[0] - x), fminf(y - g.keep_lo[1], g.keep_hi[1] - y)),
                        fminf(z - g.keep_lo[2], g.keep_hi[2] - z));
  return fmaxf(c - 0.05f * g.h, 0.0f);   // roundings of the differences: far below a twentieth of a cell
}

// rings r_first, r_first + 1, ... around cell (cx, cy, cz) through the cache; true if the query is settled
__device__ bool ring_search(const uint4 *__restrict__ sorted, const uint32_t *__restrict__ starts, const GridDesc &g, float qx, float qy,
                            float qz, int me, int cx, int cy, int cz, float clear, float moved, int r_first, float &best, int &best_j) {
  const int nx = g.n[0], ny = g.n[1], nz = g.n[2];
  bool done = false;
  for (int r = r_first; r <= AFE_WORLD_MAX_RING && !done; r++) {
    for (int dz = -r; dz <= r; dz++) {
      const int z = cz + dz;
      if (z < 0 || z >= nz) continue;
      for (int dy = -r; dy <= r; dy++) {
        const int y = cy + dy;
        if (y < 0 || y >= ny) continue;
        const int64_t row = ((int64_t)z * ny + y) * nx;
        const bool shell_row = (dz == -r || dz == r || dy == -r || dy == r) || r == 1;
        if (shell_row) {   // the whole x-run of the row is new: its cells are contiguous in the cell order
          const int xa = max(cx - r, 0), xb = min(cx + r, nx - 1);
          scan_range(sorted, starts[row + xa], starts[row + xb + 1], qx, qy, qz, me, best, best_j);
        } else {           // interior row of a wider ring: only its two end cells are new
          if (cx - r >= 0) scan_range(sorted, starts[row + cx - r], starts[row + cx - r + 1], qx, qy, qz, me, best, best_j);
          if (cx + r < nx) scan_range(sorted, starts[row + cx + r], starts[row + cx + r + 1], qx, qy, qz, me, best, best_j);
        }
      }
    }
    // 0.05 h: slack for the fp32 cell coordinates (<= 65536 per axis).  moved: the cell order may be an earlier query's
    // (world_regather_kernel) -- whatever is listed outside the rings has come at most that much closer since
    const float reach = fminf(((float)r - 0.05f) * g.h - moved, clear);
    const bool covers_all = cx - r <= 0 && cx + r >= nx - 1 && cy - r <= 0 && cy + r >= ny - 1 && cz - r <= 0 && cz + r >= nz - 1;
    done = (covers_all && !g.filtered && moved < 3.0e38f) || (reach > 0.0f && best < reach * reach);
    if (covers_all) break;
  }
  return done;
}

__device__ __forceinline__ void finish_query(bool done, int64_t local, float best, int best_j, float *__restrict__ dist2_out,
                                             int32_t *__restrict__ index_out, uint32_t *__restrict__ leftover_count,
                                             int32_t *__restrict__ leftover, uint64_t *__restrict__ leftover_keys) {
  if (done) { dist2_out[local] = best; index_out[local] = best_j; }
  else {
    const uint32_t k = atomicAdd(leftover_count, 1u);
    __hip_atomic_store(&leftover[k], (int32_t)local, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (read by the launch's last workgroup)
    leftover_keys[k] = AFE_WORLD_KEY_NONE;
  }
}

// One lane per point IN CELL ORDER (neighbouring lanes sit in the same or adjacent cells, so their
// reads share cache lines); lanes whose point is not one of this shard's vehicles retire at once.
// Exactness: every point outside the (2r+1)^3 block of cells around the query's cell is at least
// r*h away, so once the best squared distance is below ((r - 0.05) h)^2 (the margin covers the fp32
// cell assignment) no unvisited cell can improve it.  Queries still unresolved after AFE_WORLD_MAX_RING
// rings (isolated vehicles) go to a leftover list that a brute-force kernel finishes.
__device__ __forceinline__ void query_one(const uint4 *__restrict__ sorted, int64_t n_all, const uint32_t *__restrict__ starts,
                                          const GridDesc &g, int64_t first_global, int64_t n_self, float *__restrict__ dist2_out,
                                          int32_t *__restrict__ index_out, uint32_t *__restrict__ leftover_count,
                                          int32_t *__restrict__ leftover, uint64_t *__restrict__ leftover_keys,
                                          const uint32_t *__restrict__ words) {
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= n_all || s >= (int64_t)starts[g.n_cells + 1]) return;   // the end sentinel: how many points were sorted
  const uint4 qb = sorted[s];
  const float3 q = make_float3(__uint_as_float(qb.x), __uint_as_float(qb.y), __uint_as_float(qb.z));
  const int me = (int)qb.w;
  const int64_t local = (int64_t)me - first_global;
  if (local < 0 || local >= n_self) return;
  float best = 3.4e38f;
  int best_j = -1;
  if (!finite3(q.x, q.y, q.z)) { dist2_out[local] = best; index_out[local] = best_j; return; }
  int cx, cy, cz;
  cell_of(g, q.x, q.y, q.z, cx, cy, cz);
  bool done = false;
  const float moved = __uint_as_float(words[0]);   // 0 unless the cell order is an earlier query's
  const float clear = clearance(g, q.x, q.y, q.z) - moved;   // a dropped vehicle may have come that much closer
  if (clear > 0.0f) done = ring_search(sorted, starts, g, q.x, q.y, q.z, me, cx, cy, cz, clear, moved, 1, best, best_j);
  finish_query(done, local, best, best_j, dist2_out, index_out, leftover_count, leftover, leftover_keys);
}

// The workgroup that finishes LAST (a ticket) closes the query: it tells the host how many queries the rings left over
// (pinned memory, informational) and answers them itself by the brute-force definition, one after the other, all 256
// threads on each -- but only while that is little WORK: n_left x n_all distance evaluations within AFE_WORLD_TAIL_WORK
// (a handful of isolated vehicles in a small world, ~20 us of one compute unit).  Anything more is not one compute
// unit's work (twenty isolated vehicles among 2^20 points were 1.5 ms per query on one unit, 12 ms at 8 x 2^20; 10^4
// leftovers among 10^6 points 0.4 s): the count goes to `big_count`, and the launch that follows every query
// (world_brute_chunks_kernel: 1 024 workgroups that leave at once when the word is zero) shares it over the device.
// (Round-5 advisor: the gate used to be a COUNT of 32, whatever the world's size.)
#define AFE_WORLD_TAIL_WORK (1ull << 22)
__global__ void __launch_bounds__(256) world_query_kernel(const uint4 *__restrict__ sorted, int64_t n_all, const uint32_t *__restrict__ starts,
                                                          GridDesc g, int64_t first_global, int64_t n_self, float *__restrict__ dist2_out,
                                                          int32_t *__restrict__ index_out, uint32_t *leftover_count,
                                                          int32_t *leftover, uint64_t *__restrict__ leftover_keys,
                                                          uint32_t *words, const float *__restrict__ all_xyz, uint32_t *big_count, uint32_t *host_left,
                                                          uint32_t *tickets) {
  query_one(sorted, n_all, starts, g, first_global, n_self, dist2_out, index_out, leftover_count, leftover, leftover_keys, words);
  __shared__ bool last_block;
  __shared__ unsigned long long wave_key[4];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this thread's entry of the leftover list is acknowledged ...
  __syncthreads();                                   // ... and so is everybody's of this workgroup
  if (threadIdx.x == 0) last_block = last_workgroup(tickets + AFE_WORLD_TICKET_WORDS);
  __syncthreads();
  if (!last_block) return;
  const uint32_t n_left = __hip_atomic_load(leftover_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const bool in_tail = (unsigned long long)n_left * (unsigned long long)n_all <= AFE_WORLD_TAIL_WORK;     // (nothing left over: in_tail, an empty loop)
  if (threadIdx.x == 0) {
    __hip_atomic_store(host_left, n_left, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(big_count, in_tail ? 0u : n_left, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (!in_tail) return;
  for (uint32_t k = 0; k < n_left; k++) {
    const int64_t local = __hip_atomic_load(&leftover[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int me = (int)(first_global + local);
    const float x = all_xyz[me], y = all_xyz[n_all + me], z = all_xyz[2 * n_all + me];
    float best = 3.4e38f;
    int best_j = -1;
    if (finite3(x, y, z))
      for (int64_t j = threadIdx.x; j < n_all; j += 256)
        consider(dist2(all_xyz[j], all_xyz[n_all + j], all_xyz[2 * n_all + j], x, y, z), (int)j, me, best, best_j);
    unsigned long long key = best_j >= 0 ? ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)best_j : AFE_WORLD_KEY_NONE;
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) {
      const unsigned long long o = __shfl_xor(key, sft);
      key = o < key ? o : key;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) wave_key[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int wv = 1; wv < 4; wv++) key = wave_key[wv] < key ? wave_key[wv] : key;
      dist2_out[local] = __uint_as_float((uint32_t)(key >> 32));
      index_out[local] = (int32_t)(uint32_t)key;     // 0xffffffff = -1: nobody
    }
  }
}

// Brute force for listed queries.  The work is cut into (query, chunk of the ensemble) items that the whole launch
// shares -- a single left-over query is then finished by a hundred workgroups in ~0.1 ms instead of by one in 12 ms
// (8 x 2^20 positions) -- and an item's winner is merged into the query's 64-bit key with one atomic minimum:
// key = distance bits (non-negative floats order like their bit patterns) << 32 | index, so the smaller distance
// wins and among equal distances the lower index, which is the definition's tie-break.

__global__ void __launch_bounds__(256) world_brute_init_kernel(uint64_t *__restrict__ keys, const uint32_t *__restrict__ n_queries) {
  for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < *n_queries; k += gridDim.x * 256) keys[k] = AFE_WORLD_KEY_NONE;
}

// (`tickets` != nullptr: the launch's last workgroup also writes the answers out -- world_brute_finish_kernel's loop --
// so that the launch behind every grid query is ONE; it leaves at once, before any ticket, when there is nothing listed)
__global__ void __launch_bounds__(256) world_brute_chunks_kernel(const float *__restrict__ all_xyz, int64_t n_all, const int32_t *__restrict__ queries,
                                                                 const uint32_t *__restrict__ n_queries, int64_t first_global,
                                                                 unsigned long long *__restrict__ keys, uint32_t *tickets,
                                                                 float *__restrict__ dist2_out, int32_t *__restrict__ index_out) {
  __shared__ unsigned long long wave_key[4];
  __shared__ bool last_block;
  const uint32_t nq = *n_queries;
  if (nq == 0) return;
  const int64_t n_chunks = (n_all + AFE_WORLD_BRUTE_CHUNK - 1) / AFE_WORLD_BRUTE_CHUNK;
  const int64_t items = (int64_t)nq * n_chunks;
  for (int64_t item = blockIdx.x; item < items; item += gridDim.x) {
    const int64_t qi = item / n_chunks, chunk = item - qi * n_chunks;
    const int me = (int)(first_global + queries[qi]);
    const float x = all_xyz[me], y = all_xyz[n_all + me], z = all_xyz[2 * n_all + me];
    float best = 3.4e38f;
    int best_j = -1;
    if (finite3(x, y, z)) {
      const int64_t j1 = min(n_all, (chunk + 1) * AFE_WORLD_BRUTE_CHUNK);
      for (int64_t j = chunk * AFE_WORLD_BRUTE_CHUNK + threadIdx.x; j < j1; j += 256)
        consider(dist2(all_xyz[j], all_xyz[n_all + j], all_xyz[2 * n_all + j], x, y, z), (int)j, me, best, best_j);
    }
    unsigned long long key = best_j >= 0 ? ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)best_j : AFE_WORLD_KEY_NONE;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
      const unsigned long long o = __shfl_xor(key, s);
      key = o < key ? o : key;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) wave_key[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < 4; w++) key = wave_key[w] < key ? wave_key[w] : key;
      if (key != AFE_WORLD_KEY_NONE) atomicMin(&keys[qi], key);
    }
  }
  if (!tickets) return;
  __syncthreads();
  if (threadIdx.x == 0) last_block = last_workgroup(tickets);      // (its atomic minima are acknowledged: s_waitcnt inside)
  __syncthreads();
  if (!last_block) return;
  for (uint32_t k = threadIdx.x; k < nq; k += 256) {
    const int64_t local = queries[k];
    const unsigned long long key = __hip_atomic_load(&keys[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    dist2_out[local] = __uint_as_float((uint32_t)(key >> 32));
    index_out[local] = (int32_t)(uint32_t)key;   // 0xffffffff = -1: nobody
  }
}

__global__ void __launch_bounds__(256) world_brute_finish_kernel(const int32_t *__restrict__ queries, const uint32_t *__restrict__ n_queries,
                                                                 const uint64_t *__restrict__ keys, float *__restrict__ dist2_out,
                                                                 int32_t *__restrict__ index_out) {
  for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < *n_queries; k += gridDim.x * 256) {
    const int64_t local = queries[k];
    const uint64_t key = keys[k];
    dist2_out[local] = __uint_as_float((uint32_t)(key >> 32));
    index_out[local] = (int32_t)(uint32_t)key;   // 0xffffffff = -1: nobody
  }
}

// UWBNetwork::Run, UWBNetwork.cpp:66-71, for a batch of completed transactions: the noise draws do not
// depend on the positions, so the host made them in stream order (afe_uwb_range); here only
//   meas.range = float( outlier ? n * outlierStdDev : |p_req - p_res| + n * addNoiseStdDev )
// with the difference, squares, sums and square root in double like Vec3d::GetNorm2 (Vec3.hpp:113-116).
__global__ void __launch_bounds__(256) world_uwb_range_kernel(const float *__restrict__ all_xyz, int64_t n_all, const int32_t *__restrict__ req,
                                                              const int32_t *__restrict__ res, const double *__restrict__ noise_term,
                                                              const uint8_t *__restrict__ outlier, int64_t n_pairs, float *__restrict__ range) {
#pragma clang fp contract(off)
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= n_pairs) return;
  double r = noise_term[k];
  if (!outlier[k]) {
    const int a = req[k], b = res[k];
    const double dx = (double)all_xyz[a] - (double)all_xyz[b];
    const double dy = (double)all_xyz[n_all + a] - (double)all_xyz[n_all + b];
    const double dz = (double)all_xyz[2 * n_all + a] - (double)all_xyz[2 * n_all + b];
    r = sqrt(dx * dx + dy * dy + dz * dz) + r;
  }
  range[k] = (float)r;
}

}  // namespace
}  // namespace afe

using namespace afe;

// ---------------------------------------------------------------------------
// host side

struct afe_world {
  int device = 0;
  int64_t cap_points = 0;      // capacity of the per-point scratch
  int64_t cap_cells = 0;       // capacity of the per-cell scratch (cells + 1 dead bin + 1 end)
  uint32_t *counts = nullptr;  // per cell, becomes the exclusive scan (starts)
  uint32_t *block_sums = nullptr;
  uint32_t *cell = nullptr, *slot = nullptr;
  uint4 *sorted = nullptr;
  int32_t *leftover = nullptr;
  uint64_t *leftover_keys = nullptr;   // per listed query: packed (distance, index) minimum of the brute force
  int *lohi = nullptr;         // 6 ints + leftover counter [6], the explicit brute force's count [7], the query's count for the launch behind it [8]
  BoundsPartial *bounds_part = nullptr;   // one record per workgroup of the bounds kernel
  int64_t cap_self = 0;
  float *self_scratch = nullptr;
  GridDesc grid = {};
  int64_t grid_n_all = -1;      // ensemble size the grid shape was chosen for
  int64_t grid_first = -1, grid_n_self = -1;   // and the block of it the queries were for
  float grid_cell_arg = 0;      // and the caller's cell size then
  int refresh_every = 1;        // re-shape the grid every this many queries (1: always)
  int since_refresh = 0;
  // keeping the cell ORDER over several queries (world_set_sort_reuse): positions are refreshed in place, the query's
  // stopping rule allows for how far anybody has moved since the sort
  int resort_every = 1;         // sort every this many queries (1: always)
  int since_sort = 0;
  bool sort_valid = false;
  float *ref = nullptr;         // [3][cap_points]: positions at the last sort, original order
  uint32_t *words = nullptr;    // device: [0] bound on the movement since the sort (float bits)
  uint32_t *tickets = nullptr;  // device: last_workgroup counters of the regather launch, of the query launch, of the brute-force launch behind it
  uint32_t *host_flag = nullptr;   // pinned: [0] a kernel asks for a new sort, [1] queries the last finished query's rings left over (informational)
  std::string err;
};

namespace {
const int64_t kMaxCells = int64_t(1) << 22;

int wfail(afe_world *w, int status, const std::string &m) {
  if (w) w->err = m;
  return status;
}
#define W_HIP(w, call)                                                                                 \
  do {                                                                                                 \
    hipError_t err__ = (call);                                                                         \
    if (err__ != hipSuccess) return wfail((w), AFE_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(err__)); \
  } while (0)

void free_points(afe_world *w) {
  if (w->cell) (void)hipFree(w->cell);
  if (w->slot) (void)hipFree(w->slot);
  if (w->sorted) (void)hipFree(w->sorted);
  if (w->ref) (void)hipFree(w->ref);
  w->ref = nullptr; w->sort_valid = false;
  if (w->leftover) (void)hipFree(w->leftover);
  if (w->leftover_keys) (void)hipFree(w->leftover_keys);
  w->cell = w->slot = nullptr; w->sorted = nullptr; w->leftover = nullptr; w->leftover_keys = nullptr;
}

// The grid covers the CORE of the ensemble: the bounding box cut to mean +- 6 sigma per axis.  Points
// outside it are clamped into its boundary cells -- clamping never increases a cell distance, so
// the exactness argument of the query holds unchanged -- and a single fly-away vehicle no longer
// stretches the grid until everybody else shares a handful of cells.  Cell size: the caller's, or
// one that puts about two vehicles in a cell of a 4-sigma box (a uniform box is 3.5 sigma wide, a
// Gaussian blob is denser than that in its middle); flat ensembles (an orchard flight: everybody
// within a few metres of altitude) get a 2-D grid because an axis thinner than a cell collapses to
// one layer.
void choose_grid(const float lo[3], const float hi[3], const double stats[7], int64_t n, float cell_size, GridDesc &g) {
  double ext[3], core[3], rlo[3];
  const double cnt = stats[6] > 0 ? stats[6] : 1.0;
  for (int c = 0; c < 3; c++) {
    const double mean = stats[c] / cnt;
    const double var = std::max(0.0, stats[3 + c] / cnt - mean * mean);
    const double sd = std::sqrt(var);
    double a = lo[c], b = hi[c];
    if (sd > 0 && std::isfinite(sd)) { a = std::max(a, mean - 6.0 * sd); b = std::min(b, mean + 6.0 * sd); }
    if (!(b >= a)) { a = lo[c]; b = hi[c]; }
    rlo[c] = a;
    ext[c] = std::max(0.0, b - a);
    core[c] = std::min(ext[c], 4.0 * sd);
  }
  double h = cell_size;
  if (!(h > 0)) {
    bool active[3] = {true, true, true};
    h = 1.0;
    for (int iter = 0; iter < 4; iter++) {
      double vol = 1.0;
      int dims = 0;
      for (int c = 0; c < 3; c++) if (active[c]) { vol *= std::max(core[c], 1e-6); dims++; }
      if (dims == 0) { h = 1.0; break; }
      h = std::pow(vol * 2.0 / (double)std::max<int64_t>(n, 1), 1.0 / dims);
      bool changed = false;
      for (int c = 0; c < 3; c++) if (active[c] && core[c] < h) { active[c] = false; changed = true; }
      if (!changed) break;
    }
    if (!(h > 1e-6)) h = 1e-6;
  }
  for (;;) {
    int64_t cells = 1;
    bool axis_ok = true;
    for (int c = 0; c < 3; c++) {
      const double k = std::floor(ext[c] / h) + 1.0;
      // <= 65536 cells per axis: the fp32 cell coordinate (x - min) / h is then exact to ~0.01 cell,
      // which the query's stopping rule allows for (it trusts a ring only up to (r - 0.05) h)
      if (k > 65536.0) axis_ok = false;
      g.n[c] = (int)std::min(k, 65536.0);
      cells *= g.n[c];
    }
    if (axis_ok && cells <= kMaxCells) { g.n_cells = cells; break; }
    h *= 1.26;   // ~ a factor 2 fewer cells per try in 3-D
  }
  g.h = (float)h;
  g.inv_h = (float)(1.0 / h);
  for (int c = 0; c < 3; c++) g.min[c] = (float)rlo[c];
  g.filtered = 0;
  g.self_first = 0;
  g.self_count = 0;
  for (int c = 0; c < 3; c++) {   // what a filtered grid keeps of the other shards: the grid and a margin of
    // AFE_WORLD_MAX_RING + 1 cells around it (one cell more than any ring reaches, for the fp32 roundings)
    const double keep = (AFE_WORLD_MAX_RING + 1.0) * h;
    g.keep_lo[c] = (float)(rlo[c] - keep);
    g.keep_hi[c] = (float)(rlo[c] + (double)g.n[c] * h + keep);
  }
}

int ensure_capacity(afe_world *w, int64_t n_all, int64_t n_cells_total) {
  if (n_all > w->cap_points) {
    free_points(w);
    const int64_t cap = n_all + n_all / 8 + 1024;
    W_HIP(w, hipMalloc((void **)&w->cell, (size_t)cap * 4));
    W_HIP(w, hipMalloc((void **)&w->slot, (size_t)cap * 4));
    W_HIP(w, hipMalloc((void **)&w->sorted, (size_t)cap * sizeof(uint4)));
    W_HIP(w, hipMalloc((void **)&w->ref, (size_t)cap * 3 * sizeof(float)));
    W_HIP(w, hipMalloc((void **)&w->leftover, (size_t)cap * 4));
    W_HIP(w, hipMalloc((void **)&w->leftover_keys, (size_t)cap * 8));
    w->cap_points = cap;
  }
  if (n_cells_total > w->cap_cells) {
    if (w->counts) (void)hipFree(w->counts);
    if (w->block_sums) (void)hipFree(w->block_sums);
    w->counts = w->block_sums = nullptr;
    const int64_t cap = n_cells_total + 4096;
    W_HIP(w, hipMalloc((void **)&w->counts, (size_t)cap * 4));
    W_HIP(w, hipMalloc((void **)&w->block_sums, (size_t)((cap + 1023) / 1024 + 1) * 4));
    w->cap_cells = cap;
  }
  if (!w->lohi) W_HIP(w, hipMalloc((void **)&w->lohi, 16 * sizeof(int)));
  if (!w->words) { W_HIP(w, hipMalloc((void **)&w->words, 4 * sizeof(uint32_t))); W_HIP(w, hipMemset(w->words, 0, 4 * sizeof(uint32_t))); }
  if (!w->tickets) { W_HIP(w, hipMalloc((void **)&w->tickets, 3 * AFE_WORLD_TICKET_WORDS * sizeof(uint32_t))); W_HIP(w, hipMemset(w->tickets, 0, 3 * AFE_WORLD_TICKET_WORDS * sizeof(uint32_t))); }
  if (!w->host_flag) { W_HIP(w, hipHostMalloc((void **)&w->host_flag, 64, hipHostMallocCoherent | hipHostMallocMapped)); w->host_flag[0] = 0; w->host_flag[1] = 0; }
  if (!w->bounds_part) W_HIP(w, hipMalloc((void **)&w->bounds_part, AFE_WORLD_BOUNDS_BLOCKS * sizeof(BoundsPartial)));
  return AFE_OK;
}

}  // namespace

int afe::world_create(int device, afe_world **out) {
  if (!out) return AFE_ERR_INVALID_ARG;
  *out = nullptr;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return AFE_ERR_NO_DEVICE;
  if (device < 0 && hipGetDevice(&device) != hipSuccess) return AFE_ERR_NO_DEVICE;
  if (device >= n_dev) return AFE_ERR_NO_DEVICE;
  afe_world *w = new afe_world();
  w->device = device;
  *out = w;
  return AFE_OK;
}

void afe::world_destroy(afe_world *w) {
  if (!w) return;
  (void)hipSetDevice(w->device);
  free_points(w);
  if (w->counts) (void)hipFree(w->counts);
  if (w->block_sums) (void)hipFree(w->block_sums);
  if (w->lohi) (void)hipFree(w->lohi);
  if (w->bounds_part) (void)hipFree(w->bounds_part);
  if (w->self_scratch) (void)hipFree(w->self_scratch);
  if (w->words) (void)hipFree(w->words);
  if (w->tickets) (void)hipFree(w->tickets);
  if (w->host_flag) (void)hipHostFree(w->host_flag);
  delete w;
}

const char *afe::world_last_error(const afe_world *w) { return w ? w->err.c_str() : "null world"; }

int afe::world_nearest(afe_world *w, void *hip_stream, const float *all_xyz, int64_t n_all, int64_t first_global,
                                 int64_t n_self, float cell_size, float *dist2_out, int32_t *index_out) {
  if (!w || !all_xyz || n_all <= 0 || n_self <= 0 || first_global < 0 || first_global + n_self > n_all || !dist2_out || !index_out ||
      n_all > 0x7fffffff)
    return wfail(w, AFE_ERR_INVALID_ARG, "bad nearest-neighbour arguments");
  hipStream_t st = (hipStream_t)hip_stream;
  W_HIP(w, hipSetDevice(w->device));
  int rc = ensure_capacity(w, n_all, 0);
  if (rc) return rc;
  // 1. bounds and spread of the finite positions (one small read-back: the grid shape is a host decision).
  // The shape may be kept for a few queries: the query is exact for ANY grid (positions outside it clamp
  // into its boundary cells), a stale shape only costs speed, and without the read-back the whole query is
  // asynchronous on the stream (afe_set_neighbour_grid_refresh).
  // (the leftover counter is zeroed by the first kernel of the sort / of the regather)
  // A shard that queries for its own block of a larger gathered ensemble shapes the grid on ITS vehicles and sorts
  // only what can matter to them (see GridDesc::filtered): the cost of a query then follows the shard, not the
  // ensemble -- on 8 GPUs every rank would otherwise sort all 8 x 2^20 gathered positions for its 2^20 queries.
  const bool sharded = n_self < n_all;
  const bool reshape = w->grid_n_all != n_all || w->grid_first != first_global || w->grid_n_self != n_self ||
                       w->grid_cell_arg != cell_size || w->since_refresh + 1 >= w->refresh_every;
  if (reshape) {
    hipLaunchKernelGGL(world_bounds_kernel, dim3(AFE_WORLD_BOUNDS_BLOCKS), dim3(256), 0, st, all_xyz, n_all, sharded ? first_global : (int64_t)0,
                       sharded ? n_self : n_all, w->bounds_part);
    static thread_local BoundsPartial host_part[AFE_WORLD_BOUNDS_BLOCKS];
    W_HIP(w, hipMemcpyAsync(host_part, w->bounds_part, sizeof(host_part), hipMemcpyDeviceToHost, st));
    W_HIP(w, hipStreamSynchronize(st));
    int lohi[6] = {0x7fffffff, 0x7fffffff, 0x7fffffff, (int)0x80000000, (int)0x80000000, (int)0x80000000};
    double stats[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < AFE_WORLD_BOUNDS_BLOCKS; b++) {
      for (int c = 0; c < 3; c++) { lohi[c] = std::min(lohi[c], host_part[b].lo[c]); lohi[3 + c] = std::max(lohi[3 + c], host_part[b].hi[c]); }
      for (int c = 0; c < 7; c++) stats[c] += host_part[b].m[c];
    }
    float lo[3], hi[3];
    for (int c = 0; c < 3; c++) { lo[c] = unordered(lohi[c]); hi[c] = unordered(lohi[3 + c]); }
    if (lohi[0] == 0x7fffffff) { for (int c = 0; c < 3; c++) lo[c] = hi[c] = 0.0f; }   // no finite position at all
    choose_grid(lo, hi, stats, sharded ? n_self : n_all, cell_size, w->grid);
    w->grid.filtered = sharded ? 1 : 0;
    w->grid.self_first = first_global;
    w->grid.self_count = n_self;
    w->grid_n_all = n_all;
    w->grid_first = first_global;
    w->grid_n_self = n_self;
    w->grid_cell_arg = cell_size;
    w->since_refresh = 0;
  } else {
    w->since_refresh++;
  }
  const GridDesc g = w->grid;
  const int64_t m = g.n_cells + 2;   // cells, the dead bin, and the end sentinel
  const int64_t cap_before = w->cap_points;
  if ((rc = ensure_capacity(w, n_all, m))) return rc;
  const unsigned pb = (unsigned)((n_all + 255) / 256);
  if (*(volatile uint32_t *)w->host_flag) { w->sort_valid = false; *(volatile uint32_t *)w->host_flag = 0; }   // a regather met a vehicle the sort never listed
  const bool keep_order = !reshape && w->sort_valid && cap_before == w->cap_points && w->since_sort + 1 < w->resort_every;
  if (keep_order) {
    // 2'. the cell order of the last sort, today's positions (vehicles move centimetres between two queries, cells are
    // metres wide): one pass instead of the five of a sort
    w->since_sort++;
    // (the partial maxima live in the leftover-key scratch: the brute force behind the query is the next to touch it)
    float *partial = (float *)w->leftover_keys;
    hipLaunchKernelGGL(world_regather_kernel, dim3(pb), dim3(256), 0, st, all_xyz, n_all, w->cell, w->slot, w->ref, w->cap_points, w->sorted, partial,
                       (uint32_t *)(w->lohi + 6), w->words, w->host_flag, w->tickets);
  } else {
    // 2. counting sort by cell
    W_HIP(w, hipMemsetAsync(w->counts, 0, (size_t)m * 4, st));
    hipLaunchKernelGGL(world_count_kernel, dim3(pb), dim3(256), 0, st, all_xyz, n_all, g, w->counts, w->cell, w->slot, (uint32_t *)(w->lohi + 6), w->words);
    const int64_t nb = (m + 1023) / 1024;
    hipLaunchKernelGGL(world_scan_local_kernel, dim3((unsigned)nb), dim3(256), 0, st, w->counts, w->counts, w->block_sums, m);
    hipLaunchKernelGGL(world_scan_blocks_kernel, dim3(1), dim3(256), 0, st, w->block_sums, nb);
    hipLaunchKernelGGL(world_scan_add_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, w->counts, w->block_sums, m);
    hipLaunchKernelGGL(world_scatter_kernel, dim3(pb), dim3(256), 0, st, all_xyz, n_all, w->counts, w->cell, w->slot, w->sorted, w->ref, w->cap_points);
    w->sort_valid = true;
    w->since_sort = 0;
  }
  // 3. queries in cell order; isolated vehicles finish in the brute-force kernel
  uint32_t *left_count = (uint32_t *)(w->lohi + 6);
  // What the rings leave over is finished by the brute force: inside the query launch by its last workgroup while that is
  // little work (AFE_WORLD_TAIL_WORK distance evaluations: nothing or next to nothing left over in a small world), by the launch behind it otherwise,
  // which shares (query, chunk of the ensemble) items over the whole device and leaves at once when its count is zero.
  // No host-side guess is involved: a world that suddenly leaves 10^5 queries over (a fleet scattered beyond the grid's
  // rings) is answered in milliseconds by that launch, not in seconds by one compute unit.
  uint32_t *big_count = (uint32_t *)(w->lohi + 8);
  hipLaunchKernelGGL(world_query_kernel, dim3(pb), dim3(256), 0, st, w->sorted, n_all, w->counts, g, first_global, n_self, dist2_out,
                     index_out, left_count, w->leftover, w->leftover_keys, w->words, all_xyz, big_count, w->host_flag + 1, w->tickets);
  hipLaunchKernelGGL(world_brute_chunks_kernel, dim3(1024), dim3(256), 0, st, all_xyz, n_all, w->leftover, big_count, first_global,
                     (unsigned long long *)w->leftover_keys, w->tickets + 2 * AFE_WORLD_TICKET_WORDS, dist2_out, index_out);
  W_HIP(w, hipGetLastError());
  return AFE_OK;
}

int afe::world_set_refresh(afe_world *w, int every_n_queries) {
  if (!w || every_n_queries < 1) return AFE_ERR_INVALID_ARG;
  w->refresh_every = every_n_queries;
  return AFE_OK;
}

int afe::world_set_sort_reuse(afe_world *w, int every_n_queries) {
  if (!w || every_n_queries < 1) return AFE_ERR_INVALID_ARG;
  w->resort_every = every_n_queries;
  return AFE_OK;
}

int afe::world_grid_info(const afe_world *w, int dims[3], float *cell_size, int64_t *n_cells, int64_t *n_bruteforce) {
  if (!w) return AFE_ERR_INVALID_ARG;
  if (n_bruteforce) {   // how many queries of the last call the rings could not settle (synchronises)
    uint32_t left = 0;
    if (!w->lohi || hipSetDevice(w->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy(&left, w->lohi + 6, 4, hipMemcpyDeviceToHost) != hipSuccess)
      return AFE_ERR_HIP;
    *n_bruteforce = left;
  }
  if (dims) for (int c = 0; c < 3; c++) dims[c] = w->grid.n[c];
  if (cell_size) *cell_size = w->grid.h;
  if (n_cells) *n_cells = w->grid.n_cells;
  return AFE_OK;
}

// brute force over the whole ensemble for `n_queries` listed local vehicles (device array of local indices);
// the cross-check of the grid at sizes where an O(N^2) CPU loop is out of reach
int afe::world_nearest_bruteforce(afe_world *w, void *hip_stream, const float *all_xyz, int64_t n_all, int64_t first_global,
                                            const int32_t *dev_queries, int64_t n_queries, float *dist2_out, int32_t *index_out) {
  if (!w || !all_xyz || n_all <= 0 || !dev_queries || n_queries <= 0 || !dist2_out || !index_out)
    return wfail(w, AFE_ERR_INVALID_ARG, "bad brute-force arguments");
  hipStream_t st = (hipStream_t)hip_stream;
  W_HIP(w, hipSetDevice(w->device));
  int rc = ensure_capacity(w, n_queries, 0);
  if (rc) return rc;
  const uint32_t nq = (uint32_t)n_queries;
  uint32_t *cnt = (uint32_t *)(w->lohi + 7);
  W_HIP(w, hipMemcpyAsync(cnt, &nq, 4, hipMemcpyHostToDevice, st));
  W_HIP(w, hipStreamSynchronize(st));   // nq lives on this stack frame
  hipLaunchKernelGGL(world_brute_init_kernel, dim3(64), dim3(256), 0, st, w->leftover_keys, cnt);
  hipLaunchKernelGGL(world_brute_chunks_kernel, dim3(2048), dim3(256), 0, st, all_xyz, n_all, dev_queries, cnt, first_global,
                     (unsigned long long *)w->leftover_keys, (uint32_t *)nullptr, (float *)nullptr, (int32_t *)nullptr);
  hipLaunchKernelGGL(world_brute_finish_kernel, dim3(64), dim3(256), 0, st, dev_queries, cnt, w->leftover_keys, dist2_out, index_out);
  W_HIP(w, hipGetLastError());
  return AFE_OK;
}

// ---------------------------------------------------------------------------
// UWB ranging network (reference Components/Components/Simulation/UWBNetwork.{hpp,cpp})

struct afe_uwb_network {
  // the reference's generator and distributions, UWBNetwork.cpp:4-6 -- the same libstdc++ classes, so the
  // stream (engine words, generate_canonical, the polar method's cached second value) of ONE network, created
  // first in its process, is the reference's by construction.  The reference keeps them at FILE scope: a second
  // UWBNetwork there re-seeds the shared engine (rng.seed(0), :19) without clearing the normal distribution's
  // cached value, and networks running interleaved draw from one stream.  Here every network owns its stream
  // (what `rng.seed(0)` "to be repeatible" asks for); a host that has to reproduce a reference process with
  // several networks draws for all of them from one afe_uwb_network.
  std::mt19937 rng;
  std::uniform_real_distribution<double> dist_uniform{0, 1};
  std::normal_distribution<double> dist_normal{0, 1};
  double add_noise_std = 0, outlier_prob = 0, outlier_std = 0;   // UWBNetwork.cpp:15-17
  // device scratch
  int device = -1;
  int64_t cap = 0;
  int32_t *d_req = nullptr, *d_res = nullptr;
  double *d_noise = nullptr;
  uint8_t *d_out = nullptr;
  float *d_range = nullptr;
  std::vector<double> noise;
  std::vector<uint8_t> outlier;
};

extern "C" int afe_uwb_create(afe_uwb_network **out) {
  if (!out) return AFE_ERR_INVALID_ARG;
  afe_uwb_network *u = new afe_uwb_network();
  u->rng.seed(0);   // UWBNetwork.cpp:19 "to be repeatible"
  *out = u;
  return AFE_OK;
}
extern "C" void afe_uwb_destroy(afe_uwb_network *u) {
  if (!u) return;
  if (u->device >= 0) {
    (void)hipSetDevice(u->device);
    if (u->d_req) (void)hipFree(u->d_req);
    if (u->d_res) (void)hipFree(u->d_res);
    if (u->d_noise) (void)hipFree(u->d_noise);
    if (u->d_out) (void)hipFree(u->d_out);
    if (u->d_range) (void)hipFree(u->d_range);
  }
  delete u;
}
extern "C" int afe_uwb_set_noise(afe_uwb_network *u, double noise_std_dev, double outlier_probability, double outlier_std_dev) {
  if (!u) return AFE_ERR_INVALID_ARG;
  u->add_noise_std = noise_std_dev;       // UWBNetwork.hpp:28-33
  u->outlier_prob = outlier_probability;
  u->outlier_std = outlier_std_dev;
  return AFE_OK;
}

// the draws of n_pairs consecutive completed transactions, in stream order (UWBNetwork.cpp:66-71)
extern "C" int afe_uwb_draw(afe_uwb_network *u, int64_t n_pairs, double *noise_term, uint8_t *is_outlier) {
  if (!u || n_pairs < 0 || !noise_term || !is_outlier) return AFE_ERR_INVALID_ARG;
  for (int64_t k = 0; k < n_pairs; k++) {
    if (u->dist_uniform(u->rng) < u->outlier_prob) {
      is_outlier[k] = 1;
      noise_term[k] = u->dist_normal(u->rng) * u->outlier_std;
    } else {
      is_outlier[k] = 0;
      noise_term[k] = u->dist_normal(u->rng) * u->add_noise_std;
    }
  }
  return AFE_OK;
}

extern "C" int afe_uwb_range(afe_uwb_network *u, afe_engine *e, const float *dev_all_xyz, int64_t n_all,
                             const int32_t *requester, const int32_t *responder, int64_t n_pairs, float *range_out,
                             uint8_t *outlier_out) {
  if (!u || !e || !dev_all_xyz || n_all <= 0 || !requester || !responder || n_pairs <= 0 || !range_out) return AFE_ERR_INVALID_ARG;
  for (int64_t k = 0; k < n_pairs; k++)
    if (requester[k] < 0 || requester[k] >= n_all || responder[k] < 0 || responder[k] >= n_all) return AFE_ERR_OUT_OF_RANGE;
  void *stream_v = nullptr;
  int device = 0;
  engine_stream_device(e, &stream_v, &device);
  if (u->device >= 0 && u->device != device) return AFE_ERR_INVALID_ARG;
  if (hipSetDevice(device) != hipSuccess) return AFE_ERR_NO_DEVICE;
  u->device = device;
  if (n_pairs > u->cap) {
    if (u->d_req) (void)hipFree(u->d_req);
    if (u->d_res) (void)hipFree(u->d_res);
    if (u->d_noise) (void)hipFree(u->d_noise);
    if (u->d_out) (void)hipFree(u->d_out);
    if (u->d_range) (void)hipFree(u->d_range);
    u->d_req = u->d_res = nullptr; u->d_noise = nullptr; u->d_out = nullptr; u->d_range = nullptr; u->cap = 0;
    const size_t cap = (size_t)n_pairs + 1024;
    if (hipMalloc((void **)&u->d_req, cap * 4) != hipSuccess || hipMalloc((void **)&u->d_res, cap * 4) != hipSuccess ||
        hipMalloc((void **)&u->d_noise, cap * 8) != hipSuccess || hipMalloc((void **)&u->d_out, cap) != hipSuccess ||
        hipMalloc((void **)&u->d_range, cap * 4) != hipSuccess)
      return AFE_ERR_HIP;
    u->cap = (int64_t)cap;
  }
  u->noise.resize((size_t)n_pairs);
  u->outlier.resize((size_t)n_pairs);
  afe_uwb_draw(u, n_pairs, u->noise.data(), u->outlier.data());
  hipStream_t st = (hipStream_t)stream_v;
  if (hipMemcpyAsync(u->d_req, requester, (size_t)n_pairs * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(u->d_res, responder, (size_t)n_pairs * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(u->d_noise, u->noise.data(), (size_t)n_pairs * 8, hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(u->d_out, u->outlier.data(), (size_t)n_pairs, hipMemcpyHostToDevice, st) != hipSuccess)
    return AFE_ERR_HIP;
  hipLaunchKernelGGL(world_uwb_range_kernel, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, st, dev_all_xyz, n_all, u->d_req,
                     u->d_res, u->d_noise, u->d_out, n_pairs, u->d_range);
  if (hipMemcpyAsync(range_out, u->d_range, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)
    return AFE_ERR_HIP;
  if (outlier_out) std::memcpy(outlier_out, u->outlier.data(), (size_t)n_pairs);
  return AFE_OK;
}
