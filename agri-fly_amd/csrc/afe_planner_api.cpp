// afe_planner_api.cpp -- host side of the batched RAPPIDS planner (C ABI).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <mutex>
#include <random>
#include <vector>

#include "afe_planner.h"
#include "afe_host.h"   // afe_dev_env

using namespace afe;

extern "C" int afe_planner_default_config(afe_planner_config *c, int width, int height, double depth_scale,
                                          double focal_length, double true_vehicle_radius,
                                          double planning_vehicle_radius, double min_checking_dist) {
  if (!c || width <= 0 || height <= 0 || !(depth_scale > 0) || !(focal_length > 0) || !(min_checking_dist > 0))
    return AFE_ERR_INVALID_ARG;
  std::memset(c, 0, sizeof(*c));
  c->width = width; c->height = height;
  c->depth_scale = depth_scale; c->focal_length = focal_length;
  c->cx = width / 2.0; c->cy = height / 2.0;                         // main.cpp:484-488
  c->true_vehicle_radius = true_vehicle_radius;
  c->planning_vehicle_radius = planning_vehicle_radius;
  c->min_checking_dist = min_checking_dist;
  c->min_thrust = 5; c->max_thrust = 30; c->max_ang_vel = 20;       // DepthImagePlanner.cpp:43-50
  c->max_velocity = 5; c->min_section_time = 0.02;
  c->max_pyramids = 64;
  c->pixel_buffer = 2;                                               // DepthImagePlanner.cpp:59
  c->cost_type = 0;
  c->cost_vec[2] = 1.0;
  return AFE_OK;
}

extern "C" int afe_planner_samples(uint32_t seed, int width, int height, int n, double *samples4) {
  if (!samples4 || n < 0 || width <= 0 || height <= 0) return AFE_ERR_INVALID_ARG;
  // the reference's own generator types (DepthImagePlanner.hpp:349-366,417-425)
  std::uniform_real_distribution<> pixelX(0.1 * width, 0.9 * width), pixelY(0.1 * height, 0.9 * height);
  std::uniform_real_distribution<> depth(1.5, 3.0), time(2.0, 3.0);
  std::mt19937 gen(seed);
  for (int k = 0; k < n; k++) {
    // DeprojectPixelToPoint(_pixelX(_gen), _pixelY(_gen), _depth(_gen), posf): GCC evaluates
    // the arguments right to left (DepthImagePlanner.hpp:395-396); spelled out so that the
    // order does not depend on who compiles this file
    const double d = depth(gen);
    const double y = pixelY(gen);
    const double x = pixelX(gen);
    const double t = time(gen);
    samples4[4 * k + 0] = x; samples4[4 * k + 1] = y; samples4[4 * k + 2] = d; samples4[4 * k + 3] = t;
  }
  return AFE_OK;
}

namespace {
// Device scratch of a plan call.  A call needs fifteen buffers (for config 3: 0.5 GB of pyramids, 1.5 GB of
// candidate sections, ...); allocating and freeing them every call costs milliseconds at 30 plans a second, so the
// library keeps them between calls -- one set per process, grown on demand, on the device of the last call -- and
// plan calls take turns on it (they would take turns on the GPU anyway).  afe_planner_release_scratch() gives the
// memory back.
struct ScratchSlot {
  void *p = nullptr;
  size_t cap = 0;
};
std::mutex g_scratch_mutex;
ScratchSlot g_scratch[20];
int g_scratch_device = -1;

void release_scratch_locked() {
  if (g_scratch_device >= 0) (void)hipSetDevice(g_scratch_device);
  for (ScratchSlot &s : g_scratch) {
    if (s.p) (void)hipFree(s.p);
    s.p = nullptr;
    s.cap = 0;
  }
  g_scratch_device = -1;
}

struct DevBuf {
  void *p = nullptr;
  int slot;
  explicit DevBuf(int slot_) : slot(slot_) {}
  bool alloc(size_t bytes) {
    ScratchSlot &s = g_scratch[slot];
    if (bytes == 0) bytes = 1;
    if (s.cap < bytes) {
      if (s.p) (void)hipFree(s.p);
      s.p = nullptr;
      s.cap = 0;
      const size_t want = bytes + bytes / 8;
      if (hipMalloc(&s.p, want) != hipSuccess) {
        (void)hipGetLastError();
        if (hipMalloc(&s.p, bytes) != hipSuccess) { s.p = nullptr; return false; }
        s.cap = bytes;
      } else {
        s.cap = want;
      }
    }
    p = s.p;
    return true;
  }
  bool upload(const void *src, size_t bytes) { return alloc(bytes) && hipMemcpy(p, src, bytes, hipMemcpyHostToDevice) == hipSuccess; }
};
}  // namespace

extern "C" int afe_planner_release_scratch(void) {
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  release_scratch_locked();
  return AFE_OK;
}

static int plan_impl(int device, const afe_planner_config *cfg, int64_t n, const uint16_t *depth_images,
                     bool depth_on_device, int64_t n_images, const int32_t *image_index, const double *vel0,
                     const double *acc0, const double *grav, const double *cost_vec, const double *samples,
                     int n_tables, const int32_t *sample_table, int n_candidates, afe_plan_output *out,
                     uint8_t *flags, float *kernel_ms) {
  if (!cfg || n < 0 || !depth_images || n_images <= 0 || !vel0 || !acc0 || !grav || !samples || n_tables <= 0 ||
      n_candidates <= 0 || !out || cfg->max_pyramids <= 0 || cfg->width <= 0 || cfg->height <= 0)
    return AFE_ERR_INVALID_ARG;
  if (!image_index && n_images < n) return AFE_ERR_INVALID_ARG;
  if (n == 0) return AFE_OK;             // (no planners: an empty request is answered)
  // the search kernel keeps one bit per pixel in LDS (64 KB of dynamic LDS at most) and indexes
  // pixels with 24-bit arithmetic; device-resident images must allow 16-byte loads
  if ((int64_t)((cfg->width + 63) / 64) * cfg->height * 8 > 65536 || (int64_t)cfg->width * cfg->height >= (1 << 24))
    return AFE_ERR_OUT_OF_RANGE;
  if (depth_on_device && ((uintptr_t)depth_images & 15u)) return AFE_ERR_INVALID_ARG;
  if (cfg->focal_length * cfg->planning_vehicle_radius / cfg->depth_scale >= (double)(1 << 22)) return AFE_ERR_OUT_OF_RANGE;
  if (image_index)
    for (int64_t i = 0; i < n; i++) if (image_index[i] < 0 || image_index[i] >= n_images) return AFE_ERR_OUT_OF_RANGE;
  if (sample_table)
    for (int64_t i = 0; i < n; i++) if (sample_table[i] < 0 || sample_table[i] >= n_tables) return AFE_ERR_OUT_OF_RANGE;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return AFE_ERR_NO_DEVICE;
  if (device < 0 && hipGetDevice(&device) != hipSuccess) return AFE_ERR_NO_DEVICE;
  if (device >= n_dev) return AFE_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess || std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return AFE_ERR_NO_DEVICE;
  if (hipSetDevice(device) != hipSuccess) return AFE_ERR_HIP;

  std::lock_guard<std::mutex> scratch_lock(g_scratch_mutex);
  if (g_scratch_device != device) { release_scratch_locked(); (void)hipSetDevice(device); g_scratch_device = device; }
  const size_t px = (size_t)cfg->width * cfg->height;
  DevBuf d_img(0), d_imgT(1), d_cc(2), d_cb(3), d_cs(4), d_idx(5), d_v(6), d_a(7), d_g(8), d_c(9), d_s(10), d_t(11), d_pyr(12), d_out(13),
      d_flags(14), d_resume(15), d_bins(16), d_sums(17), d_order(18);
  if ((!depth_on_device && !d_img.upload(depth_images, (size_t)n_images * px * 2)) || !d_v.upload(vel0, (size_t)n * 24) ||
      !d_a.upload(acc0, (size_t)n * 24) || !d_g.upload(grav, (size_t)n * 24) ||
      !d_s.upload(samples, (size_t)n_tables * n_candidates * 32))
    return AFE_ERR_HIP;
  if (image_index && !d_idx.upload(image_index, (size_t)n * 4)) return AFE_ERR_HIP;
  if (cost_vec && !d_c.upload(cost_vec, (size_t)n * 24)) return AFE_ERR_HIP;
  if (sample_table && !d_t.upload(sample_table, (size_t)n * 4)) return AFE_ERR_HIP;
  if (!d_pyr.alloc((size_t)n * cfg->max_pyramids * sizeof(PlannerPyramid)) || !d_out.alloc((size_t)n * sizeof(PlanOutput)) ||
      !d_imgT.alloc((size_t)n_images * cfg->width * (size_t)((cfg->height + 63) / 64 * 64) * 2) || !d_cc.alloc((size_t)n * n_candidates * 8) ||
      !d_cb.alloc((size_t)n * n_candidates) || !d_cs.alloc((size_t)n * n_candidates * sizeof(CandSections)))
    return AFE_ERR_HIP;
  if (flags && !d_flags.alloc((size_t)n * n_candidates)) return AFE_ERR_HIP;
  if (!d_resume.alloc((size_t)n * sizeof(PlannerBatch::Resume))) return AFE_ERR_HIP;
  if (!d_bins.alloc(((size_t)n * PlannerBatch::kBins + 64) * sizeof(int32_t))) return AFE_ERR_HIP;
  // the per-word summaries need rows of whole 64-pixel words (and 512 B of LDS behind the bit image for their list)
  const bool whole_words = (cfg->width & 63) == 0 && (px >> 6) * 8 + 512 <= 65536 && !afe_dev_env("AFE_PLANNER_NO_SUMS");      // (lab variable: the plain sweep everywhere)
  if (whole_words && !d_sums.alloc((size_t)n_images * (px >> 6) * sizeof(uint32_t))) return AFE_ERR_HIP;
  if (cfg->max_pyramids <= 64 && !d_order.alloc((size_t)n * 64)) return AFE_ERR_HIP;

  PlannerBatch b;
  b.n = n;
  b.images = depth_on_device ? depth_images : (const uint16_t *)d_img.p;
  b.images_t = (uint16_t *)d_imgT.p;
  b.height_t = (cfg->height + 63) / 64 * 64;
  b.sums = whole_words ? (uint32_t *)d_sums.p : nullptr;
  b.pyr_order = cfg->max_pyramids <= 64 && !afe_dev_env("AFE_PLANNER_NO_LANE_LIST") ? (uint8_t *)d_order.p : nullptr;      // (lab variable: the pyramid list in HBM)
  b.cand_cost = (double *)d_cc.p;
  b.cand_bits = (uint8_t *)d_cb.p;
  b.cand_sections = (CandSections *)d_cs.p;
  b.n_images = n_images;
  b.image_index = image_index ? (const int32_t *)d_idx.p : nullptr;
  b.vel0 = (const double *)d_v.p; b.acc0 = (const double *)d_a.p; b.grav = (const double *)d_g.p;
  b.cost_vec = cost_vec ? (const double *)d_c.p : nullptr;
  b.samples = (const double *)d_s.p;
  b.sample_table = sample_table ? (const int32_t *)d_t.p : nullptr;
  b.n_candidates = n_candidates;
  b.pyramids = (PlannerPyramid *)d_pyr.p;
  b.max_pyramids = cfg->max_pyramids;
  b.out = (PlanOutput *)d_out.p;
  b.flags = flags ? (uint8_t *)d_flags.p : nullptr;
  b.resume = (PlannerBatch::Resume *)d_resume.p;
  b.budget_ticks = 0;
  b.round = 0;
  b.bin_count = (int32_t *)d_bins.p;
  b.bin_list = (int32_t *)d_bins.p + 64;
  b.ordered = 0;

  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return AFE_ERR_HIP;
  (void)hipEventRecord(e0, 0);
  const int lrc = launch_rappids(*cfg, b, nullptr);
  (void)hipEventRecord(e1, 0);
  int rc = AFE_OK;
  if (lrc != 0 || hipEventSynchronize(e1) != hipSuccess) rc = AFE_ERR_HIP;
  float ms = 0;
  if (rc == AFE_OK) (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (rc != AFE_OK) return rc;
  if (kernel_ms) *kernel_ms = ms;
  if (hipMemcpy(out, d_out.p, (size_t)n * sizeof(PlanOutput), hipMemcpyDeviceToHost) != hipSuccess) return AFE_ERR_HIP;
  if (flags && hipMemcpy(flags, d_flags.p, (size_t)n * n_candidates, hipMemcpyDeviceToHost) != hipSuccess) return AFE_ERR_HIP;
  return AFE_OK;
}

extern "C" int afe_rappids_plan(int device, const afe_planner_config *cfg, int64_t n, const uint16_t *depth_images,
                                int64_t n_images, const int32_t *image_index, const double *vel0,
                                const double *acc0, const double *grav, const double *cost_vec,
                                const double *samples, int n_tables, const int32_t *sample_table,
                                int n_candidates, afe_plan_output *out, uint8_t *flags, float *kernel_ms) {
  return plan_impl(device, cfg, n, depth_images, false, n_images, image_index, vel0, acc0, grav, cost_vec, samples,
                   n_tables, sample_table, n_candidates, out, flags, kernel_ms);
}

extern "C" int afe_rappids_plan_device(int device, const afe_planner_config *cfg, int64_t n,
                                       const void *dev_depth_images, int64_t n_images, const int32_t *image_index,
                                       const double *vel0, const double *acc0, const double *grav,
                                       const double *cost_vec, const double *samples, int n_tables,
                                       const int32_t *sample_table, int n_candidates, afe_plan_output *out,
                                       uint8_t *flags, float *kernel_ms) {
  return plan_impl(device, cfg, n, (const uint16_t *)dev_depth_images, true, n_images, image_index, vel0, acc0, grav,
                   cost_vec, samples, n_tables, sample_table, n_candidates, out, flags, kernel_ms);
}
