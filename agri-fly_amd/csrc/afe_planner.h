// afe_planner.h -- structures shared by the planner kernel and its host entry point.
#pragma once
#include <stdint.h>

#include "../../include/agrifly_engine.h"

namespace afe {

typedef afe_planner_config PlannerConfig;   // same layout as the public C struct
typedef afe_plan_output PlanOutput;

struct PlannerPyramid {   // RectangularPyramidPlanner::Pyramid (Pyramid.hpp:24-81)
  double depth;
  int right, top, left, bottom;
  double normal[4][3];
};

// the monotonic sections of one candidate (GetMonotonicSections, DepthImagePlanner.cpp:303-354): they depend on
// nothing but the candidate, so afe_rappids_candidates_kernel forms them for every admissible candidate, one lane
// each, and the search kernel's wave reads them instead of all 64 lanes repeating the quartic
struct CandSections {
  double t[5][2];                // [t0, t1] of each section, in the order IsCollisionFree pops them (last first)
  int32_t n;                     // <= 5 (a quartic has at most four roots inside (0, tf))
  uint32_t increasing;           // bit k: section k moves away from the camera
};

struct PlannerBatch {
  int64_t n;
  const uint16_t *images;        // [n_images][height][width]
  uint16_t *images_t;            // [n_images][width][height_t] scratch, filled by launch_rappids: the transposed images,
  int height_t;                  //   their columns padded to height_t = height rounded up to 64 pixels (128-byte runs start on 128-byte lines)
  int64_t n_images;
  // Per 64-pixel word of every image (only when width % 64 == 0, else null), filled by launch_rappids together with the
  // transpose: low half = the smallest depth above `ignore` (the vehicle's own radius in counts; 0xffff if none), high
  // half = the largest depth, or 0xffff if a pixel at or below `ignore` is among the 64.  Both bit images of a pyramid
  // ask "ignore < d < hi" of every pixel: a word whose smallest such depth is >= hi is all zeros, one whose high half is
  // < hi all ones -- only the words a depth edge at `hi` runs through are looked at pixel by pixel (build_mask_sum).
  uint32_t *sums;                // [n_images][height][width / 64]
  const int32_t *image_index;    // [n] or null (image i for planner i)
  const double *vel0, *acc0, *grav;  // planar [3][n], camera-fixed frame
  const double *cost_vec;        // planar [3][n] or null (cfg.cost_vec for all)
  const double *samples;         // [n_tables][n_candidates][4] = pixelX, pixelY, depth, time
  const int32_t *sample_table;   // [n] or null (table 0 for all)
  int n_candidates;
  double *cand_cost;             // [n][n_candidates] scratch: cost of every candidate
  uint8_t *cand_bits;            // [n][n_candidates] scratch: input-feasible / velocity-admissible bits
  CandSections *cand_sections;   // [n][n_candidates] scratch, filled for the admissible candidates
  PlannerPyramid *pyramids;      // [n][max_pyramids] scratch; up to 64 pyramids per plan: in the order they were made
  uint8_t *pyr_order;            // [n][64]: an interrupted planner's pyramid order by depth (slot numbers), see PyrKeys
  int max_pyramids;
  PlanOutput *out;               // [n]
  uint8_t *flags;                // [n][n_candidates] or null
  // Interruptible search (launch_rappids): in a round every unfinished planner works for at most `budget_ticks`
  // (100 MHz ticks), then writes down where it stands -- the sequential loop's own variables, nothing else -- and
  // leaves its slot; a later round picks it up.  Decisions, flags and counters are those of the uninterrupted loop by
  // construction.  (Built against the launch's tail -- a launch lasts as long as its longest planner, and planners
  // differ by two orders of magnitude --, measured slower than one launch, off by default: see launch_rappids.)
  struct Resume {
    int32_t done, base, lane;    // next candidate to look at = base + lane
    int32_t best_index, n_cost, n_feasible, n_velocity, n_free, n_pyr, pad;
    double best_cost;
  };
  Resume *resume;                // [n] or null (one uninterrupted launch)
  uint32_t budget_ticks;         // 0: no limit
  int32_t round;                 // 0: first round (nothing to resume)
  // Longest first (launch_rappids): a planner interrupted in the sizing round files itself under how many collision
  // checks it may still have to make; the finishing round starts the planners bin by bin, most work first.
  enum { kBins = 8 };
  int32_t *bin_count;            // [kBins] (zeroed before the sizing round) or null
  int32_t *bin_list;             // [kBins][n]
  int32_t ordered;               // this launch takes its planner from the bins: block j = the j-th planner in bin order
};

int launch_rappids(const PlannerConfig &cfg, const PlannerBatch &b, void *stream);

}  // namespace afe
