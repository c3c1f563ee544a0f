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
  uint16_t *images_t;            // [n_images][width][height] scratch, filled by launch_rappids
  int64_t n_images;
  const int32_t *image_index;    // [n] or null (image i for planner i)
  const double *vel0, *acc0, *grav;  // planar [3][n], camera-fixed frame
  const double *cost_vec;        // planar [3][n] or null (cfg.cost_vec for all)
  const double *samples;         // [n_tables][n_candidates][4] = pixelX, pixelY, depth, time
  const int32_t *sample_table;   // [n] or null (table 0 for all)
  int n_candidates;
  double *cand_cost;             // [n][n_candidates] scratch: cost of every candidate
  uint8_t *cand_bits;            // [n][n_candidates] scratch: input-feasible / velocity-admissible bits
  CandSections *cand_sections;   // [n][n_candidates] scratch, filled for the admissible candidates
  PlannerPyramid *pyramids;      // [n][max_pyramids] scratch
  int max_pyramids;
  PlanOutput *out;               // [n]
  uint8_t *flags;                // [n][n_candidates] or null
};

int launch_rappids(const PlannerConfig &cfg, const PlannerBatch &b, void *stream);

}  // namespace afe
