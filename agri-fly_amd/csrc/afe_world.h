// afe_world.h -- shared-world query scratch owned by an engine (afe_world.hip).
#pragma once
#include <stdint.h>

struct afe_world;

namespace afe {

int world_create(int device, afe_world **out);
void world_destroy(afe_world *w);
const char *world_last_error(const afe_world *w);
// nearest neighbour of vehicles [first_global, first_global + n_self) among all_xyz (device, planar
// fp32 [3][n_all]) through a uniform grid; cell_size <= 0: chosen from the occupied box
int world_nearest(afe_world *w, void *hip_stream, const float *all_xyz, int64_t n_all, int64_t first_global,
                  int64_t n_self, float cell_size, float *dist2_out, int32_t *index_out);
int world_set_refresh(afe_world *w, int every_n_queries);
int world_set_sort_reuse(afe_world *w, int every_n_queries);
int world_grid_info(const afe_world *w, int dims[3], float *cell_size, int64_t *n_cells, int64_t *n_bruteforce);
int world_nearest_bruteforce(afe_world *w, void *hip_stream, const float *all_xyz, int64_t n_all, int64_t first_global,
                             const int32_t *dev_queries, int64_t n_queries, float *dist2_out, int32_t *index_out);

}  // namespace afe
