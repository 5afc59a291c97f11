// search_internal.h -- pieces of search_api.hip shared inside the library (not part of the C ABI).
#pragma once
#include "common.h"

#define MZ_INTERNAL __attribute__((visibility("hidden")))

struct SearchPlan {
    mz_pool_layout L;   // offsets relative to the workspace base
    int64_t off_net_in, off_value, off_policy, off_hidden, total;
};

// workspace = [node pool as mz_pool_layout_compute lays it out, shifted by `pool_off`] [net_in n x 640 bf16] [value n f32]
//             [policy n x 65 f32] [hidden n x 512 bf16]; pbc_off < 0 keeps the pool's own pb_c table, else the table sits there
MZ_INTERNAL int mz_plan_search(int32_t n_trees, int32_t n_sims, int64_t pool_off, int64_t pbc_off, SearchPlan *out);
MZ_INTERNAL int mz_check_search_args(const char *who, const mz_tower_weights *repr, const mz_tower_weights *dyn, const mz_heads_weights *heads,
                                     const mz_search_cfg *cfg, const void *image, const int8_t *to_play, const uint64_t *legal,
                                     int32_t n_sims, int32_t n_trees, const int32_t *visits, const float *root_value);
// the launches of one search; `ws` is the scratch base (device address), P its plan, the pb_c table is already in place
MZ_INTERNAL int mz_run_search(const mz_tower_weights *repr, const mz_tower_weights *dyn, const mz_heads_weights *heads,
                              const mz_search_cfg *cfg, const void *image, const int8_t *to_play, const uint64_t *legal,
                              const double *noise, int32_t S, int32_t n, char *ws, const SearchPlan &P, int32_t *visits,
                              float *root_value, void *stream);
