// search_api.hip -- one whole search per call: the per-move body of play_game (main.py:75-87) for n trees,
// with the HIP inference launch (mz_inference_forward) as the network.
//
//   initial_inference(image)            mz_inference_forward (Representation tower + heads)
//   expand_node(root, legal) + noise    mz_mcts_root_init
//   S x [ select / recurrent_inference / expand_node / backpropagate ]
//                                       mz_mcts_select, then S x mz_inference_forward + mz_mcts_backup_select
//   root statistics                     mz_mcts_root_stats
//
// Host code only: every kernel lives in mcts_kernels.hip / tower_kernels.hip.  The node pool and the staging rows
// between the tree kernels and the inference launch (network input, value, policy, hidden) are SCRATCH: mz_search
// takes them as a caller-owned workspace.
#include "common.h"
#include "search_internal.h"

static int64_t up256(int64_t x) { return (x + 255) / 256 * 256; }

int mz_plan_search(int32_t n_trees, int32_t n_sims, int64_t pool_off, int64_t pbc_off, SearchPlan *out) {
    mz_pool_layout L;
    if (int rc = mz_pool_layout_compute(n_trees, n_sims + 1 < 2 ? 2 : n_sims + 1, MZ_BF16, &L)) return rc;
    int64_t end = L.total_bytes;
    if (pbc_off >= 0) {   // the table lives elsewhere: the pool ends where its table would have started
        end = L.off_pbc;
        L.off_pbc = pbc_off - pool_off;
    }
    L.off_stats += pool_off; L.off_hidden += pool_off; L.off_meta += pool_off; L.off_path += pool_off; L.off_pbc += pool_off;
    int64_t off = up256(pool_off + end);
    out->off_net_in = off; off = up256(off + (int64_t)n_trees * 640 * 2);
    out->off_value = off;  off = up256(off + (int64_t)n_trees * 4);
    out->off_policy = off; off = up256(off + (int64_t)n_trees * MZ_ACTIONS * 4);
    out->off_hidden = off; off = up256(off + (int64_t)n_trees * MZ_HIDDEN_ELEMS * 2);
    L.total_bytes = off;
    out->L = L;
    out->total = off;
    return MZ_OK;
}

int mz_check_search_args(const char *who, const mz_tower_weights *repr, const mz_tower_weights *dyn, const mz_heads_weights *heads,
               const mz_search_cfg *cfg, const void *image, const int8_t *to_play, const uint64_t *legal, int32_t n_sims,
               int32_t n_trees, const int32_t *visits, const float *root_value) {
    MZ_REQUIRE(repr && dyn && heads && cfg, "%s: null weights/cfg", who);
    MZ_REQUIRE(n_trees >= 0 && n_sims >= 0 && n_sims <= 65532, "%s: n_trees %d / n_sims %d out of range", who, n_trees, n_sims);
    if (n_trees == 0) return MZ_OK;
    MZ_REQUIRE(image && to_play && legal && visits && root_value, "%s: null pointer", who);
    MZ_REQUIRE(repr->cin == 9 && dyn->cin == 10, "%s: repr/dyn towers must take 9 / 10 input planes", who);
    return MZ_OK;
}

int mz_run_search(const mz_tower_weights *repr, const mz_tower_weights *dyn, const mz_heads_weights *heads, const mz_search_cfg *cfg,
               const void *image, const int8_t *to_play, const uint64_t *legal, const double *noise, int32_t S, int32_t n,
               char *ws, const SearchPlan &P, int32_t *visits, float *root_value, void *stream) {
    // Routed: hidden states never leave the pool.  The inference launch reads the leaf's parent from its slot and writes the
    // new node's state into its slot; the descents only name the slots (mz_leaf_route, carved out of the net_in staging row
    // region, of which only the action planes are still used).
    float *value = (float *)(ws + P.off_value), *policy = (float *)(ws + P.off_policy);
    char *hpool = ws + P.L.off_hidden;
    const int64_t slot_bytes = up256((int64_t)n * 4);
    static_assert(2 * 256 + 256 <= 640 * 2, "the route fits the staging row region for any n >= 1");
    const mz_leaf_route rt{(int32_t *)(ws + P.off_net_in), (int32_t *)(ws + P.off_net_in + slot_bytes), ws + P.off_net_in + 2 * slot_bytes};
    if (int rc = mz_inference_forward_routed(repr, heads, image, nullptr, nullptr, hpool, nullptr, P.L.max_nodes, MZ_BF16, value, policy, n, stream))
        return rc;   // network.py:146-153, hidden state -> every tree's root slot
    if (int rc = mz_mcts_root_init(ws, &P.L, cfg, to_play, legal, nullptr, policy, noise, nullptr, n, stream)) return rc;
    if (S > 0) {
        if (int rc = mz_mcts_select_routed(ws, &P.L, cfg, &rt, n, stream)) return rc;
        for (int s = 0; s < S; ++s) {
            if (int rc = mz_inference_forward_routed(dyn, heads, rt.act_planes, hpool, rt.src_slot, hpool, rt.dst_slot, 1, MZ_BF16, value, policy, n, stream))
                return rc;   // network.py:155-163
            const int rc = s + 1 < S ? mz_mcts_backup_select_routed(ws, &P.L, cfg, value, policy, &rt, n, stream)
                                     : mz_mcts_expand_backup(ws, &P.L, cfg, value, policy, nullptr, n, stream);
            if (rc) return rc;
        }
    }
    return mz_mcts_root_stats(ws, &P.L, visits, root_value, nullptr, nullptr, n, stream);
}

extern "C" int64_t mz_search_workspace_bytes(int32_t n_trees, int32_t n_sims) {
    SearchPlan P;
    if (n_trees < 1 || n_sims < 0 || n_sims > 65532 || mz_plan_search(n_trees, n_sims, 0, -1, &P)) return -1;
    return P.total;
}

extern "C" int mz_search(const mz_tower_weights *repr, const mz_tower_weights *dyn, const mz_heads_weights *heads,
                         const mz_search_cfg *cfg, const void *image, const int8_t *to_play, const uint64_t *legal,
                         const double *noise, int32_t n_sims, int32_t n_trees, void *workspace, int64_t workspace_bytes,
                         int32_t *visits, float *root_value, void *stream) {
    if (int rc = mz_check_search_args("mz_search", repr, dyn, heads, cfg, image, to_play, legal, n_sims, n_trees, visits, root_value)) return rc;
    if (n_trees == 0) return MZ_OK;
    SearchPlan P;
    if (int rc = mz_plan_search(n_trees, n_sims, 0, -1, &P)) return rc;
    MZ_REQUIRE(workspace && ((uintptr_t)workspace & 255) == 0, "mz_search: workspace must be a 256-byte aligned device buffer");
    MZ_REQUIRE(workspace_bytes >= P.total, "mz_search: workspace of %lld bytes, %lld needed (mz_search_workspace_bytes)",
               (long long)workspace_bytes, (long long)P.total);
    if (int rc = mz_pool_init(workspace, &P.L, cfg, stream)) return rc;   // pb_c table (host libm) + cleared meta
    return mz_run_search(repr, dyn, heads, cfg, image, to_play, legal, noise, n_sims, n_trees, (char *)workspace, P, visits, root_value, stream);
}
