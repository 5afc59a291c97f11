// replay_kernels.hip -- device-resident replay window and training-batch builder for gfx950.
//
// Replaces utilities.py:46-109 (ReplayBuffer.save_game / sample_batch) and reversi.py:314-341
// (Game.make_target), with Game.make_image (reversi.py:301-312), Action.encode (:32-36) and
// store_search_statistics' normalisation (:292-299) folded in.  Byte/bit work, HBM-bound: one
// wavefront per game (save) or per sampled position (batch), every store coalesced.
#include "common.h"

namespace {

__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Action.encode rows (reversi.py:32-36; see env_kernels.hip)
__device__ __forceinline__ unsigned encode_rows(int a) { return a >= 64 ? 0xFFu : ((1u << (a >> 3)) | (1u << (a & 7))); }

// ---- save: step-major trajectories -> game-major window slots ------------------------------------
template <class V>
__global__ void __launch_bounds__(64) k_replay_save(mz_replay rb, int slot0, const uint64_t *hw, const uint64_t *hb,
                                                    const uint8_t *actions, const int8_t *rewards, const V *visits,
                                                    const float *root_values, const int32_t *length, int length_all, int64_t stride) {
    const int g = blockIdx.x, lane = threadIdx.x;
    const int T = rb.max_moves;
    const int64_t slot = (slot0 + g) % rb.window;
    int len = length ? length[g] : length_all;
    len = len < 0 ? 0 : (len > T ? T : len);   // a per-game length outside [0, max_moves] must not reach the sampler's indexing
    for (int t = lane; t <= T; t += 64) {
        const bool in = t <= len;  // state_history has len+1 entries
        rb.white[slot * (T + 1) + t] = in ? hw[(int64_t)t * stride + g] : 0ULL;
        rb.black[slot * (T + 1) + t] = in ? hb[(int64_t)t * stride + g] : 0ULL;
    }
    for (int t = lane; t < T; t += 64) {
        const bool in = t < len;
        rb.actions[slot * T + t] = in ? actions[(int64_t)t * stride + g] : (uint8_t)0;
        rb.rewards[slot * T + t] = in ? rewards[(int64_t)t * stride + g] : (int8_t)0;
        rb.root_values[slot * T + t] = in ? root_values[(int64_t)t * stride + g] : 0.0f;
    }
    // child_visits[t][a] = N_a / sum N (reversi.py:292-298), float64 then one rounding to fp32
    for (int t = 0; t < T; ++t) {
        float *dst = rb.policy + (slot * T + t) * MZ_ACTIONS;
        if (t >= len) {
            dst[lane] = 0.0f;
            if (lane == 0) dst[64] = 0.0f;
            continue;
        }
        const V *v = visits + ((int64_t)t * stride + g) * MZ_ACTIONS;
        const int n = (int)v[lane], n64 = (int)v[64];
        const int sum = wave_sum_i32(n > 0 ? n : 0) + (n64 > 0 ? n64 : 0);
        const double ds = (double)sum;
        dst[lane] = (n > 0 && sum > 0) ? (float)((double)n / ds) : 0.0f;
        if (lane == 0) dst[64] = (n64 > 0 && sum > 0) ? (float)((double)n64 / ds) : 0.0f;
    }
    if (lane == 0) rb.length[slot] = len;
}

// ---- sample_batch: one wave per (slot, position) ---------------------------------------------------
__global__ void __launch_bounds__(64) k_replay_sample(mz_replay rb, const int32_t *slot_in, const int32_t *pos_in, int batch, int K,
                                                      float *image, float *act, float *tvalue, float *tpolicy) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int T = rb.max_moves;
    const int64_t slot = slot_in[b];
    const int pos = pos_in[b];
    // utilities.py:104 draws pos in [0, len - K].  Anything else is a caller bug: never read out of bounds, never leave the
    // sample's rows uninitialised (the caller's buffers may come from an allocator that does not clear) -- zeros, and a count
    const int len = (slot < 0 || slot >= rb.window) ? -1 : rb.length[slot];
    if (len < 0 || pos < 0 || pos + K > len || len > T) {
        float *img = image + (int64_t)b * 576;
        for (int i = lane; i < 576; i += 64) img[i] = 0.0f;
        for (int k = 0; k < K; ++k) {
            float *dst = act + ((int64_t)k * batch + b) * 128;
            dst[lane] = 0.0f;
            dst[64 + lane] = 0.0f;
        }
        for (int k = 0; k <= K; ++k) {
            float *dst = tpolicy + ((int64_t)k * batch + b) * MZ_ACTIONS;
            dst[lane] = 0.0f;
            if (lane == 0) {
                dst[64] = 0.0f;
                tvalue[(int64_t)k * batch + b] = 0.0f;
            }
        }
        if (lane == 0 && rb.refused) atomicAdd(rb.refused, 1);
        return;
    }
    // Game.make_image(pos) reversi.py:301-312: plane p, cell = lane
    const uint64_t *hw = rb.white + slot * (T + 1), *hb = rb.black + slot * (T + 1);
    float *img = image + (int64_t)b * 576;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int i = 3 - (p >> 1);  // how many states back
        uint64_t bits = 0;
        if (i < pos && i < 4) bits = ((p & 1) ? hb : hw)[pos - i];
        img[p * 64 + lane] = (float)((bits >> lane) & 1ULL);
    }
    img[8 * 64 + lane] = (pos & 1) ? 0.0f : 1.0f;
    // actions[k][b] = Action(index, real mover).encode(): layer 0 WHITE else 1; mover of step t is BLACK for even t
    for (int k = 0; k < K; ++k) {
        const int t = pos + k;
        const int a = rb.actions[slot * T + t];
        const int my_layer = (t & 1) ? 0 : 1;
        const unsigned rows = encode_rows(a);
        float *dst = act + ((int64_t)k * batch + b) * 128;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int cell = lane;  // layer h, cell lane
            dst[h * 64 + cell] = (h == my_layer && ((rows >> (cell >> 3)) & 1u)) ? 1.0f : 0.0f;
        }
    }
    // Game.make_target reversi.py:319-339
    const int last = len > 0 ? (int)rb.rewards[slot * T + len - 1] : 0;
    for (int k = 0; k <= K; ++k) {
        const int ci = pos + k;
        const bool in = ci < len;
        float *dst = tpolicy + ((int64_t)k * batch + b) * MZ_ACTIONS;
        const float *src = rb.policy + (slot * T + (in ? ci : 0)) * MZ_ACTIONS;
        dst[lane] = in ? src[lane] : 0.0f;
        if (lane == 0) {
            dst[64] = in ? src[64] : 0.0f;
            const bool same = ((ci ^ (len - 1)) & 1) == 0;  // action_history[ci].player is action_history[-1].player
            tvalue[(int64_t)k * batch + b] = in ? (float)(same ? last : -last) : 0.0f;
        }
    }
}

}  // namespace

static int check_replay(const char *who, const mz_replay *rb) {
    MZ_REQUIRE(rb, "%s: null mz_replay", who);
    MZ_REQUIRE(rb->white && rb->black && rb->actions && rb->rewards && rb->policy && rb->root_values && rb->length,
               "%s: null array in mz_replay", who);
    MZ_REQUIRE(rb->window >= 1 && rb->max_moves >= 1 && rb->max_moves <= 255, "%s: bad window/max_moves", who);
    return MZ_OK;
}

extern "C" int mz_replay_save(const mz_replay *rb, int32_t slot0, const uint64_t *hist_white, const uint64_t *hist_black,
                              const uint8_t *actions, const int8_t *rewards, const void *visits, int32_t visits_dtype,
                              const float *root_values, const int32_t *length, int32_t length_all, int64_t stride, int32_t n, void *stream) {
    if (int rc = check_replay("mz_replay_save", rb)) return rc;
    MZ_REQUIRE(n >= 0 && n <= rb->window, "mz_replay_save: n %d outside [0, window %d]", n, rb->window);
    if (n == 0) return MZ_OK;
    MZ_REQUIRE(slot0 >= 0 && slot0 < rb->window, "mz_replay_save: slot0 out of range");
    MZ_REQUIRE(hist_white && hist_black && actions && rewards && visits && root_values, "mz_replay_save: null pointer");
    MZ_REQUIRE(stride >= n, "mz_replay_save: stride < n");
    MZ_REQUIRE(length || (length_all >= 0 && length_all <= rb->max_moves), "mz_replay_save: length_all out of range");
    MZ_REQUIRE(visits_dtype == MZ_VIS_I32 || visits_dtype == MZ_VIS_U16 || visits_dtype == MZ_VIS_U8, "mz_replay_save: bad visits_dtype %d", visits_dtype);
#define MZ_SAVE(V)                                                                                                                     \
    hipLaunchKernelGGL(k_replay_save<V>, dim3(n), dim3(64), 0, (hipStream_t)stream, *rb, (int)slot0, hist_white, hist_black, actions, rewards, \
                       (const V *)visits, root_values, length, (int)length_all, stride)
    if (visits_dtype == MZ_VIS_I32) MZ_SAVE(int32_t);
    else if (visits_dtype == MZ_VIS_U16) MZ_SAVE(uint16_t);
    else MZ_SAVE(uint8_t);
#undef MZ_SAVE
    MZ_LAUNCH_CHECK("k_replay_save");
    return MZ_OK;
}

extern "C" int mz_replay_sample_batch(const mz_replay *rb, const int32_t *slot, const int32_t *pos, int32_t batch, int32_t K,
                                      float *image, float *actions, float *target_values, float *target_policies, void *stream) {
    if (int rc = check_replay("mz_replay_sample_batch", rb)) return rc;
    MZ_REQUIRE(batch >= 0 && K >= 0 && K <= rb->max_moves, "mz_replay_sample_batch: bad batch/K");
    if (batch == 0) return MZ_OK;
    MZ_REQUIRE(slot && pos && image && target_values && target_policies && (actions || K == 0), "mz_replay_sample_batch: null pointer");
    hipLaunchKernelGGL(k_replay_sample, dim3(batch), dim3(64), 0, (hipStream_t)stream, *rb, slot, pos, (int)batch, (int)K, image, actions,
                       target_values, target_policies);
    MZ_LAUNCH_CHECK("k_replay_sample");
    return MZ_OK;
}
