// env_kernels.hip -- batched 8x8 reversi on 64-bit bitboards for gfx950 (one lane per game).
//
// Replaces the per-move Python of reversi.py:81-205 (Environment) and reversi.py:32-36,182-186,
// 301-312 (Action.encode / observe / Game.make_image).  Bit i of a board <=> Action.index i =
// row*8+col.  Move generation and flipping are Kogge-Stone style directional fills: a run of
// opponent discs adjacent to the mover's disc (movegen) or to the placed disc (flip) is grown by
// shift-and-mask in log steps (1,1,2,2 -> up to 6 discs, the longest run an 8-wide board allows).
// Everything is 32-bit integer VALU work on registers; HBM traffic is the SoA state itself.
#include "common.h"

namespace {

// Instruction selection is driven by measured gfx950 issue costs (tools/valu_rate.hip, r01):
//   v_and/v_or_b32, v_bitop3_b32            ~2.4 cycles per wave64 instruction (full rate)
//   v_lsh{l,r}rev_b32, v_lsh{l,r}rev_b64,
//   v_alignbit_b32, v_and_or_b32            ~4.2 cycles (half rate)
// so a 64-bit shift must stay ONE v_lsh*rev_b64 (hipcc otherwise splits it into a 32-bit shift plus
// v_alignbit, twice the cost: hence the asm wrappers) and each mask-and-merge of a fill step is one
// v_bitop3_b32 per 32-bit half instead of v_and + v_or or the half-rate v_and_or.
template <int D> __device__ __forceinline__ uint64_t shl(uint64_t x) {
    uint64_t r;
    asm("v_lshlrev_b64 %0, %1, %2" : "=v"(r) : "i"(D), "v"(x));
    return r;
}
template <int D> __device__ __forceinline__ uint64_t shr(uint64_t x) {
    uint64_t r;
    asm("v_lshrrev_b64 %0, %1, %2" : "=v"(r) : "i"(D), "v"(x));
    return r;
}
// (a & b) | c  -- truth table 0xEA for inputs a=0xF0, b=0xCC, c=0xAA
__device__ __forceinline__ uint64_t and_or(uint64_t a, uint64_t b, uint64_t c) {
    const uint32_t lo = __builtin_amdgcn_bitop3_b32((uint32_t)a, (uint32_t)b, (uint32_t)c, 0xEA);
    const uint32_t hi = __builtin_amdgcn_bitop3_b32((uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(c >> 32), 0xEA);
    return ((uint64_t)hi << 32) | lo;
}
// a | b | c  -- truth table 0xFE
__device__ __forceinline__ uint64_t or3(uint64_t a, uint64_t b, uint64_t c) {
    const uint32_t lo = __builtin_amdgcn_bitop3_b32((uint32_t)a, (uint32_t)b, (uint32_t)c, 0xFE);
    const uint32_t hi = __builtin_amdgcn_bitop3_b32((uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(c >> 32), 0xFE);
    return ((uint64_t)hi << 32) | lo;
}

constexpr uint64_t INNER = 0x7E7E7E7E7E7E7E7EULL;  // columns 1..6: horizontal/diagonal runs never contain an edge file

// mover discs P, opponent run mask M, direction +D (<<) and -D (>>): squares that end a run
template <int D>
__device__ __forceinline__ uint64_t moves_dir(uint64_t P, uint64_t M) {
    uint64_t fl = M & shl<D>(P), fr = M & shr<D>(P);
    fl = and_or(M, shl<D>(fl), fl);
    fr = and_or(M, shr<D>(fr), fr);
    const uint64_t ml = M & shl<D>(M), mr = shr<D>(ml);
    fl = and_or(ml, shl<2 * D>(fl), fl);
    fr = and_or(mr, shr<2 * D>(fr), fr);
    fl = and_or(ml, shl<2 * D>(fl), fl);
    fr = and_or(mr, shr<2 * D>(fr), fr);
    return shl<D>(fl) | shr<D>(fr);
}

// reversi.py:81-126: legal placements of the side owning `own`
__device__ __forceinline__ uint64_t legal_moves(uint64_t own, uint64_t opp) {
    const uint64_t mo = opp & INNER;
    const uint64_t mv = or3(moves_dir<1>(own, mo), moves_dir<7>(own, mo), moves_dir<9>(own, mo)) | moves_dir<8>(own, opp);
    return mv & ~(own | opp);
}

// discs flipped along +-D by placing the single bit m (reversi.py:149-162)
template <int D>
__device__ __forceinline__ uint64_t flips_dir(uint64_t m, uint64_t P, uint64_t M) {
    uint64_t fl = M & shl<D>(m), fr = M & shr<D>(m);
    fl = and_or(M, shl<D>(fl), fl);
    fr = and_or(M, shr<D>(fr), fr);
    const uint64_t ml = M & shl<D>(M), mr = shr<D>(ml);
    fl = and_or(ml, shl<2 * D>(fl), fl);
    fr = and_or(mr, shr<2 * D>(fr), fr);
    fl = and_or(ml, shl<2 * D>(fl), fl);
    fr = and_or(mr, shr<2 * D>(fr), fr);
    // the run is captured iff the square just past its end holds a mover disc
    uint64_t r = (P & shl<D>(fl)) ? fl : 0ULL;
    r |= (P & shr<D>(fr)) ? fr : 0ULL;
    return r;
}

__device__ __forceinline__ uint64_t flips_for(uint64_t m, uint64_t own, uint64_t opp) {
    const uint64_t mo = opp & INNER;
    return or3(flips_dir<1>(m, own, mo), flips_dir<7>(m, own, mo), flips_dir<9>(m, own, mo)) | flips_dir<8>(m, own, opp);
}

struct EnvState {
    uint64_t black, white, legal;
    int to_play, steps, done;
};

// Environment.step (reversi.py:130-180).  Returns status; reward via reference.
__device__ __forceinline__ int env_step(EnvState &s, int a, int &reward, bool &wrote_turn) {
    reward = 0;
    wrote_turn = false;
    if (s.done) return MZ_ST_DONE;
    if (a > 64) return MZ_ST_ILLEGAL;  // reversi.py:26
    const bool white_moves = s.to_play == 1;
    uint64_t own = white_moves ? s.white : s.black;
    uint64_t opp = white_moves ? s.black : s.white;
    if (a != 64) {
        const uint64_t m = 1ULL << a;
        const uint64_t fl = (m & (own | opp)) ? 0ULL : flips_for(m, own, opp);
        if (fl == 0) return MZ_ST_ILLEGAL;  // reversi.py:143-144,164-165
        own |= m | fl;
        opp &= ~fl;
        s.white = white_moves ? own : opp;
        s.black = white_moves ? opp : own;
    }
    s.steps += 1;                      // reversi.py:133 (passes count)
    if (a != 64 && s.steps == 60) {    // reversi.py:170-172: player not switched, legal not refreshed
        s.done = 1;
        const int d = __popcll(own) - __popcll(opp);  // check_win :197-205 from the mover's side
        reward = (d > 0) - (d < 0);
        return MZ_ST_OK;
    }
    s.to_play = -s.to_play;            // reversi.py:137 / :176
    s.legal = legal_moves(opp, own);   // update_legal_actions for the new side to move
    wrote_turn = true;
    return MZ_ST_OK;
}

// build-defined playout RNG (python statement in tests/refspec.py)
__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t playout_rand(uint64_t seed, uint64_t game, uint64_t step) {
    return splitmix64(seed + game * 0xD1342543DE82EF95ULL + step * 0x9E3779B97F4A7C15ULL);
}
// index of the k-th (0-based, ascending) set bit; branch-free popcount descent
__device__ __forceinline__ int kth_set_bit(uint64_t x, unsigned k) {
    int pos = 0;
    unsigned c = __popc((unsigned)x);
    if (k >= c) { k -= c; x >>= 32; pos = 32; }
    unsigned w = (unsigned)x;
    c = __popc(w & 0xFFFFu);
    if (k >= c) { k -= c; w >>= 16; pos += 16; }
    c = __popc(w & 0xFFu);
    if (k >= c) { k -= c; w >>= 8; pos += 8; }
    c = __popc(w & 0xFu);
    if (k >= c) { k -= c; w >>= 4; pos += 4; }
    c = __popc(w & 0x3u);
    if (k >= c) { k -= c; w >>= 2; pos += 2; }
    if (k >= (w & 1u)) pos += 1;
    return pos;
}
__device__ __forceinline__ int playout_action(uint64_t seed, uint64_t game, uint64_t step, uint64_t legal) {
    const unsigned cnt = __popcll(legal);
    if (cnt == 0) return 64;
    const uint64_t r = playout_rand(seed, game, step);
    const unsigned k = (unsigned)(((r >> 32) * (uint64_t)cnt) >> 32);
    return kth_set_bit(legal, k);
}
__device__ __forceinline__ uint64_t digest_step(uint64_t h, const EnvState &s, int reward) {
    h = splitmix64(h ^ s.white);
    h = splitmix64(h ^ s.black);
    h = splitmix64(h ^ s.legal);
    const uint64_t meta = (uint64_t)(s.to_play & 0xFF) | ((uint64_t)(s.steps & 0xFF) << 8) | ((uint64_t)(s.done & 1) << 16) |
                          ((uint64_t)(reward & 0xFF) << 24) | ((uint64_t)(s.legal == 0) << 32);
    return splitmix64(h ^ meta);
}

constexpr uint64_t OPEN_WHITE = (1ULL << 27) | (1ULL << 36);  // reversi.py:43-44
constexpr uint64_t OPEN_BLACK = (1ULL << 28) | (1ULL << 35);  // reversi.py:45-46
constexpr uint64_t OPEN_LEGAL = (1ULL << 19) | (1ULL << 26) | (1ULL << 44) | (1ULL << 37);  // reversi.py:52

__global__ void __launch_bounds__(256) k_env_reset(uint64_t *black, uint64_t *white, int8_t *to_play, uint8_t *steps,
                                                   uint8_t *done, uint64_t *legal, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    black[i] = OPEN_BLACK;
    white[i] = OPEN_WHITE;
    to_play[i] = -1;
    steps[i] = 0;
    done[i] = 0;
    legal[i] = OPEN_LEGAL;
}

__global__ void __launch_bounds__(256) k_env_legal_mask(const uint64_t *black, const uint64_t *white, const int8_t *to_play,
                                                        uint64_t *legal, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t b = black[i], w = white[i];
    const bool white_moves = to_play[i] == 1;
    legal[i] = legal_moves(white_moves ? w : b, white_moves ? b : w);
}

__global__ void __launch_bounds__(256) k_env_step(uint64_t *black, uint64_t *white, int8_t *to_play, uint8_t *steps,
                                                  uint8_t *done, uint64_t *legal, const uint8_t *action, int8_t *reward,
                                                  uint8_t *status, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    EnvState s;
    s.black = black[i];
    s.white = white[i];
    s.to_play = to_play[i];
    s.steps = steps[i];
    s.done = done[i];
    s.legal = 0;
    int rw;
    bool turn;
    const int st = env_step(s, action[i], rw, turn);
    reward[i] = (int8_t)rw;
    status[i] = (uint8_t)st;
    if (st == MZ_ST_OK) {
        black[i] = s.black;
        white[i] = s.white;
        steps[i] = (uint8_t)s.steps;
        done[i] = (uint8_t)s.done;
        if (turn) {
            to_play[i] = (int8_t)s.to_play;
            legal[i] = s.legal;
        }
    }
}

// the same step on a PACKED per-game record: the six byte-wide streams of k_env_step (to_play, steps, done, action, reward,
// status: 64-byte requests per wave) become one u32 word (256-byte requests), next to the three u64 words.
//   meta: bits 0-6 steps | bit 7 WHITE to move | bit 8 done | bits 9-15 the action to play (in) | bits 16-17 reward + 1 (out) |
//         bits 18-19 status (out)
__global__ void __launch_bounds__(256) k_env_step_packed(uint64_t *black, uint64_t *white, uint64_t *legal, uint32_t *meta, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t m = meta[i];
    EnvState s;
    s.black = black[i];
    s.white = white[i];
    s.steps = (int)(m & 127u);
    s.to_play = (m >> 7) & 1u ? 1 : -1;
    s.done = (int)((m >> 8) & 1u);
    s.legal = 0;
    int rw;
    bool turn;
    const int st = env_step(s, (int)((m >> 9) & 127u), rw, turn);
    uint32_t out = m & 0xFFFFu;                                  // a refused step leaves the state as it was
    if (st == MZ_ST_OK) {
        black[i] = s.black;
        white[i] = s.white;
        if (turn) legal[i] = s.legal;
        out = (uint32_t)s.steps | (s.to_play == 1 ? 128u : 0u) | ((uint32_t)s.done << 8) | (m & 0xFE00u);
    }
    meta[i] = out | ((uint32_t)(rw + 1) << 16) | ((uint32_t)st << 18);
}

template <bool DIGEST, bool LOG>
__global__ void __launch_bounds__(256) k_env_random_playout(uint64_t seed, uint64_t game0, int64_t n, int n_steps, int voluntary,
                                                            uint64_t *black, uint64_t *white, int8_t *to_play, uint8_t *steps,
                                                            uint8_t *done, uint64_t *legal, uint64_t *digest, uint8_t *actions,
                                                            int8_t *last_reward) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    EnvState s;
    s.black = OPEN_BLACK;
    s.white = OPEN_WHITE;
    s.legal = OPEN_LEGAL;
    s.to_play = -1;
    s.steps = 0;
    s.done = 0;
    const uint64_t game = game0 + (uint64_t)i;
    uint64_t h = 0;
    if (DIGEST) h = digest_step(0, s, 0);
    int rw = 0;
    for (int t = 0; t < n_steps; ++t) {
        int a = playout_action(seed, game, (uint64_t)t, s.legal);
        if (voluntary && (playout_rand(seed ^ 0x5555ULL, game, (uint64_t)t) & 7) == 0) a = 64;
        bool turn;
        env_step(s, a, rw, turn);
        if (DIGEST) h = digest_step(h, s, rw);
        if (LOG) actions[i * n_steps + t] = (uint8_t)a;
    }
    black[i] = s.black;
    white[i] = s.white;
    to_play[i] = (int8_t)s.to_play;
    steps[i] = (uint8_t)s.steps;
    done[i] = (uint8_t)s.done;
    legal[i] = s.legal;
    if (DIGEST) digest[i] = h;
    if (last_reward) last_reward[i] = (int8_t)rw;
}

__global__ void __launch_bounds__(256) k_env_random_action(uint64_t seed, uint64_t game0, int step, const uint64_t *legal,
                                                           uint8_t *action, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    action[i] = (uint8_t)playout_action(seed, game0 + (uint64_t)i, (uint64_t)step, legal[i]);
}

// ---- feature planes -------------------------------------------------------------------------
template <int DT> struct Elem;
template <> struct Elem<MZ_F32> { using T = float; static __device__ T one() { return 1.0f; } static __device__ T zero() { return 0.0f; } };
template <> struct Elem<MZ_BF16> { using T = uint16_t; static __device__ T one() { return 0x3F80; } static __device__ T zero() { return 0; } };
template <> struct Elem<MZ_F16> { using T = uint16_t; static __device__ T one() { return 0x3C00; } static __device__ T zero() { return 0; } };

template <typename T> struct alignas(4 * sizeof(T)) Vec4 { T v[4]; };

// Action.encode reversi.py:32-36: rows r and c of the mover's layer (whole layer for pass)
__device__ __forceinline__ unsigned encode_rows(int a) { return a >= 64 ? 0xFFu : ((1u << (a >> 3)) | (1u << (a & 7))); }

template <int DT>
__global__ void __launch_bounds__(256) k_action_encode(const uint8_t *action, const int8_t *player, typename Elem<DT>::T *out, int64_t n) {
    using T = typename Elem<DT>::T;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // one thread per 4 cells: n * 32 threads
    if (idx >= n * 32) return;
    const int64_t g = idx >> 5;
    const int q = (int)(idx & 31), layer = q >> 4, row = (q & 15) >> 1;
    const int my_layer = player[g] == 1 ? 0 : 1;
    const bool set = layer == my_layer && ((encode_rows(action[g]) >> row) & 1);
    Vec4<T> v;
    for (int k = 0; k < 4; ++k) v.v[k] = set ? Elem<DT>::one() : Elem<DT>::zero();
    reinterpret_cast<Vec4<T> *>(out)[idx] = v;
}

// Game.make_image reversi.py:301-312: planes (3-i)*2, (3-i)*2+1 = WHITE, BLACK discs of state
// index-i for i < min(index, 4); plane 8 = 1 iff index even.  One thread per 4 cells.
// ARENA: the Controller's observation of test.py:16-29 instead -- a sliding window of the last 4
// observed states INCLUDING the opening position (state index-i for i <= index), and the last plane
// set iff the mover is BLACK, which for lock-step games from the opening is again "index even".
template <int DT, bool ARENA>
__global__ void __launch_bounds__(256) k_make_image(const uint64_t *hw, const uint64_t *hb, int64_t stride, const int32_t *sidx,
                                                    int sidx_all, typename Elem<DT>::T *out, int64_t n) {
    using T = typename Elem<DT>::T;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * 144) return;
    const int64_t g = idx / 144;
    const int rem = (int)(idx - g * 144), p = rem >> 4, q = rem & 15;
    const int si = sidx ? sidx[g] : sidx_all;
    unsigned bits = 0;
    if (p == 8) {
        bits = (si & 1) ? 0u : 0xFu;
    } else {
        const int i = 3 - (p >> 1);  // how many states back
        if ((ARENA ? i <= si : i < si) && i < 4) {
            const uint64_t *h = (p & 1) ? hb : hw;
            bits = (unsigned)(h[(int64_t)(si - i) * stride + g] >> (4 * q)) & 0xFu;
        }
    }
    Vec4<T> v;
    for (int k = 0; k < 4; ++k) v.v[k] = ((bits >> k) & 1) ? Elem<DT>::one() : Elem<DT>::zero();
    reinterpret_cast<Vec4<T> *>(out)[idx] = v;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
extern "C" int mz_env_reset(uint64_t *black, uint64_t *white, int8_t *to_play, uint8_t *steps, uint8_t *done, uint64_t *legal,
                            int64_t n, void *stream) {
    MZ_REQUIRE(n >= 0, "mz_env_reset: n < 0");
    if (n == 0) return MZ_OK;
    MZ_REQUIRE(black && white && to_play && steps && done && legal, "mz_env_reset: null pointer");
    hipLaunchKernelGGL(k_env_reset, dim3(mz_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, black, white, to_play, steps, done, legal, n);
    MZ_LAUNCH_CHECK("k_env_reset");
    return MZ_OK;
}

extern "C" int mz_env_legal_mask(const uint64_t *black, const uint64_t *white, const int8_t *to_play, uint64_t *legal, int64_t n,
                                 void *stream) {
    MZ_REQUIRE(n >= 0, "mz_env_legal_mask: n < 0");
    if (n == 0) return MZ_OK;
    MZ_REQUIRE(black && white && to_play && legal, "mz_env_legal_mask: null pointer");
    hipLaunchKernelGGL(k_env_legal_mask, dim3(mz_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, black, white, to_play, legal, n);
    MZ_LAUNCH_CHECK("k_env_legal_mask");
    return MZ_OK;
}

extern "C" int mz_env_step(uint64_t *black, uint64_t *white, int8_t *to_play, uint8_t *steps, uint8_t *done, uint64_t *legal,
                           const uint8_t *action, int8_t *reward, uint8_t *status, int64_t n, void *stream) {
    MZ_REQUIRE(n >= 0, "mz_env_step: n < 0");
    if (n == 0) return MZ_OK;
    MZ_REQUIRE(black && white && to_play && steps && done && legal && action && reward && status, "mz_env_step: null pointer");
    hipLaunchKernelGGL(k_env_step, dim3(mz_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, black, white, to_play, steps, done,
                       legal, action, reward, status, n);
    MZ_LAUNCH_CHECK("k_env_step");
    return MZ_OK;
}

extern "C" int mz_env_step_packed(uint64_t *black, uint64_t *white, uint64_t *legal, uint32_t *meta, int64_t n, void *stream) {
    MZ_REQUIRE(n >= 0, "mz_env_step_packed: n < 0");
    if (n == 0) return MZ_OK;
    MZ_REQUIRE(black && white && legal && meta, "mz_env_step_packed: null pointer");
    hipLaunchKernelGGL(k_env_step_packed, dim3(mz_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, black, white, legal, meta, n);
    MZ_LAUNCH_CHECK("k_env_step_packed");
    return MZ_OK;
}

extern "C" int mz_env_random_playout(uint64_t seed, uint64_t game0, int64_t n, int32_t n_steps, int32_t voluntary, uint64_t *black,
                                     uint64_t *white, int8_t *to_play, uint8_t *steps, uint8_t *done, uint64_t *legal,
                                     uint64_t *digest, uint8_t *actions, int8_t *last_reward, void *stream) {
    MZ_REQUIRE(n >= 0 && n_steps >= 0 && n_steps <= 255, "mz_env_random_playout: bad n/n_steps");
    if (n == 0) return MZ_OK;
    MZ_REQUIRE(black && white && to_play && steps && done && legal, "mz_env_random_playout: null state pointer");
    const dim3 grid(mz_blocks(n, 256)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define MZ_PLAYOUT(D, L)                                                                                                        \
    hipLaunchKernelGGL((k_env_random_playout<D, L>), grid, block, 0, s, seed, game0, n, (int)n_steps, (int)voluntary, black, white, \
                       to_play, steps, done, legal, digest, actions, last_reward)
    if (digest && actions) MZ_PLAYOUT(true, true);
    else if (digest) MZ_PLAYOUT(true, false);
    else if (actions) MZ_PLAYOUT(false, true);
    else MZ_PLAYOUT(false, false);
#undef MZ_PLAYOUT
    MZ_LAUNCH_CHECK("k_env_random_playout");
    return MZ_OK;
}

extern "C" int mz_env_random_action(uint64_t seed, uint64_t game0, int32_t step, const uint64_t *legal, uint8_t *action, int64_t n,
                                    void *stream) {
    MZ_REQUIRE(n >= 0, "mz_env_random_action: n < 0");
    if (n == 0) return MZ_OK;
    MZ_REQUIRE(legal && action, "mz_env_random_action: null pointer");
    hipLaunchKernelGGL(k_env_random_action, dim3(mz_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, seed, game0, (int)step, legal,
                       action, n);
    MZ_LAUNCH_CHECK("k_env_random_action");
    return MZ_OK;
}

extern "C" int mz_action_encode(const uint8_t *action, const int8_t *player, void *out, int32_t dtype, int64_t n, void *stream) {
    MZ_REQUIRE(n >= 0, "mz_action_encode: n < 0");
    if (n == 0) return MZ_OK;
    MZ_REQUIRE(action && player && out, "mz_action_encode: null pointer");
    const dim3 grid(mz_blocks(n * 32, 256)), block(256);
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case MZ_F32: hipLaunchKernelGGL(k_action_encode<MZ_F32>, grid, block, 0, s, action, player, (float *)out, n); break;
        case MZ_BF16: hipLaunchKernelGGL(k_action_encode<MZ_BF16>, grid, block, 0, s, action, player, (uint16_t *)out, n); break;
        case MZ_F16: hipLaunchKernelGGL(k_action_encode<MZ_F16>, grid, block, 0, s, action, player, (uint16_t *)out, n); break;
        default: MZ_REQUIRE(false, "mz_action_encode: bad dtype %d", dtype);
    }
    MZ_LAUNCH_CHECK("k_action_encode");
    return MZ_OK;
}

static int launch_make_image(const char *who, bool arena, const uint64_t *hist_white, const uint64_t *hist_black, int64_t hist_stride,
                             const int32_t *state_index, int32_t state_index_all, void *out, int32_t dtype, int64_t n, void *stream) {
    MZ_REQUIRE(n >= 0, "%s: n < 0", who);
    if (n == 0) return MZ_OK;
    MZ_REQUIRE(hist_white && hist_black && out, "%s: null pointer", who);
    MZ_REQUIRE(hist_stride >= n, "%s: hist_stride < n", who);
    MZ_REQUIRE(state_index || (state_index_all >= 0 && state_index_all <= 255), "%s: state_index_all out of range", who);
    const dim3 grid(mz_blocks(n * 144, 256)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define MZ_IMG(DT, T)                                                                                                              \
    if (arena) hipLaunchKernelGGL((k_make_image<DT, true>), grid, block, 0, s, hist_white, hist_black, hist_stride, state_index,       \
                                  (int)state_index_all, (T *)out, n);                                                                  \
    else hipLaunchKernelGGL((k_make_image<DT, false>), grid, block, 0, s, hist_white, hist_black, hist_stride, state_index,            \
                            (int)state_index_all, (T *)out, n)
    switch (dtype) {
        case MZ_F32: MZ_IMG(MZ_F32, float); break;
        case MZ_BF16: MZ_IMG(MZ_BF16, uint16_t); break;
        case MZ_F16: MZ_IMG(MZ_F16, uint16_t); break;
        default: MZ_REQUIRE(false, "%s: bad dtype %d", who, dtype);
    }
#undef MZ_IMG
    MZ_LAUNCH_CHECK("k_make_image");
    return MZ_OK;
}

extern "C" int mz_make_image(const uint64_t *hist_white, const uint64_t *hist_black, int64_t hist_stride, const int32_t *state_index,
                             int32_t state_index_all, void *out, int32_t dtype, int64_t n, void *stream) {
    return launch_make_image("mz_make_image", false, hist_white, hist_black, hist_stride, state_index, state_index_all, out, dtype, n, stream);
}

extern "C" int mz_controller_image(const uint64_t *hist_white, const uint64_t *hist_black, int64_t hist_stride, const int32_t *state_index,
                                   int32_t state_index_all, void *out, int32_t dtype, int64_t n, void *stream) {
    return launch_make_image("mz_controller_image", true, hist_white, hist_black, hist_stride, state_index, state_index_all, out, dtype, n,
                             stream);
}
