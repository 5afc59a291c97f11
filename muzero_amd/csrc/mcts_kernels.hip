// mcts_kernels.hip -- batched MuZero tree search for gfx950: struct-of-arrays node pool in HBM,
// ONE WAVEFRONT PER GAME TREE.
//
// Replaces main.py:99-196 (run_mcts, select_child, ucb_score, expand_node, backpropagate,
// add_exploration_noise), main.py:128-140/275-285 (select_action, softmax_sample),
// reversi.py:234-251 (Node) and utilities.py:114-129 (MinMaxStats).
//
// Mapping.  A node's 65 children are one 16-byte record each; lane l of the tree's wave owns child
// l (one coalesced 1 KiB load per tree level), child 64 (pass) is loaded wave-uniformly and scored
// redundantly by every lane.  A child's statistics are consumed by exactly one lane, so they stay
// in VGPRs -- an LDS round trip would be pure overhead -- and the argmax is wavefront ballots and
// shuffle reductions.  No LDS is used by any kernel in this file.
//
// Numerics.  The reference mixes Python floats (double) and fp32 tensors:
//   prior, pb_c, prior_score        double
//   child.value() = W/N             fp32 tensor when N > 0, Python int 0 when N == 0
//   normalize / prior_score + value fp32 if any operand is a tensor, else double
//   score comparisons               fp32 if either side is a tensor, else double
// `Mx` models that.  Builtin max() over (score, action) tuples is a sequential scan that replaces
// on "greater"; mixed comparisons make that order non-transitive, so select_argmax() below
// computes the *exact* result of the sequential scan with wave-parallel primitives (derivation in
// DESIGN.md, "select_child as a parallel scan").  log/sqrt of the pb_c term come from a host table
// (glibc, as CPython's math.log/math.sqrt) so no device libm rounding enters the score.
// This file must be compiled with -ffp-contract=off.
#include <math.h>
#include <vector>
#include "common.h"

namespace {

struct alignas(16) ChildRec {
    uint32_t nc;  // visit_count | (expanded id + 1) << 16 ; NOT_CHILD = not in parent's children dict
    float w;      // value_sum
    double p;     // prior
};
constexpr uint32_t NOT_CHILD = 0xFFFFFFFFu;
struct alignas(4) PathRec { uint16_t node; uint8_t action; uint8_t pad; };
struct alignas(16) PbcRec { double a, s; };

struct Pool {
    ChildRec *stats;
    char *hidden;
    mz_tree_meta *meta;
    PathRec *path;
    const PbcRec *pbc;
    int max_nodes, hidden_elem_bytes;
    uint32_t one16x2;  // two packed 1.0 values of the 16-bit hidden dtype
};

Pool make_pool(const void *base, const mz_pool_layout *L) {
    char *b = (char *)base;
    Pool p;
    p.stats = (ChildRec *)(b + L->off_stats);
    p.hidden = b + L->off_hidden;
    p.meta = (mz_tree_meta *)(b + L->off_meta);
    p.path = (PathRec *)(b + L->off_path);
    p.pbc = (const PbcRec *)(b + L->off_pbc);
    p.max_nodes = L->max_nodes;
    p.hidden_elem_bytes = L->hidden_elem_bytes;
    p.one16x2 = L->hidden_dtype == MZ_BF16 ? 0x3F803F80u : 0x3C003C00u;
    return p;
}

// ---- mixed Python-float / fp32-tensor numbers ---------------------------------------------------
struct Mx { double d; bool t; };  // t: value is an fp32 tensor (d holds it exactly)
__device__ __forceinline__ Mx mx_py(double d) { return Mx{d, false}; }
__device__ __forceinline__ Mx mx_t(float f) { return Mx{(double)f, true}; }
__device__ __forceinline__ float mx_f(Mx a) { return (float)a.d; }
__device__ __forceinline__ Mx mx_sub(Mx a, Mx b) { return (a.t || b.t) ? mx_t(mx_f(a) - mx_f(b)) : mx_py(a.d - b.d); }
__device__ __forceinline__ Mx mx_add(Mx a, Mx b) { return (a.t || b.t) ? mx_t(mx_f(a) + mx_f(b)) : mx_py(a.d + b.d); }
__device__ __forceinline__ Mx mx_div(Mx a, Mx b) { return (a.t || b.t) ? mx_t(mx_f(a) / mx_f(b)) : mx_py(a.d / b.d); }
__device__ __forceinline__ bool mx_gt(Mx a, Mx b) { return (a.t || b.t) ? (mx_f(a) > mx_f(b)) : (a.d > b.d); }
__device__ __forceinline__ bool mx_lt(Mx a, Mx b) { return (a.t || b.t) ? (mx_f(a) < mx_f(b)) : (a.d < b.d); }

struct MinMax { Mx lo, hi; };
// MinMaxStats.normalize utilities.py:125-129
__device__ __forceinline__ Mx mm_normalize(const MinMax &mm, Mx v) {
    if (mx_gt(mm.hi, mm.lo)) return mx_div(mx_sub(v, mm.lo), mx_sub(mm.hi, mm.lo));
    return v;
}
__device__ __forceinline__ MinMax load_minmax(const mz_tree_meta *m) {
    MinMax mm;
    const int f = m->mm_flags;
    mm.lo = Mx{m->mm_min, (f & 1) != 0};
    mm.hi = Mx{m->mm_max, (f & 2) != 0};
    return mm;
}

// ---- wave primitives ------------------------------------------------------------------------------
// Reductions on DPP (row operations + row broadcasts, the total read from lane 63): six dependent VALU steps.  __shfl_xor goes
// through ds_bpermute -- an address computation, an LDS crossbar trip and a wait per step -- and these kernels are bound by
// VALU issue and by exactly such dependent chains (profiles/r02/tree_experiment_latency.txt).  max / min are exact: same bits.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i32(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xF, false); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float v) { return __int_as_float(dpp_i32<CTRL, ROW_MASK>(__float_as_int(v), __float_as_int(v))); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
    return __hiloint2double(dpp_i32<CTRL, ROW_MASK>(__double2hiint(v), __double2hiint(v)), dpp_i32<CTRL, ROW_MASK>(__double2loint(v), __double2loint(v)));
}
// (a lane outside the row mask keeps its own value: op(v, v) = v)
#define MZ_WAVE_REDUCE(T, OP, DPP)                                                                                     \
    v = OP(v, DPP<0xB1, 0xF>(v));  /* quad_perm [1,0,3,2] */                                                           \
    v = OP(v, DPP<0x4E, 0xF>(v));  /* quad_perm [2,3,0,1] */                                                           \
    v = OP(v, DPP<0x124, 0xF>(v)); /* row_ror:4 */                                                                     \
    v = OP(v, DPP<0x128, 0xF>(v)); /* row_ror:8  -> every lane holds its 16-lane row's result */                       \
    v = OP(v, DPP<0x142, 0xA>(v)); /* row_bcast:15 into rows 1, 3 */                                                   \
    v = OP(v, DPP<0x143, 0xC>(v)); /* row_bcast:31 into rows 2, 3 -> lane 63 holds the wave's result */
__device__ __forceinline__ float wave_max_f32(float v) {
    MZ_WAVE_REDUCE(float, fmaxf, dpp_f32)
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_min_f32(float v) {
    MZ_WAVE_REDUCE(float, fminf, dpp_f32)
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ double wave_max_f64(double v) {
    MZ_WAVE_REDUCE(double, fmax, dpp_f64)
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
#undef MZ_WAVE_REDUCE
// lane `l` of a value, l wave-uniform: v_readlane instead of a ds_bpermute round trip
__device__ __forceinline__ int lane_i32(int v, int l) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(l)); }
__device__ __forceinline__ int hi_bit(uint64_t m) { return 63 - __clzll((long long)m); }
__device__ __forceinline__ int lo_bit(uint64_t m) { return __ffsll((unsigned long long)m) - 1; }
__device__ __forceinline__ float lane_f32(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// ucb_score main.py:153-160 for one child record
// The kind of the MinMaxStats bounds is the same for every lane of the tree's wave.  Once a backup has updated them they are
// fp32 tensors, and then every operation of normalize() has a tensor operand: fp32 throughout, no fp64 division.  That is the
// case of all but the first simulations of a search: the wave-uniform branch below runs it without the generic both-kinds
// arithmetic (same operations on the same values: same bits).
__device__ __forceinline__ Mx ucb_score(const ChildRec &r, const PbcRec &pb, const MinMax &mm, bool negamax) {
    const uint32_t n = r.nc & 0xFFFFu;
    const double pb_c = pb.a * (pb.s / (double)(n + 1u));
    const double prior_score = pb_c * r.p;
    Mx value = n == 0 ? mx_py(0.0) : mx_t(r.w / (float)n);  // Node.value() reversi.py:248-251
    if (negamax && n != 0) value.d = -value.d;                // non-default: the child's value seen from the parent's mover
    if (mm.lo.t && mx_gt(mm.hi, mm.lo)) {
        // normalize with a tensor minimum: (v - lo) is a tensor, so is (hi - lo), so is the quotient, so is the sum
        const float lo = mx_f(mm.lo);
        const float q = (mx_f(value) - lo) / (mx_f(mm.hi) - lo);
        return mx_t((float)prior_score + q);
    }
    return mx_add(mx_py(prior_score), mm_normalize(mm, value));
}

// select_child main.py:144-148: exact result of builtin max over (score, action) tuples scanned in
// ascending action order, computed wave-parallel.  Lane l holds child l (valid/s); child 64 is
// wave-uniform (valid64/s64).  key = fp32(score) totally orders items of different kind; items of
// Python-float kind (D) are ordered among themselves in double; a tensor-kind item (F) compares
// equal to every item with the same key.  See DESIGN.md for the proof of equivalence.
__device__ __forceinline__ int select_argmax(bool valid, Mx s, bool valid64, Mx s64, int lane, bool tie_lowest) {
    {   // max() never replaces its running result across a comparison with NaN (every such comparison is False): NaN items are
        // skipped -- the fmax reductions below do that -- but a NaN FIRST item stays the result.  (Scores are NaN when the
        // policy mass of the expanded actions is zero: priors 0/0.)
        const uint64_t mvalid = __ballot(valid), mnan = __ballot(valid && s.d != s.d);
        if (mvalid) {
            const int first = lo_bit(mvalid);
            if ((mnan >> first) & 1ULL) return first;
        } else if (valid64 && s64.d != s64.d) {
            return 64;
        }
    }
    const float key = valid ? mx_f(s) : -INFINITY;
    const float key64 = valid64 ? mx_f(s64) : -INFINITY;
    const float K = fmaxf(wave_max_f32(key), key64);
    const bool at = valid && key == K, at64 = valid64 && key64 == K;
    const uint64_t mF = __ballot(at && s.t), mD = __ballot(at && !s.t);
    const bool F64 = at64 && s64.t, D64 = at64 && !s64.t;
    if (!tie_lowest) {
        // j = last tensor-kind item at key K: it replaces whatever precedes it; after it only
        // D items remain, which are maximised in double with ties to the later index.
        const int j = F64 ? 64 : (mF ? hi_bit(mF) : -1);
        if (j == 64) return 64;
        const uint64_t below = j < 0 ? 0ULL : (j == 63 ? ~0ULL : ((2ULL << j) - 1ULL));
        const uint64_t cand = mD & ~below;
        const int ncand = __popcll(cand) + (D64 ? 1 : 0);
        if (ncand == 0) return j;
        if (ncand == 1) return D64 ? 64 : hi_bit(cand);
        const bool mine = (cand >> lane) & 1ULL;
        double dmax = wave_max_f64(mine ? s.d : -INFINITY);
        if (D64) dmax = fmax(dmax, s64.d);
        if (D64 && s64.d == dmax) return 64;
        return hi_bit(__ballot(mine && s.d == dmax));
    }
    // non-default: replace only on strictly greater -> the first item wins ties
    const uint64_t mall = mF | mD;
    if (!mall) return 64;
    const int first = lo_bit(mall);
    if ((mF >> first) & 1ULL) return first;  // a tensor-kind first item is never exceeded at key K
    const bool mine = (mD >> lane) & 1ULL;
    double dmax = wave_max_f64(mine ? s.d : -INFINITY);
    if (D64) dmax = fmax(dmax, s64.d);
    const uint64_t me = __ballot(mine && s.d == dmax);
    return me ? lo_bit(me) : 64;
}

// Action.encode rows (reversi.py:32-36; see env_kernels.hip)
__device__ __forceinline__ unsigned encode_rows(int a) { return a >= 64 ? 0xFFu : ((1u << (a >> 3)) | (1u << (a & 7))); }

// copy one hidden state (512 elements) with the whole wave, 16 B per lane per step
__device__ __forceinline__ void copy_hidden(char *dst, const char *src, int elem_bytes, int lane) {
    const uint4 *s = (const uint4 *)src;
    uint4 *d = (uint4 *)dst;
    const int n16 = (MZ_HIDDEN_ELEMS * elem_bytes) >> 4;  // 128 (f32) or 64 (16-bit)
    for (int i = lane; i < n16; i += 64) d[i] = s[i];
}

// expand_node main.py:165-175: priors over the masked actions, policy_sum sequential in double in
// ascending action order (H3).  Writes the 65 child records of `node`.
// the reference divides Python floats (main.py:171-175: .item(), then p / policy_sum): a policy mass of exactly 0.0 over the
// expanded actions raises ZeroDivisionError there.  Here the division yields NaN priors (which select_argmax treats as Python's
// max() treats NaN scores) and the tree's error word gets MZ_TREE_ERR_POLICY_MASS, as it does for a non-finite mass (NaN / inf
// network outputs: no exception in the reference, but nothing a search should go on with): callers raise on it (selfplay.py).
__device__ __forceinline__ bool bad_policy_mass(double sum) { return !(sum != 0.0) || !(fabs(sum) < (double)INFINITY); }

// Returns the policy mass it divided by.
template <bool ALL>
__device__ __forceinline__ double write_children(const Pool &P, int64_t node_row, const float *policy, uint64_t mask, bool pass_ok,
                                                 bool with_noise, double nz, double nz64, double frac, int lane) {
    const float p = policy[lane], p64 = policy[64];
    double sum = 0.0;
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        const float pi = lane_f32(p, i);
        if (ALL || ((mask >> i) & 1ULL)) sum += (double)pi;
    }
    if (ALL || pass_ok) sum += (double)p64;
    ChildRec *recs = P.stats + node_row * MZ_ACTIONS;
    const bool ex = ALL || ((mask >> lane) & 1ULL);
    double prior = (double)p / sum, prior64 = (double)p64 / sum;
    if (with_noise) {  // add_exploration_noise main.py:191-196 (nz: this lane's action, nz64: the pass action)
        prior = prior * (1 - frac) + nz * frac;
        prior64 = prior64 * (1 - frac) + nz64 * frac;
    }
    recs[lane] = ex ? ChildRec{0u, 0.0f, prior} : ChildRec{NOT_CHILD, 0.0f, 0.0};
    if (lane == 0) recs[64] = (ALL || pass_ok) ? ChildRec{0u, 0.0f, prior64} : ChildRec{NOT_CHILD, 0.0f, 0.0};
    return sum;
}

// ---- np.random.dirichlet([alpha] * n_children) (main.py:192-193) drawn on the device, by action -----------------------------
// The reference draws from numpy's global generator, which nothing seeds: any generator with the right law serves.
// Here: a counter-based stream keyed (cfg.noise_seed, tree, number of searches this tree slot has started), so a search is
// reproducible from the seed and the count, needs no generator state in memory and can sit in a captured graph.
// Gamma(alpha) per child by Marsaglia-Tsang on shape alpha + 1 times U^(1/alpha) (alpha < 1), kept as a LOGARITHM: at
// alpha = 0.03 most draws underflow a double (U^33), and the normalisation only needs differences to the largest.
__device__ __forceinline__ uint64_t mix64(uint64_t z) {   // splitmix64's finaliser (as env_kernels.hip)
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__device__ __forceinline__ double u01(uint64_t key, uint32_t draw) {   // uniform in (0, 1), 53 bits
    return ((double)(mix64(key + (uint64_t)draw * 0xD1342543DE82EF95ULL) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ double log_gamma_draw(double alpha, uint64_t key) {
    const double a = alpha < 1.0 ? alpha + 1.0 : alpha, d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    double g = d;   // (32 rejections in a row: probability < 1e-40)
    for (uint32_t it = 0; it < 32; ++it) {
        const double u1 = u01(key, 4 * it), u2 = u01(key, 4 * it + 1), u3 = u01(key, 4 * it + 2);
        const double x = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
        double v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        if (log(u3) < 0.5 * x * x + d - d * v + d * log(v)) {
            g = d * v;
            break;
        }
    }
    double lg = log(g);
    if (alpha < 1.0) lg += log(u01(key, 0xFFFFFFFFu)) / alpha;
    return lg;
}
__device__ __forceinline__ double wave_sum_f64(double v) {   // (once per search: the shuffle form will do)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// -> nz (this lane's action) and nz64 (the pass action): a Dirichlet(alpha, ..., alpha) sample over the root's children
__device__ __forceinline__ void device_dirichlet(double alpha, uint64_t seed, int t, uint32_t n_search, uint64_t mask, bool pass, int lane,
                                                 double &nz, double &nz64) {
    const uint64_t tree_key = mix64(mix64(seed) + (uint64_t)t * 0xD1342543DE82EF95ULL + (uint64_t)n_search * 0x9E3779B97F4A7C15ULL);
    const bool ex = (mask >> lane) & 1ULL;
    const double lg = ex ? log_gamma_draw(alpha, tree_key + (uint64_t)(lane + 1) * 0xA0761D6478BD642FULL) : -(double)INFINITY;
    const double lg64 = pass ? log_gamma_draw(alpha, tree_key + 65ULL * 0xA0761D6478BD642FULL) : -(double)INFINITY;   // wave-uniform
    const double m = fmax(wave_max_f64(lg), lg64);
    const double e = ex ? exp(lg - m) : 0.0, e64 = pass ? exp(lg64 - m) : 0.0;
    const double s = wave_sum_f64(e) + e64;
    nz = e / s;
    nz64 = e64 / s;
}

// ---- root set-up (play_game main.py:75-83, run_mcts :100) ----------------------------------------
__global__ void __launch_bounds__(64) k_root_init(Pool P, mz_search_cfg cfg, const int8_t *to_play, const uint64_t *legal,
                                                  const uint8_t *pass_ok, const float *policy, const double *noise, const char *hidden) {
    const int t = blockIdx.x, lane = threadIdx.x;
    const uint64_t mask = legal[t];
    const int64_t tbase = (int64_t)t * P.max_nodes;
    const bool pass = pass_ok ? pass_ok[t] != 0 : mask == 0;  // reversi.py:125-126: pass only when nothing else is legal
    const int32_t n_search = P.meta[t].pad[0];   // searches this tree slot has started (mz_pool_init zeroes it)
    double nz = 0.0, nz64 = 0.0;
    bool with_noise = noise != nullptr;
    if (noise) {
        nz = noise[(int64_t)t * MZ_ACTIONS + lane];
        nz64 = noise[(int64_t)t * MZ_ACTIONS + 64];
    } else if (cfg.device_noise && (mask != 0 || pass)) {
        with_noise = true;
        device_dirichlet(cfg.noise_alpha, cfg.noise_seed, t, (uint32_t)n_search, mask, pass, lane, nz, nz64);
    }
    const double mass = write_children<false>(P, tbase, policy + (int64_t)t * MZ_ACTIONS, mask, pass, with_noise, nz, nz64, cfg.noise_frac, lane);
    const int eb = P.hidden_elem_bytes;
    if (hidden) copy_hidden(P.hidden + tbase * (int64_t)(MZ_HIDDEN_ELEMS * eb), hidden + (int64_t)t * (MZ_HIDDEN_ELEMS * eb), eb, lane);
    if (lane == 0) {
        mz_tree_meta m;
        m.n_nodes = 1;
        m.root_n = 0;
        m.root_w = 0.0f;
        m.root_player = to_play[t];
        m.mm_min = cfg.bounds_none ? (double)INFINITY : cfg.bound_lo;  // utilities.py:117-119
        m.mm_max = cfg.bounds_none ? -(double)INFINITY : cfg.bound_hi;
        m.mm_flags = 0;
        m.path_len = 0;
        m.leaf_player = 0;
        m.error = bad_policy_mass(mass) ? MZ_TREE_ERR_POLICY_MASS : 0;
        m.leaf_parent = 0;
        m.leaf_action = 0;
        m.pad[0] = n_search + 1;
        m.pad[1] = 0;
        P.meta[t] = m;
    }
}

// ---- where a descent leaves the network input of its leaf ------------------------------------------------------------------
//   rows   (net_in != null): one (10,8,8) row per tree = parent hidden (8 planes, copied out of the pool) ++ action planes
//   routed (net_in == null): nothing is copied.  The inference launch reads the parent's hidden state from the pool slot
//          src_slot[t] and writes the new node's hidden state straight into the slot dst_slot[t] (mz_inference_forward_routed);
//          only the two action planes travel through a staging row.  The new node's id is known before it is expanded: ids
//          are handed out in order, so it is the tree's node count.
struct LeafOut {
    char *net_in;
    int32_t *src_slot, *dst_slot;
    char *act;   // (n,2,8,8) of the pool's hidden dtype
};
__device__ __forceinline__ void write_action_planes(const Pool &P, char *dst, int a, int player, int lane) {
    // Action(a, child's player).encode() (reversi.py:32-36): rows r and c of the mover's layer, pass = whole layer
    const int layer = lane >> 5, cell = (2 * lane) & 63, my_layer = player == 1 ? 0 : 1;
    const bool set = layer == my_layer && ((encode_rows(a) >> (cell >> 3)) & 1u);
    if (P.hidden_elem_bytes == 4) ((float2 *)dst)[lane] = set ? make_float2(1.0f, 1.0f) : make_float2(0.0f, 0.0f);
    else ((uint32_t *)dst)[lane] = set ? P.one16x2 : 0u;
}
__device__ __forceinline__ void write_route(const Pool &P, const LeafOut &out, int t, int64_t tbase, int node, int next_id, int a, int player, int lane) {
    if (lane == 0) {
        out.src_slot[t] = (int32_t)(tbase + node);
        out.dst_slot[t] = (int32_t)(tbase + (next_id < P.max_nodes ? next_id : P.max_nodes - 1));   // a full pool is flagged; never out of range
    }
    write_action_planes(P, out.act + (int64_t)t * (128 * P.hidden_elem_bytes), a, player, lane);
}

// ---- one simulation's descent (run_mcts main.py:103-118) ------------------------------------------
__device__ __forceinline__ void tree_select(const Pool &P, const mz_search_cfg &cfg, const LeafOut &out, int t, int lane) {
    mz_tree_meta *meta = &P.meta[t];
    const MinMax mm = load_minmax(meta);
    const int64_t tbase = (int64_t)t * P.max_nodes;
    int node = 0, depth = 0, a = 0;
    uint32_t np = (uint32_t)meta->root_n;
    int player = meta->root_player;
    const bool tie_lowest = cfg.tie_lowest != 0;
    for (;;) {
        const ChildRec *recs = P.stats + (tbase + node) * MZ_ACTIONS;
        const ChildRec r = recs[lane];
        const ChildRec r64 = recs[64];
        const PbcRec pb = P.pbc[np];
        const Mx s = ucb_score(r, pb, mm, cfg.negamax != 0);
        Mx s64 = mx_py(0.0);   // child 64 (pass) is the same record in every lane: scored only where it is a child (never beside placements at a root)
        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)r64.nc) != NOT_CHILD) s64 = ucb_score(r64, pb, mm, cfg.negamax != 0);
        a = select_argmax(r.nc != NOT_CHILD, s, r64.nc != NOT_CHILD, s64, lane, tie_lowest);
        const uint32_t nc = a == 64 ? r64.nc : (uint32_t)lane_i32((int)r.nc, a);
        if (lane == 0) P.path[tbase + depth] = PathRec{(uint16_t)node, (uint8_t)a, 0};
        ++depth;
        player = -player;  // children belong to the other player (main.py:173)
        const uint32_t child = nc >> 16;
        if (child == 0 || depth >= P.max_nodes) break;  // Node.expanded() false -> leaf
        node = (int)child - 1;
        np = nc & 0xFFFFu;
    }
    if (lane == 0) {
        meta->path_len = depth;
        meta->leaf_player = player;
        meta->leaf_parent = node;
        meta->leaf_action = a;
    }
    if (!out.net_in) {
        write_route(P, out, t, tbase, node, meta->n_nodes, a, player, lane);
        return;
    }
    // network input row: parent hidden (8 planes) ++ Action(a, child's player).encode() (2 planes)
    const int eb = P.hidden_elem_bytes;
    char *row = out.net_in + (int64_t)t * (640 * eb);
    copy_hidden(row, P.hidden + (tbase + node) * (int64_t)(MZ_HIDDEN_ELEMS * eb), eb, lane);
    write_action_planes(P, row + 512 * eb, a, player, lane);
}

// ---- expand the leaf (main.py:123, 165-175) -------------------------------------------------------
// Returns the new node's id, or -1 when the pool is full / no descent preceded (flagged in meta).
__device__ __forceinline__ int tree_expand(const Pool &P, const float *policy, const char *hidden, int t, int lane) {
    mz_tree_meta *meta = &P.meta[t];
    const int new_id = meta->n_nodes, depth = meta->path_len;
    if (new_id >= P.max_nodes || depth < 1) {  // leave the tree untouched
        if (lane == 0) meta->error |= 1;
        return -1;
    }
    const int64_t tbase = (int64_t)t * P.max_nodes;
    const double mass = write_children<true>(P, tbase + new_id, policy + (int64_t)t * MZ_ACTIONS, ~0ULL, true, false, 0.0, 0.0, 0.0, lane);
    if (lane == 0 && bad_policy_mass(mass)) meta->error |= MZ_TREE_ERR_POLICY_MASS;
    const int eb = P.hidden_elem_bytes;
    if (hidden)   // (null: the inference launch has written the node's hidden state into its slot already)
        copy_hidden(P.hidden + (tbase + new_id) * (int64_t)(MZ_HIDDEN_ELEMS * eb), hidden + (int64_t)t * (MZ_HIDDEN_ELEMS * eb), eb, lane);
    if (lane == 0) {  // link the leaf to its node: parent's child record gets the expanded id
        const PathRec pr = P.path[tbase + depth - 1];
        ChildRec *rec = P.stats + (tbase + pr.node) * MZ_ACTIONS + pr.action;
        rec->nc = (rec->nc & 0xFFFFu) | ((uint32_t)(new_id + 1) << 16);
        meta->n_nodes = new_id + 1;
    }
    return new_id;
}

// ---- backpropagate (main.py:180-186) along the stored path ----------------------------------------
// search_path[k], k = 0..depth (root .. leaf); node k>=1 lives in its parent's child record
// (path[k-1].node, path[k-1].action).  Players alternate with depth, so node k has the leaf's
// player iff (depth-k) is even.  value_k = value after k times `reward + discount*value`.
__device__ __forceinline__ void tree_backup(const Pool &P, const mz_search_cfg &cfg, float v0, int t, int lane) {
    mz_tree_meta *meta = &P.meta[t];
    const int depth = meta->path_len;
    const int64_t tbase = (int64_t)t * P.max_nodes;
    const float disc = (float)cfg.discount;
    const MinMax mm = load_minmax(meta);
    float vmin = INFINITY, vmax = -INFINITY;
    for (int k0 = 0; k0 <= depth; k0 += 64) {
        const int k = k0 + lane;
        if (k <= depth) {
            float v = v0;
            for (int i = 0; i < k; ++i) v = 0.0f + disc * v;  // node.reward (0) + discount * value
            const float sv = ((depth - k) & 1) ? -v : v;
            float nv;
            if (k == 0) {
                const float w = meta->root_w + sv;
                const int n = meta->root_n + 1;
                meta->root_w = w;
                meta->root_n = n;
                nv = w / (float)n;
            } else {
                const PathRec pr = P.path[tbase + k - 1];
                ChildRec *rec = P.stats + (tbase + pr.node) * MZ_ACTIONS + pr.action;
                const uint32_t nc = rec->nc + 1u;
                const float w = rec->w + sv;
                rec->nc = nc;
                rec->w = w;
                nv = w / (float)(nc & 0xFFFFu);
            }
            vmin = fminf(vmin, nv);
            vmax = fmaxf(vmax, nv);
        }
    }
    // MinMaxStats.update (utilities.py:121-123) with every node.value() of the path; builtin
    // min/max replace only on strictly smaller/greater, which is order independent.
    vmin = wave_min_f32(vmin);
    vmax = wave_max_f32(vmax);
    if (lane == 0) {
        int f = meta->mm_flags;
        if (mx_lt(mx_t(vmin), mm.lo)) { meta->mm_min = (double)vmin; f |= 1; }
        if (mx_gt(mx_t(vmax), mm.hi)) { meta->mm_max = (double)vmax; f |= 2; }
        meta->mm_flags = f;
    }
}

__device__ __forceinline__ void tree_expand_backup(const Pool &P, const mz_search_cfg &cfg, const float *value, const float *policy,
                                                   const char *hidden, int t, int lane) {
    if (tree_expand(P, policy, hidden, t, lane) < 0) return;
    __syncthreads();  // lane 0 linked the leaf record that the backup's lane `depth` updates next
    tree_backup(P, cfg, value[t], t, lane);
}

__global__ void __launch_bounds__(64) k_select(Pool P, mz_search_cfg cfg, LeafOut out) {
    tree_select(P, cfg, out, blockIdx.x, threadIdx.x);
}
__global__ void __launch_bounds__(64) k_expand_backup(Pool P, mz_search_cfg cfg, const float *value, const float *policy,
                                                      const char *hidden) {
    tree_expand_backup(P, cfg, value, policy, hidden, blockIdx.x, threadIdx.x);
}
// ---- the fused simulation step, latency-optimised ------------------------------------------------
// expand_node + backpropagate of simulation s and the descent of simulation s+1 for one tree, as the
// three device functions above compute them -- same arithmetic, same stores -- but organised around
// the one thing that bounds this kernel: the chain of DEPENDENT cold loads (each ~1-2 us from HBM).
//   round 0  everything whose address is known at launch is requested together: meta, the old path,
//            policy row, value, the new hidden state, and the ROOT's 65 child records
//   round 1  the backup's records (addresses from the old path) and the root's pb_c entry
//   then     the descent runs on registers: records this step has just modified are never re-read
//            from memory -- the root's and every node's on the old path are patched in registers
//            with the backup lane's new (N, W), the freshly expanded node's records are rebuilt from
//            the priors it was given, and its hidden state goes to the network input straight from
//            the registers that copied it in.  Only nodes off the old path cost a round each.
// Typical step (old depth 1-2, new depth 1-2): 3-4 rounds instead of 8-9.  Paths longer than 62
// edges (degenerate chains) take the plain three-function route.
__global__ void __launch_bounds__(64) k_backup_select(Pool P, mz_search_cfg cfg, const float *value, const float *policy,
                                                      const char *hidden, LeafOut out) {
    const int t = blockIdx.x, lane = threadIdx.x;
    mz_tree_meta *meta = &P.meta[t];
    const int depth = meta->path_len;
    if (depth > 62) {
        tree_expand_backup(P, cfg, value, policy, hidden, t, lane);
        __syncthreads();  // the descent re-reads records other lanes of this wave have just written
        tree_select(P, cfg, out, t, lane);
        return;
    }
    const int64_t tbase = (int64_t)t * P.max_nodes;
    const int eb = P.hidden_elem_bytes, n16 = (MZ_HIDDEN_ELEMS * eb) >> 4;  // 128 (f32) or 64 (16-bit) uint4 per hidden state
    // ---- round 0 ----
    const int new_id = meta->n_nodes;
    int root_n = meta->root_n;
    float root_w = meta->root_w;
    const int root_player = meta->root_player;
    MinMax mm = load_minmax(meta);
    int mm_flags = meta->mm_flags;
    const PathRec oldp = lane < P.max_nodes ? P.path[tbase + lane] : PathRec{0, 0, 0};  // old search_path edge `lane`
    const float pol = policy[(int64_t)t * MZ_ACTIONS + lane], pol64 = policy[(int64_t)t * MZ_ACTIONS + 64];
    const float v0 = value[t];
    // the new node's hidden state: copied into the pool from the inference launch's output row -- unless that launch wrote
    // it into the node's slot itself (hidden == null: the routed form, no hidden-state traffic in this kernel at all)
    uint4 hin0 = make_uint4(0, 0, 0, 0), hin1 = make_uint4(0, 0, 0, 0);
    if (hidden) {
        const uint4 *hsrc = reinterpret_cast<const uint4 *>(hidden + (int64_t)t * (MZ_HIDDEN_ELEMS * eb));
        hin0 = hsrc[lane];
        if (n16 > 64) hin1 = hsrc[lane + 64];
    }
    const ChildRec *root_recs = P.stats + tbase * MZ_ACTIONS;
    ChildRec r = root_recs[lane], r64 = root_recs[64];

    const bool ok = new_id < P.max_nodes && depth >= 1;   // tree_expand's guard
    bool bad_mass = false;
    double prior = 0.0, prior64 = 0.0;
    uint32_t b_nc = 0;     // backup lane k: the new (nc, w) of search_path[k]'s record
    float b_w = 0.0f;
    if (ok) {
        // ---- expand_node main.py:165-175 over all 65 actions (write_children<true>, keeping the priors) ----
        double sum = 0.0;
#pragma unroll
        for (int i = 0; i < 64; ++i) sum += (double)lane_f32(pol, i);
        sum += (double)pol64;
        bad_mass = bad_policy_mass(sum);
        prior = (double)pol / sum;
        prior64 = (double)pol64 / sum;
        ChildRec *nrecs = P.stats + (tbase + new_id) * MZ_ACTIONS;
        nrecs[lane] = ChildRec{0u, 0.0f, prior};
        if (lane == 0) nrecs[64] = ChildRec{0u, 0.0f, prior64};
        if (hidden) {
            uint4 *hdst = reinterpret_cast<uint4 *>(P.hidden + (tbase + new_id) * (int64_t)(MZ_HIDDEN_ELEMS * eb));
            hdst[lane] = hin0;
            if (n16 > 64) hdst[lane + 64] = hin1;
        }
        // ---- backpropagate main.py:180-186: lane k owns search_path[k], k = 0 .. depth ----
        const int prev = __shfl_up(*reinterpret_cast<const int *>(&oldp), 1, 64);   // path[k-1]
        const int pnode = prev & 0xFFFF, pact = (prev >> 16) & 0xFF;
        const float disc = (float)cfg.discount;
        float vmin = INFINITY, vmax = -INFINITY;
        if (lane <= depth) {
            float v = v0;
            for (int i = 0; i < lane; ++i) v = 0.0f + disc * v;  // node.reward (0) + discount * value
            const float sv = ((depth - lane) & 1) ? -v : v;
            float nv;
            if (lane == 0) {
                root_w = root_w + sv;
                root_n = root_n + 1;
                nv = root_w / (float)root_n;
            } else {
                ChildRec *rec = P.stats + (tbase + pnode) * MZ_ACTIONS + pact;
                uint32_t nc = rec->nc;   // ---- round 1 ----
                const float w = rec->w + sv;
                if (lane == depth) nc = (nc & 0xFFFFu) | ((uint32_t)(new_id + 1) << 16);  // tree_expand's link of the leaf to its node
                nc += 1u;
                rec->nc = nc;
                rec->w = w;
                b_nc = nc;
                b_w = w;
                nv = w / (float)(nc & 0xFFFFu);
            }
            vmin = nv;
            vmax = nv;
        }
        root_n = lane_i32(root_n, 0);
        root_w = __int_as_float(lane_i32(__float_as_int(root_w), 0));
        vmin = wave_min_f32(vmin);
        vmax = wave_max_f32(vmax);
        // MinMaxStats.update (utilities.py:121-123), as tree_backup
        if (mx_lt(mx_t(vmin), mm.lo)) { mm.lo = mx_t(vmin); mm_flags |= 1; }
        if (mx_gt(mx_t(vmax), mm.hi)) { mm.hi = mx_t(vmax); mm_flags |= 2; }
    }

    // ---- descent of the next simulation (tree_select) on registers ----
    const bool tie_lowest = cfg.tie_lowest != 0;
    int node = 0, level = 0, a = 0, player = root_player;
    uint32_t np = (uint32_t)root_n;
    for (;;) {
        if (ok && level < depth && (int)(lane_i32(*reinterpret_cast<const int *>(&oldp), level) & 0xFFFF) == node) {
            // this node's record for the old path's action was rewritten by backup lane level + 1: take it from there
            const int pa = (lane_i32(*reinterpret_cast<const int *>(&oldp), level) >> 16) & 0xFF;
            const uint32_t pnc = (uint32_t)lane_i32((int)b_nc, level + 1);
            const float pw = __int_as_float(lane_i32(__float_as_int(b_w), level + 1));
            if (pa == 64) { r64.nc = pnc; r64.w = pw; }
            else if (lane == pa) { r.nc = pnc; r.w = pw; }
        }
        const PbcRec pb = P.pbc[np];
        const Mx s = ucb_score(r, pb, mm, cfg.negamax != 0);
        Mx s64 = mx_py(0.0);   // child 64 (pass) is the same record in every lane: scored only where it is a child (never beside placements at a root)
        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)r64.nc) != NOT_CHILD) s64 = ucb_score(r64, pb, mm, cfg.negamax != 0);
        a = select_argmax(r.nc != NOT_CHILD, s, r64.nc != NOT_CHILD, s64, lane, tie_lowest);
        const uint32_t nc = a == 64 ? r64.nc : (uint32_t)lane_i32((int)r.nc, a);
        if (lane == 0) P.path[tbase + level] = PathRec{(uint16_t)node, (uint8_t)a, 0};
        ++level;
        player = -player;
        const uint32_t child = nc >> 16;
        if (child == 0 || level >= P.max_nodes) break;
        node = (int)child - 1;
        np = nc & 0xFFFFu;
        if (ok && node == new_id) {   // the node expanded a moment ago: its records are the priors it was given
            r = ChildRec{0u, 0.0f, prior};
            r64 = ChildRec{0u, 0.0f, prior64};
        } else {
            const ChildRec *recs = P.stats + (tbase + node) * MZ_ACTIONS;
            r = recs[lane];
            r64 = recs[64];
        }
    }
    if (lane == 0) {
        if (ok) {
            meta->n_nodes = new_id + 1;
            meta->root_w = root_w;
            meta->root_n = root_n;
            meta->mm_min = mm.lo.d;
            meta->mm_max = mm.hi.d;
            meta->mm_flags = mm_flags;
            if (bad_mass) meta->error |= MZ_TREE_ERR_POLICY_MASS;
        } else {
            meta->error |= 1;
        }
        meta->path_len = level;
        meta->leaf_player = player;
        meta->leaf_parent = node;
        meta->leaf_action = a;
    }
    if (!out.net_in) {
        write_route(P, out, t, tbase, node, ok ? new_id + 1 : new_id, a, player, lane);
        return;
    }
    // network input row: parent hidden (8 planes) ++ Action(a, child's player).encode() (2 planes)
    char *row = out.net_in + (int64_t)t * (640 * eb);
    uint4 *rdst = reinterpret_cast<uint4 *>(row);
    if (ok && node == new_id && hidden) {
        rdst[lane] = hin0;
        if (n16 > 64) rdst[lane + 64] = hin1;
    } else {
        const uint4 *ps = reinterpret_cast<const uint4 *>(P.hidden + (tbase + node) * (int64_t)(MZ_HIDDEN_ELEMS * eb));
        rdst[lane] = ps[lane];
        if (n16 > 64) rdst[lane + 64] = ps[lane + 64];
    }
    write_action_planes(P, row + 512 * eb, a, player, lane);
}

// ---- single-call entry points for the reference-shaped Python API (muzero_amd/compat.py) ----------
__global__ void __launch_bounds__(64) k_expand(Pool P, const float *policy, const char *hidden) {
    tree_expand(P, policy, hidden, blockIdx.x, threadIdx.x);
}
__global__ void __launch_bounds__(64) k_backup(Pool P, mz_search_cfg cfg, const float *value) {
    tree_backup(P, cfg, value[blockIdx.x], blockIdx.x, threadIdx.x);
}
// select_child / ucb_score for ONE node per tree: action_out[t] = argmax action (or -1 when the
// node is not expanded); score_out (nullable) [t][65] doubles, kind_out [t][65] bytes: 0 = Python
// float, 1 = fp32 tensor, 255 = not a child.
__global__ void __launch_bounds__(64) k_node_scores(Pool P, mz_search_cfg cfg, const int32_t *node_id, int32_t *action_out,
                                                    double *score_out, uint8_t *kind_out) {
    const int t = blockIdx.x, lane = threadIdx.x;
    const mz_tree_meta *meta = &P.meta[t];
    const MinMax mm = load_minmax(meta);
    const int64_t tbase = (int64_t)t * P.max_nodes;
    const int node = node_id[t];
    if (node < 0 || node >= meta->n_nodes) {
        if (lane == 0) action_out[t] = -1;
        return;
    }
    // the parent's visit count: root's own counter, else found in the record that links to `node`
    uint32_t np = (uint32_t)meta->root_n;
    if (node != 0) {
        np = 0;
        for (int q = 0; q < meta->n_nodes; ++q) {
            const ChildRec *rq = P.stats + (tbase + q) * MZ_ACTIONS;
            for (int a = lane; a < MZ_ACTIONS; a += 64) {
                const uint32_t nc = rq[a].nc;
                if (nc != NOT_CHILD && (int)(nc >> 16) == node + 1) np = nc & 0xFFFFu;
            }
        }
        np = (uint32_t)wave_max_f32((float)np);
    }
    const ChildRec *recs = P.stats + (tbase + node) * MZ_ACTIONS;
    const ChildRec r = recs[lane], r64 = recs[64];
    const PbcRec pb = P.pbc[np];
    const Mx s = ucb_score(r, pb, mm, cfg.negamax != 0), s64 = ucb_score(r64, pb, mm, cfg.negamax != 0);
    const int a = select_argmax(r.nc != NOT_CHILD, s, r64.nc != NOT_CHILD, s64, lane, cfg.tie_lowest != 0);
    if (lane == 0) action_out[t] = a;
    if (score_out) {
        score_out[(int64_t)t * MZ_ACTIONS + lane] = s.d;
        kind_out[(int64_t)t * MZ_ACTIONS + lane] = r.nc == NOT_CHILD ? 255 : (s.t ? 1 : 0);
        if (lane == 0) {
            score_out[(int64_t)t * MZ_ACTIONS + 64] = s64.d;
            kind_out[(int64_t)t * MZ_ACTIONS + 64] = r64.nc == NOT_CHILD ? 255 : (s64.t ? 1 : 0);
        }
    }
}

__global__ void __launch_bounds__(64) k_root_stats(Pool P, int32_t *visits, float *root_value, float *wsum, double *prior) {
    const int t = blockIdx.x, lane = threadIdx.x;
    const ChildRec *recs = P.stats + (int64_t)t * P.max_nodes * MZ_ACTIONS;
    const int64_t o = (int64_t)t * MZ_ACTIONS;
    for (int a = lane; a < MZ_ACTIONS; a += 64) {
        const ChildRec r = recs[a];
        const bool ex = r.nc != NOT_CHILD;
        visits[o + a] = ex ? (int32_t)(r.nc & 0xFFFFu) : -1;
        if (wsum) wsum[o + a] = ex ? r.w : 0.0f;
        if (prior) prior[o + a] = ex ? r.p : 0.0;
    }
    if (lane == 0) {
        const mz_tree_meta *m = &P.meta[t];
        root_value[t] = m->root_n ? m->root_w / (float)m->root_n : 0.0f;  // Node.value()
    }
}

// select_action main.py:128-140 + softmax_sample :275-285, one lane per game
__global__ void __launch_bounds__(256) k_select_action(const int32_t *visits, const int32_t *num_moves, int num_moves_all, const double *u,
                                                       uint8_t *action, int64_t n) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= n) return;
    const int32_t *v = visits + g * MZ_ACTIONS;
    const int moves = num_moves ? num_moves[g] : num_moves_all;
    int64_t sum = 0;
    int best = -1, bestv = -1, last = -1;
    for (int a = 0; a < MZ_ACTIONS; ++a) {
        const int x = v[a];
        if (x < 0) continue;
        sum += x;
        last = a;
        if (x > bestv) { bestv = x; best = a; }  // first maximum in children (ascending) order
    }
    int out = best;
    if (moves < 20 && best >= 0) {
        // np.random.choice(actions, p): cdf = cumsum(p); cdf /= cdf[-1]; searchsorted(cdf, u, 'right')
        const double uu = u[g], dsum = (double)sum;
        double c = 0.0;
        for (int a = 0; a < MZ_ACTIONS; ++a)
            if (v[a] >= 0) c += (double)v[a] / dsum;
        const double total = c;
        c = 0.0;
        out = last;
        for (int a = 0; a < MZ_ACTIONS; ++a) {
            if (v[a] < 0) continue;
            c += (double)v[a] / dsum;
            if (!(c / total <= uu)) { out = a; break; }
        }
    }
    action[g] = (uint8_t)(out < 0 ? 64 : out);
}

// softmax_sample of the arena (test.py:228-243; superset of main.py:275-285): temperature 0 = first
// maximum, 1 = proportional to N, 0 < t < 1 = proportional to N ** (1/t) (Python int ** float =
// libm pow in double).  When 1/t is a small integer the power is formed by exact multiplications
// (bit-equal to a correctly rounded pow; exact below 2^53); other temperatures use the device pow.
__global__ void __launch_bounds__(256) k_softmax_sample(const int32_t *visits, double temperature, const double *u, uint8_t *action,
                                                        int64_t n) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= n) return;
    const int32_t *v = visits + g * MZ_ACTIONS;
    int best = -1, bestv = -1, last = -1;
    for (int a = 0; a < MZ_ACTIONS; ++a) {
        const int x = v[a];
        if (x < 0) continue;
        last = a;
        if (x > bestv) { bestv = x; best = a; }
    }
    int out = best;
    if (temperature != 0.0 && best >= 0) {
        const double e = 1.0 / temperature;
        const int ei = (int)e;
        const bool int_pow = temperature != 1.0 && (double)ei == e && ei >= 1 && ei <= 16;
        auto weight = [&](int x) -> double {
            const double d = (double)x;
            if (temperature == 1.0) return d;
            if (!int_pow) return pow(d, e);
            double r = 1.0, b = d;
            for (int k = ei; k; k >>= 1) { if (k & 1) r *= b; b *= b; }
            return r;
        };
        double wsum = 0.0;
        for (int a = 0; a < MZ_ACTIONS; ++a)
            if (v[a] >= 0) wsum += weight(v[a]);
        double total = 0.0;
        for (int a = 0; a < MZ_ACTIONS; ++a)
            if (v[a] >= 0) total += weight(v[a]) / wsum;
        // np.random.choice(actions, p): cdf = cumsum(p); cdf /= cdf[-1]; searchsorted(cdf, u, 'right')
        const double uu = u[g];
        double c = 0.0;
        out = last;
        for (int a = 0; a < MZ_ACTIONS; ++a) {
            if (v[a] < 0) continue;
            c += weight(v[a]) / wsum;
            if (!(c / total <= uu)) { out = a; break; }
        }
    }
    action[g] = (uint8_t)(out < 0 ? 64 : out);
}

}  // namespace

// ------------------------------------------------------------------------------------------------
static int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

extern "C" int mz_pool_layout_compute(int32_t n_trees, int32_t max_nodes, int32_t hidden_dtype, mz_pool_layout *out) {
    MZ_REQUIRE(out, "mz_pool_layout_compute: null out");
    MZ_REQUIRE(n_trees >= 1, "mz_pool_layout_compute: n_trees < 1");
    MZ_REQUIRE(max_nodes >= 2 && max_nodes <= 65534, "mz_pool_layout_compute: max_nodes %d outside [2, 65534]", max_nodes);
    MZ_REQUIRE(hidden_dtype == MZ_F32 || hidden_dtype == MZ_BF16 || hidden_dtype == MZ_F16, "mz_pool_layout_compute: bad dtype");
    static_assert(sizeof(mz_tree_meta) == 64, "mz_tree_meta must be 64 bytes");
    static_assert(sizeof(ChildRec) == 16 && sizeof(PathRec) == 4 && sizeof(PbcRec) == 16, "record sizes");
    mz_pool_layout L;
    L.n_trees = n_trees;
    L.max_nodes = max_nodes;
    L.hidden_dtype = hidden_dtype;
    L.hidden_elem_bytes = hidden_dtype == MZ_F32 ? 4 : 2;
    const int64_t nodes = (int64_t)n_trees * max_nodes;
    int64_t off = 0;
    L.off_stats = off;  off = align_up(off + nodes * MZ_ACTIONS * 16, 256);
    L.off_hidden = off; off = align_up(off + nodes * MZ_HIDDEN_ELEMS * L.hidden_elem_bytes, 256);
    L.off_meta = off;   off = align_up(off + (int64_t)n_trees * 64, 256);
    L.off_path = off;   off = align_up(off + nodes * 4, 256);
    L.off_pbc = off;    off = align_up(off + (int64_t)(max_nodes + 1) * 16, 256);
    L.total_bytes = off;
    *out = L;
    return MZ_OK;
}

static int check_pool(const char *who, const void *pool, const mz_pool_layout *L, int32_t n_trees) {
    MZ_REQUIRE(pool && L, "%s: null pool/layout", who);
    MZ_REQUIRE(((uintptr_t)pool & 255) == 0, "%s: pool base must be 256-byte aligned", who);
    MZ_REQUIRE(n_trees >= 0 && n_trees <= L->n_trees, "%s: n_trees %d exceeds the pool's %d", who, n_trees, L->n_trees);
    return MZ_OK;
}

extern "C" int mz_pool_init(void *pool, const mz_pool_layout *L, const mz_search_cfg *cfg, void *stream) {
    if (int rc = check_pool("mz_pool_init", pool, L, 0)) return rc;
    MZ_REQUIRE(cfg, "mz_pool_init: null cfg");
    MZ_REQUIRE(cfg->pb_c_base > 0, "mz_pool_init: pb_c_base must be > 0");
    // ucb_score main.py:154-156: math.log / math.sqrt are libm's; evaluate them here with the same
    // libm in double so the device never rounds a transcendental differently.
    std::vector<PbcRec> tab((size_t)L->max_nodes + 1);
    for (int n = 0; n <= L->max_nodes; ++n) {
        tab[(size_t)n].a = log(((double)n + cfg->pb_c_base + 1) / cfg->pb_c_base) + cfg->pb_c_init;
        tab[(size_t)n].s = sqrt((double)n);
    }
    hipStream_t s = (hipStream_t)stream;
    MZ_HIP(hipMemcpyAsync((char *)pool + L->off_pbc, tab.data(), tab.size() * sizeof(PbcRec), hipMemcpyHostToDevice, s));
    MZ_HIP(hipStreamSynchronize(s));  // tab is a host temporary
    MZ_HIP(hipMemsetAsync((char *)pool + L->off_meta, 0, (size_t)L->n_trees * 64, s));
    return MZ_OK;
}

extern "C" int mz_mcts_root_init(void *pool, const mz_pool_layout *L, const mz_search_cfg *cfg, const int8_t *to_play,
                                 const uint64_t *legal, const uint8_t *pass_ok, const float *policy, const double *noise,
                                 const void *hidden, int32_t n_trees, void *stream) {
    if (int rc = check_pool("mz_mcts_root_init", pool, L, n_trees)) return rc;
    if (n_trees == 0) return MZ_OK;
    MZ_REQUIRE(cfg && to_play && legal && policy, "mz_mcts_root_init: null pointer");   // hidden null: already in the roots' pool slots
    hipLaunchKernelGGL(k_root_init, dim3(n_trees), dim3(64), 0, (hipStream_t)stream, make_pool(pool, L), *cfg, to_play, legal, pass_ok,
                       policy, noise, (const char *)hidden);
    MZ_LAUNCH_CHECK("k_root_init");
    return MZ_OK;
}

extern "C" int mz_mcts_select(void *pool, const mz_pool_layout *L, const mz_search_cfg *cfg, void *net_in, int32_t n_trees,
                              void *stream) {
    if (int rc = check_pool("mz_mcts_select", pool, L, n_trees)) return rc;
    if (n_trees == 0) return MZ_OK;
    MZ_REQUIRE(cfg && net_in, "mz_mcts_select: null pointer");
    hipLaunchKernelGGL(k_select, dim3(n_trees), dim3(64), 0, (hipStream_t)stream, make_pool(pool, L), *cfg, LeafOut{(char *)net_in, nullptr, nullptr, nullptr});
    MZ_LAUNCH_CHECK("k_select");
    return MZ_OK;
}

static int check_route(const char *who, const mz_pool_layout *L, const mz_leaf_route *r) {
    MZ_REQUIRE(r && r->src_slot && r->dst_slot && r->act_planes, "%s: null pointer in mz_leaf_route", who);
    MZ_REQUIRE((int64_t)L->n_trees * L->max_nodes < (int64_t)1 << 31, "%s: pool slots do not fit int32", who);
    return MZ_OK;
}

extern "C" int mz_mcts_select_routed(void *pool, const mz_pool_layout *L, const mz_search_cfg *cfg, const mz_leaf_route *route,
                                     int32_t n_trees, void *stream) {
    if (int rc = check_pool("mz_mcts_select_routed", pool, L, n_trees)) return rc;
    if (n_trees == 0) return MZ_OK;
    MZ_REQUIRE(cfg, "mz_mcts_select_routed: null pointer");
    if (int rc = check_route("mz_mcts_select_routed", L, route)) return rc;
    hipLaunchKernelGGL(k_select, dim3(n_trees), dim3(64), 0, (hipStream_t)stream, make_pool(pool, L), *cfg,
                       LeafOut{nullptr, route->src_slot, route->dst_slot, (char *)route->act_planes});
    MZ_LAUNCH_CHECK("k_select");
    return MZ_OK;
}

extern "C" int mz_mcts_backup_select_routed(void *pool, const mz_pool_layout *L, const mz_search_cfg *cfg, const float *value,
                                            const float *policy, const mz_leaf_route *route, int32_t n_trees, void *stream) {
    if (int rc = check_pool("mz_mcts_backup_select_routed", pool, L, n_trees)) return rc;
    if (n_trees == 0) return MZ_OK;
    MZ_REQUIRE(cfg && value && policy, "mz_mcts_backup_select_routed: null pointer");
    if (int rc = check_route("mz_mcts_backup_select_routed", L, route)) return rc;
    hipLaunchKernelGGL(k_backup_select, dim3(n_trees), dim3(64), 0, (hipStream_t)stream, make_pool(pool, L), *cfg, value, policy,
                       (const char *)nullptr, LeafOut{nullptr, route->src_slot, route->dst_slot, (char *)route->act_planes});
    MZ_LAUNCH_CHECK("k_backup_select");
    return MZ_OK;
}

extern "C" int mz_mcts_expand_backup(void *pool, const mz_pool_layout *L, const mz_search_cfg *cfg, const float *value,
                                     const float *policy, const void *hidden, int32_t n_trees, void *stream) {
    if (int rc = check_pool("mz_mcts_expand_backup", pool, L, n_trees)) return rc;
    if (n_trees == 0) return MZ_OK;
    MZ_REQUIRE(cfg && value && policy, "mz_mcts_expand_backup: null pointer");   // hidden null: already in the new nodes' pool slots
    hipLaunchKernelGGL(k_expand_backup, dim3(n_trees), dim3(64), 0, (hipStream_t)stream, make_pool(pool, L), *cfg, value, policy,
                       (const char *)hidden);
    MZ_LAUNCH_CHECK("k_expand_backup");
    return MZ_OK;
}

extern "C" int mz_mcts_backup_select(void *pool, const mz_pool_layout *L, const mz_search_cfg *cfg, const float *value,
                                     const float *policy, const void *hidden, void *net_in, int32_t n_trees, void *stream) {
    if (int rc = check_pool("mz_mcts_backup_select", pool, L, n_trees)) return rc;
    if (n_trees == 0) return MZ_OK;
    MZ_REQUIRE(cfg && value && policy && hidden && net_in, "mz_mcts_backup_select: null pointer");
    hipLaunchKernelGGL(k_backup_select, dim3(n_trees), dim3(64), 0, (hipStream_t)stream, make_pool(pool, L), *cfg, value, policy,
                       (const char *)hidden, LeafOut{(char *)net_in, nullptr, nullptr, nullptr});
    MZ_LAUNCH_CHECK("k_backup_select");
    return MZ_OK;
}

extern "C" int mz_mcts_root_stats(const void *pool, const mz_pool_layout *L, int32_t *visits, float *root_value, float *wsum,
                                  double *prior, int32_t n_trees, void *stream) {
    if (int rc = check_pool("mz_mcts_root_stats", pool, L, n_trees)) return rc;
    if (n_trees == 0) return MZ_OK;
    MZ_REQUIRE(visits && root_value, "mz_mcts_root_stats: null pointer");
    hipLaunchKernelGGL(k_root_stats, dim3(n_trees), dim3(64), 0, (hipStream_t)stream, make_pool(pool, L), visits, root_value, wsum, prior);
    MZ_LAUNCH_CHECK("k_root_stats");
    return MZ_OK;
}

extern "C" int mz_select_action(const int32_t *visits, const int32_t *num_moves, int32_t num_moves_all, const double *u,
                                uint8_t *action, int64_t n, void *stream) {
    MZ_REQUIRE(n >= 0, "mz_select_action: n < 0");
    if (n == 0) return MZ_OK;
    MZ_REQUIRE(visits && u && action, "mz_select_action: null pointer");
    hipLaunchKernelGGL(k_select_action, dim3(mz_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, visits, num_moves, (int)num_moves_all,
                       u, action, n);
    MZ_LAUNCH_CHECK("k_select_action");
    return MZ_OK;
}

extern "C" int mz_mcts_expand(void *pool, const mz_pool_layout *L, const float *policy, const void *hidden, int32_t n_trees,
                              void *stream) {
    if (int rc = check_pool("mz_mcts_expand", pool, L, n_trees)) return rc;
    if (n_trees == 0) return MZ_OK;
    MZ_REQUIRE(policy && hidden, "mz_mcts_expand: null pointer");
    hipLaunchKernelGGL(k_expand, dim3(n_trees), dim3(64), 0, (hipStream_t)stream, make_pool(pool, L), policy, (const char *)hidden);
    MZ_LAUNCH_CHECK("k_expand");
    return MZ_OK;
}

extern "C" int mz_mcts_backup(void *pool, const mz_pool_layout *L, const mz_search_cfg *cfg, const float *value, int32_t n_trees,
                              void *stream) {
    if (int rc = check_pool("mz_mcts_backup", pool, L, n_trees)) return rc;
    if (n_trees == 0) return MZ_OK;
    MZ_REQUIRE(cfg && value, "mz_mcts_backup: null pointer");
    hipLaunchKernelGGL(k_backup, dim3(n_trees), dim3(64), 0, (hipStream_t)stream, make_pool(pool, L), *cfg, value);
    MZ_LAUNCH_CHECK("k_backup");
    return MZ_OK;
}

extern "C" int mz_mcts_node_scores(const void *pool, const mz_pool_layout *L, const mz_search_cfg *cfg, const int32_t *node_id,
                                   int32_t *action, double *score, uint8_t *kind, int32_t n_trees, void *stream) {
    if (int rc = check_pool("mz_mcts_node_scores", pool, L, n_trees)) return rc;
    if (n_trees == 0) return MZ_OK;
    MZ_REQUIRE(cfg && node_id && action, "mz_mcts_node_scores: null pointer");
    MZ_REQUIRE((score == nullptr) == (kind == nullptr), "mz_mcts_node_scores: score and kind go together");
    hipLaunchKernelGGL(k_node_scores, dim3(n_trees), dim3(64), 0, (hipStream_t)stream, make_pool(pool, L), *cfg, node_id, action, score,
                       kind);
    MZ_LAUNCH_CHECK("k_node_scores");
    return MZ_OK;
}

extern "C" int mz_softmax_sample(const int32_t *visits, double temperature, const double *u, uint8_t *action, int64_t n, void *stream) {
    MZ_REQUIRE(n >= 0, "mz_softmax_sample: n < 0");
    MZ_REQUIRE(temperature >= 0.0 && temperature <= 1.0, "mz_softmax_sample: temperature %g outside [0, 1] (test.py:242-243 raises)", temperature);
    if (n == 0) return MZ_OK;
    MZ_REQUIRE(visits && action && (u || temperature == 0.0), "mz_softmax_sample: null pointer");
    hipLaunchKernelGGL(k_softmax_sample, dim3(mz_blocks(n, 256)), dim3(256), 0, (hipStream_t)stream, visits, temperature, u, action, n);
    MZ_LAUNCH_CHECK("k_softmax_sample");
    return MZ_OK;
}
