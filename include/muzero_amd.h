/*
 * muzero_amd.h -- C ABI of libmuzero_amd.so: batched reversi env + batched MCTS for MI355X (gfx950).
 *
 * The reference (ZiyuanMa/MuZero) is pure Python and has no FFI; the boundary this library sits
 * behind is the set of Python callables main.py's self-play loop uses.  Every entry
 * point below names the reference code it replaces (file:line in /root/reference).  The Python
 * binding (muzero_amd/_lib.py, ctypes) and the reference-shaped shims (muzero_amd/compat.py) sit
 * above this header; INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every data pointer is a DEVICE pointer (e.g. torch.Tensor.data_ptr()); the caller owns all
 *     memory; the library never allocates or frees device memory (safe under hipGraph capture);
 *   - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream); calls are
 *     asynchronous on it, except mz_pool_init which also performs one small synchronous H2D copy;
 *   - return 0 on success, <0 on error (MZ_E_*), never throws; mz_last_error() gives the text;
 *   - per-game problems (illegal action, finished game) are reported in status bytes, not errors;
 *   - thread-compatible: no hidden global state besides the thread-local error string.
 *
 * Data model
 *   board    : two uint64 bitboards per game, bit i <=> Action.index i = row*8+col
 *              (reversi.py:19-30); `white` = Player.WHITE(+1) discs, `black` = Player.BLACK(-1)
 *   to_play  : int8, +1 WHITE / -1 BLACK (reversi.py:14-16); BLACK moves first (reversi.py:54)
 *   steps    : uint8, Environment.steps (passes count, reversi.py:133)
 *   done     : uint8, Environment.done (set only by a placing move made at steps==60, :170-172)
 *   legal    : uint64 mask of legal placements = Environment.legal_actions; 0 <=> [Action(64)]
 *   action   : uint8 in [0,64], 64 = pass
 */
#ifndef MUZERO_AMD_H
#define MUZERO_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MZ_ACTIONS 65       /* config.py:10 action_space_size */
#define MZ_HIDDEN_ELEMS 512 /* network.py:66-69 hidden state (8,8,8) */

#define MZ_OK 0
#define MZ_E_ARG (-1)  /* null pointer / bad size / bad enum */
#define MZ_E_HIP (-2)  /* a HIP runtime call or launch failed */

/* per-game status bytes written by mz_env_step */
#define MZ_ST_OK 0
#define MZ_ST_ILLEGAL 1 /* reference would assert / raise: reversi.py:26,143-144,164-165 */
#define MZ_ST_DONE 2    /* game already finished; state left untouched (batch convention) */

/* element type of hidden states / network input buffers */
#define MZ_F32 0
#define MZ_BF16 1
#define MZ_F16 2

const char *mz_last_error(void);
int mz_version(void);

/* ------------------------------------------------------------------------------------------
 * Environment (reversi.py:39-205)
 * ---------------------------------------------------------------------------------------- */

/* Environment.__init__ (reversi.py:41-58) for n games. */
int mz_env_reset(uint64_t *black, uint64_t *white, int8_t *to_play, uint8_t *steps, uint8_t *done,
                 uint64_t *legal, int64_t n, void *stream);

/* Environment.update_legal_actions / check_position / check_direction (reversi.py:81-126) for the
 * side `to_play` on arbitrary positions: Kogge-Stone directional fills on bitboards. */
int mz_env_legal_mask(const uint64_t *black, const uint64_t *white, const int8_t *to_play,
                      uint64_t *legal, int64_t n, void *stream);

/* Environment.step (reversi.py:130-180) for n games in place: flips, pass, step counter, terminal
 * rule and check_win (:197-205), player switch and legal refresh -- including the quirks: passes
 * consume the 60-step budget and are accepted without a legality check; after the terminal move
 * to_play and legal are left stale.  reward: int8 (+1/0/-1), status: MZ_ST_*. */
int mz_env_step(uint64_t *black, uint64_t *white, int8_t *to_play, uint8_t *steps, uint8_t *done,
                uint64_t *legal, const uint8_t *action, int8_t *reward, uint8_t *status, int64_t n,
                void *stream);

/* Environment.step (reversi.py:130-180) on a packed per-game record: the same rules and quirks as mz_env_step, with the six
 * byte-wide fields in ONE 32-bit word per game, so every stream of the kernel moves 4 or 8 bytes per lane:
 *   meta: bits 0-6 steps | bit 7 WHITE to move (else BLACK) | bit 8 done | bits 9-15 the action to play, 0..64 (in) |
 *         bits 16-17 reward + 1 (out) | bits 18-19 status MZ_ST_* (out)
 * black / white / legal as in mz_env_step; a refused step (MZ_ST_ILLEGAL / MZ_ST_DONE) leaves the state untouched. */
int mz_env_step_packed(uint64_t *black, uint64_t *white, uint64_t *legal, uint32_t *meta, int64_t n, void *stream);

/* BASELINE.json config 2: games [game0, game0+n) played from the opening for n_steps steps with
 * uniformly random legal moves (counter-based RNG keyed (seed, game, step); k-th legal bit in
 * ascending index), state held in registers for the whole game.  voluntary != 0 also
 * plays a pass with probability 1/8 although moves exist (reversi.py:135-139 accepts that).
 * Outputs: final state; digest (nullable) = running 64-bit digest of all n_steps+1 states;
 * actions (nullable) = n x n_steps action log; last_reward (nullable). */
int mz_env_random_playout(uint64_t seed, uint64_t game0, int64_t n, int32_t n_steps, int32_t voluntary,
                          uint64_t *black, uint64_t *white, int8_t *to_play, uint8_t *steps, uint8_t *done,
                          uint64_t *legal, uint64_t *digest, uint8_t *actions, int8_t *last_reward,
                          void *stream);

/* the same RNG, one step at a time on HBM-resident state: action[i] = random legal move of game
 * game0+i at step `step` (64 if none).  Used with mz_env_step by the HBM-resident bench mode. */
int mz_env_random_action(uint64_t seed, uint64_t game0, int32_t step, const uint64_t *legal,
                         uint8_t *action, int64_t n, void *stream);

/* Action.encode (reversi.py:32-36) for n actions -> out (n,2,8,8) of dtype: layer 0 for WHITE else
 * 1; rows r and c of that layer set for index = r*8+c (numpy fancy-index behaviour of the
 * reference), whole layer for pass. */
int mz_action_encode(const uint8_t *action, const int8_t *player, void *out, int32_t dtype, int64_t n,
                     void *stream);

/* Game.make_image (reversi.py:301-312) + Environment.observe (:182-186) for n games.
 * hist_white/hist_black: observation history, step-major [t * hist_stride + game], t = 0..60.
 * state_index: per-game (nullable) else the uniform `state_index_all`.  out: (n,9,8,8) of dtype. */
int mz_make_image(const uint64_t *hist_white, const uint64_t *hist_black, int64_t hist_stride,
                  const int32_t *state_index, int32_t state_index_all, void *out, int32_t dtype, int64_t n,
                  void *stream);

/* The arena Controller's observation (test.py:16-29) for n games: planes 6-7 = the current state
 * (WHITE, BLACK), 4-5 the one before ... over a sliding window of the last 4 observed states that,
 * unlike make_image, includes the opening position; plane 8 = 1 iff BLACK is to move (= state_index
 * even for games played from the opening).  Arguments as mz_make_image. */
int mz_controller_image(const uint64_t *hist_white, const uint64_t *hist_black, int64_t hist_stride,
                        const int32_t *state_index, int32_t state_index_all, void *out, int32_t dtype, int64_t n,
                        void *stream);

/* ------------------------------------------------------------------------------------------
 * Tree search (main.py:99-196, 275-285; reversi.py:234-251 Node; utilities.py:114-129 MinMaxStats)
 * ---------------------------------------------------------------------------------------- */

/* config.py knobs the search reads */
typedef struct mz_search_cfg {
    double pb_c_base;   /* config.py:22 */
    double pb_c_init;   /* config.py:23 */
    double discount;    /* config.py:15 */
    double noise_frac;  /* config.py:19 root_exploration_fraction */
    double bound_lo;    /* config.py:29 known_bounds[0] */
    double bound_hi;    /* config.py:29 known_bounds[1] */
    int32_t bounds_none; /* known_bounds = None */
    int32_t tie_lowest;  /* 0 = reference: max over (score, action) tuples, ties -> HIGHEST index
                            (main.py:145-147); 1 = lowest index (BASELINE.json wording) */
    int32_t negamax;     /* 0 = reference: ucb_score adds normalize(child.value()) (main.py:159), the value from the
                            CHILD's mover's point of view -- the two-player sign slip of the MuZero pseudocode: the search
                            prefers moves that are good for the opponent.  1 = normalize(-child.value()) (non-default) */
    int32_t device_noise; /* 1: where a root expansion is given noise == NULL, draw np.random.dirichlet([noise_alpha] * n_children)
                             (main.py:192-193) on the device from a counter-based stream keyed (noise_seed, tree, searches this
                             tree slot has started); 0: NULL noise means no exploration noise (test.py's arena) */
    double noise_alpha;   /* config.py:18 root_dirichlet_alpha */
    uint64_t noise_seed;
} mz_search_cfg;

/* Node pool for n_trees independent searches of at most max_nodes-1 simulations each, laid out by
 * the library inside ONE caller-owned device buffer of total_bytes (256-B aligned base):
 *   stats  [tree][node][65] x 16 B  { uint32 n_child; float W; double P }   (Node, reversi.py:236-243)
 *            n_child = visit_count | (expanded-node id + 1) << 16 ; 0xFFFFFFFF = not a child
 *            W = value_sum (fp32, accumulated in simulation order); P = prior (float64)
 *   hidden [tree][node][512] x elem  Node.hidden_state of expanded nodes
 *   meta   [tree] x 64 B            root N/W/player, MinMaxStats, node count, path length, leaf info
 *   path   [tree][max_nodes] x 4 B  search_path of the current simulation {uint16 node; uint8 action}
 *   pbc    [max_nodes+1] x 16 B     host-computed {log((N+base+1)/base)+init, sqrt(N)} (glibc, as
 *                                   CPython's math.log/math.sqrt: main.py:154-156)
 */
typedef struct mz_pool_layout {
    int64_t total_bytes;
    int64_t off_stats, off_hidden, off_meta, off_path, off_pbc;
    int32_t n_trees, max_nodes, hidden_dtype, hidden_elem_bytes;
} mz_pool_layout;

/* tree meta record (64 B); offsets are part of the ABI so tests can inspect trees */
typedef struct mz_tree_meta {
    int32_t n_nodes;     /* expanded nodes (root = node 0) */
    int32_t root_n;      /* root.visit_count */
    float root_w;        /* root.value_sum */
    int32_t root_player; /* +1 / -1 */
    double mm_min;       /* MinMaxStats.minimum (value; fp32-exact when flagged tensor) */
    double mm_max;       /* MinMaxStats.maximum */
    int32_t mm_flags;    /* bit0: minimum is an fp32 tensor, bit1: maximum is (else Python float) */
    int32_t path_len;    /* edges of the current simulation's path */
    int32_t leaf_player; /* player of the leaf of the current simulation */
    int32_t error;       /* bit0: pool full; bit1: select without root; bit2 (MZ_TREE_ERR_POLICY_MASS): an expansion's policy mass over
                            the expanded actions was 0 (the reference raises ZeroDivisionError, main.py:171-175) or not finite */
    int32_t leaf_parent; /* expanded-node id whose hidden state feeds recurrent_inference */
    int32_t leaf_action; /* history[-1].index */
    int32_t pad[2];      /* pad[0]: searches this tree slot has started (mz_mcts_root_init counts; keys the device noise stream) */
} mz_tree_meta;

#define MZ_TREE_ERR_POLICY_MASS 4

/* host-only: compute the layout (no device access). max_nodes in [2, 65534]. */
int mz_pool_layout_compute(int32_t n_trees, int32_t max_nodes, int32_t hidden_dtype, mz_pool_layout *out);

/* upload the pb_c table for cfg and clear meta.  Re-run when cfg.pb_c_* change. */
int mz_pool_init(void *pool, const mz_pool_layout *layout, const mz_search_cfg *cfg, void *stream);

/* play_game's root set-up (main.py:75-83): root = Node(0, to_play); expand_node(root,
 * legal_actions, initial_inference output) over the legal placements in ascending index (pass iff
 * legal == 0); add_exploration_noise with caller-supplied Dirichlet noise indexed by ACTION
 * (noise nullable = none); MinMaxStats(known_bounds) (main.py:100).
 * pass_ok (nullable): per-tree override of "pass is a root child"; null = the game rule above
 * (expand_node itself accepts any action list, main.py:165-175).
 * policy: (n,65) f32 rows; hidden: (n,512) of the pool's dtype; noise: (n,65) f64. */
int mz_mcts_root_init(void *pool, const mz_pool_layout *layout, const mz_search_cfg *cfg,
                      const int8_t *to_play, const uint64_t *legal, const uint8_t *pass_ok, const float *policy,
                      const double *noise, const void *hidden, int32_t n_trees, void *stream);

/* one simulation's descent for every tree (run_mcts main.py:103-118 = select_child :144-148 +
 * ucb_score :153-160 per level), one wavefront per tree.  Writes the path, and the network input
 * for recurrent_inference: net_in (n,10,8,8) of the pool's dtype = parent hidden state (8 planes)
 * followed by Action.encode of the last action tagged with the CHILD's player (main.py:110). */
int mz_mcts_select(void *pool, const mz_pool_layout *layout, const mz_search_cfg *cfg, void *net_in,
                   int32_t n_trees, void *stream);

/* expand_node over all 65 actions (main.py:123,165-175) + backpropagate (:180-186) along the path
 * stored by mz_mcts_select.  value: (n) f32, policy: (n,65) f32, hidden: (n,512) pool dtype, or NULL when the new
 * nodes' hidden states are in their pool slots already (routed inference, below). */
int mz_mcts_expand_backup(void *pool, const mz_pool_layout *layout, const mz_search_cfg *cfg,
                          const float *value, const float *policy, const void *hidden, int32_t n_trees,
                          void *stream);

/* fused: mz_mcts_expand_backup of simulation s followed by mz_mcts_select of simulation s+1 in one
 * launch (the tree's records are still in registers / L2). */
int mz_mcts_backup_select(void *pool, const mz_pool_layout *layout, const mz_search_cfg *cfg,
                          const float *value, const float *policy, const void *hidden, void *net_in,
                          int32_t n_trees, void *stream);

/* The ROUTED form of the same calls: no hidden state is copied by the tree kernels.  run_mcts keeps every node's
 * hidden_state in the node (main.py:114-123: parent.hidden_state in, node.hidden_state out); here the inference launch
 * itself reads the parent's state from its pool slot and writes the new node's state into ITS slot
 * (mz_inference_forward_routed), and a descent only says which slots: slot = tree * max_nodes + node.  The new node's id is
 * known before the expansion (ids are handed out in order).  Only the two Action.encode planes travel through a row.
 * With routing, mz_mcts_root_init / mz_mcts_expand_backup take hidden = NULL ("already in place"). */
typedef struct mz_leaf_route {
    int32_t *src_slot;  /* [n] out: pool slot of the leaf's parent (the recurrent inference's input planes 0..7) */
    int32_t *dst_slot;  /* [n] out: pool slot of the node the next expansion creates (the inference's hidden output) */
    void *act_planes;   /* (n,2,8,8) out, pool hidden dtype: Action(a, child's player).encode() (main.py:110, reversi.py:32-36) */
} mz_leaf_route;
int mz_mcts_select_routed(void *pool, const mz_pool_layout *layout, const mz_search_cfg *cfg, const mz_leaf_route *route,
                          int32_t n_trees, void *stream);
int mz_mcts_backup_select_routed(void *pool, const mz_pool_layout *layout, const mz_search_cfg *cfg, const float *value,
                                 const float *policy, const mz_leaf_route *route, int32_t n_trees, void *stream);

/* the two halves of mz_mcts_expand_backup as separate calls, for callers that keep the reference's
 * call structure: expand_node (main.py:165-175) of the leaf reached by the last mz_mcts_select,
 * and backpropagate (main.py:180-186) of `value` along that path. */
int mz_mcts_expand(void *pool, const mz_pool_layout *layout, const float *policy, const void *hidden,
                   int32_t n_trees, void *stream);
int mz_mcts_backup(void *pool, const mz_pool_layout *layout, const mz_search_cfg *cfg, const float *value,
                   int32_t n_trees, void *stream);

/* select_child (main.py:144-148) and ucb_score (main.py:153-160) of ONE expanded node per tree:
 * action[t] = selected action of node node_id[t] (-1 if that node is not expanded);
 * score (n,65) f64 / kind (n,65) u8 nullable together: every child's ucb score and whether the
 * reference would hold it as a Python float (0) or an fp32 tensor (1); 255 = not a child. */
int mz_mcts_node_scores(const void *pool, const mz_pool_layout *layout, const mz_search_cfg *cfg,
                        const int32_t *node_id, int32_t *action, double *score, uint8_t *kind,
                        int32_t n_trees, void *stream);

/* root statistics after the search: visits (n,65) i32 (-1 = not a child) as consumed by
 * Game.store_search_statistics (reversi.py:292-299) and select_action (main.py:128-140);
 * root_value (n) f32 = root.value(); wsum (n,65) f32 / prior (n,65) f64 nullable (tests). */
int mz_mcts_root_stats(const void *pool, const mz_pool_layout *layout, int32_t *visits, float *root_value,
                       float *wsum, double *prior, int32_t n_trees, void *stream);

/* select_action + softmax_sample (main.py:128-140, 275-285): temperature 1 while num_moves < 20
 * (np.random.choice == cdf search with the caller's uniform u), else first maximum in ascending
 * action order.  num_moves: per-game (nullable) else num_moves_all. */
int mz_select_action(const int32_t *visits, const int32_t *num_moves, int32_t num_moves_all, const double *u,
                     uint8_t *action, int64_t n, void *stream);

/* softmax_sample of the arena (test.py:228-243): temperature 0 = first maximum, 1 = proportional
 * to N, 0 < t < 1 = proportional to N ** (1/t) (the arena plays t = 0.25, test.py:174); u: (n) f64
 * uniforms standing for np.random.choice's draw (nullable when temperature == 0). */
int mz_softmax_sample(const int32_t *visits, double temperature, const double *u, uint8_t *action, int64_t n,
                      void *stream);

/* ------------------------------------------------------------------------------------------
 * Replay buffer + training targets (utilities.py:46-109 ReplayBuffer; reversi.py:314-341 make_target)
 * ---------------------------------------------------------------------------------------- */

/* A window of `window` finished games, game-major, in caller-owned device arrays (host struct of
 * device pointers).  Slot order is the caller's business (the Python ReplayBuffer keeps the FIFO of
 * utilities.py:56-63).  The mover of step t is BLACK for even t: Environment switches the player on
 * every step (reversi.py:137,176) and BLACK starts (reversi.py:54). */
typedef struct mz_replay {
    uint64_t *white;    /* [window][max_moves+1] Game.state_history, plane 0 (WHITE) as bitboards */
    uint64_t *black;    /* [window][max_moves+1] plane 1 (BLACK) */
    uint8_t *actions;   /* [window][max_moves]   Game.action_history indices */
    int8_t *rewards;    /* [window][max_moves]   Game.rewards */
    float *policy;      /* [window][max_moves][65] Game.child_visits = N_a / sum N computed in double
                           (reversi.py:292-298), rounded once to fp32 (torch.FloatTensor, main.py:236) */
    float *root_values; /* [window][max_moves]   Game.root_values */
    int32_t *length;    /* [window]              len(action_history) */
    int32_t window, max_moves;
    int32_t *refused;   /* optional, may be NULL: device counter that mz_replay_sample_batch increments once per sample it
                           refuses (slot or position outside the window / the game); the caller zeroes and reads it */
} mz_replay;

/* element type of the root visit counts in a trajectory record */
#define MZ_VIS_I32 0 /* int32, -1 = not a root child (what mz_mcts_root_stats writes) */
#define MZ_VIS_U16 1 /* uint16, 0 = unvisited or not a root child */
#define MZ_VIS_U8 2  /* uint8,  0 = unvisited or not a root child (searches of at most 255 simulations) */

/* ReplayBuffer.save_game (utilities.py:56-63) for n games at once: game i of a step-major batch of
 * trajectories (the layout the self-play engine records: hist_* [t * stride + i], t = 0..max_moves;
 * actions/rewards/root_values [t * stride + i]; visits [(t * stride + i) * 65 + a] of visits_dtype: a child that was
 * never visited and an action that is no root child both normalise to 0, so the narrow types lose nothing here)
 * is written to slot (slot0 + i) mod window.  length: per-game (nullable) else length_all (values are clamped to
 * [0, max_moves]).  Game.store_search_statistics' normalisation (reversi.py:292-299) happens here. */
int mz_replay_save(const mz_replay *rb, int32_t slot0, const uint64_t *hist_white, const uint64_t *hist_black,
                   const uint8_t *actions, const int8_t *rewards, const void *visits, int32_t visits_dtype,
                   const float *root_values, const int32_t *length, int32_t length_all, int64_t stride, int32_t n,
                   void *stream);

/* ReplayBuffer.sample_batch (utilities.py:72-99) for `batch` already-drawn (slot, position) pairs
 * (the draws stay with the caller: random.sample / random.randint, utilities.py:76,104), K =
 * num_unroll_steps, position + K <= length.  One wavefront per sample.  Outputs, all fp32:
 *   image           (batch,9,8,8)    Game.make_image(position)                   utilities.py:88
 *   actions         (K,batch,2,8,8)  Action.encode of the K real moves           utilities.py:89,95
 *   target_values   (K+1,batch)      Game.make_target values                     reversi.py:331-338
 *   target_policies (K+1,batch,65)   Game.make_target policies (zeros past end)  reversi.py:335,339
 * A pair outside its game (slot outside the window, position < 0 or position + K > length) is a caller bug: the sample's rows
 * are written as zeros -- never left uninitialised -- and rb->refused, when given, counts it. */
int mz_replay_sample_batch(const mz_replay *rb, const int32_t *slot, const int32_t *pos, int32_t batch, int32_t K,
                           float *image, float *actions, float *target_values, float *target_policies, void *stream);

/* ------------------------------------------------------------------------------------------
 * Inference epilogue (network.py:35-47, 54-68, 118-128: what surrounds every nn.Conv2d)
 * ---------------------------------------------------------------------------------------- */

/* In place over a channels-last activation y (n_elems elements, `channels` innermost):
 *   y = act(y + bias[c] (+ residual)),  act = ReLU if relu != 0
 * i.e. the conv bias, the ResidualBlock skip add (network.py:46-47) and nn.ReLU in one pass instead
 * of PyTorch's three.  bias / residual nullable; fp32 arithmetic, one rounding to dtype. */
int mz_bias_act_nhwc(void *y, const void *bias, const void *residual, int64_t n_elems, int32_t channels,
                     int32_t dtype, int32_t relu, void *stream);

/* ------------------------------------------------------------------------------------------
 * Convolution tower (network.py:49-77 Representation, :113-135 Dynamics) as one LDS-resident MFMA
 * kernel: conv3x3 cin->128 + ReLU, three ResidualBlocks (network.py:35-47), conv3x3 128->8 + ReLU.
 * bf16 operands, fp32 accumulation, one bf16 rounding per layer output.  Activations never leave
 * LDS between layers (tower_kernels.hip: eight boards per workgroup, one activation image updated in place).
 * ---------------------------------------------------------------------------------------- */
typedef struct mz_tower_weights {   /* device pointers to PREPACKED weights (muzero_amd.network.pack_tower) */
    const void *w_first;   /* bf16 [9 taps][128 co][16 ci]      conv1 / dynamics conv.0, ci zero-padded; tap = ky*3+kx */
    const float *b_first;  /* f32 [128] or NULL (Dynamics' first conv has no bias, network.py:119) */
    const void *w_mid;     /* bf16 [6 layers][36 stages][128 co][4 slots][8]: stage = tap*4 + ci/32,
                              16-byte chunk c of the 32 input channels stored in slot c ^ {0,2,3,1}[(co >> 2) & 3];
                              followed by 4 stages (32 KB) of padding: the weight stream prefetches 4 stages ahead */
    const float *b_mid;    /* f32 [6][128] */
    const void *w_last;    /* bf16 [4 quarters of ci][80 rows = tap * 8 + plane (rows 72..79 zero)][4 slots][8]: chunk swizzle as w_mid;
                              20,480 bytes, read as the weight stream's last 2.5 stages */
    const float *b_last;   /* f32 [8] or NULL (network.py:127) */
    int32_t cin;           /* input planes: 9 (Representation) or 10 (Dynamics: hidden ++ action planes) */
    int64_t *debug_stamps; /* diagnostics, normally NULL: device array [ceil(n/8)][24] receiving s_memtime stamps at the
                              kernel's phase boundaries (0 start, 1 first layer, 2 weight prologue, 3+2l / 4+2l K loop /
                              epilogue of mid layer l, 15 last layer, 16 heads) */
} mz_tower_weights;

/* x: (n,cin,8,8) bf16 NCHW (e.g. the net_in rows mz_mcts_select writes); hidden: (n,8,8,8) NCHW of
 * hidden_dtype (MZ_BF16 or MZ_F32).  Persistent grid, one workgroup per CU walking 8-board tiles.  All weight pointers
 * 16-byte aligned. */
int mz_tower_forward(const mz_tower_weights *w, const void *x, void *hidden, int32_t hidden_dtype, int64_t n, void *stream);

/* Prediction heads (network.py:79-111) fused behind the tower: 1x1 conv 8->2 + ReLU + Flatten +
 * Linear(128,128) + ReLU + Linear(128,65) + Softmax (policy); 1x1 conv 8->2 + ReLU + Flatten +
 * Linear(128,1) + Tanh (value).  The two Linear matrices are bf16 in nn.Linear's own [out][in] layout (the serving
 * network's weights are bf16 values: nothing is lost) and run on the matrix cores, sixteen boards per pass, with every fp32
 * input feature split into two bf16 terms (hi + lo: 16 significant bits, exact products, fp32 sums); everything else is f32
 * arithmetic.  Within 2e-5 of the fp32 module on the same hidden state. */
typedef struct mz_heads_weights {
    const float *pc_w, *pc_b;   /* policy_head.0: [2][8], [2] */
    const void *p1_w;           /* policy_head.3 weight: bf16 [128 out][128 in]; 16-byte aligned (fetched as 16-byte MFMA fragments) */
    const float *p1_b;          /* [128] */
    const void *p2_w;           /* policy_head.5 weight: bf16 [65 out][128 in]; 16-byte aligned */
    const float *p2_b;          /* [65] */
    const float *vc_w, *vc_b;   /* value_head.0: [2][8], [2] */
    const float *v1_w, *v1_b;   /* value_head.3: [128], [1] */
} mz_heads_weights;

/* network.initial_inference / recurrent_inference (network.py:146-163) in ONE launch: tower + heads.
 * Outputs the tuple the search consumes: value (n) f32, policy (n,65) f32 softmaxed, hidden (n,8,8,8). */
int mz_inference_forward(const mz_tower_weights *w, const mz_heads_weights *heads, const void *x, void *hidden,
                         int32_t hidden_dtype, float *value, float *policy, int64_t n, void *stream);

/* The same launch with its hidden states gathered from / scattered into a pool of 512-element rows (the search pool's hidden
 * region, pool + layout.off_hidden): recurrent_inference(parent.hidden_state, action) -> node.hidden_state (main.py:114-123)
 * without the two copies around it.
 *   src_pool, src_slot : input planes 0..7 of board b = row src_slot[b] of src_pool (bf16); x then holds only the
 *                        remaining cin - 8 planes, (n, cin - 8, 8, 8) bf16.  src_pool NULL: x holds all cin planes.
 *   dst_pool, dst_slot, dst_stride : board b's hidden output goes to row dst_slot[b] of dst_pool, or, dst_slot NULL, to row
 *                        b * dst_stride (initial inference into every tree's root slot: dst_stride = max_nodes). */
int mz_inference_forward_routed(const mz_tower_weights *w, const mz_heads_weights *heads, const void *x, const void *src_pool,
                                const int32_t *src_slot, void *dst_pool, const int32_t *dst_slot, int64_t dst_stride,
                                int32_t hidden_dtype, float *value, float *policy, int64_t n, void *stream);

/* ------------------------------------------------------------------------------------------
 * One whole search per call: the per-move body of play_game (main.py:75-87) for n_trees games with the
 * HIP inference launch as the network -- initial_inference(image) (network.py:146-153), expand_node over
 * the legal moves + add_exploration_noise (main.py:82-83), n_sims x [select_child descent, recurrent_inference
 * (network.py:155-163), expand_node, backpropagate] (run_mcts, main.py:99-125), then the root statistics
 * store_search_statistics / select_action consume.  Exactly the launches a caller would make through
 * mz_inference_forward / mz_mcts_root_init / mz_mcts_select / mz_mcts_backup_select / mz_mcts_root_stats,
 * issued from C (no per-launch host-language overhead).
 *   repr / dyn : prepacked Representation (cin 9) and Dynamics (cin 10) towers; heads: prediction heads
 *   image      : (n,9,8,8) bf16 NCHW (mz_make_image with MZ_BF16); noise: (n,65) f64 by action, nullable
 *   workspace  : caller-owned device scratch of >= mz_search_workspace_bytes(n_trees, n_sims) bytes, 256-B
 *                aligned: the node pool (bf16 hidden states) and the staging rows between the tree kernels
 *                and the inference launch; contents need not be preserved between calls
 *   visits     : (n,65) i32, -1 = not a root child; root_value: (n) f32
 * Performs mz_pool_init's one small synchronous upload of the pb_c table. */
int64_t mz_search_workspace_bytes(int32_t n_trees, int32_t n_sims);   /* -1 on bad arguments */
int mz_search(const mz_tower_weights *repr, const mz_tower_weights *dyn, const mz_heads_weights *heads,
              const mz_search_cfg *cfg, const void *image, const int8_t *to_play, const uint64_t *legal,
              const double *noise, int32_t n_sims, int32_t n_trees, void *workspace, int64_t workspace_bytes,
              int32_t *visits, float *root_value, void *stream);

/* mz_search for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE (which has no native boundary at all):
 * it exists for code under hipGraph capture, tight serving loops and harnesses that forbid hipMalloc.  Same inputs,
 * outputs and results as mz_search (bit-identical visits and root values), but the workspace is a file-scope device array
 * of MZ_SEARCH_NOALLOC_BYTES inside the library -- every GPU the library runs on holds its own copy until the process
 * ends -- and the call performs no hipMalloc / hipFree / allocating library routine.
 *   limits : mz_search_workspace_bytes(n_trees, n_sims) + 64 KiB <= MZ_SEARCH_NOALLOC_BYTES (e.g. 128 trees x 50
 *            simulations, 64 x 100, 1 x 4095) and n_sims <= MZ_SEARCH_NOALLOC_MAX_SIMS; larger requests are refused with
 *            MZ_E_ARG and a message naming mz_search;
 *   state  : the array keeps the last call's contents; the call rewrites everything it reads.  All calls on one GPU
 *            share that GPU's array: calls that overlap in time on one GPU (two streams, two threads) RACE -- serialise them;
 *   first call on a device, and any call that changes cfg->pb_c_base / pb_c_init, uploads the pb_c table with one small
 *            synchronous copy: make that call before starting a graph capture. */
#define MZ_SEARCH_NOALLOC_BYTES (16 << 20)
#define MZ_SEARCH_NOALLOC_MAX_SIMS 4095
int mz_search_noalloc(const mz_tower_weights *repr, const mz_tower_weights *dyn, const mz_heads_weights *heads,
                      const mz_search_cfg *cfg, const void *image, const int8_t *to_play, const uint64_t *legal,
                      const double *noise, int32_t n_sims, int32_t n_trees, int32_t *visits, float *root_value,
                      void *stream);

/* ------------------------------------------------------------------------------------------
 * Training step: update_weights' loss.backward() (main.py:247-266) for the twelve 3x3, 128 -> 128
 * convolutions of the two towers (network.py:35-47 inside network.py:49-77 and 113-135), on the matrix cores.
 *
 * mz_conv3x3_wgrad: the weight and bias gradient autograd computes for ONE such convolution, summed over up to 8 uses of it
 * (the K unrolled Dynamics calls of main.py:247-252 share their weights):
 *     dw[co][ci][ky][kx] (+)= sum_use sum_b sum_{y,x} g[b][co][y][x] * x[b][ci][y+ky-1][x+kx-1]      (zero outside the board)
 *     db[co]             (+)= sum_use sum_b sum_{y,x} g[b][co][y][x]                                  (db may be NULL)
 *   x[k], g[k] : the k-th use's input and output gradient, (n_boards[k], 128, 8, 8) f32 NCHW contiguous, 16-byte aligned;
 *                x, g, n_boards are HOST arrays of n_seg (1..8) entries
 *   dw         : (128, 128, 3, 3) f32 -- nn.Conv2d.weight's layout; accumulate != 0 adds to what dw / db hold
 *   workspace  : device scratch of >= mz_conv3x3_wgrad_workspace(sum n_boards) bytes, 16-byte aligned (per-slice partial sums)
 * Arithmetic: operands rounded to bf16 (round to nearest even), products and sums in f32 on v_mfma_f32_16x16x32_bf16; the
 * bias gradient is summed from the f32 values.  Sums run in a fixed order: two calls on the same data return the same bits. */
int64_t mz_conv3x3_wgrad_workspace(int64_t n_boards);
int mz_conv3x3_wgrad(const float *const *x, const float *const *g, const int64_t *n_boards, int32_t n_seg, float *dw, float *db,
                     int32_t accumulate, void *workspace, int64_t workspace_bytes, void *stream);

/* The same convolutions' forward pass and data gradient (main.py:247-252 forward, :265-266 backward), one launch per layer:
 *     y[b][co][.] = bias[co] + sum_{ci,ky,kx} W[co][ci][ky][kx] * x[b][ci][. + (ky-1, kx-1)]      (packed_fwd)
 *     gx[b][ci][.] =           sum_{co,ky,kx} W[co][ci][ky][kx] * g[b][co][. - (ky-1, kx-1)]      (packed_bwd, bias NULL)
 * mz_conv3x3_pack: weight (128,128,3,3) f32 -> the two bf16 stage images the kernel streams, MZ_CONV3X3_PACKED_BYTES each
 * (16-byte aligned); repack after every optimiser step.
 * mz_conv3x3_forward: x, y (n,128,8,8) f32 NCHW contiguous, x != y; bias (128) f32 or NULL; relu != 0 clamps y at 0.
 * Arithmetic: x and W rounded to bf16 (nearest even), f32 products and sums (v_mfma_f32_16x16x32_bf16), f32 output. */
#define MZ_CONV3X3_PACKED_BYTES (40 * 4096 * 2)
int mz_conv3x3_pack(const float *w, void *packed_fwd, void *packed_bwd, void *stream);
int mz_conv3x3_forward(const void *packed, const float *bias, const float *x, float *y, int64_t n, int32_t relu, void *stream);
/* mz_conv3x3_forward with the residual block's elementwise work fused into the store phase (all tensors (n,128,8,8) f32 NCHW,
 * skip / mask nullable, mask may not alias y; x_part 0 = x rounded to bf16, 1 / 2 = see split precision below):
 *     y = [relu]( [mask > 0 ?] (conv(x) + bias [+ skip]) : 0 )
 *   forward : t = relu(conv1(h))             -> relu = 1
 *             y = relu(h + conv2(t))         -> skip = h, relu = 1                                   (network.py:35-47)
 *   backward: g_t   = dgrad2(g_pre) * (t > 0)                 -> mask = t
 *             g_pre' = (g_pre + dgrad1(g_t)) * (y_prev > 0)   -> skip = g_pre, mask = y_prev  (the previous block's gate) */
int mz_conv3x3_fused(const void *packed, const float *bias, const float *x, int32_t x_part, const float *skip, const float *mask, float *y,
                     int64_t n, int32_t relu, void *stream);

/* Split precision ("bf16x3"): the same kernels at close to the reference's fp32 arithmetic.  A value v is carried as
 * hi = bf16(v) and lo = bf16(v - hi) (16 mantissa bits together), and a product a * b as a_hi b_hi + a_lo b_hi + a_hi b_lo
 * (relative error ~2^-16 per product; sums in f32 as before).
 *   mz_conv3x3_pack_part(part = 1) packs the weights' lo images; x_part = 1 makes mz_conv3x3_fused convolve the lo part of x.
 *   One convolution = three launches accumulating through skip (which may alias y):
 *       y  = conv(x hi, W hi) + bias [+ residual skip];   y += conv(x lo, W hi);   y = epilogue(y + conv(x hi, W lo))
 *   or ONE launch, x_part = 2: packed then points at a stream [36 stages of the hi image][36 of the lo image][4 of padding] (pack the
 *   hi image to p, then the lo image to p + 36 * 4096 * 2 bytes); x goes into a hi and a lo image in LDS (4- or 2-board tiles), W_hi meets
 *   both in one K loop, W_lo x_hi follows, the epilogue runs once.
 *   mz_conv3x3_wgrad_x3 forms the three products inside one pass over the operands. */
int mz_conv3x3_pack_part(const float *w, void *packed_fwd, void *packed_bwd, int32_t part, void *stream);
/* the same for up to 16 convolutions in ONE launch (a training step repacks all twelve): w, packed_fwd, packed_bwd are HOST arrays of n pointers */
int mz_conv3x3_pack_many(const float *const *w, void *const *packed_fwd, void *const *packed_bwd, int32_t n, int32_t part, void *stream);
int mz_conv3x3_wgrad_x3(const float *const *x, const float *const *g, const int64_t *n_boards, int32_t n_seg, float *dw, float *db,
                        int32_t accumulate, void *workspace, int64_t workspace_bytes, void *stream);

/* Split-precision SERVING: the three residual blocks of a tower (network.py:35-47 as used by Representation.conv2,
 * network.py:59-61, and Dynamics' blocks, network.py:119-123) in ONE launch, LDS-resident:
 *     for blk in 0..2:   x = relu(x + conv_{2 blk + 1}(relu(conv_{2 blk}(x) + b_{2 blk})) + b_{2 blk + 1})
 * in the split precision described above (activations carried as hi / lo bf16 images in LDS, three products per convolution
 * into one f32 accumulator, the skip path in f32 registers): within ~1e-5 of the fp32 modules, no HBM traffic between layers.
 *   mz_restower_x3_pack : w = HOST array of the six convolutions' weights (128,128,3,3) f32 in order; packed = device buffer of
 *                         MZ_RESTOWER_X3_PACKED_BYTES, 16-byte aligned.  Repack when the weights change.
 *   mz_restower_x3_forward : bias (6,128) f32 contiguous, 16-byte aligned; x, y (n,128,8,8) f32 NCHW contiguous, x == y allowed;
 *                         relu_in != 0 applies ReLU to x on load (the first convolution's activation, fused). */
#define MZ_RESTOWER_X3_PACKED_BYTES ((12 * 36 + 4) * 4096 * 2)
int mz_restower_x3_pack(const float *const *w, void *packed, void *stream);
int mz_restower_x3_forward(const void *packed, const float *bias, const float *x, float *y, int64_t n, int32_t relu_in, void *stream);
/* The WHOLE tower in that form (Representation network.py:49-77, Dynamics' convolutions network.py:113-135): first convolution
 * (cin -> 128, + ReLU), the three blocks, last convolution (128 -> 8, + ReLU), one launch, planes in, hidden state out:
 *   mz_tower_x3_pack_ends : w_first (128, cin, 3, 3) f32, w_last (8, 128, 3, 3) f32 -> ends, MZ_TOWER_X3_ENDS_BYTES, 16-byte aligned
 *   mz_tower_x3_forward   : packed / bias as above; b_first (128) / b_last (8) f32 or NULL (Dynamics has none); x (n, cin, 8, 8) f32,
 *                           hidden (n, 8, 8, 8) f32, both NCHW contiguous, x != hidden. */
#define MZ_TOWER_X3_ENDS_BYTES ((2 * 5 * 128 * 32 + 2 * 4 * 80 * 32) * 2)
int mz_tower_x3_pack_ends(const float *w_first, int32_t cin, const float *w_last, void *ends, void *stream);
int mz_tower_x3_forward(const void *packed, const void *ends, const float *bias, const float *b_first, const float *b_last, int32_t cin,
                        const float *x, float *hidden, int64_t n, void *stream);
/* initial_inference / recurrent_inference (network.py:146-163) at fp32 grade in ONE launch: mz_tower_x3_forward + the prediction
 * heads in plain fp32 behind it.  heads: mz_heads_weights whose p1_w / p2_w are F32 [out][in] matrices here (nn.Linear's own
 * tensors), 16-byte aligned.  value (n) f32, policy (n,65) f32 softmaxed, hidden (n,8,8,8) f32. */
int mz_inference_x3_forward(const void *packed, const void *ends, const float *bias, const float *b_first, const float *b_last, int32_t cin,
                            const mz_heads_weights *heads, const float *x, float *hidden, float *value, float *policy, int64_t n, void *stream);
/* diagnostics: the same launch (relu_in = 0) writing shader-cycle stamps at the phase boundaries of every 4-board tile into
 * stamps, (ceil(n / 4), 32) int64 on the device (tools/restower_stamps.py reads them) */
int mz_restower_x3_stamps(const void *packed, const float *bias, const float *x, float *y, int64_t n, void *stamps, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MUZERO_AMD_H */
