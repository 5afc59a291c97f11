// tower_kernels.hip -- the MuZero convolution tower as ONE LDS-resident MFMA kernel for gfx950.
//
// Replaces, for bf16 serving, the per-layer PyTorch/MIOpen calls of network.py:49-77
// (Representation) and network.py:113-135 (Dynamics): 3x3 conv Cin->128 + ReLU, three residual
// blocks relu(x + conv(relu(conv(x)))) (network.py:35-47), 3x3 conv 128->8 + ReLU -- and, behind it in
// the same launch, the prediction heads (network.py:79-111), so one kernel returns the whole
// (value, policy, hidden) tuple of initial_inference / recurrent_inference (network.py:146-163).
//
// Why one kernel.  An 8x8 board with 128 channels is 16 KB in bf16, so EIGHT boards (512 positions, the GEMM N dimension)
// fit in 128 KB of the CU's 160 KB LDS as ONE activation image that every layer updates in place.  A workgroup therefore
// carries its eight boards through all eight layers without ever writing an activation to HBM: per tile it reads 10 KB of
// input planes, writes 8 KB of hidden state, and streams the layer weights (295 KB per 128->128 layer, L2-resident: every
// workgroup reads the same 1.9 MB) through a ring of four 8 KB LDS buffers.  What is left is MFMA work.
//
// Mapping (one workgroup = 512 threads = 8 waves, 2 per SIMD; static LDS = 163,840 B = activation image 128 KB | weight
// ring 4 x 8 KB):
//   GEMM per layer: D[co][pos] = sum_{tap,ci} W[tap][co][ci] * X[pos + shift(tap)][ci]
//   v_mfma_f32_16x16x32_bf16 (first layer: 32x32x16) with A = weights (rows = co), B = activations
//   (cols = pos), so a lane ends up holding ONE position and 4-element runs of consecutive channels: the
//   epilogue (bf16 rounding, ReLU) writes 8-byte pieces straight back into the [pos][ch] LDS
//   image the next layer reads -- no transpose, no shuffle.  The accumulators start at the bias.
//   wave w: co half (w & 1) x board pair (w >> 1)  ->  a 4 x 8 grid of 16 x 16 accumulators per wave (128 registers).
//   In place: a layer's whole output lives in the workgroup's accumulators when its K loop ends, so after one barrier
//   the epilogue overwrites the image it has just read.  The residual x + conv(relu(conv(x))) needs no second image either:
//   the epilogue of a block's first convolution reads each lane's own cells of x before it overwrites them with t and
//   leaves x + bias in the accumulators the second convolution then adds into.
//   K loop: 9 taps x 4 channel quarters = 36 stages of [128 co][32 ci] (one 8 KB weight stage feeds 32 MFMAs per wave,
//   twice what a 4-board tile gives it, and a stage's 12 fragment reads serve those 32 MFMAs: less LDS and L2 traffic per
//   MFMA is what lets the chip hold its clock -- the kernel is power-limited), filled by LDS-DMA three stages ahead;
//   one barrier per stage that waits for DMA pieces only.
//   The 3x3 halo needs no padding ring in the activation image: a lane whose shifted position leaves
//   the board reads its fragment past the end of the LDS allocation, which returns zeros (OFF_Z).
// LDS images are XOR-swizzled in 16-byte chunks so every ds_read_b128 lane group hits 16 distinct
// 4-bank slots: activations  slot = chunk ^ ((y & 1) << 3 | x),  weights (64-byte rows)  slot = chunk ^ perm[(co >> 2) & 3], perm = {0, 2, 3, 1};
// the MFMA column -> board cell map (col_cell) is chosen for the hardware's non-contiguous lane groups.
// Weights arrive PREPACKED in exactly these images (muzero_amd/network.py: pack_tower), so staging
// is a linear copy.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int TILE_BOARDS = 8;
constexpr int TILE_POS = 512;
constexpr int ACT_BYTES = TILE_POS * 128 * 2;  // 131,072: the [512 pos][128 ch] bf16 image, updated in place by every layer
constexpr int WSTAGE_BYTES = 128 * 32 * 2;     // 8,192: one [128 co][32 ci] weight stage
constexpr int OFF_X = 0, OFF_W = ACT_BYTES;
constexpr int LDS_BYTES = ACT_BYTES + 4 * WSTAGE_BYTES;  // 163,840 = all of a CU's LDS
constexpr int OFF_Z = LDS_BYTES;  // the halo's zero fragment: a DS read at or beyond the workgroup's LDS allocation returns 0
                                  // (hardware bounds check; the allocation is the whole 160 KB, so nothing lies behind it)
constexpr int N_STAGES = 36;                   // 9 taps x 4 quarters of the input channels
constexpr int FIRST_W_BYTES = 9 * 128 * 16 * 2;   // 36,864
constexpr int OFF_FIN = 0, OFF_FW = TILE_POS * 32;   // first layer scratch inside the (not yet written) image: [512 pos][16 ch] input | tap weight slabs
constexpr int LAST_ROWS = 80;                       // 9 taps x 8 planes = 72 weight rows, padded to five 16-row MFMA tiles
constexpr int LAST_W_BYTES = 4 * LAST_ROWS * 32 * 2;  // 20,480: [4 channel quarters][80 rows][32 ci], sent into ring buffers 0..2 by the last K loop
constexpr int P_WAVE_BYTES = 9 * 2 * 64 * 16;       // 18,432: one board's per-tap partial sums of the last layer, [tap][plane half][cell][4 planes] f32
static_assert(OFF_FW + FIRST_W_BYTES <= ACT_BYTES && 8 * P_WAVE_BYTES <= LDS_BYTES && LAST_W_BYTES <= 3 * WSTAGE_BYTES, "LDS plan");

struct Tower {
    const uint4 *w_first;  // [9][128 co][16 ci] bf16
    const float *b_first;  // [128] or null
    const uint4 *w_mid;    // [6][36][128 co][4 slots][8] bf16, slot = chunk ^ {0,2,3,1}[(co >> 2) & 3]; + 4 stages of padding
    const float *b_mid;    // [6][128]
    const uint4 *w_last;   // [4 quarters][80 rows = tap * 8 + plane (72 real)][4 slots][8] bf16, same 64-byte-row swizzle as w_mid
    const float *b_last;   // [8] or null
    int cin;               // real input planes (9 or 10)
    long long *stamps;     // diagnostics: [blocks][24] s_memtime stamps at the phase boundaries (null = off)
};

// prediction heads (network.py:79-111): the two Linear matrices in bf16 (the serving network's weights are bf16 values), everything else f32
struct Heads {
    const float *pc_w, *pc_b;    // policy 1x1 conv [2][8], [2]
    const uint16_t *p1_w;        // Linear(128,128) weight, bf16 [out 128][in 128] (nn.Linear layout)
    const float *p1_b;           // [128]
    const uint16_t *p2_w;        // Linear(128,65) weight, bf16 [out 65][in 128]
    const float *p2_b;           // [65]
    const float *vc_w, *vc_b;    // value 1x1 conv [2][8], [2]
    const float *v1_w, *v1_b;    // Linear(128,1) [128], [1]
    float *value, *policy;       // outputs (n), (n,65); null = heads off
};

// where the hidden states around the tower live (mz_inference_forward_routed): rows of 512 elements
struct Route {
    const uint16_t *src_pool;   // null: x holds all cin planes
    const int32_t *src_slot;
    const int32_t *dst_slot;    // null: row = board * dst_stride
    int64_t dst_stride;         // 1: the plain (n, 512) output
};
__device__ __forceinline__ int64_t dst_row(const Route &rt, int64_t b) { return rt.dst_slot ? (int64_t)rt.dst_slot[b] : b * rt.dst_stride; }

__device__ __forceinline__ int act_key(int q) { return (((q >> 3) & 1) << 3) | (q & 7); }
// MFMA 16x16x32 column c (= lane & 15) -> cell of the 16-position tile (2 board rows).  ds_read_b128 is served in the lane
// groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md, LDS): columns {0-3,12-15} of one k-chunk share a
// group with columns {4-11} of the NEXT k-chunk.  Giving {0-3,12-15} the tile's first board row and {4-11} its second
// keeps the two sets on different values of the key's row bit for every tap shift, so a group's 16 reads hit 16 distinct slots.
__device__ __forceinline__ int col_cell(int c) { return c < 4 ? c : (c < 12 ? c + 4 : c - 8); }

__device__ __forceinline__ uint32_t bf16_rne(float f) {  // finite inputs (post-ReLU)
    const uint32_t u = __float_as_uint(f);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf16_up(uint32_t v) { return __uint_as_float(v << 16); }
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
// ReLU of two packed bf16: as signed 16-bit integers negative floats (and -0) are negative, positive floats unchanged
typedef __attribute__((ext_vector_type(2))) short i16x2;
__device__ __forceinline__ uint32_t relu_bf16x2(uint32_t v) {
    const i16x2 x = __builtin_bit_cast(i16x2, v), z = {0, 0};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(x, z));
}
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {   // v_cvt_pk_bf16_f32: round to nearest even
    const f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

// validity of (tap, position tile j) for this lane: bit tap*2+j.  Position tile j of a board covers
// rows 4j..4j+3; the lane's cell is row 4j + (r >> 3), column r & 7.
__device__ __forceinline__ uint32_t halo_mask(int r) {
    uint32_t m = 0;
    const int x = r & 7;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap % 3;
        const int xx = x + kx - 1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int yy = 4 * j + (r >> 3) + ky - 1;
            if (xx >= 0 && xx <= 7 && yy >= 0 && yy <= 7) m |= 1u << (tap * 2 + j);
        }
    }
    return m;
}

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ bf16x8 lds_frag(const char *p) { return *reinterpret_cast<const bf16x8 *>(p); }

// one weight stage (8 KB) global -> LDS by LDS-DMA: each of the 8 waves moves 1 KB with ONE instruction
// (lane-linear: LDS address = wave-uniform base + 16 * lane), no VGPRs, no ds_write.  The packed image is
// already in LDS order, so the source is linear too.
// Written as inline asm in its SGPR-base + 32-bit lane offset form (m0 = the wave's LDS address, s_nop for the m0 hazard).
// With the builtin anywhere in a loop, hipcc (ROCm 7.2) treats the LDS-DMA as a FLAT access that may return out of order and
// turns every counted lgkmcnt(N) it would place in front of an MFMA into lgkmcnt(0): the K loop's streamed fragment reads
// then expose their whole LDS latency (tools/mfma_rate.hip: 83.4 % of the matrix pipe with the builtin, 91.5 % with this form).
// The compiler does not see this load's vmcnt: every wait for it is written by hand (s_waitcnt vmcnt(N) in the barriers).
__device__ __forceinline__ void dma_piece(uint32_t lds_addr_wave, const void *__restrict__ src_uniform, uint32_t lane_off) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr_wave), "v"(lane_off), "s"(src_uniform) : "memory");
}
// LDS byte address (wave-uniform, in an SGPR) of wave tid >> 6's 1 KB piece of the region at byte offset off
__device__ __forceinline__ uint32_t wave_lds_addr(char *lds, int off, int tid) {
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds + off) + (tid >> 6) * 1024);
}
__device__ __forceinline__ void dma_stage(char *ring_buf, const uint4 *__restrict__ stage, int tid) {
    dma_piece(wave_lds_addr(ring_buf, 0, tid), stage, (uint32_t)tid * 16u);
}

// ---- 128 -> 128 layer, in place in the activation image ----
// K loop: 36 stages of [128 co][32 ci] weights (8 KB = one K = 32 step = 32 MFMAs per wave) through a ring of FOUR LDS buffers
// filled by LDS-DMA.  A stage is eight j-steps (j = position tile of the wave's two boards): 4 MFMAs (the stage's 4 weight
// fragments x activation fragment j), one activation fragment read and, in the first four, one weight fragment read:
//   iteration st, behind barrier B_st:
//     j = 0..7:  MFMA(0,j) MFMA(1,j) | refill the activation slot j-step j - 1 has left with fragment j + 3 (B_st[3..7] at
//                j = 0..4, then B_st+1[0..2]) | MFMA(2,j) MFMA(3,j) | j < 4: weight fragment A_st+1[j] into the other weight set
//     behind j = 0: the DMA of stage st + 3 into the buffer stage st - 1 has left
// so activation fragments stream three j-steps ahead of their MFMAs through 4 slots, weight fragments one stage ahead through 2
// sets, and NO wait of the loop is an lgkmcnt(0): hipcc places a counted lgkmcnt(N) in front of each j-step (it can, because
// the DMA is inline asm: dma_piece), and the barrier waits for DMA pieces only.
// B_st (s_waitcnt vmcnt(1) + s_barrier) orders "every wave has issued stage st - 1's MFMAs, so it holds that stage's weight
// fragments in registers and that buffer is free" and "every wave's piece of stage st + 1 has landed" (all but the youngest
// DMA, the one of iteration st - 1), so A_st+1 is readable anywhere in iteration st.  No barrier needs the activation reads
// to be back: the image is not written during the K loop, and a fragment's consumer waits for it by data dependence.
// The weight stream is one contiguous array over all six layers (+ 4 stages of padding): it simply keeps running across layer
// boundaries, so only the first layer pays a prologue (mid_prologue).  The LAST layer's stream ends in the final convolution's
// weights instead (w_last, 20 KB = 2.5 stages): they arrive in ring buffers 0..2 as "stages 36..38".
// Halo: a lane whose shifted position leaves the board reads its fragment past the end of the LDS allocation (zeros, OFF_Z)
// -- the select is on the ADDRESS, once per tap.  Position tile j covers rows 2 (j & 3), 2 (j & 3) + 1 of board j >> 2, so the
// fragment address is (row base of the tap) + 4096 j + (chunk offset of the stage); only tiles 0 and 4 can leave their board
// at the top and tiles 3 and 7 at the bottom: three base registers per tap (main / top / bottom), one add each per stage, the
// tile's 4096 j as the instruction's immediate offset.
// Epilogue (behind a barrier: every wave is through its reads of the image): each lane rounds its 128 values to bf16, applies
// ReLU and writes them over the image; MODE 0 (first convolution of a residual block) first reads the cells' old values x and
// leaves x + bias_next in the accumulators, MODE 1 leaves bias_next: the next K loop starts from them.

// the first three weight stages of the first mid layer: issued at kernel start (the ring is idle during the
// first layer), complete by the vmcnt(0) + barrier that ends the first layer
__device__ __forceinline__ void mid_prologue(char *lds, const uint4 *__restrict__ wp, int tid) {
    dma_stage(lds + OFF_W, wp, tid);
    dma_stage(lds + OFF_W + WSTAGE_BYTES, wp + 512, tid);
    dma_stage(lds + OFF_W + 2 * WSTAGE_BYTES, wp + 2 * 512, tid);
}

#define MZ_STAMP(tw, k)                                                                              \
    do {                                                                                             \
        if ((tw).stamps && threadIdx.x == 0) (tw).stamps[(k)] = (long long)__builtin_readcyclecounter(); /* stamps: this tile's row */ \
    } while (0)

// MFMA shape of the K loop: v_mfma_f32_16x16x32_bf16.  One stage (32 input channels) is exactly one K = 32 step:
// 4 weight fragments (16 co x 32 ci) x 8 activation fragments (16 positions x 32 ci) -> 32 MFMAs into a 4 x 8 grid of
// 16 x 16 accumulators.  Same bytes, cycles and registers as 32x32x16 MFMAs, but the chip holds a higher clock on
// this shape under sustained load (MI355X_MICROARCH.md, DVFS item 7; measured here: in-kernel clock 1.87 GHz with 32x32x16).
//   A (weights):     lane l -> row co = 16 i + (l & 15), k = 8 (l >> 4) + 0..7  = the l >> 4 -th 16-byte chunk of the 64-byte row
//   B (activations): lane l -> col pos = 16 j + (l & 15), same k                 = chunk 4 kq + (l >> 4) of the 256-byte row
//   D:               lane l -> col pos, rows co = 16 i + 4 (l >> 4) + 0..3       = 8 bytes of the [pos][ch] image
typedef __attribute__((ext_vector_type(4))) float acc16_t;

// NJ = position tiles per wave: 8 (two boards per wave, the 8-board tile of the serving kernel) or 4 (one board per wave, a
// 4-board tile: the training convolution on small batches, where twice the workgroups matter more than the K loop's density)
// DUAL (the split-precision tower, k_restower_x3): the NJ position tiles are NJ / 2 real tiles read from TWO images (hi at 0, lo
// at X3_LO) that accumulate into the same NJ / 2 accumulators: one weight stage feeds W x_hi and W x_lo, 32 MFMAs per 12 fragment
// reads like the 8-board tile, on a 4-board tile.
constexpr int X3_LO = 65536;         // byte offset of the lo image
// NI = output-channel tiles per wave: 4 (wave = 64 co, two co halves x four position groups), or 2 (wave = 32 co, four co quarters x
// two position groups: with NJ = 4 a 2-board tile -- the training convolution when even 4-board tiles leave CUs idle; MODE 2 only)
template <int MODE, int NJ = 8, bool DUAL = false, int NI = 4>
__device__ __forceinline__ void layer_mid(char *lds, const uint4 *__restrict__ wp, const uint4 *__restrict__ w_last, acc16_t (&acc)[NI][DUAL ? NJ / 2 : NJ],
                                          const float *__restrict__ bias_next, int tid, const Tower &tw, int stamp_id, int xoff = OFF_X) {
    // xoff: byte offset of the image the K loop READS (the split-precision tower keeps two: k_restower_x3); the MODE 0 / 1
    // epilogues below work on the image at OFF_X
    const int w = tid >> 6, lane = tid & 63, c16 = lane & 15, kc = lane >> 4, p16 = col_cell(c16);
    constexpr int NA = DUAL ? NJ / 2 : NJ;   // accumulator tiles per wave
    constexpr int CW = 8 / NI;               // waves across the 128 output channels
    static_assert((NI == 4 || (NI == 2 && MODE == 2)) && NJ >= 4, "layer_mid shapes");
    const int cobase = 16 * NI * (w & (CW - 1)), posbase = 16 * NA * (w / CW);

    // weights: address = ring buffer + co * 64 + ((kc ^ key(co)) << 4), key = perm[(co >> 2) & 3], perm = {0, 2, 3, 1}
    int waddr[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int co = cobase + 16 * i + c16;
        waddr[i] = OFF_W + co * 64 + ((kc ^ ((0x78 >> (((co >> 2) & 3) * 2)) & 3)) << 4);
    }
    const uint32_t ring_m0 = wave_lds_addr(lds, OFF_W, tid);   // this wave's 1 KB piece of ring buffer 0
    const uint32_t lane_off = (uint32_t)tid * 16u;
    // activations, per tap: the three row bases (or the out-of-range zero block) and the chunk key; per stage: + chunk offset
    int tbm, tbt, tbb, ak, am, at, ab;
    const int x = p16 & 7, y0 = p16 >> 3;
    auto set_tap = [&](int ky, int kx) {   // tap = 3 ky + kx
        const int dq = (ky - 1) * 8 + (kx - 1);
        const bool okx = (unsigned)(x + kx - 1) < 8u;
        const int base = xoff + (posbase + p16 + dq) * 256;      // tile 0's shifted row; tile j: + 4096 j
        ak = (kc ^ act_key((p16 + dq) & 15)) << 4;               // the key's bits (row parity, column) do not depend on the tile
        tbm = okx ? base : OFF_Z;
        tbt = okx && (y0 + ky - 1) >= 0 ? base : OFF_Z;          // tiles 0, 4 = board rows 0, 1
        tbb = okx && (6 + y0 + ky - 1) < 8 ? base : OFF_Z;       // tiles 3, 7 = board rows 6, 7
    };
    auto set_kq = [&](int kq) {
        const int off = (kq * 64) ^ ak;
        am = tbm + off;
        at = tbt + off;
        ab = tbb + off;
    };
    auto bfrag = [&](int j) { return lds_frag(lds + ((j & 3) == 0 ? at : ((j & 3) == 3 ? ab : am)) + (j & (NA - 1)) * 4096 + (j / NA) * X3_LO); };

    bf16x8 a[2][NI], b[4];
    set_tap(0, 0);
    set_kq(0);
#pragma unroll
    for (int j = 0; j < 3; ++j) b[j] = bfrag(j);
#pragma unroll
    for (int i = 0; i < NI; ++i) a[0][i] = lds_frag(lds + waddr[i]);

    // 3 trips x 12 stages (3 taps): ring index, channel quarter and the register-set parity are compile-time constants
    // and the body is branch-free.  The last trip prefetches "tap 9" fragments (any address is a legal LDS read; they are
    // never used) and stages 36..38 (the next layer's first stages, or the last convolution's weights).
#pragma unroll 1
    for (int T = 0; T < 3; ++T) {
        const uint4 *wt = wp + (T * 12 + 3) * 512;
        const bool tail = w_last != nullptr && T == 2;   // wave-uniform
#pragma unroll
        for (int ss = 0; ss < 12; ++ss) {
            // all but the youngest DMA (stage st + 2, issued one iteration ago) have landed: stage st + 1 is complete
            asm volatile("s_waitcnt vmcnt(1)\n\ts_barrier" ::: "memory");
            const bf16x8(&ca)[NI] = a[ss & 1];
            bf16x8(&na)[NI] = a[(ss + 1) & 1];
            const int nbuf = ((ss + 1) & 3) * WSTAGE_BYTES;   // 12 = 0 (mod 4, mod 2): ss + 1 indexes like st + 1
#define MZ_FENCE() __builtin_amdgcn_sched_barrier(0)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
#pragma unroll
                for (int i = 0; i < NI / 2; ++i)
                    acc[i][j & (NA - 1)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ca[i], b[j & 3], acc[i][j & (NA - 1)], 0, 0, 0);
                MZ_FENCE();
                b[(j + 3) & 3] = bfrag((j + 3) & (NJ - 1));   // j <= 4: this stage's tile j + 3; j >= 5: the next stage's tile j - 5 (its addresses are set behind j = 4)
                MZ_FENCE();
#pragma unroll
                for (int i = NI / 2; i < NI; ++i)
                    acc[i][j & (NA - 1)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ca[i], b[j & 3], acc[i][j & (NA - 1)], 0, 0, 0);
                MZ_FENCE();
                if (j < NI) {
                    na[j] = lds_frag(lds + nbuf + waddr[j]);
                    MZ_FENCE();
                }
                if (j == 0) {   // stage st + 3 -> the buffer stage st - 1 left
                    if (ss < 9) {
                        dma_piece(ring_m0 + ((ss + 3) & 3) * WSTAGE_BYTES, wt + ss * 512, lane_off);
                    } else {   // stages 36..38 of the layer: the next layer's first stages, or (tail) the last convolution's 20 KB
                        const uint4 *src = tail ? w_last + (ss - 9) * 512 : wt + ss * 512;
                        if (ss < 11 || !tail || w < (LAST_W_BYTES - 2 * WSTAGE_BYTES) / 1024)
                            dma_piece(ring_m0 + ((ss + 3) & 3) * WSTAGE_BYTES, src, lane_off);
                    }
                    MZ_FENCE();
                }
                if (j == NJ - 4) {   // the addresses of stage st + 1 (next channel quarter, or the next tap's first): behind the last refill that reads stage st
                    if (((ss + 1) & 3) == 0) set_tap(ss + 1 == 12 ? T + 1 : T, ss + 1 == 12 ? 0 : (ss + 1) >> 2);
                    set_kq((ss + 1) & 3);
                    MZ_FENCE();
                }
            }
#undef MZ_FENCE
        }
    }

    MZ_STAMP(tw, stamp_id);
    // MODE 2 (the training convolution k_conv3x3, the split-precision tower): the caller stores the accumulators; nothing touches the image
    if constexpr (MODE != 2) {
    // the next layer's bias in the accumulator layout: channels cobase + 16 i + 4 kc + 0..3
    f32x4 bn[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bn[i] = bias_next ? *reinterpret_cast<const f32x4 *>(bias_next + cobase + 16 * i + 4 * kc) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    // in place: every wave must be through its reads of the image (the "tap 9" prefetches included) before any cell changes
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // epilogue: lane = position, the 4 registers of a tile = channels co0 .. co0 + 3
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = posbase + 16 * j + p16;
        char *row = lds + OFF_X + q * 256 + (kc & 1) * 8;
        const int key = act_key(q);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int chunk = (cobase >> 3) + 2 * i + (kc >> 1);   // co0 = cobase + 16 i + 4 kc
            char *p = row + ((chunk ^ key) << 4);
            uint2 xr = make_uint2(0u, 0u);
            if (MODE == 0) xr = *reinterpret_cast<const uint2 *>(p);   // x, before t overwrites it
            uint2 o;   // round, then ReLU on the packed bf16 pair (same result as ReLU then round: rounding keeps the sign)
            o.x = relu_bf16x2(pack_bf16(acc[i][j][0], acc[i][j][1]));
            o.y = relu_bf16x2(pack_bf16(acc[i][j][2], acc[i][j][3]));
            *reinterpret_cast<uint2 *>(p) = o;
            if (MODE == 0)   // the second convolution of the block accumulates onto x + its bias
                acc[i][j] = acc16_t{bf16_up(xr.x & 0xFFFFu) + bn[i][0], bf16_up(xr.x >> 16) + bn[i][1], bf16_up(xr.y & 0xFFFFu) + bn[i][2],
                                    bf16_up(xr.y >> 16) + bn[i][3]};
            else
                acc[i][j] = bn[i];
        }
    }
    // raw barrier (LDS only): the weight DMAs still in flight belong to the NEXT layer's stages 1 and 2 (or are the last
    // convolution's weights) and are retired by the next counted waits; __syncthreads() would stall every layer boundary on them
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}

// ---- first layer: input planes (global, NCHW) -> the image.  Its operands are staged INSIDE the (not yet written) image:
// [512 pos][16 ch] input at OFF_FIN, all 9 tap weight slabs [9][128 co][16 ci] at OFF_FW; the layer's whole output waits in
// the accumulators (wave = 64 co x 128 positions = 2 x 4 tiles of 32 x 32) until every wave has read its operands. --------
__device__ __forceinline__ void layer_first(char *lds, const Tower &tw, const Route &rt, const uint16_t *__restrict__ x, int64_t board0, int64_t n_boards,
                                            int tid, uint32_t vm) {
    const int w = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int cobase = 64 * (w & 1), posbase = 128 * (w >> 1);
    char *in = lds + OFF_FIN;
    char *wf = lds + OFF_FW;
    f32x4 bv[2][4];   // requested now; the accumulators start at the bias (D rows of a 32x32 tile: 8 g + 4 h + k)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bv[i][g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (tw.b_first) bv[i][g] = *reinterpret_cast<const f32x4 *>(tw.b_first + cobase + 32 * i + 8 * g + 4 * h);
        }
    {   // input: thread -> position tid, channels 0..15
        const int q = tid;
        const int64_t b = board0 + (q >> 6);
        uint32_t pk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (b < n_boards) {
            // planes 0..7 and the planes behind them: one (n, cin, 8, 8) array, or a pool row + an (n, cin - 8, 8, 8) array
            const uint16_t *plo = rt.src_pool ? rt.src_pool + (int64_t)rt.src_slot[b] * 512 + (q & 63) : x + b * tw.cin * 64 + (q & 63);
            const uint16_t *phi = rt.src_pool ? x + b * (tw.cin - 8) * 64 + (q & 63) : plo + 512;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const uint32_t v = k < tw.cin ? (uint32_t)(k < 8 ? plo[k * 64] : phi[(k - 8) * 64]) : 0u;
                pk[k >> 1] |= v << ((k & 1) * 16);
            }
        }
        *reinterpret_cast<uint4 *>(in + q * 32) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        *reinterpret_cast<uint4 *>(in + q * 32 + 16) = make_uint4(pk[4], pk[5], pk[6], pk[7]);
        for (int i = tid; i < FIRST_W_BYTES / 16; i += 512) *reinterpret_cast<uint4 *>(wf + i * 16) = tw.w_first[i];
    }
    __syncthreads();
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[i][j][k] = bv[i][k >> 2][k & 3];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int dq = (tap / 3 - 1) * 8 + (tap % 3 - 1);
        bf16x8 a[2], b[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = lds_frag(wf + (tap * 128 + cobase + 32 * i + r) * 32 + h * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) {   // 32-position tile j = rows 4 (j & 1) .. + 3 of board j >> 1
            const int q2 = (posbase + 32 * j + r + dq) & 511;
            b[j] = lds_frag(lds + (((vm >> (tap * 2 + (j & 1))) & 1u) ? OFF_FIN + q2 * 32 : OFF_Z) + h * 16);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();   // every wave has read its operands: the image may be written over them
    char *dst = lds + OFF_X;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co0 = cobase + 32 * i + 8 * g + 4 * h;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = posbase + 32 * j + r;
                char *p = dst + q * 256 + ((((co0 >> 3) ^ act_key(q))) << 4) + h * 8;
                uint2 o;   // round, then ReLU on the packed pair (as layer_mid)
                o.x = relu_bf16x2(pack_bf16(acc[i][j][4 * g], acc[i][j][4 * g + 1]));
                o.y = relu_bf16x2(pack_bf16(acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]));
                *reinterpret_cast<uint2 *>(p) = o;
            }
        }
    // the weight prologue's DMA pieces (mid_prologue: inline asm, invisible to the compiler's own waits) have landed too
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

struct HeadRegs { bf16x8 a1[4], a2[4]; };   // this wave's weight (A) fragments of the two Linear layers, one per k-step of 32 inputs

// every global read of the heads' weights, issued at the start of heads_batch.
// Linear weights are bf16 [out][in] row-major (nn.Linear's own layout): wave w owns output rows 16 w .. 16 w + 15, and
// lane l's A fragment of k-step ks is the 16 bytes at row 16 w + (l & 15), inputs 32 ks + 8 (l >> 4) .. + 7.
__device__ __forceinline__ void heads_prefetch(const Heads &hd, int tid, HeadRegs &hr) {
    const int w = tid >> 6, lane = tid & 63;
    const int r1 = 16 * w + (lane & 15);
    const int r2 = r1 < MZ_ACTIONS ? r1 : MZ_ACTIONS - 1;   // Linear(128,65): rows past 64 repeat row 64, their outputs are never read
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        hr.a1[ks] = *reinterpret_cast<const bf16x8 *>(hd.p1_w + r1 * 128 + 32 * ks + 8 * (lane >> 4));
        hr.a2[ks] = *reinterpret_cast<const bf16x8 *>(hd.p2_w + r2 * 128 + 32 * ks + 8 * (lane >> 4));
    }
}
// the heads' small parameters, one per thread (staged in LDS by heads_batch)
__device__ __forceinline__ float heads_param(const Heads &hd, int tid) {
    float pv = 0.0f;
    if (tid < 16) pv = hd.pc_w[tid];
    else if (tid < 18) pv = hd.pc_b[tid - 16];
    else if (tid < 34) pv = hd.vc_w[tid - 18];
    else if (tid < 36) pv = hd.vc_b[tid - 34];
    else if (tid >= 40 && tid < 168) pv = hd.p1_b[tid - 40];
    else if (tid >= 168 && tid < 168 + MZ_ACTIONS) pv = hd.p2_b[tid - 168];
    else if (tid >= 240 && tid < 368) pv = hd.v1_w[tid - 240];
    else if (tid == 368) pv = hd.v1_b[0];
    return pv;
}

// ---- last layer: image -> hidden (8 planes), with the nine taps in the M dimension ---------------------------------------
//   P[tap * 8 + plane][pos] = sum_ci W[tap][plane][ci] * X[pos][ci]        one GEMM, M = 72 (padded 80), N = 512, K = 128
//   hidden[plane][pos]      = relu(b[plane] + sum_tap P[tap * 8 + plane][pos + shift(tap)])     (positions off the board: 0)
// With the taps in K (as the mid layers have them) a 16-row MFMA tile would carry 8 real rows and every activation row would
// be fetched from LDS nine times; here wave w reads the 64 positions of board w ONCE (16 fragments) and the 20 weight
// fragments, and holds its 72 x 64 block of fp32 partial sums in registers.  Behind one barrier (image and ring are dead then)
// it writes them to LDS -- 18 KB per wave, private: a tap's shift stays inside the board -- and every lane sums the nine
// shifted taps of one cell x eight planes.  No barrier between the writes and the reads: the LDS serves a wave in order.
//   LDS: weights [4 quarters][80 rows][32 ci] in ring buffers 0..2 (sent there as the last K loop's "stages 36..38");
//   P of wave w at w * P_WAVE_BYTES as [tap][plane half][cell][4 planes] fp32 (16-byte entries: conflict-free both ways).
// Output: lane -> cell of board w, planes 0..7 in hv[0..7] (what store_hidden writes and the heads read).
template <int OUT_DT>
__device__ __forceinline__ void layer_last(char *lds, const Tower &tw, int tid, float (&hv)[8]) {
    const int w = tid >> 6, lane = tid & 63, c16 = lane & 15, kc = lane >> 4, p16 = col_cell(c16);
    const char *wl = lds + OFF_W;
    float *P = reinterpret_cast<float *>(lds + w * P_WAVE_BYTES);
    // retire every wave's DMA pieces of the weights
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    bf16x8 bf[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = 64 * w + 16 * j + p16;
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) bf[j][kq] = lds_frag(lds + OFF_X + q * 256 + (((4 * kq + kc) ^ act_key(q)) << 4));
    }
    acc16_t acc[5][4];
#pragma unroll
    for (int m = 0; m < 5; ++m) {
        const int row = 16 * m + c16;
        const int waddr = row * 64 + ((kc ^ ((0x78 >> (((row >> 2) & 3) * 2)) & 3)) << 4);
        bf16x8 af[4];
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) af[kq] = lds_frag(wl + kq * (LAST_ROWS * 64) + waddr);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[m][j] = acc16_t{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kq], bf[j][kq], acc[m][j], 0, 0, 0);
        }
    }
    // every wave holds its fragments in registers: the image and the ring may be overwritten
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // D: lane -> cell (its column), rows 16 m + 4 kc + 0..3 = tap 2 m + (kc >> 1), planes 4 (kc & 1) + 0..3
#pragma unroll
    for (int m = 0; m < 5; ++m) {
        const int tap = 2 * m + (kc >> 1);
        if (tap < 9) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = 16 * j + p16;
                *reinterpret_cast<f32x4 *>(P + ((tap * 2 + (kc & 1)) * 64 + q) * 4) = f32x4{acc[m][j][0], acc[m][j][1], acc[m][j][2], acc[m][j][3]};
            }
        }
    }
    {   // gather the nine shifted taps: lane -> cell q of board w, both plane halves
        const int q = lane, x = q & 7, y = q >> 3;
        f32x4 s0 = {0.0f, 0.0f, 0.0f, 0.0f}, s1 = {0.0f, 0.0f, 0.0f, 0.0f};
        if (tw.b_last) {
            s0 = *reinterpret_cast<const f32x4 *>(tw.b_last);
            s1 = *reinterpret_cast<const f32x4 *>(tw.b_last + 4);
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            const bool ok = (unsigned)(x + kx - 1) < 8u && (unsigned)(y + ky - 1) < 8u;
            const int q2 = (q + (ky - 1) * 8 + (kx - 1)) & 63;
            const f32x4 p0 = *reinterpret_cast<const f32x4 *>(P + ((tap * 2 + 0) * 64 + q2) * 4);
            const f32x4 p1 = *reinterpret_cast<const f32x4 *>(P + ((tap * 2 + 1) * 64 + q2) * 4);
            if (ok) {
                s0 += p0;
                s1 += p1;
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float v = fmaxf(k < 4 ? s0[k & 3] : s1[k & 3], 0.0f);
            hv[k] = OUT_DT == MZ_F32 ? v : bf16_up(bf16_rne(v));   // what store_hidden writes (and the heads read back)
        }
    }
    // every wave is done with its partial sums: whatever comes next (the heads, or the next tile's first layer and weight
    // prologue) may overwrite them
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// the hidden state (global, NCHW, 8 planes) from layer_last's registers: wave -> board, lane -> cell; consecutive lanes write
// consecutive cells of one plane.  Nothing in the kernel waits on these stores.
template <int OUT_DT>
__device__ __forceinline__ void store_hidden(const float (&hv)[8], void *__restrict__ hidden, const Route &rt, int64_t board0, int64_t n_boards, int tid) {
    const int64_t b = board0 + (tid >> 6);
    if (b >= n_boards) return;
    const int64_t row = dst_row(rt, b);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int64_t o = row * 512 + k * 64 + (tid & 63);
        if (OUT_DT == MZ_F32) reinterpret_cast<float *>(hidden)[o] = hv[k];
        else reinterpret_cast<uint16_t *>(hidden)[o] = (uint16_t)(__float_as_uint(hv[k]) >> 16);
    }
}

// ---- prediction heads (network.py:79-111) for a BATCH of up to 2 tiles = 16 boards of this workgroup ---------------------
// Run once per tile on the VALU the heads cost 8.6k cycles of an otherwise matrix-bound kernel, and that cost is arithmetic
// (4 x 25k fp32 MACs + bf16 unpacking per tile; batching alone bought nothing).  Sixteen boards are exactly one MFMA column
// tile, so the persistent workgroup runs the heads once per TWO tiles, on the hidden states it stored itself (slot s of the
// batch = board s & 7 of tile tile0 + (s >> 3) * tstride), with both Linear layers on v_mfma_f32_16x16x32_bf16:
//   D[out][board] = sum_in W[out][in] * f[in][board],  A = weights (bf16, exact), B = features.
// The features are fp32 (1x1 conv + ReLU outputs); each goes in as TWO bf16 terms, hi = bf16(f) and lo = bf16(f - hi), two
// MFMAs per k-step into one fp32 accumulator: 16 significant bits of f survive, products are exact, sums are fp32 -- within
// 2e-5 of the fp32 module (tests/test_gpu_net.py), where one rounding of f to bf16 would cost 1e-3.
// LDS (X image region only, dead here): hid f32 [16][8][64] (later: logits f32 [16][80]) | fp hi, lo bf16 [16][136] | fv f32 [16][128] |
// h1 hi, lo bf16 [16][136] | vsum f32 [16] | prm f32 [512]
constexpr int HB = 16;     // boards per heads batch
constexpr int FROW = 136;  // bf16 per feature row (128 + 8 of padding): 272-byte rows put the 16 boards a B-fragment read touches on 16 distinct 4-bank slots
__device__ __forceinline__ void split_bf16(float f, uint32_t &hi, uint32_t &lo) {   // f >= 0, finite
    hi = bf16_rne(f);
    lo = bf16_rne(f - bf16_up(hi));   // exact difference (|f - hi| <= ulp/2), may be negative: rne on the bits is sign-symmetric
}
// wave-wide reductions on DPP (six VALU steps, result broadcast from lane 63); __shfl_xor goes through ds_bpermute, and a
// chain of six of those per reduction is what the softmax of 16 boards was waiting on
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float old, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += dpp_f32<0xB1, 0xF>(0.0f, v);    // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E, 0xF>(0.0f, v);    // quad_perm [2,3,0,1]
    v += dpp_f32<0x124, 0xF>(0.0f, v);   // row_ror:4
    v += dpp_f32<0x128, 0xF>(0.0f, v);   // row_ror:8   -> every lane holds its 16-lane row's sum
    v += dpp_f32<0x142, 0xA>(0.0f, v);   // row_bcast:15 into rows 1, 3
    v += dpp_f32<0x143, 0xC>(0.0f, v);   // row_bcast:31 into rows 2, 3 -> lane 63 holds the total
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max_dpp(float v) {
    v = fmaxf(v, dpp_f32<0xB1, 0xF>(v, v));
    v = fmaxf(v, dpp_f32<0x4E, 0xF>(v, v));
    v = fmaxf(v, dpp_f32<0x124, 0xF>(v, v));
    v = fmaxf(v, dpp_f32<0x128, 0xF>(v, v));
    v = fmaxf(v, dpp_f32<0x142, 0xA>(v, v));
    v = fmaxf(v, dpp_f32<0x143, 0xC>(v, v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// The hidden states of the batch's EARLIER tile come back from global memory (this workgroup stored them, store_hidden, a
// whole tile ago); the loads are issued before the last tile's last layer and land under its MFMAs.  They bypass the
// CU's L1 (agent-scope relaxed atomic load = global_load sc1).  The batch's last tile hands its values over in registers.
template <int OUT_DT>
struct Readback {
    static constexpr int NW = OUT_DT == MZ_F32 ? 512 : 256;   // dwords per board
    static constexpr int R = 8 * NW / 512;                    // dwords per thread for the earlier tile = 8 boards
    uint32_t v[R];
    float pv;   // this thread's small head parameter (heads_param), requested with the readback
};
template <int OUT_DT>
__device__ __forceinline__ void readback_issue(Readback<OUT_DT> &rb, const Heads &hd, const void *hidden, const Route &rt, int64_t tile0, int64_t tstride,
                                               int nprev, int tid) {
    constexpr int NW = Readback<OUT_DT>::NW;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's own earlier stores (long complete: a formality that costs nothing here)
#pragma unroll
    for (int r = 0; r < Readback<OUT_DT>::R; ++r) {
        const int i = tid + 512 * r, slot = i / NW;
        const int sl = slot < 8 * nprev ? slot : 0;    // unused slots re-read the batch's first board (valid) and are ignored later
        const int64_t b = (tile0 + (int64_t)(sl >> 3) * tstride) * TILE_BOARDS + (sl & 7);
        rb.v[r] = __hip_atomic_load(reinterpret_cast<const uint32_t *>(hidden) + dst_row(rt, b) * NW + (i % NW), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    rb.pv = heads_param(hd, tid);
}

template <int OUT_DT>
__device__ __forceinline__ void heads_batch(char *lds, const Heads &hd, const Readback<OUT_DT> &rb, const float (&hv)[8], int64_t n_boards,
                                            int64_t tile0, int64_t tstride, int ntiles, int tid, long long *stamps) {
#define MZ_HSTAMP(k) do { if (stamps && threadIdx.x == 0) stamps[k] = (long long)__builtin_readcyclecounter(); } while (0)
    float *hid = reinterpret_cast<float *>(lds);
    uint16_t *fph = reinterpret_cast<uint16_t *>(lds + 32768), *fpl = reinterpret_cast<uint16_t *>(lds + 37376);   // [16][FROW] bf16 = 4,352 B each
    float *fv = reinterpret_cast<float *>(lds + 41984);                                                            // [16][128] f32
    uint16_t *h1h = reinterpret_cast<uint16_t *>(lds + 50176), *h1l = reinterpret_cast<uint16_t *>(lds + 54784);
    float *logit = hid;   // [16][80]: the hid rows are dead once the 1x1 convs are through (two barriers before the logits are written)
    float *vsum = reinterpret_cast<float *>(lds + 59392);
    float *prm = vsum + HB;   // small parameters: pc_w 16 | pc_b 2 | vc_w 16 | vc_b 2 (pad to 40) | p1_b 128 | p2_b 65 (pad 72) | v1_w 128 | v1_b 1
    float *s_pc_w = prm, *s_pc_b = prm + 16, *s_vc_w = prm + 18, *s_vc_b = prm + 34, *s_p1_b = prm + 40, *s_p2_b = prm + 168, *s_v1_w = prm + 240,
          *s_v1_b = prm + 368;
    static_assert(59392 + (HB + 512) * 4 <= ACT_BYTES && HB * FROW * 2 <= 4608, "the heads stay inside the X image region");
    const int w = tid >> 6, lane = tid & 63, n16 = lane & 15, kc = lane >> 4;
    const int grp = tid >> 7, k = tid & 127;
    const int nb = 8 * ntiles;                 // slots in use
    auto board_of = [&](int s) { return (tile0 + (int64_t)(s >> 3) * tstride) * TILE_BOARDS + (s & 7); };
    MZ_HSTAMP(19);
    // no barrier needed before these LDS writes: layer_last ended with one, and nothing has touched LDS since
    {   // earlier tile: the readback registers; unused slots are zero-filled
        constexpr int NW = Readback<OUT_DT>::NW;
#pragma unroll
        for (int r = 0; r < Readback<OUT_DT>::R; ++r) {
            const int i = tid + 512 * r;
            const uint32_t v = (i / NW) < nb - 8 ? rb.v[r] : 0u;
            if (OUT_DT == MZ_F32) hid[i] = __uint_as_float(v);
            else *reinterpret_cast<float2 *>(hid + 2 * i) = make_float2(bf16_up(v & 0xFFFFu), bf16_up(v >> 16));
        }
        for (int i = 8 * 512 + tid; i < HB * 512; i += 512) hid[i] = 0.0f;   // slots 8..15 (overwritten below when the batch is full)
    }
    // every global read of the heads' weights is requested here, BEHIND the staging above (the memory queue is in order: a
    // wait for the readback must not also wait for these); the matrices are consumed two phases later
    HeadRegs hr;
    heads_prefetch(hd, tid, hr);
    prm[tid] = rb.pv;
    __syncthreads();
    {   // the batch's last tile, from layer_last's registers: wave -> board, lane -> cell, planes 0..7
        float *dst = hid + (nb - 8 + w) * 512 + lane;
#pragma unroll
        for (int k = 0; k < 8; ++k) dst[k * 64] = hv[k];
    }
    __syncthreads();
    MZ_HSTAMP(21);
#pragma unroll
    for (int r = 0; r < 4; ++r) {   // 1x1 convs 8 -> 2 + ReLU, flattened channel-major (nn.Flatten of (2,8,8)); thread group = board 4 r + grp
        const int s = 4 * r + grp;
        const int c2 = k >> 6, pos = k & 63;
        float ap = s_pc_b[c2], av = s_vc_b[c2];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float xv = hid[s * 512 + c * 64 + pos];
            ap += s_pc_w[c2 * 8 + c] * xv;
            av += s_vc_w[c2 * 8 + c] * xv;
        }
        uint32_t hi, lo;
        split_bf16(fmaxf(ap, 0.0f), hi, lo);
        fph[s * FROW + k] = (uint16_t)hi;
        fpl[s * FROW + k] = (uint16_t)lo;
        fv[s * 128 + k] = fmaxf(av, 0.0f);
    }
    __syncthreads();
    MZ_HSTAMP(22);
    {   // Linear(128,128) + ReLU: wave w = output rows 16 w .. + 15, all 16 boards; D: lane -> board n16, rows 16 w + 4 kc + 0..3
        bf16x8 bh[4], bl[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bh[ks] = *reinterpret_cast<const bf16x8 *>(fph + n16 * FROW + 32 * ks + 8 * kc);
            bl[ks] = *reinterpret_cast<const bf16x8 *>(fpl + n16 * FROW + 32 * ks + 8 * kc);
        }
        acc16_t acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hr.a1[ks], bh[ks], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hr.a1[ks], bl[ks], acc, 0, 0, 0);
        }
        uint32_t hi[4], lo[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) split_bf16(fmaxf(acc[q] + s_p1_b[16 * w + 4 * kc + q], 0.0f), hi[q], lo[q]);
        *reinterpret_cast<uint2 *>(h1h + n16 * FROW + 16 * w + 4 * kc) = make_uint2(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
        *reinterpret_cast<uint2 *>(h1l + n16 * FROW + 16 * w + 4 * kc) = make_uint2(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16));
        // value Linear(128,1): wave w = boards w and w + 8
        const float v0 = wave_sum_dpp(s_v1_w[lane] * fv[w * 128 + lane] + s_v1_w[lane + 64] * fv[w * 128 + lane + 64]);
        const float v1 = wave_sum_dpp(s_v1_w[lane] * fv[(w + 8) * 128 + lane] + s_v1_w[lane + 64] * fv[(w + 8) * 128 + lane + 64]);
        if (lane == 0) {
            vsum[w] = v0;
            vsum[w + 8] = v1;
        }
    }
    __syncthreads();
    if (w < 5) {   // Linear(128,65): output rows 16 w .. + 15 (rows past 64 are padding)
        bf16x8 bh[4], bl[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bh[ks] = *reinterpret_cast<const bf16x8 *>(h1h + n16 * FROW + 32 * ks + 8 * kc);
            bl[ks] = *reinterpret_cast<const bf16x8 *>(h1l + n16 * FROW + 32 * ks + 8 * kc);
        }
        acc16_t acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hr.a2[ks], bh[ks], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hr.a2[ks], bl[ks], acc, 0, 0, 0);
        }
        *reinterpret_cast<f32x4 *>(logit + n16 * 80 + 16 * w + 4 * kc) = f32x4{acc[0], acc[1], acc[2], acc[3]};
    }
    __syncthreads();
    MZ_HSTAMP(23);
    {   // softmax over 65 + tanh: wave w = boards w and w + 8, the two chains interleaved
        float x[2], x64[2], m[2], e[2], e64[2], sum[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int s = w + 8 * h;
            x[h] = s_p2_b[lane] + logit[s * 80 + lane];
            x64[h] = s_p2_b[64] + logit[s * 80 + 64];
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) m[h] = wave_max_dpp(fmaxf(x[h], x64[h]));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            e[h] = expf(x[h] - m[h]);
            e64[h] = expf(x64[h] - m[h]);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) sum[h] = wave_sum_dpp(e[h]) + e64[h];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int s = w + 8 * h;
            const int64_t b = board_of(s);
            if (s >= nb || b >= n_boards) continue;
            float *pol = hd.policy + b * MZ_ACTIONS;
            pol[lane] = e[h] / sum[h];
            if (lane == 0) {
                pol[64] = e64[h] / sum[h];
                hd.value[b] = tanhf(vsum[s] + s_v1_b[0]);
            }
        }
    }
    // no trailing barrier: everything above lives in the X image region, which the next tile writes only behind its
    // first layer's own __syncthreads()
}

// one 8-board tile through the convolution tower: first layer, three residual blocks, last layer, hidden state stored
template <int OUT_DT>
__device__ __forceinline__ void tower_tile(char *lds, const Tower &tw_, const Heads &hd, const Route &rt, const uint16_t *__restrict__ x,
                                           void *__restrict__ hidden, int64_t n_boards, int64_t tile, int &tid, Readback<OUT_DT> &rb, int nprev, int64_t tstride,
                                           float (&hv)[8]) {
    Tower tw = tw_;
    if (tw.stamps) tw.stamps += tile * 24;   // this tile's row of the diagnostics array
    const uint32_t vm = halo_mask(tid & 31);
    const int64_t board0 = tile * TILE_BOARDS;
    MZ_STAMP(tw, 0);
    if (tw.stamps && threadIdx.x == 0) tw.stamps[17] = (long long)__builtin_amdgcn_s_memrealtime();
    mid_prologue(lds, tw.w_mid, tid);
    f32x4 bv[4];   // bias of the first mid layer in the K loop's accumulator layout (see layer_mid)
#pragma unroll
    for (int i = 0; i < 4; ++i) bv[i] = *reinterpret_cast<const f32x4 *>(tw.b_mid + 64 * ((tid >> 6) & 1) + 16 * i + 4 * ((tid & 63) >> 4));
    layer_first(lds, tw, rt, x, board0, n_boards, tid, vm);
    MZ_STAMP(tw, 1);
    MZ_STAMP(tw, 2);
    acc16_t acc[4][8];   // the K loops' accumulators: each layer's epilogue leaves the next layer's starting values in them
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = bv[i];
#pragma unroll 1
    for (int blk = 0; blk < 3; ++blk) {
        const uint4 *w0 = tw.w_mid + (int64_t)(2 * blk) * N_STAGES * 512;
        const uint4 *w1 = tw.w_mid + (int64_t)(2 * blk + 1) * N_STAGES * 512;
        layer_mid<0>(lds, w0, nullptr, acc, tw.b_mid + (2 * blk + 1) * 128, tid, tw, 3 + 4 * blk);                       // t = relu(conv(x)); acc = x + bias
        MZ_STAMP(tw, 4 + 4 * blk);
        layer_mid<1>(lds, w1, blk == 2 ? tw.w_last : nullptr, acc, blk < 2 ? tw.b_mid + (2 * blk + 2) * 128 : nullptr, tid, tw, 5 + 4 * blk);   // x = relu(x + conv(t))
        MZ_STAMP(tw, 6 + 4 * blk);
    }
    asm volatile("" : "+v"(tid));   // lane-derived addresses of the phases below are computed HERE, not held across the K loops
    if (nprev >= 0) readback_issue<OUT_DT>(rb, hd, hidden, rt, tile - nprev * tstride, tstride, nprev, tid);   // the heads follow this tile
    layer_last<OUT_DT>(lds, tw, tid, hv);
    MZ_STAMP(tw, 15);
    // a tile that the heads follow stores its hidden state AFTER them (its stores would otherwise sit in the memory queue in
    // front of the heads' weight loads, and a wait for those would wait for the stores' completion too)
    if (nprev < 0) store_hidden<OUT_DT>(hv, hidden, rt, board0, n_boards, tid);
}

// Persistent grid: one workgroup per CU walks its tiles (tile = blockIdx.x, += gridDim.x), which saves the workgroup
// turnovers a 1024-workgroup launch pays on every CU (all 160 KB of LDS per workgroup: never two resident at once).
// The arguments are NOT kept in registers across the tile loop: each iteration re-reads them from the kernarg segment
// (scalar loads, cached) through a pointer the compiler cannot see through, and the lane id is laundered the same way, so no
// address or pointer computed for one tile stays live into the K loops of the next (a plain loop spills: 20 pointers and
// the per-lane LDS addresses of every phase would be hoisted and held).
struct KArgs { Tower tw; Heads hd; const uint16_t *x; void *hidden; int64_t n_boards; Route rt; };
typedef const __attribute__((address_space(4))) char *kptr_t;
template <class T>
__device__ __forceinline__ T karg(kptr_t base, size_t off) { return *reinterpret_cast<const __attribute__((address_space(4))) T *>(base + off); }

template <int OUT_DT>
__global__ void __launch_bounds__(512) k_tower(KArgs) {
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    int in_batch = 0;   // tiles of this workgroup whose heads are still to run
#pragma unroll 1
    for (int64_t tile = blockIdx.x;; tile += gridDim.x) {
        kptr_t ka = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int64_t n_boards = karg<int64_t>(ka, offsetof(KArgs, n_boards));
        if (tile * TILE_BOARDS >= n_boards) break;
        const Tower tw{karg<const uint4 *>(ka, offsetof(KArgs, tw.w_first)), karg<const float *>(ka, offsetof(KArgs, tw.b_first)),
                       karg<const uint4 *>(ka, offsetof(KArgs, tw.w_mid)),   karg<const float *>(ka, offsetof(KArgs, tw.b_mid)),
                       karg<const uint4 *>(ka, offsetof(KArgs, tw.w_last)),  karg<const float *>(ka, offsetof(KArgs, tw.b_last)),
                       karg<int>(ka, offsetof(KArgs, tw.cin)),               karg<long long *>(ka, offsetof(KArgs, tw.stamps))};
        const Heads hd{karg<const float *>(ka, offsetof(KArgs, hd.pc_w)),    karg<const float *>(ka, offsetof(KArgs, hd.pc_b)),
                       karg<const uint16_t *>(ka, offsetof(KArgs, hd.p1_w)), karg<const float *>(ka, offsetof(KArgs, hd.p1_b)),
                       karg<const uint16_t *>(ka, offsetof(KArgs, hd.p2_w)), karg<const float *>(ka, offsetof(KArgs, hd.p2_b)),
                       karg<const float *>(ka, offsetof(KArgs, hd.vc_w)),    karg<const float *>(ka, offsetof(KArgs, hd.vc_b)),
                       karg<const float *>(ka, offsetof(KArgs, hd.v1_w)),    karg<const float *>(ka, offsetof(KArgs, hd.v1_b)),
                       karg<float *>(ka, offsetof(KArgs, hd.value)),         karg<float *>(ka, offsetof(KArgs, hd.policy))};
        // the heads run once per batch of two tiles (heads_batch): after this tile if it is the batch's second or the workgroup's last
        const bool heads_now = hd.policy != nullptr && (in_batch == 1 || (tile + gridDim.x) * TILE_BOARDS >= n_boards);
        void *hidden = karg<void *>(ka, offsetof(KArgs, hidden));
        const Route rt{karg<const uint16_t *>(ka, offsetof(KArgs, rt.src_pool)), karg<const int32_t *>(ka, offsetof(KArgs, rt.src_slot)),
                       karg<const int32_t *>(ka, offsetof(KArgs, rt.dst_slot)), karg<int64_t>(ka, offsetof(KArgs, rt.dst_stride))};
        Readback<OUT_DT> rb;
        float hv[8];
        tower_tile<OUT_DT>(lds, tw, hd, rt, karg<const uint16_t *>(ka, offsetof(KArgs, x)), hidden, n_boards, tile, tid, rb, heads_now ? in_batch : -1, gridDim.x, hv);
        ++in_batch;
        if (heads_now) {
            asm volatile("" : "+v"(tid));
            heads_batch<OUT_DT>(lds, hd, rb, hv, n_boards, tile - (int64_t)(in_batch - 1) * gridDim.x, gridDim.x, in_batch, tid,
                                tw.stamps ? tw.stamps + tile * 24 : nullptr);
            store_hidden<OUT_DT>(hv, hidden, rt, tile * TILE_BOARDS, n_boards, tid);
            in_batch = 0;
        }
        if (tw.stamps && threadIdx.x == 0) {   // diagnostics: end of this tile (and of its batch's heads, when they ran here)
            tw.stamps[tile * 24 + 16] = (long long)__builtin_readcyclecounter();
            tw.stamps[tile * 24 + 18] = (long long)__builtin_amdgcn_s_memrealtime();
        }
        // between tiles no further barrier is needed: layer_last ends with one, and heads_batch stays inside the X image region
    }
}

// ---- training: ONE 3x3, 128 -> 128 convolution on fp32 NCHW tensors, through the K loop above ------------------------------
// What update_weights' forward and backward (main.py:247-266) run for the towers' residual blocks: y = conv(x) + bias, or --
// with the flipped/transposed pack -- the data gradient of the same convolution.  One workgroup per 8-board tile: the input is
// rounded to bf16 into the [pos][ch] image (thread = position; 8 channels per 16-byte chunk), the weights stream through the
// ring, and the fp32 accumulators go straight back to NCHW (lane = position, so a store instruction writes 16-cell runs).
struct ConvArgs { const uint4 *w; const float *bias; const float *x; float *y; int64_t n_boards; int relu; const float *skip; const float *mask; int x_part; };

// Epilogue (EPI bit 0: skip, bit 1: mask), all on (n,128,8,8) f32 NCHW tensors: y = [relu]( [mask > 0 ?] (acc [+ skip]) : 0 ).
// That is the residual block's forward tail relu(h + conv(t)) and, in the backward pass, the ReLU gates and the skip path's
// gradient: g_t = dgrad(g_pre) * (t > 0);  g_pre_prev = (g_pre + dgrad(g_t)) * (y_prev > 0) -- no elementwise pass of their own.
// X3 (x_part = 2, 4-board tiles only): split precision in ONE launch -- x goes into a hi and a lo image, a.w is the stream
// [36 stages of W hi][36 of W lo][4 of padding], W_hi meets both images in one K loop (layer_mid's DUAL form), W_lo x_hi follows.
// NI = 2 (with NJ = 4): 2-board tiles, four waves across the output channels -- for batches where 4-board tiles leave CUs idle
template <int NJ, int EPI, bool X3 = false, int NI = 4>   // NJ * NI / 4 boards per workgroup: 8, 4 or 2
__global__ void __launch_bounds__(512) k_conv3x3(ConvArgs a) {
    static_assert((!X3 || NJ == 4) && (NI == 4 || (NI == 2 && NJ == 4)), "tile shapes");
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    constexpr int BOARDS = NJ * NI / 4, POS = 64 * BOARDS, SPLIT = 512 / POS, CHUNKS = 16 / SPLIT;   // SPLIT threads share a position's 16 chunks
    const int tid = threadIdx.x;
    const int64_t board0 = (int64_t)blockIdx.x * BOARDS;
    mid_prologue(lds, a.w, tid);
    {   // input: thread = position q of the tile, CHUNKS chunks of 8 channels
        const int q = tid % POS, cbase = (tid / POS) * CHUNKS, key = act_key(q);
        const int64_t b = board0 + (q >> 6);
        char *row = lds + OFF_X + q * 256;
        if (b < a.n_boards) {
            const float *src = a.x + b * 8192 + cbase * 512 + (q & 63);
#pragma unroll 1
            for (int c0 = 0; c0 < CHUNKS; c0 += 4) {
                float v[4][8];
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[c][k] = src[((c0 + c) * 8 + k) * 64];
                if (X3) {   // both images: hi here, the remainders below
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        uint32_t hi[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            hi[k] = pack_bf16(v[c][2 * k], v[c][2 * k + 1]);
                            v[c][2 * k] -= bf16_up(hi[k] & 0xFFFFu);
                            v[c][2 * k + 1] -= __uint_as_float(hi[k] & 0xFFFF0000u);
                        }
                        *reinterpret_cast<uint4 *>(row + (((cbase + c0 + c) ^ key) << 4)) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
                    }
                } else if (a.x_part) {   // the rounding remainder x - bf16(x) (exact in f32): the second term of a split-precision product
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int k = 0; k < 8; k += 2) {
                            const uint32_t hi = pack_bf16(v[c][k], v[c][k + 1]);
                            v[c][k] -= bf16_up(hi & 0xFFFFu);
                            v[c][k + 1] -= __uint_as_float(hi & 0xFFFF0000u);
                        }
                }
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    *reinterpret_cast<uint4 *>(row + (X3 ? X3_LO : 0) + (((cbase + c0 + c) ^ key) << 4)) =
                        make_uint4(pack_bf16(v[c][0], v[c][1]), pack_bf16(v[c][2], v[c][3]), pack_bf16(v[c][4], v[c][5]), pack_bf16(v[c][6], v[c][7]));
            }
        } else {
#pragma unroll
            for (int c = 0; c < CHUNKS; ++c) {
                *reinterpret_cast<uint4 *>(row + ((cbase + c) << 4)) = make_uint4(0u, 0u, 0u, 0u);
                if (X3) *reinterpret_cast<uint4 *>(row + X3_LO + ((cbase + c) << 4)) = make_uint4(0u, 0u, 0u, 0u);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the prologue's DMA pieces
    __syncthreads();
    const int w = tid >> 6, lane = tid & 63, kc = lane >> 4, p16 = col_cell(lane & 15);
    constexpr int CW = 8 / NI;   // waves across the output channels (layer_mid)
    const int cobase = 16 * NI * (w & (CW - 1)), posbase = 16 * NJ * (w / CW);
    acc16_t acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const f32x4 bv = a.bias ? *reinterpret_cast<const f32x4 *>(a.bias + cobase + 16 * i + 4 * kc) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = bv;
    }
    const Tower none{};
    if constexpr (X3) {
        layer_mid<2, 8, true, NI>(lds, a.w, nullptr, acc, nullptr, tid, none, 0);                      // W_hi x_hi + W_hi x_lo
        layer_mid<2, 4, false, NI>(lds, a.w + N_STAGES * 512, nullptr, acc, nullptr, tid, none, 0);    // W_lo x_hi
    } else {
        layer_mid<2, NJ, false, NI>(lds, a.w, nullptr, acc, nullptr, tid, none, 0);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = posbase + 16 * j + p16;
        const int64_t b = board0 + (q >> 6);
        if (b < a.n_boards) {
            const int64_t o = (b * 128 + cobase + 4 * kc) * 64 + (q & 63);
            float sk[NI][4], mk[NI][4];
            if (EPI & 1) {
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) sk[i][e] = a.skip[o + (16 * i + e) * 64];
            }
            if (EPI & 2) {
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) mk[i][e] = a.mask[o + (16 * i + e) * 64];
            }
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[i][j][e];
                    if (EPI & 1) v += sk[i][e];
                    if (EPI & 2) v = mk[i][e] > 0.0f ? v : 0.0f;
                    a.y[o + (16 * i + e) * 64] = a.relu ? fmaxf(v, 0.0f) : v;
                }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the K loop's last prefetches (padding stages) land before the wave ends
}

// ---- fp32-grade serving: the THREE RESIDUAL BLOCKS of a tower (network.py:35-47, six 3x3 128 -> 128 convolutions) in split
// precision, LDS-resident.  k_conv3x3 in split precision is three launches per convolution, each a pass over fp32 tensors in
// HBM (400 MB per launch at 4096 boards: the launches are HBM-bound).  Here a workgroup carries a 4-board tile through all six
// layers: TWO [256 pos][128 ch] bf16 images (hi = bf16(v) at 0, lo = bf16(v - hi) at 64 KB) updated in place, three K loops
// per layer into ONE set of fp32 accumulators -- W_hi (x_hi | x_lo) in one K loop, then W_lo x_hi: every product exact, fp32 sums, what is
// dropped is W_lo x_lo ~ 2^-16 of a term -- and the block input x kept in fp32 REGISTERS for the skip path (64 per lane: the
// 4-board tile's 64 accumulators leave room for it), so the residual sum sees no rounding at all.  HBM traffic: x once in, y
// once out.  The weight stream is one array of 12 x 36 stages (per layer: hi, lo; + 4 stages of padding) that runs
// through the ring without a break, like the serving kernel's.
// FULL (mz_tower_x3_forward): the whole tower -- the first convolution (cin -> 128, network.py:53 / :68) in front and the last one
// (128 -> 8, network.py:55 / :69) behind the blocks, in the same split precision: x = the (n, cin, 8, 8) input planes, y = the
// (n, 8, 8, 8) hidden state; no 128-channel tensor touches HBM at all.
struct ResArgs {
    const uint4 *w; const float *bias; const float *x; float *y; int64_t n_boards; int relu_in; long long *stamps;
    const uint16_t *ends; const float *b_first; const float *b_last; int cin;   // FULL only
    // FULL, optional (value != null): the prediction heads (network.py:79-111) behind the tower in plain fp32 on the VALU -- 25k MACs
    // per board are nothing beside the tower's 3 x 58 M -- with the Linear matrices as f32 [out][in] (nn.Linear's layout)
    const float *pc_w, *pc_b, *p1_w, *p1_b, *p2_w, *p2_b, *vc_w, *vc_b, *v1_w, *v1_b;
    float *value, *policy;
};
constexpr int X3_HEADS_OFF = 81920;   // heads scratch in LDS, behind the last layer's partial sums (4 x P_WAVE_BYTES = 73,728)
static_assert(4 * P_WAVE_BYTES <= X3_HEADS_OFF && X3_HEADS_OFF + (2048 + 3 * 512 + 4 * 80) * 4 <= LDS_BYTES, "heads scratch");
// ends = the two small convolutions' MFMA fragments, read straight from global memory (L2-resident, the same for every tile):
//   first: [2 parts][5 k-steps][128 co][32] bf16, k = 16 t2 + c -> tap 2 ks + t2, input plane c (zero beyond tap 8 / plane cin - 1)
//   last : [2 parts][4 quarters][80 rows][32 ci] bf16, row = tap * 8 + plane (72 real)          part 0 = bf16 rounding, 1 = remainder
constexpr int X3_FIRST_ELEMS = 5 * 128 * 32, X3_LAST_ELEMS = 4 * LAST_ROWS * 32, X3_ENDS_ELEMS = 2 * X3_FIRST_ELEMS + 2 * X3_LAST_ELEMS;
constexpr int X3_SEGMENTS = 12;      // weight segments per tile: per layer the bf16 rounding (hi), then its remainder (lo)

__device__ __forceinline__ void x3_store_images(char *lds, const acc16_t (&v)[4][4], int posbase, int p16, int kc, int cobase) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = posbase + 16 * j + p16;
        char *row = lds + q * 256 + (kc & 1) * 8;
        const int key = act_key(q);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int chunk = (cobase >> 3) + 2 * i + (kc >> 1);   // channels cobase + 16 i + 4 kc + 0..3
            char *p = row + ((chunk ^ key) << 4);
            uint2 hi, lo;
            hi.x = pack_bf16(v[i][j][0], v[i][j][1]);
            hi.y = pack_bf16(v[i][j][2], v[i][j][3]);
            // the rounding remainders: exact in f32, then rounded to bf16 themselves
            lo.x = pack_bf16(v[i][j][0] - bf16_up(hi.x & 0xFFFFu), v[i][j][1] - __uint_as_float(hi.x & 0xFFFF0000u));
            lo.y = pack_bf16(v[i][j][2] - bf16_up(hi.y & 0xFFFFu), v[i][j][3] - __uint_as_float(hi.y & 0xFFFF0000u));
            *reinterpret_cast<uint2 *>(p) = hi;
            *reinterpret_cast<uint2 *>(p + X3_LO) = lo;
        }
    }
}

template <bool FULL>
__global__ void __launch_bounds__(512) k_restower_x3(ResArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    const int tid = threadIdx.x;
    const int64_t board0 = (int64_t)blockIdx.x * 4;
    Tower none{};   // diagnostics only: [tiles][32] cycle stamps at the phase boundaries (tools/restower_stamps.py)
    none.stamps = a.stamps ? a.stamps + (int64_t)blockIdx.x * 32 : nullptr;
    if (none.stamps && tid == 0) none.stamps[30] = (long long)__builtin_amdgcn_s_memrealtime();
    MZ_STAMP(none, 0);
    mid_prologue(lds, a.w, tid);
    const int w = tid >> 6, lane = tid & 63, kc = lane >> 4, p16 = col_cell(lane & 15);
    const int cobase = 64 * (w & 1), posbase = 64 * (w >> 1);   // wave = co half x board
    const int64_t b = board0 + (w >> 1);
    const bool live = b < a.n_boards;
    const int64_t o = (b * 128 + cobase + 4 * kc) * 64 + p16;   // + 16 j cells, + (16 i + e) channels
    acc16_t xk[4][4], acc[4][4];
    if (FULL) {
        // first convolution.  Input planes -> hi / lo [256 pos][16 planes] bf16 at 0 / 8 KB (inside the images' region, not yet
        // written); one k-step of the 16x16x32 MFMA = two taps x 16 planes, so its output lands in the K loops' accumulator layout
        {
            const int q = tid & 255, half = tid >> 8;
            const int64_t bb = board0 + (q >> 6);
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = (bb < a.n_boards && half * 8 + k < a.cin) ? a.x[(bb * a.cin + half * 8 + k) * 64 + (q & 63)] : 0.0f;
            uint32_t hi[4], lo[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                hi[k] = pack_bf16(v[2 * k], v[2 * k + 1]);
                lo[k] = pack_bf16(v[2 * k] - bf16_up(hi[k] & 0xFFFFu), v[2 * k + 1] - __uint_as_float(hi[k] & 0xFFFF0000u));
            }
            *reinterpret_cast<uint4 *>(lds + q * 32 + half * 16) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
            *reinterpret_cast<uint4 *>(lds + 8192 + q * 32 + half * 16) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 bv = a.b_first ? *reinterpret_cast<const f32x4 *>(a.b_first + cobase + 16 * i + 4 * kc) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = bv;
        }
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) {
            const int tap = 2 * ks + (kc >> 1);   // this lane's half of the k-step
            const int ky = tap / 3, kx = tap - 3 * ky;
            bf16x8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint16_t *src = a.ends + (ks * 128 + cobase + 16 * i + (lane & 15)) * 32 + 8 * kc;
                ah[i] = *reinterpret_cast<const bf16x8 *>(src);
                al[i] = *reinterpret_cast<const bf16x8 *>(src + X3_FIRST_ELEMS);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cell = 16 * j + p16;
                const bool ok = tap < 9 && (unsigned)((cell & 7) + kx - 1) < 8u && (unsigned)((cell >> 3) + ky - 1) < 8u;
                const int addr = ok ? (posbase + cell + (ky - 1) * 8 + (kx - 1)) * 32 + (kc & 1) * 16 : OFF_Z;
                bh[j] = lds_frag(lds + addr);
                bl[j] = lds_frag(lds + addr + 8192);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
        // every wave has read the staged planes: the images may be written over them
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) xk[i][j][e] = fmaxf(acc[i][j][e], 0.0f);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = live ? a.x[o + 16 * j + (16 * i + e) * 64] : 0.0f;
                    xk[i][j][e] = a.relu_in ? fmaxf(v, 0.0f) : v;
                }
    }
    x3_store_images(lds, xk, posbase, p16, kc, cobase);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the prologue's DMA pieces
    __syncthreads();
    MZ_STAMP(none, 1);
    const uint4 *wp = a.w;
#pragma unroll 1
    for (int l = 0; l < 6; ++l) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(a.bias + l * 128 + cobase + 16 * i + 4 * kc);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (l & 1) ? xk[i][j] + bv : bv;   // a block's second convolution adds onto x
        }
        layer_mid<2, 8, true>(lds, wp, nullptr, acc, nullptr, tid, none, 2 + 3 * l);   // W_hi x_hi + W_hi x_lo
        layer_mid<2, 4>(lds, wp + N_STAGES * 512, nullptr, acc, nullptr, tid, none, 3 + 3 * l);   // W_lo x_hi
        wp += 2 * N_STAGES * 512;
        // in place: every wave must be through its reads of both images (the "tap 9" prefetches included)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = fmaxf(acc[i][j][e], 0.0f);
                if (l & 1) xk[i][j] = acc[i][j];
            }
        if (FULL || l < 5) {
            x3_store_images(lds, acc, posbase, p16, kc, cobase);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        MZ_STAMP(none, 4 + 3 * l);
    }
    if (FULL) {
        // (the lane's indices are derived again from an opaque copy of the thread id: kept live across the K loops they were spilled)
        int t2 = threadIdx.x;
        asm volatile("" : "+v"(t2));
        const int tid = t2, w = tid >> 6, lane = tid & 63, kc = lane >> 4, p16 = col_cell(lane & 15), posbase = 64 * (w >> 1);
        const int64_t b = board0 + (w >> 1);
        const bool live = b < a.n_boards;
        // last convolution, as layer_last: the nine taps in the M dimension (P[tap * 8 + plane][pos], K = 128), then every lane sums
        // the nine shifted taps of one cell.  Even waves take their board, odd waves only keep the barriers.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the K loop's last prefetches (padding stages): the ring is dead
        const bool mine = (w & 1) == 0;
        bf16x8 fh[4][4], fl[4][4];
        if (mine) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = posbase + 16 * j + p16;
#pragma unroll
                for (int kq = 0; kq < 4; ++kq) {
                    const char *p = lds + q * 256 + (((4 * kq + kc) ^ act_key(q)) << 4);
                    fh[j][kq] = lds_frag(p);
                    fl[j][kq] = lds_frag(p + X3_LO);
                }
            }
        }
        // every wave holds its fragments in registers: the images may be overwritten by the partial sums
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        float *P = reinterpret_cast<float *>(lds + (w >> 1) * P_WAVE_BYTES);
        if (mine) {
            const uint16_t *wl = a.ends + 2 * X3_FIRST_ELEMS;
#pragma unroll
            for (int m = 0; m < 5; ++m) {
                bf16x8 ah[4], al[4];
#pragma unroll
                for (int kq = 0; kq < 4; ++kq) {
                    const uint16_t *src = wl + (kq * LAST_ROWS + 16 * m + (lane & 15)) * 32 + 8 * kc;
                    ah[kq] = *reinterpret_cast<const bf16x8 *>(src);
                    al[kq] = *reinterpret_cast<const bf16x8 *>(src + X3_LAST_ELEMS);
                }
                const int tap = 2 * m + (kc >> 1);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc16_t pm = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int kq = 0; kq < 4; ++kq) {
                        pm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[kq], fh[j][kq], pm, 0, 0, 0);
                        pm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[kq], fl[j][kq], pm, 0, 0, 0);
                        pm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[kq], fh[j][kq], pm, 0, 0, 0);
                    }
                    if (tap < 9) *reinterpret_cast<f32x4 *>(P + ((tap * 2 + (kc & 1)) * 64 + 16 * j + p16) * 4) = pm;
                }
            }
            // gather: lane -> cell of the wave's board, both plane halves (the LDS serves a wave in order: no barrier needed)
            const int q = lane, x = q & 7, y = q >> 3;
            f32x4 s0 = {0.0f, 0.0f, 0.0f, 0.0f}, s1 = {0.0f, 0.0f, 0.0f, 0.0f};
            if (a.b_last) {
                s0 = *reinterpret_cast<const f32x4 *>(a.b_last);
                s1 = *reinterpret_cast<const f32x4 *>(a.b_last + 4);
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap % 3;
                const bool ok = (unsigned)(x + kx - 1) < 8u && (unsigned)(y + ky - 1) < 8u;
                const int q2 = (q + (ky - 1) * 8 + (kx - 1)) & 63;
                const f32x4 p0 = *reinterpret_cast<const f32x4 *>(P + ((tap * 2 + 0) * 64 + q2) * 4);
                const f32x4 p1 = *reinterpret_cast<const f32x4 *>(P + ((tap * 2 + 1) * 64 + q2) * 4);
                if (ok) {
                    s0 += p0;
                    s1 += p1;
                }
            }
            float *hid = reinterpret_cast<float *>(lds + X3_HEADS_OFF);   // [4 boards][8 planes][64 cells]
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float hv = fmaxf(k < 4 ? s0[k & 3] : s1[k & 3], 0.0f);
                if (live) a.y[b * 512 + k * 64 + lane] = hv;
                if (a.value) hid[(w >> 1) * 512 + k * 64 + lane] = hv;
            }
        }
        if (a.value) {   // heads: thread = (board tid >> 7, feature / output tid & 127)
            float *hid = reinterpret_cast<float *>(lds + X3_HEADS_OFF);
            float *fp = hid + 2048, *fv = fp + 512, *h1 = fv + 512, *lg = h1 + 512;   // [4][128] x 3, logits [4][80]
            const int hb = tid >> 7, k = tid & 127;
            __syncthreads();
            {   // 1x1 convolutions 8 -> 2 + ReLU, flattened channel-major (nn.Flatten of (2,8,8))
                const int c2 = k >> 6, pos = k & 63;
                float ap = a.pc_b[c2], av = a.vc_b[c2];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float xv = hid[hb * 512 + c * 64 + pos];
                    ap += a.pc_w[c2 * 8 + c] * xv;
                    av += a.vc_w[c2 * 8 + c] * xv;
                }
                fp[tid] = fmaxf(ap, 0.0f);
                fv[tid] = fmaxf(av, 0.0f);
            }
            __syncthreads();
            {   // Linear(128,128) + ReLU
                float s = a.p1_b[k];
                const f32x4 *row = reinterpret_cast<const f32x4 *>(a.p1_w + k * 128);
                const f32x4 *f = reinterpret_cast<const f32x4 *>(fp + hb * 128);
#pragma unroll 8
                for (int i = 0; i < 32; ++i) {
                    const f32x4 wv = row[i], xv = f[i];
                    s += wv[0] * xv[0];
                    s += wv[1] * xv[1];
                    s += wv[2] * xv[2];
                    s += wv[3] * xv[3];
                }
                h1[tid] = fmaxf(s, 0.0f);
            }
            if (tid < 4) {   // value: Linear(128,1) + tanh
                float s = a.v1_b[0];
                for (int i = 0; i < 128; ++i) s += a.v1_w[i] * fv[tid * 128 + i];
                if (board0 + tid < a.n_boards) a.value[board0 + tid] = tanhf(s);
            }
            __syncthreads();
            if (k < MZ_ACTIONS) {   // Linear(128,65)
                float s = a.p2_b[k];
                const f32x4 *row = reinterpret_cast<const f32x4 *>(a.p2_w + k * 128);
                const f32x4 *f = reinterpret_cast<const f32x4 *>(h1 + hb * 128);
#pragma unroll 8
                for (int i = 0; i < 32; ++i) {
                    const f32x4 wv = row[i], xv = f[i];
                    s += wv[0] * xv[0];
                    s += wv[1] * xv[1];
                    s += wv[2] * xv[2];
                    s += wv[3] * xv[3];
                }
                lg[hb * 80 + k] = s;
            }
            __syncthreads();
            if (w < 4) {   // softmax over the 65 logits of board w: lane -> action, lane 0 also action 64
                const float l0 = lg[w * 80 + lane], l1 = lane == 0 ? lg[w * 80 + 64] : -INFINITY;
                const float m = wave_max_dpp(fmaxf(l0, l1));
                const float e0 = expf(l0 - m), e1 = lane == 0 ? expf(l1 - m) : 0.0f;
                const float tot = wave_sum_dpp(e0 + e1);
                if (board0 + w < a.n_boards) {
                    a.policy[(board0 + w) * MZ_ACTIONS + lane] = e0 / tot;
                    if (lane == 0) a.policy[(board0 + w) * MZ_ACTIONS + 64] = e1 / tot;
                }
            }
        }
    } else if (live) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) a.y[o + 16 * j + (16 * i + e) * 64] = xk[i][j][e];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the K loop's last prefetches (padding stages) land before the wave ends
    MZ_STAMP(none, 20);
    if (none.stamps && tid == 0) none.stamps[31] = (long long)__builtin_amdgcn_s_memrealtime();
}

// the six convolutions' weights (128,128,3,3) f32 -> k_restower_x3's stream: [12 x 36 + 4 stages][128 co][4 slots][8] bf16, segment
// 2 l = layer l's bf16 rounding, 2 l + 1 = its remainder; stage / row / slot layout as k_conv3x3_pack's forward image
struct ResPackArgs { const float *w[6]; uint16_t *out; };
__global__ void __launch_bounds__(256) k_restower_x3_pack(ResPackArgs a) {   // blockIdx.y = segment (12 = the padding)
    const int idx = blockIdx.x * 256 + threadIdx.x;   // (stage, row, slot, e)
    const int seg = blockIdx.y;
    if (seg == X3_SEGMENTS) {
        if (idx < 4 * 4096) a.out[(size_t)X3_SEGMENTS * N_STAGES * 4096 + idx] = 0;
        return;
    }
    const float *__restrict__ wsrc = a.w[seg / 2];
    const int e = idx & 7, slot = (idx >> 3) & 3, row = (idx >> 5) & 127, stage = idx >> 12;
    const int tap = stage >> 2, quarter = stage & 3, ky = tap / 3, kx = tap % 3;
    const int ch = quarter * 32 + ((slot ^ ((0x78 >> (((row >> 2) & 3) * 2)) & 3)) << 3) + e;
    float wf = wsrc[((row * 128 + ch) * 3 + ky) * 3 + kx];
    if (seg & 1) wf -= bf16_up(bf16_rne(wf));
    a.out[(size_t)seg * N_STAGES * 4096 + idx] = (uint16_t)bf16_rne(wf);
}

// the first (128, cin, 3, 3) and last (8, 128, 3, 3) convolutions' weights -> ResArgs::ends
__global__ void __launch_bounds__(256) k_tower_x3_pack_ends(const float *__restrict__ w_first, int cin, const float *__restrict__ w_last, uint16_t *__restrict__ out) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= X3_ENDS_ELEMS) return;
    float v = 0.0f;
    int part;
    if (idx < 2 * X3_FIRST_ELEMS) {
        part = idx / X3_FIRST_ELEMS;
        const int r = idx % X3_FIRST_ELEMS, k32 = r & 31, co = (r >> 5) & 127, ks = r >> 12;
        const int tap = 2 * ks + (k32 >> 4), c = k32 & 15;
        if (tap < 9 && c < cin) v = w_first[((co * cin + c) * 3 + tap / 3) * 3 + tap % 3];
    } else {
        const int r0 = idx - 2 * X3_FIRST_ELEMS;
        part = r0 / X3_LAST_ELEMS;
        const int r = r0 % X3_LAST_ELEMS, k32 = r & 31, row = (r >> 5) % LAST_ROWS, kq = (r >> 5) / LAST_ROWS;
        const int tap = row >> 3, plane = row & 7;
        if (row < 72) v = w_last[((plane * 128 + 32 * kq + k32) * 3 + tap / 3) * 3 + tap % 3];
    }
    if (part) v -= bf16_up(bf16_rne(v));
    out[idx] = (uint16_t)bf16_rne(v);
}

// conv weight (128 co, 128 ci, 3, 3) f32 -> the two stage images k_conv3x3 streams: [36 + 4 stages][128 rows][4 slots][8] bf16,
//   forward : stage (ky*3+kx)*4 + ci/32, row co, element = W[co][ci][ky][kx]
//   backward: stage (ky*3+kx)*4 + co/32, row ci, element = W[co][ci][2-ky][2-kx]   (the data gradient as a convolution of g)
// slot s of a row holds the 8-channel chunk s ^ {0,2,3,1}[(row >> 2) & 3]; the 4 trailing stages are zero (prefetch padding)
constexpr int PACK_MANY = 16;
struct PackArgs { const float *w[PACK_MANY]; uint16_t *fwd[PACK_MANY]; uint16_t *bwd[PACK_MANY]; int part; };

__global__ void __launch_bounds__(256) k_conv3x3_pack(PackArgs a) {   // blockIdx.y = which convolution
    const int idx = blockIdx.x * 256 + threadIdx.x;   // (stage, row, slot, e)
    if (idx >= 40 * 4096) return;
    const float *__restrict__ wsrc = a.w[blockIdx.y];
    const int e = idx & 7, slot = (idx >> 3) & 3, row = (idx >> 5) & 127, stage = idx >> 12;
    uint16_t f = 0, bk = 0;
    if (stage < 36) {
        const int tap = stage >> 2, quarter = stage & 3, ky = tap / 3, kx = tap % 3;
        const int ch = quarter * 32 + ((slot ^ ((0x78 >> (((row >> 2) & 3) * 2)) & 3)) << 3) + e;
        float wf = wsrc[((row * 128 + ch) * 3 + ky) * 3 + kx], wb = wsrc[((ch * 128 + row) * 3 + (2 - ky)) * 3 + (2 - kx)];
        if (a.part) {   // the rounding remainder w - bf16(w): exact in f32, then rounded to bf16 itself
            wf -= bf16_up(bf16_rne(wf));
            wb -= bf16_up(bf16_rne(wb));
        }
        f = (uint16_t)bf16_rne(wf);
        bk = (uint16_t)bf16_rne(wb);
    }
    a.fwd[blockIdx.y][idx] = f;
    a.bwd[blockIdx.y][idx] = bk;
}

}  // namespace

extern "C" int mz_conv3x3_pack_many(const float *const *w, void *const *packed_fwd, void *const *packed_bwd, int32_t n, int32_t part, void *stream) {
    MZ_REQUIRE(w && packed_fwd && packed_bwd, "mz_conv3x3_pack: null pointer");
    MZ_REQUIRE(n >= 1 && n <= PACK_MANY, "mz_conv3x3_pack_many: %d convolutions outside [1, %d]", n, PACK_MANY);
    MZ_REQUIRE(part == 0 || part == 1, "mz_conv3x3_pack: part must be 0 (the bf16 rounding) or 1 (its remainder)");
    PackArgs a{};
    for (int i = 0; i < n; ++i) {
        MZ_REQUIRE(w[i] && packed_fwd[i] && packed_bwd[i], "mz_conv3x3_pack: null pointer (convolution %d)", i);
        MZ_REQUIRE((((uintptr_t)packed_fwd[i] | (uintptr_t)packed_bwd[i]) & 15) == 0, "mz_conv3x3_pack: outputs must be 16-byte aligned");
        a.w[i] = w[i];
        a.fwd[i] = (uint16_t *)packed_fwd[i];
        a.bwd[i] = (uint16_t *)packed_bwd[i];
    }
    a.part = (int)part;
    hipLaunchKernelGGL(k_conv3x3_pack, dim3(40 * 4096 / 256, (unsigned)n), dim3(256), 0, (hipStream_t)stream, a);
    MZ_LAUNCH_CHECK("k_conv3x3_pack");
    return MZ_OK;
}

extern "C" int mz_conv3x3_pack_part(const float *w, void *packed_fwd, void *packed_bwd, int32_t part, void *stream) {
    return mz_conv3x3_pack_many(&w, &packed_fwd, &packed_bwd, 1, part, stream);
}

extern "C" int mz_conv3x3_pack(const float *w, void *packed_fwd, void *packed_bwd, void *stream) {
    return mz_conv3x3_pack_part(w, packed_fwd, packed_bwd, 0, stream);
}

extern "C" int mz_conv3x3_fused(const void *packed, const float *bias, const float *x, int32_t x_part, const float *skip, const float *mask, float *y,
                               int64_t n, int32_t relu, void *stream) {
    MZ_REQUIRE(n >= 0, "mz_conv3x3_forward: n < 0");
    if (n == 0) return MZ_OK;
    MZ_REQUIRE(packed && x && y, "mz_conv3x3_forward: null pointer");
    MZ_REQUIRE((((uintptr_t)packed | (uintptr_t)bias) & 15) == 0, "mz_conv3x3_forward: packed weights and bias must be 16-byte aligned");
    MZ_REQUIRE(x != y, "mz_conv3x3_forward: in-place call");
    // 8-board tiles once they fill the chip; below that 4-board tiles: twice the workgroups on a less dense K loop
    int dev = 0, cus = 0;
    MZ_HIP(hipGetDevice(&dev));
    MZ_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    MZ_REQUIRE(x_part >= 0 && x_part <= 2, "mz_conv3x3_fused: x_part must be 0, 1 or 2");
    const bool small = x_part == 2 || (n + 7) / 8 < cus;
    const bool tiny = (n + 3) / 4 < cus;   // 2-board tiles: even 4-board tiles would leave CUs idle
    const int64_t n_tiles = tiny ? (n + 1) / 2 : small ? (n + 3) / 4 : (n + 7) / 8;
    MZ_REQUIRE(n_tiles <= 0x7fffffff, "mz_conv3x3_forward: too many boards");
    MZ_REQUIRE(mask != y, "mz_conv3x3_fused: mask aliases the output");   // skip may: every lane reads its own cells before it writes them
    const ConvArgs a{(const uint4 *)packed, bias, x, y, n, relu, skip, mask, (int)x_part};
    const dim3 grid((unsigned)n_tiles), block(512);
    const hipStream_t st = (hipStream_t)stream;
    switch ((tiny ? 16 : 0) | (x_part == 2 ? 8 : 0) | (small ? 4 : 0) | (skip ? 1 : 0) | (mask ? 2 : 0)) {
        case 20: hipLaunchKernelGGL((k_conv3x3<4, 0, false, 2>), grid, block, 0, st, a); break;
        case 21: hipLaunchKernelGGL((k_conv3x3<4, 1, false, 2>), grid, block, 0, st, a); break;
        case 22: hipLaunchKernelGGL((k_conv3x3<4, 2, false, 2>), grid, block, 0, st, a); break;
        case 23: hipLaunchKernelGGL((k_conv3x3<4, 3, false, 2>), grid, block, 0, st, a); break;
        case 28: hipLaunchKernelGGL((k_conv3x3<4, 0, true, 2>), grid, block, 0, st, a); break;
        case 29: hipLaunchKernelGGL((k_conv3x3<4, 1, true, 2>), grid, block, 0, st, a); break;
        case 30: hipLaunchKernelGGL((k_conv3x3<4, 2, true, 2>), grid, block, 0, st, a); break;
        case 31: hipLaunchKernelGGL((k_conv3x3<4, 3, true, 2>), grid, block, 0, st, a); break;
        case 12: hipLaunchKernelGGL((k_conv3x3<4, 0, true>), grid, block, 0, st, a); break;
        case 13: hipLaunchKernelGGL((k_conv3x3<4, 1, true>), grid, block, 0, st, a); break;
        case 14: hipLaunchKernelGGL((k_conv3x3<4, 2, true>), grid, block, 0, st, a); break;
        case 15: hipLaunchKernelGGL((k_conv3x3<4, 3, true>), grid, block, 0, st, a); break;
        case 0: hipLaunchKernelGGL((k_conv3x3<8, 0>), grid, block, 0, st, a); break;
        case 1: hipLaunchKernelGGL((k_conv3x3<8, 1>), grid, block, 0, st, a); break;
        case 2: hipLaunchKernelGGL((k_conv3x3<8, 2>), grid, block, 0, st, a); break;
        case 3: hipLaunchKernelGGL((k_conv3x3<8, 3>), grid, block, 0, st, a); break;
        case 4: hipLaunchKernelGGL((k_conv3x3<4, 0>), grid, block, 0, st, a); break;
        case 5: hipLaunchKernelGGL((k_conv3x3<4, 1>), grid, block, 0, st, a); break;
        case 6: hipLaunchKernelGGL((k_conv3x3<4, 2>), grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL((k_conv3x3<4, 3>), grid, block, 0, st, a); break;
    }
    MZ_LAUNCH_CHECK("k_conv3x3");
    return MZ_OK;
}

extern "C" int mz_conv3x3_forward(const void *packed, const float *bias, const float *x, float *y, int64_t n, int32_t relu, void *stream) {
    return mz_conv3x3_fused(packed, bias, x, 0, nullptr, nullptr, y, n, relu, stream);
}

extern "C" int mz_restower_x3_pack(const float *const *w, void *packed, void *stream) {
    MZ_REQUIRE(w && packed, "mz_restower_x3_pack: null pointer");
    MZ_REQUIRE(((uintptr_t)packed & 15) == 0, "mz_restower_x3_pack: the output must be 16-byte aligned");
    ResPackArgs a{};
    for (int i = 0; i < 6; ++i) {
        MZ_REQUIRE(w[i], "mz_restower_x3_pack: null pointer (convolution %d)", i);
        a.w[i] = w[i];
    }
    a.out = (uint16_t *)packed;
    hipLaunchKernelGGL(k_restower_x3_pack, dim3(N_STAGES * 4096 / 256, X3_SEGMENTS + 1), dim3(256), 0, (hipStream_t)stream, a);
    MZ_LAUNCH_CHECK("k_restower_x3_pack");
    return MZ_OK;
}

extern "C" int mz_tower_x3_pack_ends(const float *w_first, int32_t cin, const float *w_last, void *ends, void *stream) {
    MZ_REQUIRE(w_first && w_last && ends, "mz_tower_x3_pack_ends: null pointer");
    MZ_REQUIRE(cin >= 1 && cin <= 16, "mz_tower_x3_pack_ends: cin %d outside [1, 16]", cin);
    MZ_REQUIRE(((uintptr_t)ends & 15) == 0, "mz_tower_x3_pack_ends: the output must be 16-byte aligned");
    hipLaunchKernelGGL(k_tower_x3_pack_ends, dim3((X3_ENDS_ELEMS + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_first, (int)cin, w_last, (uint16_t *)ends);
    MZ_LAUNCH_CHECK("k_tower_x3_pack_ends");
    return MZ_OK;
}

static int tower_x3_launch(const void *packed, const void *ends, const float *bias, const float *b_first, const float *b_last, int32_t cin,
                           const mz_heads_weights *hw, const float *x, float *hidden, float *value, float *policy, int64_t n, void *stream) {
    MZ_REQUIRE(n >= 0, "mz_tower_x3_forward: n < 0");
    if (n == 0) return MZ_OK;
    MZ_REQUIRE(packed && ends && bias && x && hidden, "mz_tower_x3_forward: null pointer");
    MZ_REQUIRE(cin >= 1 && cin <= 16, "mz_tower_x3_forward: cin %d outside [1, 16]", cin);
    MZ_REQUIRE((((uintptr_t)packed | (uintptr_t)ends | (uintptr_t)bias | (uintptr_t)b_first | (uintptr_t)b_last) & 15) == 0,
               "mz_tower_x3_forward: weights and biases must be 16-byte aligned");
    MZ_REQUIRE((const void *)x != (const void *)hidden, "mz_tower_x3_forward: in-place call");
    const int64_t n_tiles = (n + 3) / 4;
    MZ_REQUIRE(n_tiles <= 0x7fffffff, "mz_tower_x3_forward: too many boards");
    ResArgs a{(const uint4 *)packed, bias, x, hidden, n, 0, nullptr, (const uint16_t *)ends, b_first, b_last, (int)cin,
              nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (hw) {
        MZ_REQUIRE(hw->pc_w && hw->pc_b && hw->p1_w && hw->p1_b && hw->p2_w && hw->p2_b && hw->vc_w && hw->vc_b && hw->v1_w && hw->v1_b && value && policy,
                   "mz_inference_x3_forward: null head pointer");
        MZ_REQUIRE((((uintptr_t)hw->p1_w | (uintptr_t)hw->p2_w) & 15) == 0, "mz_inference_x3_forward: heads p1_w / p2_w must be 16-byte aligned");
        a.pc_w = hw->pc_w; a.pc_b = hw->pc_b; a.p1_w = (const float *)hw->p1_w; a.p1_b = hw->p1_b; a.p2_w = (const float *)hw->p2_w; a.p2_b = hw->p2_b;
        a.vc_w = hw->vc_w; a.vc_b = hw->vc_b; a.v1_w = hw->v1_w; a.v1_b = hw->v1_b; a.value = value; a.policy = policy;
    }
    hipLaunchKernelGGL(k_restower_x3<true>, dim3((unsigned)n_tiles), dim3(512), 0, (hipStream_t)stream, a);
    MZ_LAUNCH_CHECK("k_restower_x3<full>");
    return MZ_OK;
}

extern "C" int mz_tower_x3_forward(const void *packed, const void *ends, const float *bias, const float *b_first, const float *b_last, int32_t cin,
                                  const float *x, float *hidden, int64_t n, void *stream) {
    return tower_x3_launch(packed, ends, bias, b_first, b_last, cin, nullptr, x, hidden, nullptr, nullptr, n, stream);
}

extern "C" int mz_inference_x3_forward(const void *packed, const void *ends, const float *bias, const float *b_first, const float *b_last, int32_t cin,
                                      const mz_heads_weights *heads, const float *x, float *hidden, float *value, float *policy, int64_t n, void *stream) {
    MZ_REQUIRE(heads, "mz_inference_x3_forward: null heads");
    return tower_x3_launch(packed, ends, bias, b_first, b_last, cin, heads, x, hidden, value, policy, n, stream);
}

static int restower_x3_launch(const void *packed, const float *bias, const float *x, float *y, int64_t n, int32_t relu_in, void *stream, long long *stamps) {
    MZ_REQUIRE(n >= 0, "mz_restower_x3_forward: n < 0");
    if (n == 0) return MZ_OK;
    MZ_REQUIRE(packed && bias && x && y, "mz_restower_x3_forward: null pointer");
    MZ_REQUIRE((((uintptr_t)packed | (uintptr_t)bias) & 15) == 0, "mz_restower_x3_forward: packed weights and bias must be 16-byte aligned");
    MZ_REQUIRE((((uintptr_t)x | (uintptr_t)y) & 3) == 0, "mz_restower_x3_forward: x and y must be f32 tensors");
    const int64_t n_tiles = (n + 3) / 4;
    MZ_REQUIRE(n_tiles <= 0x7fffffff, "mz_restower_x3_forward: too many boards");
    ResArgs a{};
    a.w = (const uint4 *)packed; a.bias = bias; a.x = x; a.y = y; a.n_boards = n; a.relu_in = relu_in ? 1 : 0; a.stamps = stamps;
    hipLaunchKernelGGL(k_restower_x3<false>, dim3((unsigned)n_tiles), dim3(512), 0, (hipStream_t)stream, a);
    MZ_LAUNCH_CHECK("k_restower_x3");
    return MZ_OK;
}

extern "C" int mz_restower_x3_forward(const void *packed, const float *bias, const float *x, float *y, int64_t n, int32_t relu_in, void *stream) {
    return restower_x3_launch(packed, bias, x, y, n, relu_in, stream, nullptr);
}
// diagnostics (tools/restower_stamps.py): the same launch with per-tile cycle stamps, (ceil(n / 4), 32) int64
extern "C" int mz_restower_x3_stamps(const void *packed, const float *bias, const float *x, float *y, int64_t n, void *stamps, void *stream) {
    MZ_REQUIRE(stamps, "mz_restower_x3_stamps: null pointer");
    return restower_x3_launch(packed, bias, x, y, n, 0, stream, (long long *)stamps);
}

static int launch_tower(const char *who, const mz_tower_weights *w, const mz_heads_weights *hw, const void *x, void *hidden,
                        int32_t hidden_dtype, float *value, float *policy, int64_t n, void *stream, const Route &rt = Route{nullptr, nullptr, nullptr, 1}) {
    MZ_REQUIRE(w, "%s: null weights", who);
    MZ_REQUIRE(n >= 0, "%s: n < 0", who);
    if (n == 0) return MZ_OK;
    MZ_REQUIRE(w->w_first && w->w_mid && w->b_mid && w->w_last && x && hidden, "%s: null pointer", who);
    MZ_REQUIRE(w->cin >= 1 && w->cin <= 16, "%s: cin %d outside [1, 16]", who, w->cin);
    MZ_REQUIRE(hidden_dtype == MZ_BF16 || hidden_dtype == MZ_F32, "%s: hidden dtype must be bf16 or f32", who);
    MZ_REQUIRE((((uintptr_t)w->w_first | (uintptr_t)w->w_mid | (uintptr_t)w->w_last | (uintptr_t)w->b_mid | (uintptr_t)w->b_first) & 15) == 0,
               "%s: weight pointers must be 16-byte aligned", who);
    Tower tw{(const uint4 *)w->w_first, w->b_first, (const uint4 *)w->w_mid, w->b_mid, (const uint4 *)w->w_last, w->b_last, w->cin,
             (long long *)w->debug_stamps};
    Heads hd{};
    if (hw) {
        MZ_REQUIRE(hw->pc_w && hw->pc_b && hw->p1_w && hw->p1_b && hw->p2_w && hw->p2_b && hw->vc_w && hw->vc_b && hw->v1_w && hw->v1_b && value && policy,
                   "%s: null head pointer", who);
        // the two Linear matrices are fetched as 16-byte MFMA fragments (heads_prefetch)
        MZ_REQUIRE((((uintptr_t)hw->p1_w | (uintptr_t)hw->p2_w) & 15) == 0, "%s: heads p1_w / p2_w must be 16-byte aligned", who);
        hd = Heads{hw->pc_w, hw->pc_b, (const uint16_t *)hw->p1_w, hw->p1_b, (const uint16_t *)hw->p2_w, hw->p2_b, hw->vc_w, hw->vc_b, hw->v1_w, hw->v1_b, value, policy};
    }
    const int64_t n_tiles = (n + TILE_BOARDS - 1) / TILE_BOARDS;
#ifndef MZ_PERSIST
#define MZ_PERSIST 1
#endif
    int64_t n_wg = n_tiles;
    if (MZ_PERSIST) {   // one workgroup per CU (each takes all of a CU's LDS), walking tiles wg, wg + n_wg, ...
        int dev = 0, cus = 0;
        MZ_HIP(hipGetDevice(&dev));
        MZ_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        if (cus > 0 && n_wg > cus) n_wg = cus;
    }
    const dim3 grid((unsigned)n_wg), block(512);
    MZ_REQUIRE(!rt.src_pool || (rt.src_slot && w->cin > 8), "%s: a source pool needs src_slot and a tower of more than 8 input planes", who);
    MZ_REQUIRE(rt.dst_slot || rt.dst_stride >= 1, "%s: dst_stride < 1", who);
    const KArgs ka{tw, hd, (const uint16_t *)x, hidden, n, rt};
    if (hidden_dtype == MZ_F32) hipLaunchKernelGGL(k_tower<MZ_F32>, grid, block, 0, (hipStream_t)stream, ka);
    else hipLaunchKernelGGL(k_tower<MZ_BF16>, grid, block, 0, (hipStream_t)stream, ka);
    MZ_LAUNCH_CHECK("k_tower");
    return MZ_OK;
}

extern "C" int mz_tower_forward(const mz_tower_weights *w, const void *x, void *hidden, int32_t hidden_dtype, int64_t n, void *stream) {
    return launch_tower("mz_tower_forward", w, nullptr, x, hidden, hidden_dtype, nullptr, nullptr, n, stream);
}

extern "C" int mz_inference_forward_routed(const mz_tower_weights *w, const mz_heads_weights *heads, const void *x, const void *src_pool,
                                           const int32_t *src_slot, void *dst_pool, const int32_t *dst_slot, int64_t dst_stride, int32_t hidden_dtype,
                                           float *value, float *policy, int64_t n, void *stream) {
    MZ_REQUIRE(heads, "mz_inference_forward_routed: null heads");
    return launch_tower("mz_inference_forward_routed", w, heads, x, dst_pool, hidden_dtype, value, policy, n, stream,
                        Route{(const uint16_t *)src_pool, src_slot, dst_slot, dst_slot ? 1 : dst_stride});
}

extern "C" int mz_inference_forward(const mz_tower_weights *w, const mz_heads_weights *heads, const void *x, void *hidden, int32_t hidden_dtype,
                                    float *value, float *policy, int64_t n, void *stream) {
    MZ_REQUIRE(heads, "mz_inference_forward: null heads");
    return launch_tower("mz_inference_forward", w, heads, x, hidden, hidden_dtype, value, policy, n, stream);
}
