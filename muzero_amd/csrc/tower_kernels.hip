// tower_kernels.hip -- the MuZero convolution tower as ONE LDS-resident MFMA kernel for gfx950.
//
// Replaces, for bf16 serving, the per-layer PyTorch/MIOpen calls of network.py:49-77
// (Representation) and network.py:113-135 (Dynamics): 3x3 conv Cin->128 + ReLU, three residual
// blocks relu(x + conv(relu(conv(x)))) (network.py:35-47), 3x3 conv 128->8 + ReLU -- and, behind it in
// the same launch, the prediction heads (network.py:79-111), so one kernel returns the whole
// (value, policy, hidden) tuple of initial_inference / recurrent_inference (network.py:146-163).
//
// Why one kernel.  An 8x8 board with 128 channels is 16 KB in bf16, so FOUR boards (256 positions,
// the GEMM N dimension) and both activation tensors a residual block needs fit in the 160 KB LDS of
// one CU.  A workgroup therefore carries its four boards through all eight layers without ever
// writing an activation to HBM: per tile it reads 5 KB of input planes, writes 4 KB of hidden
// state, and streams the layer weights (295 KB per 128->128 layer, L2-resident: every workgroup
// reads the same 1.9 MB) through a ring of four 8 KB LDS buffers.  HBM traffic drops from
// ~200 MB per layer to ~27 MB per launch; what is left is MFMA work.
//
// Mapping (one workgroup = 512 threads = 8 waves, 2 per SIMD; static LDS = 163,840 B = X image 64 KB |
// T image 64 KB | weight ring 4 x 8 KB):
//   GEMM per layer: D[co][pos] = sum_{tap,ci} W[tap][co][ci] * X[pos + shift(tap)][ci]
//   v_mfma_f32_16x16x32_bf16 (first layer: 32x32x16) with A = weights (rows = co), B = activations
//   (cols = pos), so a lane ends up holding ONE position and 4-element runs of consecutive channels: the
//   epilogue (residual, bf16 rounding, ReLU) writes 8-byte pieces straight back into the [pos][ch] LDS
//   image the next layer reads -- no transpose, no shuffle.  The accumulators start at the bias.
//   wave w: co half (w & 1) x board (w >> 1)  ->  a 4 x 4 grid of 16 x 16 accumulators per wave.
//   K loop: 9 taps x 4 channel quarters = 36 stages of [128 co][32 ci], filled by LDS-DMA three stages
//   ahead; one counted raw barrier per stage, and fragment reads run one whole stage ahead ACROSS the
//   barrier (layer_mid), so the MFMA stream never drains at a stage boundary.
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

constexpr int TILE_BOARDS = 4;
constexpr int TILE_POS = 256;
constexpr int ACT_BYTES = TILE_POS * 128 * 2;  // 65,536: one [256 pos][128 ch] bf16 image
constexpr int WSTAGE_BYTES = 128 * 32 * 2;     // 8,192: one [128 co][32 ci] weight stage
constexpr int OFF_X = 0, OFF_T = ACT_BYTES, OFF_W = 2 * ACT_BYTES;
constexpr int LDS_BYTES = 2 * ACT_BYTES + 4 * WSTAGE_BYTES;  // 163,840 = all of a CU's LDS
constexpr int OFF_Z = LDS_BYTES;  // the halo's zero fragment: a DS read at or beyond the workgroup's LDS allocation returns 0
                                  // (hardware bounds check; the allocation is the whole 160 KB, so nothing lies behind it)
constexpr int N_STAGES = 36;                   // 9 taps x 4 quarters of the input channels
constexpr int FIRST_W_BYTES = 9 * 128 * 16 * 2;   // 36,864
constexpr int LAST_W_BYTES = 36 * 16 * 32 * 2;  // 36,864

struct Tower {
    const uint4 *w_first;  // [9][128 co][16 ci] bf16
    const float *b_first;  // [128] or null
    const uint4 *w_mid;    // [6][36][128 co][4 slots][8] bf16, slot = chunk ^ {0,2,3,1}[(co >> 2) & 3]; + 4 stages of padding
    const float *b_mid;    // [6][128]
    const uint4 *w_last;   // [36][16 co (8 real)][4 slots][8] bf16, same swizzle as w_mid
    const float *b_last;   // [8] or null
    int cin;               // real input planes (9 or 10)
    long long *stamps;     // diagnostics: [blocks][24] s_memtime stamps at the phase boundaries (null = off)
};

// prediction heads (network.py:79-111): FC matrices as bf16 input pairs [in / 2][out] (one dword = inputs 2r, 2r + 1 of one
// output) so lanes (= outputs) read coalesced and a thread's 32 inputs are 16 dwords; everything else f32
struct Heads {
    const float *pc_w, *pc_b;    // policy 1x1 conv [2][8], [2]
    const uint32_t *p1_wt;       // Linear(128,128) as bf16 pairs [64][out 128]
    const float *p1_b;           // [128]
    const uint32_t *p2_wt;       // Linear(128,65)  as bf16 pairs [64][out 65]
    const float *p2_b;           // [65]
    const float *vc_w, *vc_b;    // value 1x1 conv [2][8], [2]
    const float *v1_w, *v1_b;    // Linear(128,1) [128], [1]
    float *value, *policy;       // outputs (n), (n,65); null = heads off
};

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
__device__ __forceinline__ void dma_stage(char *ring_buf, const uint4 *__restrict__ stage, int tid) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(stage + tid),
                                     (__attribute__((address_space(3))) void *)(ring_buf + (tid >> 6) * 1024), 16, 0, 0);
}

// ---- 128 -> 128 layer: src image -> dst image (dst may be the residual source, updated in place) ----
// K loop: 36 stages of [128 co][32 ci] weights (8 KB = one K = 32 step = 16 MFMAs per wave) through a
// ring of FOUR LDS buffers filled by LDS-DMA, software-pipelined one whole stage deep:
//   iteration i, after barrier B_i:  run the 16 MFMAs of stage i on the fragment set read during
//   iteration i-1, read the 8 fragments of stage i+1 into the other set (between the MFMAs), and issue
//   the DMA of stage i+3 into the buffer stage i-1 has left.
// B_i (s_waitcnt vmcnt(1) + s_barrier) orders "all waves finished reading stage i-1" (its buffer is
// free) and "every wave's DMA piece of stage i+1 has landed", so stage i+1 is readable anywhere in
// iteration i.  Nothing an MFMA consumes was requested less than a stage ago, and the barrier's
// lgkmcnt(0) finds its reads done.
// The weight stream is one contiguous array over all six layers (+ 4 stages of padding): it simply
// keeps running across layer boundaries, so only the first layer pays a prologue (mid_prologue).
// Halo: a lane whose shifted position leaves the board reads its fragment from a 256-byte block of
// zeros (OFF_Z) instead -- the select is on the ADDRESS, once per tap, so nothing waits on the data.

// the first three weight stages of the first mid layer: issued at kernel start (the ring is idle during the
// first layer), complete by the __syncthreads() that ends the first layer (it drains vmcnt)
__device__ __forceinline__ void mid_prologue(char *lds, const uint4 *__restrict__ wp, int tid) {
    dma_stage(lds + OFF_W, wp, tid);
    dma_stage(lds + OFF_W + WSTAGE_BYTES, wp + 512, tid);
    dma_stage(lds + OFF_W + 2 * WSTAGE_BYTES, wp + 2 * 512, tid);
}

#define MZ_STAMP(tw, k)                                                                              \
    do {                                                                                             \
        if ((tw).stamps && threadIdx.x == 0) (tw).stamps[(int64_t)blockIdx.x * 24 + (k)] = (long long)__builtin_readcyclecounter(); \
    } while (0)

// MFMA shape of the K loop: v_mfma_f32_16x16x32_bf16.  One stage (32 input channels) is exactly one K = 32 step:
// 4 weight fragments (16 co x 32 ci) x 4 activation fragments (16 positions x 32 ci) -> 16 MFMAs into a 4 x 4 grid of
// 16 x 16 accumulators.  Same bytes, cycles and registers as eight 32x32x16 MFMAs, but the chip holds a higher clock on
// this shape under sustained load (MI355X_MICROARCH.md, DVFS item 7; measured here: in-kernel clock 1.87 GHz with 32x32x16).
//   A (weights):     lane l -> row co = 16 i + (l & 15), k = 8 (l >> 4) + 0..7  = the l >> 4 -th 16-byte chunk of the 64-byte row
//   B (activations): lane l -> col pos = 16 j + (l & 15), same k                 = chunk 4 kq + (l >> 4) of the 256-byte row
//   D:               lane l -> col pos, rows co = 16 i + 4 (l >> 4) + 0..3       = 8 bytes of the [pos][ch] image
struct Frags16 { bf16x8 a[4], b[4]; };
typedef __attribute__((ext_vector_type(4))) float acc16_t;

template <bool RES>
__device__ __forceinline__ void layer_mid(char *lds, int src_off, int dst_off, const uint4 *__restrict__ wp, f32x4 (&bv)[4],
                                          const float *__restrict__ bias_next, const uint4 *__restrict__ tail_dma, int tid, uint32_t vm, const Tower &tw, int stamp_id) {
    const int w = tid >> 6, lane = tid & 63, c16 = lane & 15, kc = lane >> 4, p16 = col_cell(c16);
    const int cobase = 64 * (w & 1), posbase = 64 * (w >> 1);
    // the accumulators START at the bias (bv: the lane's 16 values, channels cobase + 16 i + 4 kc + 0..3, requested during
    // the previous layer's epilogue), so the epilogue has no bias add
    acc16_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = bv[i];

    char *wring = lds + OFF_W;
    // weights: address = ring buffer + co * 64 + ((kc ^ key(co)) << 4), key = perm[(co >> 2) & 3], perm = {0, 2, 3, 1}
    int waddr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int co = cobase + 16 * i + c16;
        waddr[i] = co * 64 + ((kc ^ ((0x78 >> (((co >> 2) & 3) * 2)) & 3)) << 4);
    }
    // activations per tap: row base (or the out-of-range zero block) and (kc ^ key) << 4; tile j = board rows 2j, 2j + 1
    int arow[4], ak4[4];
    const int x = p16 & 7, y0 = p16 >> 3;
    auto set_tap = [&](int ky, int kx) {   // tap = 3 ky + kx
        const int dq = (ky - 1) * 8 + (kx - 1);
        const bool okx = (unsigned)(x + kx - 1) < 8u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q2 = (posbase + 16 * j + p16 + dq) & 255;
            const bool ok = okx && (unsigned)(2 * j + y0 + ky - 1) < 8u;
            arow[j] = ok ? src_off + q2 * 256 : OFF_Z;
            ak4[j] = (kc ^ act_key(q2)) << 4;
        }
    };
    auto load_stage = [&](int st, Frags16 &f) {
        const char *wb = wring + (st & 3) * WSTAGE_BYTES;
        const int kq = st & 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) f.a[i] = lds_frag(wb + waddr[i]);
#pragma unroll
        for (int j = 0; j < 4; ++j) f.b[j] = lds_frag(lds + arow[j] + ((kq * 64) ^ ak4[j]));
    };

    Frags16 fr[2];
    set_tap(0, 0);
    load_stage(0, fr[0]);

    // 3 trips x 12 stages (3 taps): ring index, channel quarter and the register-set parity are compile-time constants
    // and the body is branch-free.  The last trip prefetches "tap 9" (every lane on the zero block) and stages 36..38
    // (the next layer's first stages, or the padding): harmless, and no tail conditions.
#pragma unroll 1
    for (int T = 0; T < 3; ++T) {
        const uint4 *wt = wp + (T * 12 + 3) * 512;
#pragma unroll
        for (int ss = 0; ss < 12; ++ss) {
            // all but the youngest DMA (stage st + 2, issued one iteration ago) have landed: stage st + 1 is complete
#ifndef MZ_EXP_NOBARRIER   // diagnostic builds only (-DMZ_EXP_NOBARRIER / -DMZ_EXP_NOWEIGHTS: wrong results, used with tools/tower_stamps.py to price the barrier and the DMA)
            asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
            // hand-ordered issue, fenced with sched_barrier(0) (hipcc otherwise sinks the prefetch reads to their use and
            // exposes every stage to the LDS latency): MFMAs first, so the matrix pipe restarts right behind the barrier;
            // two fragment reads of stage st + 1 behind every second MFMA; the last MFMAs cover those reads' latency
            // before the next barrier's lgkmcnt(0); the weight DMA (60-180 cycles of issue) goes last, into the wave's
            // own MFMA drain.
            const Frags16 &c = fr[ss & 1];
            Frags16 &nx = fr[(ss + 1) & 1];
            const char *wb = wring + ((ss + 1) & 3) * WSTAGE_BYTES;   // 12 = 0 (mod 4, mod 2): ss + 1 indexes like st + 1
            const int kq = (ss + 1) & 3;
#define MZ_MFMA(i, j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c.a[i], c.b[j], acc[i][j], 0, 0, 0)
#define MZ_FENCE() __builtin_amdgcn_sched_barrier(0)
            MZ_MFMA(0, 0);
            MZ_MFMA(0, 1);
            MZ_FENCE();
            if (((ss + 1) & 3) == 0) set_tap(ss + 1 == 12 ? T + 1 : T, ss + 1 == 12 ? 0 : (ss + 1) >> 2);
            nx.a[0] = lds_frag(wb + waddr[0]);
            nx.a[1] = lds_frag(wb + waddr[1]);
            MZ_FENCE();
            MZ_MFMA(0, 2);
            MZ_MFMA(0, 3);
            MZ_FENCE();
            nx.b[0] = lds_frag(lds + arow[0] + ((kq * 64) ^ ak4[0]));
            nx.b[1] = lds_frag(lds + arow[1] + ((kq * 64) ^ ak4[1]));
            MZ_FENCE();
            MZ_MFMA(1, 0);
            MZ_MFMA(1, 1);
            MZ_FENCE();
            nx.a[2] = lds_frag(wb + waddr[2]);
            nx.a[3] = lds_frag(wb + waddr[3]);
            MZ_FENCE();
            MZ_MFMA(1, 2);
            MZ_MFMA(1, 3);
            MZ_FENCE();
            nx.b[2] = lds_frag(lds + arow[2] + ((kq * 64) ^ ak4[2]));
            nx.b[3] = lds_frag(lds + arow[3] + ((kq * 64) ^ ak4[3]));
            MZ_FENCE();
            MZ_MFMA(2, 0);
            MZ_MFMA(2, 1);
            MZ_MFMA(2, 2);
            MZ_MFMA(2, 3);
            MZ_MFMA(3, 0);
            MZ_MFMA(3, 1);
            MZ_MFMA(3, 2);
            MZ_MFMA(3, 3);
            MZ_FENCE();
#ifndef MZ_EXP_NOWEIGHTS
            dma_stage(wring + ((ss + 3) & 3) * WSTAGE_BYTES, wt + ss * 512, tid);   // stage st + 3 -> the buffer stage st - 1 left
#endif
            MZ_FENCE();
#undef MZ_MFMA
#undef MZ_FENCE
        }
    }

    MZ_STAMP(tw, stamp_id);
    // behind the last mid layer: the LAST conv layer's 36 weight stages (36 KB, already in LDS order) go into the T image by
    // LDS-DMA now, under this epilogue.  T is free: after the K loop's last barrier no wave reads it any more (stage 35's
    // fragments were read during stage 34, and the "tap 9" prefetch reads the zero block).
    if (tail_dma) {
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int piece = w + 8 * k;   // 1 KB pieces, wave-uniform
            if (piece < LAST_W_BYTES / 1024)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(tail_dma + piece * 64 + lane),
                                                 (__attribute__((address_space(3))) void *)(lds + OFF_T + piece * 1024), 16, 0, 0);
        }
    }
    // the next layer's bias: two thousand cycles of epilogue ahead of its first use
    if (bias_next) {
#pragma unroll
        for (int i = 0; i < 4; ++i) bv[i] = *reinterpret_cast<const f32x4 *>(bias_next + cobase + 16 * i + 4 * kc);
    }
    // epilogue: lane = position, the 4 registers of a tile = channels co0 .. co0 + 3
    char *dst = lds + dst_off;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = posbase + 16 * j + p16;
        char *row = dst + q * 256 + (kc & 1) * 8;
        const int key = act_key(q);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int chunk = (cobase >> 3) + 2 * i + (kc >> 1);   // co0 = cobase + 16 i + 4 kc
            char *p = row + ((chunk ^ key) << 4);
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = acc[i][j][k];
            if (RES) {
                const uint2 xr = *reinterpret_cast<const uint2 *>(p);
                v[0] += bf16_up(xr.x & 0xFFFFu);
                v[1] += bf16_up(xr.x >> 16);
                v[2] += bf16_up(xr.y & 0xFFFFu);
                v[3] += bf16_up(xr.y >> 16);
            }
            uint2 o;   // round, then ReLU on the packed bf16 pair (same result as ReLU then round: rounding keeps the sign)
            o.x = relu_bf16x2(pack_bf16(v[0], v[1]));
            o.y = relu_bf16x2(pack_bf16(v[2], v[3]));
            *reinterpret_cast<uint2 *>(p) = o;
        }
    }
    // raw barrier (LDS only): the two weight DMAs still in flight belong to the NEXT layer's stages 1 and 2 and are
    // retired by its first counted waits; __syncthreads() would stall every layer boundary on them
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---- first layer: input planes (global, NCHW) -> X.  Uses the T region as scratch: [256 pos][16 ch]
// input image at OFF_T, all 9 tap weight slabs [9][128 co][16 ci] behind it. ---------------------------
__device__ __forceinline__ void layer_first(char *lds, const Tower &tw, const uint16_t *__restrict__ x, int64_t board0, int64_t n_boards, int tid,
                                            uint32_t vm) {
    const int w = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int cobase = 64 * (w & 1), posbase = 64 * (w >> 1);
    char *in = lds + OFF_T;
    char *wf = lds + OFF_T + TILE_POS * 32;
    f32x4 bv[2][4];   // requested now; the accumulators start at the bias (D rows of a 32x32 tile: 8 g + 4 h + k)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bv[i][g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (tw.b_first) bv[i][g] = *reinterpret_cast<const f32x4 *>(tw.b_first + cobase + 32 * i + 8 * g + 4 * h);
        }
    {   // input: thread -> position tid & 255, channels 8*(tid >> 8) .. +7
        const int q = tid & 255, part = tid >> 8;
        const int64_t b = board0 + (q >> 6);
        uint32_t pk[4] = {0, 0, 0, 0};
        if (b < n_boards) {
            const uint16_t *px = x + b * tw.cin * 64 + (q & 63);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int c = part * 8 + k;
                const uint32_t v = c < tw.cin ? (uint32_t)px[c * 64] : 0u;
                pk[k >> 1] |= v << ((k & 1) * 16);
            }
        }
        *reinterpret_cast<uint4 *>(in + q * 32 + part * 16) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        for (int i = tid; i < FIRST_W_BYTES / 16; i += 512) *reinterpret_cast<uint4 *>(wf + i * 16) = tw.w_first[i];
    }
    __syncthreads();
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[i][j][k] = bv[i][k >> 2][k & 3];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int dq = (tap / 3 - 1) * 8 + (tap % 3 - 1);
        bf16x8 a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = lds_frag(wf + (tap * 128 + cobase + 32 * i + r) * 32 + h * 16);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q2 = (posbase + 32 * j + r + dq) & 255;
            b[j] = lds_frag(lds + (((vm >> (tap * 2 + j)) & 1u) ? OFF_T + q2 * 32 : OFF_Z) + h * 16);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    char *dst = lds + OFF_X;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co0 = cobase + 32 * i + 8 * g + 4 * h;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = posbase + 32 * j + r;
                char *p = dst + q * 256 + ((((co0 >> 3) ^ act_key(q))) << 4) + h * 8;
                uint2 o;   // round, then ReLU on the packed pair (as layer_mid)
                o.x = relu_bf16x2(pack_bf16(acc[i][j][4 * g], acc[i][j][4 * g + 1]));
                o.y = relu_bf16x2(pack_bf16(acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]));
                *reinterpret_cast<uint2 *>(p) = o;
            }
        }
    __syncthreads();
}

struct HeadRegs { uint32_t w1[16], w2[16]; float pv; };   // w: bf16 pairs {input 2i (low), input 2i + 1 (high)}
__device__ __forceinline__ float pair_lo(uint32_t v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float pair_hi(uint32_t v) { return __uint_as_float(v & 0xFFFF0000u); }
// 32-input dot product of a thread's weight pairs with 32 f32 features in LDS, on packed fp32 FMAs (v_pk_fma_f32: two
// lanes of an even/odd-input accumulator pair per instruction; the library is built with -ffp-contract=off, so the FMA
// is spelled out).  The Linear layers are VALU-bound: this is a quarter of the instructions of separate mul + add.
__device__ __forceinline__ float dot32(const uint32_t (&wp)[16], const float *x) {
    const f32x4 *f = reinterpret_cast<const f32x4 *>(x);
    f32x2 a = {0.0f, 0.0f};
#pragma unroll
    for (int i4 = 0; i4 < 8; ++i4) {
        const f32x4 v = f[i4];
        a = __builtin_elementwise_fma(f32x2{pair_lo(wp[2 * i4]), pair_hi(wp[2 * i4])}, f32x2{v[0], v[1]}, a);
        a = __builtin_elementwise_fma(f32x2{pair_lo(wp[2 * i4 + 1]), pair_hi(wp[2 * i4 + 1])}, f32x2{v[2], v[3]}, a);
    }
    return a[0] + a[1];
}

// every global read of the heads, issued BEFORE the last conv layer so its latency hides under that layer's MFMAs
__device__ __forceinline__ void heads_prefetch(const Heads &hd, int tid, HeadRegs &hr) {
    const int grp = tid >> 7, k = tid & 127;
#pragma unroll
    for (int i = 0; i < 16; ++i) hr.w1[i] = hd.p1_wt[(16 * grp + i) * 128 + k];
    const int k2 = k < MZ_ACTIONS ? k : 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) hr.w2[i] = hd.p2_wt[(16 * grp + i) * MZ_ACTIONS + k2];
    float pv = 0.0f;   // one small parameter per thread
    if (tid < 16) pv = hd.pc_w[tid];
    else if (tid < 18) pv = hd.pc_b[tid - 16];
    else if (tid < 34) pv = hd.vc_w[tid - 18];
    else if (tid < 36) pv = hd.vc_b[tid - 34];
    else if (tid >= 40 && tid < 168) pv = hd.p1_b[tid - 40];
    else if (tid >= 168 && tid < 168 + MZ_ACTIONS) pv = hd.p2_b[tid - 168];
    else if (tid >= 240 && tid < 368) pv = hd.v1_w[tid - 240];
    else if (tid == 368) pv = hd.v1_b[0];
    hr.pv = pv;
}

// ---- last layer: X -> hidden (global, NCHW, 8 planes) on v_mfma_f32_16x16x32_bf16: the 8 output planes are padded
// to ONE 16-row tile (a 32x32 tile would waste 3/4 of the MFMA).  All 36 weight stages [16 co][32 ci] (36 KB, same
// 64-byte-row swizzle as the mid layers) sit in the T region at once; wave w owns positions 32w .. 32w+31 = two
// 16-position tiles.  D: lane -> position (l & 15), rows co = 4 (l >> 4) + 0..3, i.e. lanes 0..31 hold the 8 real planes.
template <int OUT_DT>
__device__ __forceinline__ void layer_last(char *lds, const Tower &tw, const Heads &hd, HeadRegs &hr, void *__restrict__ hidden,
                                           int64_t board0, int64_t n_boards, int tid, float (&hv)[2][4]) {
    const int w = tid >> 6, lane = tid & 63, c16 = lane & 15, kc = lane >> 4, p16 = col_cell(c16);
    char *wl = lds + OFF_T;
    // this layer's weights were sent to the T image by LDS-DMA during the last mid layer's epilogue (layer_mid, tail_dma):
    // retire every wave's pieces, then the heads' 65 global reads are requested and stay in flight under the MFMAs below
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (hd.policy) heads_prefetch(hd, tid, hr);
    acc16_t acc[2] = {acc16_t{0.0f, 0.0f, 0.0f, 0.0f}, acc16_t{0.0f, 0.0f, 0.0f, 0.0f}};
    const int waddr = c16 * 64 + ((kc ^ ((0x78 >> (((c16 >> 2) & 3) * 2)) & 3)) << 4);
    const int x = p16 & 7, y0 = p16 >> 3, yb = 4 * (w & 1);   // tile j = board rows yb + 2j, yb + 2j + 1
    // one tap = 4 k-steps = 12 fragments; the next tap's fragments are read (fenced) before this tap's 8 MFMAs issue
    struct TapFrags { bf16x8 a[4], b[4][2]; };
    auto load_tap = [&](int tap, TapFrags &f) {
        const int ky = tap / 3, kx = tap % 3;
        const int dq = (ky - 1) * 8 + (kx - 1);
        int arow[2], ak4[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q2 = (32 * w + 16 * j + p16 + dq) & 255;
            const bool ok = (unsigned)(x + kx - 1) < 8u && (unsigned)(yb + 2 * j + y0 + ky - 1) < 8u;
            arow[j] = ok ? OFF_X + q2 * 256 : OFF_Z;
            ak4[j] = (kc ^ act_key(q2)) << 4;
        }
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
            f.a[kq] = lds_frag(wl + (tap * 4 + kq) * 1024 + waddr);
#pragma unroll
            for (int j = 0; j < 2; ++j) f.b[kq][j] = lds_frag(lds + arow[j] + ((kq * 64) ^ ak4[j]));
        }
    };
    TapFrags tf[2];
    load_tap(0, tf[0]);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        if (tap + 1 < 9) load_tap(tap + 1, tf[(tap + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        const TapFrags &c = tf[tap & 1];
#pragma unroll
        for (int kq = 0; kq < 4; ++kq)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c.a[kq], c.b[kq][j], acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    const int64_t b = board0 + (w >> 1);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int cell = 32 * (w & 1) + 16 * j + p16;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int co = (4 * kc + k) & 7;   // lanes 32..63 hold the padding rows: computed, never stored
            const float v = fmaxf(acc[j][k] + (tw.b_last ? tw.b_last[co] : 0.0f), 0.0f);
            const uint32_t vb = bf16_rne(v);
            hv[j][k] = OUT_DT == MZ_F32 ? v : bf16_up(vb);  // what the heads see = what is stored
            if (b < n_boards && kc < 2) {
                const int64_t o = b * 512 + co * 64 + cell;
                if (OUT_DT == MZ_F32) reinterpret_cast<float *>(hidden)[o] = v;
                else reinterpret_cast<uint16_t *>(hidden)[o] = (uint16_t)vb;
            }
        }
    }
}

// ---- prediction heads for the tile's 4 boards (network.py:79-111), fp32 VALU; ~100k MACs per tile ------
// Latency, not arithmetic, is the cost here (every weight comes from L2), so: both Linear layers split
// their 128 inputs over the four 128-thread groups (thread = one output column x 32 inputs x all 4
// boards), ALL 64 weights a thread needs are requested up front (two batches in flight while the 1x1
// convolutions run), and the partial sums meet in LDS.
// LDS (the X image is dead by now), floats: hid [4][8][64] | fp [4][128] | fv [4][128] | h1 [4][128] |
// part [4 groups][4 boards][128] | vsum [4]
__device__ __forceinline__ void heads_tile(char *lds, const Heads &hd, const HeadRegs &hr, const float (&hv)[2][4], int64_t board0,
                                           int64_t n_boards, int tid) {
    float *hid = reinterpret_cast<float *>(lds + OFF_X);
    float *fp = hid + 4 * 512, *fv = fp + 4 * 128, *h1 = fv + 4 * 128, *part = h1 + 4 * 128, *vsum = part + 4 * 4 * 128;
    float *prm = vsum + 4;   // small parameters: pc_w 16 | pc_b 2 | vc_w 16 | vc_b 2 (pad to 40) | p1_b 128 | p2_b 65 (pad 72) | v1_w 128 | v1_b 1
    float *s_pc_w = prm, *s_pc_b = prm + 16, *s_vc_w = prm + 18, *s_vc_b = prm + 34, *s_p1_b = prm + 40, *s_p2_b = prm + 168, *s_v1_w = prm + 240,
          *s_v1_b = prm + 368;
    const int w = tid >> 6, lane = tid & 63;
    const int grp = tid >> 7, k = tid & 127;   // input group / board (wave-uniform), output column
    const uint32_t(&w1)[16] = hr.w1;
    const uint32_t(&w2)[16] = hr.w2;
    __syncthreads();  // every wave is done reading X
    prm[tid] = hr.pv;
    if (lane < 32) {   // layer_last's D layout: lane -> cell col_cell(lane & 15) of tile j, planes 4 (lane >> 4) + 0..3
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) hid[(w >> 1) * 512 + (4 * (lane >> 4) + q) * 64 + 32 * (w & 1) + 16 * j + col_cell(lane & 15)] = hv[j][q];
    }
    __syncthreads();
    {   // 1x1 convs 8 -> 2 + ReLU, flattened channel-major (nn.Flatten of (2,8,8)); here grp = board
        const int c2 = k >> 6, pos = k & 63;
        float ap = s_pc_b[c2], av = s_vc_b[c2];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float x = hid[grp * 512 + c * 64 + pos];
            ap += s_pc_w[c2 * 8 + c] * x;
            av += s_vc_w[c2 * 8 + c] * x;
        }
        fp[grp * 128 + k] = fmaxf(ap, 0.0f);
        fv[grp * 128 + k] = fmaxf(av, 0.0f);
    }
    __syncthreads();
    {   // Linear(128,128): partial sums over inputs 32 grp .. 32 grp + 31, all 4 boards
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            part[(grp * 4 + b) * 128 + k] = dot32(w1, fp + b * 128 + 32 * grp);
        }
        if (w < 4) {   // value Linear(128,1): wave w = board w
            float v = s_v1_w[lane] * fv[w * 128 + lane] + s_v1_w[lane + 64] * fv[w * 128 + lane + 64];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) vsum[w] = v;
        }
    }
    __syncthreads();
    h1[grp * 128 + k] = fmaxf(s_p1_b[k] + part[(0 * 4 + grp) * 128 + k] + part[(1 * 4 + grp) * 128 + k] + part[(2 * 4 + grp) * 128 + k] +
                                  part[(3 * 4 + grp) * 128 + k], 0.0f);   // here grp = board
    __syncthreads();
    if (k < MZ_ACTIONS) {   // Linear(128,65) partial sums
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            part[(grp * 4 + b) * 128 + k] = dot32(w2, h1 + b * 128 + 32 * grp);
        }
    }
    __syncthreads();
    if (w < 4 && board0 + w < n_boards) {   // softmax over 65 (one wave per board), tanh
        auto logit = [&](int a) { return s_p2_b[a] + part[(0 * 4 + w) * 128 + a] + part[(1 * 4 + w) * 128 + a] + part[(2 * 4 + w) * 128 + a] + part[(3 * 4 + w) * 128 + a]; };
        const float x = logit(lane), x64 = logit(64);
        float m = fmaxf(x, x64);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        const float e = expf(x - m), e64 = expf(x64 - m);
        float sum = e;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        sum += e64;
        float *pol = hd.policy + (board0 + w) * MZ_ACTIONS;
        pol[lane] = e / sum;
        if (lane == 0) {
            pol[64] = e64 / sum;
            hd.value[board0 + w] = tanhf(vsum[w] + s_v1_b[0]);
        }
    }
}

template <int OUT_DT>
__global__ void __launch_bounds__(512) k_tower(Tower tw, Heads hd, const uint16_t *__restrict__ x, void *__restrict__ hidden, int64_t n_boards) {
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    const int tid = threadIdx.x;
    const uint32_t vm = halo_mask(tid & 31);
    const int64_t board0 = (int64_t)blockIdx.x * TILE_BOARDS;
    MZ_STAMP(tw, 0);
    if (tw.stamps && threadIdx.x == 0) tw.stamps[(int64_t)blockIdx.x * 24 + 17] = (long long)__builtin_amdgcn_s_memrealtime();
    mid_prologue(lds, tw.w_mid, tid);
    f32x4 bv[4];   // bias of the first mid layer in the K loop's accumulator layout (see layer_mid)
#pragma unroll
    for (int i = 0; i < 4; ++i) bv[i] = *reinterpret_cast<const f32x4 *>(tw.b_mid + 64 * ((tid >> 6) & 1) + 16 * i + 4 * ((tid & 63) >> 4));
    layer_first(lds, tw, x, board0, n_boards, tid, vm);
    MZ_STAMP(tw, 1);
    MZ_STAMP(tw, 2);
#pragma unroll 1
    for (int blk = 0; blk < 3; ++blk) {
        const uint4 *w0 = tw.w_mid + (int64_t)(2 * blk) * N_STAGES * 512;
        const uint4 *w1 = tw.w_mid + (int64_t)(2 * blk + 1) * N_STAGES * 512;
        layer_mid<false>(lds, OFF_X, OFF_T, w0, bv, tw.b_mid + (2 * blk + 1) * 128, nullptr, tid, vm, tw, 3 + 4 * blk);      // t = relu(conv(x))
        MZ_STAMP(tw, 4 + 4 * blk);
        layer_mid<true>(lds, OFF_T, OFF_X, w1, bv, blk < 2 ? tw.b_mid + (2 * blk + 2) * 128 : nullptr, blk == 2 ? tw.w_last : nullptr, tid, vm, tw,
                         5 + 4 * blk);   // x = relu(x + conv(t))
        MZ_STAMP(tw, 6 + 4 * blk);
    }
    float hv[2][4];
    HeadRegs hr;
    layer_last<OUT_DT>(lds, tw, hd, hr, hidden, board0, n_boards, tid, hv);
    MZ_STAMP(tw, 15);
    if (hd.policy) heads_tile(lds, hd, hr, hv, board0, n_boards, tid);
    MZ_STAMP(tw, 16);
    if (tw.stamps && threadIdx.x == 0) tw.stamps[(int64_t)blockIdx.x * 24 + 18] = (long long)__builtin_amdgcn_s_memrealtime();
}

}  // namespace

static int launch_tower(const char *who, const mz_tower_weights *w, const mz_heads_weights *hw, const void *x, void *hidden,
                        int32_t hidden_dtype, float *value, float *policy, int64_t n, void *stream) {
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
        MZ_REQUIRE(hw->pc_w && hw->pc_b && hw->p1_wt && hw->p1_b && hw->p2_wt && hw->p2_b && hw->vc_w && hw->vc_b && hw->v1_w && hw->v1_b && value && policy,
                   "%s: null head pointer", who);
        hd = Heads{hw->pc_w, hw->pc_b, hw->p1_wt, hw->p1_b, hw->p2_wt, hw->p2_b, hw->vc_w, hw->vc_b, hw->v1_w, hw->v1_b, value, policy};
    }
    const dim3 grid((unsigned)((n + TILE_BOARDS - 1) / TILE_BOARDS)), block(512);
    if (hidden_dtype == MZ_F32) hipLaunchKernelGGL(k_tower<MZ_F32>, grid, block, 0, (hipStream_t)stream, tw, hd, (const uint16_t *)x, hidden, n);
    else hipLaunchKernelGGL(k_tower<MZ_BF16>, grid, block, 0, (hipStream_t)stream, tw, hd, (const uint16_t *)x, hidden, n);
    MZ_LAUNCH_CHECK("k_tower");
    return MZ_OK;
}

extern "C" int mz_tower_forward(const mz_tower_weights *w, const void *x, void *hidden, int32_t hidden_dtype, int64_t n, void *stream) {
    return launch_tower("mz_tower_forward", w, nullptr, x, hidden, hidden_dtype, nullptr, nullptr, n, stream);
}

extern "C" int mz_inference_forward(const mz_tower_weights *w, const mz_heads_weights *heads, const void *x, void *hidden, int32_t hidden_dtype,
                                    float *value, float *policy, int64_t n, void *stream) {
    MZ_REQUIRE(heads, "mz_inference_forward: null heads");
    return launch_tower("mz_inference_forward", w, heads, x, hidden, hidden_dtype, value, policy, n, stream);
}
