// train_kernels.hip -- the training step's 128 -> 128 convolution gradients on the matrix cores (gfx950).
//
// The reference trains through autograd (main.py:247-266: K-step unrolled loss, loss.backward()); on MI355X the fp32 step
// through PyTorch/MIOpen spends 2.7 ms of its 13.9 ms in the weight-gradient kernels of the twelve 128 -> 128 convolutions,
// another 1.2 ms in the layout transposes and concatenations around them (profiles/r03/train_step_breakdown.txt).
//
// mz_conv3x3_wgrad: dW[co][ci][ky][kx] = sum over boards b and cells (y, x) of g[b][co][y][x] * in[b][ci][y+ky-1][x+kx-1]
// (+ the bias gradient db[co] = sum g) for ONE convolution over the concatenation of up to 8 uses, reading the fp32 NCHW
// tensors autograd holds -- no transposes, no concatenation, no cast pass.
//
// The contraction runs over (board, cell), and in NCHW a channel's 64 cells are contiguous: a board row (8 cells) of one
// channel IS the 8-element k-fragment a lane feeds to v_mfma_f32_16x16x32_bf16.  So
//   A = g  (rows = co),  B = in shifted by the tap (cols = ci),  k = (board row y = 4 h + (lane >> 4), x = 0..7), h = 0, 1
//   tap (ky, kx): the B fragment is row y + ky - 1 of the SAME channel (a zero fragment outside the board), moved by
//   kx - 1 cells inside the 128-bit fragment (two shifts/alignbits per dword, zero fill at the board's edge).
// One workgroup (8 waves, 2 per SIMD) owns a co half x all 128 ci x all 9 taps: wave w holds ci tile w, 4 co tiles x 9 taps
// = 36 accumulator tiles (144 registers); a board costs 72 MFMAs per wave for 14 fragment reads.
// Boards arrive by LDS-DMA (global_load_lds_dwordx4, no registers) as raw fp32 into a ring of three 48 KB buffers
// ([64 co of this half | 128 ci][64 cells]), two boards ahead of the MFMAs; fragments are rounded to bf16 (RNE) on the way into
// the MFMA.  Rows are 256 B = the LDS's 64 banks, so the 16-byte pieces of row r are rotated by s(r & 15) (the lane sets
// ds_read_b128 is served in are {0-3, 12-15 of k-group g} + {4-11 of k-group g + 1}): conflict-free for every tap.
// Each workgroup takes a slice of the boards and writes its partial sums; k_wgrad_reduce adds the slices in a fixed order
// (deterministic, unlike MIOpen's atomics) into PyTorch's [co][ci][3][3] layout.
#include "common.h"
#include <stddef.h>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int MAX_SEG = 8;
constexpr int BOARD_FLOATS = 128 * 64;
constexpr int G_ROWS = 64, X_ROWS = 128;                 // rows of one board's buffer: this half's 64 co, then the 128 ci
constexpr int BOARD_BYTES = (G_ROWS + X_ROWS) * 256;     // 49,152
constexpr int RING = 3;
constexpr int OFF_ZERO = RING * BOARD_BYTES;             // 32 bytes of zeros: the fragment of a row outside the board
constexpr int WG_LDS = OFF_ZERO + 64;
constexpr int DMA_PER_BOARD = BOARD_BYTES / 1024 / 8;    // 6 LDS-DMA instructions per wave per board
constexpr int PART_FLOATS = 2 * 8 * 4 * 9 * 4 * 64;      // 147,456 = 128 x 128 x 9, in accumulator order [half][wave][ct][tap][lane][i]
constexpr int TARGET_SLICES = 128;

struct WgradArgs {
    const float *x[MAX_SEG];
    const float *g[MAX_SEG];
    long long start[MAX_SEG + 1];   // first global board index of each use; start[n_seg] = all boards
    int n_seg;
    int per_slice;                  // boards per workgroup
    float *partial;                 // [slices][PART_FLOATS]
    float *db_partial;              // [slices][128]
};

typedef const __attribute__((address_space(4))) char *kptr_t;
template <class T>
__device__ __forceinline__ T karg(kptr_t base, size_t off) { return *reinterpret_cast<const __attribute__((address_space(4))) T *>(base + off); }

// rotation of row r's sixteen 16-byte pieces: lanes {0-3, 12-15} of one k-group and {4-11} of the next are served together
__device__ __forceinline__ int rot(int r) { return (r < 4 || r >= 12) ? r : r - 2; }

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {   // v_cvt_pk_bf16_f32, round to nearest even
    const f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ u32x4 to_bf16(const f32x4 lo, const f32x4 hi) {
    return u32x4{pack_bf16(lo.x, lo.y), pack_bf16(lo.z, lo.w), pack_bf16(hi.x, hi.y), pack_bf16(hi.z, hi.w)};
}
// bf16(v - bf16(v)) for the eight values of a fragment whose bf16 rounding is h: the second term of a split-precision product
__device__ __forceinline__ u32x4 remainder_bf16(const f32x4 lo, const f32x4 hi, const u32x4 h) {
    auto up_lo = [](uint32_t p) { return __uint_as_float(p << 16); };
    auto up_hi = [](uint32_t p) { return __uint_as_float(p & 0xFFFF0000u); };
    return u32x4{pack_bf16(lo.x - up_lo(h.x), lo.y - up_hi(h.x)), pack_bf16(lo.z - up_lo(h.y), lo.w - up_hi(h.y)),
                 pack_bf16(hi.x - up_lo(h.z), hi.y - up_hi(h.z)), pack_bf16(hi.z - up_lo(h.w), hi.w - up_hi(h.w))};
}
// the fragment moved by one cell: element e of the result = element e + 1 (up) / e - 1 (down) of v, zero at the board's edge
__device__ __forceinline__ u32x4 cells_up(const u32x4 v) {
    return u32x4{__builtin_amdgcn_alignbit(v.y, v.x, 16), __builtin_amdgcn_alignbit(v.z, v.y, 16), __builtin_amdgcn_alignbit(v.w, v.z, 16), v.w >> 16};
}
__device__ __forceinline__ u32x4 cells_down(const u32x4 v) {
    return u32x4{v.x << 16, __builtin_amdgcn_alignbit(v.y, v.x, 16), __builtin_amdgcn_alignbit(v.z, v.y, 16), __builtin_amdgcn_alignbit(v.w, v.z, 16)};
}

// LDS-DMA, inline asm for the reason given in tower_kernels.hip (dma_piece): the compiler does not see its vmcnt
__device__ __forceinline__ void dma_piece(uint32_t lds_addr_wave, const void *__restrict__ src_uniform, uint32_t lane_off) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr_wave), "v"(lane_off), "s"(src_uniform) : "memory");
}

// X3: split precision -- every fragment as hi = bf16(v) and lo = bf16(v - hi), every product as g_hi x_hi + g_lo x_hi + g_hi x_lo
// (three MFMAs per tile instead of one, on operands that are in LDS anyway: the pass over HBM stays one)
template <bool X3>
__global__ void __launch_bounds__(512) k_conv_wgrad(WgradArgs) {
    __shared__ __attribute__((aligned(16))) char lds[WG_LDS];
    const kptr_t ka = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    const int tid = threadIdx.x, l = tid & 63, r = l & 15, kg = l >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroups b and b + 8 run on the same XCD (round-robin dispatch): a slice's two co halves sit there, so the 32 KB of input
    // planes they both read per board cross HBM once and are shared through that XCD's L2
    const int half = (blockIdx.x >> 3) & 1, slice = ((blockIdx.x >> 4) << 3) | (blockIdx.x & 7);
    const int n_seg = karg<int>(ka, offsetof(WgradArgs, n_seg));
    const int per_slice = karg<int>(ka, offsetof(WgradArgs, per_slice));
    const long long n_all = karg<long long>(ka, offsetof(WgradArgs, start) + 8 * (size_t)n_seg);
    const long long b0 = (long long)slice * per_slice;   // a slice index at or past the real count (grid rounded up to 16) has b0 >= n_all
    const long long b1 = b0 + per_slice < n_all ? b0 + per_slice : n_all;
    if (b0 >= n_all) return;   // the whole workgroup: a padding slice of the rounded-up grid owns no partial sums

    if (tid < 16) reinterpret_cast<float *>(lds + OFF_ZERO)[tid] = 0.f;
    __syncthreads();

    // DMA: wave piece wp = 8 i + w covers rows 4 wp .. 4 wp + 3 of the board buffer; lane l lands at row 4 wp + (l >> 4), slot l & 15,
    // which holds piece (slot - rot(row & 15)) & 15 of that row
    const uint32_t lane_off = (uint32_t)((l >> 4) * 256 + (((l & 15) - rot((4 * (w & 3) + (l >> 4)) & 15)) & 15) * 16);
    const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)lds);
    auto issue = [&](long long board, int slot) {
        int k = 0;
        while (k + 1 < n_seg && board >= karg<long long>(ka, offsetof(WgradArgs, start) + 8 * (size_t)(k + 1))) ++k;
        const long long lb = board - karg<long long>(ka, offsetof(WgradArgs, start) + 8 * (size_t)k);
        const char *gb = reinterpret_cast<const char *>(karg<const float *>(ka, offsetof(WgradArgs, g) + 8 * (size_t)k) + lb * BOARD_FLOATS + half * (G_ROWS * 64));
        const char *xb = reinterpret_cast<const char *>(karg<const float *>(ka, offsetof(WgradArgs, x) + 8 * (size_t)k) + lb * BOARD_FLOATS);
#pragma unroll
        for (int i = 0; i < DMA_PER_BOARD; ++i) {
            const int wp = 8 * i + w;
            const char *src = i < 2 ? gb + wp * 1024 : xb + (wp - 16) * 1024;   // 4 rows of 256 B
            dma_piece(lds0 + (uint32_t)(slot * BOARD_BYTES + wp * 1024), src, lane_off);
        }
    };

    f32x4 acc[4][9];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[ct][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float gsum = 0.f;

    // per-lane piece offsets inside a row, for board rows kg - 1 .. kg + 5 (index yy + 1); rows outside the board -> the zero chunk
    const int s = rot(r);
    const int g_row0 = r * 256, x_row0 = (G_ROWS + 16 * w + r) * 256;

    if (b0 < b1) issue(b0, 0);
    if (b0 + 1 < b1) issue(b0 + 1, 1);
    int slot = 0;
#pragma unroll 1
    for (long long n = b0; n < b1; ++n) {
        if (n + 1 < b1) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (n + 2 < b1) issue(n + 2, slot >= 1 ? slot - 1 : 2);
        const char *buf = lds + slot * BOARD_BYTES;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int y = 4 * h + kg;
            const int p0 = ((2 * y + s) & 15) * 16, p1 = ((2 * y + 1 + s) & 15) * 16;
            if (w < 4) {   // waves 0..3: co tile w of the bias gradient, from the fp32 values
                const f32x4 lo = *reinterpret_cast<const f32x4 *>(buf + g_row0 + w * 4096 + p0);
                const f32x4 hi = *reinterpret_cast<const f32x4 *>(buf + g_row0 + w * 4096 + p1);
                gsum += ((lo.x + lo.y) + (lo.z + lo.w)) + ((hi.x + hi.y) + (hi.z + hi.w));
            }
            bf16x8 af[4], afl[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const f32x4 lo = *reinterpret_cast<const f32x4 *>(buf + g_row0 + ct * 4096 + p0);
                const f32x4 hi = *reinterpret_cast<const f32x4 *>(buf + g_row0 + ct * 4096 + p1);
                const u32x4 h = to_bf16(lo, hi);
                af[ct] = __builtin_bit_cast(bf16x8, h);
                if (X3) afl[ct] = __builtin_bit_cast(bf16x8, remainder_bf16(lo, hi, h));
            }
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int yy = y + dy - 1;
                const bool inside = yy >= 0 && yy <= 7;
                const char *q0 = inside ? buf + x_row0 + ((2 * yy + s) & 15) * 16 : lds + OFF_ZERO;
                const char *q1 = inside ? buf + x_row0 + ((2 * yy + 1 + s) & 15) * 16 : lds + OFF_ZERO + 16;
                const f32x4 xlo = *reinterpret_cast<const f32x4 *>(q0), xhi = *reinterpret_cast<const f32x4 *>(q1);
                const u32x4 raw = to_bf16(xlo, xhi);
                const bf16x8 bdn = __builtin_bit_cast(bf16x8, cells_down(raw)), bmid = __builtin_bit_cast(bf16x8, raw),
                             bup = __builtin_bit_cast(bf16x8, cells_up(raw));
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    acc[ct][dy * 3 + 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ct], bdn, acc[ct][dy * 3 + 0], 0, 0, 0);
                    acc[ct][dy * 3 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ct], bmid, acc[ct][dy * 3 + 1], 0, 0, 0);
                    acc[ct][dy * 3 + 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ct], bup, acc[ct][dy * 3 + 2], 0, 0, 0);
                }
                if (X3) {
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) {   // g_lo x_hi
                        acc[ct][dy * 3 + 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afl[ct], bdn, acc[ct][dy * 3 + 0], 0, 0, 0);
                        acc[ct][dy * 3 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afl[ct], bmid, acc[ct][dy * 3 + 1], 0, 0, 0);
                        acc[ct][dy * 3 + 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afl[ct], bup, acc[ct][dy * 3 + 2], 0, 0, 0);
                    }
                    const u32x4 rl = remainder_bf16(xlo, xhi, raw);
                    const bf16x8 ldn = __builtin_bit_cast(bf16x8, cells_down(rl)), lmid = __builtin_bit_cast(bf16x8, rl),
                                 lup = __builtin_bit_cast(bf16x8, cells_up(rl));
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) {   // g_hi x_lo
                        acc[ct][dy * 3 + 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ct], ldn, acc[ct][dy * 3 + 0], 0, 0, 0);
                        acc[ct][dy * 3 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ct], lmid, acc[ct][dy * 3 + 1], 0, 0, 0);
                        acc[ct][dy * 3 + 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ct], lup, acc[ct][dy * 3 + 2], 0, 0, 0);
                    }
                }
            }
        }
        slot = slot == 2 ? 0 : slot + 1;
    }

    float *part = karg<float *>(ka, offsetof(WgradArgs, partial)) + (size_t)slice * PART_FLOATS + (size_t)((half * 8 + w) * 4) * (9 * 4 * 64);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int t = 0; t < 9; ++t) *reinterpret_cast<f32x4 *>(part + (ct * 9 + t) * 256 + l * 4) = acc[ct][t];
    if (w < 4) {   // bias gradient of co tile w: add the four k-groups (board rows) of each channel
        gsum += __shfl_xor(gsum, 16);
        gsum += __shfl_xor(gsum, 32);
        if (l < 16) karg<float *>(ka, offsetof(WgradArgs, db_partial))[(size_t)slice * 128 + half * 64 + 16 * w + l] = gsum;
    }
}

// dW[co][ci][ky][kx] (+)= sum over slices.  A block owns 64 consecutive partial elements; its four waves each add every fourth
// slice (slices q, q + 4, ...), then the four sums are added ((0 + 1) + (2 + 3)): a fixed order, four times the loads in flight.
// The blocks behind PART_FLOATS / 64 do the bias gradient.
__global__ void __launch_bounds__(256) k_wgrad_reduce(const float *__restrict__ partial, const float *__restrict__ db_partial, int slices,
                                                      float *__restrict__ dw, float *__restrict__ db, int accumulate) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    const bool bias = e >= PART_FLOATS;
    const float *src = bias ? db_partial + (e - PART_FLOATS) : partial + e;
    const size_t stride = bias ? 128 : PART_FLOATS;
    float v = 0.f;
    if (!bias || (db && e < PART_FLOATS + 128))
        for (int sl = q; sl < slices; sl += 4) v += src[(size_t)sl * stride];
    red[q][lane] = v;
    __syncthreads();
    if (q != 0) return;
    v = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    if (!bias) {
        const int i = e & 3, ln = (e >> 2) & 63;
        int rest = e >> 8;
        const int tap = rest % 9;
        rest /= 9;
        const int ct = rest & 3, w = (rest >> 2) & 7, half = rest >> 5;
        const int co = half * 64 + 16 * ct + 4 * (ln >> 4) + i, ci = 16 * w + (ln & 15);
        float *o = dw + ((size_t)co * 128 + ci) * 9 + tap;
        *o = accumulate ? *o + v : v;
    } else if (db && e < PART_FLOATS + 128) {
        const int co = e - PART_FLOATS;
        db[co] = accumulate ? db[co] + v : v;
    }
}

inline int64_t wgrad_per_slice(int64_t n_boards) {
    int64_t per = (n_boards + TARGET_SLICES - 1) / TARGET_SLICES;
    return per < 8 ? 8 : per;   // a slice's partial sums cost 1.2 MB of traffic: at least 8 boards of work behind them
}

}  // namespace

extern "C" int64_t mz_conv3x3_wgrad_workspace(int64_t n_boards) {
    if (n_boards <= 0) return 0;
    const int64_t per = wgrad_per_slice(n_boards), slices = (n_boards + per - 1) / per;
    return slices * (int64_t)(PART_FLOATS + 128) * 4;
}

static int wgrad_launch(bool x3, const float *const *x, const float *const *g, const int64_t *n_boards, int32_t n_seg, float *dw, float *db,
                        int32_t accumulate, void *workspace, int64_t workspace_bytes, void *stream) {
    MZ_REQUIRE(x && g && n_boards && dw, "mz_conv3x3_wgrad: null pointer");
    MZ_REQUIRE(n_seg >= 1 && n_seg <= MAX_SEG, "mz_conv3x3_wgrad: %d uses outside [1, %d]", n_seg, MAX_SEG);
    WgradArgs a{};
    int64_t total = 0;
    for (int k = 0; k < n_seg; ++k) {
        MZ_REQUIRE(x[k] && g[k] && n_boards[k] > 0, "mz_conv3x3_wgrad: use %d is empty or null", k);
        MZ_REQUIRE((((uintptr_t)x[k] | (uintptr_t)g[k]) & 15) == 0, "mz_conv3x3_wgrad: use %d is not 16-byte aligned", k);
        a.x[k] = x[k];
        a.g[k] = g[k];
        a.start[k] = total;
        total += n_boards[k];
    }
    a.start[n_seg] = total;
    a.n_seg = n_seg;
    const int64_t per = wgrad_per_slice(total), slices = (total + per - 1) / per;
    MZ_REQUIRE(per <= INT32_MAX && slices <= (1 << 20), "mz_conv3x3_wgrad: %lld boards are too many", (long long)total);
    const int64_t need = slices * (int64_t)(PART_FLOATS + 128) * 4;
    MZ_REQUIRE(workspace && workspace_bytes >= need, "mz_conv3x3_wgrad: workspace of %lld bytes, %lld needed (mz_conv3x3_wgrad_workspace)",
               (long long)workspace_bytes, (long long)need);
    MZ_REQUIRE(((uintptr_t)workspace & 15) == 0, "mz_conv3x3_wgrad: workspace is not 16-byte aligned");
    a.per_slice = (int)per;
    a.partial = (float *)workspace;
    a.db_partial = a.partial + slices * (int64_t)PART_FLOATS;
    if (x3) hipLaunchKernelGGL(k_conv_wgrad<true>, dim3((unsigned)(16 * ((slices + 7) / 8))), dim3(512), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_conv_wgrad<false>, dim3((unsigned)(16 * ((slices + 7) / 8))), dim3(512), 0, (hipStream_t)stream, a);
    MZ_LAUNCH_CHECK("k_conv_wgrad");
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((PART_FLOATS + 128 + 63) / 64), dim3(256), 0, (hipStream_t)stream, a.partial, a.db_partial, (int)slices, dw,
                       db, (int)accumulate);
    MZ_LAUNCH_CHECK("k_wgrad_reduce");
    return MZ_OK;
}

extern "C" int mz_conv3x3_wgrad(const float *const *x, const float *const *g, const int64_t *n_boards, int32_t n_seg, float *dw, float *db,
                                int32_t accumulate, void *workspace, int64_t workspace_bytes, void *stream) {
    return wgrad_launch(false, x, g, n_boards, n_seg, dw, db, accumulate, workspace, workspace_bytes, stream);
}

extern "C" int mz_conv3x3_wgrad_x3(const float *const *x, const float *const *g, const int64_t *n_boards, int32_t n_seg, float *dw, float *db,
                                   int32_t accumulate, void *workspace, int64_t workspace_bytes, void *stream) {
    return wgrad_launch(true, x, g, n_boards, n_seg, dw, db, accumulate, workspace, workspace_bytes, stream);
}
