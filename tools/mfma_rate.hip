// mfma_rate.hip -- what does the K loop of k_tower cost on bare hardware?  Micro-benchmark, not product code.
// One workgroup per CU (all 160 KB of LDS claimed, as k_tower does), 4 or 8 waves, each running ITER "stages" of
// 16 independent v_mfma_f32_16x16x32_bf16 (the stage of layer_mid) with, optionally, the stage's other ingredients:
//   R  8 conflict-free ds_read_b128 per stage, placed as in layer_mid (2 behind every second MFMA)
//   B  the counted raw barrier per stage
//   D  one 1-KB LDS-DMA per wave per stage
// Prints shader cycles per stage (wave 0 of every block, averaged) next to the 16-cycle-per-MFMA ideal.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o tools/mfma_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int LDS_BYTES = 163840;
constexpr int ITER = 2048;

#define CK(x)                                                                        \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

template <bool R, bool B, bool D, int P = 0>
__global__ void __launch_bounds__(512) k_rate(const uint4 *__restrict__ src, float *out, long long *cyc) {
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < LDS_BYTES / 16; i += blockDim.x) reinterpret_cast<uint4 *>(lds)[i] = src[i & 4095];
    __syncthreads();
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a[2][4], b[2][4];
    // conflict-free: 64 lanes x 16 B contiguous, a different KB per fragment and wave
    const char *base = lds + w * 8192 + lane * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[0][i] = a[1][i] = *reinterpret_cast<const bf16x8 *>(base + i * 1024);
        b[0][i] = b[1][i] = *reinterpret_cast<const bf16x8 *>(base + (4 + i) * 1024);
    }
    __syncthreads();
    uint4 stg2[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};   // P == 10: the same, loaded two stages ahead (no latency exposed)
    uint4 stg = make_uint4(0, 0, 0, 0);   // P == 5: register-staged copy (global_load_dwordx4 now, ds_write_b128 one stage later)
    const unsigned lds_base = (unsigned)(uintptr_t)lds;   // LDS byte address of the array
    const long long t0 = __builtin_readcyclecounter();
#define MF(s, i, j) acc[4 * i + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s][i], b[s][j], acc[4 * i + j], 0, 0, 0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)
#define DMA()                                                                                                                        \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + ((it + s) & 7) * 512 + tid),            \
                                     (__attribute__((address_space(3))) void *)(lds + 131072 + ((it + s) & 3) * 8192 + w * 1024), 16, 0, 0)
// P == 11: the same DMA in its SGPR-base + 32-bit lane offset form (no 64-bit address VGPR pair)
#define DMA_SADDR()                                                                                                                 \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"((unsigned)(tid * 16)),                         \
                 "s"(src + ((it + s) & 7) * 512), "s"(__builtin_amdgcn_readfirstlane((int)(lds_base + 131072 + ((it + s) & 3) * 8192 + w * 1024))) : "memory")
#define RD(dst, off) if (R) dst = *reinterpret_cast<const bf16x8 *>(base + (off) * 1024)
#pragma unroll 1
    for (int it = 0; it < ITER; it += 2) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (B && P == 6) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else if (B && P == 7) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else if (B && P == 8) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else if (B) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else if (R) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            MF(s, 0, 0); MF(s, 0, 1); FENCE();
            RD(a[s ^ 1][0], 0); RD(a[s ^ 1][1], 1); FENCE();
            MF(s, 0, 2); MF(s, 0, 3); FENCE();
            if (D && (P == 1 || (P == 2 && (w >> 2) == 0) || (P == 3 && (w & 1) == 0) || (P == 4 && (w >> 2) == 0))) { DMA(); FENCE(); }
            if (D && P == 5) { *reinterpret_cast<uint4 *>(lds + 131072 + ((it + s) & 3) * 8192 + tid * 16) = stg; FENCE(); }
            RD(b[s ^ 1][0], 4); RD(b[s ^ 1][1], 5); FENCE();
            MF(s, 1, 0); MF(s, 1, 1); FENCE();
            RD(a[s ^ 1][2], 2); RD(a[s ^ 1][3], 3); FENCE();
            MF(s, 1, 2); MF(s, 1, 3); FENCE();
            RD(b[s ^ 1][2], 6); RD(b[s ^ 1][3], 7); FENCE();
            MF(s, 2, 0); MF(s, 2, 1); MF(s, 2, 2); MF(s, 2, 3); FENCE();
            if (D && (P == 4 && (w >> 2) == 1)) { DMA(); FENCE(); }
            MF(s, 3, 0); MF(s, 3, 1); MF(s, 3, 2); MF(s, 3, 3); FENCE();
            if (D && P == 5) { stg = src[((it + s) & 7) * 512 + tid]; FENCE(); }
            if (D && P == 11) { DMA_SADDR(); FENCE(); }
            if (D && P == 10) {
                *reinterpret_cast<uint4 *>(lds + 131072 + ((it + s) & 3) * 8192 + tid * 16) = stg2[s];
                FENCE();
                stg2[s] = src[((it + s) & 7) * 512 + tid];
                FENCE();
            }
            if (D && (P == 0 || (P >= 6 && P <= 8) || (P == 2 && (w >> 2) == 1) || (P == 3 && (w & 1) == 1))) { DMA(); FENCE(); }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (D) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[(size_t)blockIdx.x * blockDim.x + tid] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
}

template <bool R, bool B, bool D, int P = 0>
static void run(const char *name, int waves, const uint4 *src, float *out, long long *cyc) {
    const int blocks = 256;
    CK(hipMemset(cyc, 0, blocks * 8 * sizeof(long long)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_rate<R, B, D, P>), dim3(blocks), dim3(64 * waves), 0, 0, src, out, cyc);   // warm-up
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_rate<R, B, D, P>), dim3(blocks), dim3(64 * waves), 0, 0, src, out, cyc);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(blocks * 8);
    CK(hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    double sum = 0, mx = 0;
    int n = 0;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < waves; ++w) {
            sum += (double)h[b * 8 + w];
            if ((double)h[b * 8 + w] > mx) mx = (double)h[b * 8 + w];
            ++n;
        }
    const double per_stage = sum / n / ITER;
    const int per_simd = waves / 4;
    const double ideal = 16.0 * 16 * per_simd;
    const double tflops = (double)blocks * waves * ITER * 16 * 16384.0 / (ms * 1e-3) / 1e12;
    printf("%-34s waves/SIMD %d  cycles/stage %7.1f (max wave %7.1f)  ideal %5.0f  -> %5.1f %% of the MFMA pipe   %.3f ms  %6.0f TFLOP/s\n", name,
           per_simd, per_stage, mx / ITER, ideal, 100.0 * ideal / per_stage, ms, tflops);
}

int main8(const uint4 *src, float *out, long long *cyc);
int main9(const uint4 *src, float *out, long long *cyc);
int main10(const uint4 *src, float *out, long long *cyc);
int main() {
    uint4 *src;
    float *out;
    long long *cyc;
    CK(hipMalloc(&src, 65536 * 16));
    CK(hipMemset(src, 0x11, 65536 * 16));
    CK(hipMalloc(&out, 256 * 512 * sizeof(float)));
    CK(hipMalloc(&cyc, 256 * 8 * sizeof(long long)));
    for (int waves = 4; waves <= 8; waves += 4) {
        run<false, false, false>("bare MFMAs", waves, src, out, cyc);
        run<true, false, false>("+ 8 ds_read_b128", waves, src, out, cyc);
        run<false, true, false>("+ barrier", waves, src, out, cyc);
        run<true, true, false>("+ reads + barrier", waves, src, out, cyc);
        run<true, true, true>("+ reads + barrier + DMA (= K loop)", waves, src, out, cyc);
        run<true, true, true, 1>("  DMA early (after MFMA 4), all", waves, src, out, cyc);
        run<true, true, true, 2>("  DMA early w<4, end w>=4", waves, src, out, cyc);
        run<true, true, true, 3>("  DMA early even w, end odd w", waves, src, out, cyc);
        run<true, true, true, 4>("  DMA after 4 (w<4) / after 12 (w>=4)", waves, src, out, cyc);
        run<true, true, true, 5>("  register-staged copy instead", waves, src, out, cyc);
        run<true, true, true, 10>("  register-staged, 2 stages ahead, end", waves, src, out, cyc);
        run<true, true, true, 11>("  DMA, SGPR base + 32-bit offset form", waves, src, out, cyc);
        run<true, true, true, 6>("  DMA at the end, wait vmcnt(2)", waves, src, out, cyc);
        run<true, true, true, 7>("  DMA at the end, wait vmcnt(3)", waves, src, out, cyc);
        run<true, true, true, 8>("  DMA at the end, no vmcnt wait", waves, src, out, cyc);
    }
    main8(src, out, cyc);
    return 0;
}

// ---- round 3: the 8-board tile.  Per stage a wave runs 4 x 8 MFMAs (8 waves, 2 per SIMD, 128 accumulator registers), j outer:
// the stage's 4 weight fragments are resident (read one stage ahead, 2 sets), the activation fragments stream through 4 slots LA
// j-steps ahead; 1 LDS-DMA piece per wave per stage at position P (0 = end of the stage, 1 = right behind the barrier, 2..4 =
// behind j-step 1, 3, 5); one counted barrier per stage, with (LG) or without lgkmcnt(0).
// (A 4-wave, 512-register form with 256 accumulators was measured too: hipcc shuttles accumulators between the two register
// files and spills, 56 % of the pipe on bare MFMAs -- not usable without hand-written assembly.)
template <int P, bool LG, int LA, bool R, bool B, bool D>
__global__ void __launch_bounds__(512) k_rate8(const uint4 *__restrict__ src, float *out, long long *cyc) {
    constexpr int NJ = 8;
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < LDS_BYTES / 16; i += blockDim.x) reinterpret_cast<uint4 *>(lds)[i] = src[i & 4095];
    __syncthreads();
    f32x4 acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a[2][4], b[4];
    const char *base = lds + w * 8192 + lane * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[0][i] = a[1][i] = *reinterpret_cast<const bf16x8 *>(base + i * 1024);
        b[i] = *reinterpret_cast<const bf16x8 *>(base + (4 + i) * 1024);
    }
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
#define DMA8()                                                                                                                             \
    do {                                                                                                                                   \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + ((it + s) & 7) * 512 + tid),               \
                                         (__attribute__((address_space(3))) void *)(lds + 131072 + ((it + s) & 3) * 8192 + w * 1024), 16, 0, 0); \
        FENCE();                                                                                                                           \
    } while (0)
#pragma unroll 1
    for (int it = 0; it < ITER; it += 2) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (B) {
                if (LG) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(1)\n\ts_barrier" ::: "memory");
            }
            if (D && P == 1) DMA8();
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s][0], b[j & 3], acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s][1], b[j & 3], acc[1][j], 0, 0, 0);
                FENCE();
                if (R && j < 4) { a[s ^ 1][j] = *reinterpret_cast<const bf16x8 *>(base + j * 1024); FENCE(); }
                acc[2][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s][2], b[j & 3], acc[2][j], 0, 0, 0);
                acc[3][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s][3], b[j & 3], acc[3][j], 0, 0, 0);
                FENCE();
                // slot (j + 3) & 3 held B[j - 1], whose MFMAs have all been issued: refill it with B[j + 3] (of this stage or the next)
                if (R && LA == 3) { b[(j + 3) & 3] = *reinterpret_cast<const bf16x8 *>(base + (4 + ((j + 3) & 3)) * 1024 + ((j + 3) & 4) * 64); FENCE(); }
                if (R && LA == 2) { b[(j + 2) & 3] = *reinterpret_cast<const bf16x8 *>(base + (4 + ((j + 2) & 3)) * 1024 + ((j + 2) & 4) * 64); FENCE(); }
                if (D && ((P == 2 && j == 1) || (P == 3 && j == 3) || (P == 4 && j == 5))) DMA8();
            }
            if (D && P == 0) DMA8();
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (D) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[(size_t)blockIdx.x * blockDim.x + tid] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
}

template <int P, bool LG, int LA, bool R, bool B, bool D>
static void run8(const char *name, const uint4 *src, float *out, long long *cyc) {
    const int blocks = 256, waves = 8;
    CK(hipMemset(cyc, 0, blocks * 8 * sizeof(long long)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_rate8<P, LG, LA, R, B, D>), dim3(blocks), dim3(512), 0, 0, src, out, cyc);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_rate8<P, LG, LA, R, B, D>), dim3(blocks), dim3(512), 0, 0, src, out, cyc);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(blocks * 8);
    CK(hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    double sum = 0;
    int n = 0;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < waves; ++w) { sum += (double)h[b * 8 + w]; ++n; }
    const double per_stage = sum / n / ITER, ideal = 16.0 * 32 * 2;
    const double tflops = (double)blocks * waves * ITER * 32 * 16384.0 / (ms * 1e-3) / 1e12;
    printf("%-58s cycles/stage %7.1f  ideal %5.0f  -> %5.1f %% of the MFMA pipe   %.3f ms  %6.0f TFLOP/s\n", name, per_stage, ideal,
           100.0 * ideal / per_stage, ms, tflops);
}

// ---- barrier every BE stages (a deeper weight ring buys that): NJ = 4 (4-board tile, 64 accumulators) or 8 (8-board tile) ----
// mode bits: 1 = reads, 2 = barrier, 4 = DMA (one piece per wave per stage, behind j-step DJ)
template <int NJ, int BE, int DJ, int MODE>
__global__ void __launch_bounds__(512) k_rate9(const uint4 *__restrict__ src, float *out, long long *cyc) {
    constexpr bool R = MODE & 1, B = MODE & 2, D = MODE & 4;
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < LDS_BYTES / 16; i += blockDim.x) reinterpret_cast<uint4 *>(lds)[i] = src[i & 4095];
    __syncthreads();
    f32x4 acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a[2][4], b[4];
    const char *base = lds + w * 8192 + lane * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[0][i] = a[1][i] = *reinterpret_cast<const bf16x8 *>(base + i * 1024);
        b[i] = *reinterpret_cast<const bf16x8 *>(base + (4 + i) * 1024);
    }
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < ITER; it += 4) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (B && (s % BE) == 0) {
                // everything this wave issued before the previous barrier interval's last BE pieces has landed
                if (BE == 1) asm volatile("s_waitcnt vmcnt(1)\n\ts_barrier" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s & 1][0], b[j & 3], acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s & 1][1], b[j & 3], acc[1][j], 0, 0, 0);
                FENCE();
                if (R && (NJ == 8 ? j < 4 : true)) {
                    a[(s & 1) ^ 1][NJ == 8 ? j : j] = *reinterpret_cast<const bf16x8 *>(base + j * 1024);
                    FENCE();
                }
                acc[2][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s & 1][2], b[j & 3], acc[2][j], 0, 0, 0);
                acc[3][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s & 1][3], b[j & 3], acc[3][j], 0, 0, 0);
                FENCE();
                if (R) { b[(j + 3) & 3] = *reinterpret_cast<const bf16x8 *>(base + (4 + ((j + 3) & 3)) * 1024 + ((j + 3) & 4) * 64); FENCE(); }
                if (D && j == DJ) {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + ((it + s) & 7) * 512 + tid),
                                                     (__attribute__((address_space(3))) void *)(lds + 98304 + ((it + s) & 7) * 8192 + w * 1024), 16, 0, 0);
                    FENCE();
                }
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (D) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[(size_t)blockIdx.x * blockDim.x + tid] = sum;
    if (lane == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
}

template <int NJ, int BE, int DJ, int MODE>
static void run9(const char *name, const uint4 *src, float *out, long long *cyc) {
    const int blocks = 256, waves = 8;
    CK(hipMemset(cyc, 0, blocks * 8 * sizeof(long long)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_rate9<NJ, BE, DJ, MODE>), dim3(blocks), dim3(512), 0, 0, src, out, cyc);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_rate9<NJ, BE, DJ, MODE>), dim3(blocks), dim3(512), 0, 0, src, out, cyc);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(blocks * 8);
    CK(hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    double sum = 0;
    int n = 0;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < waves; ++w) { sum += (double)h[b * 8 + w]; ++n; }
    const double per_stage = sum / n / ITER, ideal = 16.0 * 4 * NJ * 2;
    const double tflops = (double)blocks * waves * ITER * 4 * NJ * 16384.0 / (ms * 1e-3) / 1e12;
    printf("NJ %d barrier/%d stages DMA@j%d mode %d %-22s cycles/stage %7.1f  ideal %5.0f  -> %5.1f %%   %.3f ms  %6.0f TFLOP/s  clock %.2f GHz\n", NJ, BE, DJ, MODE, name,
           per_stage, ideal, 100.0 * ideal / per_stage, ms, tflops, per_stage * ITER / (ms * 1e-3) / 1e9);
}

int main9(const uint4 *src, float *out, long long *cyc) {
    run9<4, 1, 1, 0>("bare", src, out, cyc);
    run9<4, 1, 1, 1>("reads", src, out, cyc);
    run9<4, 1, 1, 2>("barrier", src, out, cyc);
    run9<4, 1, 1, 3>("reads+barrier", src, out, cyc);
    run9<4, 1, 1, 7>("K loop", src, out, cyc);
    run9<4, 2, 1, 7>("K loop", src, out, cyc);
    run9<4, 4, 1, 7>("K loop", src, out, cyc);
    run9<4, 4, 1, 3>("reads+barrier", src, out, cyc);
    run9<4, 4, 1, 5>("reads+DMA", src, out, cyc);
    run9<8, 1, 5, 0>("bare", src, out, cyc);
    run9<8, 1, 5, 1>("reads", src, out, cyc);
    run9<8, 1, 5, 2>("barrier", src, out, cyc);
    run9<8, 1, 5, 3>("reads+barrier", src, out, cyc);
    run9<8, 1, 5, 7>("K loop", src, out, cyc);
    run9<8, 2, 5, 7>("K loop", src, out, cyc);
    run9<8, 4, 5, 7>("K loop", src, out, cyc);
    run9<8, 2, 5, 3>("reads+barrier", src, out, cyc);
    run9<8, 4, 5, 5>("reads+DMA", src, out, cyc);
    return 0;
}

// ---- the 8-board K loop with the LDS-DMA written as inline asm (ASMDMA): with the builtin in the loop hipcc turns every counted
// lgkmcnt(N) it would place before an MFMA into lgkmcnt(0), which exposes the LDS latency of the streamed activation reads.
// The B slot freed by j-step j - 1 is refilled right behind the first MFMA pair of j-step j (B[j + 3], of this stage or the next).
template <bool ASMDMA, int DJ, int SLOTS, int NJ = 8, bool LG0 = false>
__global__ void __launch_bounds__(512) k_rate10(const uint4 *__restrict__ src, float *out, long long *cyc) {
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < LDS_BYTES / 16; i += blockDim.x) reinterpret_cast<uint4 *>(lds)[i] = src[i & 4095];
    __syncthreads();
    f32x4 acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a[2][4], b[SLOTS];
    const char *base = lds + w * 8192 + lane * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) a[0][i] = a[1][i] = *reinterpret_cast<const bf16x8 *>(base + i * 1024);
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) b[i] = *reinterpret_cast<const bf16x8 *>(base + (4 + (i & 3)) * 1024);
    __syncthreads();
    const unsigned lds_w = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds + 131072u + (unsigned)w * 1024u);   // wave-uniform, scalar
    const long long t0 = __builtin_readcyclecounter();
    int slot = 0;   // compile-time after unrolling: the B slot of (stage, j) is (8 * stage + j) % SLOTS
#pragma unroll 1
    for (int it = 0; it < ITER; it += SLOTS) {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            if (LG0) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(1)\n\ts_barrier" ::: "memory");
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int cur = (NJ * s + j) % SLOTS, fill = (NJ * s + j + SLOTS - 1) % SLOTS;   // fill = the slot B[j - 1] held = B[j + SLOTS - 1]'s
                acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s & 1][0], b[cur], acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s & 1][1], b[cur], acc[1][j], 0, 0, 0);
                FENCE();
                b[fill] = *reinterpret_cast<const bf16x8 *>(base + (4 + (fill & 3)) * 1024 + (j & 4) * 64 + (s & 1) * 128);
                FENCE();
                acc[2][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s & 1][2], b[cur], acc[2][j], 0, 0, 0);
                acc[3][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s & 1][3], b[cur], acc[3][j], 0, 0, 0);
                FENCE();
                if (j < 4) { a[(s & 1) ^ 1][j] = *reinterpret_cast<const bf16x8 *>(base + j * 1024); FENCE(); }
                if (j == DJ) {
                    if (ASMDMA) {
                        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src + ((it + s) & 7) * 512 + tid),
                                     "s"(lds_w + (unsigned)(((it + s) & 3) * 8192)) : "memory");
                    } else {
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + ((it + s) & 7) * 512 + tid),
                                                         (__attribute__((address_space(3))) void *)(lds + 131072 + ((it + s) & 3) * 8192 + w * 1024), 16, 0, 0);
                    }
                    FENCE();
                }
            }
        }
    }
    (void)slot;
    const long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[(size_t)blockIdx.x * blockDim.x + tid] = sum;
    if (lane == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
}

template <bool ASMDMA, int DJ, int SLOTS, int NJ = 8, bool LG0 = false>
static void run10(const char *name, const uint4 *src, float *out, long long *cyc) {
    const int blocks = 256, waves = 8;
    constexpr int IT = (ITER / SLOTS) * SLOTS;
    static_assert(SLOTS == 4, "ITER is a multiple of 4");
    CK(hipMemset(cyc, 0, blocks * 8 * sizeof(long long)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_rate10<ASMDMA, DJ, SLOTS, NJ, LG0>), dim3(blocks), dim3(512), 0, 0, src, out, cyc);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_rate10<ASMDMA, DJ, SLOTS, NJ, LG0>), dim3(blocks), dim3(512), 0, 0, src, out, cyc);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(blocks * 8);
    CK(hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    double sum = 0;
    int n = 0;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < waves; ++w) { sum += (double)h[b * 8 + w]; ++n; }
    const double per_stage = sum / n / IT, ideal = 128.0 * NJ;
    const double tflops = (double)blocks * waves * IT * 4 * NJ * 16384.0 / (ms * 1e-3) / 1e12;
    printf("NJ %d %-40s cycles/stage %7.1f  ideal %5.0f  -> %5.1f %%   %.3f ms  %6.0f TFLOP/s  clock %.2f GHz\n", NJ, name, per_stage, ideal,
           100.0 * ideal / per_stage, ms, tflops, per_stage * IT / (ms * 1e-3) / 1e9);
}

// sustained-load form: ~1.5 s of back-to-back launches on RANDOM bf16 data first (the clock the chip holds under this load, and
// on these operands, is what the wall time is made of: MI355X_MICROARCH.md, DVFS give-back), then 50 timed launches
template <bool ASMDMA, int DJ, int SLOTS, int NJ, bool LG0>
static void sustain10(const char *name, const uint4 *src, float *out, long long *cyc) {
    const int blocks = 256, waves = 8;
    constexpr int IT = (ITER / SLOTS) * SLOTS;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int warm = NJ == 8 ? 1500 : 3000;
    for (int i = 0; i < warm; ++i) hipLaunchKernelGGL((k_rate10<ASMDMA, DJ, SLOTS, NJ, LG0>), dim3(blocks), dim3(512), 0, 0, src, out, cyc);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((k_rate10<ASMDMA, DJ, SLOTS, NJ, LG0>), dim3(blocks), dim3(512), 0, 0, src, out, cyc);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 50;
    std::vector<long long> h(blocks * 8);
    CK(hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    double sum = 0;
    int n = 0;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < waves; ++w) { sum += (double)h[b * 8 + w]; ++n; }
    const double per_stage = sum / n / IT, ideal = 128.0 * NJ;
    const double tflops = (double)blocks * waves * IT * 4 * NJ * 16384.0 / (ms * 1e-3) / 1e12;
    printf("sustained, random data: NJ %d %-32s cycles/stage %7.1f -> %5.1f %%   %.3f ms  %6.0f TFLOP/s  clock ~%.2f GHz\n", NJ, name, per_stage,
           100.0 * ideal / per_stage, ms, tflops, per_stage * IT / (ms * 1e-3) / 1e9);
}

int main10(const uint4 *src, float *out, long long *cyc) {
    run10<false, 7, 4>("builtin DMA at the end", src, out, cyc);
    run10<false, 5, 4>("builtin DMA behind j = 5", src, out, cyc);
    run10<true, 7, 4>("asm DMA at the end", src, out, cyc);
    run10<true, 5, 4>("asm DMA behind j = 5", src, out, cyc);
    run10<true, 0, 4>("asm DMA behind j = 0", src, out, cyc);
    run10<true, 7, 4, 8, true>("asm DMA at the end, lgkmcnt(0) at the barrier", src, out, cyc);
    run10<false, 3, 4, 4>("builtin DMA at the end", src, out, cyc);
    run10<true, 3, 4, 4>("asm DMA at the end", src, out, cyc);
    run10<true, 0, 4, 4>("asm DMA behind j = 0", src, out, cyc);
    run10<true, 3, 4, 4, true>("asm DMA at the end, lgkmcnt(0) at the barrier", src, out, cyc);
    // random operands
    {
        std::vector<unsigned short> h(65536 * 8);
        unsigned long long st = 88172645463325252ull;
        for (auto &v : h) {
            st ^= st << 13; st ^= st >> 7; st ^= st << 17;
            const float f = ((int)(st >> 40 & 0xFFFF) - 32768) / 32768.0f;   // uniform (-1, 1)
            unsigned u; memcpy(&u, &f, 4);
            v = (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16);
        }
        CK(hipMemcpy((void *)src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    }
    for (int rep = 0; rep < 2; ++rep) {
        sustain10<false, 3, 4, 4, true>("builtin DMA, lgkmcnt(0) (r02 K loop)", src, out, cyc);
        sustain10<true, 0, 4, 4, false>("asm DMA behind j = 0 (r03 K loop)", src, out, cyc);
        sustain10<true, 0, 4, 8, false>("asm DMA behind j = 0 (8-board)", src, out, cyc);
    }
    return 0;
}

int main8(const uint4 *src, float *out, long long *cyc) {
    run8<0, true, 3, false, false, false>("8-board: bare MFMAs", src, out, cyc);
    run8<0, true, 3, true, true, false>("8-board: + 12 reads + barrier (lgkmcnt 0)", src, out, cyc);
    run8<0, false, 3, true, true, false>("8-board: + 12 reads + barrier (no lgkm wait)", src, out, cyc);
    run8<0, false, 2, true, true, false>("8-board: + 12 reads (2 ahead) + barrier (no lgkm wait)", src, out, cyc);
    run8<0, true, 3, true, true, true>("8-board: + DMA at the end, lgkmcnt(0)", src, out, cyc);
    run8<0, false, 3, true, true, true>("8-board: + DMA at the end", src, out, cyc);
    run8<1, false, 3, true, true, true>("8-board: + DMA behind the barrier", src, out, cyc);
    run8<2, false, 3, true, true, true>("8-board: + DMA behind j = 1", src, out, cyc);
    run8<3, false, 3, true, true, true>("8-board: + DMA behind j = 3", src, out, cyc);
    run8<4, false, 3, true, true, true>("8-board: + DMA behind j = 5", src, out, cyc);
    run8<3, false, 2, true, true, true>("8-board: + DMA behind j = 3, reads 2 ahead", src, out, cyc);
    run8<3, true, 3, true, true, true>("8-board: + DMA behind j = 3, lgkmcnt(0)", src, out, cyc);
    main9(src, out, cyc);
    main10(src, out, cyc);
    return 0;
}
