// net_kernels.hip -- fused elementwise epilogue for the PyTorch-ROCm inference step.
//
// The networks stay PyTorch (network.py:35-167); MIOpen runs the convolutions.  What PyTorch adds
// around every convolution -- bias add, residual add (ResidualBlock, network.py:46-47) and ReLU --
// are three separate full passes over a (B,128,8,8) tensor per layer (profiles/r01: 0.60 ms of a
// 1.55 ms simulation step).  This kernel does them in ONE in-place pass over the channels-last
// tensor:  y = act(y + bias[c] (+ residual)), fp32 arithmetic, one rounding.
// HBM/L3-bound: 16-byte loads/stores per lane, bias held in registers.
#include "common.h"

namespace {

template <int DT> struct Cvt;
template <> struct Cvt<MZ_BF16> {
    static __device__ __forceinline__ float up(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }
    static __device__ __forceinline__ uint16_t down(float f) {  // round to nearest even (finite inputs; NaN stays NaN)
        uint32_t u = __float_as_uint(f);
        if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40);
        return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
    }
};
template <> struct Cvt<MZ_F16> {
    static __device__ __forceinline__ float up(uint16_t v) { _Float16 h; __builtin_memcpy(&h, &v, 2); return (float)h; }
    static __device__ __forceinline__ uint16_t down(float f) { const _Float16 h = (_Float16)f; uint16_t v; __builtin_memcpy(&v, &h, 2); return v; }
};

struct alignas(16) U16x8 { uint16_t v[8]; };

// channels C (multiple of 8, <= 1024) innermost; one lane owns 8 consecutive channels of every
// pixel it visits, so its 8 bias values never change: the grid stride is a multiple of C.
template <int DT, bool RES, bool RELU>
__global__ void __launch_bounds__(256) k_bias_act16(uint16_t *y, const uint16_t *bias, const uint16_t *res, int64_t n_vec, int c_vec) {
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;  // host guarantees stride % c_vec == 0
    float b[8];
    if (bias) {
        const U16x8 bv = reinterpret_cast<const U16x8 *>(bias)[tid % c_vec];
#pragma unroll
        for (int k = 0; k < 8; ++k) b[k] = Cvt<DT>::up(bv.v[k]);
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) b[k] = 0.0f;
    }
    for (int64_t i = tid; i < n_vec; i += stride) {
        U16x8 v = reinterpret_cast<const U16x8 *>(y)[i];
        U16x8 r;
        if (RES) r = reinterpret_cast<const U16x8 *>(res)[i];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float f = Cvt<DT>::up(v.v[k]) + b[k];
            if (RES) f += Cvt<DT>::up(r.v[k]);
            if (RELU) f = fmaxf(f, 0.0f);
            v.v[k] = Cvt<DT>::down(f);
        }
        reinterpret_cast<U16x8 *>(y)[i] = v;
    }
}

template <bool RES, bool RELU>
__global__ void __launch_bounds__(256) k_bias_act32(float *y, const float *bias, const float *res, int64_t n_vec, int c_vec) {
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * 256;
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) b = reinterpret_cast<const float4 *>(bias)[tid % c_vec];
    for (int64_t i = tid; i < n_vec; i += stride) {
        float4 v = reinterpret_cast<const float4 *>(y)[i];
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        if (RES) {
            const float4 r = reinterpret_cast<const float4 *>(res)[i];
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        reinterpret_cast<float4 *>(y)[i] = v;
    }
}

}  // namespace

extern "C" int mz_bias_act_nhwc(void *y, const void *bias, const void *residual, int64_t n_elems, int32_t channels, int32_t dtype,
                                int32_t relu, void *stream) {
    MZ_REQUIRE(n_elems >= 0, "mz_bias_act_nhwc: n_elems < 0");
    if (n_elems == 0) return MZ_OK;
    MZ_REQUIRE(y, "mz_bias_act_nhwc: null y");
    const int per = dtype == MZ_F32 ? 4 : 8;  // elements per 16-byte vector
    MZ_REQUIRE(dtype == MZ_F32 || dtype == MZ_BF16 || dtype == MZ_F16, "mz_bias_act_nhwc: bad dtype %d", dtype);
    MZ_REQUIRE(channels >= per && channels % per == 0 && channels <= 2048, "mz_bias_act_nhwc: channels %d must be a multiple of %d", channels, per);
    MZ_REQUIRE(n_elems % channels == 0, "mz_bias_act_nhwc: n_elems is not a multiple of channels");
    MZ_REQUIRE(((uintptr_t)y & 15) == 0 && ((uintptr_t)bias & 15) == 0 && ((uintptr_t)residual & 15) == 0, "mz_bias_act_nhwc: pointers must be 16-byte aligned");
    const int64_t n_vec = n_elems / per;
    const int c_vec = channels / per;
    // grid: enough blocks to fill the chip, total threads a multiple of c_vec so each lane keeps its channels
    int64_t blocks = (n_vec + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    while ((blocks * 256) % c_vec) ++blocks;
    const dim3 grid((unsigned)blocks), block(256);
    hipStream_t s = (hipStream_t)stream;
#define MZ_EPI16(DT)                                                                                                             \
    do {                                                                                                                         \
        if (residual && relu) hipLaunchKernelGGL((k_bias_act16<DT, true, true>), grid, block, 0, s, (uint16_t *)y, (const uint16_t *)bias, (const uint16_t *)residual, n_vec, c_vec);   \
        else if (residual) hipLaunchKernelGGL((k_bias_act16<DT, true, false>), grid, block, 0, s, (uint16_t *)y, (const uint16_t *)bias, (const uint16_t *)residual, n_vec, c_vec);    \
        else if (relu) hipLaunchKernelGGL((k_bias_act16<DT, false, true>), grid, block, 0, s, (uint16_t *)y, (const uint16_t *)bias, (const uint16_t *)residual, n_vec, c_vec);       \
        else hipLaunchKernelGGL((k_bias_act16<DT, false, false>), grid, block, 0, s, (uint16_t *)y, (const uint16_t *)bias, (const uint16_t *)residual, n_vec, c_vec);                \
    } while (0)
    if (dtype == MZ_BF16) MZ_EPI16(MZ_BF16);
    else if (dtype == MZ_F16) MZ_EPI16(MZ_F16);
    else {
        if (residual && relu) hipLaunchKernelGGL((k_bias_act32<true, true>), grid, block, 0, s, (float *)y, (const float *)bias, (const float *)residual, n_vec, c_vec);
        else if (residual) hipLaunchKernelGGL((k_bias_act32<true, false>), grid, block, 0, s, (float *)y, (const float *)bias, (const float *)residual, n_vec, c_vec);
        else if (relu) hipLaunchKernelGGL((k_bias_act32<false, true>), grid, block, 0, s, (float *)y, (const float *)bias, (const float *)residual, n_vec, c_vec);
        else hipLaunchKernelGGL((k_bias_act32<false, false>), grid, block, 0, s, (float *)y, (const float *)bias, (const float *)residual, n_vec, c_vec);
    }
#undef MZ_EPI16
    MZ_LAUNCH_CHECK("k_bias_act");
    return MZ_OK;
}
