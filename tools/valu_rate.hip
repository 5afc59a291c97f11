// valu_rate.hip -- issue cost of the integer VALU ops the bitboard kernels are made of (gfx950).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define REP8(s) s s s s s s s s
template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, int iters) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 ^ 17, a7 = a0 ^ 19, m = 0x7E7E7E7Eu;
    uint64_t b0 = a0, b1 = a1, b2 = a2, b3 = a3;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) asm volatile(REP8("v_and_b32 %0, %8, %0\n v_and_b32 %1, %8, %1\n v_and_b32 %2, %8, %2\n v_and_b32 %3, %8, %3\n v_and_b32 %4, %8, %4\n v_and_b32 %5, %8, %5\n v_and_b32 %6, %8, %6\n v_and_b32 %7, %8, %7\n")
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        if (OP == 1) asm volatile(REP8("v_and_or_b32 %0, %8, %1, %0\n v_and_or_b32 %1, %8, %2, %1\n v_and_or_b32 %2, %8, %3, %2\n v_and_or_b32 %3, %8, %4, %3\n v_and_or_b32 %4, %8, %5, %4\n v_and_or_b32 %5, %8, %6, %5\n v_and_or_b32 %6, %8, %7, %6\n v_and_or_b32 %7, %8, %0, %7\n")
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        if (OP == 2) asm volatile(REP8("v_alignbit_b32 %0, %1, %0, 25\n v_alignbit_b32 %1, %2, %1, 25\n v_alignbit_b32 %2, %3, %2, 25\n v_alignbit_b32 %3, %4, %3, 25\n v_alignbit_b32 %4, %5, %4, 25\n v_alignbit_b32 %5, %6, %5, 25\n v_alignbit_b32 %6, %7, %6, 25\n v_alignbit_b32 %7, %0, %7, 25\n")
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        if (OP == 3) asm volatile(REP8("v_lshlrev_b32 %0, 7, %0\n v_lshlrev_b32 %1, 7, %1\n v_lshlrev_b32 %2, 7, %2\n v_lshlrev_b32 %3, 7, %3\n v_lshrrev_b32 %4, 7, %4\n v_lshrrev_b32 %5, 7, %5\n v_lshrrev_b32 %6, 7, %6\n v_lshrrev_b32 %7, 7, %7\n")
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        if (OP == 4) asm volatile(REP8("v_lshlrev_b64 %0, 7, %0\n v_lshlrev_b64 %1, 7, %1\n v_lshrrev_b64 %2, 7, %2\n v_lshrrev_b64 %3, 7, %3\n v_lshlrev_b64 %0, 9, %0\n v_lshlrev_b64 %1, 9, %1\n v_lshrrev_b64 %2, 9, %2\n v_lshrrev_b64 %3, 9, %3\n")
                                  : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
        if (OP == 5) asm volatile(REP8("v_bitop3_b32 %0, %8, %1, %0 bitop3:0xec\n v_bitop3_b32 %1, %8, %2, %1 bitop3:0xec\n v_bitop3_b32 %2, %8, %3, %2 bitop3:0xec\n v_bitop3_b32 %3, %8, %4, %3 bitop3:0xec\n v_bitop3_b32 %4, %8, %5, %4 bitop3:0xec\n v_bitop3_b32 %5, %8, %6, %5 bitop3:0xec\n v_bitop3_b32 %6, %8, %7, %6 bitop3:0xec\n v_bitop3_b32 %7, %8, %0, %7 bitop3:0xec\n")
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(b0 ^ b1 ^ b2 ^ b3);
}
template <int OP> int run(const char *name, uint32_t *out) {
    const int blocks = 256 * 32, iters = 64;  // 32 blocks of 4 waves per CU = 32 waves per SIMD over the launch
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    const double instr_per_simd = (double)blocks * 4 / 1024 * iters * 64;   // waves per SIMD x instrs per wave
    printf("%-16s %.1f us  -> %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", name, ms * 1e3, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    return 0;
}
int main() {
    uint32_t *out; CK(hipMalloc(&out, 256 * 32 * 256 * 4));
    run<0>("v_and_b32", out); run<1>("v_and_or_b32", out); run<2>("v_alignbit_b32", out); run<3>("v_lsh*rev_b32", out);
    run<4>("v_lsh*rev_b64", out); run<5>("v_bitop3_b32", out);
    return 0;
}
