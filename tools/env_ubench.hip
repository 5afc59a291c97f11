// env_ubench.hip -- A/B microbenchmark for the env step kernel variants (one process, interleaved rounds).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/env_ubench.hip -o gpurun_out/env_ubench && ./env_ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#include "../muzero_amd/csrc/env_kernels.hip"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// variant: memory traffic only (same loads/stores, trivial compute)
__global__ void __launch_bounds__(256) k_mem_only(uint64_t *black, uint64_t *white, int8_t *to_play, uint8_t *steps, uint8_t *done,
                                                  uint64_t *legal, const uint8_t *action, int8_t *reward, uint8_t *status, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint64_t b = black[i], w = white[i];
    int tp = to_play[i], st = steps[i], dn = done[i], a = action[i];
    b ^= (uint64_t)a; w ^= (uint64_t)st;
    reward[i] = (int8_t)dn; status[i] = 0;
    black[i] = b; white[i] = w; steps[i] = (uint8_t)(st + 1); done[i] = (uint8_t)dn; to_play[i] = (int8_t)-tp; legal[i] = b | w;
}

int main() {
    const int64_t n = 1 << 20;
    uint64_t *black, *white, *legal; int8_t *to_play, *reward; uint8_t *steps, *done, *action, *status;
    CK(hipMalloc(&black, n * 8)); CK(hipMalloc(&white, n * 8)); CK(hipMalloc(&legal, n * 8));
    CK(hipMalloc(&to_play, n)); CK(hipMalloc(&reward, n)); CK(hipMalloc(&steps, n)); CK(hipMalloc(&done, n));
    CK(hipMalloc(&action, n * 60)); CK(hipMalloc(&status, n));
    // action log of a random playout
    mz_env_reset(black, white, to_play, steps, done, legal, n, 0);
    for (int t = 0; t < 60; ++t) {
        mz_env_random_action(1261, 0, t, legal, action + (int64_t)t * n, n, 0);
        mz_env_step(black, white, to_play, steps, done, legal, action + (int64_t)t * n, reward, status, n, 0);
    }
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> t_step, t_mem;
    for (int round = 0; round < 7; ++round) {
        float ms;
        mz_env_reset(black, white, to_play, steps, done, legal, n, 0);
        CK(hipEventRecord(e0));
        for (int t = 0; t < 60; ++t) mz_env_step(black, white, to_play, steps, done, legal, action + (int64_t)t * n, reward, status, n, 0);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); t_step.push_back(ms / 60 * 1000);
        CK(hipEventRecord(e0));
        for (int t = 0; t < 60; ++t) hipLaunchKernelGGL(k_mem_only, dim3((n + 255) / 256), dim3(256), 0, 0, black, white, to_play, steps, done, legal, action + (int64_t)t * n, reward, status, n);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); t_mem.push_back(ms / 60 * 1000);
    }
    std::sort(t_step.begin(), t_step.end()); std::sort(t_mem.begin(), t_mem.end());
    printf("k_env_step   : median %.2f us  min %.2f us per launch (1M games)\n", t_step[3], t_step[0]);
    printf("k_mem_only   : median %.2f us  min %.2f us per launch\n", t_mem[3], t_mem[0]);
    // fused playout
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) mz_env_random_playout(1261, 0, n, 60, 0, black, white, to_play, steps, done, legal, nullptr, nullptr, nullptr, 0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("fused playout: %.1f us per 60-step game batch = %.2f us/step\n", ms / 5 * 1000, ms / 5 * 1000 / 60);
    uint64_t hb[4]; CK(hipMemcpy(hb, black, 32, hipMemcpyDeviceToHost));
    printf("check black[0..3] = %016llx %016llx %016llx %016llx\n", (unsigned long long)hb[0], (unsigned long long)hb[1], (unsigned long long)hb[2], (unsigned long long)hb[3]);
    return 0;
}
