// micro-benchmark: the shader clock in the first milliseconds of load after an idle gap.  A busy kernel (all CUs, VALU chains,
// ~0.3 ms) is launched back to back; each launch reads s_memtime (shader clock counter) and s_memrealtime (100 MHz) at its start
// and end in one wave: their ratio is the clock the kernel ran at.  Gaps of 0 / 2 / 20 / 200 ms of idle before the burst.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/clock_ramp.hip -o build_ub/clock_ramp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <chrono>
#include <thread>
#include <vector>
__global__ __launch_bounds__(256) void busy(uint64_t* stamps, float* sink, int iters, int slot)
{
    uint64_t c0 = 0, r0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { c0 = __builtin_readcyclecounter(); r0 = __builtin_amdgcn_s_memrealtime(); }
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 0.5f + i;
    for (int it = 0; it < iters; it++)
#pragma unroll
        for (int i = 0; i < 8; i++) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);
    float s = 0; for (int i = 0; i < 8; i++) s += a[i];
    if (s == 12345.f) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const uint64_t c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
        stamps[4 * slot] = c0; stamps[4 * slot + 1] = c1; stamps[4 * slot + 2] = r0; stamps[4 * slot + 3] = r1;
    }
}
int main()
{
    const int N = 160;
    uint64_t* d; float* sink; (void)hipMalloc(&d, N * 4 * 8); (void)hipMalloc(&sink, 4);
    std::vector<uint64_t> h(N * 4);
    const int gaps[4] = {0, 2, 20, 200};
    for (int w = 0; w < 300; w++) busy<<<2048, 256>>>(d, sink, 12000, 0);   // warm
    (void)hipDeviceSynchronize();
    for (int g = 0; g < 4; g++) {
        for (int w = 0; w < 300; w++) busy<<<2048, 256>>>(d, sink, 12000, 0);
        (void)hipDeviceSynchronize();
        std::this_thread::sleep_for(std::chrono::milliseconds(gaps[g]));
        for (int k = 0; k < N; k++) busy<<<2048, 256>>>(d, sink, 12000, k);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), d, N * 4 * 8, hipMemcpyDeviceToHost);
        printf("idle %3d ms, then %d kernels back to back: ms since the first | shader GHz (s_memtime / s_memrealtime at 100 MHz) | kernel ms\n", gaps[g], N);
        for (int k = 0; k < N; k += (k < 16 ? 2 : 16)) {
            const double rt = (double)(h[4 * k + 3] - h[4 * k + 2]) / 100e6, cyc = (double)(h[4 * k + 1] - h[4 * k]);
            printf("   %7.2f  %5.3f  %6.3f\n", (double)(h[4 * k + 2] - h[2]) / 100e3, cyc / rt / 1e9, rt * 1e3);
        }
    }
    return 0;
}
