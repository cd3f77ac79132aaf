// Standalone timing harness for orbm::k_match_mfma (64 frame pairs x 2000 x 2000 on random descriptors); handy
// for A/B runs of kernel edits (one gpurun call of a few seconds) and as a rocprofv3 --pmc target (tools/pmc_ub.sh).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o build_ub/mm0 tools/ubench/match_mfma.hip
#include "../../orbslamm_amd/csrc/orbm_kernels.hip"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main()
{
    const int B = 64, n = 2000, maxKp = 2104;
    const int64_t xPitch = (int64_t)((maxKp + orbm::kMfmaRowsPerBlock - 1) / orbm::kMfmaRowsPerBlock * orbm::kMfmaRowsPerBlock) * orbm::kMfmaDescBytes;
    uint8_t* d_x; int32_t* d_count; float* d_ang; int32_t *d_match, *d_hist; uint8_t* d_bin;
    CK(hipMalloc(&d_x, (B + 1) * xPitch));
    CK(hipMalloc(&d_count, (B + 1) * 4));
    CK(hipMalloc(&d_ang, (size_t)(B + 1) * maxKp * 4)); CK(hipMemset(d_ang, 0, (size_t)(B + 1) * maxKp * 4));
    CK(hipMalloc(&d_match, (size_t)B * maxKp * 4)); CK(hipMalloc(&d_bin, (size_t)B * maxKp));
    CK(hipMalloc(&d_hist, B * 32 * 4)); CK(hipMemset(d_hist, 0, B * 32 * 4));
    orbm::MatchIO io = {nullptr, 0, d_ang, maxKp, 1, d_count};
    orbm::AcceptArgs aa = {io, io, 1, 0, 0.7f, 50, 1, d_match, (int64_t)maxKp, d_bin, d_hist};
    std::vector<uint8_t> hx((B + 1) * xPitch);
    uint32_t s = 12345;
    for (auto& b : hx) { s = s * 1664525u + 1013904223u; b = ((s >> 24) & 1 ? 0x2 : 0xA) | ((s >> 25) & 1 ? 0x20 : 0xA0); }
    CK(hipMemcpy(d_x, hx.data(), hx.size(), hipMemcpyHostToDevice));
    std::vector<int32_t> hc(B + 1, n);
    CK(hipMemcpy(d_count, hc.data(), (B + 1) * 4, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void*)orbm::k_match_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, orbm::kMfmaLdsBytes));
    const int nqb = (maxKp + orbm::kMfmaRowsPerBlock - 1) / orbm::kMfmaRowsPerBlock;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; i++)
            hipLaunchKernelGGL(orbm::k_match_mfma, dim3(8 * ((B + 7) / 8) * nqb), dim3(orbm::kMfmaThreads), orbm::kMfmaLdsBytes, 0, (const uint8_t*)d_x, xPitch, aa, nqb, B, (uint2*)nullptr, (int64_t)0);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("k_match_mfma: %.1f us per launch\n", ms * 1000 / 20);
    }
    return 0;
}
