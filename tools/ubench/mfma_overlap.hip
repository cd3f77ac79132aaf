// micro-benchmark: how far v_mfma_i32_32x32x32_i8 and the matcher's fold (v_min_i32 + v_med3_i32 per accumulator element)
// overlap on gfx950 -- inside one wave and between the two waves of a SIMD.  One "iteration" = the matcher's tile step:
// 16 MFMAs in two accumulate chains, 64 fold instructions on the previous step's accumulators.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o build_ub/mfma_overlap tools/ubench/mfma_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__device__ __forceinline__ int med3i(int a, int b, int c) { return max(min(a, b), min(max(a, b), c)); }
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(int* out, int iters, int seed)
{
    v4i Q[2][8], T[8];
    for (int s = 0; s < 8; s++) {
        Q[0][s] = v4i{(int)threadIdx.x * 77 + s, seed, s, 1}; Q[1][s] = Q[0][s] + 3; T[s] = Q[0][s] * 5;
    }
    const v16i rowIdx = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15};
    v16i aE0 = rowIdx + seed, aE1 = rowIdx - seed, aO0 = aE0, aO1 = aE1;
    int b0 = 0x7FFFFFFF, s0 = b0, b1 = b0, s1 = b0;
    auto products = [&](v16i& a0, v16i& a1) {
        if (MODE == 1) return;
        a0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(T[0], Q[0][0], rowIdx, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(T[0], Q[1][0], rowIdx, 0, 0, 0);
#pragma unroll
        for (int s = 1; s < 8; s++) {
            a0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(T[s], Q[0][s], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(T[s], Q[1][s], a1, 0, 0, 0);
        }
    };
    auto fold = [&](const v16i& a0, const v16i& a1) {
        b0 -= 16; s0 -= 16; b1 -= 16; s1 -= 16;
        if (MODE == 0) { b0 = min(b0, a0[3]); b1 = min(b1, a1[5]); return; }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            s0 = med3i(b0, a0[r], s0); b0 = min(b0, a0[r]);
            s1 = med3i(b1, a1[r], s1); b1 = min(b1, a1[r]);
        }
    };
    for (int it = 0; it < iters; it += 2) {
        products(aO0, aO1); fold(aE0, aE1);
        if (MODE == 1) { aE0 += b0; aE1 += b1; }
        T[it & 7].x ^= b0 & 1;  // keep the MFMAs inside the loop
        products(aE0, aE1); fold(aO0, aO1);
        if (MODE == 1) { aO0 += b0; aO1 += b1; }
    }
    out[blockIdx.x * 256 + threadIdx.x] = b0 + s0 + b1 + s1;
}
int main()
{
    int* d; hipMalloc(&d, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 1024;
    const char* names[3] = {"MFMA only", "fold only", "MFMA + fold"};
    for (int grid : {256, 512, 1024}) {
        for (int mode = 0; mode < 3; mode++) {
            float ms = 0;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                if (mode == 0) k<0><<<grid, 256>>>(d, iters, rep);
                if (mode == 1) k<1><<<grid, 256>>>(d, iters, rep);
                if (mode == 2) k<2><<<grid, 256>>>(d, iters, rep);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            const double wavesPerSimd = grid * 4 / 1024.0;
            printf("grid %4d (%.0f wave(s) per SIMD) %-12s: %8.1f us, %.0f ns per iteration and wave slot = %.0f cycles at 2.1 GHz (16 MFMAs = 512)\n", grid, wavesPerSimd,
                   names[mode], ms * 1000, ms * 1e6 / iters, ms * 1e6 / iters * 2.1);
        }
    }
    return 0;
}
