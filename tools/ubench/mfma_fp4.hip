// Can the +-1 product run on the fp4 path of v_mfma_scale_f32_32x32x64_f8f6f4?  E2M1 holds +-1 exactly (0x2 / 0xA), the two
// block scales 2^5 make a product +-1024, sums stay integers < 2^24: exact in the f32 accumulator if the hardware adds exactly.
// Checks acc = preset + 1024 * dot against the host for random +-1 operands and times the instruction (2 chains, 2 waves per SIMD).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o build_ub/mfma_fp4 tools/ubench/mfma_fp4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ void k_check(const uint4* __restrict__ A, const uint4* __restrict__ B, float* __restrict__ out)
{
    // one wave: A = 32 rows x 64 k, B = 32 cols x 64 k; lane l owns row/col l & 31, k range 32 (l >> 5) .. + 32 (16 bytes)
    const int lane = threadIdx.x;
    const uint4 a = A[lane], b = B[lane];
    v8i va = {(int)a.x, (int)a.y, (int)a.z, (int)a.w, 0, 0, 0, 0}, vb = {(int)b.x, (int)b.y, (int)b.z, (int)b.w, 0, 0, 0, 0};
    v16f c;
    for (int r = 0; r < 16; r++) c[r] = 524288.f + (float)r;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c, 4, 4, 0, 132, 0, 132);
    for (int r = 0; r < 16; r++) out[lane * 16 + r] = c[r];
}
template <int N>
__global__ __launch_bounds__(256, 2) void k_rate(float* out, int iters, int seed)
{
    v8i a[4], b[2][4];
    for (int s = 0; s < 4; s++) { a[s] = v8i{(int)threadIdx.x * 77 + s, seed, s, 1, 0, 0, 0, 0}; b[0][s] = a[s] + 3; b[1][s] = a[s] + 7; }
    v16f c0 = {}, c1 = {};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int s = 0; s < 4; s++) {
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[s], b[0][s], c0, 4, 4, 0, 132, 0, 132);
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[s], b[1][s], c1, 4, 4, 0, 132, 0, 132);
        }
        a[it & 3][0] ^= it;
    }
    float s = 0;
    for (int r = 0; r < 16; r++) s += c0[r] + c1[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    std::vector<uint32_t> ha(64 * 4), hb(64 * 4);
    std::vector<int> sa(32 * 64), sb(32 * 64);  // +-1 values [row][k]
    srand(7);
    for (auto& v : sa) v = rand() & 1 ? 1 : -1;
    for (auto& v : sb) v = rand() & 1 ? 1 : -1;
    auto pack = [](const std::vector<int>& s, std::vector<uint32_t>& h) {
        for (int lane = 0; lane < 64; lane++)
            for (int w = 0; w < 4; w++) {
                uint32_t word = 0;
                for (int n = 0; n < 8; n++) { const int k = 32 * (lane >> 5) + 8 * w + n; word |= (uint32_t)(s[(lane & 31) * 64 + k] > 0 ? 0x2 : 0xA) << (4 * n); }
                h[lane * 4 + w] = word;
            }
    };
    pack(sa, ha); pack(sb, hb);
    uint4 *dA, *dB; float* dO;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dO, 4096 * 256 * 4);
    hipMemcpy(dA, ha.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hb.data(), 1024, hipMemcpyHostToDevice);
    k_check<<<1, 64>>>(dA, dB, dO);
    std::vector<float> ho(1024);
    hipMemcpy(ho.data(), dO, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; lane++)
        for (int r = 0; r < 16; r++) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
            int dot = 0;
            for (int k = 0; k < 64; k++) dot += sa[row * 64 + k] * sb[col * 64 + k];
            const float want = 524288.f + r + 1024.f * dot;
            if (ho[lane * 16 + r] != want) { if (bad < 5) printf("lane %d r %d: got %.1f want %.1f\n", lane, r, ho[lane * 16 + r], want); bad++; }
        }
    printf("fp4 product check: %d mismatches of 1024\n", bad);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096;
    for (int grid : {256, 512}) {
        float ms = 0;
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            k_rate<0><<<grid, 256>>>(dO, iters, rep);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double n = (double)grid * 4 * iters * 8;
        printf("grid %d: %.1f us, %.2f ns per MFMA and SIMD (%.0f TFLOP/s fp4)\n", grid, ms * 1000, ms * 1e6 / (n / 1024), n * 131072 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
