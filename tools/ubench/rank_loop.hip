// The rank-sort loop of k_distribute's careful round (E = 256 keys, 512 threads, two threads per key), alone:
// cycles for the loop as the kernel has it, with the table read at a wave-uniform address and rotated.
//   hipcc --offload-arch=gfx950 -O3 -o build_ub/rank_loop tools/ubench/rank_loop.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void k_rank(int E, unsigned long long* out, unsigned* ranks)
{
    extern __shared__ unsigned tA[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int E4 = (E + 3) & ~3;
    for (int e = tid; e < E4; e += 512) tA[e] = e < E ? (((unsigned)(e * 2654435761u) >> 14) << 13 | (8191u - e)) & 0x7FFFFFFFu : 0u;
    __syncthreads();
    int parts = 1;
    while (parts < 8 && E * parts * 2 <= 512) parts *= 2;
    const int per = ((E + parts - 1) / parts + 3) & ~3;
    const unsigned long long c0 = clock64();
    const int e = tid / parts, sub = tid % parts;
    unsigned rank = 0;
    if (e < E) {
        const unsigned ke = tA[e];
        const int base4 = sub * per / 4, n4 = (min(E4, (sub + 1) * per) - sub * per) / 4;
        if (MODE == 0) {
#pragma unroll 4
            for (int j = 0; j < n4; j++) {
                const uint4 k4 = ((const uint4*)tA)[base4 + j];
                rank += ((ke - k4.x) >> 31) + ((ke - k4.y) >> 31) + ((ke - k4.z) >> 31) + ((ke - k4.w) >> 31);
            }
        } else if (MODE == 1) {
            int idx = n4 > 0 ? (lane / parts) % n4 : 0;
#pragma unroll 4
            for (int j = 0; j < n4; j++) {
                const uint4 k4 = ((const uint4*)tA)[base4 + idx];
                rank += ((ke - k4.x) >> 31) + ((ke - k4.y) >> 31) + ((ke - k4.z) >> 31) + ((ke - k4.w) >> 31);
                if (++idx == n4) idx = 0;
            }
        } else {   // dword reads at a wave-uniform address (a true broadcast)
#pragma unroll 8
            for (int j = 0; j < n4 * 4; j++) rank += (ke - tA[base4 * 4 + j]) >> 31;
        }
    }
    for (int d = 1; d < parts; d <<= 1) rank += __shfl_xor(rank, d);
    const unsigned long long c1 = clock64();
    if (e < E && sub == 0) ranks[e] = rank;
    if (tid == 0 && blockIdx.x == 0) out[0] = c1 - c0;
}

int main()
{
    unsigned long long* out; unsigned* ranks;
    CK(hipHostMalloc(&out, 64)); CK(hipMalloc(&ranks, 4096 * 4));
    CK(hipFuncSetAttribute((const void*)k_rank<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    CK(hipFuncSetAttribute((const void*)k_rank<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    CK(hipFuncSetAttribute((const void*)k_rank<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    for (int lds : {16 * 1024, 60 * 1024, 100 * 1024, 150 * 1024})
    for (int blocks : {1, 8})
    for (int E : {256}) {
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(k_rank<0>, dim3(blocks), dim3(512), lds, 0, E, out, ranks); CK(hipDeviceSynchronize()); const unsigned long long a = out[0];
            hipLaunchKernelGGL(k_rank<1>, dim3(blocks), dim3(512), lds, 0, E, out, ranks); CK(hipDeviceSynchronize()); const unsigned long long b = out[0];
            hipLaunchKernelGGL(k_rank<2>, dim3(blocks), dim3(512), lds, 0, E, out, ranks); CK(hipDeviceSynchronize()); const unsigned long long c = out[0];
            printf("LDS %3d KB, %d blocks, E = %3d: uniform uint4 %llu cycles, rotated uint4 %llu, uniform dword %llu\n", lds / 1024, blocks, E, a, b, c);
        }
    }
    return 0;
}
