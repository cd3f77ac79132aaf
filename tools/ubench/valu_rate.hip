// micro-benchmark: issue rate of v_bcnt_u32_b32 vs v_xor_b32 vs v_pk_min_i16 on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ void k(uint32_t* out, int iters)
{
    uint32_t a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 2654435761u + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) a[i] = __builtin_popcount(a[i] ^ it) + a[(i + 1) & 7];      // xor + bcnt(with add)
            if (OP == 1) a[i] = (a[i] ^ it) + a[(i + 1) & 7];                          // xor + add
            if (OP == 2) { typedef short v2 __attribute__((ext_vector_type(2))); v2 x = *(v2*)&a[i], y = *(v2*)&a[(i + 1) & 7]; x = __builtin_elementwise_min(x, y) + (short)it; a[i] = *(uint32_t*)&x; }
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    uint32_t* d; hipMalloc(&d, 256 * 2048 * 4 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096;
    for (int op = 0; op < 3; op++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (op == 0) k<0><<<2048, 256>>>(d, iters);
            if (op == 1) k<1><<<2048, 256>>>(d, iters);
            if (op == 2) k<2><<<2048, 256>>>(d, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double waveinstr = 2048.0 * 4 * iters * 8 * 2;  // 2 VALU per element
        printf("op %d: %.3f ms, %.2f cycles per wave-instr per SIMD (at 2.3 GHz)\n", op, ms, ms * 1e-3 * 2.3e9 * 1024 / waveinstr);
    }
    return 0;
}
