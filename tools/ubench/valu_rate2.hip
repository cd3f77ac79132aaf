// micro-benchmark: issue cost of single gfx950 VALU instructions (inline asm, 8 independent chains per lane,
// 2048 x 256 threads so that every SIMD holds 8 waves): cycles per wave-instruction per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o build_ub/valu2 tools/ubench/valu_rate2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define OPS(X) \
    X(0, "v_xor_b32 %0, %0, %1") X(1, "v_add_u32 %0, %0, %1") X(2, "v_min3_i32 %0, %0, %1, %2") X(3, "v_med3_i32 %0, %0, %1, %2") \
    X(4, "v_mad_i32_i24 %0, %0, %1, %2") X(5, "v_lshl_add_u32 %0, %0, 3, %1") X(6, "v_perm_b32 %0, %0, %1, %2") \
    X(7, "v_alignbyte_b32 %0, %0, %1, 1") X(8, "v_dot4_u32_u8 %0, %0, %1, %2") X(9, "v_dot2_u32_u16 %0, %0, %1, %2") \
    X(10, "v_bcnt_u32_b32 %0, %0, %1") X(11, "v_pk_min_i16 %0, %0, %1") X(12, "v_mul_lo_u32 %0, %0, %1") X(13, "v_mul_f64 %3, %3, %4") \
    X(14, "v_add_f64 %3, %3, %4") X(15, "v_fma_f32 %0, %0, %1, %2") X(16, "v_mul_u32_u24 %0, %0, %1") X(17, "v_cndmask_b32 %0, %0, %1, vcc") \
    X(18, "v_sad_u8 %0, %0, %1, %2") X(19, "v_cndmask_b32_e64 %0, %0, %1, s[4:5]") X(20, "v_cmp_lt_u32 vcc, %0, %1") \
    X(21, "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc") X(22, "v_cmp_lt_u32 vcc, %0, %1\n v_addc_co_u32 %0, vcc, %0, %2, vcc") \
    X(23, "v_cmp_lt_u32_e64 s[6:7], %0, %1\n s_and_b64 s[8:9], s[6:7], s[4:5]\n v_cndmask_b32_e64 %0, %0, %2, s[8:9]") X(24, "v_max_i32 %0, %0, %1") \
    X(25, "v_sub_co_u32 %0, vcc, %0, %1") X(26, "v_and_or_b32 %0, %0, %1, %2") X(27, "v_bfe_u32 %0, %0, 3, 8") X(28, "v_cvt_f32_ubyte0 %0, %0")
template <int OP>
__global__ void k(uint32_t* out, int iters)
{
    uint32_t a[8];
    double dd[8];
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 2654435761u + i; dd[i] = 1.0 + 1e-9 * i; }
    const uint32_t b = threadIdx.x | 1, c = 0x03020100u;
    const double e = 1.0000001;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
#define X(N, S) if (OP == N) asm volatile(S : "+v"(a[i]) : "v"(b), "v"(c), "v"(dd[i]), "v"(e) : "vcc", "s4", "s5", "s6", "s7", "s8", "s9");
            OPS(X)
#undef X
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + (uint32_t)dd[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> float run(uint32_t* d, int iters)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) { hipEventRecord(e0); k<OP><<<2048, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); }
    return ms;
}
int main()
{
    uint32_t* d; hipMalloc(&d, 256 * 2048 * 4);
    const int iters = 2048;
    const double waveinstr = 2048.0 * 4 * iters * 8;
#define X(N, S) { float ms = run<N>(d, iters); printf("%-36s %6.2f cycles per wave-instr per SIMD (at 2.3 GHz)\n", S, ms * 1e-3 * 2.3e9 * 1024 / waveinstr); }
    OPS(X)
#undef X
    return 0;
}
