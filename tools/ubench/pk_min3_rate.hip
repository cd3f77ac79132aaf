// micro-benchmark: issue rate and exactness of the packed three-input min/max of gfx950 on small integers.
// v_pk_minimum3_f16 / v_pk_maximum3_f16 are IEEE-754-2019 minimum/maximum on two f16 lanes; the bit patterns 0..0x7BFF
// of positive f16 values order like the integers they spell, 0..255 are denormals, and a kernel's default mode keeps
// f16 denormals -- so on bytes widened to 16 bits they are an exact integer min3 / max3 of TWO values per instruction.
// Question 1: at which rate do they issue (v_pk_min_i16 issues at half rate, valu_rate.hip)?  Question 2: exact on 0..255?
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/pk_min3_rate.hip -o build_ub/pk_min3_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int OP>
__global__ void k(uint32_t* out, int iters)
{
    uint32_t a[8];
    for (int i = 0; i < 8; i++) a[i] = ((threadIdx.x * 37u + i * 11u) & 0xFF) | (((threadIdx.x * 13u + i * 7u) & 0xFF) << 16);
    uint32_t b = (threadIdx.x & 0xFF) | 0x00400000u, c = 0x00330021u;
    uint64_t w[4] = {a[0], a[1], a[2], a[3]}, w2 = 0x3f8000003f800000ull;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 1) asm volatile("v_pk_minimum3_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 2) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 3) asm volatile("v_pk_min_f16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 4) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 5) asm volatile("v_min3_u16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 6) asm volatile("v_min_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 7) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 8) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 9) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 10) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i & 3]) : "v"(b), "v"(c) : "vcc");
            if (OP == 11) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(w[i & 3]) : "v"(w2));
            if (OP == 12) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[0,1]" : "+v"(w[i & 3]) : "v"(w2));
            if (OP == 13) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 14) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 15) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 16) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 17) asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 18) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(a[i]));
            if (OP == 19) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(w[i & 3]) : "v"(w2));
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    for (int i = 0; i < 4; i++) s += (uint32_t)w[i] + (uint32_t)(w[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// exactness: every (x, y, z) in 0..255 in the low lane, (255 - x, z, y) in the high lane
__global__ void k_exact(uint32_t* bad)
{
    const uint32_t x = blockIdx.x, y = threadIdx.x;
    uint32_t nbad = 0;
    for (uint32_t z = 0; z < 256; z++) {
        const uint32_t A = x | ((255 - x) << 16), B = y | (z << 16), C = z | (y << 16);
        uint32_t mn, mx, mnswap;
        asm volatile("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(mn) : "v"(A), "v"(B), "v"(C));
        asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(mx) : "v"(A), "v"(B), "v"(C));
        // second operand with its halves exchanged: low lane takes B.hi, high lane B.lo
        asm volatile("v_pk_minimum3_f16 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(mnswap) : "v"(A), "v"(B), "v"(C));
        const uint32_t lo3 = min(min(x, y), z), hi3 = min(min(255 - x, z), y);
        const uint32_t LO3 = max(max(x, y), z), HI3 = max(max(255 - x, z), y);
        const uint32_t slo = min(min(x, z), z), shi = min(min(255 - x, y), y);
        nbad += mn != (lo3 | (hi3 << 16));
        nbad += mx != (LO3 | (HI3 << 16));
        nbad += mnswap != (slo | (shi << 16));
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main()
{
    uint32_t* d; hipMalloc(&d, 256 * 2048 * 4 * 4);
    uint32_t* bad; hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
    k_exact<<<256, 256>>>(bad);
    uint32_t hb = 1; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    printf("exactness on 0..255 (3 x 16.7 M triples, both lanes, op_sel swap): %u mismatches\n", hb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096;
    const char* names[20] = {"v_min3_i32", "v_pk_minimum3_f16", "v_pk_maximum3_f16 op_sel", "v_pk_min_f16", "v_pk_min_u16", "v_min3_u16", "v_min_i32", "v_perm_b32",
        "v_mul_lo_u32", "v_mul_i32_i24", "v_mad_u64_u32 (4 chains)", "v_pk_mul_f32 (4 chains)", "v_pk_add_f32 op_sel (4 chains)", "v_mul_f32", "v_pk_add_u16", "v_pk_mad_u16", "v_mad_u32_u24", "v_dot4_u32_u8", "v_cvt_f32_ubyte1", "v_pk_fma_f32 (4 chains)"};
    for (int op = 0; op < 20; op++) {
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            switch (op) {
            case 0: k<0><<<2048, 256>>>(d, iters); break;
            case 1: k<1><<<2048, 256>>>(d, iters); break;
            case 2: k<2><<<2048, 256>>>(d, iters); break;
            case 3: k<3><<<2048, 256>>>(d, iters); break;
            case 4: k<4><<<2048, 256>>>(d, iters); break;
            case 5: k<5><<<2048, 256>>>(d, iters); break;
            case 6: k<6><<<2048, 256>>>(d, iters); break;
            case 7: k<7><<<2048, 256>>>(d, iters); break;
            case 8: k<8><<<2048, 256>>>(d, iters); break;
            case 9: k<9><<<2048, 256>>>(d, iters); break;
            case 10: k<10><<<2048, 256>>>(d, iters); break;
            case 11: k<11><<<2048, 256>>>(d, iters); break;
            case 12: k<12><<<2048, 256>>>(d, iters); break;
            case 13: k<13><<<2048, 256>>>(d, iters); break;
            case 14: k<14><<<2048, 256>>>(d, iters); break;
            case 15: k<15><<<2048, 256>>>(d, iters); break;
            case 16: k<16><<<2048, 256>>>(d, iters); break;
            case 17: k<17><<<2048, 256>>>(d, iters); break;
            case 18: k<18><<<2048, 256>>>(d, iters); break;
            case 19: k<19><<<2048, 256>>>(d, iters); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double waveinstr = 2048.0 * 4 * iters * 8;
        printf("%-28s %.3f ms, %.2f cycles per wave-instr per SIMD (at 2.4 GHz)\n", names[op], ms, ms * 1e-3 * 2.4e9 * 1024 / waveinstr);
    }
    return 0;
}
