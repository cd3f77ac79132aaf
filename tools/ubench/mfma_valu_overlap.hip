// micro-benchmark: does VALU work issue in the shadow of v_mfma_scale_f32_32x32x64_f8f6f4 (fp4 operands) on gfx950?
// One loop iteration = 8 MFMAs in two accumulator chains (the stream matcher's tile step) + NV independent v_min3_f32 of the
// same wave, interleaved.  Waves per SIMD 1 or 2.  If the time per iteration is max(MFMA, VALU) the two overlap; if it is the sum
// they do not.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_valu_overlap.hip -o build_ub/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v8i __attribute__((ext_vector_type(8)));

template <int NV, bool MF>
__global__ __launch_bounds__(256) void k(float* out, int iters)
{
    v8i A = {(int)threadIdx.x, 2, 3, 4, 0, 0, 0, 0}, B = {5, (int)threadIdx.x * 7, 7, 8, 0, 0, 0, 0};
    v16f c0 = {}, c1 = {};
    float m[8];
    for (int i = 0; i < 8; i++) m[i] = (float)(threadIdx.x + i);
    const float x = out[0], y = out[1];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int s = 0; s < 4; s++) {
            if (MF) {
                c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c0, 4, 4, 0, 127, 0, 127);
#pragma unroll
                for (int v = 0; v < NV / 8; v++) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(m[v & 7]) : "v"(x), "v"(y));
                c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B, A, c1, 4, 4, 0, 127, 0, 127);
#pragma unroll
                for (int v = 0; v < NV / 8; v++) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(m[(v + 4) & 7]) : "v"(x), "v"(y));
            } else {
#pragma unroll
                for (int v = 0; v < NV / 4; v++) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(m[v & 7]) : "v"(x), "v"(y));
            }
        }
    }
    float r = 0;
    for (int i = 0; i < 16; i++) r += c0[i] + c1[i];
    for (int i = 0; i < 8; i++) r += m[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int NV, bool MF>
static void run(float* d, int wgPerCu, const char* what)
{
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        k<NV, MF><<<256 * wgPerCu, 256>>>(d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-22s NV=%2d waves/SIMD=%d: %.1f ns per iteration per wave-slot (%.1f ns per iteration of the SIMD)\n", what, NV, wgPerCu,
           ms * 1e6 / iters / wgPerCu, ms * 1e6 / iters);
}

// one wave, FOUR accumulator chains: 16 MFMAs + NV VALU per iteration (a wave that owns four query blocks)
template <int NV>
__global__ __launch_bounds__(256) void k4(float* out, int iters)
{
    v8i A = {(int)threadIdx.x, 2, 3, 4, 0, 0, 0, 0}, B = {5, (int)threadIdx.x * 7, 7, 8, 0, 0, 0, 0};
    v16f c[4] = {};
    float m[8];
    for (int i = 0; i < 8; i++) m[i] = (float)(threadIdx.x + i);
    const float x = out[0], y = out[1];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                c[q] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c[q], 4, 4, 0, 127, 0, 127);
#pragma unroll
                for (int v = 0; v < NV / 16; v++) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(m[(v + q * 2) & 7]) : "v"(x), "v"(y));
            }
    }
    float r = 0;
    for (int q = 0; q < 4; q++) for (int i = 0; i < 16; i++) r += c[q][i];
    for (int i = 0; i < 8; i++) r += m[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int NV>
static void run4(float* d)
{
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        k4<NV><<<256, 256>>>(d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("16 MFMA in 4 chains + VALU NV=%3d, 1 wave/SIMD: %.1f ns per iteration of the SIMD\n", NV, ms * 1e6 / iters);
}

// the matcher's real pattern: 8 MFMAs into one accumulator set while the VALU folds the OTHER set (written by the MFMAs of
// the iteration before): 40 v_min3/v_med3 reading accumulator registers.  MODE 0: fold reads the other set; 1: fold reads
// plain registers (same instruction mix, no MFMA-written sources)
template <int MODE>
__global__ __launch_bounds__(256, 2) void k5  // (…, 2): accumulators in VGPRs -- with one wave per SIMD allowed hipcc moves them to AGPRs and reads every key back with v_accvgpr_read
(float* out, int iters)
{
    v8i A = {(int)threadIdx.x, 2, 3, 4, 0, 0, 0, 0}, B = {5, (int)threadIdx.x * 7, 7, 8, 0, 0, 0, 0};
    v16f e0 = {}, e1 = {}, o0 = {}, o1 = {}, p0, p1;
    for (int i = 0; i < 16; i++) { p0[i] = out[i] + threadIdx.x; p1[i] = out[i + 16] - threadIdx.x; }
    float b0 = out[2], s0 = out[3], b1 = out[4], s1 = out[5];
    auto fold = [&](const v16f& a0, const v16f& a1) {
        b0 -= 16.f; s0 -= 16.f; b1 -= 16.f; s1 -= 16.f;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            s0 = __builtin_fminf(s0, __builtin_amdgcn_fmed3f(b0, a0[r], a0[r + 1]));
            b0 = __builtin_fminf(__builtin_fminf(b0, a0[r]), a0[r + 1]);
            s1 = __builtin_fminf(s1, __builtin_amdgcn_fmed3f(b1, a1[r], a1[r + 1]));
            b1 = __builtin_fminf(__builtin_fminf(b1, a1[r]), a1[r + 1]);
        }
    };
    auto prod = [&](v16f& a0, v16f& a1) {
#pragma unroll
        for (int s = 0; s < 4; s++) {
            a0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, a0, 4, 4, 0, 127, 0, 127);
            a1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(B, A, a1, 4, 4, 0, 127, 0, 127);
        }
    };
    for (int it = 0; it < iters; it += 2) {
        asm volatile("" : "+v"(A), "+v"(B));
        prod(e0, e1); if (MODE == 0) fold(o0, o1); else { asm volatile("" : "+v"(p0), "+v"(p1)); fold(p0, p1); }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" : "+v"(A), "+v"(B));
        prod(o0, o1); if (MODE == 0) fold(e0, e1); else { asm volatile("" : "+v"(p0), "+v"(p1)); fold(p0, p1); }
        __builtin_amdgcn_sched_barrier(0);
    }
    float r = b0 + s0 + b1 + s1;
    for (int i = 0; i < 16; i++) r += e0[i] + e1[i] + o0[i] + o1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE>
static void run5(float* d, int wgPerCu)
{
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        k5<MODE><<<256 * wgPerCu, 256>>>(d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("8 MFMA + the matcher's fold (44 VALU) %s, waves/SIMD=%d: %.1f ns per iteration of the SIMD\n",
           MODE == 0 ? "reading the other accumulator set" : "reading plain registers", wgPerCu, ms * 1e6 / iters);
}

// k5 with the kernel's operands: four tile fragments x eight query fragments in registers of their own, a chain's first
// MFMA starting from a resident constant set (C != D).  MODE 2: distinct operands; 3: + the constant C
typedef int v4i __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256, 2) void k6(float* out, int iters)
{
    v4i T[4], Q[2][4];
    for (int s = 0; s < 4; s++) { T[s] = v4i{(int)threadIdx.x + s, 2 * s, 3, 4}; Q[0][s] = v4i{5, (int)threadIdx.x * 7 + s, 7, 8}; Q[1][s] = v4i{s, 9, (int)threadIdx.x, 1}; }
    v16f e0 = {}, e1 = {}, o0 = {}, o1 = {}, row;
    for (int i = 0; i < 16; i++) row[i] = 524288.f + i;
    asm volatile("" : "+v"(row));
    float b0 = out[2], s0 = out[3], b1 = out[4], s1 = out[5];
    auto fold = [&](const v16f& a0, const v16f& a1) {
        b0 -= 16.f; s0 -= 16.f; b1 -= 16.f; s1 -= 16.f;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            s0 = __builtin_fminf(s0, __builtin_amdgcn_fmed3f(b0, a0[r], a0[r + 1]));
            b0 = __builtin_fminf(__builtin_fminf(b0, a0[r]), a0[r + 1]);
            s1 = __builtin_fminf(s1, __builtin_amdgcn_fmed3f(b1, a1[r], a1[r + 1]));
            b1 = __builtin_fminf(__builtin_fminf(b1, a1[r]), a1[r + 1]);
        }
    };
    auto mf = [](const v4i& a, const v4i& b, const v16f& c) {
        const v8i a8 = {a.x, a.y, a.z, a.w, 0, 0, 0, 0}, b8 = {b.x, b.y, b.z, b.w, 0, 0, 0, 0};
        return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c, 4, 4, 0, 132, 0, 132);
    };
    auto prod = [&](v16f& a0, v16f& a1) {
        if (MODE == 3) { a0 = mf(T[0], Q[0][0], row); a1 = mf(T[0], Q[1][0], row); }
        else { a0 = mf(T[0], Q[0][0], a0); a1 = mf(T[0], Q[1][0], a1); }
#pragma unroll
        for (int s = 1; s < 4; s++) { a0 = mf(T[s], Q[0][s], a0); a1 = mf(T[s], Q[1][s], a1); }
    };
    for (int it = 0; it < iters; it += 2) {
        for (int s = 0; s < 4; s++) asm volatile("" : "+v"(T[s]));
        prod(e0, e1); fold(o0, o1);
        __builtin_amdgcn_sched_barrier(0);
        for (int s = 0; s < 4; s++) asm volatile("" : "+v"(T[s]));
        prod(o0, o1); fold(e0, e1);
        __builtin_amdgcn_sched_barrier(0);
    }
    float r = b0 + s0 + b1 + s1;
    for (int i = 0; i < 16; i++) r += e0[i] + e1[i] + o0[i] + o1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE>
static void run6(float* d, int wgPerCu)
{
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        k6<MODE><<<256 * wgPerCu, 256>>>(d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("8 MFMA + fold, 12 operand fragments of their own%s, waves/SIMD=%d: %.1f ns per iteration of the SIMD\n",
           MODE == 3 ? ", chains start from a resident constant set" : "", wgPerCu, ms * 1e6 / iters);
}

int main()
{
    float* d; hipMalloc(&d, 256 * 2 * 256 * 4); hipMemset(d, 0, 256 * 2 * 256 * 4);
    for (int w = 1; w <= 2; w++) {
        run<0, true>(d, w, "8 MFMA");
        run<32, false>(d, w, "VALU only");
        run<64, false>(d, w, "VALU only");
        run<16, true>(d, w, "8 MFMA + VALU");
        run<32, true>(d, w, "8 MFMA + VALU");
        run<48, true>(d, w, "8 MFMA + VALU");
        run<64, true>(d, w, "8 MFMA + VALU");
        run<96, true>(d, w, "8 MFMA + VALU");
    }
    run5<0>(d, 1); run5<1>(d, 1); run5<0>(d, 2); run5<1>(d, 2);
    run6<2>(d, 2); run6<3>(d, 2);
    run4<0>(d); run4<64>(d); run4<96>(d); run4<128>(d); run4<160>(d);
    return 0;
}
