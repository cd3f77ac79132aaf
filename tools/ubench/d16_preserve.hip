// does a d16 LDS load keep the other half of its destination VGPR on this GPU (gfx950 runs with SRAM ECC, where LLVM
// assumes it does not: d16PreservesUnusedBits() is false, so the compiler never pairs two byte loads in one register)?
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/d16_preserve.hip -o build_ub/d16_preserve
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint32_t* out)
{
    __shared__ uint8_t s[256];
    s[threadIdx.x] = (uint8_t)(threadIdx.x + 1);
    __syncthreads();
    const uint32_t a0 = (uint32_t)(uintptr_t)&s[0] + threadIdx.x, a1 = (uint32_t)(uintptr_t)&s[0] + ((threadIdx.x + 7) & 63);
    uint32_t v = 0xAAAAAAAAu, w = 0xAAAAAAAAu, x = 0xAAAAAAAAu;
    asm volatile("ds_read_u8_d16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(v) : "v"(a0) : "memory");
    const uint32_t v1 = v;
    asm volatile("ds_read_u8_d16_hi %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(v) : "v"(a1) : "memory");
    asm volatile("ds_read_u8_d16_hi %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(w) : "v"(a1) : "memory");
    const uint32_t w1 = w;
    asm volatile("ds_read_u8_d16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(w) : "v"(a0) : "memory");
    // both in flight at once
    asm volatile("ds_read_u8_d16 %0, %1\n\tds_read_u8_d16_hi %0, %2\n\ts_waitcnt lgkmcnt(0)" : "+v"(x) : "v"(a0), "v"(a1) : "memory");
    out[5 * threadIdx.x + 0] = v1; out[5 * threadIdx.x + 1] = v; out[5 * threadIdx.x + 2] = w1; out[5 * threadIdx.x + 3] = w; out[5 * threadIdx.x + 4] = x;
}
int main()
{
    uint32_t* d; hipMalloc(&d, 64 * 5 * 4); uint32_t h[320];
    k<<<1, 64>>>(d); hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int ok = 0;
    for (int t = 0; t < 64; t++) {
        const uint32_t want = (uint32_t)(t + 1) | ((uint32_t)(((t + 7) & 63) + 1) << 16);
        ok += h[5 * t + 1] == want && h[5 * t + 3] == want && h[5 * t + 4] == want;
    }
    printf("lane 3: after d16 %08x, then d16_hi %08x | after d16_hi %08x, then d16 %08x | both in flight %08x ; lanes with both halves intact in all three orders: %d / 64\n", h[15], h[16], h[17], h[18], h[19], ok);
    return 0;
}
