// micro-benchmark: does the LDS serve unaligned 64-bit and 32-bit reads on gfx950 (HSA sets SH_MEM_CONFIG.ALIGNMENT_MODE = unaligned)?
// Every lane reads 8 bytes at a byte address of its own (stride 5, every residue mod 8) and compares with the bytes.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/lds_unaligned.hip -o build_ub/lds_unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
struct __attribute__((packed)) U64 { uint64_t v; };
struct __attribute__((packed)) U32 { uint32_t v; };
__global__ void k(uint32_t* out, int mode, int iters, uint32_t* dbg)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (uint8_t)(i * 37 + (i >> 8) * 11 + 5);
    __syncthreads();
    uint32_t bad = 0;
    uint64_t acc = 0;
    for (int it = 0; it < iters; it++) {
        const int a = (threadIdx.x * 5 + it * 3) & 4095;
        uint64_t v;
        if (mode == 0) { asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(uintptr_t)(lds + a)) : "memory"); }
        else if (mode == 1) v = ((const U64*)(lds + a))->v;
        else {
            uint32_t lo, hi;
            asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:4\n\ts_waitcnt lgkmcnt(0)" : "=&v"(lo), "=&v"(hi) : "v"((uint32_t)(uintptr_t)(lds + a)) : "memory");
            v = lo | ((uint64_t)hi << 32);
        }
        uint64_t e = 0;
        for (int b = 0; b < 8; b++) { const int i = a + b; e |= (uint64_t)(uint8_t)(i * 37 + (i >> 8) * 11 + 5) << (8 * b); }  // (the pattern itself: hipcc merges byte reads of LDS into a ds_read_b64 of its own)
        if (v != e && dbg) { const uint32_t slot = atomicAdd(dbg, 1u); if (slot < 12) { dbg[1 + slot * 5] = a; dbg[2 + slot * 5] = (uint32_t)v; dbg[3 + slot * 5] = (uint32_t)(v >> 32); dbg[4 + slot * 5] = (uint32_t)e; dbg[5 + slot * 5] = (uint32_t)(e >> 32); } }
        bad += v != e;
        acc += v;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = bad + (uint32_t)(acc >> 63);
}
int main()
{
    uint32_t* d; (void)hipMalloc(&d, 256 * 4);
    uint32_t h[256];
    const char* names[3] = {"ds_read_b64 (asm)", "packed uint64_t load (compiler's choice)", "2 x ds_read_b32 (asm)"};
    for (int mode = 0; mode < 3; mode++) {
        (void)hipMemset(d, 0xFF, 256 * 4);
        uint32_t* dbg; (void)hipMalloc(&dbg, 64 * 4); (void)hipMemset(dbg, 0, 64 * 4);
        k<<<1, 256>>>(d, mode, 64, dbg);
        hipError_t e = hipDeviceSynchronize();
        (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        uint32_t bad = 0; for (int i = 0; i < 256; i++) bad += h[i];
        uint32_t hd[64]; (void)hipMemcpy(hd, dbg, sizeof hd, hipMemcpyDeviceToHost);
        for (uint32_t j = 0; j < hd[0] && j < 6; j++) printf("   a=%u (mod 8 = %u) got %08x%08x want %08x%08x\n", hd[1 + j * 5], hd[1 + j * 5] & 7, hd[3 + j * 5], hd[2 + j * 5], hd[5 + j * 5], hd[4 + j * 5]);
        printf("%-44s %s, mismatches %u of %d\n", names[mode], hipGetErrorString(e), bad, 256 * 64);
    }
    return 0;
}
