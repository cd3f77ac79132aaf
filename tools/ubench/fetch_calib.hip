// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on this box for the access widths the ORB kernels use
// (MI355X_MICROARCH.md, HBM: "calibrate on a known byte count in your own access pattern"): streaming reads of
// 512 MiB (beyond the 256 MiB Infinity Cache) with 1, 4 and 16 bytes per lane, and a 4 B/lane streaming write.
// build: hipcc --offload-arch=gfx950 -O3 -o build_ub/fetch_calib tools/ubench/fetch_calib.hip
// run:   rocprofv3 --pmc FETCH_SIZE -d out -- build_ub/fetch_calib ; rocprofv3 --pmc WRITE_SIZE -d out2 -- build_ub/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void read_b1(const uint8_t* p, size_t n, uint32_t* o) { uint32_t s = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i]; if (s == 0x12345678u) o[0] = s; }
__global__ void read_b4(const uint32_t* p, size_t n, uint32_t* o) { uint32_t s = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i]; if (s == 0x12345678u) o[0] = s; }
__global__ void read_b16(const uint4* p, size_t n, uint32_t* o) { uint32_t s = 0; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; s += v.x ^ v.y ^ v.z ^ v.w; } if (s == 0x12345678u) o[0] = s; }
__global__ void write_b4(uint32_t* p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i; }
int main()
{
    const size_t bytes = 512ull << 20;
    uint8_t* d; uint32_t* o;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&o, 64) != hipSuccess) return 1;
    hipMemset(d, 1, bytes);
    hipDeviceSynchronize();
    read_b1<<<4096, 256>>>(d, bytes / 8, o);          // 64 MiB at 1 B/lane
    read_b4<<<4096, 256>>>((const uint32_t*)d, bytes / 4, o);
    read_b16<<<4096, 256>>>((const uint4*)d, bytes / 16, o);
    write_b4<<<4096, 256>>>((uint32_t*)d, bytes / 4);
    hipDeviceSynchronize();
    printf("read_b1 %zu MiB, read_b4 %zu MiB, read_b16 %zu MiB, write_b4 %zu MiB\n", bytes / 8 >> 20, bytes >> 20, bytes >> 20, bytes >> 20);
    return 0;
}
