// How fast does ONE workgroup run on an otherwise idle GPU?  The live-stream chain is a string of small launches (one to a
// few dozen workgroups each): is a dependent VALU chain / a dependent LDS chain as fast there as with the GPU busy?
//   hipcc --offload-arch=gfx950 -O3 -o build_ub/light_load_clock tools/ubench/light_load_clock.hip
// Prints ns per dependent v_add / per dependent ds_read for: a launch alone (cold, after 100 ms of idling), the 10th of a
// back-to-back string of launches, and a launch beside a kernel that keeps every CU busy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_chain(int n, int waves_dummy, unsigned long long* out, int* sink)
{
    __shared__ int tab[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) tab[i] = (i * 17 + 5) & 1023;
    __syncthreads();
    unsigned long long t0 = wall_clock64(), c0 = clock64();
    int v = threadIdx.x;
#pragma unroll 16
    for (int i = 0; i < n; i++) v = v * 3 + 1;          // dependent: one v_mad / v_lshl_add per step
    unsigned long long t1 = wall_clock64(), c1 = clock64();
    int a = threadIdx.x & 1023;
#pragma unroll 8
    for (int i = 0; i < n; i++) a = tab[a];             // dependent LDS reads
    unsigned long long t2 = wall_clock64(), c2 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = t2 - t1; out[2] = c1 - c0; out[3] = c2 - c1; }
    if (v == 12345 && a == 7) *sink = v;
}

__global__ void k_busy(float* p, int iters)
{
    float x = p[threadIdx.x];
    for (int i = 0; i < iters; i++) x = x * 1.0001f + 0.5f;
    p[threadIdx.x] = x;
}

int main()
{
    unsigned long long* out; int* sink; float* p;
    CK(hipHostMalloc(&out, 64)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&p, 4096));
    CK(hipMemset(p, 0, 4096));
    hipStream_t s, s2; CK(hipStreamCreate(&s)); CK(hipStreamCreate(&s2));
    const int n = 4096;
    auto show = [&](const char* what, int threads) {
        printf("%-44s %4d threads: %.2f ns per dependent VALU op (%.2f 'cycles'), %.1f ns per dependent LDS read (%.1f 'cycles')\n", what, threads,
               out[0] * 10.0 / n, (double)out[2] / n, out[1] * 10.0 / n, (double)out[3] / n);
    };
    for (int threads : {64, 512, 1024}) {
        for (int rep = 0; rep < 2; rep++) {
            std::this_thread::sleep_for(std::chrono::milliseconds(200));
            hipLaunchKernelGGL(k_chain, dim3(1), dim3(threads), 0, s, n, 0, out, sink); CK(hipStreamSynchronize(s));
            show("alone, after 200 ms idle", threads);
        }
        for (int i = 0; i < 50; i++) hipLaunchKernelGGL(k_chain, dim3(1), dim3(threads), 0, s, n, 0, out, sink);
        CK(hipStreamSynchronize(s));
        show("50th of a back-to-back string", threads);
        hipLaunchKernelGGL(k_busy, dim3(256 * 8), dim3(256), 0, s2, p, 4000000);
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
        hipLaunchKernelGGL(k_chain, dim3(1), dim3(threads), 0, s, n, 0, out, sink); CK(hipStreamSynchronize(s));
        show("beside a kernel that fills every CU", threads);
        CK(hipStreamSynchronize(s2));
    }
    return 0;
}
