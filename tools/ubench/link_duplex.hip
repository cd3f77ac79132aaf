// link_duplex: what the PCIe link gives the host entries' two directions when both are busy -- frames up (30.8 MB per 64-frame
// batch) beside results down (8.5 MB) -- for every pairing of {DMA engine, copy kernel} per direction.  bench.py's
// host_path.pcie_peak_gbs is the engine/engine pairing (54.6 up beside 15.1 down); the library's throughput mode runs
// engine up beside a kernel down (k_pack_host on an 8-CU queue).  This program says whether another pairing is worth building.
//   hipcc --offload-arch=gfx950 -O3 -o build_ub/link_duplex tools/ubench/link_duplex.hip ; gpurun -- build_ub/link_duplex
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_copy(const uint4* __restrict__ s, uint4* __restrict__ d, int n16)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += gridDim.x * blockDim.x) d[i] = s[i];
}
__global__ void k_copy_nt(const uint4* __restrict__ s, uint4* __restrict__ d, int n16)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += gridDim.x * blockDim.x) {
        const uint4 v = s[i];
        __builtin_nontemporal_store(v.x, &d[i].x); __builtin_nontemporal_store(v.y, &d[i].y);
        __builtin_nontemporal_store(v.z, &d[i].z); __builtin_nontemporal_store(v.w, &d[i].w);
    }
}

struct Dir {
    const char* name; bool kernel; int blocks; bool nt; hipStream_t st; const void* src; void* dst; size_t bytes;
    void go() const {
        if (!kernel) CK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, st));
        else if (nt) hipLaunchKernelGGL(k_copy_nt, dim3(blocks), dim3(256), 0, st, (const uint4*)src, (uint4*)dst, (int)(bytes / 16));
        else hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, st, (const uint4*)src, (uint4*)dst, (int)(bytes / 16));
    }
};

// both directions busy from t0 on; a direction's rate = bytes of the repetitions that ended inside the common window
static void pair(const Dir* up, const Dir* dn, int repsUp, int repsDn)
{
    hipEvent_t t0a, t0b; CK(hipEventCreate(&t0a)); CK(hipEventCreate(&t0b));
    std::vector<hipEvent_t> eu(up ? repsUp : 0), ed(dn ? repsDn : 0);
    for (auto& e : eu) CK(hipEventCreate(&e));
    for (auto& e : ed) CK(hipEventCreate(&e));
    if (up) { up->go(); CK(hipStreamSynchronize(up->st)); }
    if (dn) { dn->go(); CK(hipStreamSynchronize(dn->st)); }
    if (up) CK(hipEventRecord(t0a, up->st));
    if (dn) CK(hipEventRecord(t0b, dn->st));
    for (int i = 0; i < std::max(repsUp, repsDn); i++) {
        if (up && i < repsUp) { up->go(); CK(hipEventRecord(eu[i], up->st)); }
        if (dn && i < repsDn) { dn->go(); CK(hipEventRecord(ed[i], dn->st)); }
    }
    CK(hipDeviceSynchronize());
    auto ends = [&](std::vector<hipEvent_t>& ev, hipEvent_t t0) { std::vector<float> r; for (auto& e : ev) { float ms; CK(hipEventElapsedTime(&ms, t0, e)); r.push_back(ms); } return r; };
    std::vector<float> tu = up ? ends(eu, t0a) : std::vector<float>(), td = dn ? ends(ed, t0b) : std::vector<float>();
    float win = 1e30f;
    if (up) win = std::min(win, tu.back());
    if (dn) win = std::min(win, td.back());
    auto rate = [&](const std::vector<float>& t, size_t bytes) { int n = 0; float last = 0; for (float x : t) if (x <= win * 1.0001f) { n++; last = x; } return n ? (double)n * bytes / (last * 1e-3) / 1e9 : 0.0; };
    const double ru = up ? rate(tu, up->bytes) : 0, rd = dn ? rate(td, dn->bytes) : 0;
    // a 64-frame batch needs both: frames/s the link allows = 64 / max(t_up, t_down)
    const double tb = std::max(up ? 30801920.0 / (ru * 1e9) : 0.0, dn ? 8519680.0 / (rd * 1e9) : 0.0);
    printf("%-28s %-28s up %6.1f GB/s  down %6.1f GB/s  -> %6.1f k frames/s\n", up ? up->name : "-", dn ? dn->name : "-", ru, rd, 64.0 / tb / 1e3);
    for (auto& e : eu) CK(hipEventDestroy(e));
    for (auto& e : ed) CK(hipEventDestroy(e));
    CK(hipEventDestroy(t0a)); CK(hipEventDestroy(t0b));
}

int main()
{
    const size_t upB = 30801920, dnB = 8519680;   // bench.py host_path: one batch's frames / results
    void *hUp, *hDn, *hDnC, *dUp, *dDn;
    CK(hipHostMalloc(&hUp, upB)); CK(hipHostMalloc(&hDn, dnB)); CK(hipHostMalloc(&hDnC, dnB, hipHostMallocCoherent));
    CK(hipMalloc(&dUp, upB)); CK(hipMalloc(&dDn, dnB));
    memset(hUp, 1, upB); memset(hDn, 0, dnB); memset(hDnC, 0, dnB);
    hipStream_t sa, sb, sm8;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    std::vector<uint32_t> mask((pr.multiProcessorCount + 31) / 32, 0u); mask[0] = 0xFFu;
    if (hipExtStreamCreateWithCUMask(&sm8, (uint32_t)mask.size(), mask.data()) != hipSuccess) { (void)hipGetLastError(); sm8 = sb; }
    void *dhUp = nullptr, *dhDn = nullptr, *dhDnC = nullptr;
    CK(hipHostGetDevicePointer(&dhUp, hUp, 0)); CK(hipHostGetDevicePointer(&dhDn, hDn, 0)); CK(hipHostGetDevicePointer(&dhDnC, hDnC, 0));

    std::vector<Dir> ups = {
        {"up: engine", false, 0, false, sa, hUp, dUp, upB},
        {"up: kernel 256 blocks", true, 256, false, sa, dhUp, dUp, upB},
        {"up: kernel 1024 blocks", true, 1024, false, sa, dhUp, dUp, upB},
        {"up: kernel 64 blocks", true, 64, false, sa, dhUp, dUp, upB},
    };
    std::vector<Dir> dns = {
        {"down: engine", false, 0, false, sb, dDn, hDn, dnB},
        {"down: kernel 128 blocks", true, 128, false, sb, dDn, dhDn, dnB},
        {"down: kernel 128 bl, 8 CUs", true, 128, false, sm8, dDn, dhDn, dnB},
        {"down: kernel 128 bl coherent", true, 128, false, sb, dDn, dhDnC, dnB},
        {"down: kernel 128 bl nt", true, 128, true, sb, dDn, dhDn, dnB},
        {"down: kernel 1024 blocks", true, 1024, false, sb, dDn, dhDn, dnB},
        {"down: kernel 16 blocks", true, 16, false, sb, dDn, dhDn, dnB},
    };
    printf("# alone\n");
    for (auto& u : ups) pair(&u, nullptr, 30, 0);
    for (auto& d : dns) pair(nullptr, &d, 0, 60);
    printf("# both busy (rates inside the common window)\n");
    for (auto& u : ups) for (auto& d : dns) pair(&u, &d, 30, 110);
    return 0;
}
