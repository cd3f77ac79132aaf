// orbx_kernels.hip -- gfx950 kernels of the ORB extractor hot path.
//
// Stage map (reference: /root/reference/SingleRobotScenario/src/ORBextractor.cc):
//   k_pyramid       ComputePyramid :1107-1132 (cv::resize INTER_LINEAR, fixed point; k_resize_level = per-level fallback)
//   k_fast          ComputeKeyPointsOctTree cell loop :789-829 (cv::FAST 9/16 + NMS + minTh retry)
//   k_distribute    DistributeOctTree :539-763 + DivideNode :481-537
//   k_blur_mfma     GaussianBlur 7x7 sigma 2 :1085-1086 (two banded int8 products on the matrix cores, exact in int32)
//   k_orient_desc   IC_Angle :77-104, computeOrbDescriptor :108-147, scale/pack :837-847,1095-1101
//
// Integer/bitwise path, VALU-issue bound: the kernels are shaped to shed instructions (dot4/dot2/perm/min3/med3,
// SDWA byte operands, compactions).  64-wide waves, LDS tiles, strict IEEE fp32 where the reference computes in
// float (no contraction: explicit __fmul_rn/__fadd_rn and -ffp-contract=off).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "orbx_common.hpp"

namespace orbx {

struct FrameSrc {
    const uint8_t* img0;   // level 0 = caller's frames
    int32_t stride0;
    int64_t pitch0;        // bytes between frames
    uint8_t* pyr;          // levels >= 1
    uint8_t* blur;         // all levels
    int32_t f0;            // first frame of this launch (sub-batches run on separate streams)
};

__device__ __forceinline__ const uint8_t* level_ptr(const Geom* g, const FrameSrc& s, int f, int l, int& stride)
{
    if (l == 0) { stride = s.stride0; return s.img0 + (int64_t)f * s.pitch0; }
    stride = g->lv[l].stride;
    return s.pyr + (int64_t)f * g->pyrFrameBytes + g->lv[l].pyrOff;
}

// XCD-aware (block, frame) mapping for launches of grid (blocksPerFrame, xcd_grid_y(frames)): workgroups are dealt
// round-robin to the 8 XCDs in linear order, so without this the neighbours of one frame (which share cache
// lines: cell aprons, blur halos, overlapping keypoint patches) land on 8 different L2s and every line is fetched
// up to 8 times.  Here XCD x works through frames x, x+8, ... one after the other.
__device__ __forceinline__ bool xcd_block_frame(int nframes, int& blk, int& fr)
{
    if (nframes < 8) { blk = blockIdx.x; fr = blockIdx.y; return true; }  // too few frames to give every XCD one: grid (blocksPerFrame, frames)
    const int total = gridDim.x;
    const int lin = blockIdx.y * total + blockIdx.x;
    const int k = lin >> 3;
    fr = (k / total) * 8 + (lin & 7);
    blk = k - (k / total) * total;
    return fr < nframes;
}

__device__ __forceinline__ uint64_t lanemask_lt()
{
    const uint32_t lane = threadIdx.x & 63;
    return lane == 0 ? 0ull : (~0ull >> (64 - lane));
}

// ------------------------------------------------------------------ pyramid
// xtab[dx] = {sx, a0, a1, interp?}, ytab[dy] = {sy0, sy1, b0, b1}; built on the host
// (orbx_api.hip: build_resize_tables) exactly as cv::resize builds xofs/ialpha/yofs/ibeta.
struct ResizeTabs {
    const short4* xtab[ORBX_MAXL];
    const short4* ytab[ORBX_MAXL];
};

// One level from the previous one in HBM.  Fallback for shapes/scale factors whose halo chain
// does not fit the LDS-tiled fused kernel below (very large frames, scale factors near 2).
__global__ __launch_bounds__(256) void k_resize_level(const Geom* __restrict__ g, FrameSrc src, ResizeTabs tabs, int level)
{
    const int lw = g->lv[level].w, lh = g->lv[level].h, dstStride = g->lv[level].stride;
    const int f = blockIdx.z + src.f0;
    const int x4 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int dy = blockIdx.y * 4 + threadIdx.y;
    if (x4 >= lw || dy >= lh) return;
    int sstride;
    const uint8_t* S = level_ptr(g, src, f, level - 1, sstride);
    uint8_t* D = src.pyr + (int64_t)f * g->pyrFrameBytes + g->lv[level].pyrOff + (int64_t)dy * dstStride;
    const short4 yt = tabs.ytab[level][dy];
    const uint8_t* S0 = S + (int64_t)yt.x * sstride;
    const uint8_t* S1 = S + (int64_t)yt.y * sstride;
    const int b0 = yt.z, b1 = yt.w;
    uint32_t packed = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const short4 xt = tabs.xtab[level][min(x4 + i, lw - 1)];
        const int sx = (uint16_t)xt.x;
        int r0, r1;
        if (xt.w) {
            r0 = S0[sx] * xt.y + S0[sx + 1] * xt.z;
            r1 = S1[sx] * xt.y + S1[sx + 1] * xt.z;
        } else {
            r0 = S0[sx] * 2048;
            r1 = S1[sx] * 2048;
        }
        const int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
        packed |= (uint32_t)(v & 0xFF) << (8 * i);
    }
    *(uint32_t*)(D + x4) = packed;  // rows are 64-byte aligned and padded
}

// ------------------------------------------------------------------ fused pyramid
// All 7 resizes in ONE launch.  A block owns the same fractional rectangle of every
// level; per level it computes the pixels it owns plus the small halo the next level
// will read (ranges are chained top-down on the host, PyrRange), keeps the tile in LDS
// (ping-pong) and writes only the owned pixels to HBM.  The level-0 source tile is
// staged with aligned dword loads.  Same fixed-point arithmetic as k_resize.
constexpr int kPyrStrips = 2;   // level-0 tile of k_pyramid: strips per block (host sizing and kernel)
#ifndef ORBX_DESC_PRIO
#define ORBX_DESC_PRIO 0
#endif
#ifndef ORBX_BLUR_PRIO
#define ORBX_BLUR_PRIO 0
#endif
#ifndef ORBX_DIST_PRIO
#define ORBX_DIST_PRIO 0
#endif
#ifndef ORBX_PYR_PRIO
#define ORBX_PYR_PRIO 3
#endif
#ifndef ORBX_PYR_THREADS
#define ORBX_PYR_THREADS 256
#endif
constexpr int kPyrThreads = ORBX_PYR_THREADS;   // threads of a k_pyramid workgroup (host launch and kernel)
struct PyrRange {
    int16_t ox0, ox1, oy0, oy1;   // owned output range at this level (exclusive ends)
    int16_t nx0, nx1, ny0, ny1;   // computed range (owned + halo needed by the next level)
};

__global__ __launch_bounds__(kPyrThreads) void k_pyramid(const Geom* __restrict__ g, FrameSrc src, ResizeTabs tabs,
                                                const PyrRange* __restrict__ ranges, int bufAWords, int bufBWords,
                                                int tabCap, int xcdFrames)
{
    // one LDS array addressed with integer offsets (keeps every access in the LDS address space)
    extern __shared__ __attribute__((aligned(16))) uint32_t plds[];
#if ORBX_PYR_PRIO
    __builtin_amdgcn_s_setprio(ORBX_PYR_PRIO);
#endif
    uint8_t* const ldsb = (uint8_t*)plds;
    const int offBuf[2] = {0, bufAWords * 4};            // even / odd levels (bytes)
    uint2* const sxt = (uint2*)(plds + bufAWords + bufBWords);  // staged {sx,a0 | a1,interp}
    uint2* const syt = sxt + tabCap;                              // staged {sy0,sy1 | b0,b1}
    // xcdFrames >= 8: grid (blocks, xcd_grid_y(frames)) -- a frame's blocks on the XCD that runs its FAST cells, blur tiles and
    // descriptors (the plain grid puts block b of every frame on XCD b mod 8)
    int pblk = (int)blockIdx.x, pfr = (int)blockIdx.y;
    if (xcdFrames >= 8 && !xcd_block_frame(xcdFrames, pblk, pfr)) return;
    const int f = pfr + src.f0;
    const int nl = g->nlevels;
    const PyrRange* R = ranges + (int64_t)pblk * nl;
    const int tid = threadIdx.x;

    // Level 0 is the caller's frame: its tile (the largest, 22 KB of the 41 KB this kernel used to hold -- exactly a
    // quarter of a CU's LDS, so that a workgroup could only be placed on a CU where FAST cells had drained 40 KB) is
    // staged in kPyrStrips horizontal strips, each loaded right before the level-1 rows that read it.
    PyrRange p = R[0];
    int pstride = 0;
    const bool have0 = p.nx1 > p.nx0 && p.ny1 > p.ny0;
    int stride0 = 0;
    const uint8_t* S0 = nullptr;
    int ndw0 = 0;
    if (have0) {
        const uint8_t* S = level_ptr(g, src, f, 0, stride0);
        ndw0 = (p.nx1 - p.nx0 + 3) >> 2;
        pstride = ndw0 * 4;
        S0 = S + p.nx0;
    }
    // rows [r0, r1) of level 0 into buffer A: aligned dword loads (ranges have 4-aligned x origins).  A strip of up to
    // 4096 dwords travels through 16 registers per thread: fetched (loads issued) before the previous strip's rows are
    // computed, committed to LDS after -- only the first strip's latency is exposed.  Larger strips load in place.
    constexpr int T = kPyrThreads;
    constexpr int kStripRegs = 4096 / T;   // a strip of up to 4096 dwords through the registers
    uint32_t sreg[kStripRegs];
    auto strip_rows = [&](int strip, int chh1, const uint2* yt, int& r0, int& r1) {  // level-0 rows read by the strip's level-1 rows
        const int ya = (int)((int64_t)chh1 * strip / kPyrStrips), yb = (int)((int64_t)chh1 * (strip + 1) / kPyrStrips);
        r0 = r1 = 0;
        if (yb > ya) { r0 = (int16_t)(yt[ya].x & 0xFFFF); r1 = min((int)(int16_t)(yt[yb - 1].x >> 16) + 1, (int)p.ny1); }
    };
    // element k * T + tid of a strip is (row, dword column); stepping by 256 elements without a division per load
    const int q256 = have0 ? T / ndw0 : 0, m256 = have0 ? T - q256 * ndw0 : 0;
    const int rowT = have0 ? tid / ndw0 : 0, colT = have0 ? tid - rowT * ndw0 : 0;
    auto fetch0 = [&](int r0, int r1) {
        const int total = (r1 - r0) * ndw0;
        if (total <= 0 || total > T * kStripRegs) return;
        const uint8_t* Sr = S0 + (int64_t)r0 * stride0;
        int r = rowT, c = colT;
#pragma unroll
        for (int k = 0; k < kStripRegs; k++) {
            if (k * T < total) {   // uniform
                if (k * T + tid < total) sreg[k] = *(const uint32_t*)(Sr + (int64_t)r * stride0 + 4 * c);
                r += q256; c += m256;
                if (c >= ndw0) { c -= ndw0; r++; }
            }
        }
    };
    auto commit0 = [&](int r0, int r1) {
        const int total = (r1 - r0) * ndw0;
        if (total <= 0) return;
        if (total <= T * kStripRegs) {
#pragma unroll
            for (int k = 0; k < kStripRegs; k++) {
                const int i = k * T + tid;
                if (i < total) plds[i] = sreg[k];
            }
            return;
        }
        const uint8_t* Sr = S0 + (int64_t)r0 * stride0;
        for (int i0 = 0; i0 < total; i0 += T * 8) {
            uint32_t regs[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int i = min(i0 + k * T + tid, total - 1);
                const int r = i / ndw0, c = i - r * ndw0;
                regs[k] = *(const uint32_t*)(Sr + (int64_t)r * stride0 + 4 * c);
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int i = i0 + k * T + tid;
                if (i < total) plds[i] = regs[k];
            }
        }
    };
    for (int l = 1; l < nl; l++) {
        const PyrRange c = R[l];
        const int cw = c.nx1 - c.nx0, chh = c.ny1 - c.ny0;
        const int gpr = (cw + 3) >> 2;           // dword groups per row
        const int cstride = gpr * 4;
        const int lw = g->lv[l].w, dstStride = g->lv[l].stride;
        uint8_t* const D = src.pyr + (int64_t)f * g->pyrFrameBytes + g->lv[l].pyrOff;
        const int pnx0 = p.nx0, pny0 = p.ny0;
        const uint2* __restrict__ gx_tab = (const uint2*)tabs.xtab[l];
        const uint2* __restrict__ gy_tab = (const uint2*)tabs.ytab[l];
        // stage this level's coefficient rows (entries past the level width are never used)
        for (int i = tid; i < cstride; i += T) sxt[i] = gx_tab[min(c.nx0 + i, lw - 1)];
        for (int i = tid; i < chh; i += T) syt[i] = gy_tab[c.ny0 + i];
        __syncthreads();  // also orders the previous level's tile writes before the reads below
        const int offP = offBuf[(l - 1) & 1], offC = offBuf[l & 1];
        const bool strips = l == 1 && have0 && chh > 0;
        const int nstrips = strips ? kPyrStrips : 1;
        if (strips) { int r0, r1; strip_rows(0, chh, syt, r0, r1); fetch0(r0, r1); }
        for (int strip = 0; strip < nstrips; strip++) {
        // rows [ya, yb) of this level; for level 1 they read the level-0 rows [prow0, ..) of the strip committed just now
        const int ya = strips ? (int)((int64_t)chh * strip / nstrips) : 0, yb = strips ? (int)((int64_t)chh * (strip + 1) / nstrips) : chh;
        int prow0 = pny0;
        if (strips) {
            int r0, r1;
            strip_rows(strip, chh, syt, r0, r1);
            prow0 = r0;
            if (strip) __syncthreads();   // the previous strip's rows are read
            commit0(r0, r1);
            __syncthreads();
            if (strip + 1 < nstrips) { int q0, q1; strip_rows(strip + 1, chh, syt, q0, q1); fetch0(q0, q1); }
        }
        if (gpr > 0 && yb > ya) {
            // thread = (dword group gx, row lane): the four x-coefficient entries of the group are unpacked once
            // and reused down the rows the thread owns (rows yy0, yy0 + dr, ...)
          for (int gxb = 0; gxb < gpr; gxb += T) {  // one pass unless the tile is wider than 1024 px
            const int cols = min(T, gpr - gxb);
            const int dr = T / cols;           // row lanes
            const int yy0 = tid / cols, gx = gxb + tid - yy0 * cols;
            if (yy0 < dr) {
                int sx[4], a0[4], a1[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint2 xt = sxt[4 * gx + i];
                    sx[i] = (int)(xt.x & 0xFFFF) - pnx0;
                    a0[i] = (int16_t)(xt.x >> 16);
                    a1[i] = (int16_t)(xt.y & 0xFFFF);
                }
                const int dx = c.nx0 + 4 * gx;
                const bool ownX = dx >= c.ox0 && dx < c.ox1;
                for (int yy = ya + yy0; yy < yb; yy += dr) {
                    const uint2 yt = syt[yy];
                    const int sy0 = (int16_t)(yt.x & 0xFFFF), sy1 = (int16_t)(yt.x >> 16);
                    const int b0 = (int16_t)(yt.y & 0xFFFF), b1 = (int16_t)(yt.y >> 16);
                    const int o0 = offP + (sy0 - prow0) * pstride;
                    const int o1 = offP + (sy1 - prow0) * pstride;
                    uint32_t packed = 0;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        // columns past xmax (cv::resize's `D[dx] = S[sx] * ONE` tail) carry a0 = 2048, a1 = 0 in the table,
                        // so the two-tap form is exact there too (the second tap may read the tile's padding: times zero) --
                        // no per-pixel branch, all sixteen LDS reads of the row in flight together
                        const int s00 = ldsb[o0 + sx[i]], s10 = ldsb[o1 + sx[i]];
                        const int r0 = s00 * a0[i] + ldsb[o0 + sx[i] + 1] * a1[i];
                        const int r1 = s10 * a0[i] + ldsb[o1 + sx[i] + 1] * a1[i];
                        const int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
                        packed |= (uint32_t)(v & 0xFF) << (8 * i);
                    }
                    plds[(offC >> 2) + yy * gpr + gx] = packed;
                    const int dy = c.ny0 + yy;
                    if (ownX && dy >= c.oy0 && dy < c.oy1)
                        *(uint32_t*)(D + (int64_t)dy * dstStride + dx) = packed;  // own ranges are 4-aligned in x
                }
            }
          }
        }
        }  // strips
        p = c;
        pstride = cstride;
        __syncthreads();
    }
}

// ------------------------------------------------------------------ FAST
// One wave per reference cell.  S(p) = max over the 16 arcs of 9 contiguous ring
// pixels of min(v - p_k) resp. min(p_k - v): the pixel is a FAST-9 corner at
// threshold t iff S > t, and cv::FAST's cornerScore is then S - 1 (threshold
// independent), so the minThFAST retry re-thresholds the same S map.
//
// The kernel is VALU-issue bound (profiles/r01_pmc_sq_*.txt), so it is organised to shed
// instructions, in three compactions:
//   stage 1  compass test on every pixel, four pixels per lane from aligned LDS dwords: a
//            9-arc holds two adjacent compass pixels, i.e. one of {N,S} and one of {E,W},
//            all brighter than v+tq or all darker than v-tq.  Survivors are compacted into
//            a list tagged with their side; the few that pass on both sides go to a second
//            list growing down from the top of the same buffer (stage 2 visits them once per side).
//   stage 2  exact score on the dense lists.  A bright and a dark 9-arc cannot coexist (two
//            9-arcs of a 16-ring overlap), so one-sided survivors only evaluate their own
//            side: d_k = +-(v - p_k), S = max(0, max_k min(d_k..d_k+8)).  Pixels with S > tq
//            are compacted again, in place.
//   stage 3  3x3 non-max suppression and emission run on that corner list only.
// Window-9 minima of the 16-ring with three-input min: m3[k] = min3(d[k..k+2]),
// m9[k] = min3(m3[k], m3[k+3], m3[k+6]) -- the two-sided form of the cell whose one-sided lists do not fit (fast_S).
__device__ __forceinline__ int min3i(int a, int b, int c) { return min(min(a, b), c); }
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }

// ring offsets for a tile of row stride TSB bytes (compile-time: they become DS immediates)
template <int TSB, int K> struct RingOff {
    static constexpr int cx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
    static constexpr int cy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
    static constexpr int v = cx[K] + cy[K] * TSB;
};

__device__ __forceinline__ int arc_max_of_min(const int (&d)[16])
{
    int m3[16], m9[16];
#pragma unroll
    for (int k = 0; k < 16; k++) m3[k] = min3i(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
#pragma unroll
    for (int k = 0; k < 16; k++) m9[k] = min3i(m3[k], m3[(k + 3) & 15], m3[(k + 6) & 15]);
    int a5[5];
#pragma unroll
    for (int k = 0; k < 5; k++) a5[k] = max3i(m9[3 * k], m9[3 * k + 1], m9[3 * k + 2]);
    return max(max3i(a5[0], a5[1], a5[2]), max3i(a5[3], a5[4], m9[15]));
}

template <int TSB> __device__ __forceinline__ void ring_load(const uint8_t* __restrict__ p, int (&r)[16])
{
    r[0] = p[RingOff<TSB, 0>::v];   r[1] = p[RingOff<TSB, 1>::v];   r[2] = p[RingOff<TSB, 2>::v];   r[3] = p[RingOff<TSB, 3>::v];
    r[4] = p[RingOff<TSB, 4>::v];   r[5] = p[RingOff<TSB, 5>::v];   r[6] = p[RingOff<TSB, 6>::v];   r[7] = p[RingOff<TSB, 7>::v];
    r[8] = p[RingOff<TSB, 8>::v];   r[9] = p[RingOff<TSB, 9>::v];   r[10] = p[RingOff<TSB, 10>::v]; r[11] = p[RingOff<TSB, 11>::v];
    r[12] = p[RingOff<TSB, 12>::v]; r[13] = p[RingOff<TSB, 13>::v]; r[14] = p[RingOff<TSB, 14>::v]; r[15] = p[RingOff<TSB, 15>::v];
}

// Round 5: the same window minima two ring pixels per instruction.  gfx950's v_pk_minimum3_f16 / v_pk_maximum3_f16
// (IEEE-754-2019 minimum / maximum on two f16 lanes) issue at the full VALU rate, and on bytes widened to 16 bits they ARE
// the integer min3 / max3: positive f16 bit patterns order like the integers they spell, 0..255 are denormals, and the
// kernel's mode keeps f16 denormals (tools/ubench/pk_min3_rate.hip: all 3 x 2^24 byte triples exact, 4.8 cycles per wave
// instruction -- v_min3_i32's rate; v_min3_u16, the integer spelling, takes 8.5).  Ring pixel k rides with its antipode:
// X[k] = p[k] | p[k+8] << 16, so "index + 8" is an exchange of the two halves, which VOP3P's op_sel does for free in the
// operand fetch -- the 16 windows of three are 8 instructions, the 16 windows of nine 8 more, the fold over them 5.
// 8 (pairing) + 21 instead of 40 per visit; the visit's other ~25 instructions (list entry, addresses, ballot, stores) stay.
// operand j's halves are exchanged when bit j of SW is set (op_sel = low lane's source half, op_sel_hi = high lane's)
template <int SW> __device__ __forceinline__ uint32_t pk_min3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    if (SW == 0) asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    else if (SW == 4) asm("v_pk_minimum3_f16 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    else if (SW == 6) asm("v_pk_minimum3_f16 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    else asm("v_pk_minimum3_f16 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    static_assert(SW == 0 || SW == 2 || SW == 4 || SW == 6, "exchange masks in use");
    return r;
}
template <int SW> __device__ __forceinline__ uint32_t pk_max3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    if (SW == 0) asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    else if (SW == 4) asm("v_pk_maximum3_f16 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    else if (SW == 6) asm("v_pk_maximum3_f16 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    else asm("v_pk_maximum3_f16 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    static_assert(SW == 0 || SW == 2 || SW == 4 || SW == 6, "exchange masks in use");
    return r;
}
// max over the 16 arcs of 9 of the min over the arc (MINMAX = true) or min of max (false), of the ring in r[]
template <bool MAXMIN> __device__ __forceinline__ int arc_fold_pk(const int (&r)[16])
{
    uint32_t X[8], M3[8], M9[8];
#pragma unroll
    for (int k = 0; k < 8; k++) X[k] = (uint32_t)r[k] | ((uint32_t)r[k + 8] << 16);
#define ORBX_IN(SW, a, b, c) (MAXMIN ? pk_min3<SW>(a, b, c) : pk_max3<SW>(a, b, c))
#define ORBX_OUT(SW, a, b, c) (MAXMIN ? pk_max3<SW>(a, b, c) : pk_min3<SW>(a, b, c))
    // windows of three: M3[k] = (m3[k], m3[k+8]); X(j) for j >= 8 is X[j-8] with its halves exchanged
#pragma unroll
    for (int k = 0; k < 6; k++) M3[k] = ORBX_IN(0, X[k], X[k + 1], X[k + 2]);
    M3[6] = ORBX_IN(4, X[6], X[7], X[0]);
    M3[7] = ORBX_IN(6, X[7], X[0], X[1]);
    // windows of nine: M9[k] = (m9[k], m9[k+8]) = three windows of three, 3 apart
    M9[0] = ORBX_IN(0, M3[0], M3[3], M3[6]);
    M9[1] = ORBX_IN(0, M3[1], M3[4], M3[7]);
    M9[2] = ORBX_IN(4, M3[2], M3[5], M3[0]);
    M9[3] = ORBX_IN(4, M3[3], M3[6], M3[1]);
    M9[4] = ORBX_IN(4, M3[4], M3[7], M3[2]);
    M9[5] = ORBX_IN(6, M3[5], M3[0], M3[3]);
    M9[6] = ORBX_IN(6, M3[6], M3[1], M3[4]);
    M9[7] = ORBX_IN(6, M3[7], M3[2], M3[5]);
    const uint32_t A = ORBX_OUT(0, M9[0], M9[1], M9[2]);
    const uint32_t B = ORBX_OUT(0, M9[3], M9[4], M9[5]);
    const uint32_t C = ORBX_OUT(0, M9[6], M9[7], A);
    const uint32_t D = ORBX_OUT(4, B, C, B);      // low lane: B.lo, C.lo, B.hi
    const uint32_t E = ORBX_OUT(2, D, C, D);      // low lane: that and C.hi
#undef ORBX_IN
#undef ORBX_OUT
    return (int)(E & 0xFFFFu);
}

// one side only, on the ring's raw bytes (no difference per ring pixel):
//   dark arcs   S = max_arc min_k (v - p_k) = v - min_arc max_k p_k
//   bright arcs S = max_arc min_k (p_k - v) = max_arc min_k p_k - v
template <int TSB> __device__ __forceinline__ int fast_S_dark(const uint8_t* __restrict__ p)
{
    int r[16];
    ring_load<TSB>(p, r);
    return max((int)p[0] - arc_fold_pk<false>(r), 0);
}
template <int TSB> __device__ __forceinline__ int fast_S_bright(const uint8_t* __restrict__ p)
{
    int r[16];
    ring_load<TSB>(p, r);
    return max(arc_fold_pk<true>(r) - (int)p[0], 0);
}

// both sides (the cell whose one-sided lists would not fit)
template <int TSB> __device__ __forceinline__ int fast_S(const uint8_t* __restrict__ p)
{
    int r[16], d[16];
    ring_load<TSB>(p, r);
    const int v = p[0];
#pragma unroll
    for (int k = 0; k < 16; k++) d[k] = v - r[k];
    const int A = arc_max_of_min(d);
#pragma unroll
    for (int k = 0; k < 16; k++) d[k] = -d[k];
    return max(max(A, arc_max_of_min(d)), 0);
}

__device__ __forceinline__ uint64_t ballot64(bool b) { return __builtin_amdgcn_ballot_w64(b); }
__device__ __forceinline__ uint64_t tail_mask(int n) { return n >= 64 ? ~0ull : (1ull << n) - 1; }  // lanes < n (n > 0)
__device__ __forceinline__ int lanes_below(uint64_t bal)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0));
}

// Tile layout: ROI pixel (rx, ry) at byte ry*TSB + rx + 1, so detection pixel (x, y) = ROI (x+3, y+3)
// sits at (y+3)*TSB + x + 4 and groups of four detection pixels are dword aligned.
#ifdef ORBX_FAST_STATS
__device__ unsigned long long g_fastStats[16];  // per pass: cells, stage-2 visits, corners; [8] detection pixels
#endif
template <int TSB>
__global__ __launch_bounds__(64) void k_fast(const Geom* __restrict__ g, const Cell* __restrict__ cells,
                                            FrameSrc src, uint64_t* __restrict__ cand,
                                            int32_t* __restrict__ cellCount, int32_t* __restrict__ errFlag,
                                            int tileRows, int listCap, int smapPitch, int nframes, int cell0)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    constexpr int TSD = TSB / 4;
    int bx, fr;
    if (!xcd_block_frame(nframes, bx, fr)) return;
    bx += cell0;  // the launch covers cells [cell0, cell0 + gridDim.x)
    const Cell c = cells[bx];
    const int f = fr + src.f0;
    const int lane = threadIdx.x;
    const int level = c.level;
    const LevelGeom& L = g->lv[level];
    int stride;
    const uint8_t* base = level_ptr(g, src, f, level, stride);

#ifdef ORBX_FAST_STATS
    unsigned long long tk0 = wall_clock64(), tk1;
    const bool statCell = lane == 0 && (blockIdx.x & 63) == 0;  // one cell in 64: atomics on one address from every cell throttle the kernel they measure
#define FAST_TICK(i) do { tk1 = wall_clock64(); if (statCell) atomicAdd(&g_fastStats[i], tk1 - tk0); tk0 = tk1; } while (0)
#else
#define FAST_TICK(i) do { } while (0)
#endif
    uint32_t* tile = lds;                                           // [tileRows][TSD] dwords (+ slack)
    // score map: the detection area and a one-pixel apron of zeros, rows smapPitch bytes apart -- a map with the tile's own
    // geometry was 0.9 KB more of the 6.3 KB that decide how many cells a CU holds
    uint8_t* smap = (uint8_t*)(lds + tileRows * TSD + 4);
    uint16_t* list = (uint16_t*)(smap + (((tileRows - 4) * smapPitch + 7) & ~7));   // listCap entries
    uint8_t* const smapD = smap + smapPitch + 1;                    // S of detection pixel (x, y) at smapD[y * smapPitch + x]
    const uint8_t* tb = (const uint8_t*)tile;

    const int cw = c.w, ch = c.h;
    const int ndq = (cw + 8) >> 3;  // 8-byte columns covering the cw + 1 bytes from x0 - 1 (<= TSB / 8)
    // coalesced 8-byte loads of the ROI rows starting one byte left of the ROI (unaligned global loads are
    // fine on gfx950), 6 per lane in flight; clamped addresses so that out-of-range lanes re-store a valid
    // qword and no store needs a predicate
    {
        struct __attribute__((packed)) U64 { uint2 v; };
        const uint8_t* rowbase = base + (int64_t)c.y0 * stride + c.x0 - 1;
        for (int d0 = 0; d0 < ndq; d0 += 8) {
            const int qc = min(d0 + (lane & 7), ndq - 1);
            for (int rb = 0; rb < ch; rb += 48) {
                uint2 regs[6];
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    const int r = min(rb + (lane >> 3) + 8 * k, ch - 1);
                    regs[k] = ((const U64*)(rowbase + (int64_t)r * stride + 8 * qc))->v;
                }
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    const int r = min(rb + (lane >> 3) + 8 * k, ch - 1);
                    *(uint2*)&tile[r * TSD + 2 * qc] = regs[k];
                }
            }
        }
        uint2* sm64 = (uint2*)smap;
        for (int i = lane; i < ((ch - 4) * smapPitch + 7) >> 3; i += 64) sm64[i] = make_uint2(0u, 0u);
    }
    __syncthreads();
    FAST_TICK(9);

    const int dw = cw - 6, dh = ch - 6;  // detection area
    const int th1 = g->iniTh < 0 ? 0 : (g->iniTh > 255 ? 255 : g->iniTh);
    const int th2 = g->minTh < 0 ? 0 : (g->minTh > 255 ? 255 : g->minTh);
    constexpr int pos0 = 3 * TSB + 4;  // tile byte offset of detection pixel (0,0)
    // every cell owns a fixed segment of the candidate buffer (no atomics, deterministic layout)
    int32_t* myCount = cellCount + (int64_t)f * g->totalCells + bx;
    uint64_t* out = cand + (int64_t)f * g->candFrameRecs + L.candOff + c.candOff;
    const int candCap = ((cw - 6 + 1) >> 1) * ((ch - 6 + 1) >> 1);  // NMS bound = segment size
    // stage-1 rows of a half-wave: even rows in lanes 0..31, odd rows in lanes 32..63 -- with the 12-dword row pitch the
    // four 8-dword row segments a 32-lane group reads then fall on 32 distinct banks (rows r and r+3 of the plain
    // lane >> 3 mapping share four)
    const int rowInIter = 2 * ((lane >> 3) & 3) + (lane >> 5);

    // The reference runs cv::FAST at iniThFAST and, only if that leaves the cell empty, again at minThFAST (:808-816).
    // Same here: a pass at threshold t needs the exact score S only where S > t (a pixel with S <= t is neither a
    // corner nor a neighbour that could suppress one: a corner's S exceeds t), so the compass pre-test, the score and
    // the corner list all work at t -- at t = 20 a third of the pixels that a combined pass at min(20, 7) would score.
    // Cells without a corner at iniThFAST pay a second pass; on textured frames they are the minority.
    int total = 0;
#pragma nounroll
    for (int pass = 0; pass < 2; pass++) {
        const int th = pass ? th2 : th1;
        const bool last = pass == 1 || th1 == th2;
        if (pass) {  // the first pass's scores (all <= ... > th1) go: the map must hold zeros wherever S <= th
            __syncthreads();
            uint2* sm64 = (uint2*)smap;
            for (int i = lane; i < ((ch - 4) * smapPitch + 7) >> 3; i += 64) sm64[i] = make_uint2(0u, 0u);
            __syncthreads();
        }
        // stage 1: two lists of (y << 7 | x) entries -- the pixels that pass the compass test on the dark side grow one up
        // from the bottom of the buffer, those that pass on the bright side one down from its top (the few that pass on both
        // are in both: a pixel is a corner on one side at most, so its two visits never both write).  Ballots are taken of
        // bare compares and combined on the scalar unit (a ballot of a derived bool costs two VALU ops).  Lists that
        // meet in the middle overwrite each other -- inside the buffer -- and are thrown away below.
        int nD = 0, nBt = 0;
        for (int xb = 0; xb < dw; xb += 32) {
            const int x4 = xb + 4 * (lane & 7);
            uint64_t mX[4];
#pragma unroll
            for (int k = 0; k < 4; k++) mX[k] = ballot64(x4 + k < dw);
            for (int y0 = 0; y0 < dh; y0 += 8) {
                const int y = y0 + rowInIter;
                const bool rowOk = y < dh;
                const uint64_t mY = ballot64(y < dh);
                const uint32_t* q = tile + (rowOk ? y : 0) * TSD + (x4 >> 2);
                const uint32_t N = q[1], C0 = q[3 * TSD], C1 = q[3 * TSD + 1], C2 = q[3 * TSD + 2], S = q[6 * TSD + 1];
                // two pixels per instruction: the five dwords widened to 16-bit pairs (ten v_perm), the compass min / max on
                // v_pk_min_u16 / v_pk_max_u16 (twelve), the thresholds added in packed form (four): 26 + 8 compares for four
                // pixels against 40
                typedef unsigned short us2 __attribute__((ext_vector_type(2)));
                auto wid = [](uint32_t hiSrc, uint32_t loSrc, uint32_t sel) { return __builtin_bit_cast(us2, __builtin_amdgcn_perm(hiSrc, loSrc, sel)); };
                const us2 n2[2] = {wid(0, N, 0x0C010C00u), wid(0, N, 0x0C030C02u)}, s2[2] = {wid(0, S, 0x0C010C00u), wid(0, S, 0x0C030C02u)};
                const us2 v2[2] = {wid(0, C1, 0x0C010C00u), wid(0, C1, 0x0C030C02u)};
                const us2 w2[2] = {wid(0, C0, 0x0C020C01u), wid(C1, C0, 0x0C040C03u)};   // x - 3: C0.b1 C0.b2 | C0.b3 C1.b0
                const us2 e2[2] = {wid(C2, C1, 0x0C040C03u), wid(0, C2, 0x0C020C01u)};   // x + 3: C1.b3 C2.b0 | C2.b1 C2.b2
                const us2 th2 = {(unsigned short)th, (unsigned short)th};
                us2 hiP[2], loP[2], vtP[2];
#pragma unroll
                for (int m = 0; m < 2; m++) {
                    hiP[m] = __builtin_elementwise_min(__builtin_elementwise_max(n2[m], s2[m]), __builtin_elementwise_max(w2[m], e2[m]));
                    loP[m] = __builtin_elementwise_max(__builtin_elementwise_min(n2[m], s2[m]), __builtin_elementwise_min(w2[m], e2[m])) + th2;
                    vtP[m] = v2[m] + th2;
                }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const unsigned short hk = k & 1 ? hiP[k >> 1].y : hiP[k >> 1].x, lk = k & 1 ? loP[k >> 1].y : loP[k >> 1].x;
                    const unsigned short vtk = k & 1 ? vtP[k >> 1].y : vtP[k >> 1].x, vk = k & 1 ? v2[k >> 1].y : v2[k >> 1].x;
                    const bool br = hk > vtk, dk = vk > lk;   // hi - v > th, v - lo > th
                    const uint64_t mIn = mX[k] & mY, mBr = ballot64(br) & mIn, mDk = ballot64(dk) & mIn;
                    const bool in = rowOk && x4 + k < dw;
                    const int e = (y << 7) | (x4 + k);
                    if (mDk) {
                        if (in && dk) list[nD + lanes_below(mDk)] = (uint16_t)e;
                        nD += __popcll(mDk);
                    }
                    if (mBr) {
                        if (in && br) list[listCap - 1 - nBt - lanes_below(mBr)] = (uint16_t)e;
                        nBt += __popcll(mBr);
                    }
                }
            }
        }
        __syncthreads();
        FAST_TICK(10);
#ifdef ORBX_FAST_STATS
        if (statCell) { atomicAdd(&g_fastStats[0 + 4 * pass], 1ull); atomicAdd(&g_fastStats[1 + 4 * pass], (unsigned long long)(nD + nBt)); atomicAdd(&g_fastStats[8], (unsigned long long)(dw * dh)); }
#endif
        if (nD + nBt == 0) { if (last) break; else continue; }

        // stage 2: exact score on the dense lists, each with the arithmetic of its side; pixels with S > th (the corners of
        // this pass) are compacted in place: writes land at or below entries already consumed (the bright list is read in
        // ascending addresses and starts at listCap - nBt >= nD)
        int nC = 0;
        if (nD + nBt <= listCap) {
            for (int i0 = 0; i0 < nD; i0 += 64) {
                const int i = i0 + lane;
                const bool act = i < nD;
                const int e = act ? list[i] : 0;
                const int pos = pos0 + (e >> 7) * TSB + (e & 0x7F);
                const int Sx = fast_S_dark<TSB>(tb + pos);
                const bool corner = act && Sx > th;
                const uint64_t bal = ballot64(Sx > th) & tail_mask(nD - i0);
                if (corner) { smapD[__mul24(e >> 7, smapPitch) + (e & 0x7F)] = (uint8_t)Sx; list[nC + lanes_below(bal)] = (uint16_t)e; }
                nC += __popcll(bal);
            }
            for (int i0 = 0; i0 < nBt; i0 += 64) {
                const int i = i0 + lane;
                const bool act = i < nBt;
                const int e = act ? list[listCap - nBt + i] : 0;
                const int pos = pos0 + (e >> 7) * TSB + (e & 0x7F);
                const int Sx = fast_S_bright<TSB>(tb + pos);
                const bool corner = act && Sx > th;
                const uint64_t bal = ballot64(Sx > th) & tail_mask(nBt - i0);
                if (corner) { smapD[__mul24(e >> 7, smapPitch) + (e & 0x7F)] = (uint8_t)Sx; list[nC + lanes_below(bal)] = (uint16_t)e; }
                nC += __popcll(bal);
            }
        } else {
            // the lists ran into each other (a texture of saddle points: nearly every pixel passes on both sides): every
            // detection pixel is scored on both sides instead -- the compass test is pruning, S > th decides
            const int npx = dw * dh;
            for (int i0 = 0; i0 < npx; i0 += 64) {
                const int i = i0 + lane;
                const bool act = i < npx;
                const int y = act ? i / dw : 0, x = act ? i - y * dw : 0;
                const int e = (y << 7) | x;
                const int pos = pos0 + y * TSB + x;
                const int Sx = fast_S<TSB>(tb + pos);
                const bool corner = act && Sx > th;
                const uint64_t bal = ballot64(Sx > th) & tail_mask(npx - i0);
                if (corner) { smapD[__mul24(e >> 7, smapPitch) + (e & 0x7F)] = (uint8_t)Sx; list[nC + lanes_below(bal)] = (uint16_t)e; }
                nC += __popcll(bal);
            }
        }
        __syncthreads();
        FAST_TICK(11);
#ifdef ORBX_FAST_STATS
        if (statCell) atomicAdd(&g_fastStats[2 + 4 * pass], (unsigned long long)nC);
#endif
        if (nC == 0) { if (last) break; else continue; }

        // stage 3: 3x3 non-max suppression on M_t = (S > t ? S-1 : 0), strict >, zero outside the
        // detection area (the S map is zero there), fused with the emission.  A corner has S > t, so
        // "S > every neighbour with S > t" is simply S > max of the eight neighbours.
        total = 0;
        for (int i0 = 0; i0 < nC; i0 += 64) {
            const int i = i0 + lane;
            const int e = i < nC ? list[i] : 0;
            const uint8_t* sp = smapD + __mul24(e >> 7, smapPitch) + (e & 0x7F);
            const uint8_t* up = sp - smapPitch;
            const uint8_t* dn = sp + smapPitch;
            const int sc = sp[0];
            const int n0 = up[-1], n1 = up[0], n2 = up[1], n3 = sp[-1], n4 = sp[1], n5 = dn[-1], n6 = dn[0], n7 = dn[1];
            const int m = max3i(max3i(n0, n1, n2), max3i(n3, n4, n5), max(n6, n7));
            const bool keep = i < nC && sc > th && sc >= 2 && sc > m;
            const uint64_t bal = ballot64(sc > m) & ballot64(sc > max(th, 1)) & tail_mask(nC - i0);
            if (keep) {
                const int p = total + lanes_below(bal);
                if (p < candCap) {
                    const uint32_t xr = (e & 0x7F) + 3, yr = (e >> 7) + 3;  // ROI coordinates
                    out[p] = pack_cand(c.x0 - kMinBorder + xr, c.y0 - kMinBorder + yr, sc - 1, cand_order(c.seq, yr, xr));
                } else {
                    atomicOr(errFlag, 1);
                }
            }
            total += __popcll(bal);
        }
        FAST_TICK(12);
        if (total > 0 || last) break;
    }
    if (lane == 0) *myCount = total;
}

// ------------------------------------------------------------------ quadtree distribution
// One 256-thread workgroup per (frame, level) replays DistributeOctTree exactly:
// the node list lives in LDS in list order; every round is data-parallel
// (wave-per-node 4-way partition with ballots, block scans for the new list order).
// Node keys are ranges [start, start+cnt) of a ping-pong record buffer; children
// partition their parent's range in the other buffer, so ranges never overlap.
constexpr int kDistThreads = 512;

__device__ __forceinline__ uint32_t block_excl_scan(uint32_t* a, int m, uint32_t* wtmp)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (m + kDistThreads - 1) / kDistThreads;
    const int b = min(tid * per, m), e = min(b + per, m);
    uint32_t s = 0;
    for (int i = b; i < e; i++) s += a[i];
    const uint32_t incl = (uint32_t)wave_incl_scan((int)s);
    __syncthreads();
    if (lane == 63) wtmp[wave] = incl;
    __syncthreads();
    uint32_t woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kDistThreads / 64; w++) {
        const uint32_t t = wtmp[w];
        if (w < wave) woff += t;
        total += t;
    }
    uint32_t run = woff + incl - s;
    for (int i = b; i < e; i++) { const uint32_t t = a[i]; a[i] = run; run += t; }
    __syncthreads();
    return total;
}

// 4-way partition of one node's keys by the wave; returns child counts (uniform)
// 4 keys per lane per iteration so that four global loads are in flight per lane
// (the partition is latency-bound: few waves, dependent round trips)
__device__ __forceinline__ void wave_divide(const uint64_t* __restrict__ srcb, uint64_t* __restrict__ dstb,
                                            uint32_t start, uint32_t cnt, int xm, int ym, uint32_t c[4])
{
    constexpr int U = 4;
    const int lane = threadIdx.x & 63;
    const uint64_t lt = lanemask_lt();
    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    for (uint32_t p0 = 0; p0 < cnt; p0 += 64 * U) {
        uint64_t key[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t p = p0 + 64 * u + lane;
            key[u] = p < cnt ? srcb[start + p] : ~0ull;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t p = p0 + 64 * u + lane;
            const int q = p < cnt ? (((int)cand_x(key[u]) >= xm ? 1 : 0) | ((int)cand_y(key[u]) >= ym ? 2 : 0)) : 4;
            c0 += __popcll(__ballot(q == 0));
            c1 += __popcll(__ballot(q == 1));
            c2 += __popcll(__ballot(q == 2));
            c3 += __popcll(__ballot(q == 3));
        }
    }
    uint32_t r0 = start, r1 = start + c0, r2 = start + c0 + c1, r3 = start + c0 + c1 + c2;
    for (uint32_t p0 = 0; p0 < cnt; p0 += 64 * U) {
        uint64_t key[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t p = p0 + 64 * u + lane;
            key[u] = p < cnt ? srcb[start + p] : ~0ull;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t p = p0 + 64 * u + lane;
            const int q = p < cnt ? (((int)cand_x(key[u]) >= xm ? 1 : 0) | ((int)cand_y(key[u]) >= ym ? 2 : 0)) : 4;
            const uint64_t b0 = __ballot(q == 0), b1 = __ballot(q == 1), b2 = __ballot(q == 2), b3 = __ballot(q == 3);
            if (q == 0) dstb[r0 + __popcll(b0 & lt)] = key[u];
            else if (q == 1) dstb[r1 + __popcll(b1 & lt)] = key[u];
            else if (q == 2) dstb[r2 + __popcll(b2 & lt)] = key[u];
            else if (q == 3) dstb[r3 + __popcll(b3 & lt)] = key[u];
            r0 += __popcll(b0); r1 += __popcll(b1); r2 += __popcll(b2); r3 += __popcll(b3);
        }
    }
    c[0] = c0; c[1] = c1; c[2] = c2; c[3] = c3;
}

// LDS = true: node list and scratch live in LDS (every shipped configuration);
// LDS = false: same code over a per-block global scratch region, for very large per-level targets.
struct DistArgs {
    const Geom* g; const uint64_t* candRaw; uint64_t* candA; uint64_t* candB; const Cell* cells; const int32_t* cellCount;
    int32_t* candCount; uint64_t* kept; int32_t* keptCount; int32_t* errFlag; int cap, f0; uint32_t* gscratch; int scratchWords, l0;
    int xcdFrames;   // >= 8: grid (levels, xcd_grid_y(frames)), a frame's levels on the XCD that ran its FAST cells and will run its descriptors
};

// the body of k_distribute for block (bxLevel, by) of a (levels, frames) grid
template <bool LDS>
__device__ __forceinline__ void distribute_body(const DistArgs& da, const int bxLevel, const int by)
{
    const Geom* __restrict__ g = da.g;
    const uint64_t* __restrict__ candRaw = da.candRaw;
    uint64_t* __restrict__ candA = da.candA; uint64_t* __restrict__ candB = da.candB;
    const Cell* __restrict__ cells = da.cells;
    const int32_t* __restrict__ cellCount = da.cellCount;
    int32_t* __restrict__ candCount = da.candCount;
    uint64_t* __restrict__ kept = da.kept; int32_t* __restrict__ keptCount = da.keptCount;
    int32_t* __restrict__ errFlag = da.errFlag;
    const int cap = da.cap, f0 = da.f0, scratchWords = da.scratchWords, l0 = da.l0;
    uint32_t* __restrict__ gscratch = da.gscratch;
#ifdef ORBX_DIST_TIMING  // phase timestamps of the level-0 block of frame 0, printed at the end (tools/dist_timing.py)
    __shared__ uint64_t sStamp[96]; __shared__ uint64_t sCyc[96]; __shared__ int sStampId[96]; __shared__ int sNStamp;
    if (threadIdx.x == 0) sNStamp = 0;
#define STAMP(id) do { if (threadIdx.x == 0 && sNStamp < 96) { sStampId[sNStamp] = (id); sCyc[sNStamp] = clock64(); sStamp[sNStamp++] = wall_clock64(); } } while (0)
#else
#define STAMP(id) do {} while (0)
#endif
    extern __shared__ __attribute__((aligned(16))) uint32_t smem_lds[];
    // the launch covers levels [l0, l0 + gridDim.x): level 0 may go ahead of the others (it needs no pyramid)
    const int l = bxLevel + l0, f = by + f0;
    uint32_t* const smem = LDS ? smem_lds : gscratch + (int64_t)(by * g->nlevels + l) * scratchWords;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = kDistThreads / 64;
    const LevelGeom& L = g->lv[l];
    const int N = L.nFeat;
    // candidates of this level = the per-cell segments FAST filled; prefix of the cell counts
    // in LDS turns a flat index into (cell, slot) by binary search
    const int nCells = L.nCells;
    uint32_t* cellPref = smem + 19 * cap + 8;            // [maxCells + 1]
    uint32_t* cellOffs = cellPref + g->maxCellsPerLevel + 1;  // [maxCells]
    uint32_t* wtmp0 = smem + 19 * cap;
    for (int i = tid; i < nCells; i += kDistThreads) {
        cellPref[i] = (uint32_t)cellCount[(int64_t)f * g->totalCells + L.cellBase + i];
        cellOffs[i] = cells[L.cellBase + i].candOff;
    }
    __syncthreads();
    STAMP(0);
    int n = (int)block_excl_scan(cellPref, nCells, wtmp0);
    if (tid == 0) { cellPref[nCells] = (uint32_t)n; candCount[f * g->nlevels + l] = n; }
    __syncthreads();
    if (n > L.candCap) n = L.candCap;
    uint64_t* bufs[2] = {candA + (int64_t)f * g->candFrameRecs + L.candOff,
                         candB + (int64_t)f * g->candFrameRecs + L.candOff};

    // LDS carve-up (all u32 arrays of `cap` entries unless noted)
    // the two generations of the node list are addressed as base[set * cap + i]: a pointer picked at run time out of
    // a two-entry array makes the accesses generic (flat) instead of LDS
    short4* const nbB = (short4*)smem;       // bounds x0,x1,y0,y1   [2][cap]
    uint32_t* const nsB = smem + 4 * cap;    // key range start      [2][cap]
    uint32_t* const ncB = smem + 6 * cap;    // key count | bufId<<31 [2][cap]
    uint32_t* cc = smem + 8 * cap;        // [cap*4] child counts per E-entry
    uint32_t* ord = cc + 4 * cap;         // processing order -> list index
    uint32_t* ord2 = ord + cap;
    uint32_t* tA = ord2 + cap;            // scan scratch
    uint32_t* tB = tA + cap;
    uint32_t* proc = tB + cap;            // processed flag per list index
    uint32_t* const nkB = proc + cap;     // depth << 24 | path code of the node [2][cap]
    uint32_t* wtmp = proc + 3 * cap;      // [8]
    __shared__ int sJ;

    STAMP(1);
    // ---- roots (:543-585) and the first D levels below them in one counting sort.
    // A key's way down the tree is a function of its coordinates alone: root (int)(x / hX), then at every
    // level the quadrant against the node's midlines (:483-484).  So every key gets its path code to depth D
    // (nIni * 4^D <= 1024 bins), one histogram + one scatter put the keys in path order, and the sums of
    // the histogram over 4, 16, ... bins are the child counts of every node of depth < D: the first D
    // rounds below replay the list logic on those counts and move no key at all.
    __shared__ uint32_t hst[1368];   // levels 0..D of the histogram, level d at hoff(d) = nIni * (4^d - 1) / 3
    // The scratch of the counting sort -- scatter cursors of the leaf bins, the two code tables, the search table -- lives
    // in the part of the node-list carve-up that only the rounds use (cc .. proc, 9 * cap words; the host keeps
    // cap >= 360 for that): the workgroup's LDS footprint decides how soon it is placed beside the other kernels, and
    // the pipeline loses 2.2 % per 16 KB of it (section 5 of DESIGN.md).
    uint32_t* const hfill = smem + 8 * cap;   // [1024]
    const int nIni = L.nIni;
    const float hX = L.hX;
    if (nIni > 1024) { if (tid == 0) atomicOr(errFlag, 2); return; }
    int D = 0;
    while (D < 5 && (nIni << (2 * (D + 1))) <= 1024) D++;
    auto hoff = [&](int d) { return nIni * (((1 << (2 * d)) - 1) / 3); };
    const int nBins = nIni << (2 * D), hTotal = hoff(D) + nBins;
    for (int i = tid; i < hTotal; i += kDistThreads) hst[i] = 0;
    __syncthreads();
    const uint64_t* srcb = candRaw + (int64_t)f * g->candFrameRecs + L.candOff;  // raw FAST output stays untouched
    // The code separates: code = cx(x) + cy(y), cx = root * 4^D + the x decisions at bit 0 of every base-4 digit,
    // cy = the y decisions at bit 1 (the y descent is the same under every root).  Two small LDS tables built per
    // block (one descent per column and per row instead of one per key) turn a key's code into two byte-pair reads.
    constexpr int kLutX = 2048, kLutY = 1280;
    constexpr int kFirstCap = 1024;
    // the two code tables and the search table of the counting sort are dead once the keys are scattered -- the list
    // construction below reuses their words for its scan
    uint32_t* const sortScratch = hfill + 1024;   // [(kLutX + kLutY + kFirstCap + 2) / 2]
    static_assert((kLutX + kLutY + kFirstCap + 2) / 2 >= 1368, "the list construction scans up to 1368 flags here");
    static_assert(1024 + (kLutX + kLutY + kFirstCap + 2) / 2 <= 9 * 360, "host: nodeCap >= 360");
    uint16_t* const lutx = (uint16_t*)sortScratch;
    uint16_t* const luty = lutx + kLutX;
    uint16_t* const firstCell = luty + kLutY;
    const bool useLut = L.winW <= kLutX && L.winH <= kLutY;
    auto code_x = [&](int x) {
        int r = (int)__fdiv_rn((float)x, hX);
        if (r >= nIni) r = nIni - 1;
        int x0 = (short)(int)__fmul_rn(hX, (float)r), x1 = (short)(int)__fmul_rn(hX, (float)(r + 1));
        int code = r;
        for (int d = 0; d < D; d++) {
            const int xm = x0 + ((x1 - x0 + 1) >> 1);
            const int qx = x >= xm;
            code = code * 4 + qx;
            if (qx) x0 = xm; else x1 = xm;
        }
        return code;
    };
    auto code_y = [&](int y) {
        int y0 = 0, y1 = (short)L.winH, code = 0;
        for (int d = 0; d < D; d++) {
            const int ym = y0 + ((y1 - y0 + 1) >> 1);
            const int qy = y >= ym;
            code = code * 4 + 2 * qy;
            if (qy) y0 = ym; else y1 = ym;
        }
        return code;
    };
    STAMP(30);
    if (useLut) {
        for (int i = tid; i < L.winW; i += kDistThreads) lutx[i] = (uint16_t)code_x(i);
        for (int i = tid; i < L.winH; i += kDistThreads) luty[i] = (uint16_t)code_y(i);
        __syncthreads();
    }
    auto path_code = [&](uint64_t key) {
        const int x = (int)cand_x(key), y = (int)cand_y(key);
        return useLut ? (int)lutx[min(x, kLutX - 1)] + (int)luty[min(y, kLutY - 1)] : code_x(x) + code_y(y);
    };
    // flat index -> (cell, slot) by binary search in the cell prefix.  A chunk is 16 keys per thread, all loads
    // in flight together; a level that fits one chunk (<= 8192 candidates, every shipped shape) keeps its keys
    // in registers between the count and the scatter, larger ones read them twice.
    STAMP(31);
    constexpr int KPT = 16;
    // flat index -> (cell, slot).  A thread takes its KPT positions as four runs of four CONSECUTIVE ones (run r of thread t:
    // base + r * 4 T + 4 t ..): one table lookup and a short walk over the cell boundaries per run -- it used to be a
    // binary search for each of 16 strided positions, and the chunk load was instruction bound (5.5 of this block's 35 us) --
    // while a wave's lanes still read 2 KB in one piece (16 consecutive positions per thread: every lane on a cache line of
    // its own, the loads then wait for the texture unit's line rate).  firstCell[k] = cell of flat position k << fshift,
    // written by the cells themselves (a cell marks the blocks that start inside it: no search, no dependent chain).
    int fshift = 4;
    while (fshift < 6 && ((n + (1 << fshift) - 1) >> fshift) > kFirstCap) fshift++;
    const int nblk = (n + (1 << fshift) - 1) >> fshift;
    const bool twoLevel = nblk <= kFirstCap && nCells <= 65535;
    int topStep = 1;
    while (topStep * 2 < nCells) topStep *= 2;
    if (twoLevel) {
        const uint32_t fm = (1u << fshift) - 1;
        for (int c = tid; c < nCells; c += kDistThreads) {
            const uint32_t a = cellPref[c], b = cellPref[c + 1];
            for (uint32_t k = (a + fm) >> fshift; (k << fshift) < b; k++) firstCell[k] = (uint16_t)c;
        }
        __syncthreads();
    }
    STAMP(32);
    constexpr int kRun = 4;
    auto chunk_pos = [&](int base, int u) { return base + (u / kRun) * (kRun * kDistThreads) + tid * kRun + (u % kRun); };
    auto load_chunk = [&](int base, uint64_t (&key)[KPT], uint32_t (&code)[KPT]) {
#pragma unroll
        for (int r = 0; r < KPT / kRun; r++) {
            const int ps = min(chunk_pos(base, r * kRun), n - 1);
            int lo;
            if (twoLevel) {
                lo = (int)firstCell[ps >> fshift];
            } else {  // largest c with cellPref[c] <= ps
                lo = 0;
                for (int step = topStep; step > 0; step >>= 1) {
                    const int c = lo + step;
                    if (c < nCells && cellPref[c] <= (uint32_t)ps) lo = c;
                }
            }
            uint32_t cnext = cellPref[lo + 1];
            while ((uint32_t)ps >= cnext) { lo++; cnext = cellPref[lo + 1]; }   // (ps < n = cellPref[nCells]: ends inside the table)
            uint32_t cstart = cellPref[lo], off = cellOffs[lo];
#pragma unroll
            for (int j = 0; j < kRun; j++) {
                const uint32_t pp = (uint32_t)min(chunk_pos(base, r * kRun + j), n - 1);
                while (pp >= cnext) { lo++; cstart = cnext; cnext = cellPref[lo + 1]; off = cellOffs[lo]; }
                key[r * kRun + j] = srcb[off + (pp - cstart)];
            }
        }
#pragma unroll
        for (int u = 0; u < KPT; u++) code[u] = (uint32_t)path_code(key[u]);
    };
    auto count_chunk = [&](int base, const uint32_t (&code)[KPT]) {
#pragma unroll
        for (int u = 0; u < KPT; u++)
            if (chunk_pos(base, u) < n) atomicAdd(&hst[hoff(D) + code[u]], 1u);
    };
    auto scatter_chunk = [&](int base, const uint64_t (&key)[KPT], const uint32_t (&code)[KPT]) {
#pragma unroll
        for (int u = 0; u < KPT; u++)
            if (chunk_pos(base, u) < n) bufs[1][atomicAdd(&hfill[code[u]], 1u)] = key[u];
    };
    auto tree_sums_and_starts = [&]() {
        __syncthreads();
        STAMP(3);
        for (int d = D - 1; d >= 0; d--) {
            const int cnt = nIni << (2 * d);
            for (int i = tid; i < cnt; i += kDistThreads) {
                const uint32_t* c4 = &hst[hoff(d + 1) + 4 * i];
                hst[hoff(d) + i] = c4[0] + c4[1] + c4[2] + c4[3];
            }
            __syncthreads();
        }
        for (int i = tid; i < nBins; i += kDistThreads) hfill[i] = hst[hoff(D) + i];
        __syncthreads();
        block_excl_scan(hfill, nBins, wtmp);
        STAMP(4);
    };
    constexpr int kChunk = KPT * kDistThreads;
    if (n <= kChunk) {
        uint64_t key[KPT];
        uint32_t code[KPT];
        if (n > 0) { load_chunk(0, key, code); STAMP(33); count_chunk(0, code); }
        tree_sums_and_starts();
        if (n > 0) scatter_chunk(0, key, code);
        STAMP(34);
    } else {
        for (int base = 0; base < n; base += kChunk) { uint64_t key[KPT]; uint32_t code[KPT]; load_chunk(base, key, code); count_chunk(base, code); }
        tree_sums_and_starts();
        for (int base = 0; base < n; base += kChunk) { uint64_t key[KPT]; uint32_t code[KPT]; load_chunk(base, key, code); scatter_chunk(base, key, code); }
    }
    // ---- the list after the first r rounds, built in one step.  As long as every node with more than one key is
    // divided (main mode) the list is a function of the histogram alone: a bin of depth d is a node iff it holds a key
    // and its parent held more than one; the list after round r is C_r ++ leaves(C_{r-1}) ++ ... ++ leaves(C_0), where
    // C_d are the depth-d nodes and leaves(.) those left with one key (never divided again, :594-600), and the order
    // inside C_d follows from push_front: groups in reverse processing order, n4..n1 inside a group (:621-660), i.e.
    // ascending in T_d = (-T_{d-1}, -q_d), T_0 = root index -- the path code with every other base-4 digit
    // complemented.  The loop conditions (:669-673) after each of those rounds need only the three tallies per depth.
    // So: tally, find the first round r* after which the reference stops or turns careful (or the histogram ends),
    // one scan over the concatenated (depth, complemented code) sequence = every node's list position.  Replaces r*
    // rounds of ~5 us (a dozen barriers each) by one of ~3.
    STAMP(35);
    __shared__ int sTal[6][3];         // per depth: nodes, nodes with one key, nodes with more
    uint32_t* const sq = sortScratch;
    if (tid < 18) (&sTal[0][0])[tid] = 0;
    __syncthreads();
    auto depth_of = [&](int idx, int& code) { int d = 0; while (d < D && idx >= hoff(d + 1)) d++; code = idx - hoff(d); return d; };
    auto is_node = [&](int d, int code, uint32_t cnt) { return cnt >= 1 && (d == 0 || hst[hoff(d - 1) + (code >> 2)] > 1); };
    // a thread's (at most three) histogram entries: depth, code, count and "is a node" are worked out once, here, and kept
    // for the two passes below (each used to redo the depth search and the parent lookup)
    constexpr int kIdxPer = (1368 + kDistThreads - 1) / kDistThreads;
    int eD[kIdxPer], eCode[kIdxPer]; uint32_t eCnt[kIdxPer]; bool eNode[kIdxPer];
#pragma unroll
    for (int k = 0; k < kIdxPer; k++) {
        const int idx = tid + k * kDistThreads;
        eD[k] = 0; eCode[k] = 0; eCnt[k] = 0; eNode[k] = false;
        if (idx < hTotal) {
            eD[k] = depth_of(idx, eCode[k]);
            eCnt[k] = hst[idx];
            eNode[k] = is_node(eD[k], eCode[k], eCnt[k]);
            if (eNode[k]) { atomicAdd(&sTal[eD[k]][0], 1); atomicAdd(&sTal[eD[k]][eCnt[k] == 1 ? 1 : 2], 1); }
        }
    }
    __syncthreads();
    int rstar = 0;
    bool careful = false, finished = false;
    if (sTal[0][2] > 0) {  // the first round always runs in main mode
        int mPrev = sTal[0][0], leaves = 0;
        for (int r = 1; r <= D; r++) {
            leaves += sTal[r - 1][1];
            const int mr = sTal[r][0] + leaves, nx = sTal[r][2];
            rstar = r;
            if (mr >= N || mr == mPrev) { finished = true; break; }   // :669 / :733
            if (mr + 3 * nx > N) { careful = true; break; }          // :673
            if (nx == 0) break;                                       // nothing left to divide: the loop below ends at once
            mPrev = mr;
        }
    }
    // sequence: depth r* (all its nodes), then depths r*-1 .. 0 (their one-key nodes), each in T order
    auto seq_index = [&](int d, int code) {
        int base = 0;
        for (int j = rstar; j > d; j--) base += nIni << (2 * j);
        const int low = (1 << (2 * d)) - 1;
        const int root = code >> (2 * d);
        return base + ((((d & 1) ? nIni - 1 - root : root) << (2 * d)) | ((code & low) ^ (0x333 & low)));
    };
    const int seqLen = hoff(rstar) + (nIni << (2 * rstar));
    // (d, code) -> seq_index is one-to-one onto [0, seqLen) for the entries of depth <= r*: every slot is written, no clearing pass
    int eSeq[kIdxPer];
#pragma unroll
    for (int k = 0; k < kIdxPer; k++) {
        const int idx = tid + k * kDistThreads;
        eSeq[k] = 0;
        if (idx < seqLen) {
            eSeq[k] = seq_index(eD[k], eCode[k]);
            eNode[k] = eNode[k] && (eD[k] == rstar || eCnt[k] == 1);   // from here on: "is in the list"
            sq[eSeq[k]] = eNode[k] ? 1u : 0u;
        } else {
            eNode[k] = false;
        }
    }
    __syncthreads();
    int m = (int)block_excl_scan(sq, seqLen, wtmp);
    int cur = 0;
    if (m > cap) { if (tid == 0) atomicOr(errFlag, 2); return; }  // cannot happen (size <= max(N + 2, 4 * nIni)); guard anyway
#pragma unroll
    for (int k = 0; k < kIdxPer; k++) {
        if (!eNode[k]) continue;
        const int d = eD[k], code = eCode[k];
        const uint32_t cnt = eCnt[k];
        const int pos = (int)sq[eSeq[k]];
        const int root = code >> (2 * d);
        int x0 = (short)(int)__fmul_rn(hX, (float)root), x1 = (short)(int)__fmul_rn(hX, (float)(root + 1)), y0 = 0, y1 = (short)L.winH;
        for (int t = 1; t <= d; t++) {
            const int q = (code >> (2 * (d - t))) & 3;
            const int xm = x0 + ((x1 - x0 + 1) >> 1), ym = y0 + ((y1 - y0 + 1) >> 1);  // :483-484
            if (q & 1) x0 = xm; else x1 = xm;
            if (q & 2) y0 = ym; else y1 = ym;
        }
        const int b0 = code << (2 * (D - d));  // the node's first leaf bin; hfill holds bin ENDS after the scatter
        nbB[cur * cap + pos] = make_short4((short)x0, (short)x1, (short)y0, (short)y1);
        nsB[cur * cap + pos] = hfill[b0] - hst[hoff(D) + b0];
        ncB[cur * cap + pos] = cnt | (1u << 31);  // keys are in buffer 1
        nkB[cur * cap + pos] = ((uint32_t)d << 24) | (uint32_t)code;
    }
    __syncthreads();

    STAMP(2);
    // ---- rounds (the ones the histogram could not decide)
    for (int iter = 0; iter < 64 && !finished; iter++) {
        const int prevSize = m;
        // E = nodes with more than one key, in list order
        for (int i = tid; i < m; i += kDistThreads) tA[i] = (ncB[cur * cap + i] & 0x7FFFFFFFu) > 1 ? 1u : 0u;
        __syncthreads();
        for (int i = tid; i < m; i += kDistThreads) tB[i] = tA[i];
        __syncthreads();
        const int E = (int)block_excl_scan(tB, m, wtmp);
        STAMP(10 + (careful ? 100 : 0));
        if (E == 0) break;  // size unchanged -> finish (:669 / :733)
        for (int i = tid; i < m; i += kDistThreads) if (tA[i]) ord[tB[i]] = i;
        __syncthreads();
        STAMP(20);
        if (careful) {
            // sort by (count desc, list position asc) == reference's (size, creation) ascending
            // sort walked from the back (:684-685, tie-break see DESIGN.md)
            // rank sort: rank(e) = number of keys above key(e), key = count << 13 | (8191 - list index), all
            // distinct.  One subtract-with-borrow + add-with-carry per pair, four keys per LDS read, padded to
            // a multiple of four with zeros (never above a real key); `parts` threads share one element's scan.
            // (Boolean logic through SGPR pairs instead stalls this two-waves-per-SIMD kernel on every
            // VALU -> SALU hand-over: measured 11 us against 1.)
            if (cap > 8192 || n >= (1 << 19)) {  // key does not pack: plain two-field rank sort
                for (int e = tid; e < E; e += kDistThreads) {
                    const uint32_t i = ord[e];
                    const uint32_t ci = ncB[cur * cap + i] & 0x7FFFFFFFu;
                    int rank = 0;
                    for (int e2 = 0; e2 < E; e2++) {
                        const uint32_t i2 = ord[e2];
                        const uint32_t c2 = ncB[cur * cap + i2] & 0x7FFFFFFFu;
                        rank += (c2 > ci || (c2 == ci && i2 < i)) ? 1 : 0;
                    }
                    ord2[rank] = i;
                }
                __syncthreads();
            } else {
            const int E4 = (E + 3) & ~3;  // <= cap (a multiple of 4)
            for (int e = tid; e < E4; e += kDistThreads) {
                const uint32_t i = e < E ? ord[e] : 0;
                tA[e] = e < E ? ((ncB[cur * cap + i] & 0x7FFFFFFFu) << 13) | (8191u - i) : 0u;
            }
            __syncthreads();
            STAMP(21);
            int parts = 1;
            while (parts < 8 && E * parts * 2 <= kDistThreads) parts *= 2;
            const int per = ((E + parts - 1) / parts + 3) & ~3;
            STAMP(23);
            for (int e0 = 0; e0 < E; e0 += kDistThreads / parts) {
                const int e = e0 + tid / parts, sub = tid % parts;
                uint32_t rank = 0;
                if (e < E) {
                    const uint32_t ke = tA[e];
                    // (all lanes of a part on the same uint4: a broadcast.)  Keys are below 2^31 (count < 2^18, << 13): "ke < k" is the
                    // sign of the difference, taken with a shift -- as compares the compiler routes every one through an SGPR pair
                    const int base4 = sub * per / 4, n4 = (min(E4, (sub + 1) * per) - sub * per) / 4;
#pragma unroll 4
                    for (int j = 0; j < n4; j++) {
                        const uint4 k4 = ((const uint4*)tA)[base4 + j];
                        rank += ((ke - k4.x) >> 31) + ((ke - k4.y) >> 31) + ((ke - k4.z) >> 31) + ((ke - k4.w) >> 31);
                    }
                }
                STAMP(24);
                for (int d = 1; d < parts; d <<= 1) rank += __shfl_xor(rank, d);
                STAMP(25);
                if (e < E && sub == 0) ord2[rank] = ord[e];
            }
            STAMP(26);
            __syncthreads();
            STAMP(22);
            }
            for (int e = tid; e < E; e += kDistThreads) ord[e] = ord2[e];
            __syncthreads();
        }
        // Divide the E nodes into the other buffer.  Main mode divides all of them.  The careful
        // phase stops at the first node after which the list holds N nodes (:727-728): a split
        // adds at most 3 nodes, so nodes are divided in sorted order in chunks of
        // ceil(deficit / 3) until the cut is found -- typically a quarter of E, not all of it.
        STAMP(11);
        if (tid == 0) sJ = E - 1;
        int done = 0;
        while (done < E) {
            int chunk = E - done;
            if (careful) {
                int grown = 0;  // growth of the prefix [0, done): recomputed from tA (uniform)
                if (done > 0) grown = (int)(tB[done - 1] + tA[done - 1]) - done;
                const int deficit = N - (m + grown);
                chunk = min(chunk, max(NW, (deficit + 2) / 3));
            }
            for (int t = done + wave; t < done + chunk; t += NW) {
                const uint32_t i = ord[t];
                const short4 b = nbB[cur * cap + i];
                const uint32_t cb = ncB[cur * cap + i];
                const uint32_t cnt = cb & 0x7FFFFFFFu, bid = cb >> 31;
                const int xm = b.x + ((b.y - b.x + 1) >> 1);  // UL.x + ceil((UR.x-UL.x)/2)  :483
                const int ym = b.z + ((b.w - b.z + 1) >> 1);  // UL.y + ceil((BR.y-UL.y)/2)  :484
                uint32_t c[4];
                const uint32_t kd = nkB[cur * cap + i];
                if ((int)(kd >> 24) < D) {  // child counts straight from the histogram, keys stay where they are
                    const uint32_t* c4 = &hst[hoff((int)(kd >> 24) + 1) + 4 * (kd & 0xFFFFFFu)];
                    c[0] = c4[0]; c[1] = c4[1]; c[2] = c4[2]; c[3] = c4[3];
                } else {
                    wave_divide(bufs[bid], bufs[bid ^ 1], nsB[cur * cap + i], cnt, xm, ym, c);
                }
                if (lane == 0) { cc[4 * t] = c[0]; cc[4 * t + 1] = c[1]; cc[4 * t + 2] = c[2]; cc[4 * t + 3] = c[3]; }
            }
            __syncthreads();
            done += chunk;
            // k_t = non-empty children; prefix over processing order for everything divided so far
            for (int t = tid; t < done; t += kDistThreads) {
                const uint32_t k = (cc[4 * t] ? 1 : 0) + (cc[4 * t + 1] ? 1 : 0) + (cc[4 * t + 2] ? 1 : 0) + (cc[4 * t + 3] ? 1 : 0);
                tA[t] = k;
                tB[t] = k;
            }
            __syncthreads();
            block_excl_scan(tB, done, wtmp);  // tB[t] = sum_{u<t} k_u
            if (!careful) break;
            for (int t = tid; t < done; t += kDistThreads) {
                const int sizeAfter = m + (int)(tB[t] + tA[t]) - (t + 1);
                if (sizeAfter >= N) atomicMin(&sJ, t);
            }
            __syncthreads();
            if (sJ < E - 1 || (sJ == E - 1 && done == E)) break;  // cut found (or everything divided)
        }
        __syncthreads();
        STAMP(12);
        const int J = sJ;
        const uint32_t K = tB[J] + tA[J];  // children of processed nodes
        // survivors: old nodes not processed keep their relative order behind the new ones
        for (int i = tid; i < m; i += kDistThreads) proc[i] = 0;
        __syncthreads();
        for (int t = tid; t <= J; t += kDistThreads) proc[ord[t]] = 1;
        __syncthreads();
        for (int i = tid; i < m; i += kDistThreads) ord2[i] = proc[i] ? 0u : 1u;
        __syncthreads();
        const int nSurv = (int)block_excl_scan(ord2, m, wtmp);
        const int nxt = cur ^ 1;
        const int newSize = (int)K + nSurv;
        if (newSize > cap) {  // cannot happen (size <= max(N+2, 4*nIni)); guard anyway
            if (tid == 0) atomicOr(errFlag, 2);
            break;
        }
        for (int i = tid; i < m; i += kDistThreads) {
            if (!proc[i]) {
                const int pos = (int)K + (int)ord2[i];
                nbB[nxt * cap + pos] = nbB[cur * cap + i];
                nsB[nxt * cap + pos] = nsB[cur * cap + i];
                ncB[nxt * cap + pos] = ncB[cur * cap + i];
                nkB[nxt * cap + pos] = nkB[cur * cap + i];
            }
        }
        // children: groups in reverse processing order, inside a group n4,n3,n2,n1 (push_front, :621-660)
        int nToExpandLocal = 0;
        for (int t = tid; t <= J; t += kDistThreads) {
            const uint32_t i = ord[t];
            const short4 b = nbB[cur * cap + i];
            const uint32_t cb = ncB[cur * cap + i];
            const uint32_t kd = nkB[cur * cap + i];
            const int dep = (int)(kd >> 24);
            const bool virt = dep < D;                       // divided on the histogram: keys did not move
            const uint32_t bid = (cb >> 31) ^ (virt ? 0u : 1u);
            const uint32_t kc = ((uint32_t)(dep + 1) << 24) | (virt ? 4 * (kd & 0xFFFFFFu) : 0u);
            const int xm = b.x + ((b.y - b.x + 1) >> 1);
            const int ym = b.z + ((b.w - b.z + 1) >> 1);
            const uint32_t c0 = cc[4 * t], c1 = cc[4 * t + 1], c2 = cc[4 * t + 2], c3 = cc[4 * t + 3];
            const uint32_t st = nsB[cur * cap + i];
            int pos = (int)(K - (tB[t] + tA[t]));  // sum of k_u for u in (t, J]
            if (c3) { nbB[nxt * cap + pos] = make_short4((short)xm, b.y, (short)ym, b.w); nsB[nxt * cap + pos] = st + c0 + c1 + c2; ncB[nxt * cap + pos] = c3 | (bid << 31); nkB[nxt * cap + pos] = kc + 3; pos++; nToExpandLocal += c3 > 1; }
            if (c2) { nbB[nxt * cap + pos] = make_short4(b.x, (short)xm, (short)ym, b.w); nsB[nxt * cap + pos] = st + c0 + c1; ncB[nxt * cap + pos] = c2 | (bid << 31); nkB[nxt * cap + pos] = kc + 2; pos++; nToExpandLocal += c2 > 1; }
            if (c1) { nbB[nxt * cap + pos] = make_short4((short)xm, b.y, b.z, (short)ym); nsB[nxt * cap + pos] = st + c0; ncB[nxt * cap + pos] = c1 | (bid << 31); nkB[nxt * cap + pos] = kc + 1; pos++; nToExpandLocal += c1 > 1; }
            if (c0) { nbB[nxt * cap + pos] = make_short4(b.x, (short)xm, b.z, (short)ym); nsB[nxt * cap + pos] = st; ncB[nxt * cap + pos] = c0 | (bid << 31); nkB[nxt * cap + pos] = kc; pos++; nToExpandLocal += c0 > 1; }
        }
        // block-wide sum of nToExpand (main mode only needs it)
        {
            int v = nToExpandLocal;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
            __syncthreads();
            if (lane == 0) wtmp[wave] = (uint32_t)v;
            __syncthreads();
        }
        int nToExpand = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) nToExpand += (int)wtmp[w];
        __syncthreads();
        STAMP(13);
        cur = nxt;
        m = newSize;
#ifdef ORBX_DIST_TIMING
        if (tid == 0 && sNStamp < 96) { sStampId[sNStamp] = 1000 + m; sStamp[sNStamp++] = (uint64_t)E; }
#endif
        if (m >= N || m == prevSize) break;             // :669 / :733
        if (!careful && m + nToExpand * 3 > N) careful = true;  // :673
    }

    // ---- best response per node, first pushed wins ties (:742-760) == max record
    uint64_t* outk = kept + (int64_t)f * g->keptFrameRecs + L.keptOff;
    if (m > L.keptCap) { if (tid == 0) atomicOr(errFlag, 4); m = L.keptCap; }
    // 8 lanes per node (final nodes hold ~15 keys): 8 nodes per wave step, loads batched
    for (int i0 = wave * 8; i0 < m; i0 += NW * 8) {
        const int i = i0 + (lane >> 3), sub = lane & 7;
        uint64_t best = 0;
        if (i < m) {
            const uint32_t cb = ncB[cur * cap + i];
            const uint32_t cnt = cb & 0x7FFFFFFFu;
            const uint64_t* srcb = ((cb >> 31) ? bufs[1] : bufs[0]) + nsB[cur * cap + i];
            for (uint32_t p = sub; p < cnt; p += 32) {
                uint64_t k[4];
#pragma unroll
                for (int u = 0; u < 4; u++) k[u] = p + 8 * u < cnt ? srcb[p + 8 * u] : 0ull;
#pragma unroll
                for (int u = 0; u < 4; u++) best = k[u] > best ? k[u] : best;
            }
        }
#pragma unroll
        for (int d = 4; d >= 1; d >>= 1) {
            const uint64_t o = __shfl_xor((unsigned long long)best, d);
            best = o > best ? o : best;
        }
        if (i < m && sub == 0) outk[i] = best;
    }
    if (tid == 0) keptCount[f * g->nlevels + l] = m;
#ifdef ORBX_DIST_TIMING
    __syncthreads();
    STAMP(99);
    if (tid == 0 && l == 0 && by == 0) {
        printf("DIST n=%d N=%d D=%d\n", n, N, D);
        uint64_t prev = sStamp[0];
        for (int i = 0; i < sNStamp; i++) {
            if (sStampId[i] >= 1000) printf("   -> m=%d E=%d\n", sStampId[i] - 1000, (int)sStamp[i]);
            else { printf(" id %3d  +%6.2f us  %6lld cycles\n", sStampId[i], (double)(sStamp[i] - prev) / 100.0, i ? (long long)(sCyc[i] - sCyc[i - 1]) : 0ll); prev = sStamp[i]; }
        }
    }
#endif
#undef STAMP
}

template <bool LDS>
__global__ __launch_bounds__(kDistThreads) void k_distribute(DistArgs da)
{
#if ORBX_DIST_PRIO
    __builtin_amdgcn_s_setprio(ORBX_DIST_PRIO);
#endif
    int lvl = (int)blockIdx.x, fr = (int)blockIdx.y;
    if (da.xcdFrames >= 8 && !xcd_block_frame(da.xcdFrames, lvl, fr)) return;
    distribute_body<LDS>(da, lvl, fr);
}

// ------------------------------------------------------------------ Gaussian 7x7 sigma 2
// cv::GaussianBlur on 8U: kernel x256 -> [18,34,49,55,49,34,18], separable integer
// filter, (sum + 2^15) >> 16, saturate, BORDER_REFLECT_101.
struct BlurTiles { int32_t base[ORBX_MAXL + 1]; int32_t tilesX[ORBX_MAXL]; };

__device__ __forceinline__ int reflect101(int p, int n)
{
    if (n == 1) return 0;
    p = p < 0 ? -p : p;               // one reflection each side covers every tile of a level
    p = p >= n ? 2 * n - 2 - p : p;   // that is larger than the 3 px halo; tiny levels loop
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

// 120x32 output tile per 256-thread block; input 128 x 38 (4 px / 3 rows of halo, dword aligned) in LDS.
constexpr int kBlurTW = 120, kBlurTH = 32;

// Input tile of k_blur_mfma: (TH+6) rows x 32 dwords at a pitch of IN_STRIDE dwords, every dword XORed with XORV on the way
// in.  Global traffic is aligned dwords, all loads issued before the first LDS store.  Ends with the tile complete and
// the workgroup synchronised.
template <int IN_STRIDE, uint32_t XORV>
__device__ __forceinline__ void blur_load_tile(uint32_t* __restrict__ in, const uint8_t* __restrict__ S, int stride, int w, int h,
                                               int tx0, int ty0, int tid)
{
    constexpr int TW = kBlurTW, TH = kBlurTH;
    constexpr int IN_DW = (TW + 8) / 4;       // 32 dwords per input row (4 px margin each side)
    bool patchCols = false;
    // (TH+6)*IN_DW = 1216 dwords, 5 per thread (rows tid/32 + 8k of dword column tid%32).  Tiles that touch no image
    // border (block-uniform test) skip the reflection.
    {
        constexpr int PER = (TH + 6 + 7) / 8;
        const int c = tid & 31, r0 = tid >> 5;
        const int x0 = tx0 - 4 + 4 * c;
        uint32_t regs[PER];
        if (tx0 >= 4 && tx0 + TW + 4 <= w && ty0 >= 3 && ty0 + TH + 3 <= h) {
            const uint8_t* p = S + (int64_t)(ty0 - 3 + r0) * stride + x0;
#pragma unroll
            for (int k = 0; k < PER; k++) regs[k] = *(const uint32_t*)(p + (int64_t)min(8 * k, TH + 5 - r0) * stride);
        } else if (w >= 12 && h >= 12) {
            // Border tile, the usual case (every tile of the small levels, a third of level 0): rows are reflected in the
            // load (one reflection covers a 3 px halo), columns are loaded as clamped aligned dwords and the few halo
            // bytes that lie outside the image are patched IN LDS below -- round 1 reflected them byte by byte in the
            // load, which every wave of a left/right border tile paid with ~290 instructions (all eight waves hold a
            // lane of dword column 0 and 31).
            const int xs = min(max(x0, 0), (w - 1) & ~3);   // valid aligned address for every lane
#pragma unroll
            for (int k = 0; k < PER; k++) {
                const int yy = ty0 + min(r0 + 8 * k, TH + 5) - 3;
                const int sy = min(max(yy < 0 ? -yy : (yy >= h ? 2 * h - 2 - yy : yy), 0), h - 1);
                regs[k] = *(const uint32_t*)(S + (int64_t)sy * stride + xs);
            }
            patchCols = true;
        } else {
            const bool edge = !(x0 >= 0 && x0 + 3 < w);
            const int xs = min(max(x0, 0), (w - 1) & ~3);   // valid aligned address for every lane
#pragma unroll
            for (int k = 0; k < PER; k++) {
                const int sy = reflect101(ty0 + min(r0 + 8 * k, TH + 5) - 3, h);
                regs[k] = *(const uint32_t*)(S + (int64_t)sy * stride + xs);
            }
            if (edge) {  // dword straddles the image border: reflect byte by byte
#pragma unroll
                for (int k = 0; k < PER; k++) {
                    const uint8_t* row = S + (int64_t)reflect101(ty0 + min(r0 + 8 * k, TH + 5) - 3, h) * stride;
                    uint32_t v = 0;
#pragma unroll
                    for (int b = 0; b < 4; b++) v |= (uint32_t)row[reflect101(x0 + b, w)] << (8 * b);
                    regs[k] = v;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < PER; k++)
            if (r0 + 8 * k < TH + 6) in[(r0 + 8 * k) * IN_STRIDE + c] = regs[k] ^ XORV;
    }
    __syncthreads();
    if (patchCols && (tx0 == 0 || tx0 + TW + 4 > w)) {  // block-uniform
        uint8_t* const inb = (uint8_t*)in;
        if (tx0 == 0 && tid < TH + 6) {
            // x = -4 .. -1 (dword column 0) mirror x = 4 .. 1 (BORDER_REFLECT_101)
            const uint32_t d1 = in[tid * IN_STRIDE + 1], d2 = in[tid * IN_STRIDE + 2];
            in[tid * IN_STRIDE] = __builtin_amdgcn_perm(d2, d1, 0x01020304u);  // d2.b0, d1.b3, d1.b2, d1.b1
        }
        if (tx0 + TW + 4 > w && tid >= 64 && tid < 64 + 4 * (TH + 6)) {
            // x = w + k mirrors x = w - 2 - k; only the three columns right of the image are ever read
            const int r = (tid - 64) >> 2, k = (tid - 64) & 3;
            const int bd = w + k - (tx0 - 4), bs = w - 2 - k - (tx0 - 4);
            if (bd < 4 * IN_DW && bs >= 0) inb[r * IN_STRIDE * 4 + bd] = inb[r * IN_STRIDE * 4 + bs];
        }
        __syncthreads();
    }
}

// The four constant matrix-core operands of k_blur_mfma, one 16-byte fragment per lane each (lane = index + 32 * half, byte
// b of half h = contraction slot (h, b)):
//   [0], [1]  horizontal taps, B side: output column n of a 32-column strip reads strip columns n+1 .. n+7 (the tile
//             starts 4 px left of the outputs); slot (h, b) = strip column 16h + b of the first / second 32 columns
//   [2], [3]  vertical taps, A side: output row m reads tile rows m .. m+6; slot (h, b = 4g + j) = the tile row
//             8g + 4h + j that accumulator element 4g + j of the horizontal product holds in that lane half
struct BlurOps { uint32_t v[4][64][4]; };
constexpr BlurOps make_blur_ops()
{
    constexpr int W[7] = {18, 34, 49, 55, 49, 34, 18};
    BlurOps o{};
    for (int op = 0; op < 4; op++)
        for (int lane = 0; lane < 64; lane++) {
            const int i = lane & 31, hh = lane >> 5;
            for (int b = 0; b < 16; b++) {
                int t = 0;
                if (op < 2) t = (32 * op + 16 * hh + b) - (i + 1);
                else t = (32 * (op - 2) + 8 * (b >> 2) + 4 * hh + (b & 3)) - i;
                const uint32_t val = (t >= 0 && t <= 6) ? (uint32_t)W[t] : 0u;
                o.v[op][lane][b >> 2] |= val << (8 * (b & 3));
            }
        }
    return o;
}
__device__ const BlurOps kBlurOps = make_blur_ops();

// 7x7 Gaussian as two exact integer matrix products on the matrix cores (v_mfma_i32_32x32x32_i8): the pipeline these
// kernels run in is bound by VALU issue while the matrix pipe idles, and a separable filter is a
// pair of banded-matrix products, out = Kv . (X . Kh).  Every wave owns a 32-column strip of the tile:
//   H = X . Kh   A = 16-byte row fragments of the tile straight from LDS (pixels - 128, signed bytes; done by the
//                loader's XOR), B = the constant tap matrix; accumulator preset to 128, so G = sum - 128*257 + 128 lies
//                in [-32768, 32767] and splits exactly into a signed high byte and a signed low byte (G.b0 ^ 0x80);
//   out = Kv . H the accumulator layout of H (lane = column, element = row) IS the B-side layout of the second product,
//                contraction over rows: four v_perm per four values pack the two byte planes, no transposition, no LDS;
//                A = the constant tap matrix; high and low planes accumulate apart (the weights would not fit a byte
//                scaled by 256), joined by one v_lshl_add; the presets carry the 128*257 offsets and the 2^15 rounding.
// Tile rows 32 .. 37 (the lower halo) are a second, mostly empty 32-row product on both sides.  Results leave as
// bytes into an LDS tile (ds_write_b8_d16_hi takes bits 16 .. 23 of the saturated sum) and are stored as row dwords.
constexpr int kBlurMfmaInStride = 36;      // dwords: 144-byte rows keep the 16-byte fragment reads conflict free
constexpr int kBlurMfmaOutStride = 33;     // dwords per output row (128 bytes + 4)
constexpr int kBlurMfmaInWords = (kBlurTH + 6) * kBlurMfmaInStride + 8, kBlurMfmaOutWords = kBlurTH * kBlurMfmaOutStride;

// one tile (bx of frame fr) by the 256 threads `tid`; ONE workgroup barrier inside, reached by every thread (valid or not)
__device__ __forceinline__ void blur_mfma_tile(const Geom* __restrict__ g, const FrameSrc& src, const BlurTiles& bt, const int bx, const int fr,
                                               const bool valid, const int tid, uint32_t* in, uint32_t* outT)
{
    constexpr int TW = kBlurTW, TH = kBlurTH;
    constexpr int IN_STRIDE = kBlurMfmaInStride, OUT_STRIDE = kBlurMfmaOutStride;
    typedef int b4i __attribute__((ext_vector_type(4)));
    typedef int b16i __attribute__((ext_vector_type(16)));
    if (!valid) { __syncthreads(); return; }
    const int f = fr + src.f0;
    int l = 0;
    while (l + 1 < g->nlevels && bx >= bt.base[l + 1]) l++;
    const int tIdx = bx - bt.base[l];
    const int tx0 = (tIdx % bt.tilesX[l]) * TW, ty0 = (tIdx / bt.tilesX[l]) * TH;
    const LevelGeom& L = g->lv[l];
    const int w = L.w, h = L.h;
    int stride;
    const uint8_t* S = level_ptr(g, src, f, l, stride);
    const int wave = tid >> 6, lane = tid & 63, hh = lane >> 5, m = lane & 31;
    const b4i* ops = (const b4i*)kBlurOps.v;
    const b4i hB0 = ops[lane], hB1 = ops[64 + lane], vA0 = ops[128 + lane], vA1 = ops[192 + lane];

    blur_load_tile<IN_STRIDE, 0x80808080u>(in, S, stride, w, h, tx0, ty0, tid);

    if (tx0 + 32 * wave < w) {  // strips right of the image have nothing to add (wave-uniform)
        const uint8_t* inb = (const uint8_t*)in + 32 * wave + 16 * hh;
        b16i G0, G1;
#pragma unroll
        for (int r = 0; r < 16; r++) G0[r] = G1[r] = 128;
        {
            const b4i a0 = *(const b4i*)(inb + m * (IN_STRIDE * 4)), a1 = *(const b4i*)(inb + m * (IN_STRIDE * 4) + 32);
            G0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, hB0, G0, 0, 0, 0);
            G0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, hB1, G0, 0, 0, 0);
        }
        {
            const int m1 = min(32 + m, TH + 5);
            const b4i a0 = *(const b4i*)(inb + m1 * (IN_STRIDE * 4)), a1 = *(const b4i*)(inb + m1 * (IN_STRIDE * 4) + 32);
            G1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, hB0, G1, 0, 0, 0);
            G1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, hB1, G1, 0, 0, 0);
        }
        b4i lo0, hi0, lo1 = {0, 0, 0, 0}, hi1 = {0, 0, 0, 0};
#pragma unroll
        for (int gq = 0; gq < 4; gq++) {
            const uint32_t t0 = __builtin_amdgcn_perm((uint32_t)G0[4 * gq + 1], (uint32_t)G0[4 * gq], 0x05010400u);
            const uint32_t t1 = __builtin_amdgcn_perm((uint32_t)G0[4 * gq + 3], (uint32_t)G0[4 * gq + 2], 0x05010400u);
            lo0[gq] = (int)(__builtin_amdgcn_perm(t1, t0, 0x05040100u) ^ 0x80808080u);
            hi0[gq] = (int)__builtin_amdgcn_perm(t1, t0, 0x07060302u);
        }
        {   // tile rows 32 .. 37 sit in elements 0 .. 3 of the second product; the taps of every other slot are zero
            const uint32_t t0 = __builtin_amdgcn_perm((uint32_t)G1[1], (uint32_t)G1[0], 0x05010400u);
            const uint32_t t1 = __builtin_amdgcn_perm((uint32_t)G1[3], (uint32_t)G1[2], 0x05010400u);
            lo1[0] = (int)(__builtin_amdgcn_perm(t1, t0, 0x05040100u) ^ 0x80808080u);
            hi1[0] = (int)__builtin_amdgcn_perm(t1, t0, 0x07060302u);
        }
        b16i aHi, aLo;
#pragma unroll
        for (int r = 0; r < 16; r++) { aHi[r] = 0; aLo[r] = 257 * 32896 + 32768; }
        aHi = __builtin_amdgcn_mfma_i32_32x32x32_i8(vA0, hi0, aHi, 0, 0, 0);
        aHi = __builtin_amdgcn_mfma_i32_32x32x32_i8(vA1, hi1, aHi, 0, 0, 0);
        aLo = __builtin_amdgcn_mfma_i32_32x32x32_i8(vA0, lo0, aLo, 0, 0, 0);
        aLo = __builtin_amdgcn_mfma_i32_32x32x32_i8(vA1, lo1, aLo, 0, 0, 0);
        uint8_t* ob = (uint8_t*)outT + (4 * hh) * (OUT_STRIDE * 4) + 32 * wave + m;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            // (sum + 2^15) >> 16, saturated (the sum can reach 257 * 257 * 255)
            const uint32_t v = min((uint32_t)((aHi[r] << 8) + aLo[r]), 0xFFFFFFu);
            ob[((r & 3) + 8 * (r >> 2)) * (OUT_STRIDE * 4)] = (uint8_t)(v >> 16);
        }
    }
    __syncthreads();

    const int cq = tid & 31, rq = tid >> 5;
    const int x = tx0 + 4 * cq;
    if (cq >= TW / 4 || x >= w) return;
    uint8_t* D = src.blur + (int64_t)f * g->blurFrameBytes + L.blurOff + (int64_t)(ty0 + rq) * L.blurStride + x;
    const int nrows = min(TH, h - ty0);
#pragma unroll
    for (int k = 0; k < TH / 8; k++)
        if (rq + 8 * k < nrows) *(uint32_t*)(D + (int64_t)(8 * k) * L.blurStride) = outT[(rq + 8 * k) * OUT_STRIDE + cq];
}

__global__ __launch_bounds__(256, 2) void k_blur_mfma(const Geom* __restrict__ g, FrameSrc src, BlurTiles bt, int nframes)
{
#if ORBX_BLUR_PRIO
    __builtin_amdgcn_s_setprio(ORBX_BLUR_PRIO);
#endif
    __shared__ __attribute__((aligned(16))) uint32_t in[kBlurMfmaInWords];
    __shared__ uint32_t outT[kBlurMfmaOutWords];
    int bx, fr;
    if (!xcd_block_frame(nframes, bx, fr)) return;
    blur_mfma_tile(g, src, bt, bx, fr, true, (int)threadIdx.x, in, outT);
}

// The quadtree and the Gaussian of a robot's live frame in ONE launch: the first blocks of the grid are k_distribute's (one per
// level), the others blur two tiles each (a half of the 512 threads per tile).  The two do not depend on each other -- the
// quadtree needs FAST's candidates, the blur the pyramid -- but in the one queue a live chain runs on they could only follow
// each other: 5.6 us of blur behind 28 us of quadtree that keeps 8 of the 256 CUs busy.  (Two queues cost an event each
// way, ~5 us apiece; hipExtAnyOrderLaunch is ignored on gfx9.)  Latency mode only: grid (levels + ceil(tiles / 2), frames).
__global__ __launch_bounds__(kDistThreads) void k_distribute_blur(DistArgs da, FrameSrc src, BlurTiles bt, int nTiles)
{
    static_assert(kDistThreads == 512, "two blur tiles of 256 threads per block");
    const int nl = da.g->nlevels;
    if ((int)blockIdx.x < nl) { distribute_body<true>(da, (int)blockIdx.x, (int)blockIdx.y); return; }
    __shared__ __attribute__((aligned(16))) uint32_t in2[2][kBlurMfmaInWords];
    __shared__ uint32_t outT2[2][kBlurMfmaOutWords];
    const int half = (int)threadIdx.x >> 8;
    const int tile = 2 * ((int)blockIdx.x - nl) + half;
    blur_mfma_tile(da.g, src, bt, tile, (int)blockIdx.y, tile < nTiles, (int)threadIdx.x & 255, in2[half], outT2[half]);
}

// ------------------------------------------------------------------ orientation + rBRIEF + pack
// dword load (any byte alignment) in the SGPR-base + 32-bit lane offset form.  The compiler does not track it: the
// consumer waits with an explicit s_waitcnt vmcnt (loads return in issue order).
__device__ __forceinline__ void gload_sbase(uint32_t& dst, uint32_t voff, const uint8_t* sbase)
{
    asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase));
}

__device__ const int8_t d_pattern[1024] __attribute__((aligned(16))) = {
#include "brief_pattern.inc"
};

// cv::fastAtan2 (degrees), every operation a separate binary32 rounding
__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    constexpr float k180pi = (float)(180.0 / 3.1415926535897932384626433832795);
    constexpr float p1 = 0.9997878412794807f * k180pi;
    constexpr float p3 = -0.3258083974640975f * k180pi;
    constexpr float p5 = 0.1555786518463281f * k180pi;
    constexpr float p7 = -0.04432655554792128f * k180pi;
    constexpr float eps = (float)2.2204460492503131e-16;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, eps));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, eps));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

// sinf / cosf of the steering angle: the reference's `(float)cos(angle)` has a float argument under `using namespace
// std` (ORBextractor.cc:65,112-113) = cosf.  orbx_sincosf.h restates glibc's sinf/cosf (one binary64 polynomial, rounded
// once); tests/cpp/sincos_check.c runs the same text against the host libm over every binary32 angle in [0, 2pi].
#define ORBX_HD __device__ __forceinline__
#include "orbx_sincosf.h"

constexpr int kKpPerBlock = 16;
struct KpBlocks { int32_t base[ORBX_MAXL + 1]; };  // block index -> level (kKpPerBlock keypoints per block)

// The blurred patch a keypoint's 512 steered samples can reach is a DISC, not the 39 x 40 rectangle round 4 fetched: a
// pattern point at radius r lands, rotated by any angle and rounded, within |Y| <= r + 0.5 and, in row Y, within
// |X| <= sqrt(r^2 - (|Y| - 0.5)^2) + 0.5; the pattern's largest radius is 18.385 (13, 13), so rows +-19 are never read and
// row Y needs |X| <= kDiscHalf[|Y|] (a CPU test under tests/ recomputes the table from the pattern and from a sweep of the angle).  On the
// kernel's grid of dwords from byte cx - 18 that is 308 of 370 (row, dword) items: FIVE gather loads per keypoint
// instead of seven (a load instruction costs the CU's memory pipe ~7.5 cycles and ~1.1 more per cache line it touches,
// tools/orient_npi_exp.sh; the kernel is bound by exactly that).  The LDS patch keeps its rectangular 40-byte rows --
// a test's address stays one v_mad -- the slots outside the disc are simply never written or read.
constexpr int kDiscR = 18;
constexpr int kDiscHalf[kDiscR + 1] = {18, 18, 18, 18, 18, 18, 18, 17, 17, 16, 16, 15, 14, 13, 12, 11, 10, 8, 6};
constexpr int kDiscLoads = 5;
struct DiscItems { uint16_t rc[kDiscLoads * 64]; int n; };   // row << 4 | dword of item t; the tail repeats the last item
constexpr DiscItems make_disc_items()
{
    DiscItems d{};
    int n = 0;
    for (int r = 0; r <= 2 * kDiscR; r++) {
        const int ay = r < kDiscR ? kDiscR - r : r - kDiscR, h = kDiscHalf[ay];
        for (int c = (kDiscR - h) / 4; c <= (kDiscR + h) / 4; c++) d.rc[n++] = (uint16_t)(r << 4 | c);
    }
    d.n = n;
    for (int t = n; t < kDiscLoads * 64; t++) d.rc[t] = d.rc[n - 1];
    return d;
}
__device__ const DiscItems kDisc = make_disc_items();
static_assert(make_disc_items().n <= kDiscLoads * 64 && make_disc_items().n > (kDiscLoads - 1) * 64, "five loads, not four");

// Four keypoints per wave, one per quarter-wave (16 lanes): about half of a keypoint's instructions are
// quarter-uniform (level and slot bookkeeping, fastAtan2, the binary64 sin/cos, the keypoint record) and cost a
// full wave instruction however many lanes need them, so one wave now pays them for four keypoints.
//  * IC_Angle: the 31 x 32 patch as 248 (row, dword) items, 16 per lane, unaligned dword loads all in flight;
//    the masked moments are v_dot4_u32_u8 sums against per-item byte weights kept in LDS:
//    m10 = sum dot4(px, u + 16) - 16 * sum dot4(px, 1), m01 = sum v * dot4(px, 1); butterfly over 16 lanes.
//  * steered BRIEF: lane j of the quarter owns tests j, j + 16, ...; one ballot serves the four keypoints, its
//    16-bit field of the quarter is descriptor word t, parked in lane t and stored as 16 x 2 bytes.
__global__ __launch_bounds__(256) void k_orient_desc(const Geom* __restrict__ g, FrameSrc src, KpBlocks kb,
                                                    const uint64_t* __restrict__ kept,
                                                    const int32_t* __restrict__ keptCount,
                                                    OrbxKeyPointDev* __restrict__ outKps,
                                                    uint8_t* __restrict__ outDesc, int32_t* __restrict__ outCount, int nframes,
                                                    uint8_t* __restrict__ outX, int64_t xPitch, int64_t xAngOff)
{
#if ORBX_DESC_PRIO
    __builtin_amdgcn_s_setprio(ORBX_DESC_PRIO);
#endif
#ifdef ORBX_ORIENT_TIMING
    uint64_t ts[10]; int nts = 0;
#define OSTAMP() ts[nts++] = wall_clock64()
#else
#define OSTAMP() do {} while (0)
#endif
    OSTAMP();
    __shared__ float4 spat[256];           // 256 tests x (x0, y0, x1, y1)
    __shared__ uint32_t swu[256], sw1[256];  // per item: byte weights u + 16 (0 outside the disc), disc flags
    constexpr int PR = kDiscR, PDW = 10, PROWS = 2 * PR + 1;  // blurred patch: rows cy-18 .. cy+18, bytes cx-18 .. cx+21, of which the disc is fetched
    __shared__ uint32_t spatch[kKpPerBlock][PROWS * PDW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, ql = lane & 15;
    int bx, fr;
    if (!xcd_block_frame(nframes, bx, fr)) return;
    const int f = fr + src.f0;
    // The wave's life is a chain of memory latencies (pattern table -> counts -> record -> patches).  The first three
    // are independent and issued together (the record is read unconditionally -- the block grid stays inside the
    // frame's kept segment + slack -- and masked afterwards); the patch loads go out as soon as the record is in, and
    // the block's tables and their barrier are built underneath them.
    int l = 0;
    while (l + 1 < g->nlevels && bx >= kb.base[l + 1]) l++;
    const int idx = (bx - kb.base[l]) * kKpPerBlock + wave * 4 + q;
    const int nl = g->nlevels;
    const LevelGeom& L = g->lv[l];
    const uint64_t rec0 = kept[(int64_t)f * g->keptFrameRecs + L.keptOff + idx];
    int32_t pk = ((const int32_t*)d_pattern)[tid];
    const int tv = (tid >> 3) - kHalfPatch;
    int um = g->umax[(tv < 0 ? -tv : tv) & 15];
    int before = 0, totalAll = 0, mine = 0;
    for (int i = 0; i < nl; i++) {
        const int c = keptCount[f * nl + i];
        if (i < l) before += c;
        if (i == l) mine = c;
        totalAll += c;
    }
    if (bx == 0 && tid == 0) outCount[f] = totalAll < g->maxKp ? totalAll : g->maxKp;
    // the disc's (row, dword) items of this lane, 64 per load (the items past the disc's end repeat the last one: same
    // address, same LDS slot, so neither the loads nor the stores need a predicate)
    int ditem[kDiscLoads];
#pragma unroll
    for (int i = 0; i < kDiscLoads; i++) ditem[i] = kDisc.rc[64 * i + lane];
    asm volatile("" : "+v"(pk), "+v"(um), "+v"(ditem[0]), "+v"(ditem[1]), "+v"(ditem[2]), "+v"(ditem[3]), "+v"(ditem[4]));  // landed here: no compiler-tracked load is in flight next to the untracked ones below
    static_assert(kDiscLoads == 5, "operand list above");
    const int o = before + idx;
    const bool active = idx < mine && o < g->maxKp;   // uniform over the quarter
    const bool anyActive = __builtin_amdgcn_ballot_w64(active) != 0;
    OSTAMP();
    const uint64_t rec = active ? rec0 : 0;
    const int cx = (int)cand_x(rec) + kMinBorder, cy = (int)cand_y(rec) + kMinBorder;  // :843-844
    int stride;
    const uint8_t* img = level_ptr(g, src, f, l, stride);
    const int bs = L.blurStride;
    const uint8_t* blur = src.blur + (int64_t)f * g->blurFrameBytes + L.blurOff;
    const int ctrOff = cy * stride + cx, bctrOff = cy * bs + cx;  // uniform over the quarter

    OSTAMP();
    // Both patches of the wave's four keypoints are fetched by ALL 64 lanes, one keypoint after the other: the lane ->
    // (row, dword) map is then the same for every load, the keypoint's origin is a scalar (v_readlane) and the loads
    // take the SGPR-base form -- no per-load vector address arithmetic (it was a quarter of this kernel's instructions).
    //  * IC_Angle patch: rows cy-15 .. cy+16 x 8 dwords from cx-16, 8 rows per load (row cy+16 carries zero weights)
    //  * blurred patch:  the disc of rows cy-18 .. cy+18 the steered tests can reach (kDisc: 308 dwords, five loads); it
    //    does not depend on the angle, so it is in flight during the moments and lands in LDS before the trigonometry.
    // (16 bytes per lane -- 12 loads instead of 44 -- is slower: the patch origins have byte alignment, and what bounds
    // the kernel is the ~90 cache lines a keypoint touches, not the number of load instructions.)
    uint32_t dw[4][4];
    if (anyActive) {
        const uint32_t voff = (uint32_t)((lane >> 3) * stride + 4 * (lane & 7));
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint8_t* pk = img + (__builtin_amdgcn_readlane(ctrOff, 16 * k) - kHalfPatch * stride - 16);
#pragma unroll
            for (int j = 0; j < 4; j++) gload_sbase(dw[k][j], voff, pk + 8 * j * stride);
        }
    }
    // the blurred disc, five loads per keypoint; an idle quarter repeats keypoint 0 (always live)
    constexpr int NPI = kDiscLoads;
    uint32_t pd[4][NPI];
    int sidx[NPI];
    if (anyActive) {
        uint32_t voff[NPI];
#pragma unroll
        for (int i = 0; i < NPI; i++) {
            const int it = ditem[i], r = it >> 4, c = it & 15;
            voff[i] = (uint32_t)(r * bs + 4 * c);
            sidx[i] = r * PDW + c;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int so = __builtin_amdgcn_readlane((int)active, 16 * k) ? __builtin_amdgcn_readlane(bctrOff, 16 * k) : __builtin_amdgcn_readlane(bctrOff, 0);
            const uint8_t* pk = blur + (so - PR * bs - PR);
#pragma unroll
            for (int i = 0; i < NPI; i++) gload_sbase(pd[k][i], voff[i], pk);
        }
    }
    // the block's tables are built while the patches are in flight
    {
        spat[tid] = make_float4((float)(int8_t)(pk & 0xFF), (float)(int8_t)((pk >> 8) & 0xFF),
                                (float)(int8_t)((pk >> 16) & 0xFF), (float)(int8_t)((pk >> 24) & 0xFF));
        const int u0 = 4 * (tid & 7) - 16;
        const int d = tid < 248 ? um : -1;
        uint32_t wu = 0, w1 = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int u = u0 + j;
            if (u >= -d && u <= d) { wu |= (uint32_t)(u + 16) << (8 * j); w1 |= 1u << (8 * j); }
        }
        swu[tid] = wu; sw1[tid] = w1;
    }
    __syncthreads();
    if (!anyActive) return;
    OSTAMP();
    // IC_Angle: m10 = sum u I, m01 = sum v I over the disc, as v_dot4_u32_u8 sums against byte weights u + 16, v + 16
    // and the disc flags (one set of weights per lane serves the four keypoints)
    int m10, m01;
    asm volatile("s_waitcnt vmcnt(20)"  // the 16 IC_Angle dwords are in; the 20 patch dwords may still be in flight
                 : "+v"(dw[0][0]), "+v"(dw[0][1]), "+v"(dw[0][2]), "+v"(dw[0][3]), "+v"(dw[1][0]), "+v"(dw[1][1]), "+v"(dw[1][2]), "+v"(dw[1][3]),
                   "+v"(dw[2][0]), "+v"(dw[2][1]), "+v"(dw[2][2]), "+v"(dw[2][3]), "+v"(dw[3][0]), "+v"(dw[3][1]), "+v"(dw[3][2]), "+v"(dw[3][3]));
    static_assert(4 * NPI == 20, "vmcnt above");
    {
        int A[4], B[4];
        uint32_t wu[4], w1[4], wv[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            wu[j] = swu[lane + 64 * j];
            w1[j] = sw1[lane + 64 * j];
            wv[j] = w1[j] * (uint32_t)(8 * j + (lane >> 3) + 1);  // (v + 16) per flagged byte, v = row - 15
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t su = 0, sv = 0, s1 = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                su = __builtin_amdgcn_udot4(dw[k][j], wu[j], su, false);
                sv = __builtin_amdgcn_udot4(dw[k][j], wv[j], sv, false);
                s1 = __builtin_amdgcn_udot4(dw[k][j], w1[j], s1, false);
            }
            A[k] = (int)su - 16 * (int)s1;
            B[k] = (int)sv - 16 * (int)s1;
        }
        // reduce-scatter over the wave: v_permlane32_swap leaves keypoints {0,1} in lanes 0..31 and {2,3} in 32..63,
        // v_permlane16_swap leaves keypoint q in quarter q; four row rotations finish the sum inside the quarter
        auto swap32 = [](int a, int b) { const auto r = __builtin_amdgcn_permlane32_swap((unsigned)a, (unsigned)b, false, false); return (int)(r[0] + r[1]); };
        auto swap16 = [](int a, int b) { const auto r = __builtin_amdgcn_permlane16_swap((unsigned)a, (unsigned)b, false, false); return (int)(r[0] + r[1]); };
        m10 = swap16(swap32(A[0], A[2]), swap32(A[1], A[3]));
        m01 = swap16(swap32(B[0], B[2]), swap32(B[1], B[3]));
        m10 += __builtin_amdgcn_update_dpp(0, m10, 0x128, 0xF, 0xF, false);  // row_ror:8
        m01 += __builtin_amdgcn_update_dpp(0, m01, 0x128, 0xF, 0xF, false);
        m10 += __builtin_amdgcn_update_dpp(0, m10, 0x124, 0xF, 0xF, false);
        m01 += __builtin_amdgcn_update_dpp(0, m01, 0x124, 0xF, 0xF, false);
        m10 += __builtin_amdgcn_update_dpp(0, m10, 0x122, 0xF, 0xF, false);
        m01 += __builtin_amdgcn_update_dpp(0, m01, 0x122, 0xF, 0xF, false);
        m10 += __builtin_amdgcn_update_dpp(0, m10, 0x121, 0xF, 0xF, false);
        m01 += __builtin_amdgcn_update_dpp(0, m01, 0x121, 0xF, 0xF, false);
    }
    OSTAMP();
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(pd[0][0]), "+v"(pd[0][1]), "+v"(pd[0][2]), "+v"(pd[0][3]), "+v"(pd[0][4]),
                   "+v"(pd[1][0]), "+v"(pd[1][1]), "+v"(pd[1][2]), "+v"(pd[1][3]), "+v"(pd[1][4]),
                   "+v"(pd[2][0]), "+v"(pd[2][1]), "+v"(pd[2][2]), "+v"(pd[2][3]), "+v"(pd[2][4]),
                   "+v"(pd[3][0]), "+v"(pd[3][1]), "+v"(pd[3][2]), "+v"(pd[3][3]), "+v"(pd[3][4]));
    static_assert(NPI == 5, "operand list above");
    // park the blurred discs
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int i = 0; i < NPI; i++) spatch[wave * 4 + k][sidx[i]] = pd[k][i];
    const float angle = fast_atan2_deg((float)m01, (float)m10);

    // steered BRIEF on the blurred level
    constexpr float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    const float ang = __fmul_rn(angle, factorPI);
    float a, b;  // a = cosf(ang), b = sinf(ang)
    orbx_sincosf_0_2pi(ang, &b, &a);
    OSTAMP();
    // The 512 sample points of a keypoint lie within radius 18.4 of it, inside the LDS patch (gathering the bytes
    // straight from memory costs one cache-line access per lane and test: the address rate of the vector-memory pipe).
    // cvRound (:118-120) = round-half-even = one fp32 add of 1.5 * 2^23: the integer sits in the low mantissa bits,
    // v_mad_i32_i24 reads exactly those, and the biases are folded into the patch offset.
    const uint8_t* const sp8 = (const uint8_t*)&spatch[0][0];
    constexpr float kRnd = 12582912.f;  // 0x4B400000
    const uint32_t adj = (uint32_t)((wave * 4 + q) * (PROWS * PDW * 4) + PR * (PDW * 4) + PR) - 0x400000u * (uint32_t)(PDW * 4) - 0x4B400000u;
    static_assert(PDW * 4 == 40, "row pitch is an inline constant of the mad below");
    uint32_t myWord = 0;
    // Round 5: the rotation of a sample point as four packed-fp32 instructions instead of eight scalar ones.  v_pk_mul_f32 /
    // v_pk_add_f32 issue at the full VALU rate on gfx950 (tools/ubench/pk_min3_rate.hip) and round each lane like their
    // scalar forms; op_sel broadcasts x (or y) to both lanes and exchanges (cos, sin), neg_lo turns the low lane's factor
    // into -sin -- x a - y b is x a + (-(y b)) bit for bit, signed zeros included.
    //   U = (x a, x b), V = (-(y b), y a), W = U + V = (x a - y b, x b + y a), Z = W + 1.5 * 2^23 = (column, row) bits
    typedef float f2v __attribute__((ext_vector_type(2)));
    const f2v AB = {a, b}, KR = {kRnd, kRnd};
    auto rot = [&](const f2v p, int& rx, int& ry) {
        f2v U, V, W, Z;
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(U) : "v"(p), "v"(AB));
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(V) : "v"(p), "v"(AB));
        asm("v_pk_add_f32 %0, %1, %2" : "=v"(W) : "v"(U), "v"(V));
        asm("v_pk_add_f32 %0, %1, %2" : "=v"(Z) : "v"(W), "v"(KR));
        rx = __float_as_int(Z.x); ry = __float_as_int(Z.y);
    };
#pragma unroll
    for (int t = 0; t < 16; t++) {
        const float4 pt = spat[ql + 16 * t];
        int rx0, ry0, rx1, ry1;
        rot(f2v{pt.x, pt.y}, rx0, ry0);
        rot(f2v{pt.z, pt.w}, rx1, ry1);
        uint32_t o0, o1;
        asm("v_mad_i32_i24 %0, %1, 40, %2" : "=v"(o0) : "v"(ry0), "v"(rx0));
        asm("v_mad_i32_i24 %0, %1, 40, %2" : "=v"(o1) : "v"(ry1), "v"(rx1));
        const int t0 = sp8[o0 + adj], t1 = sp8[o1 + adj];
        const uint64_t bal = __builtin_amdgcn_ballot_w64(t0 < t1);
        const uint32_t w16 = (uint32_t)(bal >> (16 * q)) & 0xFFFFu;  // tests 16t .. 16t+15 of this quarter's keypoint
        if (ql == t) myWord = w16;
    }
    OSTAMP();
#ifdef ORBX_ORIENT_TIMING
    if (lane == 0 && wave == 0 && fr == 0 && (bx % 29) == 0)
        printf("ORIENT bx %d l %d: barrier %.2f counts %.2f rec %.2f moments %.2f trig %.2f brief %.2f us\n", bx, l, (ts[1]-ts[0])/100.0, (ts[2]-ts[1])/100.0, (ts[3]-ts[2])/100.0, (ts[4]-ts[3])/100.0, (ts[5]-ts[4])/100.0, (ts[6]-ts[5])/100.0);
#endif
    // The +-1 form the stream matcher's matrix-core scan reads (orbm_kernels.hip: E2M1 nibbles in MFMA tile order, 128 bytes
    // per descriptor), written here instead of by a launch of its own (k_expand_desc: 41 MB of traffic and a kernel per step).
    // The wave's four descriptors go through its OWN patch words in LDS (no block barrier: nobody else reads them), then
    // lane (keypoint kq, 32-bit chunk c) expands one chunk to 16 bytes at block (o >> 5), item c * 32 + (o & 31).
    if (outX) {
        uint16_t* const sw = (uint16_t*)&spatch[wave * 4 + q][0];
        sw[ql] = (uint16_t)myWord;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int kq = (lane >> 3) & 3, c = lane & 7;
        const int oq = __shfl(o, 16 * kq);
        const bool aq = __shfl((int)active, 16 * kq) != 0;
        if (lane < 32 && aq) {
            const uint32_t bits = spatch[wave * 4 + kq][c];
            auto pm1 = [](uint32_t b8) {  // 8 bits -> 8 nibbles, set = +1 (0x2), clear = -1 (0xA)
                uint32_t x = (b8 | (b8 << 12)) & 0x000F000Fu;
                x = (x | (x << 6)) & 0x03030303u;
                x = (x | (x << 3)) & 0x11111111u;
                return 0x22222222u | ((x ^ 0x11111111u) << 3);
            };
            const uint4 e = make_uint4(pm1(bits & 255), pm1((bits >> 8) & 255), pm1((bits >> 16) & 255), pm1(bits >> 24));
            *(uint4*)(outX + (int64_t)f * xPitch + (int64_t)(oq >> 5) * 4096 + (int64_t)(c * 32 + (oq & 31)) * 16) = e;
        }
    }
    if (active) {
        ((uint16_t*)(outDesc + ((int64_t)f * g->maxKp + o) * 32))[ql] = (uint16_t)myWord;
        if (ql == 1 && outX) *(float*)(outX + (int64_t)f * xPitch + xAngOff + (int64_t)o * 4) = angle;   // the stream matcher's compact angle array
        if (ql == 0) {
            OrbxKeyPointDev kp;
            kp.x = __fmul_rn((float)cx, L.scale);  // level 0: scale == 1.0f, identity (:1095-1101)
            kp.y = __fmul_rn((float)cy, L.scale);
            kp.size = L.kpSize;
            kp.angle = angle;
            kp.response = (float)cand_resp(rec);
            kp.octave = l;
            kp.class_id = -1;
            outKps[(int64_t)f * g->maxKp + o] = kp;
        }
    }
}

// ------------------------------------------------------------------ small movers of the host path
// One or two host frames into HBM by a KERNEL instead of the DMA engine (the one-frame-per-call entry): the frames sit in
// pinned host memory the device can address, every lane fetches 16 bytes over the link, all loads independent.  The
// copy is no faster than the engine's (~12 against 16 us for a 1241x376 frame) -- but the chain's first kernel follows
// it in the same queue with nothing in between, where the engine's completion reaches the compute queue ~20 us late
// (tools/live_dma_gap.sh).
__global__ __launch_bounds__(256) void k_upload(const uint4* __restrict__ src, uint4* __restrict__ dst, int n16)
{
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n16) ((u32x4*)dst)[i] = __builtin_nontemporal_load((const u32x4*)src + i);
}

// the same for frames that do NOT lie back to back (the cameras of several robots in one call: a ring buffer each):
// blockIdx.y = frame, one source pointer per frame
struct UploadSrcs { const uint4* src[8]; };
__global__ __launch_bounds__(256) void k_upload_frames(UploadSrcs s, uint4* __restrict__ dst, int n16)
{
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int i = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y;
    if (i < n16) ((u32x4*)dst)[(size_t)f * n16 + i] = __builtin_nontemporal_load((const u32x4*)s.src[f] + i);
}

// raised behind a copy kernel that wrote a caller's results into pinned host memory: the host polls it instead of waiting
// for the stream
__global__ void k_raise_flag(int32_t* flag, int32_t value)
{
    if (threadIdx.x == 0) { __threadfence_system(); __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
}

// The stream's last frame becomes the previous frame of the next batch: slot `src` of one result set -> slot 0 of the
// other (one launch instead of three device-to-device copies in front of the download).
__global__ __launch_bounds__(256) void k_roll_prev(const uint32_t* __restrict__ kpsSrc, const uint32_t* __restrict__ descSrc,
                                                  const int32_t* __restrict__ countSrc, uint32_t* __restrict__ kpsDst,
                                                  uint32_t* __restrict__ descDst, int32_t* __restrict__ countDst,
                                                  const uint4* __restrict__ xSrc, uint4* __restrict__ xDst, int xAng16)
{
    const int n = *countSrc;
    const int t = blockIdx.x * 256 + threadIdx.x, step = gridDim.x * 256;
    for (int i = t; i < 7 * n; i += step) kpsDst[i] = kpsSrc[i];
    for (int i = t; i < 8 * n; i += step) descDst[i] = descSrc[i];
    if (xSrc) {  // the +-1 form travels with the descriptors (whole 32-feature blocks of 4 KiB)
        const int n16 = ((n + 31) >> 5) * 256;
        for (int i = t; i < n16; i += step) xDst[i] = xSrc[i];
        for (int i = t; i < (n + 3) >> 2; i += step) xDst[xAng16 + i] = xSrc[xAng16 + i];   // and the angles behind them
    }
    if (t == 0) *countDst = n;
}

// One-frame-per-call entry: the exact n keypoint records, descriptors and match entries of each frame straight into the
// caller-visible pinned host buffer (device writes over PCIe) -- one launch instead of six copies of ~18 us each.
struct PackArgs {
    const uint32_t* kps; const uint32_t* desc; const int32_t* count;   // first frame of the batch (slot 1 of the set)
    const int32_t* match; const int32_t* nmatch; int32_t* err;   // match / nmatch may be null; err: the batch's error word (read and cleared)
    uint32_t* hKps; uint32_t* hDesc; int32_t* hN; int32_t* hMatch; int32_t* hNmatch; int32_t* hErr;
    int32_t* hFlag; int32_t flagValue; int32_t* blocksDone;  // the last block to finish raises the host flag
    int maxKp;       // pitch (records per frame) of the device arrays
    int hostPitch;   // and of the host arrays: the slot's own block (= maxKp) or the caller's (orbx_submit_batch_into)
};
// frame f's part `part` of `nparts` (each a workgroup of blockDim.x threads); `total` workgroups take part in all
__device__ __forceinline__ void pack_host_part(const PackArgs& a, int f, int part, int nparts, int total)
{
    const int nAll = a.count[f];
    const int n = min(nAll, a.hostPitch);   // never past the caller's row; hN carries the true count
    const int t = part * blockDim.x + threadIdx.x, step = nparts * blockDim.x;
    const int64_t o = (int64_t)f * a.maxKp, oh = (int64_t)f * a.hostPitch;
    // 16 bytes per lane where the frame's slot is 16-byte aligned (maxKp a multiple of 4): a wave instruction then writes
    // a contiguous KiB towards the host
    auto copy_words = [&](uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, int nw) {
        int i0 = 0;
        if ((((uintptr_t)dst | (uintptr_t)src) & 15) == 0) {
            const int n4 = nw >> 2;
            for (int i = t; i < n4; i += step) ((uint4*)dst)[i] = ((const uint4*)src)[i];
            i0 = n4 << 2;
        }
        for (int i = i0 + t; i < nw; i += step) dst[i] = src[i];
    };
    copy_words(a.hKps + oh * 7, a.kps + o * 7, 7 * n);
    copy_words(a.hDesc + oh * 8, a.desc + o * 8, 8 * n);
    if (a.match) copy_words((uint32_t*)a.hMatch + oh, (const uint32_t*)a.match + o, n);
    if (t == 0) {
        a.hN[f] = nAll;
        if (a.nmatch) a.hNmatch[f] = a.nmatch[f];
        if (f == 0 && part == 0) *a.hErr = atomicExch(a.err, 0);  // the batch's own word: handed over and cleared
    }
    // Without a flag the consumer waits for the kernel's completion event, and the end of the kernel publishes: no
    // system-scope fence, no arrival count.
    if (!a.hFlag) return;
    // every block's writes are performed system-wide before its arrival is counted; the last arrival publishes
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        if (atomicAdd(a.blocksDone, 1) == total - 1) {
            *a.blocksDone = 0;
            __threadfence_system();
            __hip_atomic_store(a.hFlag, a.flagValue, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
__global__ __launch_bounds__(256) void k_pack_host(PackArgs a)
{
    pack_host_part(a, blockIdx.y, blockIdx.x, gridDim.x, gridDim.x * gridDim.y);
}

// ------------------------------------------------------------------ Frame::ComputeStereoMatches (Frame.cc:466-638)
// One wave per left keypoint.  Phase 1: the right keypoints whose row band (:481-490) covers the left keypoint's row,
// within one octave and inside the disparity range, scanned by the 64 lanes in index order; best = least Hamming
// distance, lowest index on ties (the reference's row lists hold ascending indices and compare with strict <).
// Phase 2: the 11-position sliding window of :548-587 -- sum of absolute differences of the two centre-subtracted
// 11 x 11 patches on the keypoint's pyramid level -- two patch pixels per lane, then the parabola fit on lane 0.
struct StereoArgs {
    const OrbxKeyPointDev* kL; const uint8_t* dL; const int32_t* nL;
    const OrbxKeyPointDev* kR; const uint8_t* dR; const int32_t* nR;
    const Geom* gL; const Geom* gR;
    FrameSrc srcL, srcR;
    int fL, fR;
    float sf[ORBX_MAXL], isf[ORBX_MAXL];
    float mb, mbf;
    float* uRight; float* depth; int32_t* sad;
};

__global__ __launch_bounds__(256) void k_stereo_match(StereoArgs a)
{
    const int lane = threadIdx.x & 63;
    const int iL = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int N = *a.nL, Nr = *a.nR;
    if (iL >= N) return;
    const OrbxKeyPointDev kp = a.kL[iL];
    const int levelL = kp.octave;
    const float vL = kp.y, uL = kp.x;
    const int nRows = a.gL->lv[0].h;
    const float maxD = __fdiv_rn(a.mbf, a.mb);
    const float minU = __fsub_rn(uL, maxD), maxU = __fadd_rn(uL, 3.0f);  // uL - minD, minD = -3
    float outU = -1.0f, outD = -1.0f;
    int outSad = -1;
    const int row = (int)vL;
    uint32_t key = 0xFFFFFFFFu;
    if (row >= 0 && row < nRows && !(maxU < 0)) {
        uint32_t q[8];
        const uint32_t* qp = (const uint32_t*)(a.dL + (int64_t)iL * 32);
#pragma unroll
        for (int i = 0; i < 8; i++) q[i] = qp[i];
        for (int j = lane; j < Nr; j += 64) {
            const OrbxKeyPointDev r = a.kR[j];
            const float rr = __fmul_rn(2.0f, a.sf[r.octave]);
            int maxr = (int)ceilf(__fadd_rn(r.y, rr)), minr = (int)floorf(__fsub_rn(r.y, rr));
            minr = max(minr, 0); maxr = min(maxr, nRows - 1);
            if (row < minr || row > maxr) continue;
            if (r.octave < levelL - 1 || r.octave > levelL + 1) continue;
            if (!(r.x >= minU && r.x <= maxU)) continue;
            const uint32_t* tp = (const uint32_t*)(a.dR + (int64_t)j * 32);
            int d = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) d += __popc(q[i] ^ tp[i]);
            if (d < 100) key = min(key, ((uint32_t)d << 16) | (uint32_t)j);  // bestDist starts at TH_HIGH, strict <
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) key = min(key, (uint32_t)__shfl_xor((int)key, m));
    if (key != 0xFFFFFFFFu) {
        const int bestIdxR = (int)(key & 0xFFFFu);
        const float uR0 = a.kR[bestIdxR].x;
        const float scaleFactor = a.isf[levelL];
        const float scaleduL = roundf(__fmul_rn(kp.x, scaleFactor));
        const float scaledvL = roundf(__fmul_rn(kp.y, scaleFactor));
        const float scaleduR0 = roundf(__fmul_rn(uR0, scaleFactor));
        constexpr int w = 5, L = 5;
        const int lw = a.gL->lv[levelL].w, lh = a.gL->lv[levelL].h, rw = a.gR->lv[levelL].w, rh = a.gR->lv[levelL].h;
        const int y0 = (int)__fsub_rn(scaledvL, (float)w), x0 = (int)__fsub_rn(scaleduL, (float)w);
        bool ok = !(y0 < 0 || y0 + 2 * w + 1 > lh || y0 + 2 * w + 1 > rh || x0 < 0 || x0 + 2 * w + 1 > lw);
        const float iniu = __fsub_rn(__fadd_rn(scaleduR0, (float)L), (float)w), endu = __fadd_rn(__fadd_rn(__fadd_rn(scaleduR0, (float)L), (float)w), 1.0f);
        ok = ok && !(iniu < 0 || endu >= (float)rw);
        const int xr0 = (int)__fsub_rn(__fsub_rn(scaleduR0, (float)L), (float)w);
        ok = ok && !(xr0 < 0 || (int)endu > rw);
        if (ok) {
            int sL, sR;
            const uint8_t* IL = level_ptr(a.gL, a.srcL, a.fL, levelL, sL) + (int64_t)y0 * sL + x0;
            const uint8_t* IRb = level_ptr(a.gR, a.srcR, a.fR, levelL, sR) + (int64_t)y0 * sR + xr0;  // column of shift -L
            const int cL = IL[w * sL + w];
            // lane owns patch pixels p = lane and lane + 64 (121 in all)
            int pl[2], off[2];
            bool has[2];
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int p = lane + 64 * k;
                has[k] = p < 121;
                const int yy = has[k] ? p / 11 : 0, xx = has[k] ? p - 11 * (p / 11) : 0;
                pl[k] = (int)IL[yy * sL + xx] - cL;
                off[k] = yy * sR + xx;
            }
            int bestSad = 0x7FFFFFFF, bestincR = 0;
            float vD[11];
#pragma unroll
            for (int s = 0; s <= 2 * L; s++) {
                const int cR = IRb[w * sR + w + s];
                int sad = 0;
#pragma unroll
                for (int k = 0; k < 2; k++)
                    if (has[k]) sad += abs(pl[k] - ((int)IRb[off[k] + s] - cR));
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) sad += __shfl_xor(sad, m);
                vD[s] = (float)sad;
                if (sad < bestSad) { bestSad = sad; bestincR = s - L; }
            }
            if (!(bestincR == -L || bestincR == L)) {
                float dist1 = 0.f, dist2 = 0.f, dist3 = 0.f;
#pragma unroll
                for (int s = 1; s < 2 * L; s++)
                    if (s == bestincR + L) { dist1 = vD[s - 1]; dist2 = vD[s]; dist3 = vD[s + 1]; }
                const float deltaR = __fdiv_rn(__fsub_rn(dist1, dist3),
                                               __fmul_rn(2.0f, __fsub_rn(__fadd_rn(dist1, dist3), __fmul_rn(2.0f, dist2))));
                if (!(deltaR < -1 || deltaR > 1)) {
                    float bestuR = __fmul_rn(a.sf[levelL], __fadd_rn(__fadd_rn(scaleduR0, (float)bestincR), deltaR));
                    float disparity = __fsub_rn(uL, bestuR);
                    if (disparity >= 0 && disparity < maxD) {
                        if (disparity <= 0) { disparity = 0.01f; bestuR = (float)__dsub_rn((double)uL, 0.01); }  // Frame.cc:611: uL - 0.01 is a double subtraction
                        outD = __fdiv_rn(a.mbf, disparity);
                        outU = bestuR;
                        outSad = bestSad;
                    }
                }
            }
        }
    }
    if (lane == 0) { a.uRight[iL] = outU; a.depth[iL] = outD; a.sad[iL] = outSad; }
}

// :626-637: sort the accepted (SAD, index) pairs, take the element at size/2 as median, drop everything at or above
// 1.5 * 1.4 * median.  One workgroup; the median by rank counting on packed keys SAD << 16 | index.
__global__ __launch_bounds__(1024) void k_stereo_median(const int32_t* __restrict__ nL, const int32_t* __restrict__ sad,
                                                       float* __restrict__ uRight, float* __restrict__ depth, int32_t* __restrict__ nAccepted)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t keys[];
    __shared__ int sN;
    __shared__ uint32_t sMedian;
    const int N = *nL, tid = threadIdx.x;
    if (tid == 0) sN = 0;
    __syncthreads();
    for (int i = tid; i < N; i += 1024)
        if (sad[i] >= 0) keys[atomicAdd(&sN, 1)] = ((uint32_t)sad[i] << 16) | (uint32_t)i;
    __syncthreads();
    const int n = sN;
    if (tid == 0) *nAccepted = n;
    if (n == 0) return;
    for (int e = tid; e < n; e += 1024) {
        const uint32_t k = keys[e];
        int rank = 0;
        for (int j = 0; j < n; j++) rank += keys[j] < k;
        if (rank == n / 2) sMedian = k;
    }
    __syncthreads();
    const float median = (float)(int)(sMedian >> 16);
    const float thDist = __fmul_rn(1.5f * 1.4f, median);
    for (int e = tid; e < n; e += 1024) {
        const uint32_t k = keys[e];
        if (!((float)(int)(k >> 16) < thDist)) { uRight[k & 0xFFFFu] = -1.0f; depth[k & 0xFFFFu] = -1.0f; }
    }
}

}  // namespace orbx
