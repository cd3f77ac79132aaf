// orbx_common.hpp -- geometry shared by host code and gfx950 kernels.
//
// Everything here is derived once per (params, frame shape) on the host with the
// reference's own formulas (file:line cited at each step; paths under
// /root/reference/SingleRobotScenario/src/) and handed to the kernels by value.
#pragma once

#include <stdint.h>

#define ORBX_MAXL 16

namespace orbx {

constexpr int kPatchSize = 31;       // ORBextractor.cc:72
constexpr int kHalfPatch = 15;       // :73
constexpr int kEdgeThreshold = 19;   // :74
constexpr int kMinBorder = kEdgeThreshold - 3;  // :773  minBorderX = EDGE_THRESHOLD-3

// One pyramid level.  Level 0 aliases the caller's frame (no copy); levels >= 1
// live in the handle's pyramid buffer with 64-byte aligned rows.
struct LevelGeom {
    int32_t w, h;          // cvRound(W*inv), cvRound(H*inv)                       :1111-1112
    int32_t stride;        // bytes between rows (levels >= 1)
    int32_t pyrOff;        // byte offset inside one frame's pyramid buffer (levels >= 1)
    int32_t blurOff;       // byte offset inside one frame's blurred buffer (all levels)
    int32_t blurStride;
    int32_t winW, winH;    // maxBorderX-minBorderX, maxBorderY-minBorderY          :775-782
    int32_t nCols, nRows;  // width/W, height/W (W = 30)                            :784-785
    int32_t wCell, hCell;  // ceil(width/nCols), ceil(height/nRows)                 :786-787
    int32_t cellBase;      // index of this level's first cell in the cell table
    int32_t nCells;
    int32_t candOff;       // record offset inside one frame's candidate buffer
    int32_t candCap;
    int32_t keptOff;       // record offset inside one frame's kept buffer
    int32_t keptCap;
    int32_t nFeat;         // mnFeaturesPerLevel[level]                              :435-446
    int32_t nIni;          // round(width/height)                                    :543
    float hX;              // width/nIni                                             :545
    float scale;           // mvScaleFactor[level]                                   :419-423
    float kpSize;          // (float)(int)(PATCH_SIZE*mvScaleFactor[level])          :837
};

struct Geom {
    int32_t nlevels;
    int32_t w0, h0;
    int32_t iniTh, minTh;
    int32_t totalCells;
    int32_t pyrFrameBytes;    // per frame
    int32_t blurFrameBytes;
    int32_t candFrameRecs;
    int32_t keptFrameRecs;
    int32_t maxKp;            // output capacity per frame
    int32_t maxCellsPerLevel;
    int32_t umax[16];         // :452-469
    LevelGeom lv[ORBX_MAXL];
};

// One FAST cell = one cv::FAST call of the reference (:789-829).  ROI in level
// coordinates; detection area is the ROI minus a 3 px rim.
struct Cell {
    uint16_t level;
    uint16_t x0, y0;   // iniX, iniY
    uint16_t w, h;     // maxX-iniX, maxY-iniY
    uint16_t ci, cj;   // i (row), j (col)
    uint32_t seq;      // rank of the cell in the reference's visiting order (within level)
    uint32_t candOff;  // first record of this cell's segment inside the level's candidate segment
};

// Candidate record (u64), see DESIGN.md:
//   [63:56] response (FAST score 0..254)
//   [55:26] ~order (30 bit)  order = cellSeq<<14 | yInRoi<<7 | xInRoi  = reference push order
//   [25:13] y   [12:0] x     window-relative level coordinates (ORBextractor.cc:822-823)
// max() over a node's records == "largest response, first pushed wins" (:748-757).
__host__ __device__ inline uint64_t pack_cand(uint32_t x, uint32_t y, uint32_t resp, uint32_t order)
{
    return ((uint64_t)resp << 56) | ((uint64_t)((~order) & 0x3FFFFFFFu) << 26) | ((uint64_t)y << 13) | x;
}
// cellSeq 16 bit, y/x inside the ROI 7 bit each
__host__ __device__ inline uint32_t cand_order(uint32_t cellSeq, uint32_t yr, uint32_t xr)
{
    return (cellSeq << 14) | (yr << 7) | xr;
}
__host__ __device__ inline uint32_t cand_x(uint64_t r) { return (uint32_t)(r & 0x1FFFu); }
__host__ __device__ inline uint32_t cand_y(uint64_t r) { return (uint32_t)((r >> 13) & 0x1FFFu); }
__host__ __device__ inline uint32_t cand_resp(uint64_t r) { return (uint32_t)(r >> 56); }

// cv::KeyPoint layout (28 bytes), identical to OrbxKeyPoint of the C ABI
struct OrbxKeyPointDev {
    float x, y, size, angle, response;
    int32_t octave, class_id;
};


// Inclusive prefix sum over the 64 lanes of a wave on the DPP path (row_shr 1/2/4/8 inside a row of 16, then row_bcast:15
// and row_bcast:31 across the rows): six VALU adds.  The same with __shfl_up is six ds_bpermute round trips through the
// LDS unit (~100 cycles each), and the single-workgroup kernels of a live stream's chain (quadtree, frame build,
// candidates) are strings of such scans.  All 64 lanes must be active.
__device__ __forceinline__ int wave_incl_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);   // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);   // row_bcast:31 -> rows 2, 3
    return v;
}

}  // namespace orbx
