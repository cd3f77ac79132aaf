/* orb_stereo.c -- CPU ORACLE (test infrastructure) for Frame::ComputeStereoMatches,
 * /root/reference/SingleRobotScenario/src/Frame.cc:466-638, restated line by line.
 * Third-party arithmetic inside it: cv::Mat::convertTo(CV_32F), `IL - IL.at<float>(w,w) * ones`, cv::norm(IL, IR,
 * NORM_L1) -- on 8-bit patches these are exact integer operations (|values| <= 255, 121 terms), so nothing here is
 * "unpinned": the result is a sum of absolute differences of centre-subtracted patches.
 * Behaviour the reference leaves undefined, defined here (and identically in the product):
 *  - a window that leaves the pyramid level (cv::Mat::rowRange / colRange would throw; real keypoints sit >= 16 px
 *    inside) skips the keypoint;
 *  - a right keypoint whose row band leaves [0, rows) (vRowIndices[yi] out of range, :489-490) is clipped to the image;
 *  - no accepted match at all (vDistIdx[0] read on an empty vector, :627) leaves every entry at -1. */
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "orb_oracle.h"

typedef struct { int dist, idx; } DistIdx;
static int cmp_distidx(const void* a, const void* b)
{
    const DistIdx *x = (const DistIdx*)a, *y = (const DistIdx*)b;
    if (x->dist != y->dist) return x->dist < y->dist ? -1 : 1;   /* std::sort of pair<int,int> */
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}

int orc_compute_stereo_matches(const OrcKeyPoint* keysL, const uint8_t* descL, int N,
                               const OrcKeyPoint* keysR, const uint8_t* descR, int Nr,
                               const OrcPyramid* pyrL, const OrcPyramid* pyrR,
                               const float* mvScaleFactors, const float* mvInvScaleFactors,
                               float mb, float mbf, float* mvuRight, float* mvDepth)
{
    for (int i = 0; i < N; i++) { mvuRight[i] = -1.0f; mvDepth[i] = -1.0f; }
    const int nRows = pyrL->h[0];                                            /* :471 */
    /* :474-491 row table: right keypoint iR is a candidate of every row in [floor(y - r), ceil(y + r)] */
    int* rowCount = (int*)calloc((size_t)nRows + 1, sizeof(int));
    int* minr = (int*)malloc(sizeof(int) * (size_t)(Nr + 1));
    int* maxr = (int*)malloc(sizeof(int) * (size_t)(Nr + 1));
    for (int iR = 0; iR < Nr; iR++) {
        const float kpY = keysR[iR].y;
        const float r = 2.0f * mvScaleFactors[keysR[iR].octave];
        maxr[iR] = (int)ceil((double)(kpY + r));
        minr[iR] = (int)floor((double)(kpY - r));
        if (minr[iR] < 0) minr[iR] = 0;
        if (maxr[iR] > nRows - 1) maxr[iR] = nRows - 1;
        for (int yi = minr[iR]; yi <= maxr[iR]; yi++) rowCount[yi]++;
    }
    int* rowStart = (int*)malloc(sizeof(int) * (size_t)(nRows + 1));
    rowStart[0] = 0;
    for (int y = 0; y < nRows; y++) rowStart[y + 1] = rowStart[y] + rowCount[y];
    int* rowIdx = (int*)malloc(sizeof(int) * (size_t)(rowStart[nRows] + 1));
    memset(rowCount, 0, sizeof(int) * (size_t)nRows);
    for (int iR = 0; iR < Nr; iR++)
        for (int yi = minr[iR]; yi <= maxr[iR]; yi++) rowIdx[rowStart[yi] + rowCount[yi]++] = iR;   /* push order = iR ascending */

    const float minZ = mb, minD = -3, maxD = mbf / minZ;                      /* :494-496 */
    DistIdx* vDistIdx = (DistIdx*)malloc(sizeof(DistIdx) * (size_t)(N + 1));
    int nDist = 0;
    for (int iL = 0; iL < N; iL++) {
        const OrcKeyPoint* kpL = &keysL[iL];
        const int levelL = kpL->octave;
        const float vL = kpL->y, uL = kpL->x;
        const int row = (int)vL;                                              /* vRowIndices[vL]: float -> size_t */
        if (row < 0 || row >= nRows) continue;
        const int cs = rowStart[row], ce = rowStart[row + 1];
        if (cs == ce) continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = 100;                                                   /* ORBmatcher::TH_HIGH */
        int bestIdxR = 0;
        for (int c = cs; c < ce; c++) {
            const int iR = rowIdx[c];
            const OrcKeyPoint* kpR = &keysR[iR];
            if (kpR->octave < levelL - 1 || kpR->octave > levelL + 1) continue;
            const float uR = kpR->x;
            if (uR >= minU && uR <= maxU) {
                const int dist = orc_descriptor_distance(descL + 32 * (size_t)iL, descR + 32 * (size_t)iR);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
            }
        }
        if (bestDist < 100) {                                                 /* :541 subpixel match by correlation */
            const float uR0 = keysR[bestIdxR].x;
            const float scaleFactor = mvInvScaleFactors[kpL->octave];
            const float scaleduL = roundf(kpL->x * scaleFactor);
            const float scaledvL = roundf(kpL->y * scaleFactor);
            const float scaleduR0 = roundf(uR0 * scaleFactor);
            const int w = 5, L = 5;
            const int lw = pyrL->w[levelL], lh = pyrL->h[levelL], rw = pyrR->w[levelL], rh = pyrR->h[levelL];
            const int y0 = (int)(scaledvL - w), x0 = (int)(scaleduL - w);
            if (y0 < 0 || y0 + 2 * w + 1 > lh || y0 + 2 * w + 1 > rh || x0 < 0 || x0 + 2 * w + 1 > lw) continue;   /* see header */
            const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
            if (iniu < 0 || endu >= rw) continue;                             /* :565-568 */
            const int xr0 = (int)(scaleduR0 - L - w);
            if (xr0 < 0 || (int)(scaleduR0 + L + w + 1) > rw) continue;      /* see header */
            const uint8_t* IL = pyrL->data[levelL] + (size_t)y0 * pyrL->stride[levelL] + x0;
            const int cL = IL[(size_t)w * pyrL->stride[levelL] + w];
            int bestSad = INT_MAX, bestincR = 0;
            float vDists[11];
            for (int incR = -L; incR <= L; incR++) {
                const uint8_t* IR = pyrR->data[levelL] + (size_t)y0 * pyrR->stride[levelL] + (int)(scaleduR0 + incR - w);
                const int cR = IR[(size_t)w * pyrR->stride[levelL] + w];
                int sad = 0;
                for (int yy = 0; yy < 2 * w + 1; yy++)
                    for (int xx = 0; xx < 2 * w + 1; xx++)
                        sad += abs((IL[(size_t)yy * pyrL->stride[levelL] + xx] - cL) - (IR[(size_t)yy * pyrR->stride[levelL] + xx] - cR));
                const float dist = (float)sad;                                /* float dist = cv::norm(IL, IR, NORM_L1) */
                if (dist < (float)bestSad) { bestSad = (int)dist; bestincR = incR; }
                vDists[L + incR] = dist;
            }
            if (bestincR == -L || bestincR == L) continue;
            const float dist1 = vDists[L + bestincR - 1], dist2 = vDists[L + bestincR], dist3 = vDists[L + bestincR + 1];
            const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
            if (deltaR < -1 || deltaR > 1) continue;
            float bestuR = mvScaleFactors[kpL->octave] * ((float)scaleduR0 + (float)bestincR + deltaR);
            float disparity = (uL - bestuR);
            if (disparity >= 0 && disparity < maxD) {
                if (disparity <= 0) { disparity = 0.01f; bestuR = (float)((double)uL - 0.01); } /* ref:611: `uL-0.01` is computed in double */
                mvDepth[iL] = mbf / disparity;
                mvuRight[iL] = bestuR;
                vDistIdx[nDist].dist = bestSad; vDistIdx[nDist].idx = iL; nDist++;
            }
        }
    }
    if (nDist > 0) {                                                          /* :626-637 */
        qsort(vDistIdx, (size_t)nDist, sizeof(DistIdx), cmp_distidx);
        const float median = (float)vDistIdx[nDist / 2].dist;
        const float thDist = 1.5f * 1.4f * median;
        for (int i = nDist - 1; i >= 0; i--) {
            if ((float)vDistIdx[i].dist < thDist) break;
            mvuRight[vDistIdx[i].idx] = -1;
            mvDepth[vDistIdx[i].idx] = -1;
        }
    }
    free(vDistIdx); free(rowIdx); free(rowStart); free(maxr); free(minr); free(rowCount);
    return nDist;
}

/* ------------------------------------------------------------------ ref: Frame.cc:641-663 Frame::ComputeStereoFromRGBD
 * imDepth.at<float>(v, u) with float v, u: the arguments convert to int (truncation).  Deterministic refinement: a keypoint
 * outside the depth image (out-of-bounds read in the reference) counts as d = 0. */
void orc_stereo_from_rgbd(const OrcKeyPoint* keys, const OrcKeyPoint* keysUn, int n, const float* depth, int w, int h, int stride,
                          float mbf, float* mvuRight, float* mvDepth)
{
    for (int i = 0; i < n; i++) {
        mvuRight[i] = -1; mvDepth[i] = -1;                    /* :643-644 */
        const int v = (int)keys[i].y, u = (int)keys[i].x;     /* :651-654 */
        const float d = (u >= 0 && u < w && v >= 0 && v < h) ? depth[(size_t)v * stride + u] : 0.f;
        if (d > 0) {                                          /* :656-660 */
            mvDepth[i] = d;
            mvuRight[i] = keysUn[i].x - mbf / d;
        }
    }
}
