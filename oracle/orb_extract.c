/*
 * orb_extract.c -- CPU ORACLE (test infrastructure only), extractor half.
 * PARITY UNPINNED at the OpenCV boundary -- see orb_oracle.h.
 *
 * Restates /root/reference/SingleRobotScenario/src/ORBextractor.cc (cited as
 * "ref:LINE") plus the OpenCV 3.0.0 primitives it calls (cited as "cv3.0 <fn>",
 * restated from the published algorithm).
 * Build with -ffp-contract=off (strict binary32).
 */
#include "orb_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static const int8_t k_pattern[1024] = {
#include "brief_pattern.inc"
};

enum { PATCH_SIZE = 31, HALF_PATCH_SIZE = 15, EDGE_THRESHOLD = 19 }; /* ref:72-74 */

/* cvRound: round-half-to-even (cvtsd2si / lrint), cv3.0 fast_math.hpp */
static int cv_round(double v) { return (int)lrint(v); }

/* ------------------------------------------------------------------ ctor, ref:410-470 */
int orc_extractor_init(OrcExtractor* ex, int nfeatures, float scaleFactor_f, int nlevels,
                       int iniThFAST, int minThFAST)
{
    if (nlevels < 1 || nlevels > ORC_MAX_LEVELS || nfeatures < 1) return -1;
    memset(ex, 0, sizeof(*ex));
    ex->nfeatures = nfeatures;
    ex->scaleFactor = (double)scaleFactor_f;
    ex->nlevels = nlevels;
    ex->iniThFAST = iniThFAST;
    ex->minThFAST = minThFAST;

    ex->mvScaleFactor[0] = 1.0f;
    ex->mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) { /* ref:419-423: float * double -> float */
        ex->mvScaleFactor[i] = (float)((double)ex->mvScaleFactor[i - 1] * ex->scaleFactor);
        ex->mvLevelSigma2[i] = ex->mvScaleFactor[i] * ex->mvScaleFactor[i];
    }
    for (int i = 0; i < nlevels; i++) { /* ref:427-431 */
        ex->mvInvScaleFactor[i] = 1.0f / ex->mvScaleFactor[i];
        ex->mvInvLevelSigma2[i] = 1.0f / ex->mvLevelSigma2[i];
    }

    /* ref:435-446 */
    float factor = (float)(1.0 / ex->scaleFactor); /* 1.0f / double -> double -> float */
    float nDesired = (float)nfeatures * (1 - factor) /
                     (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int level = 0; level < nlevels - 1; level++) {
        ex->mnFeaturesPerLevel[level] = cv_round(nDesired);
        sum += ex->mnFeaturesPerLevel[level];
        nDesired *= factor;
    }
    ex->mnFeaturesPerLevel[nlevels - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;

    /* ref:452-469: end of each row of the circular patch */
    int v, v0;
    int vmax = (int)floor((double)((float)HALF_PATCH_SIZE * sqrtf(2.f) / 2 + 1));
    int vmin = (int)ceil((double)((float)HALF_PATCH_SIZE * sqrtf(2.f) / 2));
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) ex->umax[v] = cv_round(sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (ex->umax[v0] == ex->umax[v0 + 1]) ++v0;
        ex->umax[v] = v0;
        ++v0;
    }
    return 0;
}

/* ref:1111-1112 */
void orc_level_size(const OrcExtractor* ex, int w, int h, int level, int* lw, int* lh)
{
    float scale = ex->mvInvScaleFactor[level];
    *lw = cv_round((double)((float)w * scale));
    *lh = cv_round((double)((float)h * scale));
}

/* ------------------------------------------------------------------ cv3.0 resize INTER_LINEAR, 8UC1
 * fixed point: coefficients scaled by 2048 (INTER_RESIZE_COEF_BITS = 11), horizontal
 * pass into int32, vertical pass ((b0*(r0>>4))>>16 + (b1*(r1>>4))>>16 + 2) >> 2.
 * scale == 2 exactly would switch to INTER_AREA in OpenCV; not reachable for the
 * scale factors this path is used with (asserted by the caller tests). */
static short sat_s16_round(float v)
{
    long r = lrintf(v);
    if (r > 32767) r = 32767;
    if (r < -32768) r = -32768;
    return (short)r;
}

void orc_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride,
                          uint8_t* dst, int dw, int dh, int dstride)
{
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int* xofs = (int*)malloc(sizeof(int) * dw);
    int* yofs = (int*)malloc(sizeof(int) * dh);
    short* ialpha = (short*)malloc(sizeof(short) * 2 * dw);
    short* ibeta = (short*)malloc(sizeof(short) * 2 * dh);
    int xmax = dw;

    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) {
            if (dx < xmax) xmax = dx;
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        }
        xofs[dx] = sx;
        ialpha[dx * 2] = sat_s16_round((1.f - fx) * 2048);
        ialpha[dx * 2 + 1] = sat_s16_round(fx * 2048);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= sy;
        yofs[dy] = sy;
        ibeta[dy * 2] = sat_s16_round((1.f - fy) * 2048);
        ibeta[dy * 2 + 1] = sat_s16_round(fy * 2048);
    }

    int* row0 = (int*)malloc(sizeof(int) * dw);
    int* row1 = (int*)malloc(sizeof(int) * dw);
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = yofs[dy], sy1 = yofs[dy] + 1;
        /* rows clipped to [0, sh-1] when fetched */
        sy0 = sy0 < 0 ? 0 : (sy0 < sh ? sy0 : sh - 1);
        sy1 = sy1 < 0 ? 0 : (sy1 < sh ? sy1 : sh - 1);
        const uint8_t* S0 = src + (size_t)sy0 * sstride;
        const uint8_t* S1 = src + (size_t)sy1 * sstride;
        for (int dx = 0; dx < dw; dx++) {
            int sx = xofs[dx];
            if (dx < xmax) {
                row0[dx] = S0[sx] * ialpha[dx * 2] + S0[sx + 1] * ialpha[dx * 2 + 1];
                row1[dx] = S1[sx] * ialpha[dx * 2] + S1[sx + 1] * ialpha[dx * 2 + 1];
            } else {
                row0[dx] = S0[sx] * 2048;
                row1[dx] = S1[sx] * 2048;
            }
        }
        int b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
        uint8_t* D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; dx++)
            D[dx] = (uint8_t)((((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2);
    }
    free(row0); free(row1); free(xofs); free(yofs); free(ialpha); free(ibeta);
}

/* ------------------------------------------------------------------ cv3.0 FAST, TYPE_9_16, nonmax = true
 * Bresenham circle radius 3, 16 pixels, ring extended to 25 entries. */
static const int k_circle[16][2] = {
    {0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
    {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

static int fast_is_corner(const uint8_t* p, const int* off, int t)
{
    const int v = p[0];
    const int lo = v - t, hi = v + t;
#define CLS(k) (p[off[k]] < lo ? 1 : (p[off[k]] > hi ? 2 : 0))
    /* high-speed test: any 9-arc contains one pixel of every opposite pair */
    int d = CLS(0) | CLS(8);
    if (!d) return 0;
    d &= CLS(2) | CLS(10);
    d &= CLS(4) | CLS(12);
    d &= CLS(6) | CLS(14);
    if (!d) return 0;
    d &= CLS(1) | CLS(9);
    d &= CLS(3) | CLS(11);
    d &= CLS(5) | CLS(13);
    d &= CLS(7) | CLS(15);
#undef CLS
    if (d & 1) { /* darker arc of >= 9 on the 25-entry ring */
        int run = 0;
        for (int k = 0; k < 25; k++) {
            if (p[off[k]] < lo) { if (++run > 8) return 1; } else run = 0;
        }
    }
    if (d & 2) { /* brighter arc */
        int run = 0;
        for (int k = 0; k < 25; k++) {
            if (p[off[k]] > hi) { if (++run > 8) return 1; } else run = 0;
        }
    }
    return 0;
}

/* cv3.0 cornerScore<16>: largest threshold for which the pixel stays a corner, minus... (see App. A.1) */
static int fast_corner_score(const uint8_t* p, const int* off, int threshold)
{
    int d[25];
    const int v = p[0];
    for (int k = 0; k < 25; k++) d[k] = v - p[off[k]];

    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
        if (d[k + 3] < a) a = d[k + 3];
        if (a <= a0) continue;
        for (int m = 4; m <= 8; m++) if (d[k + m] < a) a = d[k + m];
        int x = a < d[k] ? a : d[k];
        if (x > a0) a0 = x;
        x = a < d[k + 9] ? a : d[k + 9];
        if (x > a0) a0 = x;
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
        for (int m = 3; m <= 5; m++) if (d[k + m] > b) b = d[k + m];
        if (b >= b0) continue;
        for (int m = 6; m <= 8; m++) if (d[k + m] > b) b = d[k + m];
        int x = b > d[k] ? b : d[k];
        if (x < b0) b0 = x;
        x = b > d[k + 9] ? b : d[k + 9];
        if (x < b0) b0 = x;
    }
    return -b0 - 1;
}

/* corner predicate only (before score / non-max suppression), for the independent
 * scikit-image fixture tests/golden/fast_detect_skimage.npz */
void orc_fast_corner_mask(const uint8_t* img, int w, int h, int stride, int threshold, uint8_t* mask)
{
    int off[25];
    for (int k = 0; k < 16; k++) off[k] = k_circle[k][0] + k_circle[k][1] * stride;
    for (int k = 16; k < 25; k++) off[k] = off[k - 16];
    memset(mask, 0, (size_t)w * h);
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++)
            mask[(size_t)y * w + x] = (uint8_t)fast_is_corner(img + (size_t)y * stride + x, off, threshold);
}

int orc_fast9_16(const uint8_t* img, int w, int h, int stride, int threshold,
                 OrcCorner* out, int cap)
{
    if (w < 7 || h < 7) return 0;
    if (threshold < 0) threshold = 0;
    if (threshold > 255) threshold = 255;
    int off[25];
    for (int k = 0; k < 16; k++) off[k] = k_circle[k][0] + k_circle[k][1] * stride;
    for (int k = 16; k < 25; k++) off[k] = off[k - 16];

    /* score map, zero outside the detection area [3,w-3) x [3,h-3) (the OpenCV row
     * buffers are zero-initialised and only written at corner positions) */
    uint8_t* score = (uint8_t*)calloc((size_t)w * h * 2, 1);
    uint8_t* flag = score + (size_t)w * h;
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            const uint8_t* p = img + (size_t)y * stride + x;
            if (fast_is_corner(p, off, threshold)) {
                flag[(size_t)y * w + x] = 1;
                score[(size_t)y * w + x] = (uint8_t)fast_corner_score(p, off, threshold);
            }
        }
    /* 3x3 non-max suppression, strict >, row-major output order */
    int n = 0;
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            const uint8_t* s = score + (size_t)y * w + x;
            const int c = s[0];
            if (!flag[(size_t)y * w + x]) continue;
            if (c > s[-1] && c > s[1] && c > s[-w - 1] && c > s[-w] && c > s[-w + 1] &&
                c > s[w - 1] && c > s[w] && c > s[w + 1]) {
                if (n < cap) { out[n].x = x; out[n].y = y; out[n].score = c; }
                n++;
            }
        }
    free(score);
    return n < cap ? n : cap;
}

/* ------------------------------------------------------------------ cv3.0 GaussianBlur 7x7 sigma 2, 8UC1
 * getGaussianKernel(7, 2, CV_32F) -> fixed point x256 -> separable int filter,
 * (sum + 2^15) >> 16, BORDER_REFLECT_101. */
static void gauss7_kernel_q8(int k[7])
{
    float cf[7];
    double sum = 0;
    const double scale2X = -0.5 / (2.0 * 2.0);
    for (int i = 0; i < 7; i++) {
        double x = i - 3.0;
        double t = exp(scale2X * x * x);
        cf[i] = (float)t;
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < 7; i++) {
        cf[i] = (float)(cf[i] * sum);
        k[i] = (int)lrint((double)cf[i] * 256.0);
    }
}

static int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) {
        if (p < 0) p = -p;
        else p = 2 * n - 2 - p;
    }
    return p;
}

void orc_gaussian7_u8(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride)
{
    int k[7];
    gauss7_kernel_q8(k);
    int* tmp = (int*)malloc(sizeof(int) * (size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t* S = src + (size_t)y * sstride;
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int i = 0; i < 7; i++) s += k[i] * S[reflect101(x + i - 3, w)];
            tmp[(size_t)y * w + x] = s;
        }
    }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int i = 0; i < 7; i++) s += k[i] * tmp[(size_t)reflect101(y + i - 3, h) * w + x];
            s = (s + (1 << 15)) >> 16;
            dst[(size_t)y * dstride + x] = (uint8_t)(s < 0 ? 0 : (s > 255 ? 255 : s));
        }
    free(tmp);
}

/* ------------------------------------------------------------------ cv3.0 fastAtan2 (degrees) */
float orc_fast_atan2(float y, float x)
{
    static const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
    static const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
    static const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
    static const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* ref:77-104 IC_Angle (integer moments over the radius-15 disc, raw level image) */
float orc_ic_angle(const uint8_t* img, int stride, int x, int y, const int* umax)
{
    int m_01 = 0, m_10 = 0;
    const uint8_t* center = img + (size_t)y * stride + x;
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0;
        int d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * stride], val_minus = center[u - v * stride];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return orc_fast_atan2((float)m_01, (float)m_10);
}

/* ref:107-147 computeOrbDescriptor (steered BRIEF on the blurred level) */
void orc_brief(const uint8_t* img, int stride, int x, int y, float angle_deg, uint8_t desc[32])
{
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    float angle = angle_deg * factorPI;
    /* ref:65 `using namespace std;` + ref:113 `(float)cos(angle)` with a float argument: std::cos(float) = cosf
       (g++ emits `call cosf`), NOT cos((double)angle) rounded -- the two differ for 0.135 % of the binary32 angles
       in [0, 2pi] on glibc 2.35.  This step is libm-version dependent in the reference itself (orb_oracle.h F6). */
    float a = cosf(angle), b = sinf(angle);
    const uint8_t* center = img + (size_t)y * stride + x;
    const int8_t* pat = k_pattern;
    for (int i = 0; i < 32; ++i, pat += 32) {
        int val = 0;
        for (int k = 0; k < 8; k++) {
            const int8_t* q = pat + 4 * k;
            /* two separately rounded products and one rounded sum per coordinate */
            float r0 = (float)q[0] * b + (float)q[1] * a, c0 = (float)q[0] * a - (float)q[1] * b;
            float r1 = (float)q[2] * b + (float)q[3] * a, c1 = (float)q[2] * a - (float)q[3] * b;
            int t0 = center[cv_round((double)r0) * stride + cv_round((double)c0)];
            int t1 = center[cv_round((double)r1) * stride + cv_round((double)c1)];
            val |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)val;
    }
}

/* ------------------------------------------------------------------ ref:765-829 cell loop */
int orc_level_candidates(const OrcExtractor* ex, const uint8_t* img, int w, int h, int stride,
                         OrcCorner* out, int cap)
{
    const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
    const int maxBorderX = w - EDGE_THRESHOLD + 3, maxBorderY = h - EDGE_THRESHOLD + 3;
    const float W = 30;
    const float width = (float)(maxBorderX - minBorderX);
    const float height = (float)(maxBorderY - minBorderY);
    const int nCols = (int)(width / W), nRows = (int)(height / W);
    if (nCols < 1 || nRows < 1) return 0; /* deterministic refinement, see header */
    const int wCell = (int)ceil((double)(width / nCols));
    const int hCell = (int)ceil((double)(height / nRows));

    OrcCorner* cell = (OrcCorner*)malloc(sizeof(OrcCorner) * 64 * 64);
    int n = 0;
    for (int i = 0; i < nRows; i++) {
        const float iniY = (float)(minBorderY + i * hCell);
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBorderY - 3) continue;
        if (maxY > maxBorderY) maxY = (float)maxBorderY;
        for (int j = 0; j < nCols; j++) {
            const float iniX = (float)(minBorderX + j * wCell);
            float maxX = iniX + wCell + 6;
            if (iniX >= maxBorderX - 6) continue;
            if (maxX > maxBorderX) maxX = (float)maxBorderX;
            const int y0 = (int)iniY, y1 = (int)maxY, x0 = (int)iniX, x1 = (int)maxX;
            const uint8_t* roi = img + (size_t)y0 * stride + x0;
            int nc = orc_fast9_16(roi, x1 - x0, y1 - y0, stride, ex->iniThFAST, cell, 64 * 64);
            if (nc == 0)
                nc = orc_fast9_16(roi, x1 - x0, y1 - y0, stride, ex->minThFAST, cell, 64 * 64);
            for (int k = 0; k < nc; k++) {
                if (n < cap) {
                    out[n].x = cell[k].x + j * wCell; /* ref:822-823 */
                    out[n].y = cell[k].y + i * hCell;
                    out[n].score = cell[k].score;
                }
                n++;
            }
        }
    }
    free(cell);
    return n < cap ? n : -1;
}

/* How many cells of a level the reference's loop visits, how many of them come back empty from cv::FAST at iniThFAST and
 * are run again at minThFAST (ref:808-816), and how many stay empty: the statistic bench.py's `natural` block reports beside
 * its throughput (the synthetic frames retry 0.14 % of their cells, photographs and low-texture scenes far more). */
void orc_level_cell_stats(const OrcExtractor* ex, const uint8_t* img, int w, int h, int stride, int* ncells, int* nretry, int* nempty)
{
    const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
    const int maxBorderX = w - EDGE_THRESHOLD + 3, maxBorderY = h - EDGE_THRESHOLD + 3;
    const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
    const int nCols = (int)(width / 30.f), nRows = (int)(height / 30.f);
    *ncells = *nretry = *nempty = 0;
    if (nCols < 1 || nRows < 1) return;
    const int wCell = (int)ceil((double)(width / nCols)), hCell = (int)ceil((double)(height / nRows));
    OrcCorner* cell = (OrcCorner*)malloc(sizeof(OrcCorner) * 64 * 64);
    for (int i = 0; i < nRows; i++) {
        const float iniY = (float)(minBorderY + i * hCell);
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBorderY - 3) continue;
        if (maxY > maxBorderY) maxY = (float)maxBorderY;
        for (int j = 0; j < nCols; j++) {
            const float iniX = (float)(minBorderX + j * wCell);
            float maxX = iniX + wCell + 6;
            if (iniX >= maxBorderX - 6) continue;
            if (maxX > maxBorderX) maxX = (float)maxBorderX;
            const uint8_t* roi = img + (size_t)(int)iniY * stride + (int)iniX;
            const int rw = (int)maxX - (int)iniX, rh = (int)maxY - (int)iniY;
            (*ncells)++;
            if (orc_fast9_16(roi, rw, rh, stride, ex->iniThFAST, cell, 64 * 64) == 0) {
                (*nretry)++;
                if (orc_fast9_16(roi, rw, rh, stride, ex->minThFAST, cell, 64 * 64) == 0) (*nempty)++;
            }
        }
    }
    free(cell);
}

/* ------------------------------------------------------------------ ref:481-763 quadtree */
typedef struct QNode {
    int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
    int* keys;
    int nkeys;
    int noMore;
    long seq;
    struct QNode *prev, *next;
} QNode;

typedef struct { QNode *head, *tail; int size; long next_seq; } QList;

static QNode* qnode_new(int cap_keys)
{
    QNode* n = (QNode*)calloc(1, sizeof(QNode));
    n->keys = (int*)malloc(sizeof(int) * (cap_keys > 0 ? cap_keys : 1));
    return n;
}
static void qnode_free(QNode* n) { free(n->keys); free(n); }
static void qlist_push_front(QList* l, QNode* n)
{
    n->seq = l->next_seq++;
    n->prev = NULL; n->next = l->head;
    if (l->head) l->head->prev = n; else l->tail = n;
    l->head = n; l->size++;
}
static void qlist_push_back(QList* l, QNode* n)
{
    n->seq = l->next_seq++;
    n->next = NULL; n->prev = l->tail;
    if (l->tail) l->tail->next = n; else l->head = n;
    l->tail = n; l->size++;
}
static QNode* qlist_erase(QList* l, QNode* n) /* returns successor */
{
    QNode* nx = n->next;
    if (n->prev) n->prev->next = n->next; else l->head = n->next;
    if (n->next) n->next->prev = n->prev; else l->tail = n->prev;
    l->size--;
    qnode_free(n);
    return nx;
}

/* ref:481-537 */
static void divide_node(const QNode* p, const OrcCorner* kp, QNode* c[4])
{
    const int halfX = (int)ceil((double)((float)(p->URx - p->ULx) / 2));
    const int halfY = (int)ceil((double)((float)(p->BRy - p->ULy) / 2));
    for (int i = 0; i < 4; i++) c[i] = qnode_new(p->nkeys);
    QNode *n1 = c[0], *n2 = c[1], *n3 = c[2], *n4 = c[3];
    n1->ULx = p->ULx; n1->ULy = p->ULy;
    n1->URx = p->ULx + halfX; n1->URy = p->ULy;
    n1->BLx = p->ULx; n1->BLy = p->ULy + halfY;
    n1->BRx = p->ULx + halfX; n1->BRy = p->ULy + halfY;

    n2->ULx = n1->URx; n2->ULy = n1->URy;
    n2->URx = p->URx; n2->URy = p->URy;
    n2->BLx = n1->BRx; n2->BLy = n1->BRy;
    n2->BRx = p->URx; n2->BRy = p->ULy + halfY;

    n3->ULx = n1->BLx; n3->ULy = n1->BLy;
    n3->URx = n1->BRx; n3->URy = n1->BRy;
    n3->BLx = p->BLx; n3->BLy = p->BLy;
    n3->BRx = n1->BRx; n3->BRy = p->BLy;

    n4->ULx = n3->URx; n4->ULy = n3->URy;
    n4->URx = n2->BRx; n4->URy = n2->BRy;
    n4->BLx = n3->BRx; n4->BLy = n3->BRy;
    n4->BRx = p->BRx; n4->BRy = p->BRy;

    for (int i = 0; i < p->nkeys; i++) {
        const int id = p->keys[i];
        const float x = (float)kp[id].x, y = (float)kp[id].y;
        QNode* t;
        if (x < (float)n1->URx) t = (y < (float)n1->BRy) ? n1 : n3;
        else t = (y < (float)n1->BRy) ? n2 : n4;
        t->keys[t->nkeys++] = id;
    }
    for (int i = 0; i < 4; i++) if (c[i]->nkeys == 1) c[i]->noMore = 1;
}

typedef struct { int size; QNode* node; } SizeNode;
/* Test hook: 0 = the defined tie-break (creation sequence ascending), 1 = the opposite order among equal sizes.  The
 * reference breaks such ties by heap address (ref:684) -- ANY order can come out of it; tests use the hook to show where
 * the choice can and cannot change the result (tests/test_oracle_invariants.py). */
static int g_tie_break_reversed = 0;
void orc_set_tie_break_reversed(int on) { g_tie_break_reversed = on; }
static int cmp_sizenode(const void* a, const void* b)
{
    const SizeNode *x = (const SizeNode*)a, *y = (const SizeNode*)b;
    if (x->size != y->size) return x->size < y->size ? -1 : 1;
    /* reference compares heap pointers here (ref:684); oracle: creation sequence */
    const int c = x->node->seq < y->node->seq ? -1 : (x->node->seq > y->node->seq ? 1 : 0);
    return g_tie_break_reversed ? -c : c;
}

/* push children of a divided node (ref:621-660 / 694-730); records expandable ones */
static void push_children(QList* l, QNode* c[4], SizeNode* vec, int* nvec, int* nToExpand)
{
    for (int i = 0; i < 4; i++) {
        if (c[i]->nkeys > 0) {
            qlist_push_front(l, c[i]);
            if (c[i]->nkeys > 1) {
                if (nToExpand) (*nToExpand)++;
                vec[*nvec].size = c[i]->nkeys;
                vec[*nvec].node = c[i];
                (*nvec)++;
            }
        } else {
            qnode_free(c[i]);
        }
    }
}

int orc_distribute(const OrcCorner* in, int n, int minX, int maxX, int minY, int maxY,
                   int N, OrcCorner* out, int cap)
{
    if (n <= 0) return 0;
    /* ref:543-545 */
    const int nIni = (int)roundf((float)(maxX - minX) / (float)(maxY - minY));
    if (nIni < 1) return -2; /* reference divides by zero here (hX = inf) */
    const float hX = (float)(maxX - minX) / (float)nIni;

    QList l = {0};
    QNode** ini = (QNode**)malloc(sizeof(QNode*) * nIni);
    for (int i = 0; i < nIni; i++) { /* ref:552-563 */
        QNode* ni = qnode_new(n);
        ni->ULx = (int)(hX * (float)i); ni->ULy = 0;
        ni->URx = (int)(hX * (float)(i + 1)); ni->URy = 0;
        ni->BLx = ni->ULx; ni->BLy = maxY - minY;
        ni->BRx = ni->URx; ni->BRy = maxY - minY;
        qlist_push_back(&l, ni);
        ini[i] = ni;
    }
    for (int i = 0; i < n; i++) { /* ref:566-570 */
        size_t r = (size_t)((float)in[i].x / hX);
        if (r >= (size_t)nIni) r = (size_t)nIni - 1; /* unreachable for in-window x; UB in ref */
        ini[r]->keys[ini[r]->nkeys++] = i;
    }
    free(ini);
    for (QNode* it = l.head; it;) { /* ref:572-585 */
        if (it->nkeys == 1) { it->noMore = 1; it = it->next; }
        else if (it->nkeys == 0) it = qlist_erase(&l, it);
        else it = it->next;
    }

    int finish = 0;
    SizeNode* vec = (SizeNode*)malloc(sizeof(SizeNode) * (size_t)(4 * (n + 4)));
    SizeNode* prevvec = (SizeNode*)malloc(sizeof(SizeNode) * (size_t)(4 * (n + 4)));
    int nvec = 0;
    while (!finish) { /* ref:594-739 */
        int prevSize = l.size;
        int nToExpand = 0;
        nvec = 0;
        /* nodes pushed at the front during this pass are not visited by it */
        for (QNode* it = l.head; it;) {
            if (it->noMore) { it = it->next; continue; }
            QNode* c[4];
            divide_node(it, in, c);
            push_children(&l, c, vec, &nvec, &nToExpand);
            it = qlist_erase(&l, it);
        }
        if (l.size >= N || l.size == prevSize) {
            finish = 1;
        } else if (l.size + nToExpand * 3 > N) {
            while (!finish) {
                prevSize = l.size;
                int nprev = nvec;
                memcpy(prevvec, vec, sizeof(SizeNode) * (size_t)nprev);
                nvec = 0;
                qsort(prevvec, (size_t)nprev, sizeof(SizeNode), cmp_sizenode);
                for (int j = nprev - 1; j >= 0; j--) {
                    QNode* c[4];
                    divide_node(prevvec[j].node, in, c);
                    push_children(&l, c, vec, &nvec, NULL);
                    qlist_erase(&l, prevvec[j].node);
                    if (l.size >= N) break;
                }
                if (l.size >= N || l.size == prevSize) finish = 1;
            }
        }
    }
    free(vec); free(prevvec);

    /* ref:742-760 best response per node, first wins ties, list order */
    int m = 0;
    for (QNode* it = l.head; it; it = it->next) {
        int best = it->keys[0];
        for (int k = 1; k < it->nkeys; k++)
            if (in[it->keys[k]].score > in[best].score) best = it->keys[k];
        if (m < cap) out[m] = in[best];
        m++;
    }
    for (QNode* it = l.head; it;) it = qlist_erase(&l, it);
    return m <= cap ? m : -1;
}

/* ------------------------------------------------------------------ ref:1043-1132 operator() */
int orc_extract(const OrcExtractor* ex, const uint8_t* img, int w, int h, int stride,
                OrcKeyPoint* kps, uint8_t* desc, int cap,
                uint8_t* pyr_out, int* cand_counts, int* kept_counts)
{
    if (!img || w <= 0 || h <= 0) return 0; /* ref:1046-1047 */
    const int L = ex->nlevels;
    uint8_t* lv[ORC_MAX_LEVELS];
    int lw[ORC_MAX_LEVELS], lh[ORC_MAX_LEVELS];
    /* ComputePyramid ref:1107-1132 (the 19 px border is never read on this path) */
    for (int l = 0; l < L; l++) {
        orc_level_size(ex, w, h, l, &lw[l], &lh[l]);
        if (lw[l] < 1 || lh[l] < 1) { for (int k = 0; k < l; k++) free(lv[k]); return -3; }
        lv[l] = (uint8_t*)malloc((size_t)lw[l] * lh[l]);
        if (l == 0) for (int y = 0; y < h; y++) memcpy(lv[0] + (size_t)y * w, img + (size_t)y * stride, (size_t)w);
        else orc_resize_linear_u8(lv[l - 1], lw[l - 1], lh[l - 1], lw[l - 1], lv[l], lw[l], lh[l], lw[l]);
    }
    if (pyr_out) {
        size_t o = 0;
        for (int l = 0; l < L; l++) { memcpy(pyr_out + o, lv[l], (size_t)lw[l] * lh[l]); o += (size_t)lw[l] * lh[l]; }
    }

    int total = 0, err = 0;
    for (int l = 0; l < L && !err; l++) {
        const int minB = EDGE_THRESHOLD - 3;
        const int maxBX = lw[l] - EDGE_THRESHOLD + 3, maxBY = lh[l] - EDGE_THRESHOLD + 3;
        int ccap = (lw[l] * lh[l]) / 4 + 16;
        OrcCorner* cand = (OrcCorner*)malloc(sizeof(OrcCorner) * (size_t)ccap);
        OrcCorner* kept = (OrcCorner*)malloc(sizeof(OrcCorner) * (size_t)(ex->nfeatures + 16));
        int nc = orc_level_candidates(ex, lv[l], lw[l], lh[l], lw[l], cand, ccap);
        int nk = 0;
        if (nc < 0) err = -4;
        else if (nc > 0) {
            nk = orc_distribute(cand, nc, minB, maxBX, minB, maxBY, ex->mnFeaturesPerLevel[l],
                                kept, ex->nfeatures + 16);
            if (nk < 0) err = -5;
        }
        if (cand_counts) cand_counts[l] = nc;
        if (kept_counts) kept_counts[l] = nk;
        if (!err && nk > 0) {
            if (total + nk > cap) err = -6;
            else {
                const int scaledPatchSize = (int)((float)PATCH_SIZE * ex->mvScaleFactor[l]); /* ref:837 */
                uint8_t* blur = (uint8_t*)malloc((size_t)lw[l] * lh[l]);
                orc_gaussian7_u8(lv[l], lw[l], lh[l], lw[l], blur, lw[l]); /* ref:1085-1086 */
                for (int i = 0; i < nk; i++) {
                    OrcKeyPoint* kp = &kps[total + i];
                    const int x = kept[i].x + minB, y = kept[i].y + minB; /* ref:843-844 */
                    kp->size = (float)scaledPatchSize;
                    kp->response = (float)kept[i].score;
                    kp->octave = l;
                    kp->class_id = -1;
                    kp->angle = orc_ic_angle(lv[l], lw[l], x, y, ex->umax); /* raw level, ref:852 */
                    orc_brief(blur, lw[l], x, y, kp->angle, desc + (size_t)(total + i) * 32);
                    kp->x = (float)x; kp->y = (float)y;
                    if (l != 0) { /* ref:1095-1101 */
                        const float scale = ex->mvScaleFactor[l];
                        kp->x *= scale; kp->y *= scale;
                    }
                }
                free(blur);
                total += nk;
            }
        }
        free(cand); free(kept);
    }
    for (int l = 0; l < L; l++) free(lv[l]);
    return err ? err : total;
}
