// check_vs_opencv.cpp -- the reference's ORBextractor (unmodified source, real OpenCV) against this repository's CPU
// oracle, stage by stage, on PGM frames.  Build: tools/check_vs_opencv/CMakeLists.txt.  Not built in the development
// image (no OpenCV there); whoever has OpenCV 3.0 runs it once and the word "unpinned" in DESIGN.md section 2 goes away
// -- or the first difference it prints says which restated stage to fix.
//
// Stages, each compared on its own so that a difference is attributed to ONE OpenCV primitive:
//   1 cv::resize(INTER_LINEAR)         every pyramid level from the previous one      ORBextractor.cc:1120  <-> orc_resize_linear_u8
//   2 cv::FAST(thr, nonmax)            on every level, thresholds 20 and 7            :809,814              <-> orc_fast9_16
//   3 cv::GaussianBlur(7x7, 2, 2, 101) on every level                                 :1086                 <-> orc_gaussian7_u8
//   4 cv::fastAtan2                    on a dense grid of (y, x) incl. the IC ranges  :103                  <-> orc_fast_atan2
//   5 ORBextractor::operator()         keypoints (28-byte records) and descriptors    :1043-1105            <-> orc_extract
// Stage 5 can differ where stages 1-4 agree only through the quadtree's tie among equal-sized nodes (sort by
// (size, pointer), :684 -- heap addresses) and through cosf/sinf of another libm; the program says which keypoints differ.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#include <opencv2/imgproc/imgproc.hpp>

#include "ORBextractor.h"   // the reference's header (SingleRobotScenario/include)

extern "C" {
#include "orb_oracle.h"
}

static bool read_pgm(const char* path, cv::Mat& out)
{
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    char magic[3] = {0, 0, 0};
    int w = 0, h = 0, maxv = 0;
    bool ok = std::fscanf(f, "%2s %d %d %d", magic, &w, &h, &maxv) == 4 && std::strcmp(magic, "P5") == 0 && maxv == 255 && w > 0 && h > 0;
    if (ok) {
        std::fgetc(f);  // the single whitespace after the header
        out.create(h, w, CV_8UC1);
        ok = std::fread(out.data, 1, (size_t)w * h, f) == (size_t)w * h;
    }
    std::fclose(f);
    return ok;
}

struct Tally { long compared = 0, differing = 0; void add(long c, long d) { compared += c; differing += d; } };

static long diff_bytes(const cv::Mat& a, const std::vector<uint8_t>& b, int w, int h, const char* what, int level)
{
    long bad = 0;
    int fx = -1, fy = -1;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            if (a.at<uchar>(y, x) != b[(size_t)y * w + x]) { if (!bad) { fx = x; fy = y; } bad++; }
    if (bad) std::printf("    %s level %d: %ld of %d pixels differ, first at (%d, %d): OpenCV %d, oracle %d\n", what, level, bad, w * h, fx, fy,
                         (int)a.at<uchar>(fy, fx), (int)b[(size_t)fy * w + fx]);
    return bad;
}

int main(int argc, char** argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: %s frame.pgm [frame.pgm ...] [--nfeatures N]\n", argv[0]); return 2; }
    int nfeatures = 0;
    std::vector<std::string> files;
    for (int i = 1; i < argc; i++) {
        if (!std::strcmp(argv[i], "--nfeatures") && i + 1 < argc) nfeatures = std::atoi(argv[++i]);
        else files.push_back(argv[i]);
    }
    std::printf("OpenCV %s\n", CV_VERSION);
    Tally tResize, tFast, tBlur, tAtan, tKp, tDesc;

    // ---- stage 4 first: it needs no image
    {
        long bad = 0, n = 0;
        float fy = 0, fx = 0;
        for (int iy = -2000; iy <= 2000; iy += 7)
            for (int ix = -2000; ix <= 2000; ix += 5) {
                // the moments of IC_Angle are integers up to 15 * 255 * 749 in magnitude: a grid of small values plus scaled ones
                for (int s = 0; s < 2; s++) {
                    const float y = (float)(s ? iy * 1431 : iy), x = (float)(s ? ix * 1431 : ix);
                    const float a = cv::fastAtan2(y, x), b = orc_fast_atan2(y, x);
                    n++;
                    if (std::memcmp(&a, &b, 4) != 0) { if (!bad) { fy = y; fx = x; } bad++; }
                }
            }
        if (bad) std::printf("  fastAtan2: %ld of %ld arguments differ, first at (y=%g, x=%g): OpenCV %.9g, oracle %.9g\n", bad, n, fy, fx,
                             cv::fastAtan2(fy, fx), orc_fast_atan2(fy, fx));
        tAtan.add(n, bad);
    }

    for (const std::string& path : files) {
        cv::Mat im;
        if (!read_pgm(path.c_str(), im)) { std::fprintf(stderr, "cannot read %s (binary PGM, maxval 255)\n", path.c_str()); return 2; }
        const int W = im.cols, H = im.rows;
        const int nf = nfeatures ? nfeatures : (W >= 1000 ? 2000 : 1000);
        std::printf("%s: %dx%d, nfeatures %d\n", path.c_str(), W, H, nf);
        OrcExtractor oex;
        orc_extractor_init(&oex, nf, 1.2f, 8, 20, 7);

        // ---- stages 1-3 on the pyramid OpenCV builds (each level from OpenCV's previous level: a resize difference
        // at level l must not be blamed on levels > l)
        cv::Mat prev = im;
        for (int l = 0; l < 8; l++) {
            int lw, lh;
            orc_level_size(&oex, W, H, l, &lw, &lh);
            cv::Mat lvl;
            if (l == 0) lvl = im;
            else {
                cv::resize(prev, lvl, cv::Size(lw, lh), 0, 0, cv::INTER_LINEAR);   // ORBextractor.cc:1120
                std::vector<uint8_t> o((size_t)lw * lh);
                orc_resize_linear_u8(prev.data, prev.cols, prev.rows, (int)prev.step, o.data(), lw, lh, lw);
                tResize.add((long)lw * lh, diff_bytes(lvl, o, lw, lh, "resize", l));
            }
            for (int thr : {20, 7}) {
                std::vector<cv::KeyPoint> kp;
                cv::FAST(lvl, kp, thr, true);                                        // :809,814 (there per 30 px cell; here the whole level)
                std::vector<OrcCorner> oc((size_t)lw * lh / 4 + 16);
                const int on = orc_fast9_16(lvl.data, lw, lh, (int)lvl.step, thr, oc.data(), (int)oc.size());
                long bad = (long)kp.size() != on;
                for (size_t i = 0; i < kp.size() && !bad; i++)
                    bad = (int)kp[i].pt.x != oc[i].x || (int)kp[i].pt.y != oc[i].y || (int)kp[i].response != oc[i].score;
                if (bad) std::printf("    FAST level %d threshold %d: OpenCV %zu corners, oracle %d (or position / score / order differ)\n", l, thr, kp.size(), on);
                tFast.add(1, bad);
            }
            {
                cv::Mat blurred = lvl.clone();
                cv::GaussianBlur(blurred, blurred, cv::Size(7, 7), 2, 2, cv::BORDER_REFLECT_101);   // :1086
                std::vector<uint8_t> o((size_t)lw * lh);
                orc_gaussian7_u8(lvl.data, lw, lh, (int)lvl.step, o.data(), lw);
                tBlur.add((long)lw * lh, diff_bytes(blurred, o, lw, lh, "GaussianBlur", l));
            }
            prev = lvl;
        }

        // ---- stage 5: the reference's class itself
        iORB_SLAM::ORBextractor ref(nf, 1.2f, 8, 20, 7);
        std::vector<cv::KeyPoint> kps;
        cv::Mat desc;
        ref(im, cv::Mat(), kps, desc);
        std::vector<OrcKeyPoint> ok((size_t)nf * 3);
        std::vector<uint8_t> od((size_t)nf * 3 * 32);
        const int on = orc_extract(&oex, im.data, W, H, (int)im.step, ok.data(), od.data(), nf * 3, nullptr, nullptr, nullptr);
        std::printf("  operator(): reference %zu keypoints, oracle %d\n", kps.size(), on);
        long badK = 0, badD = 0;
        const size_t n = std::min(kps.size(), (size_t)(on < 0 ? 0 : on));
        static_assert(sizeof(cv::KeyPoint) == sizeof(OrcKeyPoint), "cv::KeyPoint is the 28-byte record");
        for (size_t i = 0; i < n; i++) {
            if (std::memcmp(&kps[i], &ok[i], sizeof(OrcKeyPoint)) != 0) {
                if (badK < 5) std::printf("    keypoint %zu: reference (%.3f, %.3f, oct %d, angle %.4f, resp %.0f)  oracle (%.3f, %.3f, oct %d, angle %.4f, resp %.0f)\n", i,
                                          kps[i].pt.x, kps[i].pt.y, kps[i].octave, kps[i].angle, kps[i].response, ok[i].x, ok[i].y, ok[i].octave, ok[i].angle, ok[i].response);
                badK++;
            } else if (std::memcmp(desc.ptr(i), &od[i * 32], 32) != 0) badD++;
        }
        badK += (long)std::max(kps.size(), (size_t)(on < 0 ? 0 : on)) - (long)n;
        tKp.add((long)n, badK); tDesc.add((long)n, badD);
        if (badD) std::printf("    %ld descriptors differ on identical keypoints (cosf/sinf of this libm, or the blur)\n", badD);
    }
    std::printf("\nsummary (differing / compared)\n  resize       %ld / %ld pixels\n  FAST         %ld / %ld level passes\n  GaussianBlur %ld / %ld pixels\n"
                "  fastAtan2    %ld / %ld arguments\n  keypoints    %ld / %ld\n  descriptors  %ld / %ld\n",
                tResize.differing, tResize.compared, tFast.differing, tFast.compared, tBlur.differing, tBlur.compared, tAtan.differing, tAtan.compared,
                tKp.differing, tKp.compared, tDesc.differing, tDesc.compared);
    const bool pinned = !(tResize.differing | tFast.differing | tBlur.differing | tAtan.differing | tKp.differing | tDesc.differing);
    std::printf("%s\n", pinned ? "PINNED: the oracle equals the reference with this OpenCV on these frames"
                               : "DIFFERENCES: see above; stages 1-4 name the OpenCV primitive, stage 5 alone points at the quadtree tie-break or libm");
    return pinned ? 0 : 1;
}
