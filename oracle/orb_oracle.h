/*
 * orb_oracle.h -- CPU ORACLE for the ORBSLAMM per-frame hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is a plain-C restatement of the reference algorithm
 *   /root/reference/SingleRobotScenario/src/ORBextractor.cc   (extractor)
 *   /root/reference/SingleRobotScenario/src/ORBmatcher.cc     (matchers)
 *   /root/reference/SingleRobotScenario/src/Frame.cc:230-245,327-392 (grid)
 * Each function cites the file:line it follows.
 *
 * !! PARITY UNPINNED !!  The arithmetic of five extractor stages lives in OpenCV
 * (cv::FAST, cv::resize, cv::GaussianBlur, cv::fastAtan2, cvRound), which is NOT
 * vendored in the reference and NOT installed in this image.  The
 * reference target is OpenCV 3.0.0 (CMake asks for >=3.0, fallback
 * 2.4.3); its published generic C++ algorithm is restated here.
 * The reference holds no tests / golden vectors for this path, and
 * the extractor / matcher / vocabulary sources cannot be built here without writing
 * stand-ins for OpenCV.  oracle/_ref holds the one part that can: DBoW2's BowVector.cpp
 * and FeatureVector.cpp (standard library only), compiled from the reference where they
 * lie (Makefile target _ref) -- orb_vocab.c's BowVector / FeatureVector assembly is held
 * to that object code bit for bit (tests/test_oracle_vs_reference_cpu.py).  Also pinned: the FAST-9 corner predicate (fixture from
 * scikit-image's independent implementation), the BRIEF pattern (sha256), the umax
 * table, the per-level feature split, the matcher arithmetic (fully visible in
 * the reference source) -- see tests/test_oracle_*.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * anything in this directory.  The product (orbslamm_amd/) never links it.
 *
 * Deterministic refinements of reference behaviour that is undefined/UB:
 *  - DistributeOctTree sorts pair<int,ExtractorNode*>; ties on size are broken by
 *    heap pointer (ORBextractor.cc:684).  Oracle: tie -> node creation sequence
 *    number, ascending (so walking from the back expands the latest-created first).
 *  - floating point: strict IEEE binary32, no FMA contraction (-ffp-contract=off).
 *  - F6 (corrects SURVEY.md F6, which read the step as "double libm, cast"): the steering angle's
 *    `(float)cos(angle)`, `(float)sin(angle)` (ORBextractor.cc:113) take a FLOAT argument under the file-scope
 *    `using namespace std;` (:65), so they are std::cos(float) = cosf / sinf.  orc_brief calls the HOST libm's
 *    cosf/sinf: this step is libm-version dependent in the reference itself (glibc >= 2.28 rounds one binary64
 *    polynomial; older glibc and other libms may differ in the last bit for some angles).  The product restates
 *    glibc's algorithm (orbslamm_amd/csrc/orbx_sincosf.h); tests/cpp/sincos_check.c compares the two over every
 *    binary32 angle in [0, 2pi].
 *  - levels whose FAST window is narrower/lower than 30 px (nCols or nRows == 0;
 *    float division by zero in the reference, :785-786) yield no keypoints.
 */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_LEVELS 16

/* layout-identical to cv::KeyPoint as filled by the reference (28 bytes) */
typedef struct {
    float x, y;      /* pt */
    float size;
    float angle;
    float response;
    int32_t octave;
    int32_t class_id;
} OrcKeyPoint;

typedef struct {
    int nfeatures;
    double scaleFactor;           /* member is double, initialised from a float (ORBextractor.h:93) */
    int nlevels;
    int iniThFAST, minThFAST;
    float mvScaleFactor[ORC_MAX_LEVELS];
    float mvInvScaleFactor[ORC_MAX_LEVELS];
    float mvLevelSigma2[ORC_MAX_LEVELS];
    float mvInvLevelSigma2[ORC_MAX_LEVELS];
    int mnFeaturesPerLevel[ORC_MAX_LEVELS];
    int umax[16];
} OrcExtractor;

/* FAST candidate in level (window-relative) coordinates, before distribution */
typedef struct {
    int32_t x, y;
    int32_t score;
} OrcCorner;

/* ---- extractor ---- */
int  orc_extractor_init(OrcExtractor* ex, int nfeatures, float scaleFactor, int nlevels,
                        int iniThFAST, int minThFAST);
void orc_level_size(const OrcExtractor* ex, int w, int h, int level, int* lw, int* lh);

void orc_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride,
                          uint8_t* dst, int dw, int dh, int dstride);
int  orc_fast9_16(const uint8_t* img, int w, int h, int stride, int threshold,
                  OrcCorner* out, int cap);
void orc_fast_corner_mask(const uint8_t* img, int w, int h, int stride, int threshold, uint8_t* mask);
void orc_gaussian7_u8(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride);
float orc_fast_atan2(float y, float x);
float orc_ic_angle(const uint8_t* img, int stride, int x, int y, const int* umax);
void orc_brief(const uint8_t* blurred, int stride, int x, int y, float angle_deg, uint8_t desc[32]);

/* candidates of one level, in reference order (cells row-major, inside a cell row-major);
 * coordinates are window-relative (minBorder not added).  returns count (<= cap) */
int  orc_level_candidates(const OrcExtractor* ex, const uint8_t* img, int w, int h, int stride,
                          OrcCorner* out, int cap);
/* cells visited / cells run again at minThFAST / cells empty at both thresholds (ref:808-816) */
void orc_level_cell_stats(const OrcExtractor* ex, const uint8_t* img, int w, int h, int stride, int* ncells, int* nretry, int* nempty);
/* DistributeOctTree: in = candidates (window-relative), out = kept, list order */
int  orc_distribute(const OrcCorner* in, int n, int minX, int maxX, int minY, int maxY,
                    int N, OrcCorner* out, int cap);

/* test hook: reverse the order among equal-sized quadtree nodes (the reference's order there is a heap-address accident) */
void orc_set_tie_break_reversed(int on);

/* whole operator(): returns number of keypoints (or <0 on error).  desc is N x 32.
 * If pyr_out != NULL it receives the concatenated (tight, stride = level width)
 * pyramid levels 0..nlevels-1; if cand_counts != NULL it receives the per-level
 * number of FAST candidates; kept_counts the per-level kept keypoints. */
int  orc_extract(const OrcExtractor* ex, const uint8_t* img, int w, int h, int stride,
                 OrcKeyPoint* kps, uint8_t* desc, int cap,
                 uint8_t* pyr_out, int* cand_counts, int* kept_counts);

/* ---- matcher ---- */
int  orc_descriptor_distance(const uint8_t a[32], const uint8_t b[32]);
void orc_three_maxima(const int* hist_sizes, int L, int* ind1, int* ind2, int* ind3);
int  orc_rot_bin(float angle_q, float angle_t);

/* headline brute force "match vs previous frame" (SURVEY.md 8d): for every query
 * (current frame) best/second over ALL train descriptors in index order, accept
 * best<=th_low && (float)best < nnratio*(float)second, rotation histogram + top-3
 * pruning.  match[q] = train index or -1.  returns number of matches. */
int  orc_match_bruteforce(const uint8_t* qdesc, const float* qangle, int nq,
                          const uint8_t* tdesc, const float* tangle, int nt,
                          float nnratio, int th_low, int check_ori, int32_t* match);

/* Frame grid (Frame.cc:230-245, 382-392) as CSR: cell = ix*rows+iy */
typedef struct {
    float minX, minY, invW, invH;  /* mnMinX, mnMinY, mfGridElementWidthInv, mfGridElementHeightInv */
    int cols, rows;                /* 64, 48 */
} OrcGridParams;
void orc_grid_build(const OrcGridParams* gp, const OrcKeyPoint* keys_un, int n,
                    int32_t* cell_start /* cols*rows+1 */, int32_t* cell_idx /* n */);
int  orc_features_in_area(const OrcGridParams* gp, const OrcKeyPoint* keys_un,
                          const int32_t* cell_start, const int32_t* cell_idx,
                          float x, float y, float r, int minLevel, int maxLevel,
                          int32_t* out, int cap);

/* SearchByBoW, flattened (ORBmatcher.cc:159-290 when out_by_train=1, :524-657 when 0).
 * Feature vectors are CSR: node ids ascending, per-node index lists in stored order. */
typedef struct {
    int n_nodes;
    const uint32_t* node_id;   /* ascending */
    const int32_t* start;      /* n_nodes+1 */
    const int32_t* idx;        /* feature indices */
} OrcFeatVec;
int  orc_search_by_bow(const uint8_t* qdesc, const float* qangle, const uint8_t* qvalid, int nq,
                       const OrcFeatVec* qfv,
                       const uint8_t* tdesc, const float* tangle, const uint8_t* tvalid, int nt,
                       const OrcFeatVec* tfv,
                       float nnratio, int check_ori, int out_by_train,
                       int32_t* match /* nt if out_by_train else nq */);

/* SearchByProjection family, flattened (projection itself stays in the caller).
 * mode 3: ORBmatcher.cc:45-129   (Frame, vector<MapPoint*>)  best+second w/ levels
 * mode 4: ORBmatcher.cc:1330-1472 (Cur, Last)                best only, rot-hist
 * mode 5: ORBmatcher.cc:1474-1601 (Cur, KF, set)             best<=ORBdist, rot-hist
 * mode 6: ORBmatcher.cc:292-405   (KF, Scw)                  best<=TH_LOW
 * per query: u,v,r,minLevel,maxLevel, valid, obs_pos (MapPoint::Observations()>0).
 * t_occ in: initial "skip" flag per train feature.  assign[t] (in/out): query index
 * now held by train feature t, -1 if none / unchanged initial.  returns nmatches. */
typedef struct {
    int mode;
    float nnratio;
    int check_ori;
    int th_dist;   /* TH_HIGH=100 (3,4), ORBdist (5), TH_LOW=50 (6) */
} OrcProjParams;
int  orc_search_by_projection(const OrcProjParams* pp,
                              const float* q_uvr /* nq x 3 */, const int8_t* q_lvl /* nq x 2 */,
                              const uint8_t* qdesc, const float* qangle,
                              const uint8_t* qvalid, const uint8_t* q_obs_pos, int nq,
                              const OrcGridParams* gp, const OrcKeyPoint* t_keys_un,
                              const int32_t* cell_start, const int32_t* cell_idx,
                              const uint8_t* tdesc, int nt,
                              uint8_t* t_occ, int32_t* assign);

int  orc_search_by_projection_stereo(const OrcProjParams* pp,
                              const float* q_uvr, const float* q_ur, const int8_t* q_lvl,
                              const uint8_t* qdesc, const float* qangle,
                              const uint8_t* qvalid, const uint8_t* q_obs_pos, int nq,
                              const OrcGridParams* gp, const OrcKeyPoint* t_keys_un,
                              const int32_t* cell_start, const int32_t* cell_idx,
                              const uint8_t* tdesc, const float* t_uright, int nt,
                              uint8_t* t_occ, int32_t* assign);

/* ---- the remaining matchers on the same primitive ---- */

/* Independent windowed best search shared by Fuse (ORBmatcher.cc:827-975, chi2=1),
 * Fuse(KF,Scw,...) (:977-1102, chi2=0) and both passes of SearchBySim3 (:1104-1328, chi2=0):
 * candidates = KeyFrame::GetFeaturesInArea(u,v,radius) (no level filter), keep octave in
 * [pred-1, pred], optional reprojection chi-square test, best = strict <, first wins.
 * best_idx[q] = -1 when no candidate survives (best_dist then 256). */
int  orc_window_best(const float* q_uvr, const float* q_ur, const int8_t* q_pred,
                     const uint8_t* qdesc, const uint8_t* qvalid, int nq,
                     const OrcGridParams* gp, const OrcKeyPoint* tk,
                     const int32_t* cell_start, const int32_t* cell_idx,
                     const uint8_t* tdesc, const float* t_uright, int nt,
                     const float* inv_sigma2, int chi2,
                     int32_t* best_idx, int32_t* best_dist);

/* SearchForInitialization, ORBmatcher.cc:407-522.  q_xy = vbPrevMatched; only octave-0
 * queries search; a better later match steals the train feature (:466-470). */
int  orc_search_for_initialization(const float* q_xy, float window, const OrcKeyPoint* qk,
                                   const uint8_t* qdesc, int nq,
                                   const OrcGridParams* gp, const OrcKeyPoint* tk,
                                   const int32_t* cell_start, const int32_t* cell_idx,
                                   const uint8_t* tdesc, int nt,
                                   float nnratio, int check_ori, int32_t* matches12);

/* SearchForTriangulation, ORBmatcher.cc:659-825.  skip1/skip2: feature already has a
 * MapPoint; uright = mvuRight (NULL for mono); F12 row-major 3x3; (ex,ey) epipole in
 * image 2; sf2/sigma2_2 = pKF2->mvScaleFactors / mvLevelSigma2. */
int  orc_search_for_triangulation(const OrcKeyPoint* k1, const uint8_t* d1, const uint8_t* skip1,
                                  const float* uright1, int n1, const OrcFeatVec* fv1,
                                  const OrcKeyPoint* k2, const uint8_t* d2, const uint8_t* skip2,
                                  const float* uright2, int n2, const OrcFeatVec* fv2,
                                  const float F12[9], float ex, float ey,
                                  const float* sf2, const float* sigma2_2,
                                  int only_stereo, int check_ori, int32_t* matches12);

/* SURVEY.md 8(f) rank 1, last item: Frame::ComputeStereoMatches (src/Frame.cc:466-638), orb_stereo.c.
 * pyramids: per level a tight or strided 8-bit image (the un-blurred levels of ComputePyramid, no border needed:
 * the 11 x 21 search window stays inside the level for keypoints >= 16 px from its edge).
 * returns the number of accepted matches before the median filter. */
typedef struct {
    int nlevels;
    const uint8_t* data[ORC_MAX_LEVELS];
    int w[ORC_MAX_LEVELS], h[ORC_MAX_LEVELS], stride[ORC_MAX_LEVELS];
} OrcPyramid;
int  orc_compute_stereo_matches(const OrcKeyPoint* keysL, const uint8_t* descL, int N,
                                const OrcKeyPoint* keysR, const uint8_t* descR, int Nr,
                                const OrcPyramid* pyrL, const OrcPyramid* pyrR,
                                const float* mvScaleFactors, const float* mvInvScaleFactors,
                                float mb, float mbf, float* mvuRight, float* mvDepth);

/* Frame::UndistortKeyPoints (src/Frame.cc:404-434), OpenCV 3.0 undistortPoints (unpinned) */
void orc_stereo_from_rgbd(const OrcKeyPoint* keys, const OrcKeyPoint* keysUn, int n, const float* depth, int w, int h, int stride,
                          float mbf, float* mvuRight, float* mvDepth);
void orc_undistort_keypoints(const OrcKeyPoint* in, int n, const float K[4], const float D[5], OrcKeyPoint* out);

/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:242-307), batched */
int  orc_distinctive_descriptors(const uint8_t* desc, const int32_t* start, int npoints, int32_t* best_idx);

/* ---- vocabulary-tree descent (DBoW2 transform), orb_vocab.c ---- */
typedef struct OrcVocab OrcVocab;
/* nodes in loadFromTextFile order: line i becomes node id i+1 (root = 0); parent ids refer to
 * that numbering; leaves get word ids in file order. */
OrcVocab* orc_vocab_create(int k, int L, int scoring, int weighting, int n, const int32_t* parent,
                           const uint8_t* is_leaf, const uint8_t* desc, const double* weight);
void orc_vocab_free(OrcVocab* v);
void orc_vocab_transform_one(const OrcVocab* v, const uint8_t* f, int levelsup, uint32_t* word, double* w, uint32_t* nid);
/* BowVector as (word_id ascending, weight), FeatureVector as CSR (node ascending, feature
 * indices ascending).  Capacities: n entries each, fv_start n+1. */
int  orc_vocab_transform(const OrcVocab* v, const uint8_t* desc, int n, int levelsup,
                         uint32_t* word_id, double* word_w, int* n_words,
                         uint32_t* fv_node, int32_t* fv_start, int32_t* fv_idx, int* n_fv);

#ifdef __cplusplus
}
#endif
#endif
