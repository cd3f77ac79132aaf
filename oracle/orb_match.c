/*
 * orb_match.c -- CPU ORACLE (test infrastructure only), matcher half.
 *
 * Restates /root/reference/SingleRobotScenario/src/ORBmatcher.cc ("ref:LINE") and
 * the Frame grid of src/Frame.cc ("frame:LINE").  The matcher arithmetic is fully
 * visible in the reference source (no third-party arithmetic on the Hamming path),
 * so this half is a direct restatement; the object-graph walking (MapPoint flags,
 * projections) is flattened into arrays by the caller.
 */
#include "orb_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

enum { TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30 }; /* ref:37-39 */

/* ref:1649-1665: 8 x 32-bit SWAR popcount of the xor */
int orc_descriptor_distance(const uint8_t a[32], const uint8_t b[32])
{
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb;
        memcpy(&pa, a + 4 * i, 4);
        memcpy(&pb, b + 4 * i, 4);
        uint32_t v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555u);
        v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
        dist += (int)((((v + (v >> 4)) & 0xF0F0F0Fu) * 0x1010101u) >> 24);
    }
    return dist;
}

/* ref:1603-1644 */
void orc_three_maxima(const int* histo, int L, int* ind1, int* ind2, int* ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    *ind1 = *ind2 = *ind3 = -1;
    for (int i = 0; i < L; i++) {
        const int s = histo[i];
        if (s > max1) {
            max3 = max2; max2 = max1; max1 = s;
            *ind3 = *ind2; *ind2 = *ind1; *ind1 = i;
        } else if (s > max2) {
            max3 = max2; max2 = s;
            *ind3 = *ind2; *ind2 = i;
        } else if (s > max3) {
            max3 = s; *ind3 = i;
        }
    }
    if ((float)max2 < 0.1f * (float)max1) { *ind2 = -1; *ind3 = -1; }
    else if ((float)max3 < 0.1f * (float)max1) { *ind3 = -1; }
}

/* ref:238-245 etc.: factor = 1.0f/HISTO_LENGTH (sic), C round() */
int orc_rot_bin(float angle_q, float angle_t)
{
    const float factor = 1.0f / HISTO_LENGTH;
    float rot = angle_q - angle_t;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)round((double)(rot * factor));
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}

/* rotation histogram as 30 growable index lists */
typedef struct { int* v[HISTO_LENGTH]; int n[HISTO_LENGTH]; int cap[HISTO_LENGTH]; } RotHist;
static void rh_init(RotHist* h) { memset(h, 0, sizeof(*h)); }
static void rh_push(RotHist* h, int bin, int val)
{
    if (h->n[bin] == h->cap[bin]) {
        h->cap[bin] = h->cap[bin] ? 2 * h->cap[bin] : 64;
        h->v[bin] = (int*)realloc(h->v[bin], sizeof(int) * (size_t)h->cap[bin]);
    }
    h->v[bin][h->n[bin]++] = val;
}
static void rh_free(RotHist* h) { for (int i = 0; i < HISTO_LENGTH; i++) free(h->v[i]); }
/* null everything outside the three dominant bins (ref:269-287); returns #removed pushes */
static int rh_prune(RotHist* h, int32_t* arr)
{
    int i1, i2, i3, removed = 0;
    orc_three_maxima(h->n, HISTO_LENGTH, &i1, &i2, &i3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
        if (i == i1 || i == i2 || i == i3) continue;
        for (int j = 0; j < h->n[i]; j++) { arr[h->v[i][j]] = -1; removed++; }
    }
    return removed;
}

int orc_match_bruteforce(const uint8_t* qdesc, const float* qangle, int nq,
                         const uint8_t* tdesc, const float* tangle, int nt,
                         float nnratio, int th_low, int check_ori, int32_t* match)
{
    RotHist rh; rh_init(&rh);
    int nmatches = 0;
    for (int q = 0; q < nq; q++) {
        match[q] = -1;
        int best1 = 256, best2 = 256, bestIdx = -1;
        for (int t = 0; t < nt; t++) {
            const int dist = orc_descriptor_distance(qdesc + 32 * (size_t)q, tdesc + 32 * (size_t)t);
            if (dist < best1) { best2 = best1; best1 = dist; bestIdx = t; }
            else if (dist < best2) best2 = dist;
        }
        if (best1 <= th_low && (float)best1 < nnratio * (float)best2) {
            match[q] = bestIdx;
            if (check_ori) rh_push(&rh, orc_rot_bin(qangle[q], tangle[bestIdx]), q);
            nmatches++;
        }
    }
    if (check_ori) nmatches -= rh_prune(&rh, match);
    rh_free(&rh);
    return nmatches;
}

/* ------------------------------------------------------------------ Frame grid */
/* frame:382-392 PosInGrid uses round(); keypoints falling on col==cols / row==rows are dropped */
static int pos_in_grid(const OrcGridParams* gp, float x, float y, int* px, int* py)
{
    *px = (int)round((double)((x - gp->minX) * gp->invW));
    *py = (int)round((double)((y - gp->minY) * gp->invH));
    if (*px < 0 || *px >= gp->cols || *py < 0 || *py >= gp->rows) return 0;
    return 1;
}

/* frame:230-245 */
void orc_grid_build(const OrcGridParams* gp, const OrcKeyPoint* k, int n,
                    int32_t* cell_start, int32_t* cell_idx)
{
    const int nc = gp->cols * gp->rows;
    int32_t* cnt = (int32_t*)calloc((size_t)nc + 1, sizeof(int32_t));
    for (int i = 0; i < n; i++) {
        int px, py;
        if (pos_in_grid(gp, k[i].x, k[i].y, &px, &py)) cnt[px * gp->rows + py]++;
    }
    cell_start[0] = 0;
    for (int c = 0; c < nc; c++) cell_start[c + 1] = cell_start[c] + cnt[c];
    memset(cnt, 0, sizeof(int32_t) * (size_t)nc);
    for (int i = 0; i < n; i++) {
        int px, py;
        if (pos_in_grid(gp, k[i].x, k[i].y, &px, &py)) {
            const int c = px * gp->rows + py;
            cell_idx[cell_start[c] + cnt[c]++] = i;
        }
    }
    free(cnt);
}

/* frame:327-380 (KeyFrame.cc:618-657 is the minLevel=-1,maxLevel=-1 case) */
int orc_features_in_area(const OrcGridParams* gp, const OrcKeyPoint* k,
                         const int32_t* cell_start, const int32_t* cell_idx,
                         float x, float y, float r, int minLevel, int maxLevel,
                         int32_t* out, int cap)
{
    int n = 0;
    int nMinCellX = (int)floor((double)((x - gp->minX - r) * gp->invW));
    if (nMinCellX < 0) nMinCellX = 0;
    if (nMinCellX >= gp->cols) return 0;
    int nMaxCellX = (int)ceil((double)((x - gp->minX + r) * gp->invW));
    if (nMaxCellX > gp->cols - 1) nMaxCellX = gp->cols - 1;
    if (nMaxCellX < 0) return 0;
    int nMinCellY = (int)floor((double)((y - gp->minY - r) * gp->invH));
    if (nMinCellY < 0) nMinCellY = 0;
    if (nMinCellY >= gp->rows) return 0;
    int nMaxCellY = (int)ceil((double)((y - gp->minY + r) * gp->invH));
    if (nMaxCellY > gp->rows - 1) nMaxCellY = gp->rows - 1;
    if (nMaxCellY < 0) return 0;

    const int bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
            const int c = ix * gp->rows + iy;
            for (int j = cell_start[c]; j < cell_start[c + 1]; j++) {
                const OrcKeyPoint* kp = &k[cell_idx[j]];
                if (bCheckLevels) {
                    if (kp->octave < minLevel) continue;
                    if (maxLevel >= 0 && kp->octave > maxLevel) continue;
                }
                const float distx = kp->x - x, disty = kp->y - y;
                if (fabsf(distx) < r && fabsf(disty) < r) {
                    if (n < cap) out[n] = cell_idx[j];
                    n++;
                }
            }
        }
    return n < cap ? n : cap;
}

/* ------------------------------------------------------------------ SearchByBoW, ref:159-290 / 524-657 */
int orc_search_by_bow(const uint8_t* qdesc, const float* qangle, const uint8_t* qvalid, int nq,
                      const OrcFeatVec* qfv,
                      const uint8_t* tdesc, const float* tangle, const uint8_t* tvalid, int nt,
                      const OrcFeatVec* tfv,
                      float nnratio, int check_ori, int out_by_train, int32_t* match)
{
    const int nout = out_by_train ? nt : nq;
    for (int i = 0; i < nout; i++) match[i] = -1;
    uint8_t* matched_t = (uint8_t*)calloc((size_t)nt + 1, 1); /* vpMapPointMatches[t]!=NULL / vbMatched2 */
    RotHist rh; rh_init(&rh);
    int nmatches = 0;

    int a = 0, b = 0; /* lock-step walk of the two sorted node lists (ref:180-266) */
    while (a < qfv->n_nodes && b < tfv->n_nodes) {
        if (qfv->node_id[a] == tfv->node_id[b]) {
            for (int iq = qfv->start[a]; iq < qfv->start[a + 1]; iq++) {
                const int q = qfv->idx[iq];
                if (qvalid && !qvalid[q]) continue; /* !pMP || pMP->isBad() */
                int best1 = 256, best2 = 256, bestIdx = -1;
                for (int it = tfv->start[b]; it < tfv->start[b + 1]; it++) {
                    const int t = tfv->idx[it];
                    if (matched_t[t]) continue;
                    if (tvalid && !tvalid[t]) continue; /* ref:583 (!pMP2), KF-KF only */
                    const int dist = orc_descriptor_distance(qdesc + 32 * (size_t)q, tdesc + 32 * (size_t)t);
                    if (dist < best1) { best2 = best1; best1 = dist; bestIdx = t; }
                    else if (dist < best2) best2 = dist;
                }
                if (best1 <= TH_LOW && (float)best1 < nnratio * (float)best2) {
                    matched_t[bestIdx] = 1;
                    if (out_by_train) match[bestIdx] = q; else match[q] = bestIdx;
                    if (check_ori)
                        rh_push(&rh, orc_rot_bin(qangle[q], tangle[bestIdx]), out_by_train ? bestIdx : q);
                    nmatches++;
                }
            }
            a++; b++;
        } else if (qfv->node_id[a] < tfv->node_id[b]) {
            while (a < qfv->n_nodes && qfv->node_id[a] < tfv->node_id[b]) a++; /* lower_bound */
        } else {
            while (b < tfv->n_nodes && tfv->node_id[b] < qfv->node_id[a]) b++;
        }
    }
    if (check_ori) nmatches -= rh_prune(&rh, match);
    rh_free(&rh);
    free(matched_t);
    return nmatches;
}

/* ------------------------------------------------------------------ SearchByProjection family */
int orc_search_by_projection(const OrcProjParams* pp,
                             const float* q_uvr, const int8_t* q_lvl,
                             const uint8_t* qdesc, const float* qangle,
                             const uint8_t* qvalid, const uint8_t* q_obs_pos, int nq,
                             const OrcGridParams* gp, const OrcKeyPoint* tk,
                             const int32_t* cell_start, const int32_t* cell_idx,
                             const uint8_t* tdesc, int nt,
                             uint8_t* t_occ, int32_t* assign)
{
    return orc_search_by_projection_stereo(pp, q_uvr, NULL, q_lvl, qdesc, qangle, qvalid, q_obs_pos, nq, gp, tk, cell_start,
                                           cell_idx, tdesc, NULL, nt, t_occ, assign);
}

/* with the stereo gate of ref:91-96 (mode 3: er = |mTrackProjXR - mvuRight|, er > r*scale -> skip) and ref:1409-1415
 * (mode 4: ur = u - mbf*invzc, er > radius -> skip); q_ur / t_uright NULL = mono */
int orc_search_by_projection_stereo(const OrcProjParams* pp,
                             const float* q_uvr, const float* q_ur, const int8_t* q_lvl,
                             const uint8_t* qdesc, const float* qangle,
                             const uint8_t* qvalid, const uint8_t* q_obs_pos, int nq,
                             const OrcGridParams* gp, const OrcKeyPoint* tk,
                             const int32_t* cell_start, const int32_t* cell_idx,
                             const uint8_t* tdesc, const float* t_uright, int nt,
                             uint8_t* t_occ, int32_t* assign)
{
    RotHist rh; rh_init(&rh);
    int32_t* cand = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nt + 1));
    int nmatches = 0;
    const int use_rot = pp->check_ori && (pp->mode == 4 || pp->mode == 5);
    for (int q = 0; q < nq; q++) {
        if (qvalid && !qvalid[q]) continue;
        const float u = q_uvr[3 * q], v = q_uvr[3 * q + 1], r = q_uvr[3 * q + 2];
        const int nc = orc_features_in_area(gp, tk, cell_start, cell_idx, u, v, r,
                                            q_lvl[2 * q], q_lvl[2 * q + 1], cand, nt);
        if (nc == 0) continue;
        int best = 256, best2 = 256, bestLevel = -1, bestLevel2 = -1, bestIdx = -1;
        for (int c = 0; c < nc; c++) {
            const int t = cand[c];
            if (t_occ[t]) continue;
            if (t_uright && t_uright[t] > 0) {
                const float er = fabsf(q_ur[q] - t_uright[t]);
                if (er > r) continue;
            }
            const int dist = orc_descriptor_distance(qdesc + 32 * (size_t)q, tdesc + 32 * (size_t)t);
            if (dist < best) {
                best2 = best; best = dist;
                bestLevel2 = bestLevel; bestLevel = tk[t].octave;
                bestIdx = t;
            } else if (dist < best2) {
                bestLevel2 = tk[t].octave; best2 = dist;
            }
        }
        if (best <= pp->th_dist) {
            if (pp->mode == 3 && bestLevel == bestLevel2 && (float)best > pp->nnratio * (float)best2)
                continue; /* ref:120-121 */
            assign[bestIdx] = q;
            /* modes 3/4: a later query may take the feature again unless this MapPoint
             * has observations (ref:87-89, 1405-1407); modes 5/6: any non-null blocks */
            if (pp->mode == 3 || pp->mode == 4) { if (!q_obs_pos || q_obs_pos[q]) t_occ[bestIdx] = 1; }
            else t_occ[bestIdx] = 1;
            nmatches++;
            if (use_rot) rh_push(&rh, orc_rot_bin(qangle[q], tk[bestIdx].angle), bestIdx);
        }
    }
    if (use_rot) nmatches -= rh_prune(&rh, assign);
    rh_free(&rh);
    free(cand);
    return nmatches;
}

/* ------------------------------------------------------------------ SURVEY 8(f).1 */
/* Fuse :908-946 / :1053-1079, SearchBySim3 :1196-1222 / :1276-1302 */
int orc_window_best(const float* q_uvr, const float* q_ur, const int8_t* q_pred,
                    const uint8_t* qdesc, const uint8_t* qvalid, int nq,
                    const OrcGridParams* gp, const OrcKeyPoint* tk,
                    const int32_t* cell_start, const int32_t* cell_idx,
                    const uint8_t* tdesc, const float* t_uright, int nt,
                    const float* inv_sigma2, int chi2,
                    int32_t* best_idx, int32_t* best_dist)
{
    int32_t* cand = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nt + 1));
    for (int q = 0; q < nq; q++) {
        best_idx[q] = -1;
        best_dist[q] = 256;
        if (qvalid && !qvalid[q]) continue;
        const float u = q_uvr[3 * q], v = q_uvr[3 * q + 1], radius = q_uvr[3 * q + 2];
        const int pred = q_pred[q];
        const int nc = orc_features_in_area(gp, tk, cell_start, cell_idx, u, v, radius, -1, -1, cand, nt);
        int bestDist = 256, bestIdx = -1;
        for (int c = 0; c < nc; c++) {
            const int idx = cand[c];
            const int kpLevel = tk[idx].octave;
            if (kpLevel < pred - 1 || kpLevel > pred) continue;
            if (chi2) {
                const float ex = u - tk[idx].x, ey = v - tk[idx].y;
                if (t_uright && t_uright[idx] >= 0) {
                    const float er = q_ur[q] - t_uright[idx];
                    const float e2 = ex * ex + ey * ey + er * er;
                    if ((double)(e2 * inv_sigma2[kpLevel]) > 7.8) continue;
                } else {
                    const float e2 = ex * ex + ey * ey;
                    if ((double)(e2 * inv_sigma2[kpLevel]) > 5.99) continue;
                }
            }
            const int dist = orc_descriptor_distance(qdesc + 32 * (size_t)q, tdesc + 32 * (size_t)idx);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        best_idx[q] = bestIdx;
        best_dist[q] = bestDist;
    }
    free(cand);
    return 0;
}

/* :407-522 */
int orc_search_for_initialization(const float* q_xy, float window, const OrcKeyPoint* qk,
                                  const uint8_t* qdesc, int nq,
                                  const OrcGridParams* gp, const OrcKeyPoint* tk,
                                  const int32_t* cell_start, const int32_t* cell_idx,
                                  const uint8_t* tdesc, int nt,
                                  float nnratio, int check_ori, int32_t* m12)
{
    int nmatches = 0;
    RotHist rh; rh_init(&rh);
    int32_t* cand = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nt + 1));
    int* matchedDist = (int*)malloc(sizeof(int) * (size_t)(nt + 1));
    int* m21 = (int*)malloc(sizeof(int) * (size_t)(nt + 1));
    for (int i = 0; i < nt; i++) { matchedDist[i] = 0x7FFFFFFF; m21[i] = -1; }
    for (int i = 0; i < nq; i++) m12[i] = -1;
    for (int i1 = 0; i1 < nq; i1++) {
        const int level1 = qk[i1].octave;
        if (level1 > 0) continue;
        const int nc = orc_features_in_area(gp, tk, cell_start, cell_idx, q_xy[2 * i1], q_xy[2 * i1 + 1],
                                            window, level1, level1, cand, nt);
        if (nc == 0) continue;
        int bestDist = 0x7FFFFFFF, bestDist2 = 0x7FFFFFFF, bestIdx2 = -1;
        for (int c = 0; c < nc; c++) {
            const int i2 = cand[c];
            const int dist = orc_descriptor_distance(qdesc + 32 * (size_t)i1, tdesc + 32 * (size_t)i2);
            if (matchedDist[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= TH_LOW) {
            if ((float)bestDist < (float)bestDist2 * nnratio) {
                if (m21[bestIdx2] >= 0) { m12[m21[bestIdx2]] = -1; nmatches--; }
                m12[i1] = bestIdx2;
                m21[bestIdx2] = i1;
                matchedDist[bestIdx2] = bestDist;
                nmatches++;
                if (check_ori) rh_push(&rh, orc_rot_bin(qk[i1].angle, tk[bestIdx2].angle), i1);
            }
        }
    }
    if (check_ori) {
        int i1, i2, i3;
        orc_three_maxima(rh.n, HISTO_LENGTH, &i1, &i2, &i3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == i1 || i == i2 || i == i3) continue;
            for (int j = 0; j < rh.n[i]; j++) {
                const int idx1 = rh.v[i][j];
                if (m12[idx1] >= 0) { m12[idx1] = -1; nmatches--; }
            }
        }
    }
    rh_free(&rh);
    free(cand); free(matchedDist); free(m21);
    return nmatches;
}

/* :141-157 CheckDistEpipolarLine */
static int check_dist_epipolar(const OrcKeyPoint* kp1, const OrcKeyPoint* kp2, const float* F, const float* sigma2_2)
{
    const float a = kp1->x * F[0] + kp1->y * F[3] + F[6];
    const float b = kp1->x * F[1] + kp1->y * F[4] + F[7];
    const float c = kp1->x * F[2] + kp1->y * F[5] + F[8];
    const float num = a * kp2->x + b * kp2->y + c;
    const float den = a * a + b * b;
    if (den == 0) return 0;
    const float dsqr = num * num / den;
    return (double)dsqr < 3.84 * (double)sigma2_2[kp2->octave];
}

/* :659-825.  vbMatched2 is never set in the reference, so queries are independent. */
int orc_search_for_triangulation(const OrcKeyPoint* k1, const uint8_t* d1, const uint8_t* skip1,
                                 const float* uright1, int n1, const OrcFeatVec* fv1,
                                 const OrcKeyPoint* k2, const uint8_t* d2, const uint8_t* skip2,
                                 const float* uright2, int n2, const OrcFeatVec* fv2,
                                 const float F12[9], float ex, float ey,
                                 const float* sf2, const float* sigma2_2,
                                 int only_stereo, int check_ori, int32_t* m12)
{
    int nmatches = 0;
    RotHist rh; rh_init(&rh);
    for (int i = 0; i < n1; i++) m12[i] = -1;
    int a = 0, b = 0;
    while (a < fv1->n_nodes && b < fv2->n_nodes) {
        if (fv1->node_id[a] == fv2->node_id[b]) {
            for (int i1 = fv1->start[a]; i1 < fv1->start[a + 1]; i1++) {
                const int idx1 = fv1->idx[i1];
                if (skip1 && skip1[idx1]) continue;
                const int bStereo1 = uright1 && uright1[idx1] >= 0;
                if (only_stereo && !bStereo1) continue;
                int bestDist = TH_LOW, bestIdx2 = -1;
                for (int i2 = fv2->start[b]; i2 < fv2->start[b + 1]; i2++) {
                    const int idx2 = fv2->idx[i2];
                    if (skip2 && skip2[idx2]) continue;
                    const int bStereo2 = uright2 && uright2[idx2] >= 0;
                    if (only_stereo && !bStereo2) continue;
                    const int dist = orc_descriptor_distance(d1 + 32 * (size_t)idx1, d2 + 32 * (size_t)idx2);
                    if (dist > TH_LOW || dist > bestDist) continue;
                    if (!bStereo1 && !bStereo2) {
                        const float distex = ex - k2[idx2].x, distey = ey - k2[idx2].y;
                        if (distex * distex + distey * distey < 100 * sf2[k2[idx2].octave]) continue;
                    }
                    if (check_dist_epipolar(&k1[idx1], &k2[idx2], F12, sigma2_2)) { bestIdx2 = idx2; bestDist = dist; }
                }
                if (bestIdx2 >= 0) {
                    m12[idx1] = bestIdx2;
                    nmatches++;
                    if (check_ori) rh_push(&rh, orc_rot_bin(k1[idx1].angle, k2[bestIdx2].angle), idx1);
                }
            }
            a++; b++;
        } else if (fv1->node_id[a] < fv2->node_id[b]) {
            while (a < fv1->n_nodes && fv1->node_id[a] < fv2->node_id[b]) a++;
        } else {
            while (b < fv2->n_nodes && fv2->node_id[b] < fv1->node_id[a]) b++;
        }
    }
    if (check_ori) nmatches -= rh_prune(&rh, m12);
    rh_free(&rh);
    return nmatches;
}

/* ------------------------------------------------------------------ SURVEY 8(f).4
 * MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:242-307), batched over map points:
 * all-pairs distances, per row the element [0.5*(N-1)] of the sorted row, least median wins
 * (strict <, first index).  desc = concatenated observation descriptors, start = CSR. */
static int cmp_int(const void* a, const void* b) { return *(const int*)a - *(const int*)b; }
int orc_distinctive_descriptors(const uint8_t* desc, const int32_t* start, int npoints, int32_t* best_idx)
{
    for (int p = 0; p < npoints; p++) {
        const int N = start[p + 1] - start[p];
        best_idx[p] = -1;
        if (N <= 0) continue;
        const uint8_t* D = desc + 32 * (size_t)start[p];
        int* row = (int*)malloc(sizeof(int) * (size_t)N);
        int bestMedian = 0x7FFFFFFF, bestIdx = 0;
        for (int i = 0; i < N; i++) {
            for (int j = 0; j < N; j++) row[j] = i == j ? 0 : orc_descriptor_distance(D + 32 * (size_t)i, D + 32 * (size_t)j);
            qsort(row, (size_t)N, sizeof(int), cmp_int);
            const int median = row[(int)(0.5 * (N - 1))];
            if (median < bestMedian) { bestMedian = median; bestIdx = i; }
        }
        free(row);
        best_idx[p] = bestIdx;
    }
    return 0;
}

/* ------------------------------------------------------------------ SURVEY 8(f).3
 * Frame::UndistortKeyPoints (src/Frame.cc:404-434) = cv::undistortPoints(mat, mat, mK, mDistCoef,
 * cv::Mat(), mK): OpenCV 3.0 cvUndistortPoints restated from its published algorithm (UNPINNED,
 * like the other OpenCV primitives): normalise with 1/fx, 5 fixed-point iterations of the
 * radial-tangential model in double, re-project with mK, store as float.  K = fx, fy, cx, cy;
 * D = k1, k2, p1, p2, k3.  D[0] == 0 -> plain copy (:406-410). */
void orc_undistort_keypoints(const OrcKeyPoint* in, int n, const float K[4], const float D[5], OrcKeyPoint* out)
{
    for (int i = 0; i < n; i++) out[i] = in[i];
    if (D[0] == 0.0f) return;
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
    const double ifx = 1. / fx, ify = 1. / fy;
    const double k[8] = {D[0], D[1], D[2], D[3], D[4], 0, 0, 0};
    for (int i = 0; i < n; i++) {
        double x = in[i].x, y = in[i].y, x0, y0;
        x0 = x = (x - cx) * ifx;
        y0 = y = (y - cy) * ify;
        for (int j = 0; j < 5; j++) {
            const double r2 = x * x + y * y;
            const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
            const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        const double xx = fx * x + 0.0 * y + cx, yy = 0.0 * x + fy * y + cy, ww = 1. / (0.0 * x + 0.0 * y + 1.0);
        out[i].x = (float)(xx * ww);
        out[i].y = (float)(yy * ww);
    }
}
