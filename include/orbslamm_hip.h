/*
 * orbslamm_hip.h -- C ABI of the MI355X-native ORB front-end (liborbslamm_hip.so).
 *
 * Drop-in boundary for the per-frame hot path of HDaoud/ORBSLAMM.  Every entry
 * point names the reference interface it replaces (paths under
 * /root/reference/SingleRobotScenario/).  Plain pointers and sizes only; no
 * exceptions cross the ABI; every function returns 0 on success or a negative
 * ORBX_E_* code (orbx_last_error() gives the text).  All compute runs in HIP
 * kernels on gfx950 -- there is no CPU fallback; without a GPU every compute
 * entry returns ORBX_E_NO_DEVICE.
 *
 * Threading (mirrors the reference): one handle = one HIP stream;
 * calls on one handle must be serialised by the caller, distinct handles are
 * fully concurrent and may live on different devices.
 */
#ifndef ORBSLAMM_HIP_H
#define ORBSLAMM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORBX_MAX_LEVELS 16

enum {
    ORBX_OK = 0,
    ORBX_E_INVALID = -1,      /* bad argument */
    ORBX_E_NO_DEVICE = -2,    /* no HIP device / extension unusable: fail loudly, never fall back */
    ORBX_E_HIP = -3,          /* HIP runtime error (text in orbx_last_error) */
    ORBX_E_CAPACITY = -4,     /* caller buffer too small; nothing written past cap */
    ORBX_E_UNSUPPORTED = -5   /* image shape the reference itself cannot handle */
};

const char* orbx_last_error(void);
/* number of visible HIP devices (0 when none); never fails */
int orbx_device_count(void);
/* PCI bus id of a HIP device ("0000:c1:00.0", cap >= 16): /sys/bus/pci/devices/<id>/numa_node names the NUMA node whose
 * cores and memory sit next to it (a rank that feeds host frames should be pinned there) */
int orbx_device_pci_bus_id(int device, char* out, int cap);
/* the shader clock (MHz) the device runs at now UNDER LOAD, measured by a ~0.1 ms probe kernel that fills every SIMD (s_memtime
 * against the 100 MHz s_memrealtime in its first wave).  Diagnostics for multi-GPU launches (one robot per GPU, MultipleRobotsScenario/Examples/Monocular/
 * mono_kitti.cc:80-101): a rank that reports it right behind its timed steps tells a cold or throttled GPU from a slow
 * pipeline.  Synchronous; uses the device's null stream. */
int orbx_device_shader_clock_mhz(int device, float* mhz);

/* ---------------------------------------------------------------- extractor
 * replaces class ORBextractor (include/ORBextractor.h:45-111). */

/* ORBextractor::ORBextractor(int nfeatures, float scaleFactor, int nlevels,
 *                            int iniThFAST, int minThFAST)  src/ORBextractor.cc:410-470 */
typedef struct {
    int32_t nfeatures;
    float scaleFactor;
    int32_t nlevels;
    int32_t iniThFAST;
    int32_t minThFAST;
} OrbxParams;

/* layout-identical to cv::KeyPoint as the reference fills it (28 bytes):
 * pt.x, pt.y, size, angle, response, octave, class_id (=-1)   src/ORBextractor.cc:841-847,1094-1103 */
typedef struct {
    float x, y;
    float size;
    float angle;
    float response;
    int32_t octave;
    int32_t class_id;
} OrbxKeyPoint;

typedef struct orbx_handle orbx_t;

/* Allocates every device buffer once (no allocation on the per-frame path).
 * max_w/max_h: largest frame; max_batch: frames in flight per call. */
int orbx_create(const OrbxParams* params, int max_w, int max_h, int max_batch, int device, orbx_t** out);
/* The same for the LIVE frames of up to max_cameras (1..8) cameras: every call of 1..max_cameras frames runs as ONE
 * latency-mode chain (one queue, frames up through a copy kernel, results behind a flag the caller polls) -- the handle
 * a hub creates that puts the frames of several robots' tracking threads through the GPU together
 * (include/orbslamm_hub.hpp; MultipleRobotsScenario/Examples/Monocular/mono_kitti.cc:83-101 with more robots than the
 * GPU runs queues).  orbx_create keeps the latency mode to calls of one or two frames. */
int orbx_create_live(const OrbxParams* params, int max_w, int max_h, int max_cameras, int device, orbx_t** out);
void orbx_destroy(orbx_t* h);

/* GetLevels / GetScaleFactor / GetScaleFactors / GetInverseScaleFactors /
 * GetScaleSigmaSquares / GetInverseScaleSigmaSquares  (include/ORBextractor.h:60-83).
 * Each array receives nlevels floats; NULL pointers are skipped.  Works without a GPU. */
int orbx_levels(const orbx_t* h);
float orbx_scale_factor(const orbx_t* h);
int orbx_scale_tables(const orbx_t* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2);
/* mnFeaturesPerLevel (src/ORBextractor.cc:435-446) and umax[16] (:452-469) */
int orbx_features_per_level(const orbx_t* h, int32_t* out);
int orbx_umax(const orbx_t* h, int32_t out[16]);
/* upper bound of keypoints one frame can yield (sum over levels of N_l + slack) */
int orbx_max_keypoints(const orbx_t* h);

/* void ORBextractor::operator()(InputArray image, InputArray mask,
 *        vector<KeyPoint>& keypoints, OutputArray descriptors)   src/ORBextractor.cc:1043-1105
 * image: 8-bit single channel, host memory (mask is ignored by the reference).
 * kps[cap], desc[cap*32]; *n_out = number of keypoints.  Empty image (w<=0||h<=0||!img):
 * returns ORBX_OK with outputs untouched, like the reference (:1046-1047) but *n_out = 0. */
int orbx_extract(orbx_t* h, const uint8_t* img, int w, int h_, int stride,
                 OrbxKeyPoint* kps, uint8_t* desc, int cap, int* n_out);

/* Batched form: B independent frames of identical shape, host memory.
 * kps[B*cap], desc[B*cap*32], n_out[B].  Synchronous: = orbx_submit_batch + orbx_collect_batch. */
int orbx_extract_batch(orbx_t* h, const uint8_t* const* imgs, int B, int w, int h_, int stride,
                       OrbxKeyPoint* kps, uint8_t* desc, int cap, int* n_out);

/* ---- pipelined host-buffer entry --------------------------------------------------------------------------------
 * What a multi-robot front-end does with Frame::ExtractORB (src/Frame.cc:247-253) + the match against the previous
 * frame: up to three batches of one stream in flight, the upload of batch n+1 and the download of batch n-1 running
 * (on two copy streams) beside the kernels of batch n.  Tickets are collected in submission order.
 *   - frames in memory from orbx_host_alloc_frames (pinned, rows 64-byte aligned: the device layout) are read by the
 *     DMA engine in place; other pinned / registered memory (orbx_host_register) goes through a strided DMA;
 *     pageable frames are first copied to the handle's pinned staging (row bands over a few threads).
 *     The caller must not touch the frames of a ticket before it has been collected.
 *   - match_prev: every frame is also matched against the previous frame of the stream (rule of
 *     orbx_match_prev_batch_device) and the match tables come back with the batch. */
typedef struct {
    int32_t match_prev;   /* != 0: match vs the previous frame of the stream */
    float nnratio;        /* 0.7 */
    int32_t th_low;       /* TH_LOW = 50 */
    int32_t check_ori;
} OrbxStreamOpts;
/* results of one ticket in handle-owned pinned host memory, valid until orbx_release(ticket) */
typedef struct {
    int32_t B, cap;
    const int32_t* n;         /* [B] keypoints per frame */
    const OrbxKeyPoint* kps;  /* [B][cap] */
    const uint8_t* desc;      /* [B][cap][32] */
    const int32_t* match;     /* [B][cap] index into the previous frame or -1; NULL without match_prev */
    const int32_t* nmatch;    /* [B] */
} OrbxBatchView;
int orbx_submit_batch(orbx_t* h, const uint8_t* const* imgs, int B, int w, int h_, int stride,
                      const OrbxStreamOpts* opts /* may be NULL: extract only */, int* ticket);
/* waits for the ticket; zero-copy view of its results */
int orbx_collect_view(orbx_t* h, int ticket, OrbxBatchView* view);
int orbx_release(orbx_t* h, int ticket);
/* waits, copies the exact n entries per frame to the caller's arrays (kps[B*cap], desc[B*cap*32], match[B*cap];
 * NULL pointers are skipped) and releases the ticket */
int orbx_collect_batch(orbx_t* h, int ticket, OrbxKeyPoint* kps, uint8_t* desc, int cap, int* n_out,
                       int32_t* match, int* nmatch);
/* synchronous form of the pair above (the per-frame drop-in entry when B = 1) */
int orbx_extract_match_batch(orbx_t* h, const uint8_t* const* imgs, int B, int w, int h_, int stride,
                             const OrbxStreamOpts* opts, OrbxKeyPoint* kps, uint8_t* desc, int cap, int* n_out,
                             int32_t* match, int* nmatch);
/* Results straight into the CALLER'S containers, no second pass on the host: the result kernel writes exactly n records
 * per frame into kps[B*cap], desc[B*cap*32], n[B] (and match[B*cap], nmatch[B] with match_prev) over the link.  The
 * arrays must be memory the device can address: from orbx_host_alloc, or registered once with orbx_host_register
 * (a std::vector<cv::KeyPoint>'s storage can be registered after reserve()).  A frame with more than cap keypoints
 * gets its first cap written, n[] holds the true count, orbx_collect returns ORBX_E_CAPACITY.  orbx_collect waits for
 * the ticket and releases it. */
typedef struct {
    OrbxKeyPoint* kps; uint8_t* desc; int32_t* n;
    int32_t* match; int32_t* nmatch;   /* may be NULL without match_prev */
    int32_t cap;
} OrbxBatchOut;
int orbx_submit_batch_into(orbx_t* h, const uint8_t* const* imgs, int B, int w, int h_, int stride,
                           const OrbxStreamOpts* opts, const OrbxBatchOut* out, int* ticket);
int orbx_collect(orbx_t* h, int ticket);
/* pinned (device-addressable, coherent) host memory for such arrays; orbx_host_free releases it */
int orbx_host_alloc(orbx_t* h, size_t bytes, void** p);
/* pinned host frames in the device layout (a cv::Mat can wrap them: cv::Mat(h, w, CV_8UC1, ptr, stride)) */
int orbx_host_alloc_frames(orbx_t* h, int B, int w, int h_, uint8_t** frames, int* stride, size_t* pitch);
int orbx_host_free(orbx_t* h, void* p);
/* pin caller-owned frame memory (e.g. a capture ring buffer) for in-place DMA */
int orbx_host_register(orbx_t* h, void* p, size_t bytes);
int orbx_host_unregister(orbx_t* h, void* p);

/* Device-resident form: frames already in HBM (d_imgs + f*frame_pitch, rows `stride`
 * bytes apart; base, stride and frame_pitch must be multiples of 4).  Results stay
 * in handle-owned HBM (see orbx_device_results); asynchronous on the handle's streams (orbx_sync waits). */
int orbx_extract_batch_device(orbx_t* h, const uint8_t* d_imgs, int B, int w, int h_,
                              int stride, size_t frame_pitch);
/* device pointers of the last batch: kps[B][cap], desc[B][cap][32], counts[B].  Results alternate between two
 * sets: the pointers are those of the batch just extracted and stay untouched until the next-but-one extraction. */
int orbx_device_results(orbx_t* h, OrbxKeyPoint** d_kps, uint8_t** d_desc, int32_t** d_counts, int* cap);
/* blocking D2H copy of one frame of the last batch */
int orbx_download(orbx_t* h, int frame, OrbxKeyPoint* kps, uint8_t* desc, int cap, int* n_out);

/* std::vector<cv::Mat> mvImagePyramid (include/ORBextractor.h:85): lazy D2H of one
 * level (tight rows, no 19 px border -- the border is never read on the mono path).
 * dst may be NULL to query the size.  blurred!=0 returns the 7x7 Gaussian-smoothed
 * working image the descriptors were sampled from (src/ORBextractor.cc:1085-1086). */
int orbx_pyramid_level(orbx_t* h, int frame, int level, int blurred, uint8_t* dst, int* w, int* h_);
/* void Frame::ComputeStereoMatches()   src/Frame.cc:466-638, on the frames the two extractors (left, right: same
 * device, same shape and parameters -- Tracking's mpORBextractorLeft / mpORBextractorRight) extracted last: row-band
 * candidates, best descriptor within one octave and the disparity range [-3, mbf/mb], 11-position SAD search on the
 * keypoint's pyramid level (the consumer of mvImagePyramid, :561,578), parabola sub-pixel fit, median-distance filter.
 * u_right[cap] = mvuRight, depth[cap] = mvDepth (-1: no match); *n_left = left keypoints. */
int orbx_compute_stereo_matches(orbx_t* left, orbx_t* right, int frame, float mb, float mbf,
                                float* u_right, float* depth, int cap, int* n_left);

/* per-stage dump for parity tests: FAST candidates of one level, packed records
 * (see DESIGN.md "candidate record"), unordered; returns count in *n */
int orbx_level_candidates(orbx_t* h, int frame, int level, uint64_t* dst, int cap, int* n);

/* waits for everything the handle has enqueued; returns the device error flag.  The handles of one device and process share
 * their pipeline streams (four for the batched mode, a pool of four chains for the one-frame-per-call mode), so this call --
 * like a call with a new frame shape, which synchronises before it re-configures -- also waits for the work OTHER handles
 * of the device have in those streams: correct, but a robot that syncs in its loop couples its latency to its neighbours'.
 * The results of a call are published behind flags / events of their own (orbx_collect_*, orbm_track_results): a steady
 * loop needs no orbx_sync. */
int orbx_sync(orbx_t* h);

/* Convenience for hosts that do not link HIP themselves (the reference does not):
 * device buffers for the device-resident entry points, on the handle's GPU.
 * orbx_upload is a blocking host->device copy. */
int orbx_device_alloc(orbx_t* h, size_t bytes, void** d_ptr);
int orbx_device_free(orbx_t* h, void* d_ptr);
int orbx_upload(orbx_t* h, void* d_dst, const void* h_src, size_t bytes);

/* "match vs previous frame" for the stream held by this handle:
 * frame f of the last extracted batch is matched against frame f-1 (frame 0 against
 * the last frame of the previous batch; no previous frame -> no matches).
 * Brute-force best/second Hamming over ALL previous-frame descriptors in index order,
 * accept best<=th_low && (float)best < nnratio*(float)second, rotation histogram and
 * three-maxima pruning exactly as ORBmatcher::SearchByBoW does (src/ORBmatcher.cc:230-287).
 * Results stay in HBM: match[B][cap] (int32, -1 = none), nmatch[B].
 * Call it once after every orbx_extract_batch_device of the stream: it also rolls the batch's last frame into
 * the previous-frame slot the next batch is matched against. */
int orbx_match_prev_batch_device(orbx_t* h, float nnratio, int th_low, int check_ori);
int orbx_device_matches(orbx_t* h, int32_t** d_match, int32_t** d_nmatch);
int orbx_download_matches(orbx_t* h, int frame, int32_t* match, int cap, int* nmatch);
/* forget the previous frame (start of a new stream).  A call with a frame shape other than the handle's current one does
 * the same: a new shape starts a new stream (the first frame of the new shape has no previous frame). */
int orbx_reset_stream(orbx_t* h);

/* per-kernel timing with HIP events on the handle's stream.  enable=1 starts
 * collecting (adds two event records per launch); orbx_profile_read returns the
 * accumulated milliseconds and launch count per kernel since the last reset. */
#define ORBX_PROF_MAX 16
typedef struct {
    int32_t n;
    const char* name[ORBX_PROF_MAX];
    double ms[ORBX_PROF_MAX];
    int64_t launches[ORBX_PROF_MAX];
} OrbxProfile;
/* serial != 0: every kernel of a call runs on the handle's main stream (no overlap of
 * blur / matching / sub-batches); used to time kernels in isolation. */
int orbx_set_serial(orbx_t* h, int serial);
int orbx_profile_enable(orbx_t* h, int enable);
/* bracket only the launches of one kernel (name as reported by orbx_profile_read; NULL = all again): the
 * event records between dependent launches cost ~4 % of a stream-overlapped batch, one kernel's ~1 % */
int orbx_profile_select(orbx_t* h, const char* kernel);
int orbx_profile_read(orbx_t* h, OrbxProfile* out, int reset);

/* Diagnostics: which kernels slow each other down when they share the GPU (tools/pair_overlap.py).
 * Kernel i is launched back to back on one stream while kernel j keeps a second stream busy; co_ms[i * n + j] = time per
 * launch of i beside j, alone_ms[i] = alone; nb = frames per launch.  Runs on the buffers of the last extracted and
 * matched batch.  lds_bytes / wg_threads / wgs: workgroup footprint and count of each kernel at that nb. */
int orbx_debug_pair_overlap(orbx_t* h, int nb, float target_ms, int* n_kernels, const char** names,
                            float* alone_ms, float* co_ms, int32_t* lds_bytes, int32_t* wg_threads, int32_t* wgs);

/* Diagnostics: the rate of plain pinned hipMemcpyAsync copies of up_bytes (host to device) and down_bytes (device to host) on
 * this box, each direction alone and both at once on two streams -- the ceiling the host-buffer entries are read against
 * (bench.py host_path.pcie_*).  GB/s of payload. */
int orbx_debug_link_rate(orbx_t* h, size_t up_bytes, size_t down_bytes, int reps, float* h2d_gbs, float* d2h_gbs,
                         float* both_up_gbs, float* both_down_gbs);

/* Diagnostics: the row copy the host-buffer entries stage pageable frames with (streaming stores where the destination
 * rows are 32-byte aligned, memcpy otherwise) -- `rows` rows of `w` bytes from src (pitch spitch) to dst (pitch dpitch).
 * Needs no device: the CPU suite holds it to a plain copy for odd widths, unaligned sources and row tails.
 * Returns 1 if the streaming form ran, 0 for memcpy, < 0 on a bad argument. */
int orbx_debug_stage_rows(uint8_t* dst, size_t dpitch, const uint8_t* src, size_t spitch, size_t w, int rows);

/* ---------------------------------------------------------------- matcher
 * replaces class ORBmatcher (include/ORBmatcher.h:37-102).  The object-graph
 * walking (MapPoint flags, mutex-guarded getters, camera projection) stays in the
 * C++ adapter; these entry points take the flattened arrays (host memory). */
typedef struct orbm_handle orbm_t;
int orbm_create(int device, orbm_t** out);
void orbm_destroy(orbm_t* h);
/* The reference builds an ORBmatcher as a STACK TEMPORARY at every call site (src/Tracking.cc:639, 809, 914, 1242, 1415,
 * 1454, src/LocalMapping.cc:215, 483, src/LoopClosing.cc:245, 603, MultiMapper.cc:180, 687) holding only (mfNNratio,
 * mbCheckOrientation) -- include/ORBmatcher.h:96-97.  The device state a search needs (queue, grow-only scratch, pinned
 * staging block, the pool of frame blocks) therefore belongs to the calling THREAD: this returns the thread's handle for
 * `device`, made at the thread's first call and released when the thread ends (frames made through it keep it alive).  Its
 * queue is one of the device's four shared chain streams, not one more queue per thread (same priority as the latency
 * extractors' chains: high, or normal under ORBX_LAT_PRIO=0).  The price of sharing: the handle's synchronous entries wait on
 * a stream that other robots' chains and matcher calls are queued on too -- with more than four robot threads on a device a
 * call can wait for another robot's work (latency coupling, never a wrong result).  A caller that wants a queue of its own
 * takes orbm_create.  Never pass the thread handle to orbm_destroy. */
int orbm_thread_handle(int device, orbm_t** out);
/* Allocation counters: device allocations (scratch growth + frame blocks) and pinned allocations of handle h (0 for a
 * null h), and the number of matcher handles the process has made.  A steady-state loop leaves all three unchanged. */
int orbm_alloc_stats(orbm_t* h, int64_t* device_allocs, int64_t* host_allocs, int64_t* handles_made);

/* static int ORBmatcher::DescriptorDistance(const cv::Mat&, const cv::Mat&)
 * src/ORBmatcher.cc:1649-1665.  dist[nq*nt], row-major. */
int orbm_distance_matrix(orbm_t* h, const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* dist);

/* brute-force matcher on host arrays (same rule as orbx_match_prev_batch_device) */
int orbm_match_bruteforce(orbm_t* h, const uint8_t* qdesc, const float* qangle, int nq,
                          const uint8_t* tdesc, const float* tangle, int nt,
                          float nnratio, int th_low, int check_ori, int32_t* match, int* nmatches);

/* DBoW2::FeatureVector flattened to CSR (Thirdparty/DBoW2/DBoW2/FeatureVector.h:21-47):
 * node ids ascending, per-node feature-index lists in stored order. */
typedef struct {
    int32_t n_nodes;
    const uint32_t* node_id;
    const int32_t* start;   /* n_nodes + 1 */
    const int32_t* idx;
} OrbmFeatVec;

/* int ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&)     src/ORBmatcher.cc:159-290 (out_by_train=1)
 * int ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vector<MapPoint*>&)  src/ORBmatcher.cc:524-657 (out_by_train=0)
 * qvalid: query feature has a good MapPoint; tvalid (KF-KF form only): train feature has one.
 * match: nt entries (query index per train feature) if out_by_train else nq entries. */
int orbm_search_by_bow(orbm_t* h,
                       const uint8_t* qdesc, const float* qangle, const uint8_t* qvalid, int nq,
                       const OrbmFeatVec* qfv,
                       const uint8_t* tdesc, const float* tangle, const uint8_t* tvalid, int nt,
                       const OrbmFeatVec* tfv,
                       float nnratio, int check_ori, int out_by_train,
                       int32_t* match, int* nmatches);

/* Frame grid: Frame::AssignFeaturesToGrid / PosInGrid / GetFeaturesInArea
 * src/Frame.cc:230-245, 382-392, 327-380 (KeyFrame::GetFeaturesInArea src/KeyFrame.cc:618-657) */
typedef struct {
    float minX, minY;   /* mnMinX, mnMinY */
    float invW, invH;   /* mfGridElementWidthInv, mfGridElementHeightInv */
    int32_t cols, rows; /* FRAME_GRID_COLS 64, FRAME_GRID_ROWS 48 */
} OrbmGrid;

/* SearchByProjection family (projection done by the caller):
 * mode 3: SearchByProjection(Frame&, const vector<MapPoint*>&, th)        src/ORBmatcher.cc:45-129
 * mode 4: SearchByProjection(Frame& Cur, const Frame& Last, th, bMono)    src/ORBmatcher.cc:1330-1472
 * mode 5: SearchByProjection(Frame&, KeyFrame*, set<MapPoint*>&, th, ORBdist) src/ORBmatcher.cc:1474-1601
 * mode 6: SearchByProjection(KeyFrame*, cv::Mat Scw, ...)                 src/ORBmatcher.cc:292-405
 * Per query: (u, v, radius), level window [minLevel,maxLevel] with GetFeaturesInArea's
 * convention, descriptor, angle (modes 4/5), valid, obs_pos (MapPoint::Observations()>0).
 * t_occ (in/out, nt): train feature must be skipped.  assign (in/out, nt): query index
 * holding train feature t, -1 = NULL.
 * Sizes (the reference's loops take any, its frames hold 1000-5000 features): up to 8192 train features the frame's grid and
 * the resolve's tables live in one CU's LDS; beyond -- up to 65535 features per frame and 65536 queries per call, the width
 * of a feature / query index in the candidate lists (ORBX_E_UNSUPPORTED above) -- the same lists and rounds run from memory
 * (slower; frames built by the general-size kernels).  orbv_transform / orbm_frameset_compute_bow: any number of descriptors
 * (the sort's keys in memory above 8192); orbx_compute_stereo_matches: 65535 keypoints per frame. */
typedef struct {
    int32_t mode;
    float nnratio;
    int32_t check_ori;
    int32_t th_dist;
} OrbmProjParams;
int orbm_search_by_projection(orbm_t* h, const OrbmProjParams* pp,
                              const float* q_uvr, const int8_t* q_lvl,
                              const uint8_t* qdesc, const float* qangle,
                              const uint8_t* qvalid, const uint8_t* q_obs_pos, int nq,
                              const OrbmGrid* grid, const OrbxKeyPoint* t_keys_un,
                              const uint8_t* tdesc, int nt,
                              uint8_t* t_occ, int32_t* assign, int* nmatches);

/* The same with the stereo / RGB-D gate of modes 3 and 4 (src/ORBmatcher.cc:91-96, :1409-1415): a train feature with
 * t_uright[t] > 0 (Frame::mvuRight) is skipped when |q_ur[q] - t_uright[t]| > radius of the query (q_uvr[3q+2]);
 * q_ur = MapPoint::mTrackProjXR (mode 3) resp. u - mbf * invzc (mode 4).  Both NULL = the mono call above. */
int orbm_search_by_projection_stereo(orbm_t* h, const OrbmProjParams* pp,
                                     const float* q_uvr, const float* q_ur, const int8_t* q_lvl,
                                     const uint8_t* qdesc, const float* qangle,
                                     const uint8_t* qvalid, const uint8_t* q_obs_pos, int nq,
                                     const OrbmGrid* grid, const OrbxKeyPoint* t_keys_un,
                                     const uint8_t* tdesc, const float* t_uright, int nt,
                                     uint8_t* t_occ, int32_t* assign, int* nmatches);
/* Optional hint of the drop-in members (include/ORBmatcher_hip.hpp): the train side of the NEXT orbm_search_by_projection(_stereo)
 * on this handle -- the very arrays, count and grid that call will pass -- is uploaded and its grid built NOW, asynchronously, so
 * that the device works while the caller walks its MapPoints (the reference's loop head, ORBmatcher.cc:53-76, :1355-1392: mutex-
 * guarded getters, ~100 us for a frame's worth).  The search compares pointers, count and grid with what was prepared and does
 * its own upload when they differ; one search per prepare.  The arrays must not change between the two calls. */
int orbm_projection_prepare(orbm_t* h, const OrbmGrid* grid, const OrbxKeyPoint* t_keys_un, const uint8_t* tdesc, int nt);


/* ---- the remaining ORBmatcher entry points on the same primitive ---- */

/* Independent windowed best search, the device part of
 *   int ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, float th)               src/ORBmatcher.cc:827-975  (chi2 = 1)
 *   int ORBmatcher::Fuse(KeyFrame*, cv::Mat Scw, const vector<MapPoint*>&, float, vector<MapPoint*>&)  :977-1102 (chi2 = 0)
 *   int ORBmatcher::SearchBySim3(KeyFrame*, KeyFrame*, vector<MapPoint*>&, s12, R12, t12, th)   :1104-1328 (both passes, chi2 = 0)
 * Per projected map point: (u, v, radius), predicted level (window [pred-1, pred]), descriptor,
 * valid flag; optional stereo terms (q_ur, t_uright = mvuRight, NULL for mono).
 * inv_sigma2 = mvInvLevelSigma2 (nlevels floats, needed when chi2).  best_idx = -1 if none
 * (best_dist = 256).  The caller applies bestDist <= TH_LOW / TH_HIGH and edits the map. */
int orbm_window_best(orbm_t* h, const float* q_uvr, const float* q_ur, const int8_t* q_pred,
                     const uint8_t* qdesc, const uint8_t* qvalid, int nq,
                     const OrbmGrid* grid, const OrbxKeyPoint* t_keys_un, const uint8_t* tdesc,
                     const float* t_uright, int nt, const float* inv_sigma2, int nlevels, int chi2,
                     int32_t* best_idx, int32_t* best_dist);

/* int ORBmatcher::SearchForInitialization(Frame& F1, Frame& F2, vector<cv::Point2f>& vbPrevMatched,
 *                                         vector<int>& vnMatches12, int windowSize)   src/ORBmatcher.cc:407-522
 * q_xy = vbPrevMatched (nq x 2).  matches12[nq] = index in F2 or -1; the caller refreshes
 * vbPrevMatched from it (:517-519). */
int orbm_search_for_initialization(orbm_t* h, const float* q_xy, float window_size,
                                   const OrbxKeyPoint* q_keys_un, const uint8_t* qdesc, int nq,
                                   const OrbmGrid* grid, const OrbxKeyPoint* t_keys_un, const uint8_t* tdesc, int nt,
                                   float nnratio, int check_ori, int32_t* matches12, int* nmatches);

/* int ORBmatcher::SearchForTriangulation(KeyFrame*, KeyFrame*, cv::Mat F12,
 *              vector<pair<size_t,size_t>>& vMatchedPairs, bool bOnlyStereo)          src/ORBmatcher.cc:659-825
 * skip1/skip2: the feature already has a MapPoint; uright = mvuRight (NULL for mono);
 * F12 row-major; (ex, ey) = epipole in image 2 (:666-672, cv::Mat algebra stays in the caller);
 * sf2 / sigma2_2 = pKF2->mvScaleFactors / mvLevelSigma2.  matches12[n1] = index in KF2 or -1. */
int orbm_search_for_triangulation(orbm_t* h,
                                  const OrbxKeyPoint* k1, const uint8_t* d1, const uint8_t* skip1, const float* uright1, int n1,
                                  const OrbmFeatVec* fv1,
                                  const OrbxKeyPoint* k2, const uint8_t* d2, const uint8_t* skip2, const float* uright2, int n2,
                                  const OrbmFeatVec* fv2,
                                  const float F12[9], float ex, float ey, const float* sf2, const float* sigma2_2, int nlevels,
                                  int only_stereo, int check_ori, int32_t* matches12, int* nmatches);

/* ---- a Frame's matcher-side state held in HBM ------------------------------------------------
 * The tail of Frame::Frame (src/Frame.cc:196-210: UndistortKeyPoints + AssignFeaturesToGrid) run on DEVICE-resident
 * extractor output -- e.g. d_keys = kps + f*cap, d_desc = desc + f*cap*32 from orbx_device_results -- so that the
 * projection searches consume it without a round trip through the host.  The frame copies what it needs (the
 * extractor's result sets are reused two batches later) into ONE block taken from the creating handle's pool and given
 * back by orbm_frame_destroy (from any thread): a Frame per image allocates nothing in steady state, and neither call
 * waits for the device.  Any matcher handle of the same device may search a frame; the creating handle lives until its
 * last frame is destroyed.
 * LIFETIME OF THE INPUTS: orbm_frame_create returns with the build ENQUEUED on the creating handle's stream; d_keys and
 * d_desc must stay untouched until the build has read them -- i.e. until the first search, BoW or download on the frame
 * has returned (each returns after its own results, enqueued behind the build, have landed), or until
 * orbm_frame_settle(f).  A caller that recycles the input buffers at once (orbx_upload into them, the extractor's next
 * but one batch) calls orbm_frame_settle first; Tracking's shape -- create, then search -- needs nothing. */
typedef struct orbm_frame orbm_frame_t;
int orbm_frame_create(orbm_t* h, const OrbxKeyPoint* d_keys, const uint8_t* d_desc, int n,
                      const float K[4], const float D[5], const OrbmGrid* grid, orbm_frame_t** out);
int orbm_frame_destroy(orbm_frame_t* f);
int orbm_frame_size(const orbm_frame_t* f);
/* waits (once; later calls return at once) until the frame's build has finished reading d_keys / d_desc */
int orbm_frame_settle(orbm_frame_t* f);
/* mvKeysUn for the host side of Tracking (pose optimisation reads it) */
int orbm_frame_download_keys_un(orbm_frame_t* f, OrbxKeyPoint* keys_un);
/* orbm_search_by_projection with the frame as train side (CurrentFrame of modes 3-5, the KeyFrame of mode 6) */
int orbm_search_by_projection_frame(orbm_t* h, const OrbmProjParams* pp,
                                    const float* q_uvr, const int8_t* q_lvl, const uint8_t* qdesc, const float* qangle,
                                    const uint8_t* qvalid, const uint8_t* q_obs_pos, int nq,
                                    orbm_frame_t* train, uint8_t* t_occ, int32_t* assign, int* nmatches);

/* rounds and candidates of the handle's last projection search (diagnostics: the queries of the reference's sequential
 * loop are resolved in parallel rounds, see orbt_kernels.hip) */
int orbm_last_search_stats(orbm_t* h, int* rounds, int* candidates);

/* ---- the Tracking-shaped path, batched and without host round trips (round 3) -----------------------------------
 * A frame set holds `slots` device-resident frames of at most `cap` features: mvKeysUn, descriptors, mGrid.
 *   orbm_frameset_build*: the tail of Frame::Frame (src/Frame.cc:196-210: UndistortKeyPoints + AssignFeaturesToGrid) for
 *     n frames in ONE launch from device-resident extractor output (frame i at d_keys + i*src_cap, d_desc + i*src_cap*32,
 *     d_counts[i]; counts are read on the device), into slots (slot0 + i) % slots.  Asynchronous.
 *   orbm_track_frames: int ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, float th, bool bMono)
 *     src/ORBmatcher.cc:1330-1472 as Tracking::TrackWithMotionModel calls it (src/Tracking.cc:925-936, th = 15 then 30,
 *     bMono = true), for npairs (CurrentFrame, LastFrame) slot pairs in ONE launch.  Every LastFrame feature is a query
 *     (as if it held a MapPoint with observations, none an outlier) projected with the identity pose: u, v =
 *     mvKeysUn[i].pt, skipped outside `bounds` (:1375-1378), radius = th * mvScaleFactors[octave], octave window
 *     [octave-1, octave+1] (:1381-1392) -- SURVEY.md 8(d)'s grid-windowed variant.  A caller with a pose uses
 *     orbm_search_by_projection_frame on the same data.  pp->mode must be 4.  Asynchronous.
 *   orbm_track_results: waits for the last (back = 0) or an earlier (back = 1 .. 3) orbm_track_frames; assign[p*cap + t] =
 *     LastFrame feature index held by CurrentFrame feature t (or -1), nmatches[p]; both point into pinned host memory the
 *     kernel wrote (four result sets in rotation: valid until three more calls have been issued).
 * bounds = mnMinX, mnMaxX, mnMinY, mnMaxY (Frame::ComputeImageBounds); scale_factors = mvScaleFactors. */
typedef struct orbm_frameset orbm_frameset_t;
int orbm_frameset_create(orbm_t* h, int slots, int cap, const float K[4], const float D[5], const OrbmGrid* grid,
                         const float bounds[4], const float* scale_factors, int nlevels, orbm_frameset_t** out);
int orbm_frameset_destroy(orbm_frameset_t* fs);
int orbm_frameset_build(orbm_frameset_t* fs, int slot0, int n, const OrbxKeyPoint* d_keys, const uint8_t* d_desc,
                        const int32_t* d_counts, int src_cap);
/* the frames of the extractor's last batch (orbx_extract_batch_device / orbx_submit_batch), ordered behind its kernels on
 * the device; the extractor will not reuse that result set before the build has read it. */
int orbm_frameset_build_from_extractor(orbm_frameset_t* fs, int slot0, orbx_t* ex);
int orbm_frameset_sync(orbm_frameset_t* fs);
/* The live stream -- one frame per robot per call, what the reference's main loops do
 * (MultipleRobotsScenario/Examples/Monocular/mono_kitti.cc:80-101 -> Tracking::GrabImageMonocular, src/Tracking.cc:240-267):
 * after orbm_frameset_attach(fs, ex) the frame set's kernels are enqueued on the extractor's own stream, so that
 *   orbx_submit_batch(ex, one frame)  ->  orbm_frameset_build_from_extractor(fs, slot, ex)  ->  orbm_track_frames(fs, ...)
 * is ONE chain of kernels on the device: Frame::Frame (ExtractORB, UndistortKeyPoints, AssignFeaturesToGrid, src/Frame.cc:175-210)
 * followed by TrackWithMotionModel's SearchByProjection (src/Tracking.cc:925-936) with no host round trip and no cross-stream
 * hand-over in between; the keypoints / descriptors come back through the ticket (flag-polled), the match table through
 * orbm_track_results (flag-polled for up to 8 pairs).  The search of frame t needs only the pose PREDICTED from frame t-1
 * (mVelocity * mLastFrame.mTcw, src/Tracking.cc:905), so it can be submitted together with the frame.  ex = NULL detaches.
 * Attach before the first orbm_frameset_compute_bow.  Calls on the extractor and on the attached frame set must come from
 * one thread at a time (they share a stream). */
int orbm_frameset_attach(orbm_frameset_t* fs, orbx_t* ex);
/* mvKeysUn / descriptors of one slot for the host side (pose optimisation reads mvKeysUn) */
int orbm_frameset_download(orbm_frameset_t* fs, int slot, OrbxKeyPoint* keys_un, uint8_t* desc, int cap, int* n_out);
int orbm_track_frames(orbm_frameset_t* fs, const OrbmProjParams* pp, float th, const int32_t* cur_slots,
                      const int32_t* last_slots, int npairs);
/* int ORBmatcher::SearchByProjection(Frame& F, const vector<MapPoint*>& vpMapPoints, const float th)   src/ORBmatcher.cc:45-129
 * as Tracking::SearchLocalPoints runs it on every frame (src/Tracking.cc:1242-1249: th = 1, or 3 shortly after a
 * relocalisation, nnratio 0.8), with the frame resident in slot `slot` of the set.  Per local MapPoint in view
 * (Frame::isInFrustum, src/Frame.cc:269-325): q_uvr = mTrackProjX, mTrackProjY, r = RadiusByViewingCos(mTrackViewCos) * th *
 * mvScaleFactors[mnTrackScaleLevel]; q_lvl = mnTrackScaleLevel - 1, mnTrackScaleLevel; its descriptor; qvalid =
 * mbTrackInView && !isBad (NULL: all); q_obs_pos = Observations() > 0 (NULL: all); t_occ[cap] = F.mvpMapPoints[t] holds a
 * MapPoint with observations (NULL: none).  One pinned upload, two launches, no sync: asynchronous like orbm_track_frames,
 * the table comes back through orbm_track_results as a search of ONE pair (assign[t] = index of the MapPoint taken by
 * feature t or -1; the caller writes F.mvpMapPoints).  pp->mode 3 (modes 4 - 6 run without the rotation check: there are no
 * query angles here).  nq <= slots * cap. */
int orbm_track_local_points(orbm_frameset_t* fs, int slot, const OrbmProjParams* pp, const float* q_uvr, const int8_t* q_lvl,
                            const uint8_t* qdesc, const uint8_t* qvalid, const uint8_t* q_obs_pos, int nq, const uint8_t* t_occ);
/* int ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono)
 * src/ORBmatcher.cc:1330-1472 as Tracking::TrackWithMotionModel calls it (src/Tracking.cc:925-936) with a REAL pose:
 * the caller projects LastFrame's MapPoints with CurrentFrame.mTcw (:1362-1378 -- cv::Mat algebra, stays on the host) and
 * passes, per LastFrame feature i < nq: q_uvr = u, v, radius = th * mvScaleFactors[octave]; q_lvl = the octave window
 * (:1381-1392); qvalid = holds a MapPoint that is no outlier, projects with positive depth into the image bounds
 * (:1349-1378); q_obs_pos (NULL: all); t_occ[cap] as orbm_track_local_points.  LastFrame's descriptors and angles
 * (rotation check, pp->check_ori) and CurrentFrame's grid stay in HBM: 14 bytes per feature go up.  pp->mode 4 (or 5: the
 * relocalisation search, same shape).  Asynchronous; the table comes back through orbm_track_results as ONE pair. */
int orbm_track_frame_projected(orbm_frameset_t* fs, int cur_slot, int last_slot, const OrbmProjParams* pp, const float* q_uvr,
                               const int8_t* q_lvl, const uint8_t* qvalid, const uint8_t* q_obs_pos, int nq, const uint8_t* t_occ);
int orbm_track_results(orbm_frameset_t* fs, int back, const int32_t** assign, const int32_t** nmatches, int* npairs, int* cap);
int orbm_track_stats(orbm_frameset_t* fs, int pair, int* rounds, int* candidates);

/* void Frame::UndistortKeyPoints()   src/Frame.cc:404-434  (cv::undistortPoints(mat, mat, mK, mDistCoef, Mat(), mK)).
 * K = fx, fy, cx, cy; D = k1, k2, p1, p2, k3.  D[0] == 0: plain copy (:406-410).  Only pt changes. */
int orbm_undistort_keypoints(orbm_t* h, const OrbxKeyPoint* keys, int n, const float K[4], const float D[5],
                             OrbxKeyPoint* keys_un);

/* void Frame::ComputeStereoFromRGBD(const cv::Mat& imDepth)   src/Frame.cc:641-663 (the RGB-D Frame constructor, :104-142).
 * depth: the CV_32F depth image (w x hh floats, rows `stride` floats apart, already scaled by mDepthMapFactor as Tracking::
 * GrabImageRGBD does, src/Tracking.cc:215-216); keys = mvKeys (the lookup uses the DISTORTED position, float -> int by
 * truncation like Mat::at<float>(v, u)), keys_un = mvKeysUn.  uright[i] = kpU.pt.x - mbf / d and depth_out[i] = d where
 * d > 0, else both -1.  A keypoint outside the depth image (the reference reads out of bounds) counts as d = 0. */
int orbm_compute_stereo_from_rgbd(orbm_t* h, const OrbxKeyPoint* keys, const OrbxKeyPoint* keys_un, int n, const float* depth, int w, int hh,
                                  int stride, float mbf, float* uright, float* depth_out);

/* void MapPoint::ComputeDistinctiveDescriptors()   src/MapPoint.cc:242-307, batched over map points.
 * desc = the observations' descriptors of all points back to back, start[npoints+1] = CSR.
 * best_idx[p] = index (inside point p's list) of the descriptor with the least median distance
 * to the others, -1 for a point without observations. */
int orbm_distinctive_descriptors(orbm_t* h, const uint8_t* desc, const int32_t* start, int npoints, int32_t* best_idx);

/* The descriptor text of the map file: `os << pKF->mDescriptors`, `os << pMP->GetDescriptor()`  src/MapSerializer.cc:344-347,
 * 429-431 (cv::Mat's stream operator, OpenCV 3.0 default formatter: "[%3d, %3d, ...;\n %3d, ...]").  Host-side
 * formatting of host data; needs no device.  out == NULL queries *len (bytes without the terminating 0). */
int orbm_descriptors_to_text(const uint8_t* desc, int n, int cols, char* out, size_t cap, size_t* len);
/* the inverse, for map loaders (desc == NULL counts the rows) */
int orbm_descriptors_from_text(const char* text, uint8_t* desc, int cap_rows, int cols, int* n_rows);

/* GetFeaturesInArea on the device grid, for tests: out[cap] indices in reference order */
int orbm_features_in_area(orbm_t* h, const OrbmGrid* grid, const OrbxKeyPoint* keys_un, int n,
                          float x, float y, float r, int minLevel, int maxLevel,
                          int32_t* out, int cap, int* n_out);

/* ---------------------------------------------------------------- vocabulary
 * replaces ORBVocabulary::transform as called by Frame::ComputeBoW (src/Frame.cc:395-402):
 *   void TemplatedVocabulary::transform(const vector<TDescriptor>&, BowVector&, FeatureVector&, int levelsup)
 *   Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1194 (+ :1218-1259)
 * The tree is uploaded once; nodes come in loadFromTextFile order (:1338-1425): entry i is node
 * id i+1 (root = 0), leaves receive word ids in file order.  scoring/weighting are DBoW2's enums. */
typedef struct orbv_handle orbv_t;
int orbv_create(int device, int k, int L, int scoring, int weighting, int n_nodes,
                const int32_t* parent, const uint8_t* is_leaf, const uint8_t* desc /* n x 32 */,
                const double* weight, orbv_t** out);
/* bool TemplatedVocabulary::loadFromTextFile(const std::string&)  (ORBvoc.txt format) */
int orbv_load_text(int device, const char* path, orbv_t** out);
void orbv_destroy(orbv_t* v);
/* BowVector as (word id ascending, value); FeatureVector as CSR (node id ascending, feature
 * indices ascending).  Capacities: n for word_id/word_value/fv_node/fv_idx, n+1 for fv_start. */
int orbv_transform(orbv_t* v, const uint8_t* desc, int n, int levelsup,
                   uint32_t* word_id, double* word_value, int* n_words,
                   uint32_t* fv_node, int32_t* fv_start, int32_t* fv_idx, int* n_fv_nodes);

/* Frame::ComputeBoW (src/Frame.cc:394-402) on a device-resident frame: BowVector to the host (word id ascending,
 * value; capacity orbm_frame_size), FeatureVector kept with the frame.  Vocabulary and frame must share the device. */
int orbm_frame_compute_bow(orbm_frame_t* f, orbv_t* voc, int levelsup, uint32_t* word_id, double* word_value, int* n_words);
/* orbm_search_by_bow between two frames that ran orbm_frame_compute_bow (query = KeyFrame / pKF1, train = Frame / pKF2) */
int orbm_search_by_bow_frames(orbm_t* h, orbm_frame_t* q, const uint8_t* qvalid, orbm_frame_t* t, const uint8_t* tvalid,
                              float nnratio, int check_ori, int out_by_train, int32_t* match, int* nmatches);

/* ---- the BoW side of the Tracking-shaped path on a frame set (round 3) ----
 * orbm_frameset_compute_bow: void Frame::ComputeBoW() (src/Frame.cc:394-402: mpORBvocabulary->transform(vCurrentDesc, mBowVec,
 *   mFeatVec, 4)) for slots (slot0 + i) % slots, i < n, in two launches; asynchronous.  FeatureVectors stay in HBM;
 *   orbm_frameset_bow_vector hands one slot's mBowVec to the host (word id ascending, value).
 * orbm_bow_frames: int ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches)
 *   src/ORBmatcher.cc:159-290 as Tracking::TrackReferenceKeyFrame calls it (src/Tracking.cc:805-812), for npairs
 *   (KeyFrame slot, Frame slot) pairs in two launches; every KeyFrame feature counts as holding a good MapPoint.
 *   Asynchronous.  orbm_bow_results as orbm_track_results: match[p*cap + t] = KeyFrame feature matched to Frame feature
 *   t or -1, nmatches[p], in pinned host memory (four result sets in rotation). */
int orbm_frameset_compute_bow(orbm_frameset_t* fs, orbv_t* voc, int slot0, int n, int levelsup);
int orbm_frameset_bow_vector(orbm_frameset_t* fs, int slot, uint32_t* word_id, double* word_value, int cap, int* n_words);
int orbm_bow_frames(orbm_frameset_t* fs, const int32_t* kf_slots, const int32_t* frame_slots, int npairs, float nnratio, int check_ori);
int orbm_bow_results(orbm_frameset_t* fs, int back, const int32_t** match, const int32_t** nmatches, int* npairs, int* cap);

/* orbm_window_best with a device-resident frame as train side (the KeyFrame of Fuse / Fuse(Scw) / SearchBySim3; mono) */
int orbm_window_best_frame(orbm_t* h, const float* q_uvr, const int8_t* q_pred, const uint8_t* qdesc, const uint8_t* qvalid, int nq,
                           orbm_frame_t* train, const float* inv_sigma2, int nlevels, int chi2,
                           int32_t* best_idx, int32_t* best_dist);
/* orbm_search_for_triangulation between two device-resident frames that ran orbm_frame_compute_bow (mono);
 * skip1 / skip2 (host): the feature already has a MapPoint */
int orbm_search_for_triangulation_frames(orbm_t* h, orbm_frame_t* f1, const uint8_t* skip1, orbm_frame_t* f2, const uint8_t* skip2,
                                         const float F12[9], float ex, float ey, const float* sf2, const float* sigma2_2, int nlevels,
                                         int check_ori, int32_t* matches12, int* nmatches);

/* orbm_search_for_initialization between two device-resident frames (F1 = the initial frame, F2 = the current one);
 * q_xy = vbPrevMatched (host, F1 size), matches12: F1 size */
int orbm_search_for_initialization_frames(orbm_t* h, const float* q_xy, float window_size, orbm_frame_t* f1, orbm_frame_t* f2,
                                          float nnratio, int check_ori, int32_t* matches12, int* nmatches);

#ifdef __cplusplus
}
#endif
#endif
