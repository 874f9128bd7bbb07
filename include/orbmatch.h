/*
 * orbmatch.h -- C ABI of liborbmatch_hip.so: MI355X (gfx950) implementation of the ORB-SLAM2
 * descriptor-matching hot path (ORB_SLAM2::ORBmatcher).
 *
 * The reference has no FFI; its boundary is the C++ class ORBmatcher (ORBmatcher.h:39-257).
 * Each entry point below is the flat (pointer + size) form of one reference method; the
 * drop-in C++ class in orbslam2_amd/host/ORBmatcher.{h,cc} flattens the pointer graph, calls
 * these and re-applies the pointer side effects.  All citations are file:line under the
 * reference tree.
 *
 * Conventions
 *  - plain C types only; every pointer is borrowed for the duration of the call;
 *  - return value >= 0 is success (a count where the reference returns one), < 0 is an
 *    ORBM_E* code; orbm_last_error() returns the text of the last failure on that context;
 *  - a context (orbm_ctx) is one calling thread's state: device, HIP stream, grow-only device
 *    workspace, frame table.  The reference matcher is called concurrently from three threads
 *    (System.cc:112,117); give each thread its own context.  Calls on one context must not
 *    overlap;
 *  - "io_on_device": 0 = the index lists and outputs of a batch call are HOST pointers (the
 *    call stages them and synchronises before returning); 1 = they are DEVICE pointers and the
 *    call only enqueues work on the context's stream (no allocation-free guarantee: the
 *    workspace may grow on the first call of a given size).
 */
#ifndef ORBMATCH_H
#define ORBMATCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ORBmatcher.cc:37-39 */
#define ORBM_TH_HIGH 100
#define ORBM_TH_LOW 50
#define ORBM_HISTO_LENGTH 30
/* Frame.h:37-38 */
#define ORBM_GRID_ROWS 48
#define ORBM_GRID_COLS 64
#define ORBM_GRID_CELLS (ORBM_GRID_ROWS * ORBM_GRID_COLS)
#define ORBM_MAX_LEVELS 16

#define ORBM_OK 0
#define ORBM_EINVAL (-1)  /* bad argument (NULL, negative size, id out of range, stride too small) */
#define ORBM_EHIP (-2)    /* a HIP runtime call failed; see orbm_last_error */
#define ORBM_ENOMEM (-3)  /* host or device allocation failed */
#define ORBM_ELIMIT (-4)  /* input exceeds a documented limit of the kernel */

/* mp_flags bits (per keypoint): state of Frame::mvpMapPoints[i] / KeyFrame::GetMapPoint(i). */
#define ORBM_MP_PRESENT 1u /* pointer non-NULL */
#define ORBM_MP_BAD 2u     /* MapPoint::isBad() (MapPoint.cc:315) */
#define ORBM_MP_OBS 4u     /* MapPoint::Observations() > 0 (MapPoint.cc:212) */

/* query flags for orbm_proj_queries */
#define ORBM_Q_VALID 1u /* mbTrackInView && !isBad()  (ORBmatcher.cc:69-73) */
#define ORBM_Q_OBS 2u   /* Observations() > 0 of the query map point (seen by later queries through
                           F.mvpMapPoints[idx]->Observations(), ORBmatcher.cc:113-115) */

/*
 * Flat view of what the matcher reads from a Frame / KeyFrame (Frame.h:235-291, KeyFrame.h:232-282).
 * For a KeyFrame pass min_x..max_y already truncated to int as KeyFrame.cc:49-50 does.
 */
typedef struct orbm_frame_view {
    int32_t n;                  /* N */
    const uint8_t *desc;        /* mDescriptors, n x 32 bytes, row i = keypoint i (Frame.h:253) */
    const float *x;             /* mvKeysUn[i].pt.x */
    const float *y;             /* mvKeysUn[i].pt.y */
    const float *angle;         /* mvKeysUn[i].angle, degrees */
    const float *angle_raw;     /* mvKeys[i].angle (ORBmatcher.cc:323 reads mvKeys); NULL => angle */
    const int32_t *octave;      /* mvKeysUn[i].octave */
    const float *uright;        /* mvuRight; NULL => all -1 (monocular) */
    const uint8_t *mp_flags;    /* ORBM_MP_* per keypoint; NULL => all 0 */
    float min_x, min_y, max_x, max_y; /* mnMinX, mnMinY, mnMaxX, mnMaxY */
    float inv_w, inv_h;         /* mfGridElementWidthInv, mfGridElementHeightInv */
    int32_t n_levels;           /* mnScaleLevels (<= ORBM_MAX_LEVELS) */
    const float *scale_factors; /* mvScaleFactors[n_levels] */
    const float *level_sigma2;  /* mvLevelSigma2[n_levels] */
    /* DBoW2::FeatureVector as CSR: nodes ascending, indices in map order (ORBmatcher.cc:243-254) */
    int32_t n_nodes;
    const uint32_t *fv_node;    /* n_nodes node ids, strictly ascending */
    const uint32_t *fv_off;     /* n_nodes+1 offsets into fv_idx */
    const uint32_t *fv_idx;     /* keypoint indices */
} orbm_frame_view;

/* Flat view of the map points handed to SearchByProjection(Frame&, vector<MapPoint*>&, th)
 * after Frame::isInFrustum filled the tracking fields (Frame.cc:407-412, MapPoint.h:183-198). */
typedef struct orbm_proj_queries {
    int32_t n;
    const uint8_t *desc;    /* MapPoint::GetDescriptor(), n x 32 (MapPoint.cc:453) */
    const float *proj_x;    /* mTrackProjX */
    const float *proj_y;    /* mTrackProjY */
    const float *proj_xr;   /* mTrackProjXR */
    const int32_t *level;   /* mnTrackScaleLevel */
    const float *view_cos;  /* mTrackViewCos */
    const uint8_t *flags;   /* ORBM_Q_* ; NULL => all ORBM_Q_VALID|ORBM_Q_OBS */
} orbm_proj_queries;

/* Flat view of the queries of the best-1 windowed searches (the three other SearchByProjection
 * overloads, ORBmatcher.cc:401-542, 1768-1950, 1963-2123).  The host does the float 3-D
 * projection and the gates that precede GetFeaturesInArea; the device does the window walk,
 * the Hamming loop, the claims and the rotation histogram. */
typedef struct orbm_win_queries {
    int32_t n;
    const uint8_t *desc;     /* n x 32 */
    const float *u;          /* window centre */
    const float *v;
    const float *radius;     /* window half size (already th*scale) */
    const int32_t *min_level;/* level arguments of GetFeaturesInArea / the [l-1,l] octave filter */
    const int32_t *max_level;
    const float *ur;         /* u - bf*invz for the stereo gate; NULL => gate off */
    const float *angle;      /* query keypoint angle for the rotation histogram; NULL => no histogram */
    const uint8_t *flags;    /* ORBM_Q_VALID (query passed the host gates), ORBM_Q_OBS */
} orbm_win_queries;

typedef struct orbm_ctx orbm_ctx;

/* ---- context ------------------------------------------------------------------------- */
orbm_ctx *orbm_create(int device);           /* NULL on failure (orbm_last_error(NULL) has the text) */
void orbm_destroy(orbm_ctx *ctx);
const char *orbm_last_error(const orbm_ctx *ctx);
/* adopt a caller's hipStream_t (NULL is the HIP null stream); (void*)-1 returns to the context's own stream */
int orbm_set_stream(orbm_ctx *ctx, void *hip_stream);
/* Measurement (no counterpart in the reference; bench.py's roofline figure): after orbm_kernel_timing(ctx, n) the library
 * records a pair of HIP events on the context's stream around the matching kernel of each of the next n brute-force
 * launches (orbm_bf2nn_batch, orbm_allpairs_batch kind 0) -- the kernel alone, not the call's memsets or the record
 * compaction -- without synchronising anything.  orbm_kernel_times synchronises the stream, writes the elapsed
 * milliseconds of the recorded launches (at most cap) and returns how many were recorded; recording starts over.
 * orbm_kernel_timing(ctx, 0) switches it off.  Creates and destroys events: not for the allocation-free callers. */
int orbm_kernel_timing(orbm_ctx *ctx, int max_launches);
int orbm_kernel_times(orbm_ctx *ctx, float *ms, int cap);
int orbm_synchronize(orbm_ctx *ctx);
int orbm_device_info(orbm_ctx *ctx, int32_t *n_cu, int32_t *clock_khz, char *name, int name_len);

/* ---- device-resident Frame / KeyFrame storage (Frame.h:253,267,249; KeyFrame.h:232,236,282) --
 * Uploads descriptors (packed 32-byte rows), keypoint SoA, FeatureVector CSR and builds the
 * 64x48 grid-cell CSR ON THE DEVICE with the round() rule of Frame::PosInGrid
 * (Frame.cc:278-298, 506-519).  out_ids receive small integer handles. */
int orbm_frames_upload(orbm_ctx *ctx, int n_frames, const orbm_frame_view *views, int32_t *out_ids);
int orbm_frames_free(orbm_ctx *ctx, int n_frames, const int32_t *ids);
/* replaces mp_flags of one frame (map points change while descriptors do not) */
int orbm_frame_set_mp_flags(orbm_ctx *ctx, int32_t id, const uint8_t *mp_flags);
/* copies the device-built grid back: cell_off[ORBM_GRID_CELLS+1], cell_idx[n]; cell = ix*48+iy */
int orbm_frame_grid_download(orbm_ctx *ctx, int32_t id, uint32_t *cell_off, uint32_t *cell_idx);

/* query sets of SearchByProjection(F, vpMapPoints, th): device copy = the descriptors (32 B per query) + ONE 16-byte
 * record per query {proj_x, proj_y, proj_xr, level | flags << 8 | (view_cos >= 0.998f) << 16} (the comparison of
 * ORBmatcher.cc:168-174 is evaluated here; a search gathers a query with one load).  48 bytes per query in HBM. */
int orbm_proj_queries_upload(orbm_ctx *ctx, int n_sets, const orbm_proj_queries *sets, int32_t *out_ids);
int orbm_win_queries_upload(orbm_ctx *ctx, int n_sets, const orbm_win_queries *sets, int32_t *out_ids);
int orbm_queries_free(orbm_ctx *ctx, int n_sets, const int32_t *ids);

/* ---- a1: ORBmatcher::DescriptorDistance (ORBmatcher.cc:2184-2200), n independent pairs ---- */
int orbm_hamming256_pairs(orbm_ctx *ctx, int n, const uint8_t *a, const uint8_t *b, int32_t *dist,
                          int io_on_device);

/* ---- the scoring inside KeyFrameDatabase::DetectLoopCandidates / DetectRelocalizationCandidates
 * (KeyFrameDataBase.cc:109-285, 287-431): `mpVoc->score(query.mBowVec, pKFi->mBowVec)` (:188, :356) for many keyframes per
 * call, and the number of shared words (mnLoopWords / mnRelocWords).  DBoW2's L1Scoring (ORBvoc.txt's scoring type).
 * A BowVector arrives flat: n ascending word ids and their values (DBoW2::BowVector is a std::map<WordId, double>). */
typedef struct orbm_bowvec_view {
    int32_t n;                   /* at most 16384 words */
    const uint32_t *word;        /* strictly ascending */
    const double *value;
} orbm_bowvec_view;
int orbm_bowvecs_upload(orbm_ctx *ctx, int n_vecs, const orbm_bowvec_view *views, int32_t *out_ids);
int orbm_bowvecs_free(orbm_ctx *ctx, int n_vecs, const int32_t *ids);
/* score[p] = L1Scoring::score(vec_a[p], vec_b[p]) (double; the reference narrows it to float), common[p] (nullable) = number
 * of words both hold.  With io_on_device the id lists and outputs are device addresses and the ids are not checked. */
int orbm_bow_score_batch(orbm_ctx *ctx, int n_pairs, const int32_t *vec_a, const int32_t *vec_b, double *score,
                         int32_t *common, int io_on_device);
/* Copies a device-resident BowVector to the host (n words; the arrays hold `capacity` entries). */
int orbm_bowvec_download(orbm_ctx *ctx, int32_t id, int32_t *n, uint32_t *word, double *value, int32_t capacity);

/* ---- DBoW2 TemplatedVocabulary::transform(features, ..., FeatureVector, levelsup), word / node part,
 * as called by Frame::ComputeBoW (Frame.cc:526-533) and KeyFrame::ComputeBoW (KeyFrame.cc:69-78) with levelsup = 4.
 * DBoW2 is not part of the reference tree; the tree arrives flat.  Node 0 is the root; the children of node i are
 * child_idx[child_off[i] .. child_off[i+1]) in DBoW2's stored order (ties go to the first child), every child id is
 * larger than its parent's and the root has children; desc = 32 bytes per node (row 0 unused); word_id = the word of a
 * leaf, -1 for inner nodes; weight = the node weights (nullable = all positive): transform() leaves a feature whose word
 * has weight <= 0 out of the FeatureVector ("if (w > 0)"); depth = L (levels below the root).  */
typedef struct orbm_vocab_view {
    int32_t n_nodes, depth;
    const uint32_t *child_off;   /* n_nodes + 1 */
    const uint32_t *child_idx;
    const uint8_t *desc;         /* n_nodes x 32 */
    const int32_t *word_id;      /* n_nodes */
    const double *weight;        /* n_nodes or NULL */
    int32_t weighting;           /* DBoW2 WeightingType: 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY (ORBvoc.txt: 0) */
    int32_t scoring;             /* DBoW2 ScoringType: 0 L1_NORM, 1 L2_NORM, 2 CHI_SQUARE, 3 KL, 4 BHATTACHARYYA, 5 DOT_PRODUCT (ORBvoc.txt: 0) */
} orbm_vocab_view;
int orbm_vocab_upload(orbm_ctx *ctx, const orbm_vocab_view *vocab, int32_t *vocab_id);
int orbm_vocab_free(orbm_ctx *ctx, int32_t vocab_id);
/* For every listed frame (host array of ids, each at most once): descends the tree with each descriptor and REPLACES the
 * frame's device-resident FeatureVector with the one DBoW2 would build (node at level depth - levelsup, nodes ascending,
 * feature indices ascending inside a node), so that orbm_search_bow_batch / orbm_search_triang_batch need no
 * FeatureVector from the host.  word_id / node_id (host, n_frames x stride, nullable) receive the per-feature word id
 * and node id (node id -1: the word's weight is not positive, the feature is in no node); rows beyond a frame's n are
 * -1.  bowvec_ids (host, n_frames, nullable): also builds each frame's BowVector on the device (word weights added per
 * feature in feature order, L1-normalised in word order, as transform() does for the vocabulary's weighting / scoring;
 * not for L2_NORM) and returns their ids for orbm_bow_score_batch; release them with orbm_bowvecs_free.
 * Limit: 8192 features per frame. */
int orbm_frames_compute_bow(orbm_ctx *ctx, int32_t vocab_id, int n_frames, const int32_t *frames, int levelsup,
                            int32_t stride, int32_t *word_id, int32_t *node_id, int32_t *bowvec_ids);
/* The same descent for n bare descriptors (32 bytes each; host, or device and 16-byte aligned when io_on_device):
 * word_id[n], node_id[n] as above.  What a per-call ORBVocabulary::transform needs before a frame exists. */
int orbm_vocab_transform(orbm_ctx *ctx, int32_t vocab_id, int n, const uint8_t *desc, int levelsup, int32_t *word_id,
                         int32_t *node_id, int io_on_device);
/* Copies a frame's device-resident FeatureVector to the host: fv_node[n_nodes], fv_off[n_nodes + 1], fv_idx[fv_off[n_nodes]];
 * the arrays must hold n, n + 1 and n entries (n = the frame's feature count). */
int orbm_frame_fv_download(orbm_ctx *ctx, int32_t frame, int32_t *n_nodes, uint32_t *fv_node, uint32_t *fv_off, uint32_t *fv_idx);

/* ---- descriptor stage of Frame::ComputeStereoMatches (Frame.cc:632-728), batched over stereo
 * frames.  frame_left / frame_right are uploaded views whose x, y, octave, desc are mvKeys / mvKeysRight and
 * mDescriptors / mDescriptorsRight (raw, not undistorted, coordinates); scale_factors as in the frame.
 * max_disparity[p] = mbf / mb (:664-666).  For left keypoint iL of pair p:
 *   best_dist[p*stride+iL] = least Hamming distance below TH_HIGH (100 when none), best_idx_r[...] = that right
 *   keypoint or -1; candidates = right keypoints whose row band [floor(y-2*sf), ceil(y+2*sf)] holds row (int)vL,
 *   octave within +-1, uL - max_disparity <= uR <= uL, ascending index, strict '<'.
 * Entries between the left frame's n and stride are written as no match (-1, 100).  The caller continues with :734 (SAD refinement) where
 * best_dist < (TH_HIGH+TH_LOW)/2.  Row bands reaching outside the image are undefined behaviour in the reference;
 * here they simply never match a row that does not exist. */
int orbm_stereo_match_batch(orbm_ctx *ctx, int n_pairs, const int32_t *frame_left, const int32_t *frame_right,
                            const float *max_disparity, int32_t stride, int32_t *best_idx_r, int32_t *best_dist,
                            int io_on_device);

/* ---- MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:363-449), batched over map points.
 * Point p owns descriptors desc[32*off[p] .. 32*off[p+1]) (the rows of its non-bad observing keyframes, in the
 * order the reference's map iteration yields them).  best_idx[p] = index, within the point's own list, of the
 * descriptor with the least median Hamming distance to all of them (median = element (int)(0.5*(N-1)) of the sorted
 * row, first row wins ties), or -1 for an empty list (the reference returns without touching mDescriptor).  The
 * caller then sets mDescriptor = that row.  Replaces the per-point calls at LocalMapping.cc:289,780,926,
 * LoopClosing.cc:918, Tracking.cc:673,864,1461.  Any number of observations per point (up to 512 a point is handled in
 * registers + LDS, beyond that by a slower streaming form of the same kernel; same results). */
int orbm_distinctive_descriptors_batch(orbm_ctx *ctx, int n_points, const int32_t *off, const uint8_t *desc,
                                       int32_t *best_idx, int io_on_device);

/* ---- BASELINE config 2: brute-force 2-NN + threshold + ratio test over keyframe pairs --------
 * For pair p and every row i of frame_a[p]: scan all rows j of frame_b[p] in ascending j with the
 * strict-< best / second-best update of ORBmatcher.cc:290-300, accept iff best <= th and
 * (float)best < ratio*(float)second (ORBmatcher.cc:304-308).
 * match[p*stride+i] = j or -1 (rows between the a-frame's n and the stride: -1, on both I/O paths); best_d / second_d
 * (nullable) get the raw distances (256 = none; their rows beyond n are not written);
 * counts[p] = accepted rows.  stride >= max n of the a-frames.
 * The distances are computed on the matrix cores (every descriptor bit as +-1 in FP4, expanded in registers from the
 * frames' raw rows; exact) and are identical to the xor + popcount form, which the environment variable
 * ORBMATCH_BF2NN_KERNEL=valu selects instead.  ORBMATCH_BF2NN_KERNEL=lds selects round 1's form of the same product, which
 * keeps a pre-expanded image of each frame (128 B per descriptor, built on the device the first time a frame is used).
 * Large batches (about a thousand 2000 x 2000 pairs and more) that ask for matches only (best_d == second_d == NULL)
 * under an acceptance rule whose bound stays below 73 (th 50 at ratio 0.75: 67) multiply the last 64 bits of a comparison
 * only for blocks of 32 candidates in which some 192-bit partial distance is already below its bound (identical matches,
 * about a fifth faster); ORBMATCH_BF2NN_KERNEL=full multiplies every bit of every comparison regardless. */
int orbm_bf2nn_batch(orbm_ctx *ctx, int n_pairs, const int32_t *frame_a, const int32_t *frame_b,
                     float ratio, int th, int32_t stride, int32_t *match, int32_t *best_d,
                     int32_t *second_d, int32_t *counts, int io_on_device);
/* Only for the forms that keep a +-1 image per frame (ORBMATCH_BF2NN_KERNEL=lds, the int8 form): builds the images of every
 * live frame NOW (one launch + one stream synchronisation) instead of at the first orbm_bf2nn_batch call that meets a
 * frame.  A no-op for the default kernel. */
int orbm_bf2nn_prepare(orbm_ctx *ctx);

/* ---- BASELINE config 5 / SURVEY 8(e): batched keyframe-pair matching that leaves COMPACT match records ----------------
 * The loops this replaces: LoopClosing.cc:433-477 (SearchByBoW against every consistent candidate, a pair is kept only
 * if nmatches >= 20, :459) and Tracking.cc:1778-1801 (relocalisation candidates, >= 15, :1787); all-pairs matching is
 * the same loop over the whole pair triangle.  Record format (what travels to the host or to rank 0):
 *   summaries[p] = { pair_id[p] (or p when pair_id is NULL), nmatches, 0 }          8 bytes for EVERY pair
 *   lists        = (u16 row | u16 value << 16) x nmatches, ascending row, ONLY for pairs with nmatches >= threshold
 *                  (and > 0), pair after pair in batch order: a pair's list starts at the running sum of the accepted
 *                  pairs' nmatches before it.  row / value = index / content of the pair's match row (idx1 -> idx2 for
 *                  the brute-force search and SearchByBoW(KF,KF); idxF -> idxKF for SearchByBoW(KF,F)).
 *   totals[0] = number of accepted pairs, totals[1] = list entries of the batch.  If totals[1] > list_cap the lists
 *   were cut at list_cap entries (the summaries are complete): call again with a larger buffer. */
#define ORBM_PAIR_PAD 0xFFFFFFFFu   /* pair_id of a padding entry (fixed-size batches): reported with n = 0 and never listed */
typedef struct orbm_pair_summary {
    uint32_t pair_id;
    uint16_t n;      /* nmatches of the pair */
    uint16_t pad;    /* 0 */
} orbm_pair_summary;
/* Compacts match rows + counts as the searches above return them (entries of a row beyond its frame's n must be -1). */
int orbm_pack_records(orbm_ctx *ctx, int n_pairs, const uint32_t *pair_id, const int32_t *counts, const int32_t *match,
                      int32_t stride, int threshold, orbm_pair_summary *summaries, uint32_t *lists, uint32_t list_cap,
                      uint32_t *totals, int io_on_device);
/* Match + compact in one call; the match rows stay in a workspace of the context.  kind 0 = brute-force 2-NN
 * (ratio, th as orbm_bf2nn_batch), 1 = SearchByBoW(KF,KF), 2 = SearchByBoW(KF,F) (ratio, check_ori as orbm_search_bow_batch).
 * stride >= the rows a pair needs (n of the a-frame; of the b-frame for kind 2), at most 65536.  The VALUE half of a
 * list entry indexes the pair's other frame (the b-frame; the a-frame for kind 2), which must have at most 65536
 * keypoints as well: checked (ORBM_ELIMIT) when the pair lists are host memory; with io_on_device the ids cannot be read
 * here and the limit is the caller's contract. */
int orbm_allpairs_batch(orbm_ctx *ctx, int kind, int n_pairs, const int32_t *frame_a, const int32_t *frame_b,
                        const uint32_t *pair_id, float ratio, int th, int check_ori, int32_t stride, int threshold,
                        orbm_pair_summary *summaries, uint32_t *lists, uint32_t list_cap, uint32_t *totals, int io_on_device);
/* The same over SEVERAL GPUs from one host process (SURVEY 8(b) `orbm_allpairs(ctx[], n_gpus, ...)`; the reference's
 * candidate loops LoopClosing.cc:433-477, Tracking.cc:1778-1801 run over shards).  ctxs[0 .. n_ctx): one context per GPU (or
 * several on one GPU), handles[c][k] = handle of keyframe k in context c (every context holds the keyframes it is given;
 * a store replicated with orbm_frames_upload in each context has the same handles everywhere).  kf_a / kf_b name
 * keyframes by k; all arrays are HOST memory.  The pair list is cut into n_ctx contiguous slices, one worker thread per
 * context matches its slice (orbm_allpairs_batch) and the records are concatenated in pair order: the result equals one
 * context's for the whole list.  No collective is involved.  Errors: the failing context's code; its message is copied to
 * ctxs[0]. */
int orbm_allpairs(orbm_ctx *const *ctxs, int n_ctx, const int32_t *const *handles, int kind, int n_pairs, const int32_t *kf_a,
                  const int32_t *kf_b, const uint32_t *pair_id, float ratio, int th, int check_ori, int32_t stride, int threshold,
                  orbm_pair_summary *summaries, uint32_t *lists, uint32_t list_cap, uint32_t *totals);
/* Block-cyclic sharding of the pair triangle {(a, b): 0 <= a < b < n_kf} over `world` GPUs (host arithmetic, no context):
 * the triangle is cut into tile x tile blocks of keyframes (row-major over the upper-triangular blocks), block k belongs to
 * rank k % world, pairs inside a block ascend in a, then b: a block's pairs touch 2 x tile keyframes, whose descriptors sit
 * in one L2.  Returns the number of pairs of the rank's shard; with pair_a / pair_b non-NULL fills entries
 * [first, first + max_pairs) of the shard's enumeration.  No data-path collective: every GPU holds the whole store. */
int64_t orbm_allpairs_shard(int n_kf, int tile, int rank, int world, int64_t first, int64_t max_pairs, int32_t *pair_a,
                            int32_t *pair_b);

/* ---- a7 / a8: SearchByBoW (ORBmatcher.cc:218-385 and 703-854), batched over pairs -------------
 * variant 0 = (KeyFrame*, Frame&): a = KF, b = F; match[p*stride + idxF] = idxKF whose map point
 *             is assigned to F's keypoint, or -1 (vpMapPointMatches, ORBmatcher.cc:312).
 * variant 1 = (KeyFrame*, KeyFrame*): match[p*stride + idx1] = idx2 or -1 (vpMatches12, :798).
 * Validity comes from the frames' mp_flags.  counts[p] = returned nmatches.
 * stride >= n of every b-frame (variant 0) / a-frame (variant 1). */
int orbm_search_bow_batch(orbm_ctx *ctx, int variant, int n_pairs, const int32_t *frame_a,
                          const int32_t *frame_b, float ratio, int check_ori, int32_t stride,
                          int32_t *match, int32_t *counts, int io_on_device);

/* ---- allocation-free variant of orbm_search_bow_batch.  NO counterpart in the reference: it exists for callers that cannot
 * allocate device memory (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  Same
 * results as orbm_search_bow_batch(..., io_on_device = 1): frame_a, frame_b, match, counts are DEVICE pointers, the
 * call only enqueues work on the context's stream, and it calls no hipMalloc / hipFree (the context's workspaces are
 * not touched).  The per-pair accumulators (rotation histogram + match count) live in a static device array of the
 * library, 256 KB per GPU, held until the process ends: at most 2048 pairs per call (ORBM_ELIMIT beyond).  Every call
 * on one GPU shares that array: calls that overlap on one GPU (two contexts / streams) race; calls on one stream do not. */
int orbm_search_bow_batch_noalloc(orbm_ctx *ctx, int variant, int n_pairs, const int32_t *frame_a,
                                  const int32_t *frame_b, float ratio, int check_ori, int32_t stride,
                                  int32_t *match, int32_t *counts);

/* ---- a9: SearchForTriangulation (ORBmatcher.cc:868-1093), batched over pairs ------------------
 * f12: n_pairs x 9 row-major F12; epipole: n_pairs x 2 (ex, ey) of ORBmatcher.cc:883-884 (the
 * 3-D pose arithmetic stays on the host).  match12[p*stride + idx1] = idx2 or -1. */
int orbm_search_triang_batch(orbm_ctx *ctx, int n_pairs, const int32_t *frame_a,
                             const int32_t *frame_b, const float *f12, const float *epipole,
                             int only_stereo, int check_ori, int32_t stride, int32_t *match12,
                             int32_t *counts, int io_on_device);

/* ---- a6: SearchByProjection(Frame&, const vector<MapPoint*>&, th) (ORBmatcher.cc:57-166) ------
 * Instance k matches query set qset[k] into frame frame[k].  slot_flags (n_inst x stride bytes of
 * ORBM_MP_*, nullable => the frames' own mp_flags) is the initial state of the slot vector the
 * method reads and writes (F.mvpMapPoints; skip iff PRESENT and OBS, ORBmatcher.cc:113-115).
 * owner[k*stride + idx] = index of the query whose map point ends up in F.mvpMapPoints[idx]
 * (last writer), or -1 if the call did not write the slot.  counts[k] = returned nmatches. */
int orbm_search_proj_points_batch(orbm_ctx *ctx, int n_inst, const int32_t *frame, const int32_t *qset,
                                  float th, float ratio, const uint8_t *slot_flags, int32_t stride,
                                  int32_t *owner, int32_t *counts, int io_on_device);

/* ---- a11: the best-1 windowed overloads, one kernel, three modes -----------------------------
 * mode 0 = (Frame& Cur, const Frame& Last, th, bMono)  ORBmatcher.cc:1768-1950: Frame grid with level
 *          arguments, slot skipped iff PRESENT and OBS, stereo gate fabs(ur-uRight) > radius,
 *          accept best <= dist_th (TH_HIGH), rotation histogram pushes the slot;
 * mode 1 = (Frame& Cur, KeyFrame*, sAlreadyFound, th, ORBdist) :1963-2123: slot skipped iff PRESENT,
 *          accept best <= dist_th (ORBdist), rotation histogram pushes the slot;
 * mode 2 = (KeyFrame*, Scw, vpPoints, vpMatched, th) :401-542: KeyFrame grid (no level argument,
 *          KeyFrame.cc:766-812) then octave in [min_level,max_level], slot skipped iff PRESENT,
 *          accept best <= dist_th (TH_LOW), no histogram.
 * slot_flags: as for orbm_search_proj_points_batch (CurrentFrame.mvpMapPoints / vpMatched).
 * owner[k*stride + idx] = query index whose map point ends up in the slot; -1 = the call never wrote the
 * slot; -2 = the slot was written and then set to NULL by the rotation filter (:1942, :2115). */
int orbm_search_window_batch(orbm_ctx *ctx, int mode, int n_inst, const int32_t *frame,
                             const int32_t *qset, int dist_th, int check_ori,
                             const uint8_t *slot_flags, int32_t stride, int32_t *owner,
                             int32_t *counts, int io_on_device);

/* ---- a12: claim-free best-1 windowed search: the inner loops of SearchBySim3 (ORBmatcher.cc:1586-1626,
 * 1676-1710) and of both Fuse overloads (:1185-1257, :1390-1421).  KeyFrame grid (KeyFrame.cc:766-812), octave in
 * [min_level, max_level], strict-< best-1; queries are independent (no slot state is read or written).
 * chi2_gate != 0 adds Fuse's reprojection gate (:1212-1238): with uRight >= 0,
 * (ex^2+ey^2+er^2)*invSigma2[octave] > 7.8 skips, else (ex^2+ey^2)*invSigma2[octave] > 5.99 skips (ur from the
 * query set).  best_idx[k*stride_q + q] = keypoint index if its distance <= dist_th, else -1.  The pointer-graph
 * side effects of Fuse (Replace / AddObservation, :1259-1283) are replayed on the host in query order. */
int orbm_search_window_free_batch(orbm_ctx *ctx, int n_inst, const int32_t *frame, const int32_t *qset, int dist_th,
                                  int chi2_gate, int32_t stride_q, int32_t *best_idx, int io_on_device);
/* The same for query sets that are used once (the two directions of one SearchBySim3 call :1586-1710, the projections of one
 * Fuse call :1185-1257 / :1390-1421): queries[k] travels with the call (host arrays, at most stride_q queries each; one
 * pinned copy in, one out, one synchronisation; nothing uploaded beforehand, nothing to free).  Results as
 * orbm_search_window_free_batch on uploaded copies of the same sets. */
int orbm_search_window_free_once(orbm_ctx *ctx, int n_inst, const int32_t *frame, const orbm_win_queries *queries, int dist_th,
                                 int chi2_gate, int32_t stride_q, int32_t *best_idx);

/* ---- a6 / a11 as ONE call per search: what a tracking thread issues per frame (Tracking.cc:1152, 1549-1560, 1865 -- one
 * fresh query set into one resident frame).  The query arrays travel with the call (one pinned host->device copy together
 * with the id lists, nothing is allocated or freed, one stream synchronisation); results equal orbm_*_queries_upload +
 * orbm_search_proj_points_batch / orbm_search_window_batch with n_inst = 1 + orbm_queries_free.  frame: a handle of
 * orbm_frames_upload; slot_flags: NULL or n bytes (n = the frame's keypoints) replacing its mp_flags for this search;
 * owner: n ints out; count: nmatches out.  All pointers are HOST memory. */
int orbm_search_proj_points_once(orbm_ctx *ctx, int32_t frame, const orbm_proj_queries *queries, float th, float ratio,
                                 const uint8_t *slot_flags, int32_t *owner, int32_t *count);
int orbm_search_window_once(orbm_ctx *ctx, int mode, int32_t frame, const orbm_win_queries *queries, int dist_th, int check_ori,
                            const uint8_t *slot_flags, int32_t *owner, int32_t *count);

/* ---- a5: SearchForInitialization (ORBmatcher.cc:562-689) --------------------------------------
 * prev_xy: n1 x 2 floats, vbPrevMatched, in/out (:684-686); matches12: n1 ints, -1 = none.
 * Host pointers.  Returns nmatches. */
int orbm_search_init(orbm_ctx *ctx, int32_t frame1, int32_t frame2, float *prev_xy, int window,
                     float ratio, int check_ori, int32_t *matches12);

/* ---- measurement helper (K0 of SURVEY appendix B): sustained v_xor_b32 + v_bcnt_u32_b32 rate ---
 * Runs `iters` unrolled xor+bcnt pairs per lane on every SIMD and returns lane-ops/s in *rate. */
int orbm_popcount_peak(orbm_ctx *ctx, int iters, double *lane_ops_per_s);

#ifdef __cplusplus
}
#endif
#endif /* ORBMATCH_H */
