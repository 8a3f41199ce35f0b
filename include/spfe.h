/*
 * spfe.h — C ABI of libspfe.so, the MI355X-native SuperPoint feature front-end.
 *
 * Drop-in boundary for ONE path of HyHuang1995/sp_orb_slam: the extractor call
 *     (*mpORBextractorLeft)(im, cv::Mat(), mvKeys, mDescriptors)
 * (orb_slam2/src/type/frame.cpp:296-314), i.e. SPExtractor::operator()
 * (orb_slam2/src/cv/sp_extractor.cpp:361-514) and everything below it
 * (SPFrontend::forward :79-159, nms :161-250, computeCovariance :252-340).
 * Plain pointers and sizes only: no torch, OpenCV or Eigen types.  The C++
 * adaptor that restores the BaseExtractor signature
 * (include/orb_slam/cv/base_extractor.h:54-56) is include/spfe_extractor.hpp;
 * INTEGRATION.md shows the reference-side change.
 *
 * All functions return SPFE_OK (0) or a negative SPFE_E* code; the message for
 * the last failure on the calling thread is spfe_last_error().
 */
#ifndef SPFE_H
#define SPFE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define SPFE_API __attribute__((visibility("default")))
#else
#define SPFE_API
#endif

#define SPFE_OK 0
#define SPFE_EINVAL (-1)   /* bad argument / size not a multiple of 8 / wrong batch */
#define SPFE_EEMPTY (-2)   /* null or empty image: sp_extractor.cpp:364-365 throws here */
#define SPFE_EHIP (-3)     /* HIP runtime error (no GPU, OOM, launch failure) */
#define SPFE_EWEIGHTS (-4) /* weight blob/file missing or malformed */

/* spfe_config.precision */
#define SPFE_PRECISION_F32 0  /* exact f32 on v_mfma_f32_32x32x2_f32; bit-identical to the CPU oracle */
#define SPFE_PRECISION_BF16 1 /* ALL twelve convolutions — conv1a..conv4b, convPa / convDa and both 1x1 heads convPb /
                                 convDb — on v_mfma_f32_32x32x16_bf16: bf16 activations and weights, f32 accumulate /
                                 bias / ReLU / pool.  The heads read the bf16 ReLU(convPa) / ReLU(convDa) buffer and
                                 write f32 logits / coarse descriptors, so the detector logits carry bf16 rounding.
                                 Softmax, NMS, descriptor sampling, heat and covariance stay f32 (BASELINE configs[3]:
                                 "bf16 conv path with fp32 NMS"): everything behind the logits is bit-exact GIVEN the
                                 logits.  Keypoints / descriptors match the f32 path within tolerance, not bitwise
                                 (keypoint-set Jaccard ~0.93, descriptor cosine >= 0.9999 at 1280x720). */

/* spfe_config.flags */
#define SPFE_FLAG_HEAT 1u /* also produce heat / heat_inv (H*W floats each), sp_extractor.cpp:461-474 */
#define SPFE_FLAG_ASYNC_COV 2u /* spfe_extract_batch_device only: the covariance stage (cov2, cov2_inv, status of
                                  the records) runs on a library-owned side stream and is NOT ordered into the
                                  caller's stream by the call; order it with spfe_wait_records(ticket) before
                                  reading the records.  Lets the latency-bound covariance of batch i overlap the
                                  convolutions of batch i+1.  Pass a different record buffer to consecutive calls. */

#define SPFE_FLAG_DESC_BF16 4u /* the records (and spfe_result) carry the descriptors as bf16 [kmax][256] — the f32
                                  descriptor of sp_extractor.cpp:512-513 rounded to nearest even — instead of f32: a record
                                  shrinks from ~1.1 MB to ~0.6 MB (1000 features), and with it the D2H copy of the host
                                  calls and the all-gather of the multi-GPU path.  spfe_result.desc is NULL then and
                                  spfe_result.desc_bf16 is set; spfe_record_layout.desc_elem_bytes says which.  Everything
                                  else in the record is unchanged.  The entry points that read descriptors from records
                                  (spfe_match_records_device, spfe_match_patches_record_device,
                                  spfe_track_dust_record_device) widen the rows on load: distances are the f32 arithmetic
                                  of the reference on those rounded values. */

#define SPFE_FLAG_LAZY_HEAT_INV 8u /* with SPFE_FLAG_HEAT: the host calls (spfe_extract*, spfe_submit_batch) bring back
                                     `heat` only; spfe_result.heat_inv is NULL and spfe_fetch_heat_inv() copies a frame's
                                     map on demand.  In the reference heat_inv_ is read by computeCovariance alone
                                     (sp_extractor.cpp:508) — which runs on the device here — and by no caller (SURVEY.md
                                     §8b "public but unread elsewhere"), so its 4 H W bytes per frame need not cross PCIe
                                     on every call (752x480: 1.44 MB of the 4.0 MB a call with heat maps brings back). */

/* spfe_result.status / record header word 2 */
#define SPFE_STATUS_COV_OVERFLOW 1 /* Set only when ONE covariance region has more pops than the device's last-resort list
                                      holds (SPFE_COV_FALLBACK_CAP, 4 M by default — the reference's own loop would spend
                                      ~0.1 s in that one BFS): cov2 / cov2_inv of that frame are then not valid.  Everything
                                      short of that is handled on the device, exactly, with status 0, for host calls, device
                                      records and all-gathered records alike: a walk longer than SPFE_COV_QCAP = 1024 pops
                                      reruns in one of SPFE_COV_OVF_SLOTS = 16 lists of SPFE_COV_OVF_CAP = 16384 pops per
                                      frame; a frame that exhausts those (or whose hills leave the staged window) is redone
                                      sequentially by cov_fallback_kernel.  The library has no host compute routine. */

/* ABI of this header.  Bumped whenever a struct below changes size or layout or an entry point changes its signature
 * (4: spfe_result.desc_bf16, spfe_record_layout.desc_elem_bytes — round 3 — and this check).  A caller built against an older
 * header would hand the library arrays of the wrong stride; spfe_check_abi(SPFE_ABI_VERSION, sizeof(spfe_config),
 * sizeof(spfe_result), sizeof(spfe_record_layout)) refuses that up front (the C++ adaptor and the Python loader call it). */
#define SPFE_ABI_VERSION 5

#define SPFE_DESC_DIM 256
#define SPFE_NUM_PARAMS 1300865 /* sp_extractor.cpp:16-43; order = register_module order :46-62 */

typedef struct spfe_handle_s *spfe_handle;

/*
 * Replaces the SPExtractor constructor (sp_extractor.cpp:342-359), which reads
 * tracking::num_features (tracker.cpp:131), camera::height/width and
 * common::model_path (:354-355) from globals.
 */
typedef struct {
  int height;               /* camera::height, multiple of 8 (:70); height x width: up to 262,143 cells of 8x8 and 2^31 bytes of
                               first-layer activations (64 channels a pixel, 4 / 2 bytes each in f32 / bf16 mode) per frame:
                               3840x2160 fits in both */
  int width;                /* camera::width, multiple of 8 */
  int num_features;         /* tracking::num_features; up to num_features+1 keypoints (:211-213); 1 .. 10000 (the
                               covariance link stage keeps 16 bytes per keypoint in one workgroup's LDS; the
                               shipped configurations use 800 - 1000) */
  int max_batch;            /* frames per spfe_extract_batch* call (>=1) */
  int device;               /* HIP device ordinal */
  int precision;            /* SPFE_PRECISION_* */
  unsigned flags;           /* SPFE_FLAG_* */
  const float *weights;     /* flat fp32 blob, SPFE_NUM_PARAMS floats, or NULL */
  const char *weights_path; /* "SPFW" file (sp_orb_slam_amd/weights.py), used if weights==NULL */
} spfe_config;

/*
 * Everything the caller reads after operator() (frame.cpp:296-314): keypoints,
 * descriptors and the SPExtractor side outputs (sp_extractor.h:61-73).
 * Pointers are library-owned host buffers, valid until the next call on the
 * same handle (the reference hands out cv::Mat headers over freed tensor
 * storage, sp_extractor.cpp:432-433,448-451; this does not).
 */
typedef struct {
  int K;                    /* keypoints emitted, raster order (:220-238) */
  int n_candidates;         /* cells with score >= 0.007 (:122) */
  int status;               /* 0, or SPFE_STATUS_* bits */
  int reserved;
  const float *kp_xy;       /* [K][2] pt.x, pt.y (integer valued); size=1, octave=0, angle=-1 (:231-232) */
  const float *kp_response; /* [K] heat_inv at the keypoint (:271) */
  const float *desc;        /* [K][256] unit-L2 rows, CV_32FC1 (:512-513) */
  const float *cov2;        /* [K][2] (:332) */
  const float *cov2_inv;    /* [K][2] getCov2Inv() (sp_extractor.h:67) */
  const int16_t *occ_grid;  /* [H/8][W/8] CV_16SC1, -1 = empty (:178,227-228) */
  const float *dense_dust;  /* [H/8][W/8] softmax dustbin (:107,450) */
  const float *semi_dust;   /* [H/8][W/8] raw dustbin logit (:106,448) */
  const float *heat;        /* [H][W] or NULL without SPFE_FLAG_HEAT (:467) */
  const float *heat_inv;    /* [H][W] or NULL without SPFE_FLAG_HEAT / with SPFE_FLAG_LAZY_HEAT_INV (:468) */
  const uint16_t *desc_bf16; /* [K][256] bf16 bit patterns with SPFE_FLAG_DESC_BF16 (then desc is NULL), else NULL */
} spfe_result;

SPFE_API int spfe_create(const spfe_config *cfg, spfe_handle *out);
SPFE_API void spfe_destroy(spfe_handle h);

/* SPExtractor::operator() for one CV_8UC1 frame of the configured size.
 * `stride` = bytes between rows (cv::Mat::step). */
SPFE_API int spfe_extract(spfe_handle h, const uint8_t *image, int stride, spfe_result *out);

/* n independent frames (n <= max_batch); outs[i] valid until the next call. */
SPFE_API int spfe_extract_batch(spfe_handle h, const uint8_t *const *images, int stride, int n,
                       spfe_result *outs);

/* The synchronous call in three parts, for a caller that copies the outputs out of the library's buffers (the drop-in
 * class does: the reference's members are deep cv::Mat copies, sp_extractor.cpp:436-474): the two H x W maps are complete
 * after the network's tail, ~0.15 ms before the record (selection, sampling, covariance follow), and their D2H runs beside
 * that work (SPFE_EARLY_HEAT_COPY) — so the caller's own copy of the maps can run beside it too.
 *   spfe_extract_begin   what spfe_extract_batch does up to the end of its enqueueing; returns at once
 *   spfe_extract_maps    blocks until the maps asked for (either argument may be NULL: not asked for) are in host
 *                        memory; *heat / *heat_inv = frame 0's map, frame i at + i * H * W floats (*heat_inv = NULL with
 *                        SPFE_FLAG_LAZY_HEAT_INV).  `heat` arrives first, `heat_inv` behind it: a caller copying both asks
 *                        for heat alone, copies it, then asks for heat_inv.  NULL results (and SPFE_OK) when the maps do not
 *                        travel ahead of the record in this call (no SPFE_FLAG_HEAT, SPFE_EARLY_HEAT_COPY=0):
 *                        spfe_extract_finish delivers them as spfe_extract_batch does.  Optional, any number of times.
 *   spfe_extract_rows    blocks until frame `frame`'s descriptor rows are in host memory — final behind the sampling, while
 *                        the covariance of the call still runs: *K rows of 256 floats at *desc (where spfe_result.desc will
 *                        point).  *desc = NULL (and SPFE_OK) when the rows travel with the record in this call (batches on
 *                        the side-stream chain, SPFE_FLAG_DESC_BF16, SPFE_EARLY_HEAT_COPY=0).  Optional.
 *   spfe_extract_finish  the rest of spfe_extract_batch: blocks, fills outs[0 .. n) (same pointers, same lifetime)
 * begin + finish == spfe_extract_batch, bit for bit.  Between the two no other call on the handle (SPFE_EINVAL from begin
 * while a call is open, from maps / finish when none is). */
SPFE_API int spfe_extract_begin(spfe_handle h, const uint8_t *const *images, int stride, int n);
SPFE_API int spfe_extract_maps(spfe_handle h, const float **heat, const float **heat_inv);
SPFE_API int spfe_extract_rows(spfe_handle h, int frame, int *K, const float **desc);
SPFE_API int spfe_extract_finish(spfe_handle h, spfe_result *outs);

/* Pipelined host path.  spfe_extract_batch is synchronous like the reference's operator() (upload
 * sp_extractor.cpp:379-390, blocking D2H :427-433).  A host that has the next frames while the current ones
 * are being processed (a dataset player, a multi-camera rig, the batch path) submits instead:
 *   spfe_submit_batch   copies the frames into pinned staging, enqueues H2D (copy stream), the whole path
 *                       (compute + side streams) and the D2H of the records (+ heat maps with SPFE_FLAG_HEAT;
 *                       second copy stream), and returns at once with a ticket;
 *   spfe_collect_batch  blocks until that batch is back in host memory; outs[i] are valid until three
 *                       further batches have been submitted.
 * Up to 3 batches may be in flight (submit fails with SPFE_EINVAL when the oldest has not been collected), so
 * the H2D of batch i + 1 and the D2H of batch i - 1 overlap the compute of batch i.  Collect in any order. */
SPFE_API int spfe_submit_batch(spfe_handle h, const uint8_t *const *images, int stride, int n, long *ticket);
SPFE_API int spfe_collect_batch(spfe_handle h, long ticket, spfe_result *outs);

/* Everything after the network (sp_extractor.cpp:105-148 detector tail and
 * descriptor sampling, :461-514 host glue, nms, computeCovariance) for n frames
 * whose raw head outputs the caller provides as HOST arrays: semi [n][H/8][W/8][65]
 * (convPb logits, channels last) and coarse [n][H/8][W/8][256] (convDb output,
 * not normalised).  Used by the tests to drive the selection kernels with
 * hand-made logits; also the entry for a caller with its own network. */
SPFE_API int spfe_postprocess(spfe_handle h, const float *semi, const float *coarse, int n,
                              spfe_result *outs);

/*
 * Device-resident batch path (multi-GPU pipeline): d_images is a DEVICE pointer
 * to n contiguous u8 frames [n][H][W]; d_records is a DEVICE buffer of
 * n * spfe_record_bytes(h) bytes that receives one fixed-stride record per
 * frame (layout: spfe_record_layout).  Work is enqueued on `stream`
 * (hipStream_t, NULL = the handle's own stream) and NOT synchronised: the caller
 * may all-gather d_records with RCCL on the same stream.
 */
typedef struct {
  size_t bytes;    /* record stride, multiple of 256 */
  int kmax;        /* num_features + 1 */
  size_t off_hdr;  /* int32 K, int32 n_candidates, int32 status, int32 reserved */
  size_t off_xy;   /* float [kmax][2] */
  size_t off_resp; /* float [kmax] */
  size_t off_cov;  /* float [kmax][2] */
  size_t off_cinv; /* float [kmax][2] */
  size_t off_desc; /* float [kmax][256]; bf16 [kmax][256] with SPFE_FLAG_DESC_BF16 */
  size_t off_occ;  /* int16 [H/8][W/8] */
  size_t off_dd;   /* float [H/8][W/8] dense_dust */
  size_t off_sd;   /* float [H/8][W/8] semi_dust */
  int desc_elem_bytes; /* 4, or 2 with SPFE_FLAG_DESC_BF16 */
} spfe_record_layout;

SPFE_API int spfe_get_record_layout(spfe_handle h, spfe_record_layout *out);
SPFE_API size_t spfe_record_bytes(spfe_handle h);
SPFE_API int spfe_extract_batch_device(spfe_handle h, const void *d_images, int n, void *d_records,
                              void *stream);
/* (Batches of >= 2 frames issue the layers behind conv1b as two half batches, the second on a library-owned stream that must
 * not share a hardware queue with `stream`: the first call that brings a new `stream` measures that with two 150 us spin
 * kernels that time-stamp themselves on the device clock, and synchronises `stream` once while doing so (a stream under
 * capture is not probed: no split).  The answer is kept per hipStream_t value for the life of the handle — a stream destroyed
 * and re-created at the same address inherits it (a performance matter only).  spfe_debug_read("split_streams") reports the
 * outcome.  SPFE_F32_SPLIT=0 / SPFE_F32_SPLIT_PROBE=0 switch the split / the measurement off.) */
/* Ticket of the most recent spfe_extract_batch_device call on this handle (0, 1, 2, ...), and the
 * ordering point for SPFE_FLAG_ASYNC_COV: makes `stream` (NULL = the handle's stream) wait until the
 * records of call `ticket` (one of the last 4 calls) AND OF EVERY EARLIER CALL are complete (also where the handle runs two
 * side chains on two sets of buffers — large bf16 frames, DESIGN.md 5.1: tickets stay one sequence).  Without the flag the
 * call itself does this and spfe_wait_records is a no-op dependency. */
SPFE_API long spfe_last_ticket(spfe_handle h);
SPFE_API int spfe_wait_records(spfe_handle h, long ticket, void *stream);

/* ---- multi-GPU batch path: RCCL all-gather of the records (BASELINE configs[2]) ----------
 * One process per GPU, one handle per process.  Frames are independent, so a batch shards over ranks with no
 * collective in the data path; the ONLY exchange is this all-gather of the fixed-stride records (K lives in
 * the record header, so no count exchange).  The C++ SLAM host uses these directly; no torch involved.
 *   rank 0: spfe_comm_unique_id(id) -> ship the 128 bytes to every rank (MPI, a socket, a file ...)
 *   every rank: spfe_comm_init(h, id, rank, world)            [ncclCommInitRank on the handle's device]
 *   per batch:  spfe_extract_batch_device(h, imgs, n, d_local, stream); t = spfe_last_ticket(h);
 *               spfe_allgather_records(h, t, d_local, d_all, n);   [d_all: world * n * spfe_record_bytes(h)]
 *               spfe_comm_wait(h, consumer_stream);                [or hipStreamSynchronize(spfe_comm_stream(h))]
 * The collective is issued on the library's side stream, right behind the covariance kernels of the batch it
 * gathers: call spfe_allgather_records for batch i BEFORE enqueueing batch i + 1 (with SPFE_FLAG_ASYNC_COV the gather
 * of batch i then overlaps the convolutions of batch i + 1; called later it still is correct, it just queues behind
 * batch i + 1's covariance).  No stream waits in a hardware queue for an event (a waiting stream of its own can land
 * on the compute stream's hardware queue and stall it; SPFE_COMM_OWN_STREAM=1 restores that form).  d_local / d_all
 * must stay untouched until the gather has completed (order the next writer with spfe_comm_wait).  librccl is loaded on first use (dlopen), so single-GPU users of
 * libspfe.so do not need it.  rank-major output: global frame g = rank * n + i. */
#define SPFE_COMM_ID_BYTES 128
SPFE_API int spfe_comm_unique_id(void *id, size_t cap);
SPFE_API int spfe_comm_init(spfe_handle h, const void *id, int rank, int world);
SPFE_API int spfe_comm_destroy(spfe_handle h);
SPFE_API int spfe_allgather_records(spfe_handle h, long ticket, const void *d_local, void *d_all, int frames_per_rank);
SPFE_API int spfe_comm_wait(spfe_handle h, void *stream);
SPFE_API void *spfe_comm_stream(spfe_handle h); /* hipStream_t of the collective, NULL before spfe_comm_init */
SPFE_API int spfe_comm_count(spfe_handle h, int *count); /* ncclCommCount: the rank count RCCL itself reports */

/* Host view of ONE record that the caller copied to host memory. */
SPFE_API int spfe_view_record(spfe_handle h, const void *host_record, spfe_result *out);

/* Test/diagnostic tap: copy an intermediate device buffer of frame `frame` of
 * the last call to host. Names: "semi" [hc][wc][65], "coarse" [hc][wc][256],
 * "heat_log" [H][W], "feat" [hc][wc][128], "act<i>" layer outputs.
 * The descriptor branch (convDa, convDb: sp_extractor.cpp:99-100) computes only the rows of the coarse map that the
 * emitted keypoints' bilinear taps read (:134-148 reads nothing else); "coarse" completes the map first (one dense
 * pass over the last call's activations), "coarse_sparse" is the map as the call left it, "db_total" [1] int /
 * "db_list" ints the cells it computed (b * hc * wc + cell; frame ignored), "da_gathered" [1] int whether convDa ran on
 * those cells only as well.
 * Returns the number of bytes copied or a negative error. */
SPFE_API long spfe_debug_read(spfe_handle h, const char *name, int frame, void *dst, size_t cap);

/* ---- descriptor matching ------------------------------------------------
 * Replaces  cv::BFMatcher::create(cv::NORM_L2, crossCheck)->match(desc_query, matches)  with
 * desc_train added, as called by SPMatcher::SearchByBruteForce (orb_slam2/src/cv/sp_matcher.cpp:
 * 1642-1674; the distance is SPMatcher::DescriptorDistance, :1636-1640 = L2 norm of a - b).
 * Descriptors are rows of 256 floats.  For every query row i: train_idx[i] = matched train row or
 * -1, distance[i] = its L2 distance (FLT_MAX when unmatched) — i.e. cv::DMatch{queryIdx = i,
 * trainIdx = train_idx[i], distance}, with the unmatched queries (which OpenCV omits) marked -1.
 *   cross_check != 0 (the reference's setting): OpenCV's batchDistance rule — every train row votes
 *     for its nearest query (lowest query index on ties); a query is matched to the closest train
 *     row that voted for it (lowest train index on ties), queries without votes stay unmatched.
 *   cross_check == 0: plain nearest train row per query (lowest index on ties).
 * NaN / infinite distances never match.  n_query or n_train == 0: all -1, SPFE_OK. */
SPFE_API int spfe_match(spfe_handle h, const float *query, int n_query, const float *train, int n_train,
                        int cross_check, int32_t *train_idx, float *distance);
/* Replaces  matcher->knnMatch(desc_query, matches, 2)  on a cv::FlannBasedMatcher that holds desc_train — the k = 2
 * search behind the ratio tests of KeyFrame::matchMps (orb_slam2/src/type/keyframe.cpp:447-470, index built in
 * buildIndexesMps :421-445) and SPMatcher's keyframe matching (src/cv/sp_matcher.cpp:195-215, :264-280) — by the
 * EXACT two nearest train rows (the reference's randomised kd-trees, matching::ntree / nchecks, are approximate:
 * this returns what they approximate, so it has no bit-parity target, only the exact-search oracle's).
 * train_idx / distance: [n_query][2], nearest first; ties -> lower train index; -1 / FLT_MAX when n_train < 2
 * (or a distance is NaN / infinite).  Distances as in spfe_match. */
SPFE_API int spfe_match_knn2(spfe_handle h, const float *query, int n_query, const float *train, int n_train,
                             int32_t *train_idx, float *distance);
/* Device-resident form: matches the descriptors of n_pairs query records against n_pairs train
 * records (both arrays of spfe_record_bytes()-strided records in HBM, e.g. the outputs of two
 * spfe_extract_batch_device calls), reading K from the record headers on the device; no host
 * synchronisation.  d_out: n_pairs blocks of spfe_match_out_bytes(h) bytes, each
 * int32 train_idx[kmax] followed by float distance[kmax] (kmax = num_features + 1; entries >= the
 * query record's K are -1 / FLT_MAX).  stream: hipStream_t (NULL = the handle's stream). */
SPFE_API int spfe_match_records_device(spfe_handle h, const void *d_query_records, const void *d_train_records,
                                       int n_pairs, int cross_check, void *d_out, void *stream);
SPFE_API size_t spfe_match_out_bytes(spfe_handle h);

/* Patch-wise association of projected map points — the loop of Tracker::trackFrameDustKFLocal,
 * orb_slam2/src/tracking/tracker_dust.cpp:113-172 (the caller filters `!in_view || isBad()` points out).
 * Map point i sits at dust-map position (mp_uv[2i], mp_uv[2i+1]) in CELL units (dust_proj_u / _v) and
 * carries descriptor mp_desc[i]; it examines the keypoints of cells (floor(u)+du, floor(v)+dv),
 * du, dv in {0,1}, du outer, and takes the one with the smallest L2 distance below max_dist (0.75 in
 * the reference; first one on ties); a taken keypoint is gone for the map points after it (the
 * reference clears its occ_grid cell).  kp_idx[i] = keypoint index or -1.  Distances are
 * (float) cv::norm(a, b, NORM_L2): squared differences accumulated in double.  Cells outside the grid
 * hold no keypoint (the reference does not check).  n_points <= 4096.
 * occ_grid: int16 [height/8][width/8] and kp_desc: [n_keypoints][256] of the frame (spfe_result). */
SPFE_API int spfe_match_patches(spfe_handle h, const float *mp_desc, const float *mp_uv, int n_points,
                                const int16_t *occ_grid, const float *kp_desc, int n_keypoints, float max_dist,
                                int32_t *kp_idx);
/* The same against ONE record resident in HBM (occ_grid, descriptors and K read on the device);
 * d_mp_desc / d_mp_uv / d_kp_idx are device arrays; enqueued on `stream`, no host synchronisation. */
SPFE_API int spfe_match_patches_record_device(spfe_handle h, const void *d_mp_desc, const void *d_mp_uv,
                                              int n_points, const void *d_record, float max_dist, void *d_kp_idx,
                                              void *stream);

/* ---- direct "dust" alignment ------------------------------------------------
 * Replaces  Optimizer::PoseOptimizationDust(Frame *pFrame, const std::vector<MapPoint *> &mps,
 *                                           std::vector<bool> &is_visible)
 * (orb_slam2/src/mapping/optimizer_dust.cpp:170-294, called from Tracker::trackFrameDustKFLocal,
 * tracker_dust.cpp:91): a 6-DoF Levenberg-Marquardt (g2o: OptimizationAlgorithmLevenberg, one VertexSE3Expmap,
 * one EdgeSE3ProjectDustOnlyPose per map point — src/optimization/types_dust_tracking.cpp:37-140 — with
 * RobustKernelHuber(0.9), optimize(40)) that moves the camera pose so that the map points project onto cells
 * with a low dustbin probability.  dense_dust = Frame::dust_ (the extractor's dense_dust_, [H/8][W/8]);
 * points_xyz[i] = mps[i]->GetWorldPos() (3 floats); Tcw = Frame::mTcw (CV_32F 4x4, row-major); fx..cy =
 * Frame::fx.. (full resolution; the /8 and -3.5 of :223-226 are applied inside).
 * Outputs: Tcw_out (what pFrame->SetPose receives, :287), inlier[i] (is_visible[i] / in_view, :262-266: level 0
 * and chi2 <= inlier_chi2), proj_uv[i] = (dust_proj_u, dust_proj_v) of the inliers (:267-268), *n_inlier (the
 * return value), *iterations (optimize()'s).  n <= SPFE_DUST_MAX_POINTS (the tracker keeps 150-200).
 * Degenerate calls, as g2o behaves: max_iterations == 0 evaluates nothing — every edge keeps level 0 / error 0, so all n
 * points are reported as inliers with proj_uv = (0, 0) and the pose is echoed (use >= 1 for meaningful flags; the
 * reference always passes 40); n == 0 has no active edge — the pose is echoed, *iterations = 0, *n_inlier = 0.
 * The arithmetic is pinned by tests/golden/dust_*.npz (an independent f64 numpy / scipy statement).
 * g2o is not part of the reference snapshot: its algorithm is restated (include/spfe_dust_math.h) — results
 * equal the CPU oracle's up to the device's sin / cos in the exponential map. */
#define SPFE_DUST_MAX_POINTS 512
typedef struct {
  float fx, fy, cx, cy;
  int max_iterations;   /* 40 (:243) */
  double huber_delta;   /* 0.9 (:221) */
  double inlier_chi2;   /* 0.9 (:260) */
} spfe_dust_params;
SPFE_API int spfe_align_dust(spfe_handle h, const float *dense_dust, const float *points_xyz, int n,
                             const float *Tcw, const spfe_dust_params *prm, float *Tcw_out, uint8_t *inlier,
                             float *proj_uv, int *n_inlier, int *iterations);
/* The same against the dense_dust of ONE record resident in HBM; d_points_xyz / d_Tcw are device arrays; d_out
 * receives SPFE_DUST_OUT_BYTES: float Tcw_out[16] | int32 n_inlier | int32 iterations | pad to
 * SPFE_DUST_OFF_UV: float proj_uv[512][2] | SPFE_DUST_OFF_INLIER: uint8 inlier[512].  Enqueued on `stream`
 * (NULL = the handle's), no host synchronisation; order it after the record with spfe_wait_records. */
#define SPFE_DUST_OFF_UV 128
#define SPFE_DUST_OFF_INLIER (128 + SPFE_DUST_MAX_POINTS * 8)
#define SPFE_DUST_OUT_BYTES (128 + SPFE_DUST_MAX_POINTS * 9)
SPFE_API int spfe_align_dust_record_device(spfe_handle h, const void *d_record, const void *d_points_xyz, int n,
                                           const void *d_Tcw, const spfe_dust_params *prm, void *d_out,
                                           void *stream);

/* The tracker's per-frame chain behind the extraction, on ONE record resident in HBM — Tracking::trackFrameDustKFLocal,
 * orb_slam2/src/tracking/tracker_dust.cpp:92-172: PoseOptimizationDust(&mCurrentFrame, mps_for_track, is_visible) (:92-94),
 * give up when n_inlier < min_inliers (tracking::dust::th_ninlier, :97-102), else the patch-wise association of the in_view
 * map points at their dust_proj_u / v (:113-172; see spfe_match_patches).  Map point i = d_points_xyz[3 i..] with track
 * descriptor d_mp_desc[256 i..] (MapPoint::getDescTrack()).  d_dust_out receives the SPFE_DUST_OUT_BYTES block above;
 * d_kp_idx[i] (int32) = index of the keypoint map point i takes (mCurrentFrame.mvpMapPoints[idx] = mp), -1 for points that
 * are not in view, find nothing below max_dist (0.75f, :121), or when the alignment had too few inliers.  Two kernels behind
 * each other on `stream`: projections, flags and n_inlier never leave HBM; no host synchronisation.  n <= 512. */
SPFE_API int spfe_track_dust_record_device(spfe_handle h, const void *d_record, const void *d_points_xyz,
                                           const void *d_mp_desc, int n, const void *d_Tcw, const spfe_dust_params *prm,
                                           int min_inliers, float max_dist, void *d_dust_out, void *d_kp_idx, void *stream);

/* The batch path's form: n_frames independent solves in ONE launch, one workgroup each (a single solve is a chain of
 * dependent double-precision operations — latency, not throughput: 256 of them side by side take as long as one).
 * Frame f aligns d_n_points[f] points at d_points_xyz + f * SPFE_DUST_MAX_POINTS * 3 floats, starting from the pose
 * d_Tcw + 16 f, against the dense_dust of record f of d_records (spfe_record_bytes() strided, e.g. the output of
 * spfe_extract_batch_device or the all-gathered array); d_out + f * SPFE_DUST_OUT_BYTES receives the block described
 * above.  All arrays in device memory; enqueued on `stream`, no host synchronisation. */
SPFE_API int spfe_align_dust_batch_device(spfe_handle h, const void *d_records, int n_frames, const void *d_points_xyz,
                                          const void *d_n_points, const void *d_Tcw, const spfe_dust_params *prm,
                                          void *d_out, void *stream);

/* ---- input staging -----------------------------------------------------
 * Replaces, per frame, the host OpenCV sequence in front of the extractor:
 *   cv::remap(mono, mono, m1, m2, cv::INTER_LINEAR)       orb_slam2/src/io/data_loader.cc:519-521
 *   mono(cv::Rect(0, 0, camera::width, camera::height))   orb_slam2/src/system.cpp:160-161
 *   cvtColor(im, im, CV_BGR2GRAY / CV_RGB2GRAY / CV_BGRA2GRAY / CV_RGBA2GRAY)
 *                                                         orb_slam2/src/tracking/mono_tracker.cpp:18-28
 * with one gather kernel that writes the gray H x W frame conv1a reads (OpenCV 3.x integer
 * arithmetic: 5-bit sub-pixel positions, 15-bit bilinear weights, BORDER_CONSTANT 0; gray =
 * (1868 B + 9617 G + 4899 R + 8192) >> 14).  The maps are the CV_32FC1 pair of
 * cv::initUndistortRectifyMap (data_loader.cc:485-486), src_height x src_width, copied to the
 * device once; map_x == map_y == NULL means "no remap" (crop + gray only).  src_height >= height
 * and src_width >= width of the handle (the crop keeps the top-left corner). */
typedef struct spfe_staging {
  int src_height, src_width; /* camera image == map size */
  int channels;              /* 1, 3 or 4 interleaved 8-bit channels (cv::imread gives 3: BGR) */
  int rgb;                   /* 0: blue first (mbRGB == false), 1: red first */
  const float *map_x, *map_y;
} spfe_staging;
SPFE_API int spfe_set_staging(spfe_handle h, const spfe_staging *st);
/* spfe_extract / spfe_extract_batch on raw camera frames (stride in bytes >= src_width * channels) */
SPFE_API int spfe_extract_staged(spfe_handle h, const uint8_t *src, int stride, spfe_result *out);
SPFE_API int spfe_extract_batch_staged(spfe_handle h, const uint8_t *const *srcs, int stride, int n,
                                       spfe_result *outs);
/* Device form: n packed raw frames [n][src_height][src_width][channels] in HBM -> gray frames
 * [n][height][width] (d_gray, ready for spfe_extract_batch_device), enqueued on `stream`. */
SPFE_API int spfe_stage_batch_device(spfe_handle h, const void *d_src, int n, void *d_gray, void *stream);

/* Per-stage GPU time (ms, HIP events recorded on the launch stream around every
 * kernel), averaged over the calls since spfe_stage_reset (the library keeps the
 * last 128 calls).  Enabled by SPFE_STAGE_TIMING=1 in the environment at
 * spfe_create (SPFE_STAGE_TIMING=2: only the dominant kernel, conv1b, is bracketed — two events
 * per call instead of sixteen; the other stages then read 0); returns the number of stages written (names: spfe_stage_name). */
SPFE_API int spfe_stage_times(spfe_handle h, float *ms, int cap);
SPFE_API int spfe_stage_reset(spfe_handle h);
SPFE_API const char *spfe_stage_name(int i);

/* heat_inv (sp_extractor.cpp:468) of frame `frame` of the LAST synchronous host call (spfe_extract, spfe_extract_batch,
 * spfe_extract_staged) on this handle, copied to the library's pinned buffer on demand: *out is a view valid until the next
 * call on the handle.  Meant for handles created with SPFE_FLAG_LAZY_HEAT_INV (without it the map is in spfe_result already;
 * the call works all the same).  SPFE_EINVAL without SPFE_FLAG_HEAT, before the first call, or for a frame the call did not hold. */
SPFE_API int spfe_fetch_heat_inv(spfe_handle h, int frame, const float **out);

/* The H x W maps of the synchronous host calls (spfe_extract*, spfe_postprocess, the three-part call) straight into memory of
 * the caller: heat / heat_inv = max_batch * H * W floats each, page-locked by the library for as long as they are set
 * (hipHostRegister; the caller keeps them allocated until the next spfe_set_map_buffers or spfe_destroy), and what
 * spfe_result.heat / .heat_inv (and spfe_extract_maps, spfe_fetch_heat_inv) point at from then on.  NULL = the library's own
 * buffer for that map again.  For a caller whose outputs are deep copies anyway — the drop-in class keeps heat_ / heat_inv_ as
 * cv::Mat members the reference fills per call (sp_extractor.cpp:461-474): with the members' own storage set here the maps
 * land in them by DMA and the copies (2 x 1.44 MB at 752x480) disappear from the call.  The pipelined host path
 * (spfe_submit_batch) keeps its own buffers.  SPFE_EINVAL without SPFE_FLAG_HEAT or while a call is open. */
SPFE_API int spfe_set_map_buffers(spfe_handle h, float *heat, float *heat_inv);

/* Test hook: evaluates the device forms of spfe_expf(x) and spfe_logf(|x|)
 * (include/spfe_exact_math.h) on n host floats, so tests can compare GPU bits
 * with host bits. */
SPFE_API int spfe_math_probe(const float *in, float *out_exp, float *out_log, int n);

SPFE_API const char *spfe_last_error(void);
SPFE_API const char *spfe_version(void);
SPFE_API int spfe_abi_version(void); /* SPFE_ABI_VERSION of the header the library was built from */
SPFE_API int spfe_check_abi(int abi_version, size_t sizeof_config, size_t sizeof_result, size_t sizeof_record_layout);

#ifdef __cplusplus
}
#endif
#endif /* SPFE_H */
