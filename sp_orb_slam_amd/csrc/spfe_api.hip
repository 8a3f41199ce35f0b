// spfe_api.hip — host side of libspfe.so: handle, weight packing, the per-batch
// launch sequence and the C ABI of include/spfe.h.
//
// One handle = one GPU, one stream, one set of buffers (SURVEY.md §8b
// "Threading"): the object SPExtractor's constructor builds
// (/root/reference/orb_slam2/src/cv/sp_extractor.cpp:342-359) and whose
// operator() (:361-514) the extract calls replace.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <functional>
#include <cfloat>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/spfe.h"
#include "../../include/spfe_exact_math.h"
#include "spfe_kernels.h"

// The few RCCL types and signatures the gather needs, declared here so that building libspfe.so needs no RCCL development
// headers: librccl is dlopen'ed by spfe_comm_init (a single-GPU host never loads it).  Values as in rccl.h (NCCL 2 ABI).
extern "C" {
typedef struct ncclComm *ncclComm_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1 } ncclDataType_t;
typedef ncclResult_t (*pfn_ncclGetUniqueId)(ncclUniqueId *);
typedef ncclResult_t (*pfn_ncclCommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
typedef ncclResult_t (*pfn_ncclCommDestroy)(ncclComm_t);
typedef ncclResult_t (*pfn_ncclCommCount)(const ncclComm_t, int *);
typedef ncclResult_t (*pfn_ncclAllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
typedef const char *(*pfn_ncclGetErrorString)(ncclResult_t);
}


namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess)                                                               \
      return fail(SPFE_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                  __LINE__);                                                            \
  } while (0)

constexpr int kKcAuto = 0;      // layers the K-chain kernel takes by default on single frames: none (measured, conv_f32_kc.hip's header)
constexpr int NSTAGE = 15;
const char *kStageNames[NSTAGE] = {"conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b",
                                   "conv4a", "conv4b", "convPaDa", "convPb", "convDb", "tail",
                                   "select", "post_side", "total"};  // post_side = select + heat_norm + desc + cov (side stream)

struct ConvLayer {
  int cin, cout_real, nblk, ks;
  bool pool, relu, small_tile;
  float *d_w = nullptr, *d_b = nullptr;
  const float *in = nullptr;
  int in_stride = 0, in_choff = 0;
  float *out = nullptr;
  int out_stride = 0, out_choff = 0;
  int H = 0, W = 0;  // input resolution of this layer
};

}  // namespace

struct spfe_handle_s {
  spfe_config cfg{};
  int H = 0, W = 0, hc = 0, wc = 0, C = 0, kmax = 0, B = 0;
  hipStream_t stream = nullptr;
  // covariance runs on a side stream: with SPFE_FLAG_ASYNC_COV it overlaps the next
  // call's convolutions (it is latency bound and uses a handful of CUs)
  hipStream_t side = nullptr;
  static constexpr int NTICKET = 4;
  hipEvent_t ev_post[NTICKET] = {}, ev_cov[NTICKET] = {};
  hipEvent_t ev_db = nullptr;    // launch stream: this call's convDb is done (when it is launched behind the detector tail)
  bool defer_db = true;          // SPFE_DEFER_DB=0: convDb in layer order
  hipEvent_t ev_desc = nullptr;  // side stream: the last call's descriptor sampling (reader of d_coarse) is done
  // f32, batches of >= 2 frames: the layers behind conv1b run as TWO half batches on two streams (SPFE_F32_SPLIT), so that the
  // workgroups of one half's kernel fill the CUs the other half's kernel leaves idle in its last, partial round of work items.
  // (Tried on top and removed: conv1a of call i + 1 on the handle's idle stream beside the later layers of call i — it fits
  // on every CU beside a convolution workgroup, but what it saves as a stage the matrix-bound kernels lose beside it: +-0.)
  // No other stream is created for the convolutions: HIP maps streams onto a few hardware queues, and ONE more stream in the
  // process moved this one onto the launch stream's queue — -4 % instead of +2 %.
  hipStream_t conv2 = nullptr;
  std::vector<hipStream_t> conv2_pool;   // candidates tried so far (kept: destroying one would reshuffle the queue mapping)
  struct Conv2Choice { hipStream_t for_stream, conv2; bool ok; };
  std::vector<Conv2Choice> conv2_known;    // per launch stream seen so far: the candidate that shares no hardware queue with it
  bool conv2_ok = false;                   // or with the side stream (ok = false: none found, no split on that stream)
  long long *probe_stamp = nullptr;        // pinned: the queue probe's device time stamps
  bool split_last = false;                 // the last call issued the layers behind conv1b as two half batches
  int split_probe = -3;                    // outcome of the last probe: 1 free queue found, 0 none, -1 not measurable, -2 stream
                                           // under capture, 2 probe switched off (first candidate trusted), -3 never probed
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // pipelined calls: the launch stream does not wait for the second stream's half batch at the end of a step — the side chain
  // does, and the launch stream only in front of the NEXT call's conv1b (its conv1a runs beside the other half's last kernels)
  bool join_pending = false;
  bool defer_join = true;   // SPFE_DEFER_JOIN=0: the join at the end of the step, on the launch stream
  int desc_in_replay = 1;   // SPFE_DESC_IN_REPLAY
  int f32_split = 2;   // parts (0 = off)
  int bf16_split = -1;  // SPFE_BF16_SPLIT: the same for the bf16 stack; -1 = frames of fewer than 10,000 cells (752x480: +2 %; 1280x720: +-0)
  bool desc_recorded = false;
  long ticket = 0;          // calls so far; call t uses slot t % NTICKET
  bool cov_inflight = false;
  std::vector<void *> dev_allocs;
  std::vector<void *> host_allocs;
  uint8_t *d_img = nullptr;
  float *d_w1a = nullptr, *d_b1a = nullptr;
  unsigned short *d_w1a_tab = nullptr;   // bf16 mode: conv1a's weights as the MFMA operand table of conv1a_mfma.h
  float *act[8] = {};
  float *d_head = nullptr, *d_semi = nullptr, *d_coarse = nullptr;
  unsigned short *d_hd = nullptr;    // bf16 mode: ReLU(convPa) | ReLU(convDa), [B][C][512] bf16 (input of the two bf16 heads)
  unsigned char *d_wdb = nullptr, *d_wpb = nullptr;   // bf16 mode: convDb / convPb weights, head_bf16.hip layout
  float *d_wdb32 = nullptr, *d_wpb32 = nullptr;       // f32 mode: the same for head_f32.hip (SPFE_F32_HEADS=1; default: generic kernel)
  bool f32_heads = false;
  // f32: convPb and the detector tail in one launch (pbtail_f32.hip): the two full 32-channel tiles on the MFMA, the dustbin
  // channel as the contract's fmaf chain on the VALU, the tail on the logits while they are still in LDS.  SPFE_PBTAIL=0:
  // convPb as a launch of the generic kernel + tail_kernel (same bits)
  bool pbtail = true;
  float *d_wpb_dust = nullptr;                        // convPb's row 64 (the dustbin channel), [256]
  // "sparse convDb": the descriptor head runs BEHIND the selection, on the cells some emitted keypoint's bilinear taps read
  // (<= 4 per keypoint: 28 % of a 1280x720 frame at 1000 keypoints), gathered through select_kernel's list; d_coarse keeps
  // the dense layout, only the rows anybody reads are written.  SPFE_SPARSE_DB=0: the dense head in the launch stream.
  bool sparse_db = true;
  bool sparse_db_sync_only = false;   // ... in synchronous calls only (bf16 frames below 10,000 cells; SPFE_SPARSE_DB=2)
  bool sparse_last = false;      // the last call left d_coarse sparse (spfe_debug_read("coarse") completes it on demand)
  // ... and convDa with it (bf16 mode, da_gather_bf16.hip): the dense launch computes convPa only, the descriptor branch
  // runs on the listed cells from conv4b's output on.  SPFE_SPARSE_DA=0: convPa|Da dense, only convDb gathered.
  bool sparse_da = false;
  int sparse_da_mode = 1;        // SPFE_SPARSE_DA: 0 never, 1 synchronous calls only (default), 2 pipelined calls too
  bool sparse_da_call = false;   // ... this / the last call
  int *d_db_list = nullptr, *d_db_total = nullptr;
  int db_cap = 0;                // list entries per frame: min(4 kmax, C)
  int db_tiles_per_wg = 4;       // SPFE_DB_TILES_PER_WG: the gathered head's grid = listed tiles / this (a workgroup's weights: 128 KB)
  hipEvent_t ev_sel = nullptr;   // side stream: this call's selection (and its cell list) is done
  hipEvent_t ev_dbs[2] = {};     // by ticket parity: that call's gathered head (reader of the head activations / of conv4b's output) is done
  bool dbs_recorded[2] = {};
  // sparse_da: conv4b's output exists twice (by ticket parity), so that the NEXT call's conv4b does not wait for this
  // call's gathered convDa, which runs behind the selection on the side stream
  float *act7_alt = nullptr;
  float *d_wda32 = nullptr;          // f32 mode: convDa's weights in da_gather_f32.hip's order
  const float *feat_cur = nullptr;   // conv4b's output of the call being enqueued / of the last call
  // what the detector tail (launch stream) hands to the side chain exists twice, by ticket parity: batch i + 1's tail then
  // only has to wait for batch i - 1's side chain, not for batch i's (which runs beside batch i + 1's convolutions)
  float *d_heat_log[2] = {}, *d_heat = nullptr, *d_heat_inv = nullptr;
  float *d_minmax[2] = {}, *d_cell_score[2] = {}, *d_heat_consts = nullptr;
  uint8_t *d_cell_k[2] = {}, *d_cell_mask = nullptr;
  const uint8_t *rec_of[NTICKET] = {};   // record buffer of each ticket (same buffer twice in a row: the old ordering)
  int *d_kp_cell = nullptr;
  int select_lean = 0;                // SPFE_SELECT_LEAN: 1 = select_kernel keeps 2 bytes a cell in LDS on every frame size, 0 = only
                                      // above 16,384 cells (default), -1 = in pipelined calls.  Measured (round 4, same-box A/B, 8
                                      // frames per call, pipelined): the lean form starts beside a convolution workgroup instead of
                                      // waiting for a free CU, and that is NOT a gain — f32 752x480 2107 / 2116 -> 2085 / 2082
                                      // frames/s (it now sits beside conv1b: 0.87 -> 0.83 of peak), bf16 1280x720 7687 / 7690 ->
                                      // 7670 / 7684, bf16 752x480 14,888 / 14,908 -> 14,846 / 14,826: the selection's 390 us
                                      // "overlapped" were waiting time off the critical path
  int *d_sel_slot = nullptr;          // frames of more than 16,384 cells: select_kernel's global scratch (tail_select.hip)
  uint16_t *d_sel_list = nullptr;
  uint8_t *d_records = nullptr;
  spfe::CovScratch cov{};
  ConvLayer layers[10];
  spfe::RecordLayout rl{};
  // host side
  uint8_t *h_img = nullptr, *h_records = nullptr;
  float *h_heat = nullptr, *h_heat_inv = nullptr;
  int last_n = 0;
  int num_cus = 256;
  int small_maxh = -1;
  // input staging (spfe_set_staging)
  spfe_staging st{};
  bool st_set = false;
  float *d_map_x = nullptr, *d_map_y = nullptr;
  uint8_t *d_raw = nullptr, *h_raw = nullptr;
  // descriptor matching (spfe_match*): scratch grown on demand
  unsigned long long *m_best_t = nullptr, *m_best_q = nullptr;
  uint8_t *m_stage_q = nullptr, *m_stage_t = nullptr, *m_out = nullptr, *m_out2 = nullptr;
  int *p_cidx = nullptr;           // patch association scratch: [4096][4] candidates, distances, host staging
  float *p_cdist = nullptr;
  uint8_t *p_stage = nullptr;
  size_t p_stage_bytes = 0;
  int m_pairs = 0, m_cap = 0;      // capacity of m_best_* ([pairs][cap])
  int m_host_cap = 0;              // rows the host-API staging blocks / m_out hold
  unsigned tile2_mask = 0;   // SPFE_TILE2_MASK: f32 layers forced onto 2-row tiles (probe knob)
  bool tile2_auto = true;    // SPFE_TILE2_AUTO=0: never choose 2-row tiles
  // f32, a single frame: a POOLED low-resolution layer (conv3b: 180 eight-row items on 256 CUs — one round of the longest
  // items, 70 % of the CUs busy) as UN-pooled 2-row tiles (720 items: three rounds of quarter-size items) into a scratch
  // buffer + a 2x2 max-pool pass (pool2x2_f32_kernel; bias / ReLU / max commute exactly: same bits).  SPFE_POOL_SPLIT:
  // -1 cost model, 0 never, 1 wherever the shapes allow (tests)
  int pool_split = -1;
  // f32, a single frame: the low-resolution layers without a pool (conv3a, conv4a, conv4b, convPa [| convDa]) on the K-chain
  // kernel (conv_f32_kc.hip, v_mfma_f32_16x16x4_f32: every output's fmaf chain advances 4 k per 32-cycle issue and the layer is
  // cut into 16 x 16 chains, 3 per wavefront, so that every SIMD has work).  SPFE_KC: -1 = the layers it measured faster on,
  // 0 = never, else a mask of conv layer indices (bit 3 = conv3a, 5 = conv4a, 6 = conv4b, 7 = convPa | Da)
  int kc_mask = -1;
  float *d_wkc[8] = {};      // their weights in conv_f32_kc_pack_weights order
  float *d_unpooled = nullptr;   // [<= 2 frames][H / 4][W / 4][128]
  unsigned tile16_mask = 0;  // f32 layers (bit i = conv layer i of enqueue()) on 16-row / 8-wave tiles
  int conv1b_split_rows = -1; // ... and, when that launch was cut in a 16-row and an 8-row part, the 16-row part's tile rows ("conv1b_split_rows")
  int conv1b_tile_rows = 8;  // rows per tile of the last call's conv1b launch (f32; spfe_debug_read("conv1b_tile_rows"))
  int tile16x4 = 1;          // SPFE_TILE16X4: conv1b on 16-row tiles of 4 wavefronts x 4 rows (0 never, 1 by the cost model — possibly
                             // cut in a 16-row and an 8-row launch —, 2 always in one launch, 3 cost model without the cut)
  bool fuse1a = false;  // f32: conv1a computed inside conv1b (opt-in: SPFE_FUSE_CONV1A=1; measured perf-neutral)
  bool fuse1a_bf16 = true;  // bf16: conv1a computed by the producer waves of the wave-specialised conv1b (SPFE_BF16_FUSE_CONV1A=0 to split)
  uint8_t *dust_scratch = nullptr;   // spfe_align_dust: dust map | points | pose | output block (device)
  uint8_t *dust_host = nullptr;      // pinned mirror of the output block
  // pipelined host path (spfe_submit_batch / spfe_collect_batch): NPIPE batches in flight, each with its own
  // pinned input / output staging and device frame / record buffers; H2D and D2H on copy streams
  static constexpr int NPIPE = 3;
  struct PipeSlot {
    uint8_t *h_img = nullptr, *d_img = nullptr, *d_rec = nullptr, *h_rec = nullptr;
    float *h_heat = nullptr, *h_heat_inv = nullptr;
    hipEvent_t ev_h2d = nullptr, ev_done = nullptr;
    long ticket = -1;   // records ticket of the batch in this slot, -1 = free
    int n = 0;
  } pipe[NPIPE];
  bool pipe_ready = false, pipe_mode = false;
  hipStream_t s_h2d = nullptr, s_d2h = nullptr;
  long pipe_submitted = 0;
  // RCCL all-gather of the records (spfe_comm_init / spfe_allgather_records): librccl is loaded on demand
  void *rccl_lib = nullptr;
  ncclComm_t comm = nullptr;
  int comm_rank = 0, comm_world = 0;
  hipStream_t comm_stream = nullptr;   // the stream of the collective: the side stream (default) or one of its own
  bool comm_own_stream = false;        // SPFE_COMM_OWN_STREAM=1
  hipEvent_t ev_gather = nullptr;      // the last gather on comm_stream is done
  bool gather_recorded = false;
  pfn_ncclCommInitRank p_ncclCommInitRank = nullptr;
  pfn_ncclCommDestroy p_ncclCommDestroy = nullptr;
  pfn_ncclCommCount p_ncclCommCount = nullptr;
  pfn_ncclAllGather p_ncclAllGather = nullptr;
  pfn_ncclGetErrorString p_ncclGetErrorString = nullptr;
  unsigned ws_mask = 15u;   // bf16 layers (bit i = conv layer i of enqueue(), Cin = 64 only) that may use the wave-specialised kernel
  int ws_min_items = 11;    // ... when the launch has at least this many (tile, 64-channel block) items per workgroup (pipelined calls)
  int ws_min_items_sync = 5;   // ... the same for synchronous calls
  unsigned char *d_wws[4] = {};  // their weights in conv_bf16_ws.hip's layout
  unsigned char *d_wrw[4] = {};  // bf16 Cin = 128 layers (conv3b, 4a, 4b, Pa|Da): weights in conv_bf16_rw.hip's fragment order
  bool bf16_rw = true;           // SPFE_BF16_RW: register-resident-weights kernel for those layers
  int rw_rows3 = 1;              // SPFE_BF16_RW_ROWS3
  int rw_min4 = 3, rw_min2 = 2;  // ... 4-row tiles from this many tiles per workgroup, 2-row tiles from this many, else conv_bf16.hip
  int side_cus_default = 0;      // SPFE_SIDE_CUS
  bool bf16_dyn = true;          // SPFE_BF16_DYN_QUEUE
  int tile16_min_items = 3;      // SPFE_BF16_TILE16_MIN_ITEMS (0 = 8-row tiles only)
  int tile_rows_big = 12;        // SPFE_BF16_TILE_ROWS (12 | 16)
  int *d_tile_ctr = nullptr;     // [8 layers][16] tile-queue counters, zero at the start of every enqueue(): cleared by the previous
                                 // call's detector tail (a launch of its own cost 9 us between two 1 ms steps), or by a launch when that did not happen
  bool tile_ctr_clean = false;
  bool act0_missing = false;  // the last call computed conv1a inside conv1b
  bool bf16 = false;  // SPFE_PRECISION_BF16: all twelve convolutions (1x1 heads included) as bf16 GEMMs with f32 accumulation; f32 tail
  // per-stage timing: a ring of event sets, one set per enqueue() call
  bool timing = false;
  bool timing_all = true;  // false (SPFE_STAGE_TIMING=2): events around the dominant kernel (conv1b) only
  static constexpr int EVSETS = 128;
  std::vector<hipEvent_t> evpool;  // [EVSETS][NSTAGE + 1]
  long calls = 0, calls_at_reset = 0;
  hipEvent_t *ev = nullptr;        // set used by the current call
};

namespace {

template <class T>
int dev_alloc(spfe_handle h, T **p, size_t count) {
  void *q = nullptr;
  HIP_TRY(hipMalloc(&q, count * sizeof(T) + 256));
  h->dev_allocs.push_back(q);
  *p = reinterpret_cast<T *>(q);
  return SPFE_OK;
}
template <class T>
int host_alloc(spfe_handle h, T **p, size_t count) {
  void *q = nullptr;
  HIP_TRY(hipHostMalloc(&q, count * sizeof(T) + 256, hipHostMallocDefault));
  h->host_allocs.push_back(q);
  *p = reinterpret_cast<T *>(q);
  return SPFE_OK;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Order stream `s` behind `ev` — but only if `ev` has not fired yet.  A wait is a barrier packet in the compute queue and
// costs ~10 us of idle queue whether or not the event is long done (measured on kernel timelines of the pipelined steps:
// conv1a -> [wait] -> conv1b 12 us apart); the waits below guard buffers against work TWO batches back, which in steady state
// finished long ago: one hipEventQuery on the host replaces the packet.  (Not under stream capture: a query is illegal there,
// and a captured wait is a graph edge, not a packet.)
hipError_t wait_if_pending(hipStream_t s, hipEvent_t ev) {
  static const bool always = getenv("SPFE_ALWAYS_WAIT") && atoi(getenv("SPFE_ALWAYS_WAIT")) != 0;   // A/B knob
  if (!always) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone) {
      const hipError_t q = hipEventQuery(ev);
      if (q == hipSuccess) return hipSuccess;
      if (q != hipErrorNotReady) (void)hipGetLastError();   // (e.g. an event never recorded: fall through to the wait)
    }
  }
  return hipStreamWaitEvent(s, ev, 0);
}

void make_layout(int kmax, int C, bool desc_bf16, spfe::RecordLayout *r) {
  size_t o = 0;
  r->kmax = kmax;
  r->desc_bf16 = desc_bf16 ? 1 : 0;
  r->off_hdr = o; o += 16;
  r->off_xy = o; o = align_up(o + (size_t)kmax * 2 * 4, 16);
  r->off_resp = o; o = align_up(o + (size_t)kmax * 4, 16);
  r->off_cov = o; o = align_up(o + (size_t)kmax * 2 * 4, 16);
  r->off_cinv = o; o = align_up(o + (size_t)kmax * 2 * 4, 16);
  r->off_desc = o; o = align_up(o + (size_t)kmax * SPFE_DESC_DIM * (desc_bf16 ? 2 : 4), 16);
  r->off_occ = o; o = align_up(o + (size_t)C * 2, 16);
  r->off_dd = o; o = align_up(o + (size_t)C * 4, 16);
  r->off_sd = o; o = align_up(o + (size_t)C * 4, 16);
  r->bytes = align_up(o, 256);
}

// offsets into the flat blob (register_module order, sp_extractor.cpp:46-62)
size_t blob_weight_offset(int l) {
  size_t off = 0;
  for (int i = 0; i < l; ++i) {
    const spfe_layer_t &L = SPFE_LAYERS[i];
    off += (size_t)L.cout * L.cin * L.ksize * L.ksize + L.cout;
  }
  return off;
}

// pack OIHW weights of one or two layers (concatenated along cout) into slabs
// [nblk][chunk][n-tile(2)][tap][KC][32] (K order of spfe_exact_math.h) + padded bias
int pack_layer(spfe_handle h, const float *blob, const int *lids, int nl, ConvLayer *out) {
  const spfe_layer_t &L0 = SPFE_LAYERS[lids[0]];
  const int cin = L0.cin, ks = L0.ksize, taps = ks * ks;
  int cout = 0;
  for (int i = 0; i < nl; ++i) cout += SPFE_LAYERS[lids[i]].cout;
  const int kc = spfe::conv_kc(ks), nchunk = cin / kc, nblk = (cout + 63) / 64;
  std::vector<float> w((size_t)nblk * nchunk * taps * kc * 64, 0.0f), bia((size_t)nblk * 64, 0.0f);
  int co_base = 0;
  for (int i = 0; i < nl; ++i) {
    const spfe_layer_t &L = SPFE_LAYERS[lids[i]];
    const float *W = blob + blob_weight_offset(lids[i]);
    const float *Bv = W + (size_t)L.cout * L.cin * taps;
    for (int co = 0; co < L.cout; ++co) {
      const int g = co_base + co, nb = g / 64, j = g % 64;
      bia[g] = Bv[co];
      for (int ci = 0; ci < cin; ++ci) {
        const int ch = ci / kc, c = ci % kc;
        for (int t = 0; t < taps; ++t)
          w[(((((size_t)nb * nchunk + ch) * 2 + j / 32) * taps + t) * kc + c) * 32 + j % 32] =
              W[((size_t)co * cin + ci) * taps + t];
      }
    }
    co_base += L.cout;
  }
  int rc;
  if ((rc = dev_alloc(h, &out->d_w, w.size()))) return rc;
  if ((rc = dev_alloc(h, &out->d_b, bia.size()))) return rc;
  HIP_TRY(hipMemcpy(out->d_w, w.data(), w.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(out->d_b, bia.data(), bia.size() * 4, hipMemcpyHostToDevice));
  out->cin = cin;
  out->cout_real = cout;
  out->nblk = nblk;
  out->ks = ks;
  return SPFE_OK;
}

unsigned short host_bf16_rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

// bf16 slabs for conv_bf16.hip: [nblk][chunk of 32 channels][tap][n 64][80-byte row: 32 bf16 + pad],
// each slab padded to conv_bf16_slab_bytes(); bias stays f32
int pack_layer_bf16(spfe_handle h, const float *blob, const int *lids, int nl, ConvLayer *out) {
  const spfe_layer_t &L0 = SPFE_LAYERS[lids[0]];
  const int cin = L0.cin, taps = 9;
  int cout = 0;
  for (int i = 0; i < nl; ++i) cout += SPFE_LAYERS[lids[i]].cout;
  const int nchunk = cin / 32, nblk = (cout + 63) / 64;
  const size_t slab = spfe::conv_bf16_slab_bytes();
  std::vector<unsigned char> w((size_t)nblk * nchunk * slab, 0);
  std::vector<float> bia((size_t)nblk * 64, 0.0f);
  int co_base = 0;
  for (int i = 0; i < nl; ++i) {
    const spfe_layer_t &L = SPFE_LAYERS[lids[i]];
    const float *W = blob + blob_weight_offset(lids[i]);
    const float *Bv = W + (size_t)L.cout * L.cin * taps;
    for (int co = 0; co < L.cout; ++co) {
      // row of the 64-channel block: even channels fill accumulator tile 0, odd ones tile 1 (the kernels pack a lane's
      // channel pair into one dword store); the bias stays in channel order
      const int g = co_base + co, nb = g / 64, c64 = g % 64, j = (c64 & 1) * 32 + (c64 >> 1);
      bia[g] = Bv[co];
      for (int ci = 0; ci < cin; ++ci) {
        const int ch = ci / 32, c = ci % 32;
        for (int t = 0; t < taps; ++t) {
          const unsigned short v = host_bf16_rne(W[((size_t)co * cin + ci) * taps + t]);
          // row (tap, cout) = 64 B: 4 pieces of 8 channels, piece g in slot g ^ ((cout >> 2) & 3) (conv_bf16.hip's LDS layout)
          memcpy(&w[((size_t)nb * nchunk + ch) * slab + ((size_t)t * 64 + j) * 64 + (((c / 8) ^ ((j >> 2) & 3)) * 16) + (c % 8) * 2], &v, 2);
        }
      }
    }
    co_base += L.cout;
  }
  int rc;
  unsigned char *dw = nullptr;
  if ((rc = dev_alloc(h, &dw, w.size()))) return rc;
  if ((rc = dev_alloc(h, &out->d_b, bia.size()))) return rc;
  HIP_TRY(hipMemcpy(dw, w.data(), w.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(out->d_b, bia.data(), bia.size() * 4, hipMemcpyHostToDevice));
  out->d_w = reinterpret_cast<float *>(dw);
  out->cin = cin;
  out->cout_real = cout;
  out->nblk = nblk;
  out->ks = 3;
  return SPFE_OK;
}

// conv_bf16_ws.hip layout: [nblk][tap][cout 64][8 pieces of 8 cin, piece g in slot g ^ ((cout >> 1) & 7)]
int pack_layer_bf16_ws(spfe_handle h, const float *blob, int lid, unsigned char **out) {
  const spfe_layer_t &L = SPFE_LAYERS[lid];
  if (L.cin != 64 || L.ksize != 3 || L.cout % 64) return fail(SPFE_EINVAL, "internal: layer %d is not a Cin = 64 3x3 layer", lid);
  const size_t blk = spfe::conv_bf16_ws_weight_bytes();
  std::vector<unsigned char> w((size_t)(L.cout / 64) * blk, 0);
  const float *W = blob + blob_weight_offset(lid);
  for (int co = 0; co < L.cout; ++co)
    for (int ci = 0; ci < 64; ++ci)
      for (int t = 0; t < 9; ++t) {
        const unsigned short v = host_bf16_rne(W[((size_t)co * 64 + ci) * 9 + t]);
        // row of the block: even channels fill accumulator tile 0, odd ones tile 1 (conv_bf16_ws.hip's epilogue
        // packs a lane's channel pair into one dword store)
        const int c64 = co % 64, j = (c64 & 1) * 32 + (c64 >> 1), slot = (ci / 8) ^ ((j >> 1) & 7);
        memcpy(&w[(size_t)(co / 64) * blk + ((size_t)t * 64 + j) * 128 + slot * 16 + (ci % 8) * 2], &v, 2);
      }
  int rc;
  if ((rc = dev_alloc(h, out, w.size()))) return rc;
  HIP_TRY(hipMemcpy(*out, w.data(), w.size(), hipMemcpyHostToDevice));
  return SPFE_OK;
}

// conv_bf16_rw.hip layout for a Cin = 128 layer (or two concatenated ones: convPa | convDa)
int pack_layer_bf16_rw(spfe_handle h, const float *blob, const int *lids, int nl, unsigned char **out) {
  int cout = 0;
  for (int i = 0; i < nl; ++i) {
    const spfe_layer_t &L = SPFE_LAYERS[lids[i]];
    if (L.cin != 128 || L.ksize != 3) return fail(SPFE_EINVAL, "internal: layer %d is not a Cin = 128 3x3 layer", lids[i]);
    cout += L.cout;
  }
  if (cout % 128) return fail(SPFE_EINVAL, "internal: %d output channels are not whole 128-channel groups", cout);
  std::vector<unsigned short> wb((size_t)cout * 128 * 9);
  size_t o = 0;
  for (int i = 0; i < nl; ++i) {
    const spfe_layer_t &L = SPFE_LAYERS[lids[i]];
    const float *W = blob + blob_weight_offset(lids[i]);
    for (size_t k = 0; k < (size_t)L.cout * 128 * 9; ++k) wb[o++] = host_bf16_rne(W[k]);
  }
  std::vector<unsigned char> w((size_t)(cout / 128) * spfe::conv_bf16_rw_weight_bytes());
  spfe::conv_bf16_rw_pack_weights(wb.data(), cout, w.data());
  int rc;
  if ((rc = dev_alloc(h, out, w.size()))) return rc;
  HIP_TRY(hipMemcpy(*out, w.data(), w.size(), hipMemcpyHostToDevice));
  return SPFE_OK;
}

int load_blob(const spfe_config *cfg, std::vector<float> *blob) {
  blob->resize(SPFE_NUM_PARAMS);
  if (cfg->weights) {
    memcpy(blob->data(), cfg->weights, (size_t)SPFE_NUM_PARAMS * 4);
    return SPFE_OK;
  }
  if (!cfg->weights_path) return fail(SPFE_EWEIGHTS, "no weights: both weights and weights_path are NULL");
  FILE *f = fopen(cfg->weights_path, "rb");
  if (!f) return fail(SPFE_EWEIGHTS, "cannot open weight file %s", cfg->weights_path);
  unsigned char head[16];
  uint32_t ver = 0;
  uint64_t n = 0;
  bool ok = fread(head, 1, 16, f) == 16 && memcmp(head, "SPFW", 4) == 0;
  if (ok) {
    memcpy(&ver, head + 4, 4);
    memcpy(&n, head + 8, 8);
    ok = ver == 1 && n == SPFE_NUM_PARAMS && fread(blob->data(), 4, n, f) == n;
  }
  fclose(f);
  if (!ok) return fail(SPFE_EWEIGHTS, "%s is not a valid SPFW v1 file with %d params", cfg->weights_path, SPFE_NUM_PARAMS);
  return SPFE_OK;
}

int build(spfe_handle h, const spfe_config *cfg) {
  h->cfg = *cfg;
  h->H = cfg->height; h->W = cfg->width;
  h->hc = h->H / 8; h->wc = h->W / 8; h->C = h->hc * h->wc;
  h->kmax = cfg->num_features + 1;
  h->B = cfg->max_batch;
  h->bf16 = cfg->precision == SPFE_PRECISION_BF16;
  const int H = h->H, W = h->W, B = h->B, C = h->C;
  HIP_TRY(hipSetDevice(cfg->device));
  {
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, cfg->device));
    h->num_cus = prop.multiProcessorCount;
    const char *genv = getenv("SPFE_CONV_GRID");
    if (genv) h->num_cus = atoi(genv);
  }
  HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  {
    // SPFE_SIDE_PRIORITY (probe knob): -1 = the side stream at the device's highest priority, 1 = lowest, unset / 0 = default
    const char *pe = getenv("SPFE_SIDE_PRIORITY");
    const int want = pe ? atoi(pe) : 0;
    int lo = 0, hi = 0;   // (numerically: greatest priority = lowest value)
    // SPFE_SIDE_CUS=N: the side stream (selection, descriptors, covariance) confined to the last N of the device's CUs
    // (hipExtStreamCreateWithCUMask; mask bit i <-> CU i / 8 of XCD i % 8: tools/microbench/cumask_probe.hip), so that its
    // long-lived small workgroups cannot sit on every CU while the convolutions of the next batch want whole CUs
    const char *ce = getenv("SPFE_SIDE_CUS");
    int side_cus = ce ? atoi(ce) : h->side_cus_default;
    hipDeviceProp_t prop;
    if (side_cus > 0 && hipGetDeviceProperties(&prop, cfg->device) == hipSuccess && prop.multiProcessorCount >= 64 &&
        side_cus < prop.multiProcessorCount) {
      const int ncu = prop.multiProcessorCount;
      std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
      for (int b = ncu - side_cus; b < ncu; ++b) mask[b / 32] |= 1u << (b % 32);
      if (hipExtStreamCreateWithCUMask(&h->side, (uint32_t)mask.size(), mask.data()) != hipSuccess) h->side = nullptr;
    }
    if (h->side) {
    } else if (want && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi)
      HIP_TRY(hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, want < 0 ? hi : lo));
    else
      HIP_TRY(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
  }
  for (int i = 0; i < spfe_handle_s::NTICKET; ++i) {
    HIP_TRY(hipEventCreateWithFlags(&h->ev_post[i], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&h->ev_cov[i], hipEventDisableTiming));
  }
  HIP_TRY(hipEventCreateWithFlags(&h->ev_desc, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
  if (const char *e = getenv("SPFE_F32_SPLIT")) h->f32_split = atoi(e);
  if (const char *e = getenv("SPFE_DESC_IN_REPLAY")) h->desc_in_replay = atoi(e);
  if (const char *e = getenv("SPFE_BF16_SPLIT")) h->bf16_split = atoi(e);
  HIP_TRY(hipEventCreateWithFlags(&h->ev_db, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&h->ev_sel, hipEventDisableTiming));
  for (int i = 0; i < 2; ++i) HIP_TRY(hipEventCreateWithFlags(&h->ev_dbs[i], hipEventDisableTiming));
  // Measured, pipelined, 8 frames per call (same-box A/B): bf16 1280x720 +2.5 ... 3.8 % (7590 -> 7780, 7322 -> 7604 frames/s),
  // f32 752x480 +0.4 ... 0.7 %, bf16 752x480 -2 ... 3 %: there the launch stream runs as two half batches on two streams, the
  // dense head (HBM-bound) hid completely beside the other half's convolutions (removing it altogether gains nothing), and
  // the gathered launch is pure extra work for the chip.  So: f32, and bf16 frames of >= 10,000 cells (= no two-stream split).
  // SYNCHRONOUS calls of those small bf16 frames take the gathered branch all the same (round 4): there is no other half
  // batch to hide the dense head beside, and the gathered form brings the inline chain with it (enqueue_post) — 752x480:
  // a single frame's p50 0.287 -> 0.264 ... 0.274 ms, 8 frames per synchronous call +0.5 ... 0.8 %; 640x480: 0.311 -> 0.288
  // ms, +2.4 %.  SPFE_SPARSE_DB = 0 never, 1 every call, 2 synchronous calls only
  h->sparse_db = true;
  h->sparse_db_sync_only = h->bf16 && h->C < 10000;
  h->db_tiles_per_wg = h->bf16 ? 4 : 1;
  if (const char *e = getenv("SPFE_SPARSE_DB")) { h->sparse_db = atoi(e) != 0; h->sparse_db_sync_only = atoi(e) == 2; }
  // the gathered kernels form row byte offsets in 32 bits (the head activations' rows are 2048 / 1024 bytes, 0x80000000 is their
  // out-of-range marker): batches beyond that take the dense head (the launchers refuse them as well)
  if ((size_t)cfg->max_batch * h->C * (h->bf16 ? 1024 : 2048) >= ((size_t)1 << 31)) h->sparse_db = false;
  if (const char *e = getenv("SPFE_DB_TILES_PER_WG")) h->db_tiles_per_wg = atoi(e);
  if (const char *de = getenv("SPFE_DEFER_DB")) h->defer_db = atoi(de) != 0;
  if (const char *de = getenv("SPFE_DEFER_JOIN")) h->defer_join = atoi(de) != 0;
  {
    const char *fenv = getenv("SPFE_FUSE_CONV1A");
    h->fuse1a = fenv && atoi(fenv) != 0;
    const char *menv = getenv("SPFE_TILE16_MASK");
    if (menv) h->tile16_mask = (unsigned)strtoul(menv, nullptr, 0);
    if (const char *m2 = getenv("SPFE_TILE2_MASK")) h->tile2_mask = (unsigned)strtoul(m2, nullptr, 0);
    if (const char *m4 = getenv("SPFE_TILE16X4")) h->tile16x4 = atoi(m4);
    if (const char *a2 = getenv("SPFE_TILE2_AUTO")) h->tile2_auto = atoi(a2) != 0;
    if (const char *ps = getenv("SPFE_POOL_SPLIT")) h->pool_split = atoi(ps);
    if (const char *km = getenv("SPFE_KC")) h->kc_mask = (int)strtol(km, nullptr, 0);
    const char *wenv = getenv("SPFE_BF16_WS_MASK");
    if (wenv) h->ws_mask = (unsigned)strtoul(wenv, nullptr, 0) & 0xfu;
    const char *f16env = getenv("SPFE_BF16_FUSE_CONV1A");
    if (f16env) h->fuse1a_bf16 = atoi(f16env) != 0;
    // Synchronous calls (latency): the wave-specialised kernel wins from ~5 items per workgroup (batch 1 at 752x480: 0.43 ->
    // 0.385 ms, conv1a fused).  Pipelined calls (SPFE_FLAG_ASYNC_COV): it holds all of a CU's LDS, the side-stream kernels of
    // the previous batch cannot start beside it, and at 752x480 x 8 (0.65 ms steps) their chain becomes the critical path
    // when the quarter-resolution layers take it too (12,450 -> 12,050 frames/s): those keep the higher bar.
    // (the bar is picked per call: spfe_submit_batch pipelines on a handle created without the flag)
    const char *ienv = getenv("SPFE_BF16_WS_MIN_ITEMS");
    if (ienv) h->ws_min_items = h->ws_min_items_sync = atoi(ienv);
    const char *t16env = getenv("SPFE_BF16_TILE16_MIN_ITEMS");
    if (t16env) h->tile16_min_items = atoi(t16env);
    const char *trenv = getenv("SPFE_BF16_TILE_ROWS");
    if (trenv) h->tile_rows_big = atoi(trenv);
    const char *denv = getenv("SPFE_BF16_DYN_QUEUE");
    if (denv) h->bf16_dyn = atoi(denv) != 0;
    if (const char *e = getenv("SPFE_BF16_RW")) h->bf16_rw = atoi(e) != 0;
    if (const char *e = getenv("SPFE_BF16_RW_MIN4")) h->rw_min4 = atoi(e);
    if (const char *e = getenv("SPFE_BF16_RW_ROWS3")) h->rw_rows3 = atoi(e);
    if (const char *e = getenv("SPFE_BF16_RW_MIN2")) h->rw_min2 = atoi(e);
  }
  const char *tenv = getenv("SPFE_STAGE_TIMING");
  h->timing = tenv && atoi(tenv) != 0;
  h->timing_all = !(tenv && atoi(tenv) == 2);
  if (h->timing) {
    h->evpool.resize((size_t)spfe_handle_s::EVSETS * (NSTAGE + 1), nullptr);
    for (auto &e : h->evpool) HIP_TRY(hipEventCreate(&e));
  }

  std::vector<float> blob;
  int rc = load_blob(cfg, &blob);
  if (rc) return rc;

  // conv1a weights: [tap][64]
  {
    const float *Wt = blob.data() + blob_weight_offset(0);
    std::vector<float> w(9 * 64), bv(64);
    for (int co = 0; co < 64; ++co) {
      for (int t = 0; t < 9; ++t) w[t * 64 + co] = Wt[co * 9 + t];
      bv[co] = Wt[64 * 9 + co];
    }
    if ((rc = dev_alloc(h, &h->d_w1a, w.size()))) return rc;
    if ((rc = dev_alloc(h, &h->d_b1a, bv.size()))) return rc;
    HIP_TRY(hipMemcpy(h->d_w1a, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->d_b1a, bv.data(), bv.size() * 4, hipMemcpyHostToDevice));
    if (h->bf16) {
      // conv1a_mfma.h: [j 2][lane 64][e 8] = bf16(w[channel 32 j + (lane & 31)][tap 8 (lane >> 5) + e]), 0 for taps >= 9
      std::vector<unsigned short> tab(2 * 64 * 8, 0);
      for (int j = 0; j < 2; ++j)
        for (int ln = 0; ln < 64; ++ln)
          for (int e = 0; e < 8; ++e) {
            const int t = 8 * (ln >> 5) + e, co = 32 * j + (ln & 31);
            if (t < 9) tab[(j * 64 + ln) * 8 + e] = host_bf16_rne(Wt[co * 9 + t]);
          }
      if ((rc = dev_alloc(h, &h->d_w1a_tab, tab.size()))) return rc;
      HIP_TRY(hipMemcpy(h->d_w1a_tab, tab.data(), tab.size() * 2, hipMemcpyHostToDevice));
    }
  }

  // activations (NHWC f32).  act[0]=conv1a .. act[7]=conv4b
  const int lh[8] = {H, H / 2, H / 2, H / 4, H / 4, H / 8, H / 8, H / 8};
  const int lw[8] = {W, W / 2, W / 2, W / 4, W / 4, W / 8, W / 8, W / 8};
  const int lc[8] = {64, 64, 64, 64, 128, 128, 128, 128};
  for (int i = 0; i < 8; ++i)
    if ((rc = dev_alloc(h, &h->act[i], (size_t)B * lh[i] * lw[i] * lc[i]))) return rc;
  if ((rc = dev_alloc(h, &h->d_img, (size_t)B * H * W))) return rc;
  if (!h->bf16 && h->pool_split != 0 && !(H & 15) && !(W & 15))   // (the un-pooled output of the largest pooled layer this serves: conv2b / conv3b of <= 2 frames)
    if ((rc = dev_alloc(h, &h->d_unpooled, (size_t)std::min(B, 2) * (H / 2) * (W / 2) * 64))) return rc;
  if ((rc = dev_alloc(h, &h->d_head, (size_t)B * C * 512))) return rc;
  if ((rc = dev_alloc(h, &h->d_semi, (size_t)B * C * SPFE_SEMI_CH))) return rc;
  if ((rc = dev_alloc(h, &h->d_coarse, (size_t)B * C * SPFE_DESC_DIM))) return rc;
  for (int k = 0; k < 2; ++k) {
    if ((rc = dev_alloc(h, &h->d_heat_log[k], (size_t)B * H * W))) return rc;
    if ((rc = dev_alloc(h, &h->d_minmax[k], (size_t)B * spfe::tail_parts(h->H, h->W) * 2))) return rc;
    if ((rc = dev_alloc(h, &h->d_cell_score[k], (size_t)B * C))) return rc;
    if ((rc = dev_alloc(h, &h->d_cell_k[k], (size_t)B * C))) return rc;
  }
  if ((rc = dev_alloc(h, &h->d_heat_inv, (size_t)B * H * W))) return rc;
  if (cfg->flags & SPFE_FLAG_HEAT)
    if ((rc = dev_alloc(h, &h->d_heat, (size_t)B * H * W))) return rc;
  if ((rc = dev_alloc(h, &h->d_heat_consts, (size_t)B * 4))) return rc;
  if ((rc = dev_alloc(h, &h->d_cell_mask, (size_t)B * C))) return rc;
  if ((rc = dev_alloc(h, &h->d_kp_cell, (size_t)B * h->kmax))) return rc;
  if (h->sparse_db) {
    h->db_cap = (int)std::min<size_t>((size_t)4 * h->kmax, (size_t)C);
    if ((rc = dev_alloc(h, &h->d_db_list, (size_t)B * h->db_cap))) return rc;
    if ((rc = dev_alloc(h, &h->d_db_total, 16))) return rc;
    HIP_TRY(hipMemset(h->d_db_total, 0, 16 * sizeof(int)));
  }
  {   // select_kernel's global scratch: frames of more than 16,384 cells, and the lean form of pipelined calls
    if ((rc = dev_alloc(h, &h->d_sel_slot, (size_t)B * C))) return rc;
    if ((rc = dev_alloc(h, &h->d_sel_list, (size_t)B * C))) return rc;
    if (const char *e = getenv("SPFE_SELECT_LEAN")) h->select_lean = atoi(e);
  }
  {
    const char *qenv = getenv("SPFE_COV_QCAP");
    h->cov.qcap = qenv ? atoi(qenv) : 1024;
    if (h->cov.qcap < 16) h->cov.qcap = 16;
    if ((rc = dev_alloc(h, &h->cov.claim, (size_t)B * H * W))) return rc;
    if ((rc = dev_alloc(h, &h->cov.done, (size_t)B * H * W))) return rc;
    if ((rc = dev_alloc(h, &h->cov.queue, (size_t)B * h->kmax * h->cov.qcap))) return rc;
    if ((rc = dev_alloc(h, &h->cov.qval, (size_t)B * h->kmax * h->cov.qcap))) return rc;
    if ((rc = dev_alloc(h, &h->cov.npop, (size_t)B * h->kmax))) return rc;
    if ((rc = dev_alloc(h, &h->cov.dirty, (size_t)B * h->kmax))) return rc;
    if ((rc = dev_alloc(h, &h->cov.nxt, (size_t)B * h->kmax))) return rc;
    if ((rc = dev_alloc(h, &h->cov.nxy, (size_t)B * h->kmax * 2))) return rc;
    if ((rc = dev_alloc(h, &h->cov.workers, (size_t)B * h->kmax))) return rc;
    if ((rc = dev_alloc(h, &h->cov.counters, (size_t)B * 4))) return rc;
    h->cov.ecap = 32 * h->kmax;   // (~24 pops per keypoint on the dense synthetic detector, a quarter of the keypoints dirty)
    if (getenv("SPFE_COV_EDGES") && atoi(getenv("SPFE_COV_EDGES")) == 0) h->cov.ecap = 0;   // A/B: the link kernel walks the pop lists
    if (const char *e = getenv("SPFE_COV_ECAP")) h->cov.ecap = std::max(0, atoi(e));        // (tests: a list that overflows)
    if (h->cov.ecap && (rc = dev_alloc(h, &h->cov.edges, (size_t)B * h->cov.ecap * 2))) return rc;
    const char *oenv = getenv("SPFE_COV_OVF_SLOTS"), *cenv = getenv("SPFE_COV_OVF_CAP");
    h->cov.ovf_slots = oenv ? atoi(oenv) : 16;
    h->cov.ovf_cap = cenv ? atoi(cenv) : 16384;
    if (h->cov.ovf_slots < 0) h->cov.ovf_slots = 0;
    if (h->cov.ovf_cap < h->cov.qcap) h->cov.ovf_cap = h->cov.qcap;
    if ((rc = dev_alloc(h, &h->cov.ovf_slot, (size_t)B * h->kmax))) return rc;
    if ((rc = dev_alloc(h, &h->cov.ovf_q, (size_t)B * h->cov.ovf_slots * h->cov.ovf_cap + 1))) return rc;
    if ((rc = dev_alloc(h, &h->cov.ovf_v, (size_t)B * h->cov.ovf_slots * h->cov.ovf_cap + 1))) return rc;
    // the device-side last resort (cov.hip, cov_fallback_kernel): one list for the batch, 4 M pops by default (48 MB)
    const char *fenv = getenv("SPFE_COV_FALLBACK_CAP");
    h->cov.fb_cap = fenv ? atoi(fenv) : (1 << 22);
    if (h->cov.fb_cap < 1024) h->cov.fb_cap = 1024;
    if ((rc = dev_alloc(h, &h->cov.fb_q, (size_t)h->cov.fb_cap))) return rc;
    if ((rc = dev_alloc(h, &h->cov.fb_v, (size_t)h->cov.fb_cap))) return rc;
  }
  make_layout(h->kmax, C, (cfg->flags & SPFE_FLAG_DESC_BF16) != 0, &h->rl);
  if ((rc = dev_alloc(h, &h->d_records, (size_t)B * h->rl.bytes))) return rc;
  HIP_TRY(hipMemset(h->d_records, 0, (size_t)B * h->rl.bytes));

  // the MFMA conv chain
  struct Spec { int nl, l0, l1, src, dst; bool pool; };
  // src/dst index into act[]; -1 = head buffer
  const Spec specs[8] = {{1, 1, 0, 0, 1, true},  {1, 2, 0, 1, 2, false}, {1, 3, 0, 2, 3, true},
                         {1, 4, 0, 3, 4, false}, {1, 5, 0, 4, 5, true},  {1, 6, 0, 5, 6, false},
                         {1, 7, 0, 6, 7, false}, {2, 8, 10, 7, -1, false}};
  const char *senv = getenv("SPFE_SMALL_TILE_MAXH");  // override of the per-call choice in enqueue()
  const int small_maxh = senv ? atoi(senv) : -1;
  h->small_maxh = small_maxh;
  for (int i = 0; i < 8; ++i) {
    ConvLayer &L = h->layers[i];
    const int lids[2] = {specs[i].l0, specs[i].l1};
    if (h->bf16) rc = pack_layer_bf16(h, blob.data(), lids, specs[i].nl, &L);
    else rc = pack_layer(h, blob.data(), lids, specs[i].nl, &L);
    if (rc) return rc;
    L.pool = specs[i].pool;
    L.relu = true;
    L.H = lh[specs[i].src];
    L.W = lw[specs[i].src];
    L.small_tile = L.H <= small_maxh;
    L.in = h->act[specs[i].src];
    L.in_stride = lc[specs[i].src];
    L.in_choff = 0;
    if (specs[i].dst >= 0) { L.out = h->act[specs[i].dst]; L.out_stride = lc[specs[i].dst]; }
    else { L.out = h->d_head; L.out_stride = 512; }
    L.out_choff = 0;
  }
  {  // convPb: head[0:256] -> semi (65)
    ConvLayer &L = h->layers[8];
    const int lids[1] = {9};
    if ((rc = pack_layer(h, blob.data(), lids, 1, &L))) return rc;
    L.pool = false; L.relu = false; L.small_tile = true; L.H = H / 8; L.W = W / 8;
    L.in = h->d_head; L.in_stride = 512; L.in_choff = 0;
    L.out = h->d_semi; L.out_stride = SPFE_SEMI_CH; L.out_choff = 0;
  }
  {  // convDb: head[256:512] -> coarse (256)
    ConvLayer &L = h->layers[9];
    const int lids[1] = {11};
    if ((rc = pack_layer(h, blob.data(), lids, 1, &L))) return rc;
    L.pool = false; L.relu = false; L.small_tile = true; L.H = H / 8; L.W = W / 8;
    L.in = h->d_head; L.in_stride = 512; L.in_choff = 256;
    L.out = h->d_coarse; L.out_stride = SPFE_DESC_DIM; L.out_choff = 0;
  }
  if (h->bf16) {  // Cin = 64 layers selected for the wave-specialised kernel (conv1b by default)
    for (int i = 0; i < 4; ++i)
      if ((h->ws_mask >> i) & 1)
        if ((rc = pack_layer_bf16_ws(h, blob.data(), specs[i].l0, &h->d_wws[i]))) return rc;
    if (h->bf16_rw)
      for (int i = 4; i < 8; ++i) {
        const int lids2[2] = {specs[i].l0, specs[i].l1};   // (convPa | convDa for the last one)
        if ((rc = pack_layer_bf16_rw(h, blob.data(), lids2, specs[i].nl, &h->d_wrw[i - 4]))) return rc;
      }
    if ((rc = dev_alloc(h, &h->d_tile_ctr, 8 * 64))) return rc;   // [layer][part of the batch][32]
    h->sparse_da = h->sparse_db && h->d_wrw[3] && (size_t)B * C * 1024 < ((size_t)1 << 31);
    // Measured at 1280x720 x 8 (da_gather_bf16.hip): 25 us alone against the 43 us the dense launch loses without convDa, a
    // single-frame call's p50 0.357 -> 0.352 ms; but pipelined 7640 -> 7500 frames/s — a workgroup needs a whole CU (148 KB
    // of LDS, 380 registers), so beside the next batch's convolutions it only starts where one of theirs has ended, and
    // then holds that CU for its ~6 tiles.  So: synchronous calls only.
    if (const char *e = getenv("SPFE_SPARSE_DA")) h->sparse_da_mode = atoi(e);
    h->sparse_da = h->sparse_da && h->sparse_da_mode != 0;
    if (h->sparse_da && (rc = dev_alloc(h, &h->act7_alt, (size_t)B * C * 128))) return rc;
  }
  if (!h->bf16 && h->kc_mask != 0) {   // the K-chain kernel's weight tables (conv_f32_kc.hip): conv3a, conv4a, conv4b, convPa | convDa
    const int kl[4] = {3, 5, 6, 7};
    for (int q = 0; q < 4; ++q) {
      const int i = kl[q];
      const int l0 = specs[i].l0, l1 = specs[i].l1, nl = specs[i].nl;
      const spfe_layer_t &La = SPFE_LAYERS[l0];
      const int cout = La.cout + (nl == 2 ? SPFE_LAYERS[l1].cout : 0);
      if (!spfe::conv_f32_kc_supports(h->layers[i].H, h->layers[i].W, La.cin, cout)) continue;
      std::vector<float> wsrc((size_t)cout * La.cin * 9), wdst((size_t)cout * La.cin * 9);
      memcpy(wsrc.data(), blob.data() + blob_weight_offset(l0), (size_t)La.cout * La.cin * 9 * 4);
      if (nl == 2) memcpy(wsrc.data() + (size_t)La.cout * La.cin * 9, blob.data() + blob_weight_offset(l1), (size_t)SPFE_LAYERS[l1].cout * La.cin * 9 * 4);
      spfe::conv_f32_kc_pack_weights(wsrc.data(), La.cin, cout, wdst.data());
      if ((rc = dev_alloc(h, &h->d_wkc[i], wdst.size()))) return rc;
      HIP_TRY(hipMemcpy(h->d_wkc[i], wdst.data(), wdst.size() * 4, hipMemcpyHostToDevice));
    }
  }
  if (!h->bf16) {  // f32 heads with register-resident weights (head_f32.hip), bit-identical to the generic kernel — opt-in:
    // measured 63 + 38.5 us per eight 752x480 frames against 72 + 35.5 for the generic kernel (matrix-bound: 47 us at the peak)
    const char *fe = getenv("SPFE_F32_HEADS");
    if (fe) h->f32_heads = atoi(fe) != 0;
    if (const char *e = getenv("SPFE_PBTAIL")) h->pbtail = atoi(e) != 0;
    if (h->f32_heads) h->pbtail = false;
    if (h->pbtail) {
      const float *Wp = blob.data() + blob_weight_offset(9);   // layer 9 = convPb, [65][256]
      if ((rc = dev_alloc(h, &h->d_wpb_dust, 256))) return rc;
      HIP_TRY(hipMemcpy(h->d_wpb_dust, Wp + (size_t)64 * 256, 256 * 4, hipMemcpyHostToDevice));
    }
    for (int which = 0; which < 2; ++which) {
      if (!h->f32_heads && !(which == 0 && h->sparse_db) && !(which == 1 && h->pbtail)) continue;   // (the gathered descriptor head is head_f32.hip's kernel; pbtail_f32.hip reads convPb's table)
      const int lid = which ? 9 : 11;
      const spfe_layer_t &Ld = SPFE_LAYERS[lid];
      std::vector<float> w(spfe::head_f32_weight_bytes(Ld.cout) / 4, 0.0f);
      spfe::head_f32_pack_weights(blob.data() + blob_weight_offset(lid), Ld.cout, w.data());
      float **dst = which ? &h->d_wpb32 : &h->d_wdb32;
      if ((rc = dev_alloc(h, dst, w.size()))) return rc;
      HIP_TRY(hipMemcpy(*dst, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    }
    // convDa gathered as well (da_gather_f32.hip), pipelined calls included: 752x480 x 8, 19 k of 45 k cells listed: 122 us
    // (two workgroups per CU) against the 210 us the dense launch loses without convDa — pipelined 2035 -> 2057 ... 2075
    // frames/s, a single-frame call's p50 -1.4 %
    h->sparse_da_mode = 2;
    h->sparse_da = h->sparse_db && (size_t)B * C * 2048 < ((size_t)1 << 31);
    if (const char *e = getenv("SPFE_SPARSE_DA")) h->sparse_da_mode = atoi(e);
    h->sparse_da = h->sparse_da && h->sparse_da_mode != 0;
    if (h->sparse_da) {
      std::vector<float> w(spfe::da_gather_f32_weight_bytes() / 4);
      spfe::da_gather_f32_pack_weights(blob.data() + blob_weight_offset(10), w.data());   // layer 10 = convDa
      if ((rc = dev_alloc(h, &h->d_wda32, w.size()))) return rc;
      HIP_TRY(hipMemcpy(h->d_wda32, w.data(), w.size() * 4, hipMemcpyHostToDevice));
      if ((rc = dev_alloc(h, &h->act7_alt, (size_t)B * C * 128))) return rc;
    }
  }
  if (h->bf16) {  // both heads in bf16: convPa | convDa write bf16, convPb and convDb are head_bf16.hip's GEMMs
    if (const char *e = getenv("SPFE_PBTAIL")) h->pbtail = atoi(e) != 0;   // (convPb inside the tail's launch: pbtail_bf16.hip)
    if ((rc = dev_alloc(h, &h->d_hd, (size_t)B * C * 512))) return rc;
    for (int which = 0; which < 2; ++which) {
      const int lid = which ? 9 : 11;
      const spfe_layer_t &Ld = SPFE_LAYERS[lid];
      const float *Wd = blob.data() + blob_weight_offset(lid);
      std::vector<unsigned char> w(spfe::head_bf16_weight_bytes(Ld.cout), 0);
      std::vector<unsigned short> wb((size_t)Ld.cout * Ld.cin);
      for (size_t k = 0; k < wb.size(); ++k) wb[k] = host_bf16_rne(Wd[k]);
      spfe::head_bf16_pack_weights(wb.data(), Ld.cout, w.data());
      unsigned char **dst = which ? &h->d_wpb : &h->d_wdb;
      if ((rc = dev_alloc(h, dst, w.size()))) return rc;
      HIP_TRY(hipMemcpy(*dst, w.data(), w.size(), hipMemcpyHostToDevice));
    }
  }

  // pinned host mirrors for the host-facing calls
  if ((rc = host_alloc(h, &h->h_img, (size_t)B * H * W))) return rc;
  if ((rc = host_alloc(h, &h->h_records, (size_t)B * h->rl.bytes))) return rc;
  if ((rc = host_alloc(h, &h->h_heat_inv, (size_t)B * H * W))) return rc;
  if (cfg->flags & SPFE_FLAG_HEAT)
    if ((rc = host_alloc(h, &h->h_heat, (size_t)B * H * W))) return rc;
  HIP_TRY(hipDeviceSynchronize());
  return SPFE_OK;
}

#define STAGE_MARK(i) \
  do { if (h->timing && (h->timing_all || (i) == 1 || (i) == 2)) HIP_TRY(hipEventRecord(h->ev[i], s)); } while (0)

int enqueue_post(spfe_handle h, int n, uint8_t *d_records, hipStream_t s, const std::function<int()> *conv_db = nullptr, bool sparse = false, bool fused_pb = false, bool tail_done = false);
spfe::FrameBufs frame_bufs(spfe_handle h, uint8_t *d_records, bool sparse);
int tail_waits(spfe_handle h, uint8_t *d_records, hipStream_t s);
int launch_db_gathered(spfe_handle h, int n, hipStream_t s);

// D2H of the records by a kernel of our own that writes the pinned (device-mapped) host buffer: 8.9 MB in ~0.18 ms, no LDS,
// fits beside the persistent convolution workgroups; the runtime's own D2H path cost 0.36 ms more per batch in the pipeline
// (SPFE_PIPE_COPY_KERNEL=0 selects it)
}  // namespace
namespace spfe {   // (named, so that kernel traces show them: an anonymous namespace prints as "(anonymous namespace)::")
__global__ void copy_records_kernel(uint4 *dst, const uint4 *src, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
  __threadfence_system();
}

// ---- which stream for the second half batch?  HIP maps streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default;
// the assignment depends on what else the process has created), and two streams on ONE queue run their kernels one after the
// other: a second-half stream that shares the launch stream's queue (or the side stream's, whose kernels wait for events)
// turns the +2 % of the split into -3 %.  The runtime offers no query, so the library measures — on the DEVICE clock: two
// 150 us spin kernels, one per stream, each writing the wall_clock64 (100 MHz, one counter for the whole device) of its first
// and last instruction.  On different queues the two intervals overlap; on one queue the second starts when the first has
// ended.  No host timer is involved, so a preempted host thread cannot change the answer; the outcome is
// readable through spfe_debug_read("split_streams").  Once per launch stream (the first call that brings it synchronises that
// stream), up to four candidates; without a free queue — or when the stream is being captured — the split stays off.
__global__ void queue_probe_spin_kernel(long long ticks, long long *stamp) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) { stamp[0] = t0; stamp[1] = wall_clock64(); }
}
__global__ void zero_tile_counters_kernel(int *p, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0;
}
}  // namespace spfe
namespace {
// 1 = the two streams share a hardware queue, 0 = they do not, -1 = could not be measured (error / ambiguous twice)
int streams_share_a_queue(hipStream_t a, hipStream_t b, long long *h_stamp /* pinned, 4 entries */) {
  constexpr long long kTicks = 15000;   // 150 us
  if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return -1;
  for (int rep = 0; rep < 3; ++rep) {   // (rep 0 includes the kernel's code load: its stamps are not used)
    for (int i = 0; i < 4; ++i) h_stamp[i] = 0;
    hipLaunchKernelGGL(spfe::queue_probe_spin_kernel, dim3(1), dim3(64), 0, a, kTicks, h_stamp);
    hipLaunchKernelGGL(spfe::queue_probe_spin_kernel, dim3(1), dim3(64), 0, b, kTicks, h_stamp + 2);
    if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return -1;
    if (rep == 0) continue;
    const long long a0 = h_stamp[0], a1 = h_stamp[1], b0 = h_stamp[2], b1 = h_stamp[3];
    if (a1 <= a0 || b1 <= b0) continue;   // (a stamp did not arrive: try once more)
    // overlap of the two intervals against the spin length: none = one queue; more than half = two queues
    const long long ov = std::min(a1, b1) - std::max(a0, b0);
    if (ov <= kTicks / 10) return 1;
    if (ov >= kTicks / 2) return 0;
  }
  return -1;
}
int pick_conv2(spfe_handle h, hipStream_t s) {
  for (const auto &k : h->conv2_known)
    if (k.for_stream == s) { h->conv2 = k.conv2; h->conv2_ok = k.ok; h->split_probe = k.ok ? 1 : 0; return SPFE_OK; }
  if (h->conv2_known.size() >= 16) { h->conv2_ok = false; return SPFE_OK; }   // (a caller that keeps making streams: no split)
  h->conv2_ok = false;
  {   // a stream under capture cannot be synchronised or probed: no split for this call, and no answer is remembered
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { h->split_probe = -2; return SPFE_OK; }
  }
  struct Remember {   // whatever the outcome below, it is this stream's answer from now on
    spfe_handle h; hipStream_t s;
    ~Remember() { h->conv2_known.push_back({s, h->conv2, h->conv2_ok}); }
  } remember{h, s};
  if (!h->probe_stamp) {
    void *q = nullptr;
    HIP_TRY(hipHostMalloc(&q, 4 * sizeof(long long), hipHostMallocDefault));
    h->host_allocs.push_back(q);
    h->probe_stamp = reinterpret_cast<long long *>(q);
  }
  if (const char *e = getenv("SPFE_F32_SPLIT_PROBE"))   // 0: trust the first candidate (no measurement, no synchronisation)
    if (atoi(e) == 0) {
      if (h->conv2_pool.empty()) { hipStream_t c; HIP_TRY(hipStreamCreateWithFlags(&c, hipStreamNonBlocking)); h->conv2_pool.push_back(c); }
      h->conv2 = h->conv2_pool[0];
      h->conv2_ok = true;
      h->split_probe = 2;
      return SPFE_OK;
    }
  h->split_probe = 0;
  for (int k = 0; k < 4; ++k) {
    if ((int)h->conv2_pool.size() <= k) {
      hipStream_t c = nullptr;
      HIP_TRY(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
      h->conv2_pool.push_back(c);
    }
    hipStream_t c = h->conv2_pool[k];
    const int q1 = streams_share_a_queue(c, s, h->probe_stamp);
    const int q2 = q1 == 0 ? streams_share_a_queue(c, h->side, h->probe_stamp) : q1;
    if (q1 < 0 || q2 < 0) h->split_probe = -1;   // (could not be measured: counts as shared)
    if (q1 == 0 && q2 == 0) {
      h->conv2 = c;
      h->conv2_ok = true;
      h->split_probe = 1;
      break;
    }
  }
  return SPFE_OK;
}

// Enqueue the whole path for n frames already in device memory.
// (see spfe_handle_s::join_pending) orders `s` behind the half batch the last pipelined call left on the second stream
int settle_join(spfe_handle h, hipStream_t s) {
  if (h->join_pending) {
    HIP_TRY(wait_if_pending(s, h->ev_join));
    h->join_pending = false;
  }
  return SPFE_OK;
}

int enqueue(spfe_handle h, const uint8_t *d_images, int n, uint8_t *d_records, hipStream_t s) {
  const int H = h->H, W = h->W;
  if (h->timing) h->ev = h->evpool.data() + (size_t)(h->calls % spfe_handle_s::EVSETS) * (NSTAGE + 1);
  h->calls++;
  STAGE_MARK(0);
  // (a kernel of our own, not hipMemsetAsync: the runtime's fill is a blit that queues behind its other blits — the
  // pipelined host path's D2H copy of the PREVIOUS batch — and held the whole next batch back by 0.6 ms at 752x480 bf16)
  // bf16: conv1a is inside conv1b and the tile-queue counters may be in use by the half batch still running: the join first
  if (h->bf16) { const int rcj = settle_join(h, s); if (rcj) return rcj; }
  if (h->d_tile_ctr && !h->tile_ctr_clean) {
    hipLaunchKernelGGL(spfe::zero_tile_counters_kernel, dim3(1), dim3(256), 0, s, h->d_tile_ctr, 8 * 64);
    HIP_TRY(hipGetLastError());
  }
  h->tile_ctr_clean = false;   // (until this call's tail has been enqueued)
  const bool fused = !h->bf16 && h->fuse1a;  // f32: conv1b computes conv1a's outputs itself
  // bf16: when conv1b takes the wave-specialised kernel, its producer waves compute conv1a (no conv1a launch, no act0)
  const int grid_ws0 = std::max(16, (h->num_cus > 0 ? h->num_cus : 256) & ~15);
  const int ws_min = ((h->cfg.flags & SPFE_FLAG_ASYNC_COV) || h->pipe_mode) ? h->ws_min_items : h->ws_min_items_sync;
  const bool ws_layer0 = h->bf16 && h->d_wws[0] && W >= 32 &&
                         (long)((W + 31) / 32) * ((H + 7) / 8) * n >= (long)ws_min * grid_ws0;
  const bool fused16 = ws_layer0 && h->fuse1a_bf16;
  h->act0_missing = fused || fused16;
  // Pipelined two-half-batch steps: what the tails wait for (the side chain two tickets back: long finished, but the host
  // runs many steps ahead of the device, so these are real wait packets) is waited for in FRONT of conv1a — the packets are
  // then processed while the other half batch of the last step still runs, not between conv1a and conv1b with the chip idle.
  // Predicted from the last call's schedule; a wrong guess only repeats the (satisfied) waits later.  SPFE_EARLY_WAITS=0: off
  bool early_waits = false;
  {
    static const bool ew_env = !(getenv("SPFE_EARLY_WAITS") && atoi(getenv("SPFE_EARLY_WAITS")) == 0);
    if (ew_env && h->split_last && h->pbtail && n >= 2 && ((h->cfg.flags & SPFE_FLAG_ASYNC_COV) || h->pipe_mode) && !(h->timing && h->timing_all)) {
      const int rcw = tail_waits(h, d_records, s);
      if (rcw) return rcw;
      early_waits = true;
    }
  }
  if (h->bf16 && !fused16) HIP_TRY(spfe::launch_conv1a_bf16(d_images, h->d_w1a_tab, h->d_b1a, h->act[0], n, H, W, s));
  else if (!h->bf16 && !fused) HIP_TRY(spfe::launch_conv1a(d_images, h->d_w1a, h->d_b1a, h->act[0], n, H, W, s));
  // f32: conv1a (HBM-bound, reads the new frames, writes what conv1b of the last call has long read) runs beside the last
  // kernels of the half batch on the second stream; everything behind it waits for that half
  { const int rcj = settle_join(h, s); if (rcj) return rcj; }
  STAGE_MARK(1);
  // frames [f0, f0 + nfr) of the batch on stream `s` (the whole batch on the caller's stream by default)
  const int n_all = n;
  hipStream_t const s_all = s;
  // the descriptor head (bf16: convDa too) runs gathered, in enqueue_post
  const bool sparse = h->sparse_db && h->d_db_list && !(h->sparse_db_sync_only && ((h->cfg.flags & SPFE_FLAG_ASYNC_COV) || h->pipe_mode));
  const bool sparse_da = sparse && h->sparse_da && (h->sparse_da_mode >= 2 || !((h->cfg.flags & SPFE_FLAG_ASYNC_COV) || h->pipe_mode));
  h->sparse_da_call = sparse_da;
  const int par_db = (int)(h->ticket & 1);
  h->feat_cur = sparse_da && par_db ? h->act7_alt : h->act[7];
  bool tail_per_half = false;   // set below, when the layers behind conv1b run as two half batches
  auto run_layer = [&](int i, int f0 = 0, int nfr = -1, hipStream_t s_use = nullptr) -> int {
    const ConvLayer &L = h->layers[i];
    hipStream_t s = s_use ? s_use : s_all;
    const int n = nfr < 0 ? n_all : nfr;
    // convDb overwrites the coarse descriptor map the PREVIOUS call's descriptor sampling reads on
    // the side stream (pipelined callers): order it after that, by event, not by timing
    if (i == 9 && h->desc_recorded) HIP_TRY(wait_if_pending(s, h->ev_desc));
    // convPa | convDa overwrite the head activations the PREVIOUS call's gathered descriptor head reads (side stream).
    // sparse_da: the dense launch writes convPa's channels only, the gathered convDa / convDb touch the others; what the
    // gathered convDa reads is conv4b's output — kept twice, so conv4b waits for the call TWO tickets back
    if (!sparse_da && i == 7 && h->dbs_recorded[par_db ^ 1]) HIP_TRY(wait_if_pending(s, h->ev_dbs[par_db ^ 1]));
    if (sparse_da && i == 6 && h->dbs_recorded[par_db]) HIP_TRY(wait_if_pending(s, h->ev_dbs[par_db]));
    spfe::ConvParams p;
    p.in = L.in; p.in_stride = L.in_stride; p.in_choff = L.in_choff;
    p.wpack = L.d_w; p.bias = L.d_b;
    p.out = L.out; p.out_stride = L.out_stride; p.out_choff = L.out_choff; p.cout_real = L.cout_real;
    if (sparse_da && i == 6) p.out = const_cast<float *>(h->feat_cur);
    if (sparse_da && i == 7) p.in = h->feat_cur;
    p.B = n; p.H = L.H; p.W = L.W;
    const int part = f0 > 0 ? 1 : 0;
    // first frame of this part: byte offsets (bf16 activations are 2-byte elements behind the float pointers)
    auto shift = [&](const float *base, size_t elems) -> const float * {
      return reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + elems * (h->bf16 ? 2 : 4));
    };
    if (f0 > 0) {
      p.in = shift(p.in, (size_t)f0 * L.H * L.W * L.in_stride);
      p.out = const_cast<float *>(shift(p.out, (size_t)f0 * (L.pool ? (L.H / 2) * (L.W / 2) : L.H * L.W) * L.out_stride));
    }
    p.img = nullptr; p.w1a = nullptr; p.b1a = nullptr; p.tile_ctr = nullptr;
    if (i == 0 && fused) { p.img = d_images; p.w1a = h->d_w1a; p.b1a = h->d_b1a; }
    // tile height per layer and batch: 8-row tiles do 4 MFMAs per K step and wave
    // (better hidden side work), 4-row tiles give twice the work items; pick the
    // one with the shorter critical path over the persistent grid
    if (h->bf16 && i < 8) {
      // bf16 stack: 8-row tiles only; convPa/Da (i == 7) write f32 for the f32 heads
      p.tiles_x = (L.W + 31) / 32; p.tiles_y = (L.H + 7) / 8; p.nblk = L.nblk;
      p.num_cus = h->num_cus;
      // the wave-specialised kernel has the higher rate but ~8 us more start-up (512-thread workgroups, two
      // barriers before the first MFMA): it takes the launches with enough work items per workgroup
      // (tools/microbench/conv_ws_probe: the crossover is at ~10 items)
      const int grid_ws = (h->num_cus > 0 ? h->num_cus : 256) & ~15;
      if (i < 4 && h->d_wws[i] && L.W >= 32 && (long)p.tiles_x * p.tiles_y * n * p.nblk >= (long)ws_min * (grid_ws < 16 ? 16 : grid_ws)) {
        p.wpack = reinterpret_cast<const float *>(h->d_wws[i]);
        p.tile_ctr = h->d_tile_ctr + 64 * i + 32 * part;
        if (i == 0 && fused16) { p.img = d_images; p.w1a = reinterpret_cast<const float *>(h->d_w1a_tab); p.b1a = h->d_b1a; }
        // (experiment knobs, pipelined calls: conv1b / all Cin = 64 layers on fewer workgroups than CUs, so that the previous
        // batch's selection — 143 KB of LDS per workgroup, nothing fits beside this kernel's 158 KB — starts beside conv1b
        // instead of behind it.  1280x720 x 8: conv1b on 224 workgroups +0.3 ... 2 % whole path with conv1b at 0.51 - 0.53 of
        // peak instead of 0.57; 240 / 208 / 192: -2 / -1 / -3 %.  Not taken: HISTORY.md "Round 4")
        static const int ws_grid0 = getenv("SPFE_BF16_CONV1B_GRID") ? atoi(getenv("SPFE_BF16_CONV1B_GRID")) : 0;
        static const int ws_grid = getenv("SPFE_BF16_WS_GRID") ? atoi(getenv("SPFE_BF16_WS_GRID")) : 0;
        if (((h->cfg.flags & SPFE_FLAG_ASYNC_COV) || h->pipe_mode) && (i == 0 && ws_grid0 ? ws_grid0 : ws_grid) > 0)
          p.num_cus = i == 0 && ws_grid0 ? ws_grid0 : ws_grid;
        HIP_TRY(spfe::launch_conv_bf16_ws(p, L.pool, i == 0 ? (fused16 ? 2 : 1) : 0, s));
        STAGE_MARK(2 + i);
        return SPFE_OK;
      }
      // streamed-weight layers (Cin = 128): work items in queue order (conv_bf16.hip, CtlB::dyn); SPFE_BF16_DYN_QUEUE=0: static
      // (launches with a handful of items per workgroup stay static: the queue costs them more than it balances)
      // taller tiles for the streamed-weight layers when that still leaves every workgroup >= tile16_min_items items
      // (conv_bf16.hip, MT = 3 / 4: a stage's weight chunk feeds 1.5x / 2x the MFMAs).  Measured: 12-row tiles (layers
      // without a pool) -3...5 % on convPa|Da; 16-row tiles need 512 VGPRs + spills and lose 35 %: not the default.
      if (i == 7) {   // convPa | convDa: one launch, 512 output channels, bf16 (both 1x1 heads are bf16 GEMMs)
        p.out = reinterpret_cast<float *>(h->d_hd + (size_t)f0 * h->C * 512); p.out_stride = 512; p.out_choff = 0;
        if (sparse_da) p.nblk = L.nblk / 2;   // convPa only: convDa runs gathered, behind the selection (da_gather_bf16.hip)
      }
      // Cin = 128: weights resident in registers (conv_bf16_rw.hip) when every workgroup of a 128-channel group gets enough
      // tiles; 4-row tiles, or 2-row tiles for the small launches (twice the tiles)
      if (L.cin == 128 && h->bf16_rw && h->d_wrw[i - 4] && L.W >= 32 && !(L.W & 1) && !(L.pool && (L.H & 1))) {
        const int ncg = p.nblk / 2;
        const long wgs = std::max(8L * ncg, (long)((h->num_cus > 0 ? h->num_cus : 256) / (8 * ncg)) * 8 * ncg) / ncg;
        const long t4 = (long)p.tiles_x * ((L.H + 3) / 4) * n, t2 = (long)p.tiles_x * ((L.H + 1) / 2) * n;
        int tr = t4 >= (long)h->rw_min4 * wgs ? 4 : t2 >= (long)h->rw_min2 * wgs ? 2 : 0;
        // layers without a pool may take 3-row tiles: whichever of 4 / 3 rows needs fewer row-rounds on the slowest workgroup
        // (convPa|Da at 1280x720 x 8: 920 four-row tiles over 64 workgroups = 15 rounds of 4 rows, 1200 three-row tiles = 19 of 3)
        if (tr == 4 && !L.pool && h->rw_rows3) {
          const long t3 = (long)p.tiles_x * ((L.H + 2) / 3) * n;
          if (((t3 + wgs - 1) / wgs) * 3 < ((t4 + wgs - 1) / wgs) * 4) tr = 3;
        }
        if (tr) {
          p.wpack = reinterpret_cast<const float *>(h->d_wrw[i - 4]);
          p.nblk = ncg;
          p.tiles_y = (L.H + tr - 1) / tr;
          p.tile_ctr = h->d_tile_ctr + 64 * i + 32 * part;
          HIP_TRY(spfe::launch_conv_bf16_rw(p, L.pool, tr, s));
          STAGE_MARK(2 + i);
          return SPFE_OK;
        }
      }
      int tile_rows = 8;
      if (L.cin == 128 && h->tile16_min_items > 0) {
        const int tr = h->tile_rows_big > 0 ? h->tile_rows_big : 16;
        if ((tr == 16 || !L.pool) &&
            (long)p.tiles_x * ((L.H + tr - 1) / tr) * n * p.nblk >= (long)h->tile16_min_items * (grid_ws < 8 ? 8 : grid_ws)) {
          tile_rows = tr;
          p.tiles_y = (L.H + tr - 1) / tr;
        }
      }
      if (L.cin == 128 && h->bf16_dyn && (long)p.tiles_x * p.tiles_y * n * p.nblk >= 5L * (grid_ws < 8 ? 8 : grid_ws))
        p.tile_ctr = h->d_tile_ctr + 64 * i + 32 * part;
      HIP_TRY(spfe::launch_conv_bf16(p, L.cin, L.pool, false, s, tile_rows));
      STAGE_MARK(2 + i);
      return SPFE_OK;
    }
    if (h->bf16 && i >= 8) {  // convPb (65 logits) and convDb (256 descriptor channels): bf16 GEMMs over all cells of the batch
      const unsigned short *hd = h->d_hd + (size_t)f0 * h->C * 512;
      if (i == 8 && h->pbtail) {   // (inside the detector tail's launch: pbtail_bf16.hip; enqueue_post, or here per half batch)
        if (tail_per_half) {
          const spfe::FrameBufs fb = frame_bufs(h, d_records, sparse);
          // (each half clears ITS tile-queue counters [layer][part][32] for the next call: the other half's may be in use)
          HIP_TRY(spfe::launch_pbtail_bf16(h->d_hd, h->d_wpb, h->layers[8].d_b, h->d_semi, fb, h->rl, n, H, W, s, f0,
                                           h->d_tile_ctr ? h->d_tile_ctr + 32 * part : nullptr, h->d_tile_ctr ? 8 * 32 : 0, 64));
          if (h->d_tile_ctr) h->tile_ctr_clean = true;
        }
      }
      else if (i == 8) HIP_TRY(spfe::launch_head1x1_bf16(hd, h->d_wpb, L.d_b, h->d_semi + (size_t)f0 * h->C * SPFE_SEMI_CH, n * h->C, 65, s));
      else HIP_TRY(spfe::launch_head1x1_bf16(hd, h->d_wdb, L.d_b, h->d_coarse + (size_t)f0 * h->C * SPFE_DESC_DIM, n * h->C, 256, s));
      STAGE_MARK(2 + i);
      return SPFE_OK;
    }
    if (!h->bf16 && i == 8 && h->pbtail) {   // convPb runs inside the detector tail's launch (pbtail_f32.hip)
      if (tail_per_half) {   // two half batches on two streams: each half's tail right behind its convPa, beside the other half's layers
        const spfe::FrameBufs fb = frame_bufs(h, d_records, sparse);
        HIP_TRY(spfe::launch_pbtail_f32(h->d_head, h->d_wpb32, h->d_wpb_dust, h->layers[8].d_b, h->d_semi, fb, h->rl, n, H, W, s, f0));
      }
      STAGE_MARK(2 + i);
      return SPFE_OK;
    }
    if (!h->bf16 && i >= 8 && h->f32_heads) {  // convPb / convDb in f32: head_f32.hip (weights in registers)
      if (i == 8) HIP_TRY(spfe::launch_head1x1_f32(h->d_head, h->d_wpb32, L.d_b, h->d_semi, n * h->C, 65, s));
      else HIP_TRY(spfe::launch_head1x1_f32(h->d_head, h->d_wdb32, L.d_b, h->d_coarse, n * h->C, 256, s));
      STAGE_MARK(2 + i);
      return SPFE_OK;
    }
    if (!h->bf16 && L.ks == 3 && !L.pool && L.relu && i < 8 && h->d_wkc[i] && n_all == 1 && !(i == 0 && fused) &&
        (h->kc_mask > 0 ? ((h->kc_mask >> i) & 1) != 0 : h->kc_mask < 0 && ((kKcAuto >> i) & 1) &&
         // ... where at least two of its workgroups share a CU (they fill each other's staging stalls: convPa of a 752x480
         // frame, 480 workgroups, 45 -> 41 us; conv4a, 240 workgroups = one per CU, 27 -> 29 us: not taken)
         (long)((L.H + (2 * ((L.W + 15) / 16) <= 12 ? 2 : 1) - 1) / (2 * ((L.W + 15) / 16) <= 12 ? 2 : 1)) *
                 ((i == 7 && sparse_da ? L.cout_real / 2 : L.cout_real) / 16) * 2 >= 3L * (h->num_cus > 0 ? h->num_cus : 256))) {
      const int cout = i == 7 && sparse_da ? L.cout_real / 2 : L.cout_real;   // (convPa alone when convDa runs gathered)
      p.B = n; p.H = L.H; p.W = L.W;
      HIP_TRY(spfe::launch_conv_f32_kc(p, L.cin, cout, h->d_wkc[i], L.d_b, s));
      STAGE_MARK(2 + i);
      return SPFE_OK;
    }
    bool small_tile = L.small_tile, tiny_tile = false;
    if (L.ks == 3 && h->small_maxh < 0) {
      const long tx = (L.W + 31) / 32;
      const long nblk_eff = i == 7 && sparse_da ? L.nblk / 2 : L.nblk;   // (convPa alone when convDa runs gathered)
      const long items_big = tx * ((L.H + 7) / 8) * nblk_eff * n, items_small = tx * ((L.H + 3) / 4) * nblk_eff * n;
      const long g = h->num_cus > 0 ? h->num_cus : 256;
      const double cost_big = (double)((items_big + g - 1) / g) * 2.0 * 0.93;
      const double cost_small = (double)((items_small + g - 1) / g);
      small_tile = cost_small < cost_big;
      // 2-row tiles (layers without a pool): a single frame's low-resolution layers are 90 ... 360 four-row items on 256
      // CUs — one round of long items with CUs idle.  Half-height items cost 0.56 of a 4-row one (measured, batch 1:
      // conv4a / 4b 45 -> 27 us, convPa|Da 80 -> 64, conv3a 46 -> 38); at 8 frames per call the model keeps the taller tiles
      if (!L.pool && L.relu && !(i == 0 && fused) && h->tile2_auto) {
        const long items_tiny = tx * ((L.H + 1) / 2) * nblk_eff * n;
        const double cost_tiny = (double)((items_tiny + g - 1) / g) * 0.56;
        tiny_tile = cost_tiny < (cost_small < cost_big ? cost_small : cost_big);
      }
    }
    // a pooled layer as un-pooled 2-row tiles + a pool pass (single frames; see spfe_handle_s::pool_split)
    bool pool_split = false;
    if (L.ks == 3 && L.pool && L.relu && i > 0 && i < 7 && h->d_unpooled && h->pool_split != 0 && n_all == 1 &&   // (one scratch buffer: single-frame calls)
         !(L.H & 1) && !(L.W & 1) &&
        (size_t)n * L.H * L.W * L.out_stride <= (size_t)std::min(h->B, 2) * (H / 2) * (W / 2) * 64) {
      const long tx = (L.W + 31) / 32, g = h->num_cus > 0 ? h->num_cus : 256;
      const long items_big = tx * ((L.H + 7) / 8) * L.nblk * n, items_small = tx * ((L.H + 3) / 4) * L.nblk * n;
      const long items_tiny = tx * ((L.H + 1) / 2) * L.nblk * n;
      const double cost_big = (double)((items_big + g - 1) / g) * 2.0 * 0.93, cost_small = (double)((items_small + g - 1) / g);
      // (0.56: a 2-row item against a 4-row one, measured; 0.12: the pool pass — ~6 us against the ~50 us of a 4-row round at K = 1152)
      const double cost_tiny = (double)((items_tiny + g - 1) / g) * 0.56 + 0.12;
      pool_split = h->pool_split > 0 || cost_tiny < (cost_small < cost_big ? cost_small : cost_big) - 0.02;
    }
    if (pool_split) {
      spfe::ConvParams q = p;
      q.out = h->d_unpooled; q.out_stride = L.out_stride; q.out_choff = 0;
      q.tiles_x = (L.W + 31) / 32; q.tiles_y = (L.H + 1) / 2; q.nblk = L.nblk; q.num_cus = h->num_cus;
      HIP_TRY(spfe::launch_conv_f32(q, L.cin, L.ks, false, true, 3, 0, s));
      HIP_TRY(spfe::launch_pool2x2_f32(h->d_unpooled, p.out, n, L.H, L.W, L.out_stride, s));
      STAGE_MARK(2 + i);
      return SPFE_OK;
    }
    if (i == 0 && fused) small_tile = false;  // the fused first layer exists for 8-row tiles only
    int tile_mode = small_tile ? 1 : 0;
    if (L.ks == 3 && !(i == 0 && fused) && ((h->tile16_mask >> i) & 1)) tile_mode = 2;
    // conv1b: 16-row tiles of 4 wavefronts x 4 rows x 64 channels (6 operand reads per 8 MFMAs instead of 8; bit-identical):
    // measured on conv1b 2.2 ... 2.5 % per tile (640x480: 0.863 -> 0.882 of peak; 752x480: the coarser list costs 45 -> 46
    // round equivalents and it still gains 0.3 %; whole path +0.6 ... 0.8 %) — taken when its rounds are not more than 2.5 %
    // longer than the 8-row list's.  SPFE_TILE16X4=0: never, 2: always (one launch)
    // ... and when neither list divides well, BOTH: the first k tile rows (of 16) of the batch as 16-row tiles, the rest as
    // 8-row tiles in a second launch — 752x480 x 8: 224 of 240 tile rows = 21 rounds exactly + 768 eight-row tiles = 3
    // rounds exactly = 45 round equivalents, 42 of them at the 16-row rate (46 with 16-row tiles alone).  SPFE_TILE16X4=3:
    // no second launch
    long split16_rows = -1;   // >= 0: conv1b in two launches, 16-row tiles for the first split16_rows tile rows of the batch
    if (i == 0 && !fused && L.pool && h->tile16x4 && (tile_mode == 0 || h->tile16x4 == 2)) {
      const long g = h->num_cus > 0 ? h->num_cus : 256;
      const long tx = (L.W + 31) / 32, ty8 = (L.H + 7) / 8, ty16 = (L.H + 15) / 16;
      const long r8 = (tx * ty8 * n + g - 1) / g, r16 = (tx * ty16 * n + g - 1) / g;
      const double c8 = (double)r8, c16 = 2.0 * r16 * 0.975;
      double best = c8 < c16 ? c8 : c16;
      if (c16 < c8 || h->tile16x4 == 2) tile_mode = 4;
      if (h->tile16x4 == 1 && r8 >= 8)   // (large launches only: the second launch costs a kernel boundary)
        for (long k = ty16 * n - 1; k > 0 && k >= ty16 * n - 4 * ty16; --k) {
          const long f = k / ty16, r = k % ty16;                    // frames before f whole, r tile rows of frame f
          const long rows8 = (ty8 - std::min(2 * r, ty8)) + (n - f - 1) * ty8;
          const double c = 2.0 * ((tx * k + g - 1) / g) * 0.975 + (double)((tx * rows8 + g - 1) / g) + 0.3;
          if (c < best - 0.2) { best = c; split16_rows = k; }
        }
    }
    if (tiny_tile && tile_mode != 2 && tile_mode != 4) tile_mode = 3;
    if (L.ks == 3 && !L.pool && L.relu && ((h->tile2_mask >> i) & 1)) tile_mode = 3;
    const int th = spfe::conv_tile_rows(tile_mode);
    if (i == 0) { h->conv1b_tile_rows = th; h->conv1b_split_rows = (int)split16_rows; }
    p.tiles_x = (L.W + 31) / 32; p.tiles_y = (L.H + th - 1) / th; p.nblk = L.nblk;
    if (i == 7 && sparse_da) p.nblk = L.nblk / 2;   // convPa only: convDa runs gathered, behind the selection (da_gather_f32.hip)
    p.num_cus = h->num_cus;
    if (split16_rows > 0) {   // conv1b: 16-row tiles for the first split16_rows tile rows, 8-row tiles for the rest
      const int ty8 = (L.H + 7) / 8, ty16 = (L.H + 15) / 16;
      const long f = split16_rows / ty16, r = split16_rows % ty16;
      spfe::ConvParams p16 = p;
      p16.tiles_y = ty16;
      p16.item_lo = 0; p16.item_hi = (int)(p.tiles_x * split16_rows);
      HIP_TRY(spfe::launch_conv_f32(p16, L.cin, L.ks, L.pool, L.relu, 4, 1, s));
      // SPFE_STAGE_TIMING=2 brackets the dominant KERNEL: the 16-row launch (split16_rows of the batch's tile rows), not the pair
      const bool kernel_bracket = h->timing && !h->timing_all;
      if (kernel_bracket) HIP_TRY(hipEventRecord(h->ev[2], s));
      p.tiles_y = ty8;
      p.item_lo = (int)((f * ty8 + std::min<long>(2 * r, ty8)) * p.tiles_x);
      p.item_hi = p.tiles_x * ty8 * n;
      h->conv1b_tile_rows = 16;
      HIP_TRY(spfe::launch_conv_f32(p, L.cin, L.ks, L.pool, L.relu, 0, 1, s));
      if (!kernel_bracket) STAGE_MARK(2 + i);
      return SPFE_OK;
    }
    HIP_TRY(spfe::launch_conv_f32(p, L.cin, L.ks, L.pool, L.relu, tile_mode, i == 0 ? (fused ? 2 : 1) : 0, s));
    STAGE_MARK(2 + i);
    return SPFE_OK;
  };
  // The descriptor head (convDb) feeds only the descriptor sampling; the detector branch — tail, selection, heat
  // normalisation, covariance — does not wait for it.  So it is launched BEHIND the detector tail and runs beside the side
  // chain's first kernels (a synchronous single-frame call: p50 0.83 -> 0.80 ms at 752x480 f32, 0.38 -> 0.365 ms at 1280x720 bf16).  With
  // per-stage events (SPFE_STAGE_TIMING=1) the launch order stays the table's order.
  // Synchronous calls only: in the pipelined modes the side chain runs beside the NEXT batch anyway, and the deferred order
  // measured 0.3 ... 0.7 % slower there.
  const bool defer_db = !sparse && !(h->timing && h->timing_all) && h->defer_db && !((h->cfg.flags & SPFE_FLAG_ASYNC_COV) || h->pipe_mode);
  const int nlayers = sparse ? 9 : 10;   // sparse: convDb is enqueue_post's gathered launch behind the selection
  // f32, >= 2 frames: conv1b for the whole batch (its work list divides evenly over the CUs), then everything behind it as
  // two half batches on two streams: a layer's work list is 5.6 / 11.25 / 2.8 items per workgroup at 8 frames of 752x480, its
  // last round leaves most CUs idle, and the other half's kernel — independent frames — starts on exactly those CUs
  bool split = (h->bf16 ? (h->bf16_split >= 1 || (h->bf16_split < 0 && h->C < 10000)) : (h->f32_split >= 1 && !h->f32_heads)) && n >= 2 && !(h->timing && h->timing_all) && !defer_db;
  if (split) {
    const int rcp = pick_conv2(h, s);
    if (rcp) return rcp;
    split = h->conv2_ok;
  }
  h->split_last = split;
  if (split) {
    // The detector tail rides in convPb's launch, and with the halves on two streams each half's tail can run right behind its
    // convPa instead of behind the join (SPFE_TAIL_PER_HALF=0: behind the join) — what it must wait for (the side chain two
    // tickets back) is waited for HERE, on the launch stream in front of conv1b; the second stream forks behind conv1b
    static const bool tph_env = !(getenv("SPFE_TAIL_PER_HALF") && atoi(getenv("SPFE_TAIL_PER_HALF")) == 0);
    tail_per_half = h->pbtail && tph_env;
    if (tail_per_half && !early_waits) {
      const int rcw = tail_waits(h, d_records, s);
      if (rcw) return rcw;
    }
    int rc = run_layer(0);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(h->ev_fork, s));
    HIP_TRY(hipStreamWaitEvent(h->conv2, h->ev_fork, 0));
    // (SPFE_F32_SPLIT = number of parts, alternating between the two streams; 2 = halves)
    const int parts = 2;
    for (int q = 0; q < parts; q += 2)
      for (int i = 1; i < nlayers; ++i)
        for (int r = q; r < std::min(q + 2, parts); ++r) {
          const int f0 = (int)((long)n * r / parts), f1 = (int)((long)n * (r + 1) / parts);
          if ((rc = run_layer(i, f0, f1 - f0, (r & 1) ? h->conv2 : s))) return rc;
        }
    HIP_TRY(hipEventRecord(h->ev_join, h->conv2));
    // Pipelined calls whose tails ran per half: nothing on the launch stream needs the other half any more — the side chain
    // waits for it (enqueue_post), the launch stream in front of the next call's conv1b (settle_join).  A step's last kernel
    // — the second half's tail, ~20 us alone on the chip — and the event hop behind it (~13 us) leave the critical path
    // (f32 752x480 x 8: 3735 us steps, +0.9 %).
    if (tail_per_half && h->defer_join && ((h->cfg.flags & SPFE_FLAG_ASYNC_COV) || h->pipe_mode)) h->join_pending = true;
    else HIP_TRY(hipStreamWaitEvent(s, h->ev_join, 0));
    return enqueue_post(h, n, d_records, s, nullptr, sparse, h->pbtail, tail_per_half);
  }
  for (int i = 0; i < (defer_db ? 9 : nlayers); ++i) {
    const int rc = run_layer(i);
    if (rc) return rc;
  }
  if (sparse) STAGE_MARK(2 + 9);   // ("convDb" reads 0 on the launch stream: the gathered head is part of post_side)
  if (!defer_db) return enqueue_post(h, n, d_records, s, nullptr, sparse, h->pbtail);
  const std::function<int()> conv_db = [&]() -> int { return run_layer(9); };
  return enqueue_post(h, n, d_records, s, &conv_db, false, h->pbtail);
}

// The descriptor head on select_kernel's cell list (stream `s`, behind the selection of the same call).
int launch_db_gathered(spfe_handle h, int n, hipStream_t s) {
  const ConvLayer &L = h->layers[9];
  const int max_total = n * h->db_cap;
  if (h->bf16 && h->sparse_da_call)
    HIP_TRY(spfe::launch_da_gather_bf16(h->feat_cur, h->d_wrw[3], h->layers[7].d_b, h->d_hd, h->d_db_list, h->d_db_total, max_total, n, h->hc, h->wc, h->num_cus, s));
  if (!h->bf16 && h->sparse_da_call)
    HIP_TRY(spfe::launch_da_gather_f32(h->feat_cur, h->d_wda32, h->layers[7].d_b + 256, h->d_head, h->d_db_list, h->d_db_total, max_total, n, h->hc, h->wc, h->num_cus, s));
  if (h->bf16) HIP_TRY(spfe::launch_head1x1_bf16_gather(h->d_hd, h->d_wdb, L.d_b, h->d_coarse, n * h->C, h->d_db_list, h->d_db_total, max_total, h->db_tiles_per_wg, s));
  else HIP_TRY(spfe::launch_head1x1_f32_gather(h->d_head, h->d_wdb32, L.d_b, h->d_coarse, n * h->C, h->d_db_list, h->d_db_total, max_total, h->db_tiles_per_wg, s));
  return SPFE_OK;
}

// The dense descriptor head over the last call's head activations (spfe_debug_read("coarse") after a sparse call).
int launch_db_dense(spfe_handle h, int n, hipStream_t s) {
  const ConvLayer &L = h->layers[9];
  if (h->sparse_da_call) {   // convDa was gathered too: the same kernel over a list of ALL cells (a debug path)
    const int all = n * h->C;
    std::vector<int> cells((size_t)all + 1);
    for (int i = 0; i < all; ++i) cells[i] = i;
    cells[all] = all;
    int *d_tmp = nullptr;
    HIP_TRY(hipMalloc(&d_tmp, cells.size() * sizeof(int)));
    hipError_t e = hipMemcpy(d_tmp, cells.data(), cells.size() * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess)
      e = h->bf16 ? spfe::launch_da_gather_bf16(h->feat_cur, h->d_wrw[3], h->layers[7].d_b, h->d_hd, d_tmp, d_tmp + all, all, n, h->hc, h->wc, h->num_cus, s)
                  : spfe::launch_da_gather_f32(h->feat_cur, h->d_wda32, h->layers[7].d_b + 256, h->d_head, d_tmp, d_tmp + all, all, n, h->hc, h->wc, h->num_cus, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_tmp);
    HIP_TRY(e);
  }
  if (h->bf16) HIP_TRY(spfe::launch_head1x1_bf16(h->d_hd, h->d_wdb, L.d_b, h->d_coarse, n * h->C, 256, s));
  else HIP_TRY(spfe::launch_head1x1_f32(h->d_head, h->d_wdb32, L.d_b, h->d_coarse, n * h->C, 256, s));
  return SPFE_OK;
}

// The buffers of the call being enqueued (ticket h->ticket) as the tail / selection / covariance kernels see them.
spfe::FrameBufs frame_bufs(spfe_handle h, uint8_t *d_records, bool sparse) {
  spfe::FrameBufs f{};
  f.semi = h->d_semi; f.coarse = h->d_coarse;
  if (sparse) { f.db_list = h->d_db_list; f.db_total = h->d_db_total; }
  const int par = (int)(h->ticket & 1);
  f.heat_log = h->d_heat_log[par]; f.heat = h->d_heat; f.heat_inv = h->d_heat_inv;
  f.minmax = reinterpret_cast<uint32_t *>(h->d_minmax[par]);
  f.cell_score = h->d_cell_score[par]; f.cell_k = h->d_cell_k[par]; f.cell_mask = h->d_cell_mask; f.kp_cell = h->d_kp_cell;
  f.sel_slot = h->d_sel_slot; f.sel_list = h->d_sel_list;
  f.records = d_records; f.heat_consts = h->d_heat_consts;
  return f;
}

// What the detector tail of the call being enqueued must wait for (stream s).  A side chain still in flight: heat_inv, the
// covariance scratch and everything else that only side-stream kernels touch is ordered by that stream.  This call's tail
// writes the buffers of its ticket parity — last read by the chain two tickets back — and the dust maps inside the record
// buffer, so it waits for the previous chain only when the caller passes the same record buffer twice in a row.
int tail_waits(spfe_handle h, uint8_t *d_records, hipStream_t s) {
  if (h->cov_inflight) {
    const int NT = spfe_handle_s::NTICKET;
    if (h->ticket >= 2) HIP_TRY(wait_if_pending(s, h->ev_cov[(h->ticket - 2) % NT]));
    const int prev = (int)((h->ticket + NT - 1) % NT);
    static const bool old_order = getenv("SPFE_TAIL_WAITS_PREV") && atoi(getenv("SPFE_TAIL_WAITS_PREV"));   // A/B knob
    if (h->rec_of[prev] == d_records || old_order) HIP_TRY(wait_if_pending(s, h->ev_cov[prev]));
  }
  return SPFE_OK;
}

// Detector tail, selection, descriptors, covariance for n frames whose semi /
// coarse maps are in the handle's buffers.  tail_done: the detector tail (inside pbtail_f32_kernel) was launched per half
// batch by enqueue(), behind tail_waits().
int enqueue_post(spfe_handle h, int n, uint8_t *d_records, hipStream_t s, const std::function<int()> *conv_db, bool sparse, bool fused_pb, bool tail_done) {
  const int H = h->H, W = h->W;
  spfe::FrameBufs f = frame_bufs(h, d_records, sparse);
  h->sparse_last = sparse;
  const int par = (int)(h->ticket & 1);
  if (h->timing && !h->ev) return fail(SPFE_EINVAL, "internal: no event set");
  const int slot = (int)(h->ticket % spfe_handle_s::NTICKET);
  if (!tail_done) {
    const int rcw = tail_waits(h, d_records, s);
    if (rcw) return rcw;
  }
  h->rec_of[slot] = d_records;
  if (tail_done) {}
  else if (fused_pb && h->bf16) {
    HIP_TRY(spfe::launch_pbtail_bf16(h->d_hd, h->d_wpb, h->layers[8].d_b, h->d_semi, f, h->rl, n, H, W, s, 0, h->d_tile_ctr, h->d_tile_ctr ? 8 * 64 : 0));
    static const bool zit = !(getenv("SPFE_ZERO_IN_TAIL") && atoi(getenv("SPFE_ZERO_IN_TAIL")) == 0);   // A/B knob
    if (h->d_tile_ctr && zit) h->tile_ctr_clean = true;
  } else if (fused_pb) HIP_TRY(spfe::launch_pbtail_f32(h->d_head, h->d_wpb32, h->d_wpb_dust, h->layers[8].d_b, h->d_semi, f, h->rl, n, H, W, s));
  else {
    HIP_TRY(spfe::launch_tail(f, h->rl, n, H, W, s, h->d_tile_ctr, h->d_tile_ctr ? 8 * 64 : 0));
    static const bool zit = !(getenv("SPFE_ZERO_IN_TAIL") && atoi(getenv("SPFE_ZERO_IN_TAIL")) == 0);   // A/B knob
    if (h->d_tile_ctr && zit) h->tile_ctr_clean = true;
  }
  STAGE_MARK(12);
  // Synchronous calls with the gathered descriptor branch (a single frame's operator(): BASELINE configs[1]): the detector
  // branch is the critical path — tail -> selection -> covariance walk / classify / link / replay, a chain of latency-bound
  // kernels — and every cross-stream event hop on it costs ~13 us (measured on a batch-1 kernel timeline: tail -> side stream
  // 13.6 us, head -> replay 12.6 us), as much as the kernels it orders.  So the chain stays on the LAUNCH stream, without a
  // hop, and the descriptor branch (gathered convDa / convDb + sampling: needs the selection's cell list, shorter than the
  // covariance chain) takes the side stream: one hop at its start, beside the covariance kernels, and a join at the end that
  // has long been signalled.  (Round 3 ran it the other way round and let the replay launch carry the sampling: the replay
  // then waited for the gathered head — 28 us of a 0.80 ms call.)  SPFE_INLINE_CHAIN=0 restores that order.
  static const bool inline_env = !(getenv("SPFE_INLINE_CHAIN") && atoi(getenv("SPFE_INLINE_CHAIN")) == 0);
  if (inline_env && sparse && !conv_db && !((h->cfg.flags & SPFE_FLAG_ASYNC_COV) || h->pipe_mode) && !(h->timing && h->timing_all)) {
    if (h->cov_inflight) {   // (a pipelined call's chain still on the side stream — it owns heat_inv and the covariance scratch)
      HIP_TRY(hipStreamWaitEvent(s, h->ev_cov[(h->ticket + spfe_handle_s::NTICKET - 1) % spfe_handle_s::NTICKET], 0));
      h->cov_inflight = false;
    }
    STAGE_MARK(13);
    // (the event the side stream waits for is the selection's own completion signal: a hipEventRecord here put a marker
    // packet between the selection and the covariance walk — 7.6 us on the chain; SPFE_SEL_EXT_EVENT=0: that record)
    static const bool sel_ext_env = !(getenv("SPFE_SEL_EXT_EVENT") && atoi(getenv("SPFE_SEL_EXT_EVENT")) == 0);
    // (under stream capture the record it is: the stop event of an extended launch is not a capture node, the side stream
    // would not join the capture and its kernels would run once, at capture time)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool sel_ext = sel_ext_env && hipStreamIsCapturing(s, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone;
    HIP_TRY(spfe::launch_select(f, h->rl, n, H, W, h->cfg.num_features, s, &h->cov, h->rl.kmax, h->select_lean == 1,
                                sel_ext ? h->ev_sel : nullptr));
    if (!sel_ext) HIP_TRY(hipEventRecord(h->ev_sel, s));
    HIP_TRY(hipStreamWaitEvent(h->side, h->ev_sel, 0));
    int rc = launch_db_gathered(h, n, h->side);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(h->ev_dbs[par], h->side));
    h->dbs_recorded[par] = true;
    HIP_TRY(spfe::launch_desc(f, h->rl, n, H, W, h->side));
    HIP_TRY(hipEventRecord(h->ev_desc, h->side));
    h->desc_recorded = true;
    HIP_TRY(spfe::launch_cov(f, h->rl, h->cov, n, H, W, s, false, nullptr));
    HIP_TRY(hipStreamWaitEvent(s, h->ev_desc, 0));    // the join: records complete in `s` order
    HIP_TRY(hipEventRecord(h->ev_cov[slot], s));
    h->cov_inflight = false;
    h->ticket++;
    h->last_n = n;
    return SPFE_OK;
  }
  // Everything that only the finished record needs — selection (one latency-bound workgroup per frame), heat
  // normalisation (input of the covariance), descriptor sampling, covariance — goes to the side stream, ordered
  // after this call's detector tail: small kernels that run beside the next call's convolutions.
  HIP_TRY(hipEventRecord(h->ev_post[slot], s));
  HIP_TRY(hipStreamWaitEvent(h->side, h->ev_post[slot], 0));
  if (h->join_pending) HIP_TRY(hipStreamWaitEvent(h->side, h->ev_join, 0));   // (the other half batch: its tail ran on the second stream)
  STAGE_MARK(13);   // ("select" reads 0 on the launch stream: it is part of post_side)
  // (the heat normalisation rides in the selection's first launch: both depend on the detector tail only)
  {
    const bool pipelined = (h->cfg.flags & SPFE_FLAG_ASYNC_COV) || h->pipe_mode;
    HIP_TRY(spfe::launch_select(f, h->rl, n, H, W, h->cfg.num_features, h->side, &h->cov, h->rl.kmax,
                                h->select_lean == 1 || (h->select_lean < 0 && pipelined)));
  }
  // Synchronous calls: the descriptor sampling rides in the covariance replay launch (the chain's longest kernel) instead of
  // standing in front of the chain; pipelined calls keep it early — the NEXT call's convDb waits for it, and behind a replay
  // that shares the chip with that call's convolutions it would wait too long (0.5 ms steps in bf16 mode).
  const bool sync_call = !((h->cfg.flags & SPFE_FLAG_ASYNC_COV) || h->pipe_mode);
  // pipelined calls, sparse: nothing on the launch stream waits for the sampling any more (the dense convDb of the NEXT call
  // did), so it may ride in the replay launch there too: bf16 1280x720 +0.2 %, f32 752x480 -0.7 % (kept early in f32 mode)
  static const int sparse_dir_env = getenv("SPFE_SPARSE_DESC_IN_REPLAY") ? atoi(getenv("SPFE_SPARSE_DESC_IN_REPLAY")) : -1;
  const bool sparse_dir = sparse_dir_env < 0 ? h->bf16 : sparse_dir_env != 0;
  const bool desc_in_replay = h->desc_in_replay && (sync_call || (sparse && sparse_dir)) && !(h->timing && h->timing_all);
  hipEvent_t before_replay = nullptr;
  if (conv_db) {   // the descriptor head, launched behind the detector tail (enqueue()): the sampling waits for it
    const int rc = (*conv_db)();
    if (rc) return rc;
    HIP_TRY(hipEventRecord(h->ev_db, s));
    if (desc_in_replay) before_replay = h->ev_db;
    else HIP_TRY(hipStreamWaitEvent(h->side, h->ev_db, 0));
  }
  if (sparse) {
    // The gathered descriptor head.  Synchronous calls: on the launch stream, behind the selection, beside the covariance
    // chain's first kernels; the replay launch (which carries the sampling) waits for it.  Pipelined calls: in the side chain.
    if (desc_in_replay && sync_call) {
      HIP_TRY(hipEventRecord(h->ev_sel, h->side));
      HIP_TRY(hipStreamWaitEvent(s, h->ev_sel, 0));
      const int rc = launch_db_gathered(h, n, s);
      if (rc) return rc;
      HIP_TRY(hipEventRecord(h->ev_dbs[par], s));
      before_replay = h->ev_dbs[par];
    } else {
      const int rc = launch_db_gathered(h, n, h->side);
      if (rc) return rc;
      HIP_TRY(hipEventRecord(h->ev_dbs[par], h->side));
    }
    h->dbs_recorded[par] = true;
  }
  if (!desc_in_replay) {
    HIP_TRY(spfe::launch_desc(f, h->rl, n, H, W, h->side));
    HIP_TRY(hipEventRecord(h->ev_desc, h->side));  // d_coarse may be overwritten after this (next call's convDb)
    h->desc_recorded = true;
  }
  {
    // bf16 pipelined calls: fat replay workgroups (8 components each), so that the previous batch's replay holds ~120 CUs
    // instead of a wavefront on nearly every CU — a register-resident-weights convolution workgroup of THIS batch needs a
    // whole CU's registers (SPFE_REPLAY_WAVES=2|8 overrides)
    static const int rw_env = getenv("SPFE_REPLAY_WAVES") ? atoi(getenv("SPFE_REPLAY_WAVES")) : 0;
    // (measured, same-box A/B, 8 frames per call: bf16 1280x720 +0.7 %, bf16 752x480 -1.8 %, f32 -1 %: large bf16 frames only)
    const int rwv = rw_env ? rw_env : (h->bf16 && !sync_call && h->C >= 10000 ? 8 : 2);
    HIP_TRY(spfe::launch_cov(f, h->rl, h->cov, n, H, W, h->side, desc_in_replay, before_replay, rwv));
  }
  if (desc_in_replay) {
    HIP_TRY(hipEventRecord(h->ev_desc, h->side));
    h->desc_recorded = true;
  }
  HIP_TRY(hipEventRecord(h->ev_cov[slot], h->side));
  if (h->timing && h->timing_all) HIP_TRY(hipEventRecord(h->ev[14], h->side));
  h->cov_inflight = true;
  h->ticket++;
  if (!(h->cfg.flags & SPFE_FLAG_ASYNC_COV) && !h->pipe_mode) {
    // synchronous contract: the records are complete in `s` order when the call returns
    HIP_TRY(hipStreamWaitEvent(s, h->ev_cov[slot], 0));
    h->cov_inflight = false;
  }
  h->last_n = n;
  return SPFE_OK;
}

void view_record(const spfe_handle h, const uint8_t *rec, const float *heat, const float *heat_inv,
                 spfe_result *out) {
  const spfe::RecordLayout &r = h->rl;
  const int *hdr = reinterpret_cast<const int *>(rec + r.off_hdr);
  out->K = hdr[0];
  out->n_candidates = hdr[1];
  out->status = hdr[2];
  out->kp_xy = reinterpret_cast<const float *>(rec + r.off_xy);
  out->kp_response = reinterpret_cast<const float *>(rec + r.off_resp);
  out->desc = r.desc_bf16 ? nullptr : reinterpret_cast<const float *>(rec + r.off_desc);
  out->desc_bf16 = r.desc_bf16 ? reinterpret_cast<const uint16_t *>(rec + r.off_desc) : nullptr;
  out->cov2 = reinterpret_cast<const float *>(rec + r.off_cov);
  out->cov2_inv = reinterpret_cast<const float *>(rec + r.off_cinv);
  out->occ_grid = reinterpret_cast<const int16_t *>(rec + r.off_occ);
  out->dense_dust = reinterpret_cast<const float *>(rec + r.off_dd);
  out->semi_dust = reinterpret_cast<const float *>(rec + r.off_sd);
  out->heat = heat;
  out->heat_inv = heat_inv;
}

}  // namespace

extern "C" {

const char *spfe_last_error(void) { return g_err.c_str(); }
const char *spfe_version(void) { return "spfe 0.4 abi 4 (gfx950, f32-mfma + bf16-mfma)"; }
int spfe_abi_version(void) { return SPFE_ABI_VERSION; }
int spfe_check_abi(int abi_version, size_t sizeof_config, size_t sizeof_result, size_t sizeof_record_layout) {
  if (abi_version != SPFE_ABI_VERSION)
    return fail(SPFE_EINVAL, "ABI mismatch: the caller was built against spfe.h ABI %d, this library is ABI %d", abi_version, SPFE_ABI_VERSION);
  if (sizeof_config != sizeof(spfe_config) || sizeof_result != sizeof(spfe_result) || sizeof_record_layout != sizeof(spfe_record_layout))
    return fail(SPFE_EINVAL, "ABI mismatch: struct sizes config %zu / result %zu / record_layout %zu, library %zu / %zu / %zu",
                sizeof_config, sizeof_result, sizeof_record_layout, sizeof(spfe_config), sizeof(spfe_result), sizeof(spfe_record_layout));
  return SPFE_OK;
}
const char *spfe_stage_name(int i) { return (i >= 0 && i < NSTAGE) ? kStageNames[i] : ""; }

int spfe_create(const spfe_config *cfg, spfe_handle *out) {
  if (!cfg || !out) return fail(SPFE_EINVAL, "spfe_create: null argument");
  *out = nullptr;
  if (cfg->height <= 0 || cfg->width <= 0 || cfg->height % 8 || cfg->width % 8)
    return fail(SPFE_EINVAL, "image size %dx%d must be positive multiples of 8 (sp_extractor.cpp:70)",
                cfg->width, cfg->height);
  if (cfg->height < 16 || cfg->width < 16)
    return fail(SPFE_EINVAL, "image size %dx%d too small", cfg->width, cfg->height);
  if (cfg->num_features < 1 || cfg->num_features > 10000)
    return fail(SPFE_EINVAL, "num_features %d out of range (1..10000: the covariance link stage keeps 16 bytes per "
                             "keypoint in one workgroup's 160 KB of LDS)", cfg->num_features);
  if (cfg->max_batch < 1) return fail(SPFE_EINVAL, "max_batch must be >= 1");
  if (cfg->precision != SPFE_PRECISION_F32 && cfg->precision != SPFE_PRECISION_BF16)
    return fail(SPFE_EINVAL, "unsupported precision %d", cfg->precision);
  if ((size_t)(cfg->height / 8) * (cfg->width / 8) > spfe::select_max_cells() ||
      spfe::select_lds_bytes(cfg->height, cfg->width) > 160 * 1024)
    return fail(SPFE_EINVAL, "image %dx%d has more than 65,535 cells (e.g. 2560x1632): too large for the selection stage "
                             "(16-bit cell indices; 1920x1080 and 2560x1440 fit)", cfg->width, cfg->height);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(SPFE_EHIP, "no HIP device available (libspfe has no CPU path)");
  if (cfg->device < 0 || cfg->device >= ndev)
    return fail(SPFE_EINVAL, "device %d out of range (%d devices)", cfg->device, ndev);
  spfe_handle h = new spfe_handle_s();
  int rc = build(h, cfg);
  if (rc) {
    std::string keep = g_err;
    spfe_destroy(h);
    g_err = keep;
    return rc;
  }
  *out = h;
  return SPFE_OK;
}

void spfe_destroy(spfe_handle h) {
  if (!h) return;
  (void)hipSetDevice(h->cfg.device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->side) (void)hipStreamSynchronize(h->side);
  for (hipStream_t c : h->conv2_pool) (void)hipStreamSynchronize(c);
  for (int i = 0; i < spfe_handle_s::NTICKET; ++i) {
    if (h->ev_post[i]) (void)hipEventDestroy(h->ev_post[i]);
    if (h->ev_cov[i]) (void)hipEventDestroy(h->ev_cov[i]);
  }
  if (h->ev_desc) (void)hipEventDestroy(h->ev_desc);
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  for (hipStream_t c : h->conv2_pool) (void)hipStreamDestroy(c);
  if (h->ev_db) (void)hipEventDestroy(h->ev_db);
  if (h->ev_sel) (void)hipEventDestroy(h->ev_sel);
  for (int i = 0; i < 2; ++i) if (h->ev_dbs[i]) (void)hipEventDestroy(h->ev_dbs[i]);
  (void)spfe_comm_destroy(h);
  for (auto &ps : h->pipe) {
    if (ps.ev_h2d) (void)hipEventDestroy(ps.ev_h2d);
    if (ps.ev_done) (void)hipEventDestroy(ps.ev_done);
  }
  if (h->s_h2d) { (void)hipStreamSynchronize(h->s_h2d); (void)hipStreamDestroy(h->s_h2d); }
  if (h->s_d2h) { (void)hipStreamSynchronize(h->s_d2h); (void)hipStreamDestroy(h->s_d2h); }
  if (h->side) (void)hipStreamDestroy(h->side);
  for (void *p : {(void *)h->d_map_x, (void *)h->d_map_y, (void *)h->d_raw})
    if (p) (void)hipFree(p);
  if (h->h_raw) (void)hipHostFree(h->h_raw);
  for (void *p : {(void *)h->p_cidx, (void *)h->p_cdist, (void *)h->p_stage})
    if (p) (void)hipFree(p);
  for (void *p : {(void *)h->m_best_t, (void *)h->m_best_q, (void *)h->m_stage_q, (void *)h->m_stage_t,
                  (void *)h->m_out, (void *)h->m_out2})
    if (p) (void)hipFree(p);
  for (void *p : h->dev_allocs) (void)hipFree(p);
  for (void *p : h->host_allocs) (void)hipHostFree(p);
  for (auto &e : h->evpool)
    if (e) (void)hipEventDestroy(e);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int spfe_get_record_layout(spfe_handle h, spfe_record_layout *o) {
  if (!h || !o) return fail(SPFE_EINVAL, "null argument");
  const spfe::RecordLayout &r = h->rl;
  o->bytes = r.bytes; o->kmax = r.kmax; o->off_hdr = r.off_hdr; o->off_xy = r.off_xy;
  o->off_resp = r.off_resp; o->off_cov = r.off_cov; o->off_cinv = r.off_cinv;
  o->off_desc = r.off_desc; o->off_occ = r.off_occ; o->off_dd = r.off_dd; o->off_sd = r.off_sd;
  o->desc_elem_bytes = r.desc_bf16 ? 2 : 4;
  return SPFE_OK;
}

size_t spfe_record_bytes(spfe_handle h) { return h ? h->rl.bytes : 0; }

int spfe_extract_batch_device(spfe_handle h, const void *d_images, int n, void *d_records, void *stream) {
  if (!h) return fail(SPFE_EINVAL, "null handle");
  if (!d_images) return fail(SPFE_EEMPTY, "input image is empty");
  if (n < 1 || n > h->B) return fail(SPFE_EINVAL, "batch %d not in [1, %d]", n, h->B);
  HIP_TRY(hipSetDevice(h->cfg.device));
  hipStream_t s = stream ? reinterpret_cast<hipStream_t>(stream) : h->stream;
  uint8_t *rec = d_records ? reinterpret_cast<uint8_t *>(d_records) : h->d_records;
  return enqueue(h, reinterpret_cast<const uint8_t *>(d_images), n, rec, s);
}

int finish_host(spfe_handle h, int n, spfe_result *outs);

int spfe_postprocess(spfe_handle h, const float *semi, const float *coarse, int n, spfe_result *outs) {
  if (!h || !outs || !semi || !coarse) return fail(SPFE_EINVAL, "null argument");
  if (n < 1 || n > h->B) return fail(SPFE_EINVAL, "batch %d not in [1, %d]", n, h->B);
  HIP_TRY(hipSetDevice(h->cfg.device));
  hipStream_t s = h->stream;
  { const int rcj = settle_join(h, s); if (rcj) return rcj; }
  if (h->desc_recorded) HIP_TRY(hipStreamWaitEvent(s, h->ev_desc, 0));  // previous call's reader of d_coarse
  HIP_TRY(hipMemcpyAsync(h->d_semi, semi, (size_t)n * h->C * SPFE_SEMI_CH * 4, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(h->d_coarse, coarse, (size_t)n * h->C * SPFE_DESC_DIM * 4, hipMemcpyHostToDevice, s));
  if (h->timing) h->ev = h->evpool.data() + (size_t)(h->calls % spfe_handle_s::EVSETS) * (NSTAGE + 1);
  if (h->timing) for (int i = 0; i <= 11; ++i) HIP_TRY(hipEventRecord(h->ev[i], s));
  h->calls++;
  int rc = enqueue_post(h, n, h->d_records, s);
  if (rc) return rc;
  return finish_host(h, n, outs);
}

int spfe_extract_batch(spfe_handle h, const uint8_t *const *images, int stride, int n, spfe_result *outs) {
  if (!h || !outs) return fail(SPFE_EINVAL, "null argument");
  if (!images) return fail(SPFE_EEMPTY, "input image is empty");
  if (n < 1 || n > h->B) return fail(SPFE_EINVAL, "batch %d not in [1, %d]", n, h->B);
  const int H = h->H, W = h->W;
  if (stride < W) return fail(SPFE_EINVAL, "stride %d smaller than width %d", stride, W);
  for (int i = 0; i < n; ++i) {
    if (!images[i]) return fail(SPFE_EEMPTY, "input image is empty");  // sp_extractor.cpp:364-365
    for (int y = 0; y < H; ++y)
      memcpy(h->h_img + ((size_t)i * H + y) * W, images[i] + (size_t)y * stride, W);
  }
  HIP_TRY(hipSetDevice(h->cfg.device));
  hipStream_t s = h->stream;
  HIP_TRY(hipMemcpyAsync(h->d_img, h->h_img, (size_t)n * H * W, hipMemcpyHostToDevice, s));
  int rc = enqueue(h, h->d_img, n, h->d_records, s);
  if (rc) return rc;
  return finish_host(h, n, outs);
}

// D2H of the records (+ maps), sync, host views.
int finish_host(spfe_handle h, int n, spfe_result *outs) {
  const int H = h->H, W = h->W;
  hipStream_t s = h->stream;
  if (h->cov_inflight) {
    const int prev = (int)((h->ticket + spfe_handle_s::NTICKET - 1) % spfe_handle_s::NTICKET);
    HIP_TRY(hipStreamWaitEvent(s, h->ev_cov[prev], 0));
    h->cov_inflight = false;
  }
  const bool want = (h->cfg.flags & SPFE_FLAG_HEAT) != 0;
  // (the synchronous path keeps the runtime's copy: a copy kernel as in spfe_submit_batch measured +4 % in f32 and -4 % in
  // bf16 mode here, nothing for a single frame)
  HIP_TRY(hipMemcpyAsync(h->h_records, h->d_records, (size_t)n * h->rl.bytes, hipMemcpyDeviceToHost, s));
  if (want) {
    HIP_TRY(hipMemcpyAsync(h->h_heat_inv, h->d_heat_inv, (size_t)n * H * W * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(h->h_heat, h->d_heat, (size_t)n * H * W * 4, hipMemcpyDeviceToHost, s));
  }
  HIP_TRY(hipStreamSynchronize(s));
  for (int i = 0; i < n; ++i) {
    uint8_t *rec = h->h_records + (size_t)i * h->rl.bytes;
    float *hinv = h->h_heat_inv + (size_t)i * H * W;
    view_record(h, rec, want ? h->h_heat + (size_t)i * H * W : nullptr, want ? hinv : nullptr, &outs[i]);
  }
  return SPFE_OK;
}

int spfe_extract(spfe_handle h, const uint8_t *image, int stride, spfe_result *out) {
  if (!image) return fail(SPFE_EEMPTY, "input image is empty");
  const uint8_t *imgs[1] = {image};
  return spfe_extract_batch(h, imgs, stride, 1, out);
}

long spfe_last_ticket(spfe_handle h) { return h ? h->ticket - 1 : -1; }

int spfe_wait_records(spfe_handle h, long ticket, void *stream) {
  if (!h) return fail(SPFE_EINVAL, "null handle");
  if (ticket < 0 || ticket >= h->ticket || ticket + spfe_handle_s::NTICKET <= h->ticket)
    return fail(SPFE_EINVAL, "ticket %ld is not one of the last %d calls", ticket, spfe_handle_s::NTICKET);
  HIP_TRY(hipSetDevice(h->cfg.device));
  hipStream_t s = stream ? reinterpret_cast<hipStream_t>(stream) : h->stream;
  HIP_TRY(hipStreamWaitEvent(s, h->ev_cov[ticket % spfe_handle_s::NTICKET], 0));
  return SPFE_OK;
}

int spfe_view_record(spfe_handle h, const void *host_record, spfe_result *out) {
  if (!h || !host_record || !out) return fail(SPFE_EINVAL, "null argument");
  view_record(h, reinterpret_cast<const uint8_t *>(host_record), nullptr, nullptr, out);
  return SPFE_OK;
}

long spfe_debug_read(spfe_handle h, const char *name, int frame, void *dst, size_t cap) {
  if (!h || !name || !dst) return fail(SPFE_EINVAL, "null argument");
  if (frame < 0 || frame >= h->B) return fail(SPFE_EINVAL, "frame %d out of range", frame);
  const size_t C = h->C, HW = (size_t)h->H * h->W;
  if (std::string(name) == "conv1b_split_rows") {
    if (cap < sizeof(int)) return fail(SPFE_EINVAL, "buffer 'conv1b_split_rows' needs 4 bytes");
    *reinterpret_cast<int *>(dst) = h->conv1b_split_rows;
    return (long)sizeof(int);
  }
  if (std::string(name) == "conv1b_tile_rows") {   // f32: which conv1b instantiation the last call launched (8 or 16 rows per tile)
    if (cap < sizeof(int)) return fail(SPFE_EINVAL, "buffer 'conv1b_tile_rows' needs 4 bytes");
    *reinterpret_cast<int *>(dst) = h->conv1b_tile_rows;
    return (long)sizeof(int);
  }
  if (std::string(name) == "split_streams") {   // [2] int: outcome of the queue probe (see spfe_handle_s::split_probe), and whether the last call ran as two half batches
    if (cap < 2 * sizeof(int)) return fail(SPFE_EINVAL, "buffer 'split_streams' needs 8 bytes");
    reinterpret_cast<int *>(dst)[0] = h->split_probe;
    reinterpret_cast<int *>(dst)[1] = h->split_last ? 1 : 0;
    return (long)(2 * sizeof(int));
  }
  if (std::string(name) == "da_gathered") {   // 1: the last call ran convDa on the listed cells only (host-side flag)
    if (cap < sizeof(int)) return fail(SPFE_EINVAL, "buffer 'da_gathered' needs 4 bytes");
    *reinterpret_cast<int *>(dst) = h->sparse_last && h->sparse_da_call ? 1 : 0;
    return (long)sizeof(int);
  }
  const void *src = nullptr;
  size_t bytes = 0;
  bool bf16_src = false;
  std::string nm(name);
  if (nm == "semi") { src = h->d_semi + frame * C * SPFE_SEMI_CH; bytes = C * SPFE_SEMI_CH * 4; }
  else if (nm == "coarse" || nm == "coarse_sparse") {
    src = h->d_coarse + frame * C * SPFE_DESC_DIM; bytes = C * SPFE_DESC_DIM * 4;
    if (nm == "coarse" && h->sparse_last) {   // the last call wrote only the rows its keypoints read: complete the map
      if (hipSetDevice(h->cfg.device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return fail(SPFE_EHIP, "debug read of 'coarse' failed");
      const int rc = launch_db_dense(h, h->last_n, h->stream);
      if (rc) return rc;
      h->sparse_last = false;
    }
  }
  else if (nm == "db_total" && h->d_db_total) { src = h->d_db_total; bytes = 4; }
  else if (nm == "db_list" && h->d_db_list) { src = h->d_db_list; bytes = (size_t)h->B * h->db_cap * 4; }
  else if (nm == "head") {
    if (h->bf16) return fail(SPFE_EINVAL, "'head' is f32 only: the bf16 mode keeps ReLU(convPa) | ReLU(convDa) as bf16");
    src = h->d_head + frame * C * 512; bytes = C * 512 * 4;
  }
  else if (nm == "heat_log") { src = h->d_heat_log[(h->ticket + 1) & 1] + frame * HW; bytes = HW * 4; }
  else if (nm == "heat_inv") { src = h->d_heat_inv + frame * HW; bytes = HW * 4; }
  else if (nm == "heat" && h->d_heat) { src = h->d_heat + frame * HW; bytes = HW * 4; }
  else if (nm == "image") { src = h->d_img + frame * HW; bytes = HW; }
  else if (nm == "cell_score") { src = h->d_cell_score[(h->ticket + 1) & 1] + frame * C; bytes = C * 4; }
  else if (nm == "cov_counters") { src = h->cov.counters + frame * 4; bytes = 16; }
  else if (nm == "cov_nxt") { src = h->cov.nxt + (size_t)frame * h->kmax; bytes = (size_t)h->kmax * 4; }
  else if (nm == "cov_workers") { src = h->cov.workers + (size_t)frame * h->kmax; bytes = (size_t)h->kmax * 4; }
  else if (nm == "cov_npop") { src = h->cov.npop + (size_t)frame * h->kmax; bytes = (size_t)h->kmax * 4; }
  else if (nm == "feat") { src = (h->feat_cur ? h->feat_cur : h->act[7]) + frame * C * 128; bytes = C * 128 * 4; bf16_src = h->bf16; }
  else if (nm.size() == 4 && nm.compare(0, 3, "act") == 0 && nm[3] >= '0' && nm[3] <= '7') {
    const int i = nm[3] - '0';
    if (i == 0 && h->act0_missing)
      return fail(SPFE_EINVAL, "act0 is not materialised: conv1a was fused into conv1b in the last call");
    const int lh[8] = {1, 2, 2, 4, 4, 8, 8, 8};
    const int lc[8] = {64, 64, 64, 64, 128, 128, 128, 128};
    const size_t per = (size_t)(h->H / lh[i]) * (h->W / lh[i]) * lc[i];
    // (conv4b's output exists twice, by ticket parity, when convDa runs gathered: "act7" is the last call's, like "feat")
    src = (i == 7 && h->feat_cur ? h->feat_cur : h->act[i]) + frame * per; bytes = per * 4; bf16_src = h->bf16;
  } else return fail(SPFE_EINVAL, "unknown debug buffer '%s'", name);
  if (bytes > cap) return fail(SPFE_EINVAL, "buffer '%s' needs %zu bytes, cap %zu", name, bytes, cap);
  // both streams: with SPFE_FLAG_ASYNC_COV heat / heat_inv are written on the side stream
  if (hipSetDevice(h->cfg.device) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess ||
      hipStreamSynchronize(h->side) != hipSuccess)
    return fail(SPFE_EHIP, "debug read of '%s' failed", name);
  if (bf16_src) {
    // bf16 mode keeps the conv stack's activations as bf16 NHWC (half the elements' bytes, the frame offset
    // in bf16 elements): read them as such and widen to the f32 the caller expects
    const size_t n = bytes / 4;
    std::vector<unsigned short> tmp(n);
    const unsigned short *bsrc = reinterpret_cast<const unsigned short *>(
        nm == "feat" || nm == "act7" ? (const void *)(h->feat_cur ? h->feat_cur : h->act[7]) : (const void *)h->act[nm[3] - '0']) + (size_t)frame * n;
    if (hipMemcpy(tmp.data(), bsrc, n * 2, hipMemcpyDeviceToHost) != hipSuccess)
      return fail(SPFE_EHIP, "debug read of '%s' failed", name);
    uint32_t *d32 = reinterpret_cast<uint32_t *>(dst);
    for (size_t k = 0; k < n; ++k) d32[k] = (uint32_t)tmp[k] << 16;
    return (long)bytes;
  }
  if (hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) != hipSuccess)
    return fail(SPFE_EHIP, "debug read of '%s' failed", name);
  return (long)bytes;
}

int spfe_stage_reset(spfe_handle h) {
  if (!h) return fail(SPFE_EINVAL, "null handle");
  h->calls_at_reset = h->calls;
  return SPFE_OK;
}

// Average per-stage GPU time (ms) over the calls since spfe_stage_reset (at most
// the last EVSETS calls).  Needs SPFE_STAGE_TIMING=1 at spfe_create.
int spfe_stage_times(spfe_handle h, float *ms, int cap) {
  if (!h || !ms) return fail(SPFE_EINVAL, "null argument");
  if (!h->timing || h->calls == h->calls_at_reset) return 0;
  if (hipSetDevice(h->cfg.device) != hipSuccess) return fail(SPFE_EHIP, "hipSetDevice failed");
  long first = h->calls_at_reset;
  if (h->calls - first > spfe_handle_s::EVSETS) first = h->calls - spfe_handle_s::EVSETS;
  const int nst = cap < NSTAGE ? cap : NSTAGE;
  std::vector<double> acc(NSTAGE, 0.0);
  for (long c = first; c < h->calls; ++c) {
    hipEvent_t *ev = h->evpool.data() + (size_t)(c % spfe_handle_s::EVSETS) * (NSTAGE + 1);
    if (!h->timing_all) {  // only the bracket of the dominant kernel was recorded
      float t = 0;
      if (hipEventSynchronize(ev[2]) != hipSuccess) return fail(SPFE_EHIP, "event sync failed");
      (void)hipEventElapsedTime(&t, ev[1], ev[2]);
      acc[1] += t;
      continue;
    }
    if (hipEventSynchronize(ev[NSTAGE - 1]) != hipSuccess) return fail(SPFE_EHIP, "event sync failed");
    for (int i = 0; i < NSTAGE - 1; ++i) {
      float t = 0;
      (void)hipEventElapsedTime(&t, ev[i], ev[i + 1]);
      acc[i] += t;
    }
    float t = 0;
    (void)hipEventElapsedTime(&t, ev[0], ev[NSTAGE - 1]);
    acc[NSTAGE - 1] += t;
  }
  for (int i = 0; i < nst; ++i) ms[i] = (float)(acc[i] / (double)(h->calls - first));
  return nst;
}

// ---- direct "dust" alignment (optimizer_dust.cpp:170-294) -----------------
namespace {
int dust_check(spfe_handle h, int n, const spfe_dust_params *prm) {
  if (n < 0 || n > SPFE_DUST_MAX_POINTS) return fail(SPFE_EINVAL, "n_points %d not in [0, %d]", n, SPFE_DUST_MAX_POINTS);
  if (prm->max_iterations < 0 || prm->max_iterations > 1000) return fail(SPFE_EINVAL, "max_iterations %d", prm->max_iterations);
  if (!(prm->huber_delta > 0)) return fail(SPFE_EINVAL, "huber_delta must be positive");
  if (spfe::dust_lds_bytes(h->hc, h->wc) > 160 * 1024) return fail(SPFE_EINVAL, "dust map %dx%d too large for LDS", h->wc, h->hc);
  return SPFE_OK;
}
int dust_launch(spfe_handle h, const float *d_dust, const float *d_pts, int n, const float *d_T,
                const spfe_dust_params *prm, uint8_t *d_out, hipStream_t s, int nframes = 1, size_t dust_stride = 0,
                const int *d_n = nullptr) {
  spfe::DustArgs a{};
  a.nframes = nframes; a.dust_stride = dust_stride; a.pts_stride = (size_t)SPFE_DUST_MAX_POINTS * 12; a.pose_stride = 64;
  a.out_stride = SPFE_DUST_OUT_BYTES; a.n_dev = d_n;
  a.dust = d_dust; a.hc = h->hc; a.wc = h->wc; a.pts = d_pts; a.n = n; a.Tcw_in = d_T;
  a.fx = prm->fx; a.fy = prm->fy; a.cx = prm->cx; a.cy = prm->cy;
  a.max_iterations = prm->max_iterations; a.delta = prm->huber_delta; a.inlier_chi2 = prm->inlier_chi2;
  a.Tcw_out = reinterpret_cast<float *>(d_out);
  a.counts = reinterpret_cast<int *>(d_out + 64);
  a.uv = reinterpret_cast<float *>(d_out + SPFE_DUST_OFF_UV);
  a.inlier = d_out + SPFE_DUST_OFF_INLIER;
  HIP_TRY(spfe::launch_dust_align(a, s));
  return SPFE_OK;
}
}  // namespace

int spfe_align_dust_record_device(spfe_handle h, const void *d_record, const void *d_points_xyz, int n,
                                  const void *d_Tcw, const spfe_dust_params *prm, void *d_out, void *stream) {
  if (!h || !d_record || !d_Tcw || !prm || !d_out || (n > 0 && !d_points_xyz)) return fail(SPFE_EINVAL, "null argument");
  int rc = dust_check(h, n, prm);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(h->cfg.device));
  hipStream_t s = stream ? reinterpret_cast<hipStream_t>(stream) : h->stream;
  const float *d_dust = reinterpret_cast<const float *>(reinterpret_cast<const uint8_t *>(d_record) + h->rl.off_dd);
  return dust_launch(h, d_dust, reinterpret_cast<const float *>(d_points_xyz), n, reinterpret_cast<const float *>(d_Tcw),
                     prm, reinterpret_cast<uint8_t *>(d_out), s);
}

int spfe_align_dust_batch_device(spfe_handle h, const void *d_records, int n_frames, const void *d_points_xyz,
                                 const void *d_n_points, const void *d_Tcw, const spfe_dust_params *prm, void *d_out,
                                 void *stream) {
  if (!h || !d_records || !d_Tcw || !prm || !d_out || !d_points_xyz || !d_n_points) return fail(SPFE_EINVAL, "null argument");
  if (n_frames < 1 || n_frames > 65535) return fail(SPFE_EINVAL, "n_frames %d", n_frames);
  int rc = dust_check(h, 0, prm);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(h->cfg.device));
  hipStream_t s = stream ? reinterpret_cast<hipStream_t>(stream) : h->stream;
  const float *d_dust = reinterpret_cast<const float *>(reinterpret_cast<const uint8_t *>(d_records) + h->rl.off_dd);
  return dust_launch(h, d_dust, reinterpret_cast<const float *>(d_points_xyz), 0, reinterpret_cast<const float *>(d_Tcw), prm,
                     reinterpret_cast<uint8_t *>(d_out), s, n_frames, h->rl.bytes, reinterpret_cast<const int *>(d_n_points));
}

int spfe_align_dust(spfe_handle h, const float *dense_dust, const float *points_xyz, int n, const float *Tcw,
                    const spfe_dust_params *prm, float *Tcw_out, uint8_t *inlier, float *proj_uv, int *n_inlier,
                    int *iterations) {
  if (!h || !dense_dust || !Tcw || !prm || !Tcw_out || (n > 0 && !points_xyz)) return fail(SPFE_EINVAL, "null argument");
  int rc = dust_check(h, n, prm);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(h->cfg.device));
  const size_t map_b = (size_t)h->C * 4, pts_b = (size_t)SPFE_DUST_MAX_POINTS * 12, out_off = map_b + pts_b + 64;
  if (!h->dust_scratch) {
    if ((rc = dev_alloc(h, &h->dust_scratch, out_off + SPFE_DUST_OUT_BYTES))) return rc;
    if ((rc = host_alloc(h, &h->dust_host, (size_t)SPFE_DUST_OUT_BYTES))) return rc;
  }
  hipStream_t s = h->stream;
  uint8_t *d = h->dust_scratch;
  HIP_TRY(hipMemcpyAsync(d, dense_dust, map_b, hipMemcpyHostToDevice, s));
  if (n > 0) HIP_TRY(hipMemcpyAsync(d + map_b, points_xyz, (size_t)n * 12, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d + map_b + pts_b, Tcw, 64, hipMemcpyHostToDevice, s));
  rc = dust_launch(h, reinterpret_cast<const float *>(d), reinterpret_cast<const float *>(d + map_b), n,
                   reinterpret_cast<const float *>(d + map_b + pts_b), prm, d + out_off, s);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(h->dust_host, d + out_off, SPFE_DUST_OUT_BYTES, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  memcpy(Tcw_out, h->dust_host, 64);
  const int *cnt = reinterpret_cast<const int *>(h->dust_host + 64);
  if (n_inlier) *n_inlier = cnt[0];
  if (iterations) *iterations = cnt[1];
  if (proj_uv && n > 0) memcpy(proj_uv, h->dust_host + SPFE_DUST_OFF_UV, (size_t)n * 8);
  if (inlier && n > 0) memcpy(inlier, h->dust_host + SPFE_DUST_OFF_INLIER, (size_t)n);
  return SPFE_OK;
}

// ---- pipelined host path ------------------------------------------------------------------------
// The host boundary of SPExtractor::operator() (upload sp_extractor.cpp:379-390, six synchronous D2H copies
// :427-433) as a depth-NPIPE pipeline: pinned staging, H2D of batch i + 1 and D2H of batch i - 1 on copy
// streams beside the compute of batch i, covariance on the side stream.
namespace {
int pipe_setup(spfe_handle h) {
  if (h->pipe_ready) return SPFE_OK;
  const size_t img = (size_t)h->B * h->H * h->W, rec = (size_t)h->B * h->rl.bytes;
  const bool want = (h->cfg.flags & SPFE_FLAG_HEAT) != 0;
  int rc;
  HIP_TRY(hipStreamCreateWithFlags(&h->s_h2d, hipStreamNonBlocking));
  HIP_TRY(hipStreamCreateWithFlags(&h->s_d2h, hipStreamNonBlocking));
  for (auto &ps : h->pipe) {
    if ((rc = host_alloc(h, &ps.h_img, img))) return rc;
    if ((rc = host_alloc(h, &ps.h_rec, rec))) return rc;
    if ((rc = dev_alloc(h, &ps.d_img, img))) return rc;
    if ((rc = dev_alloc(h, &ps.d_rec, rec))) return rc;
    if (want) {
      if ((rc = host_alloc(h, &ps.h_heat, img))) return rc;
      if ((rc = host_alloc(h, &ps.h_heat_inv, img))) return rc;
    }
    HIP_TRY(hipEventCreateWithFlags(&ps.ev_h2d, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&ps.ev_done, hipEventDisableTiming));
  }
  h->pipe_ready = true;
  return SPFE_OK;
}
}  // namespace

int spfe_submit_batch(spfe_handle h, const uint8_t *const *images, int stride, int n, long *ticket) {
  if (!h || !ticket) return fail(SPFE_EINVAL, "null argument");
  if (!images) return fail(SPFE_EEMPTY, "input image is empty");
  if (n < 1 || n > h->B) return fail(SPFE_EINVAL, "batch %d not in [1, %d]", n, h->B);
  const int H = h->H, W = h->W;
  if (stride < W) return fail(SPFE_EINVAL, "stride %d smaller than width %d", stride, W);
  HIP_TRY(hipSetDevice(h->cfg.device));
  int rc = pipe_setup(h);
  if (rc) return rc;
  spfe_handle_s::PipeSlot &ps = h->pipe[h->pipe_submitted % spfe_handle_s::NPIPE];
  if (ps.ticket >= 0)
    return fail(SPFE_EINVAL, "pipeline full: %d batches in flight, collect ticket %ld first", spfe_handle_s::NPIPE, ps.ticket);
  for (int i = 0; i < n; ++i) {
    if (!images[i]) return fail(SPFE_EEMPTY, "input image is empty");  // sp_extractor.cpp:364-365
    if (stride == W) memcpy(ps.h_img + (size_t)i * H * W, images[i], (size_t)H * W);
    else
      for (int y = 0; y < H; ++y) memcpy(ps.h_img + ((size_t)i * H + y) * W, images[i] + (size_t)y * stride, W);
  }
  HIP_TRY(hipMemcpyAsync(ps.d_img, ps.h_img, (size_t)n * H * W, hipMemcpyHostToDevice, h->s_h2d));
  HIP_TRY(hipEventRecord(ps.ev_h2d, h->s_h2d));
  hipStream_t s = h->stream;
  HIP_TRY(hipStreamWaitEvent(s, ps.ev_h2d, 0));
  const bool want = (h->cfg.flags & SPFE_FLAG_HEAT) != 0;
  if (want && h->pipe_submitted > 0) {
    // the heat maps are single buffers: this batch's heat_norm (side stream) must not overwrite them
    // before the previous batch's copy has left
    const spfe_handle_s::PipeSlot &pp = h->pipe[(h->pipe_submitted - 1) % spfe_handle_s::NPIPE];
    if (pp.ticket >= 0) HIP_TRY(hipStreamWaitEvent(h->side, pp.ev_done, 0));
  }
  h->pipe_mode = true;
  rc = enqueue(h, ps.d_img, n, ps.d_rec, s);
  h->pipe_mode = false;
  if (rc) return rc;
  const long t = h->ticket - 1;
  // D2H on the SIDE stream, behind the covariance kernels it has to follow anyway.  (A copy stream of its own, waiting
  // for the covariance event, looked cleaner and cost half the throughput in bf16 mode: HIP maps streams onto a few
  // hardware queues, the waiting copy stream shared one with the compute stream, and its barrier packet held the NEXT
  // batch's convolutions until the previous batch's covariance had finished — tools/microbench/run_hosttrace.sh.)
  hipStream_t sc = h->side;
  {
    // f32 mode: a copy kernel of our own (2038 against 2000 ... 2028 frames/s with the runtime's copy at 752x480 x 8).  bf16
    // mode: the runtime's copy engine — the kernel's 64 workgroups sit on the chip for the 0.2 ms the PCIe transfer takes,
    // beside convolutions that are 4x shorter than the f32 ones: 6895 against 7990 frames/s at 1280x720 x 8 (= the
    // device-resident rate)
    static const int copy_env = getenv("SPFE_PIPE_COPY_KERNEL") ? atoi(getenv("SPFE_PIPE_COPY_KERNEL")) : -1;
    const int copy_mode = copy_env >= 0 ? copy_env : (h->bf16 && h->C >= 10000 ? 0 : 1);   // (bf16 752x480: kernel 11,560, engine 11,250)
    if (copy_mode == 1) {          // a copy kernel of our own writing the pinned buffer
      const size_t n16 = ((size_t)n * h->rl.bytes + 15) / 16;
      hipLaunchKernelGGL(spfe::copy_records_kernel, dim3(64), dim3(256), 0, sc, reinterpret_cast<uint4 *>(ps.h_rec),
                         reinterpret_cast<const uint4 *>(ps.d_rec), n16);
      HIP_TRY(hipGetLastError());
    } else if (copy_mode != 2) {   // (2: no copy at all, timing probe)
      HIP_TRY(hipMemcpyAsync(ps.h_rec, ps.d_rec, (size_t)n * h->rl.bytes, hipMemcpyDeviceToHost, sc));
    }
  }
  if (want) {
    const size_t m16 = (size_t)n * H * W * 4 / 16;   // (H, W multiples of 8)
    hipLaunchKernelGGL(spfe::copy_records_kernel, dim3(64), dim3(256), 0, sc, reinterpret_cast<uint4 *>(ps.h_heat_inv),
                       reinterpret_cast<const uint4 *>(h->d_heat_inv), m16);
    hipLaunchKernelGGL(spfe::copy_records_kernel, dim3(64), dim3(256), 0, sc, reinterpret_cast<uint4 *>(ps.h_heat),
                       reinterpret_cast<const uint4 *>(h->d_heat), m16);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipEventRecord(ps.ev_done, sc));
  ps.ticket = t;
  ps.n = n;
  h->pipe_submitted++;
  *ticket = t;
  return SPFE_OK;
}

int spfe_collect_batch(spfe_handle h, long ticket, spfe_result *outs) {
  if (!h || !outs) return fail(SPFE_EINVAL, "null argument");
  spfe_handle_s::PipeSlot *ps = nullptr;
  for (auto &c : h->pipe)
    if (c.ticket == ticket && ticket >= 0) ps = &c;
  if (!ps) return fail(SPFE_EINVAL, "ticket %ld is not in flight", ticket);
  HIP_TRY(hipSetDevice(h->cfg.device));
  HIP_TRY(hipEventSynchronize(ps->ev_done));
  const int H = h->H, W = h->W;
  const bool want = (h->cfg.flags & SPFE_FLAG_HEAT) != 0;
  for (int i = 0; i < ps->n; ++i) {
    uint8_t *rec = ps->h_rec + (size_t)i * h->rl.bytes;
    view_record(h, rec, want ? ps->h_heat + (size_t)i * H * W : nullptr, want ? ps->h_heat_inv + (size_t)i * H * W : nullptr,
                &outs[i]);
  }
  ps->ticket = -1;   // the views stay valid until NPIPE further submits reuse the slot
  return SPFE_OK;
}

// ---- multi-GPU: RCCL all-gather of the records -----------------------------------
namespace {
void *open_rccl() {
  // an already loaded librccl (e.g. the one torch.distributed brought) is reused by soname; the handle is kept for the
  // life of the process (one dlopen, never closed: communicators may outlive any one extractor handle)
  static void *const lib = []() -> void * {   // (function-local static: initialised once, also under concurrent first calls)
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
      if (void *l = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) return l;
    return nullptr;
  }();
  return lib;
}
}  // namespace

int spfe_comm_unique_id(void *id, size_t cap) {
  if (!id || cap < NCCL_UNIQUE_ID_BYTES) return fail(SPFE_EINVAL, "unique id buffer must hold %d bytes", NCCL_UNIQUE_ID_BYTES);
  void *lib = open_rccl();
  if (!lib) return fail(SPFE_EHIP, "librccl not found: %s", dlerror());
  auto get = reinterpret_cast<pfn_ncclGetUniqueId>(dlsym(lib, "ncclGetUniqueId"));
  auto err = reinterpret_cast<pfn_ncclGetErrorString>(dlsym(lib, "ncclGetErrorString"));
  if (!get || !err) return fail(SPFE_EHIP, "librccl lacks ncclGetUniqueId");
  ncclUniqueId u;
  const ncclResult_t r = get(&u);
  if (r != ncclSuccess) return fail(SPFE_EHIP, "ncclGetUniqueId: %s", err(r));
  memcpy(id, &u, NCCL_UNIQUE_ID_BYTES);
  return SPFE_OK;
}

int spfe_comm_init(spfe_handle h, const void *id, int rank, int world) {
  if (!h || !id) return fail(SPFE_EINVAL, "null argument");
  if (world < 1 || rank < 0 || rank >= world) return fail(SPFE_EINVAL, "rank %d / world %d", rank, world);
  if (h->comm) return fail(SPFE_EINVAL, "communicator already initialised (spfe_comm_destroy first)");
  HIP_TRY(hipSetDevice(h->cfg.device));
  if (!h->rccl_lib) {
    h->rccl_lib = open_rccl();
    if (!h->rccl_lib) return fail(SPFE_EHIP, "librccl not found: %s", dlerror());
    h->p_ncclCommInitRank = reinterpret_cast<pfn_ncclCommInitRank>(dlsym(h->rccl_lib, "ncclCommInitRank"));
    h->p_ncclCommDestroy = reinterpret_cast<pfn_ncclCommDestroy>(dlsym(h->rccl_lib, "ncclCommDestroy"));
    h->p_ncclCommCount = reinterpret_cast<pfn_ncclCommCount>(dlsym(h->rccl_lib, "ncclCommCount"));
    h->p_ncclAllGather = reinterpret_cast<pfn_ncclAllGather>(dlsym(h->rccl_lib, "ncclAllGather"));
    h->p_ncclGetErrorString = reinterpret_cast<pfn_ncclGetErrorString>(dlsym(h->rccl_lib, "ncclGetErrorString"));
    if (!h->p_ncclCommInitRank || !h->p_ncclCommDestroy || !h->p_ncclCommCount || !h->p_ncclAllGather || !h->p_ncclGetErrorString)
      return fail(SPFE_EHIP, "librccl lacks a required entry point");
  }
  ncclUniqueId u;
  memcpy(&u, id, NCCL_UNIQUE_ID_BYTES);
  const ncclResult_t r = h->p_ncclCommInitRank(&h->comm, world, u, rank);
  if (r != ncclSuccess) {
    h->comm = nullptr;
    return fail(SPFE_EHIP, "ncclCommInitRank(rank %d of %d, device %d): %s", rank, world, h->cfg.device,
                h->p_ncclGetErrorString(r));
  }
  // The collective runs on the SIDE stream, behind the covariance kernels of the batch it gathers (call
  // spfe_allgather_records for batch i before enqueueing batch i + 1, as parallel.ShardedExtractor does): no stream sits in
  // a hardware queue waiting for the covariance event.  HIP maps streams onto a few hardware queues; a waiting stream that
  // lands on the compute stream's queue holds the NEXT batch's convolutions back (measured on the host path: half the
  // throughput).  SPFE_COMM_OWN_STREAM=1: a communication stream of its own that waits for the batch's event.
  h->comm_own_stream = getenv("SPFE_COMM_OWN_STREAM") && atoi(getenv("SPFE_COMM_OWN_STREAM")) != 0;
  if (!h->comm_stream) {
    if (h->comm_own_stream) HIP_TRY(hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
    else h->comm_stream = h->side;
  }
  if (!h->ev_gather) HIP_TRY(hipEventCreateWithFlags(&h->ev_gather, hipEventDisableTiming));
  h->comm_rank = rank;
  h->comm_world = world;
  h->gather_recorded = false;
  return SPFE_OK;
}

int spfe_comm_destroy(spfe_handle h) {
  if (!h) return fail(SPFE_EINVAL, "null handle");
  if (h->comm_stream) (void)hipStreamSynchronize(h->comm_stream);
  if (h->comm && h->p_ncclCommDestroy) (void)h->p_ncclCommDestroy(h->comm);
  h->comm = nullptr;
  if (h->ev_gather) { (void)hipEventDestroy(h->ev_gather); h->ev_gather = nullptr; }
  if (h->comm_stream && h->comm_own_stream) (void)hipStreamDestroy(h->comm_stream);
  h->comm_stream = nullptr;
  h->comm_world = 0;
  h->gather_recorded = false;
  return SPFE_OK;
}

void *spfe_comm_stream(spfe_handle h) { return h ? reinterpret_cast<void *>(h->comm_stream) : nullptr; }

int spfe_comm_count(spfe_handle h, int *count) {
  if (!h || !count) return fail(SPFE_EINVAL, "null argument");
  if (!h->comm) return fail(SPFE_EINVAL, "spfe_comm_init has not been called");
  const ncclResult_t r = h->p_ncclCommCount(h->comm, count);   // what RCCL itself says, not what the caller passed in
  if (r != ncclSuccess) return fail(SPFE_EHIP, "ncclCommCount: %s", h->p_ncclGetErrorString(r));
  return SPFE_OK;
}

int spfe_allgather_records(spfe_handle h, long ticket, const void *d_local, void *d_all, int frames_per_rank) {
  if (!h || !d_local || !d_all) return fail(SPFE_EINVAL, "null argument");
  if (!h->comm) return fail(SPFE_EINVAL, "spfe_comm_init has not been called");
  if (frames_per_rank < 1) return fail(SPFE_EINVAL, "frames_per_rank %d", frames_per_rank);
  if (ticket < 0 || ticket >= h->ticket || ticket + spfe_handle_s::NTICKET <= h->ticket)
    return fail(SPFE_EINVAL, "ticket %ld is not one of the last %d calls", ticket, spfe_handle_s::NTICKET);
  HIP_TRY(hipSetDevice(h->cfg.device));
  // on the side stream the gather simply follows the batch's covariance kernels (and everything enqueued there since:
  // gather batch i before enqueueing batch i + 1); a stream of its own waits for exactly this batch's records.  Either
  // way the gather of batch i runs beside the convolutions of batch i + 1
  // (always: in pipelined calls the covariance kernels sit on the side stream in front of the gather and the event has been
  // recorded there — a wait that is satisfied when it is reached; in synchronous calls the chain runs on the launch stream
  // (round 4) and this wait is what orders the gather behind it)
  HIP_TRY(hipStreamWaitEvent(h->comm_stream, h->ev_cov[ticket % spfe_handle_s::NTICKET], 0));
  const size_t count = (size_t)frames_per_rank * h->rl.bytes;   // bytes as ncclUint8; RCCL counts are size_t
  const ncclResult_t r = h->p_ncclAllGather(d_local, d_all, count, ncclUint8, h->comm, h->comm_stream);
  if (r != ncclSuccess) return fail(SPFE_EHIP, "ncclAllGather(%zu bytes per rank): %s", count, h->p_ncclGetErrorString(r));
  HIP_TRY(hipEventRecord(h->ev_gather, h->comm_stream));
  h->gather_recorded = true;
  return SPFE_OK;
}

int spfe_comm_wait(spfe_handle h, void *stream) {
  if (!h) return fail(SPFE_EINVAL, "null handle");
  if (!h->gather_recorded) return SPFE_OK;
  HIP_TRY(hipSetDevice(h->cfg.device));
  hipStream_t s = stream ? reinterpret_cast<hipStream_t>(stream) : h->stream;
  HIP_TRY(hipStreamWaitEvent(s, h->ev_gather, 0));
  return SPFE_OK;
}

// ---- input staging ------------------------------------------------------
int spfe_set_staging(spfe_handle h, const spfe_staging *st) {
  if (!h || !st) return fail(SPFE_EINVAL, "null argument");
  if (st->channels != 1 && st->channels != 3 && st->channels != 4)
    return fail(SPFE_EINVAL, "staging: %d channels unsupported (1, 3, 4)", st->channels);
  if (st->src_height < h->H || st->src_width < h->W)
    return fail(SPFE_EINVAL, "staging: source %dx%d smaller than the extractor's %dx%d (system.cpp:160 crop)",
                st->src_width, st->src_height, h->W, h->H);
  if (st->src_height > 32767 || st->src_width > 32767) return fail(SPFE_EINVAL, "staging: source too large");
  if ((st->map_x == nullptr) != (st->map_y == nullptr)) return fail(SPFE_EINVAL, "staging: one map is null");
  HIP_TRY(hipSetDevice(h->cfg.device));
  HIP_TRY(hipDeviceSynchronize());
  for (void **p : {(void **)&h->d_map_x, (void **)&h->d_map_y, (void **)&h->d_raw})
    if (*p) { (void)hipFree(*p); *p = nullptr; }
  if (h->h_raw) { (void)hipHostFree(h->h_raw); h->h_raw = nullptr; }
  h->st_set = false;
  const size_t npx = (size_t)st->src_height * st->src_width;
  if (st->map_x) {
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->d_map_x), npx * 4));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->d_map_y), npx * 4));
    HIP_TRY(hipMemcpy(h->d_map_x, st->map_x, npx * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->d_map_y, st->map_y, npx * 4, hipMemcpyHostToDevice));
  }
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->d_raw), (size_t)h->B * npx * st->channels));
  HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&h->h_raw), (size_t)h->B * npx * st->channels,
                        hipHostMallocDefault));
  h->st = *st;
  h->st.map_x = h->st.map_y = nullptr;  // the caller's arrays are not kept
  h->st_set = true;
  return SPFE_OK;
}

namespace {
int enqueue_stage(spfe_handle h, const uint8_t *d_src, int n, uint8_t *d_gray, hipStream_t s) {
  spfe::StageParams p{};
  p.src = d_src;
  p.src_stride = h->st.src_width * h->st.channels;
  p.src_frame_bytes = (size_t)h->st.src_height * p.src_stride;
  p.src_h = h->st.src_height;
  p.src_w = h->st.src_width;
  p.map_x = h->d_map_x;
  p.map_y = h->d_map_y;
  p.rgb = h->st.rgb;
  p.gray = d_gray;
  p.H = h->H;
  p.W = h->W;
  HIP_TRY(spfe::launch_stage_input(p, h->st.channels, n, s));
  return SPFE_OK;
}
}  // namespace

int spfe_stage_batch_device(spfe_handle h, const void *d_src, int n, void *d_gray, void *stream) {
  if (!h || !d_gray) return fail(SPFE_EINVAL, "null argument");
  if (!h->st_set) return fail(SPFE_EINVAL, "spfe_set_staging has not been called");
  if (!d_src) return fail(SPFE_EEMPTY, "input image is empty");
  if (n < 1 || n > h->B) return fail(SPFE_EINVAL, "batch %d not in [1, %d]", n, h->B);
  HIP_TRY(hipSetDevice(h->cfg.device));
  hipStream_t s = stream ? reinterpret_cast<hipStream_t>(stream) : h->stream;
  return enqueue_stage(h, reinterpret_cast<const uint8_t *>(d_src), n, reinterpret_cast<uint8_t *>(d_gray), s);
}

int spfe_extract_batch_staged(spfe_handle h, const uint8_t *const *srcs, int stride, int n, spfe_result *outs) {
  if (!h || !outs) return fail(SPFE_EINVAL, "null argument");
  if (!h->st_set) return fail(SPFE_EINVAL, "spfe_set_staging has not been called");
  if (!srcs) return fail(SPFE_EEMPTY, "input image is empty");
  if (n < 1 || n > h->B) return fail(SPFE_EINVAL, "batch %d not in [1, %d]", n, h->B);
  const int row = h->st.src_width * h->st.channels;
  if (stride < row) return fail(SPFE_EINVAL, "stride %d smaller than a source row (%d bytes)", stride, row);
  const size_t frame = (size_t)h->st.src_height * row;
  for (int i = 0; i < n; ++i) {
    if (!srcs[i]) return fail(SPFE_EEMPTY, "input image is empty");  // sp_extractor.cpp:364-365
    for (int y = 0; y < h->st.src_height; ++y)
      memcpy(h->h_raw + i * frame + (size_t)y * row, srcs[i] + (size_t)y * stride, row);
  }
  HIP_TRY(hipSetDevice(h->cfg.device));
  hipStream_t s = h->stream;
  HIP_TRY(hipMemcpyAsync(h->d_raw, h->h_raw, n * frame, hipMemcpyHostToDevice, s));
  int rc = enqueue_stage(h, h->d_raw, n, h->d_img, s);
  if (rc) return rc;
  rc = enqueue(h, h->d_img, n, h->d_records, s);
  if (rc) return rc;
  return finish_host(h, n, outs);
}

int spfe_extract_staged(spfe_handle h, const uint8_t *src, int stride, spfe_result *out) {
  if (!src) return fail(SPFE_EEMPTY, "input image is empty");
  const uint8_t *one[1] = {src};
  return spfe_extract_batch_staged(h, one, stride, 1, out);
}

// ---- patch-wise association (tracker_dust.cpp:113-172) -------------------------------------------
namespace {
constexpr int kPatchMax = 4096;
int patch_scratch(spfe_handle h) {
  if (h->p_cidx) return SPFE_OK;
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->p_cidx), (size_t)kPatchMax * 4 * sizeof(int)));
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->p_cdist), (size_t)kPatchMax * 4 * sizeof(float)));
  return SPFE_OK;
}
}  // namespace

int spfe_match_patches_record_device(spfe_handle h, const void *d_mp_desc, const void *d_mp_uv, int n_points,
                                     const void *d_record, float max_dist, void *d_kp_idx, void *stream) {
  if (!h || !d_record || !d_kp_idx) return fail(SPFE_EINVAL, "null argument");
  if (n_points < 0 || n_points > kPatchMax) return fail(SPFE_EINVAL, "n_points %d not in [0, %d]", n_points, kPatchMax);
  if (n_points == 0) return SPFE_OK;
  if (!d_mp_desc || !d_mp_uv) return fail(SPFE_EINVAL, "null argument");
  HIP_TRY(hipSetDevice(h->cfg.device));
  int rc = patch_scratch(h);
  if (rc) return rc;
  hipStream_t s = stream ? reinterpret_cast<hipStream_t>(stream) : h->stream;
  const uint8_t *rec = reinterpret_cast<const uint8_t *>(d_record);
  spfe::PatchArgs a{};
  a.mp_desc = reinterpret_cast<const float *>(d_mp_desc);
  a.mp_uv = reinterpret_cast<const float *>(d_mp_uv);
  a.n_points = n_points;
  a.occ = reinterpret_cast<const int16_t *>(rec + h->rl.off_occ);
  a.hc = h->hc; a.wc = h->wc;
  a.kp_desc = reinterpret_cast<const float *>(rec + h->rl.off_desc);
  a.kp_desc_bf16 = h->rl.desc_bf16;
  a.k_ptr = reinterpret_cast<const int *>(rec + h->rl.off_hdr);
  a.k_imm = 0;
  HIP_TRY(spfe::launch_match_patches(a, h->kmax, max_dist, h->p_cidx, h->p_cdist,
                                     reinterpret_cast<int32_t *>(d_kp_idx), s));
  return SPFE_OK;
}

int spfe_track_dust_record_device(spfe_handle h, const void *d_record, const void *d_points_xyz, const void *d_mp_desc, int n,
                                  const void *d_Tcw, const spfe_dust_params *prm, int min_inliers, float max_dist,
                                  void *d_dust_out, void *d_kp_idx, void *stream) {
  if (!h || !d_record || !d_Tcw || !prm || !d_dust_out || !d_kp_idx || (n > 0 && (!d_points_xyz || !d_mp_desc)))
    return fail(SPFE_EINVAL, "null argument");
  int rc = dust_check(h, n, prm);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(h->cfg.device));
  if ((rc = patch_scratch(h))) return rc;
  hipStream_t s = stream ? reinterpret_cast<hipStream_t>(stream) : h->stream;
  const uint8_t *rec = reinterpret_cast<const uint8_t *>(d_record);
  uint8_t *dout = reinterpret_cast<uint8_t *>(d_dust_out);
  // PoseOptimizationDust(&mCurrentFrame, mps_for_track, is_visible)   tracker_dust.cpp:92-94
  rc = dust_launch(h, reinterpret_cast<const float *>(rec + h->rl.off_dd), reinterpret_cast<const float *>(d_points_xyz), n,
                   reinterpret_cast<const float *>(d_Tcw), prm, dout, s);
  if (rc || n == 0) return rc;
  // the patch-wise association of the in_view points at their dust_proj_u / v   :113-172, on the same stream: the
  // projections, the flags and n_inlier are read where the alignment left them
  spfe::PatchArgs a{};
  a.mp_desc = reinterpret_cast<const float *>(d_mp_desc);
  a.mp_uv = reinterpret_cast<const float *>(dout + SPFE_DUST_OFF_UV);
  a.n_points = n;
  a.occ = reinterpret_cast<const int16_t *>(rec + h->rl.off_occ);
  a.hc = h->hc; a.wc = h->wc;
  a.kp_desc = reinterpret_cast<const float *>(rec + h->rl.off_desc);
  a.kp_desc_bf16 = h->rl.desc_bf16;
  a.k_ptr = reinterpret_cast<const int *>(rec + h->rl.off_hdr);
  a.k_imm = 0;
  a.in_view = dout + SPFE_DUST_OFF_INLIER;
  a.gate_ptr = reinterpret_cast<const int *>(dout + 64);
  a.gate_min = min_inliers;
  HIP_TRY(spfe::launch_match_patches(a, h->kmax, max_dist, h->p_cidx, h->p_cdist, reinterpret_cast<int32_t *>(d_kp_idx), s));
  return SPFE_OK;
}

int spfe_match_patches(spfe_handle h, const float *mp_desc, const float *mp_uv, int n_points,
                       const int16_t *occ_grid, const float *kp_desc, int n_keypoints, float max_dist,
                       int32_t *kp_idx) {
  if (!h || !kp_idx) return fail(SPFE_EINVAL, "null argument");
  if (n_points < 0 || n_points > kPatchMax) return fail(SPFE_EINVAL, "n_points %d not in [0, %d]", n_points, kPatchMax);
  if (n_keypoints < 0 || n_keypoints > 32767) return fail(SPFE_EINVAL, "n_keypoints %d out of range", n_keypoints);
  for (int i = 0; i < n_points; ++i) kp_idx[i] = -1;
  if (n_points == 0 || n_keypoints == 0) return SPFE_OK;
  if (!mp_desc || !mp_uv || !occ_grid || !kp_desc) return fail(SPFE_EINVAL, "null argument");
  HIP_TRY(hipSetDevice(h->cfg.device));
  int rc = patch_scratch(h);
  if (rc) return rc;
  const size_t cells = (size_t)h->hc * h->wc;
  const size_t o_mp = 0, o_uv = o_mp + (size_t)n_points * 1024, o_occ = align_up(o_uv + (size_t)n_points * 8, 16),
               o_kp = align_up(o_occ + cells * 2, 16), o_out = o_kp + (size_t)n_keypoints * 1024,
               total = o_out + (size_t)n_points * 4;
  if (total > h->p_stage_bytes) {
    HIP_TRY(hipDeviceSynchronize());
    if (h->p_stage) (void)hipFree(h->p_stage);
    h->p_stage = nullptr;
    h->p_stage_bytes = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->p_stage), total));
    h->p_stage_bytes = total;
  }
  hipStream_t s = h->stream;
  uint8_t *d = h->p_stage;
  HIP_TRY(hipMemcpyAsync(d + o_mp, mp_desc, (size_t)n_points * 1024, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d + o_uv, mp_uv, (size_t)n_points * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d + o_occ, occ_grid, cells * 2, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(d + o_kp, kp_desc, (size_t)n_keypoints * 1024, hipMemcpyHostToDevice, s));
  spfe::PatchArgs a{};
  a.mp_desc = reinterpret_cast<const float *>(d + o_mp);
  a.mp_uv = reinterpret_cast<const float *>(d + o_uv);
  a.n_points = n_points;
  a.occ = reinterpret_cast<const int16_t *>(d + o_occ);
  a.hc = h->hc; a.wc = h->wc;
  a.kp_desc = reinterpret_cast<const float *>(d + o_kp);
  a.k_ptr = nullptr;
  a.k_imm = n_keypoints;
  HIP_TRY(spfe::launch_match_patches(a, n_keypoints, max_dist, h->p_cidx, h->p_cdist,
                                     reinterpret_cast<int32_t *>(d + o_out), s));
  HIP_TRY(hipMemcpyAsync(kp_idx, d + o_out, (size_t)n_points * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return SPFE_OK;
}

// ---- descriptor matching ------------------------------------------------
namespace {
int match_scratch(spfe_handle h, int pairs, int cap) {
  if (pairs <= h->m_pairs && cap <= h->m_cap) return SPFE_OK;
  pairs = std::max(pairs, h->m_pairs);
  cap = std::max(cap, h->m_cap);
  HIP_TRY(hipDeviceSynchronize());
  if (h->m_best_t) (void)hipFree(h->m_best_t);
  if (h->m_best_q) (void)hipFree(h->m_best_q);
  h->m_best_t = h->m_best_q = nullptr;
  h->m_pairs = h->m_cap = 0;
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->m_best_t), (size_t)pairs * cap * 8));
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->m_best_q), (size_t)pairs * cap * 8));
  h->m_pairs = pairs;
  h->m_cap = cap;
  return SPFE_OK;
}
constexpr size_t kMatchHdr = 16;  // staging block of the host API: int32 count, pad, then rows
}  // namespace

size_t spfe_match_out_bytes(spfe_handle h) { return h ? (size_t)h->kmax * 8 : 0; }

int spfe_match_records_device(spfe_handle h, const void *d_query_records, const void *d_train_records, int n_pairs,
                              int cross_check, void *d_out, void *stream) {
  if (!h || !d_query_records || !d_train_records || !d_out) return fail(SPFE_EINVAL, "null argument");
  if (n_pairs < 1) return fail(SPFE_EINVAL, "n_pairs %d must be >= 1", n_pairs);
  HIP_TRY(hipSetDevice(h->cfg.device));
  int rc = match_scratch(h, n_pairs, h->kmax);
  if (rc) return rc;
  hipStream_t s = stream ? reinterpret_cast<hipStream_t>(stream) : h->stream;
  spfe::MatchSide q{reinterpret_cast<const uint8_t *>(d_query_records), h->rl.bytes, h->rl.off_hdr, h->rl.off_desc,
                    h->kmax};
  spfe::MatchSide t{reinterpret_cast<const uint8_t *>(d_train_records), h->rl.bytes, h->rl.off_hdr, h->rl.off_desc,
                    h->kmax};
  q.desc_bf16 = t.desc_bf16 = h->rl.desc_bf16;   // (records made with SPFE_FLAG_DESC_BF16: bf16 rows, widened on load)
  HIP_TRY(spfe::launch_match(q, t, n_pairs, cross_check != 0, h->m_best_t, h->m_best_q,
                             reinterpret_cast<uint8_t *>(d_out), (size_t)h->kmax * 8, s));
  return SPFE_OK;
}

int spfe_match(spfe_handle h, const float *query, int n_query, const float *train, int n_train, int cross_check,
               int32_t *train_idx, float *distance) {
  if (!h || !train_idx || !distance) return fail(SPFE_EINVAL, "null argument");
  if (n_query < 0 || n_train < 0) return fail(SPFE_EINVAL, "negative descriptor count");
  if ((n_query && !query) || (n_train && !train)) return fail(SPFE_EINVAL, "null descriptor array");
  for (int i = 0; i < n_query; ++i) { train_idx[i] = -1; distance[i] = FLT_MAX; }
  if (n_query == 0 || n_train == 0) return SPFE_OK;
  HIP_TRY(hipSetDevice(h->cfg.device));
  const int cap = std::max(n_query, n_train);
  if (cap > h->m_host_cap) {
    HIP_TRY(hipDeviceSynchronize());
    for (uint8_t **p : {&h->m_stage_q, &h->m_stage_t, &h->m_out, &h->m_out2})
      if (*p) { (void)hipFree(*p); *p = nullptr; }
    h->m_host_cap = 0;
    const int want = std::max(cap, h->kmax);
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->m_stage_q), kMatchHdr + (size_t)want * 1024));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->m_stage_t), kMatchHdr + (size_t)want * 1024));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->m_out), (size_t)want * 8));
    h->m_host_cap = want;
  }
  int rc = match_scratch(h, 1, std::max(cap, h->kmax));
  if (rc) return rc;
  hipStream_t s = h->stream;
  const int32_t hq[4] = {n_query, 0, 0, 0}, ht[4] = {n_train, 0, 0, 0};
  HIP_TRY(hipMemcpyAsync(h->m_stage_q, hq, 16, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(h->m_stage_t, ht, 16, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(h->m_stage_q + kMatchHdr, query, (size_t)n_query * 1024, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(h->m_stage_t + kMatchHdr, train, (size_t)n_train * 1024, hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));  // hq / ht live on this frame
  spfe::MatchSide q{h->m_stage_q, 0, 0, kMatchHdr, n_query};
  spfe::MatchSide t{h->m_stage_t, 0, 0, kMatchHdr, n_train};
  HIP_TRY(spfe::launch_match(q, t, 1, cross_check != 0, h->m_best_t, h->m_best_q, h->m_out, 0, s));
  HIP_TRY(hipMemcpyAsync(train_idx, h->m_out, (size_t)n_query * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(distance, h->m_out + (size_t)n_query * 4, (size_t)n_query * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return SPFE_OK;
}

// knnMatch(query, matches, 2): the two nearest train rows of every query, exactly (the FLANN kd-tree the
// reference builds for this is approximate and randomised)
int spfe_match_knn2(spfe_handle h, const float *query, int n_query, const float *train, int n_train,
                    int32_t *train_idx, float *distance) {
  if (!h || !train_idx || !distance) return fail(SPFE_EINVAL, "null argument");
  if (n_query < 0 || n_train < 0) return fail(SPFE_EINVAL, "negative descriptor count");
  if ((n_query && !query) || (n_train && !train)) return fail(SPFE_EINVAL, "null descriptor array");
  for (int i = 0; i < 2 * n_query; ++i) { train_idx[i] = -1; distance[i] = FLT_MAX; }
  if (n_query == 0 || n_train == 0) return SPFE_OK;
  HIP_TRY(hipSetDevice(h->cfg.device));
  const int cap = std::max(n_query, n_train);
  if (cap > h->m_host_cap || !h->m_out2) {
    HIP_TRY(hipDeviceSynchronize());
    for (uint8_t **p : {&h->m_stage_q, &h->m_stage_t, &h->m_out, &h->m_out2})
      if (*p) { (void)hipFree(*p); *p = nullptr; }
    h->m_host_cap = 0;
    const int want = std::max(cap, h->kmax);
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->m_stage_q), kMatchHdr + (size_t)want * 1024));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->m_stage_t), kMatchHdr + (size_t)want * 1024));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->m_out), (size_t)want * 8));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&h->m_out2), (size_t)want * 16));
    h->m_host_cap = want;
  }
  int rc = match_scratch(h, 1, std::max(cap, h->kmax));
  if (rc) return rc;
  hipStream_t s = h->stream;
  const int32_t hq[4] = {n_query, 0, 0, 0}, ht[4] = {n_train, 0, 0, 0};
  HIP_TRY(hipMemcpyAsync(h->m_stage_q, hq, 16, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(h->m_stage_t, ht, 16, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(h->m_stage_q + kMatchHdr, query, (size_t)n_query * 1024, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(h->m_stage_t + kMatchHdr, train, (size_t)n_train * 1024, hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));  // hq / ht live on this frame
  spfe::MatchSide q{h->m_stage_q, 0, 0, kMatchHdr, n_query};
  spfe::MatchSide t{h->m_stage_t, 0, 0, kMatchHdr, n_train};
  // scratch: best_q holds the first neighbours, best_t (>= cap entries) the second
  HIP_TRY(spfe::launch_match_knn2(q, t, 1, h->m_best_q, h->m_best_t, h->m_out2, 0, s));
  // device layout idx1 | dist1 | idx2 | dist2 -> host layout [n_query][2]
  std::vector<int32_t> hi(2 * (size_t)n_query);
  std::vector<float> hd(2 * (size_t)n_query);
  HIP_TRY(hipMemcpyAsync(hi.data(), h->m_out2, (size_t)n_query * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(hd.data(), h->m_out2 + (size_t)n_query * 4, (size_t)n_query * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(hi.data() + n_query, h->m_out2 + (size_t)n_query * 8, (size_t)n_query * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(hd.data() + n_query, h->m_out2 + (size_t)n_query * 12, (size_t)n_query * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  for (int i = 0; i < n_query; ++i) {
    train_idx[2 * i] = hi[i]; train_idx[2 * i + 1] = hi[n_query + i];
    distance[2 * i] = hd[i]; distance[2 * i + 1] = hd[n_query + i];
  }
  return SPFE_OK;
}

// test hook: run the exact-math device functions on n floats (host buffers)
int spfe_math_probe(const float *in, float *out_exp, float *out_log, int n) {
  float *d_in = nullptr, *d_e = nullptr, *d_l = nullptr;
  HIP_TRY(hipMalloc(&d_in, n * 4));
  HIP_TRY(hipMalloc(&d_e, n * 4));
  HIP_TRY(hipMalloc(&d_l, n * 4));
  HIP_TRY(hipMemcpy(d_in, in, n * 4, hipMemcpyHostToDevice));
  HIP_TRY(spfe::launch_math_probe(d_in, d_e, d_l, n, nullptr));
  HIP_TRY(hipMemcpy(out_exp, d_e, n * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(out_log, d_l, n * 4, hipMemcpyDeviceToHost));
  (void)hipFree(d_in); (void)hipFree(d_e); (void)hipFree(d_l);
  return SPFE_OK;
}

}  // extern "C"
