// spfe_host.h — what the host-side translation units of libspfe.so share: the handle, the stage table, allocation and error
// helpers, and the entry points of one unit that another calls.
//   spfe_pack.hip      weight blob -> device tables, buffers, streams, events: build() (the handle's construction)
//   spfe_schedule.hip  the per-batch launch sequence: enqueue() / enqueue_post() (streams, events, ticket parity)
//   spfe_comm.hip      multi-GPU: RCCL all-gather of the records (spfe_comm_*, spfe_allgather_records)
//   spfe_widen.hip     the rows SURVEY.md §8f widens into: dust alignment, input staging, descriptor matching (C ABI)
//   spfe_api.hip       the C ABI of the path itself: create / destroy / extract* / submit + collect / debug reads / timing
// One handle = one GPU, one stream, one set of buffers (SURVEY.md §8b "Threading"): the object SPExtractor's constructor
// builds (/root/reference/orb_slam2/src/cv/sp_extractor.cpp:342-359) and whose operator() (:361-514) the extract calls replace.
#pragma once
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


namespace spfe_host {

extern thread_local std::string g_err;
int fail(int code, const char *fmt, ...);

#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess)                                                               \
      return ::spfe_host::fail(SPFE_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                  __LINE__);                                                            \
  } while (0)

constexpr int NSTAGE = 15;
extern const char *const kStageNames[NSTAGE];

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


}  // namespace spfe_host
using spfe_host::ConvLayer;
using spfe_host::NSTAGE;

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
  bool defer_db = true;          // synchronous dense calls: convDb launched behind the detector tail
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
  int split_mode = -1;      // SPFE_SPLIT: -1 = f32 always, bf16 frames of fewer than 10,000 cells (752x480: +2 %; 1280x720: +-0); 0 never; 1 always
  // the other schedule switches (read_switches(), spfe_pack.hip; README "Environment switches"), defaults = the product's order
  bool inline_chain = true, tail_per_half = true, early_waits = true, sel_ext_event = true, zero_in_tail = true;
  // TWO SIDE CHAINS IN FLIGHT (round 5).  The side chain of a batch is a serial string of latency-bound kernels on one stream,
  // and what it hands over internally (heat_inv, the covariance scratch, the cell lists, the coarse map) exists once — so
  // chain i + 1 starts when chain i has ended, and on large bf16 frames a chain (0.7 - 1.0 ms beside the convolutions of a
  // 1.0 ms step; its selection cannot even start before conv1b ends: LDS) is as long as the step: the side stream, not the
  // convolutions, then sets the step (1280x720 x 8: 7,530 - 8,290 frames/s depending on how many long replay chains the frames
  // hold).  With a second, complete set of buffers and a second side stream the chains of consecutive batches overlap: the
  // handle owns a TWIN — a whole second handle built from the same configuration — and pipelined device calls alternate
  // between the two (measured first with two handles in one process, tools/microbench/two_chains.py: 7,530 -> 8,115 and
  // 7,745 -> 8,125 frames/s at 1280x720 bf16; 752x480: -3 % bf16, -2 % f32: there the chain is short and a step runs as two
  // half batches).  Tickets stay one sequence: tmap says which of the two ran a ticket, and under which ticket of its own.
  // SPFE_TWO_CHAINS: -1 by workload (bf16 frames of >= 10,000 cells in short calls or beyond select_kernel: make_twin, spfe_api.hip), 0 never, 1 always.
  spfe_handle twin = nullptr;
  bool is_twin = false;
  bool twin_failed = false;                  // the twin could not be built (e.g. out of memory): one side chain, and no second attempt
  int two_chains_env = -1;
  // The parsed weight blob (register_module order), kept for as long as a twin may still be built from it: spfe_config's
  // weights / weights_path are the CALLER's memory and need not outlive spfe_create (the lazily built twin used to
  // read them again at the first spfe_submit_batch) — the stored cfg carries neither.
  std::vector<float> blob;
  long last_seq = 0;                         // value of the process-wide call counter at this handle's last enqueue_post (debug reads
                                             // of a pair come from the one that ran last)
  long g_ticket = 0;                         // tickets handed out by this handle when it has a twin
  struct TicketRef { spfe_handle who; long local; } tmap[8] = {};
  int replay_waves = 0;     // SPFE_REPLAY_WAVES: 0 = by workload
  int sparse_db_env = -1, sparse_da_env = -1, pbtail_env = 1, fuse1a_env = -1, pipe_copy_kernel = -1, cov_ecap_env = -1;
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
  int sparse_da_mode = 1;        // SPFE_SPARSE_DA: 0 never, 1 synchronous calls only, 2 pipelined calls too (the default of both precisions since round 5)
  bool sparse_da_call = false;   // ... this / the last call
  int *d_db_list = nullptr, *d_db_total = nullptr;
  int db_cap = 0;                // list entries per frame: min(4 kmax, C)
  int db_tiles_per_wg = 4;       // the gathered head's grid = listed tiles / this (a workgroup's weights: 128 KB)
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
  // (select_kernel's 2-bytes-a-cell LDS form is for frames above 16,384 cells only.  Forced onto smaller frames in pipelined
  // calls it starts beside a convolution workgroup instead of waiting for a free CU, and that measured as NOT a gain — f32
  // 752x480 2107 / 2116 -> 2085 / 2082 frames/s, conv1b 0.87 -> 0.83 of peak: HISTORY.md "Round 4" — the switch is gone)
  int *d_sel_slot = nullptr;          // frames of more than 16,384 cells: select_kernel's global scratch (tail_select.hip)
  uint16_t *d_sel_list = nullptr;
  // frames of more than 65,535 cells (3840x2160), or SPFE_SELECT_HUGE=1: select_huge_kernel, everything per cell in global scratch
  bool select_huge = false;
  int select_huge_env = 0;
  uint8_t *d_sel_state = nullptr;
  int *d_sel_list32 = nullptr;
  uint8_t *d_records = nullptr;
  spfe::CovScratch cov{};
  int cov_gen_code = 0;        // generation code of the last chain (CovScratch::gen = code << 16); 0: none yet
  int cov_gen_start = 32766;   // SPFE_COV_CAPS field 6 (tests reach the wrap)
  int cov_frames_clean = 0;    // leading frames whose claim / done maps hold tagged (or reset) entries
  int cov_captured_min = 0;    // lowest generation code a captured (hipGraph) call froze; 0: never captured (enqueue_post)
  ConvLayer layers[10];
  spfe::RecordLayout rl{};
  // host side
  uint8_t *h_img = nullptr, *h_records = nullptr;
  float *h_heat = nullptr, *h_heat_inv = nullptr;
  float *own_heat = nullptr, *own_heat_inv = nullptr;   // spfe_set_map_buffers: the library's buffers while h_heat / h_heat_inv
  float *usr_heat = nullptr, *usr_heat_inv = nullptr;   // point at the caller's (registered) memory
  int last_n = 0;
  int host_sync_n = 0;   // frames of the last synchronous host call (spfe_fetch_heat_inv)
  // Synchronous host calls with SPFE_FLAG_HEAT (the drop-in's operator(): Frame clones heat_, frame.cpp:304): the heat maps are
  // final when the heat normalisation ends — ~150 us before a single frame's record is (selection + covariance behind it) — so
  // their D2H (2 x 4 H W bytes: 2.9 MB at 752x480, ~100 us of PCIe) starts THERE, on a copy stream of its own behind the
  // normalisation's completion signal, and runs beside the chain instead of behind it (round 6; SPFE_EARLY_HEAT_COPY=0: behind
  // the record's copy, as before).
  hipStream_t s_heat = nullptr;
  hipEvent_t ev_heat = nullptr, ev_heat_copied1 = nullptr, ev_heat_copied = nullptr;   // normalisation done / heat in host memory / both maps
  bool early_heat_copy = true;   // SPFE_EARLY_HEAT_COPY
  int open_n = 0;                // frames of a call begun by spfe_extract_begin and not yet finished
  bool host_sync_call = false;   // set by the synchronous host entry points around enqueue()
  bool heat_early = false;       // this call's maps were sent ahead: finish_host waits for ev_heat_copied instead of copying
  // ... and the descriptor rows of the record (kmax x 256 floats: 1.0 of a 752x480 record's 1.1 MB) leave right behind the
  // descriptor sampling, on the side stream that ran it, beside the covariance replay (the chain's longest kernel) instead of
  // behind it; finish_host then copies the record's two small ends only.  Inline-chain calls (a synchronous call with the
  // gathered descriptor branch: the single-frame operator()); same switch.
  bool desc_early = false;
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
  unsigned tile2_mask = 0;   // SPFE_TILE2_MASK > 0: f32 layers forced onto 2-row tiles
  bool tile2_auto = true;    // SPFE_TILE2_MASK=0: never choose 2-row tiles
  // f32, a single frame: a POOLED low-resolution layer (conv3b: 180 eight-row items on 256 CUs — one round of the longest
  // items, 70 % of the CUs busy) as UN-pooled 2-row tiles (720 items: three rounds of quarter-size items) into a scratch
  // buffer + a 2x2 max-pool pass (pool2x2_f32_kernel; bias / ReLU / max commute exactly: same bits).  SPFE_POOL_SPLIT:
  // -1 cost model, 0 never, 1 wherever the shapes allow (tests)
  int pool_split = -1;
  float *d_unpooled = nullptr;   // [<= 2 frames][H / 4][W / 4][128]
  int conv1b_split_rows = -1; // ... and, when that launch was cut in a 16-row and an 8-row part, the 16-row part's tile rows ("conv1b_split_rows")
  int conv1b_tile_rows = 8;  // rows per tile of the last call's conv1b launch (f32; spfe_debug_read("conv1b_tile_rows"))
  int tile16x4 = 1;          // SPFE_TILE16X4: conv1b on 16-row tiles of 4 wavefronts x 4 rows (0 never, 1 by the cost model — possibly
                             // cut in a 16-row and an 8-row launch —, 2 always in one launch, 3 cost model without the cut)
  bool fuse1a = false;  // f32: conv1a computed inside conv1b in every call (SPFE_FUSE_CONV1A=1; perf-neutral on batches); unset: single-frame synchronous calls only
  bool fuse1a_bf16 = true;  // bf16: conv1a computed by the producer waves of the wave-specialised conv1b (SPFE_FUSE_CONV1A=0 to split)
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
    spfe_handle who = nullptr;   // which of a twin pair ran the batch (its heat maps, its side stream)
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

namespace spfe_host {

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

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// tickets as the caller of the C ABI sees them (one sequence per handle, twin or not)
inline long api_tickets(const spfe_handle h) { return h->twin ? h->g_ticket : h->ticket; }
inline spfe_handle_s::TicketRef ticket_ref(spfe_handle h, long t) {
  if (!h->twin) return {h, t};
  return h->tmap[t % 8];
}
inline spfe_handle last_caller(spfe_handle h) {   // the one of the pair whose buffers hold the last call's intermediates
  return h->twin && h->twin->last_seq > h->last_seq ? h->twin : h;
}

hipError_t wait_if_pending(hipStream_t s, hipEvent_t ev);
void make_layout(int kmax, int C, bool desc_bf16, spfe::RecordLayout *r);
// spfe_pack.hip
// (sibling: the handle whose twin `h` is to be — weights and environment switches are taken from it, not from the caller's
// pointers or the environment of that later moment)
int build(spfe_handle h, const spfe_config *cfg, spfe_handle sibling = nullptr);
// spfe_schedule.hip
int enqueue(spfe_handle h, const uint8_t *d_images, int n, uint8_t *d_records, hipStream_t s);
int enqueue_post(spfe_handle h, int n, uint8_t *d_records, hipStream_t s, const std::function<int()> *conv_db = nullptr, bool sparse = false, bool fused_pb = false, bool tail_done = false);
spfe::FrameBufs frame_bufs(spfe_handle h, uint8_t *d_records, bool sparse);
int tail_waits(spfe_handle h, uint8_t *d_records, hipStream_t s);
int launch_db_gathered(spfe_handle h, int n, hipStream_t s);
int launch_db_dense(spfe_handle h, int n, hipStream_t s);
int settle_join(spfe_handle h, hipStream_t s);
// spfe_api.hip
void view_record(const spfe_handle h, const uint8_t *rec, const float *heat, const float *heat_inv, spfe_result *out);

}  // namespace spfe_host

// spfe_api.hip: D2H of the last synchronous batch's records (+ heat maps) and the result views (not exported: hidden visibility)
extern "C" int finish_host(spfe_handle h, int n, spfe_result *outs);

namespace spfe {
__global__ void copy_records_kernel(uint4 *dst, const uint4 *src, size_t n16);
}
