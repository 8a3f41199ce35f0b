// spfe_schedule.hip — the per-batch launch sequence of the path (SPFrontend::forward + the host glue of operator(),
// /root/reference/orb_slam2/src/cv/sp_extractor.cpp:79-159, :361-514): which kernel on which stream, ordered by which
// event; the side chain (selection, descriptors, covariance) of batch i beside the convolutions of batch i + 1.
#include "spfe_host.h"

namespace spfe_host {

// Order stream `s` behind `ev` — but only if `ev` has not fired yet.  A wait is a barrier packet in the compute queue and
// costs ~10 us of idle queue whether or not the event is long done (measured on kernel timelines of the pipelined steps:
// conv1a -> [wait] -> conv1b 12 us apart); the waits below guard buffers against work TWO batches back, which in steady state
// finished long ago: one hipEventQuery on the host replaces the packet.  (Not under stream capture: a query is illegal there,
// and a captured wait is a graph edge, not a packet.)
hipError_t wait_if_pending(hipStream_t s, hipEvent_t ev) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone) {
    const hipError_t q = hipEventQuery(ev);
    if (q == hipSuccess) return hipSuccess;
    if (q != hipErrorNotReady) (void)hipGetLastError();   // (e.g. an event never recorded: fall through to the wait)
  }
  return hipStreamWaitEvent(s, ev, 0);
}

#define STAGE_MARK(i) \
  do { if (h->timing && (h->timing_all || (i) == 1 || (i) == 2)) HIP_TRY(hipEventRecord(h->ev[i], s)); } while (0)

// D2H of the records by a kernel of our own that writes the pinned (device-mapped) host buffer: 8.9 MB in ~0.18 ms, no LDS,
// fits beside the persistent convolution workgroups; the runtime's own D2H path cost 0.36 ms more per batch in the pipeline
// (SPFE_PIPE_COPY_KERNEL=0 selects it)
}  // namespace spfe_host
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
// ended (within microseconds: a larger gap is a late launch, not a shared queue, and counts as "not measurable").  No host
// timer is involved; the outcome is
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
namespace spfe_host {
// 1 = the two streams share a hardware queue, 0 = they do not, -1 = could not be measured (error / ambiguous)
int streams_share_a_queue(hipStream_t a, hipStream_t b, long long *h_stamp /* pinned, 4 entries */) {
  constexpr long long kTicks = 15000;   // 150 us
  if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return -1;
  int shared_votes = 0;
  for (int rep = 0; rep < 3; ++rep) {   // (rep 0 includes the kernel's code load: its stamps are not used)
    for (int i = 0; i < 4; ++i) h_stamp[i] = 0;
    hipLaunchKernelGGL(spfe::queue_probe_spin_kernel, dim3(1), dim3(64), 0, a, kTicks, h_stamp);
    hipLaunchKernelGGL(spfe::queue_probe_spin_kernel, dim3(1), dim3(64), 0, b, kTicks, h_stamp + 2);
    if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return -1;
    if (rep == 0) continue;
    const long long a0 = h_stamp[0], a1 = h_stamp[1], b0 = h_stamp[2], b1 = h_stamp[3];
    if (a1 <= a0 || b1 <= b0) continue;   // (a stamp did not arrive: try once more)
    // overlap of the two intervals against the spin length: more than half = two queues.  None at all has two causes that
    // look alike from the overlap alone: ONE queue — the second spin starts right where the first ended, a few us
    // apart — or a host thread that was held up for > 150 us between the two launches — the second spin then starts whenever
    // it was launched.  The gap tells them apart; "shared" is only returned when both measured repetitions say so.
    const long long ov = std::min(a1, b1) - std::max(a0, b0);
    if (ov >= kTicks / 2) return 0;
    const long long gap = b0 >= a1 ? b0 - a1 : a0 - b1;   // between the end of one spin and the start of the other
    if (ov <= kTicks / 10 && gap >= 0 && gap <= 3000) ++shared_votes;   // (<= 30 us: back to back on one queue)
  }
  return shared_votes == 2 ? 1 : -1;
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
  if (h->open_n) return fail(SPFE_EINVAL, "a call begun by spfe_extract_begin is open: spfe_extract_finish first");
  if (h->timing) h->ev = h->evpool.data() + (size_t)(h->calls % spfe_handle_s::EVSETS) * (NSTAGE + 1);
  h->calls++;
  h->host_sync_n = 0;   // (heat_inv is about to be rewritten: finish_host() says when a synchronous host call's maps are complete)
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
  // f32: conv1b computes conv1a's outputs itself — on request (SPFE_FUSE_CONV1A=1; perf-neutral on batches), and by default in
  // single-frame synchronous calls, where a launch and its boundary less are worth 5 us (752x480: p50 0.7266 -> 0.7213 ms)
  const bool fused = !h->bf16 && (h->fuse1a || (h->fuse1a_env < 0 && n == 1 && !((h->cfg.flags & SPFE_FLAG_ASYNC_COV) || h->pipe_mode) &&
                                                !(h->timing && h->timing_all)));
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
  // Predicted from the last call's schedule; a wrong guess only repeats the (satisfied) waits later.
  bool early_waits = false;
  {
    if (h->early_waits && h->split_last && h->pbtail && n >= 2 && ((h->cfg.flags & SPFE_FLAG_ASYNC_COV) || h->pipe_mode) && !(h->timing && h->timing_all)) {
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
        // (measured and not kept: conv1b / all Cin = 64 layers on fewer workgroups than CUs in pipelined calls, so that the
        // previous batch's selection — 143 KB of LDS, nothing fits beside this kernel's 158 KB — starts beside conv1b: 1280x720
        // x 8 on 224 workgroups +0.3 ... 2 % with conv1b at 0.51 - 0.53 of peak instead of 0.57; 248 / 240 / 208 / 192: -1 / -2
        // / -1 / -3 %.  HISTORY.md "Round 4"; re-measured in round 5)
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
                                           h->d_tile_ctr ? h->d_tile_ctr + 32 * part : nullptr, h->d_tile_ctr ? 8 * 32 : 0, 64,
                                           h->pbtail_env > 1 ? h->pbtail_env : 0));
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
  bool split = (h->split_mode >= 0 ? h->split_mode >= 1 : (!h->bf16 || h->C < 10000)) && !(!h->bf16 && h->f32_heads) && n >= 2 && !(h->timing && h->timing_all) && !defer_db;
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
    tail_per_half = h->pbtail && h->tail_per_half;
    if (tail_per_half && !early_waits) {
      const int rcw = tail_waits(h, d_records, s);
      if (rcw) return rcw;
    }
    int rc = run_layer(0);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(h->ev_fork, s));
    HIP_TRY(hipStreamWaitEvent(h->conv2, h->ev_fork, 0));
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
  f.heat_log = h->d_heat_log[par]; f.heat = h->d_heat;
  f.heat_inv = (h->cfg.flags & SPFE_FLAG_HEAT) ? h->d_heat_inv : nullptr;   // materialised as an output only: the covariance kernels read heat_log
  f.minmax = reinterpret_cast<uint32_t *>(h->d_minmax[par]);
  f.cell_score = h->d_cell_score[par]; f.cell_k = h->d_cell_k[par]; f.cell_mask = h->d_cell_mask; f.kp_cell = h->d_kp_cell;
  f.sel_slot = h->d_sel_slot; f.sel_list = h->d_sel_list;
  f.sel_state = h->d_sel_state; f.sel_list32 = h->d_sel_list32; f.sel_huge = h->select_huge ? 1 : 0;
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
    if (h->rec_of[prev] == d_records) HIP_TRY(wait_if_pending(s, h->ev_cov[prev]));
  }
  return SPFE_OK;
}

// Synchronous host calls with heat maps: `ev` fires when the heat normalisation has completed; the maps' D2H goes to a copy
// stream of its own behind it (spfe_host.h: s_heat).  Returns the event to hand to launch_select, or null when this call does
// not send its maps ahead.
static hipEvent_t heat_event_for(spfe_handle h, hipStream_t s) {
  h->heat_early = false;
  if (!h->host_sync_call || !h->early_heat_copy || !(h->cfg.flags & SPFE_FLAG_HEAT) || (h->timing && h->timing_all)) return nullptr;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return nullptr;
  if (!h->s_heat) {
    if (hipStreamCreateWithFlags(&h->s_heat, hipStreamNonBlocking) != hipSuccess) { h->s_heat = nullptr; return nullptr; }
    if (hipEventCreateWithFlags(&h->ev_heat, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_heat_copied1, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_heat_copied, hipEventDisableTiming) != hipSuccess) { h->early_heat_copy = false; return nullptr; }
  }
  return h->ev_heat;
}
static int send_heat_maps_ahead(spfe_handle h, int n) {
  const size_t bytes = (size_t)n * h->H * h->W * 4;
  HIP_TRY(hipStreamWaitEvent(h->s_heat, h->ev_heat, 0));
  HIP_TRY(hipMemcpyAsync(h->h_heat, h->d_heat, bytes, hipMemcpyDeviceToHost, h->s_heat));
  HIP_TRY(hipEventRecord(h->ev_heat_copied1, h->s_heat));   // (`heat` alone: spfe_extract_maps hands it out while heat_inv travels)
  if (!(h->cfg.flags & SPFE_FLAG_LAZY_HEAT_INV))
    HIP_TRY(hipMemcpyAsync(h->h_heat_inv, h->d_heat_inv, bytes, hipMemcpyDeviceToHost, h->s_heat));
  HIP_TRY(hipEventRecord(h->ev_heat_copied, h->s_heat));
  h->heat_early = true;
  return SPFE_OK;
}

// Detector tail, selection, descriptors, covariance for n frames whose semi /
// coarse maps are in the handle's buffers.  tail_done: the detector tail (inside pbtail_f32_kernel) was launched per half
// batch by enqueue(), behind tail_waits().
int enqueue_post(spfe_handle h, int n, uint8_t *d_records, hipStream_t s, const std::function<int()> *conv_db, bool sparse, bool fused_pb, bool tail_done) {
  const int H = h->H, W = h->W;
  if (h->open_n) return fail(SPFE_EINVAL, "a call begun by spfe_extract_begin is open: spfe_extract_finish first");
  spfe::FrameBufs f = frame_bufs(h, d_records, sparse);
  h->sparse_last = sparse;
  h->desc_early = false;   // (set below when this call sends its descriptor rows ahead)
  { static long g_call_seq = 0; h->last_seq = ++g_call_seq; }   // (handles are driven by one thread each; a pair by the same one)
  {   // this chain's generation of the claim / done maps (cov.hip): one code per chain, counting down; a full reset of the maps
      // only before a frame's first use and when the codes are used up
    h->cov_gen_code = h->cov_gen_code > 1 ? h->cov_gen_code - 1 : 0;
    h->cov.reset_maps = 0;
    if (h->cov_gen_code == 0 || n > h->cov_frames_clean) {
      // (after a wrap only the frames reset NOW are clean: a frame this call does not touch keeps entries of the old cycle,
      // whose codes come round again — it is reset before its next use, like a frame never used)
      if (h->cov_gen_code == 0) { h->cov_gen_code = h->cov_gen_start; h->cov_frames_clean = 0; }
      h->cov.reset_maps = 1;
      h->cov_frames_clean = std::max(h->cov_frames_clean, n);
    }
    h->cov.gen = h->cov_gen_code << 16;
    // a captured call is replayed with the arguments of the capture — the same generation G every time: it clears its maps
    // itself, as every call did before round 5, and every replay leaves entries tagged G behind.  A later direct call with a
    // LOWER code reads them as "nobody" and wins every atomicMin against them.  Once the codes have wrapped (they restart at
    // cov_gen_start > G) that is no longer true — a direct call behind a replay would lose its claims to the stale, lower
    // G-tagged entries — so a handle that has been captured remembers the lowest code it was captured with, and
    // direct calls clear the maps themselves while their code is not below it.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
      h->cov.reset_maps = 1;
      h->cov_captured_min = h->cov_captured_min ? std::min(h->cov_captured_min, h->cov_gen_code) : h->cov_gen_code;
    } else if (h->cov_captured_min && h->cov_gen_code >= h->cov_captured_min) {
      h->cov.reset_maps = 1;
      h->cov_frames_clean = std::max(h->cov_frames_clean, n);
    }
  }
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
    HIP_TRY(spfe::launch_pbtail_bf16(h->d_hd, h->d_wpb, h->layers[8].d_b, h->d_semi, f, h->rl, n, H, W, s, 0, h->d_tile_ctr, h->d_tile_ctr ? 8 * 64 : 0, 32,
                                     h->pbtail_env > 1 ? h->pbtail_env : 0));
    if (h->d_tile_ctr && h->zero_in_tail) h->tile_ctr_clean = true;
  } else if (fused_pb) HIP_TRY(spfe::launch_pbtail_f32(h->d_head, h->d_wpb32, h->d_wpb_dust, h->layers[8].d_b, h->d_semi, f, h->rl, n, H, W, s));
  else {
    HIP_TRY(spfe::launch_tail(f, h->rl, n, H, W, s, h->d_tile_ctr, h->d_tile_ctr ? 8 * 64 : 0));
    if (h->d_tile_ctr && h->zero_in_tail) h->tile_ctr_clean = true;
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
  if (h->inline_chain && sparse && !conv_db && !((h->cfg.flags & SPFE_FLAG_ASYNC_COV) || h->pipe_mode) && !(h->timing && h->timing_all)) {
    if (h->cov_inflight) {   // (a pipelined call's chain still on the side stream — it owns heat_inv and the covariance scratch)
      HIP_TRY(hipStreamWaitEvent(s, h->ev_cov[(h->ticket + spfe_handle_s::NTICKET - 1) % spfe_handle_s::NTICKET], 0));
      h->cov_inflight = false;
    }
    STAGE_MARK(13);
    // (the event the side stream waits for is the selection's own completion signal: a hipEventRecord here put a marker
    // packet between the selection and the covariance walk — 7.6 us on the chain; SPFE_SEL_EXT_EVENT=0: that record)
    // (under stream capture the record it is: the stop event of an extended launch is not a capture node, the side stream
    // would not join the capture and its kernels would run once, at capture time)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool sel_ext = h->sel_ext_event && hipStreamIsCapturing(s, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone;
    hipEvent_t const heat_ev = heat_event_for(h, s);
    HIP_TRY(spfe::launch_select(f, h->rl, n, H, W, h->cfg.num_features, s, &h->cov, h->rl.kmax, false,
                                sel_ext ? h->ev_sel : nullptr, heat_ev));
    if (heat_ev) { const int rch = send_heat_maps_ahead(h, n); if (rch) return rch; }
    if (!sel_ext) HIP_TRY(hipEventRecord(h->ev_sel, s));
    HIP_TRY(hipStreamWaitEvent(h->side, h->ev_sel, 0));
    int rc = launch_db_gathered(h, n, h->side);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(h->ev_dbs[par], h->side));
    h->dbs_recorded[par] = true;
    HIP_TRY(spfe::launch_desc(f, h->rl, n, H, W, h->side));
    h->desc_early = false;
    if (h->host_sync_call && h->early_heat_copy && d_records == h->d_records && cap == hipStreamCaptureStatusNone) {
      // a synchronous host call: the descriptor rows are final here — their D2H rides the side stream beside the covariance
      // chain (spfe_host.h: desc_early); rows of the n records, a 2-D copy with the record stride as pitch
      const size_t seg = h->rl.off_occ - h->rl.off_desc;
      HIP_TRY(hipMemcpy2DAsync(h->h_records + h->rl.off_desc, h->rl.bytes, h->d_records + h->rl.off_desc, h->rl.bytes, seg, (size_t)n,
                               hipMemcpyDeviceToHost, h->side));
      // ... and the headers, for K (final with the selection; finish_host brings the whole head again): spfe_extract_rows
      HIP_TRY(hipMemcpy2DAsync(h->h_records + h->rl.off_hdr, h->rl.bytes, h->d_records + h->rl.off_hdr, h->rl.bytes, 16, (size_t)n,
                               hipMemcpyDeviceToHost, h->side));
      h->desc_early = true;
    }
    HIP_TRY(hipEventRecord(h->ev_desc, h->side));
    h->desc_recorded = true;
    HIP_TRY(spfe::launch_cov(f, h->rl, h->cov, n, H, W, s, false, nullptr, 2, true));   // (a synchronous call: moments deferred, cov.hip)
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
    hipEvent_t const heat_ev = heat_event_for(h, h->side);
    HIP_TRY(spfe::launch_select(f, h->rl, n, H, W, h->cfg.num_features, h->side, &h->cov, h->rl.kmax, false, nullptr, heat_ev));
    if (heat_ev) { const int rch = send_heat_maps_ahead(h, n); if (rch) return rch; }
  }
  // Synchronous calls: the descriptor sampling rides in the covariance replay launch (the chain's longest kernel) instead of
  // standing in front of the chain; pipelined calls keep it early — the NEXT call's convDb waits for it, and behind a replay
  // that shares the chip with that call's convolutions it would wait too long (0.5 ms steps in bf16 mode).
  const bool sync_call = !((h->cfg.flags & SPFE_FLAG_ASYNC_COV) || h->pipe_mode);
  // pipelined calls, sparse: nothing on the launch stream waits for the sampling any more (the dense convDb of the NEXT call
  // did), so it may ride in the replay launch there too: bf16 1280x720 +0.2 %, f32 752x480 -0.7 % (kept early in f32 mode)
  const bool desc_in_replay = (sync_call || (sparse && h->bf16)) && !(h->timing && h->timing_all);
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
    // (measured, same-box A/B, 8 frames per call: bf16 1280x720 +0.7 %, bf16 752x480 -1.8 %, f32 -1 %: large bf16 frames only)
    const int rwv = h->replay_waves ? h->replay_waves : (h->bf16 && !sync_call && h->C >= 10000 ? 8 : 2);
    HIP_TRY(spfe::launch_cov(f, h->rl, h->cov, n, H, W, h->side, desc_in_replay, before_replay, rwv, sync_call));
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

}  // namespace spfe_host
