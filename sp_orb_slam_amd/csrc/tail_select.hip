// tail_select.hip — detector tail, keypoint selection and descriptor sampling.
//
// Replaces, on the GPU, what the reference does with ~15 small ATen kernels plus
// host loops (/root/reference/orb_slam2/src/cv/sp_extractor.cpp):
//   tail_kernel       softmax / dustbin slices / per-cell arg-max / threshold /
//                     log-heat + pixel_shuffle               (:105-131)
//   heat_norm_kernel  to_heat affine normalisation            (:461-474)
//   select_kernel     score sort + greedy NMS + border reject + raster order +
//                     occ_grid                               (:489-498, :161-250)
//   desc_kernel       coarse L2-normalise + bilinear grid_sample + L2-normalise
//                     for the emitted keypoints only          (:102-103, :134-148)
// Float steps follow include/spfe_exact_math.h so every integer decision is
// bit-identical to the CPU oracle given identical logits.
#include <hip/hip_ext.h>

#include "spfe_kernels.h"
#include "../../include/spfe_exact_math.h"

#include "desc_body.h"
#include "tail_body.h"

namespace spfe {

#define TAIL_THREADS (TAIL_CELLS_PER_WG * 4)  // tail_kernel: 2 waves x 16 cells per workgroup (a DPP quad per cell, tail_body.h)

// The wave first copies its 16 x 65 contiguous logits into LDS with coalesced loads, each lane then reads its 16 (+ the
// dustbin).  min/max of the log-heat: one partial per workgroup (no atomics, no init).
__global__ __launch_bounds__(TAIL_THREADS) void tail_kernel(FrameBufs f, RecordLayout rl, int H, int W, int nparts, int *zero_ints, int nzero) {
  // (the bf16 convolutions' tile-queue counters, for the NEXT call: every convolution of this call is behind us in stream order)
  if (zero_ints && blockIdx.x == 0 && blockIdx.y == 0)
    for (int i = threadIdx.x; i < nzero; i += TAIL_THREADS) zero_ints[i] = 0;
  constexpr int CPW = 16, NW = TAIL_THREADS / 64;  // cells per wave, waves per workgroup
  const int wc = W >> 3, hc = H >> 3, C = hc * wc;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane & 3, lc = lane >> 2;
  const int b = blockIdx.y;
  const float *semi = f.semi + (size_t)b * C * SPFE_SEMI_CH;
  float *heat_log = f.heat_log + (size_t)b * H * W;
  uint8_t *rec = f.records + (size_t)b * rl.bytes;
  float *dense_dust = reinterpret_cast<float *>(rec + rl.off_dd);
  float *semi_dust = reinterpret_cast<float *>(rec + rl.off_sd);
  __shared__ float sm[NW][CPW * SPFE_SEMI_CH];
  __shared__ float smin[NW], smax[NW];

  const int cell0 = (blockIdx.x * NW + wave) * CPW;            // first cell of this wave
  const int ncell = C - cell0 < CPW ? C - cell0 : CPW;         // (<= 0: nothing to do)
  float lmin = 0.0f, lmax = -1e30f;                            // log-heat is <= 0
  if (ncell > 0) {
    const float *g = semi + (size_t)cell0 * SPFE_SEMI_CH;
    const int nfl = ncell * SPFE_SEMI_CH;
    for (int i = lane; i < nfl; i += 64) sm[wave][i] = g[i];
  }
  __builtin_amdgcn_wave_barrier();
  if (lc < ncell)
    tail_cell(&sm[wave][lc * SPFE_SEMI_CH], q, cell0 + lc, wc, W, heat_log, semi_dust, dense_dust, f.cell_score + (size_t)b * C,
              f.cell_k + (size_t)b * C, lmin, lmax);
  lmin = wave_min64(lmin);
  lmax = wave_max64(lmax);
  if (lane == 0) { smin[wave] = lmin; smax[wave] = lmax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = smin[0], c = smax[0];
    for (int i = 1; i < NW; ++i) { a = smin[i] < a ? smin[i] : a; c = smax[i] > c ? smax[i] : c; }
    float *part = reinterpret_cast<float *>(f.minmax) + ((size_t)b * nparts + blockIdx.x) * 2;
    part[0] = a;
    part[1] = c;
  }
}

int tail_parts(int H, int W) {  // workgroups (= min/max partials) per frame, of tail_kernel and of pbtail_f32_kernel alike
  const int C = (H / 8) * (W / 8);
  return (C + TAIL_CELLS_PER_WG - 1) / TAIL_CELLS_PER_WG;
}

hipError_t launch_tail(const FrameBufs &f, const RecordLayout &r, int B, int H, int W, hipStream_t s, int *zero_ints, int nzero) {
  const int nparts = tail_parts(H, W);
  hipLaunchKernelGGL(tail_kernel, dim3(nparts, B), dim3(TAIL_THREADS), 0, s, f, r, H, W, nparts, zero_ints, nzero);
  return hipGetLastError();
}

// to_heat (:461-474): img = -L, min/max as doubles, one affine map per pixel with
// float scale/shift, float multiply then float add (oracle_heat has the rule).
// The same pass also resets the covariance stage's scratch (claim / done maps to "nobody", counters to 0,
// overflow slots to "none"): it touches every pixel anyway, and four memset launches leave the latency-bound
// side chain.
__device__ __forceinline__ void heat_norm_body(const FrameBufs &f, int H, int W, int nparts, const CovScratch &cs, int kmax, int b,
                                               int blk, int nblk) {
  if (blk == 0) {
    if (threadIdx.x < 4) cs.counters[b * 4 + threadIdx.x] = 0;
    for (int k = threadIdx.x; k < kmax; k += 256) cs.ovf_slot[(size_t)b * kmax + k] = -1;
  }
  __shared__ float sc[4];
  if (threadIdx.x < 64) {
    const float *part = reinterpret_cast<const float *>(f.minmax) + (size_t)b * nparts * 2;
    float lo = 0.0f, hi = -1e30f;
    for (int i = threadIdx.x; i < nparts; i += 64) {
      lo = part[i * 2] < lo ? part[i * 2] : lo;
      hi = part[i * 2 + 1] > hi ? part[i * 2 + 1] : hi;
    }
    lo = wave_min64(lo);
    hi = wave_max64(hi);
    if (threadIdx.x == 0) {
      // m = -L : min(m) = -max(L), max(m) = -min(L)
      const double dmin = (double)(-hi), dmax = (double)(-lo);
      const double inv = 1.0 / (dmax - dmin);
      sc[0] = (float)(-inv);
      sc[1] = (float)(-dmin * inv);
      sc[2] = (float)(inv);
      sc[3] = (float)(dmax * inv);
      if (blk == 0) {
        float *hc4 = f.heat_consts + (size_t)b * 4;
        hc4[0] = sc[0]; hc4[1] = sc[1]; hc4[2] = sc[2]; hc4[3] = sc[3];
      }
    }
  }
  __syncthreads();
  const float a_h = sc[0], b_h = sc[1], a_i = sc[2], b_i = sc[3];
  const size_t n4 = (size_t)H * W / 4;
  const float4 *L4 = reinterpret_cast<const float4 *>(f.heat_log + (size_t)b * H * W);
  float4 *hi4 = f.heat_inv ? reinterpret_cast<float4 *>(f.heat_inv + (size_t)b * H * W) : nullptr;   // (an output only: SPFE_FLAG_HEAT)
  if (!hi4 && !cs.reset_maps && !f.heat) return;   // nothing to write per pixel: the covariance kernels form heat_inv where they read it
  float4 *h4 = f.heat ? reinterpret_cast<float4 *>(f.heat + (size_t)b * H * W) : nullptr;
  int4 *cl4 = reinterpret_cast<int4 *>(cs.claim + (size_t)b * H * W), *dn4 = reinterpret_cast<int4 *>(cs.done + (size_t)b * H * W);
  const int4 none = {COV_RESET, COV_RESET, COV_RESET, COV_RESET};
  const bool reset_maps = cs.reset_maps != 0;   // (else the entries' generation tags make earlier batches' read as "nobody": cov.hip)
  for (size_t i = (size_t)blk * 256 + threadIdx.x; i < n4; i += (size_t)nblk * 256) {
    if (reset_maps) { cl4[i] = none; dn4[i] = none; }
    if (!hi4 && !h4) continue;
    const float4 L = L4[i];
    float4 o;
    if (hi4) {
      o.x = L.x * a_i + b_i; o.y = L.y * a_i + b_i; o.z = L.z * a_i + b_i; o.w = L.w * a_i + b_i;
      hi4[i] = o;
    }
    if (h4) {
      o.x = L.x * a_h + b_h; o.y = L.y * a_h + b_h; o.z = L.z * a_h + b_h; o.w = L.w * a_h + b_h;
      h4[i] = o;
    }
  }
}

__global__ __launch_bounds__(256) void heat_norm_kernel(FrameBufs f, int H, int W, int nparts, CovScratch cs, int kmax) {
  heat_norm_body(f, H, W, nparts, cs, kmax, blockIdx.y, blockIdx.x, gridDim.x);
}

hipError_t launch_heat_norm(const FrameBufs &f, const CovScratch &cs, int kmax, int B, int H, int W, hipStream_t s) {
  const int blocks = !f.heat_inv && !f.heat && !cs.reset_maps ? 1 : (int)(((size_t)H * W / 4 + 255) / 256);
  hipLaunchKernelGGL(heat_norm_kernel, dim3(blocks < 128 ? blocks : 128, B), dim3(256), 0, s, f, H, W, tail_parts(H, W), cs, kmax);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Selection: one 1024-thread workgroup per frame, all state in LDS.
// The reference sorts candidates by score and runs a sequential greedy NMS on a
// byte image (:489-502, :161-250).  There is at most one candidate per 8x8 cell
// and the suppression window is Chebyshev radius 4 < 8, so a candidate interacts
// only with its 8 neighbouring cells: the greedy result is the unique fixed point
// of "alive iff no higher-ranked alive candidate within the window", reached by
// parallel rounds (no sort needed).  The break after num_features+1 survivors
// (:211-213) keeps the num_features+1 best-ranked survivors; border reject
// (:222-224) and raster numbering (:226-236) follow.
// ---------------------------------------------------------------------------
#ifdef SPFE_SELECT_PROBE   // phase timestamps of frame 0 (tools/microbench/run_selprof.sh builds a probe library with it)
#define SEL_TP(i) do { __syncthreads(); if (threadIdx.x == 0) tp[i] = __builtin_readcyclecounter(); } while (0)
#else
#define SEL_TP(i) do { } while (0)
#endif
enum : uint8_t { ST_NONE = 0, ST_UNDEC = 1, ST_ALIVE = 2, ST_DEAD = 3, ST_KEPT = 4 };

// Which neighbours CAN suppress a candidate (inside the window and ranked before it) never changes during the rounds,
// only their states do.  That 8-bit mask per candidate is local work — 8 neighbours x (score, k) from L2 — so it runs
// over the whole chip in front of the one-workgroup-per-frame kernel (where it was 41 k of 186 k cycles at 1280x720:
// one CU's VALU, 14 cells per thread).  Neighbour q: 0..2 row above (dx -1, 0, +1), 3 / 4 left / right, 5..7 row below.
__device__ __forceinline__ void nms_mask_body(const FrameBufs &f, int hc, int wc, int b, int blk) {
  const int C = hc * wc;
  const int c = blk * 256 + threadIdx.x;
  if (c >= C) return;
  const float *gscore = f.cell_score + (size_t)b * C;
  const uint8_t *gk = f.cell_k + (size_t)b * C;
  const float sc = gscore[c];
  unsigned m = 0;
  if (sc > 0.0f) {
    const int cy = c / wc, cx = c - cy * wc;
    const int k = gk[c];
    const int x = cx * 8 + (k & 7), y = cy * 8 + (k >> 3);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int dy = q < 3 ? -1 : (q < 5 ? 0 : 1);
      const int dx = q < 3 ? q - 1 : (q == 3 ? -1 : (q == 4 ? 1 : q - 6));
      const int ny = cy + dy, nx = cx + dx;
      const bool valid = (unsigned)ny < (unsigned)hc && (unsigned)nx < (unsigned)wc;
      const int n = valid ? ny * wc + nx : c;
      const float sn = gscore[n];   // 0 where there is no candidate
      const int kn = gk[n];
      const int ddx = nx * 8 + (kn & 7) - x, ddy = ny * 8 + (kn >> 3) - y;
      const bool close = ddx <= SPFE_NMS_DIST && ddx >= -SPFE_NMS_DIST && ddy <= SPFE_NMS_DIST && ddy >= -SPFE_NMS_DIST;
      if (valid && sn > 0.0f && close && spfe_ranks_before(sn, n, sc, c)) m |= 1u << q;
    }
  }
  f.cell_mask[(size_t)b * C + c] = (uint8_t)m;
}

__global__ __launch_bounds__(256) void nms_mask_kernel(FrameBufs f, int hc, int wc) {
  if (f.db_total && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *f.db_total = 0;   // select_kernel (next launch) appends
  nms_mask_body(f, hc, wc, blockIdx.y, blockIdx.x);
}
// The neighbour masks (input of the selection) and the heat normalisation (input of the covariance stage) both depend on the
// detector tail only: ONE launch, the first `nmask` blocks of a frame do the masks, the rest the normalisation — a kernel
// boundary less on the latency-bound side chain.
__global__ __launch_bounds__(256) void mask_and_heat_norm_kernel(FrameBufs f, int H, int W, int nparts, CovScratch cs, int kmax, int nmask) {
  if (f.db_total && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *f.db_total = 0;   // select_kernel (next launch) appends
  if ((int)blockIdx.x < nmask) nms_mask_body(f, H >> 3, W >> 3, blockIdx.y, blockIdx.x);
  else heat_norm_body(f, H, W, nparts, cs, kmax, blockIdx.y, (int)blockIdx.x - nmask, (int)gridDim.x - nmask);
}

// LDS of select_kernel.  Up to SELECT_SMALL_CELLS cells everything the kernel touches per cell sits in LDS (9 bytes a
// cell).  Larger frames (1920x1080 = 32,400 cells ... 65,535 cells) keep only what OTHER threads read during the rounds —
// the cell states and the neighbour masks, 2 bytes a cell — and the thread-private per-cell data (score, arg-max, the
// slot / index hand-off, the tie / layout list) in global scratch (FrameBufs::sel_slot / sel_list), which a single
// workgroup reads back coherently through its CU's L1 behind __syncthreads().
constexpr int SELECT_SMALL_CELLS = 16384;   // (also the register-resident key path of the cut: 16 cells per thread)
constexpr int SELECT_DB_WORDS = 2048 + 32;   // bitmap of the cells the descriptor head must compute (<= 65,535 cells) + scan scratch
static size_t select_fixed_lds(int H) { return ((size_t)(H / 8) + 16) * 4 * 2 + 64 + 1024 * 4 + 64 + 64 + SELECT_DB_WORDS * 4; }
bool select_big(int H, int W) { return (size_t)(H / 8) * (W / 8) > (size_t)SELECT_SMALL_CELLS; }
size_t select_lds_bytes(int H, int W, bool lean) {
  const size_t C = (size_t)(H / 8) * (W / 8);
  const size_t Cp = (C + 15) & ~(size_t)15;
  if (lean || select_big(H, W)) return Cp + Cp + select_fixed_lds(H);
  return Cp * 4 + Cp * 2 + Cp + Cp + Cp + select_fixed_lds(H);
}
size_t select_max_cells() { return 65535; }   // 16-bit cell indices (sList, row_of), 64 cells per thread

template <bool BIG>
__global__ __launch_bounds__(1024) void select_kernel(FrameBufs f, RecordLayout rl, int H, int W,
                                                      int num_features) {
  const int wc = W >> 3, hc = H >> 3, C = hc * wc;
  const int Cp = (C + 15) & ~15;
  const int b = blockIdx.x, tid = threadIdx.x;
  // c / wc for c < 2^16 as one multiply-high (exact: the error of ceil(2^32 / wc) stays below 2^-16 < 1 / wc)
  const unsigned wc_magic = wc > 1 ? 0xffffffffu / (unsigned)wc + 1u : 0u;
  auto row_of = [&](int c) -> int { return wc > 1 ? (int)__umulhi((unsigned)c, wc_magic) : c; };
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  // small frames: [score f32 | list u16 | k u8 | state u8 | mask u8 | rows ...]; big frames: [state u8 | mask u8 | rows ...]
  float *sScoreL = reinterpret_cast<float *>(smem);
  uint16_t *sListL = reinterpret_cast<uint16_t *>(sScoreL + Cp);
  uint8_t *sKL = reinterpret_cast<uint8_t *>(sListL + Cp);
  uint8_t *sState = BIG ? smem : sKL + Cp;
  uint8_t *sMask = sState + Cp;                      // per candidate: which of its 8 neighbours can suppress it
  int *sRow = reinterpret_cast<int *>(sMask + Cp);   // [hc+1] counts, then [hc+1] bases
  int *sRowBase = sRow + (hc + 16);
  int *sCnt = sRowBase + (hc + 16);  // [0] undecided flag, [1] survivors, [2] candidates, [3] K

  const float *gscore = f.cell_score + (size_t)b * C;
  const uint8_t *gk = f.cell_k + (size_t)b * C;
  uint8_t *rec = f.records + (size_t)b * rl.bytes;
  int *hdr = reinterpret_cast<int *>(rec + rl.off_hdr);
  float *kp_xy = reinterpret_cast<float *>(rec + rl.off_xy);
  int16_t *occ = reinterpret_cast<int16_t *>(rec + rl.off_occ);
  int *kp_cell = f.kp_cell + (size_t)b * rl.kmax;
  // per-cell data only the owner thread touches (plus one hand-off behind a barrier): LDS, or global scratch when BIG
  int *const slotp = BIG ? f.sel_slot + (size_t)b * C : reinterpret_cast<int *>(sScoreL);
  uint16_t *const sList = BIG ? f.sel_list + (size_t)b * C : sListL;
  auto score_of = [&](int c) -> float { return BIG ? gscore[c] : sScoreL[c]; };
  auto k_of = [&](int c) -> int { return BIG ? (int)gk[c] : (int)sKL[c]; };

#ifdef SPFE_SELECT_PROBE
  unsigned long long tp[12];
#endif
  SEL_TP(0);
  if (tid < 8) sCnt[tid] = 0;
  for (int i = tid; i < hc + 1; i += 1024) sRow[i] = 0;
  unsigned *const sBits = reinterpret_cast<unsigned *>(sRowBase + (hc + 16) + 16 + 1024 + 16);   // behind sHist: [2048] bits, [16] wave totals, [1] base
  if (f.db_list)
    for (int i = tid; i < 2048; i += 1024) sBits[i] = 0;
  __syncthreads();
  // a candidate nobody can suppress is alive from the start; `und` = this thread's undecided cells (bit j <-> cell
  // tid + 1024 j; C <= 65535), so that the rounds only touch what is still open
  const uint8_t *gmask = f.cell_mask + (size_t)b * C;
  const int lane = tid & 63, wave = tid >> 6;
  int ncand = 0;
  uint64_t und = 0, alive = 0;   // alive: this thread's cells in state ALIVE (the later passes walk bits, not LDS)
#pragma unroll 4
  for (int c = tid, j = 0; c < C; c += 1024, ++j) {
    const float s = gscore[c];
    const uint8_t m = gmask[c];
    if constexpr (!BIG) { sScoreL[c] = s; sKL[c] = gk[c]; }
    sMask[c] = m;
    sState[c] = s > 0.0f ? (m ? ST_UNDEC : ST_ALIVE) : ST_NONE;
    und |= (uint64_t)(s > 0.0f && m) << j;
    alive |= (uint64_t)(s > 0.0f && !m) << j;
    ncand += s > 0.0f;
  }
  if (ncand) atomicAdd(&sCnt[2], ncand);
  __syncthreads();
  SEL_TP(1);
  // ---- NMS fixed point ----
  SEL_TP(2);
  // every round decides at least the best-ranked undecided candidate, so C rounds always suffice (a strictly
  // rank-ordered staircase across the cells is the worst case); real frames need a handful
  // ONE barrier per round: the "somebody is still undecided" flag rotates through four slots (slot r & 3 is written in
  // round r, read behind its barrier, cleared in round r + 2 — after every thread has passed barrier r + 1 — and written
  // again in round r + 4), and the states need none: they only move UNDEC -> ALIVE / DEAD, both decisions below rest on
  // final states of the neighbours, so reading a state a thread has just written in this very round is as good as reading it
  // in the next (the fixed point is unique).  The rounds were three barriers of a 1024-thread workgroup each.
  int *sFlag = sCnt + 4;   // [4], zero on entry; sCnt[4..6] are re-used by the cut below
  for (int round = 0; round < C; ++round) {
    if (tid == 0) sFlag[(round + 2) & 3] = 0;
    for (uint64_t rest = und; rest;) {
      const int j = __ffsll((long long)rest) - 1;
      rest &= rest - 1;
      const int c = tid + j * 1024;
      const unsigned m = sMask[c];
      bool dead = false, blocked = false;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int off = (q < 3 ? -wc : (q < 5 ? 0 : wc)) + (q < 3 ? q - 1 : (q == 3 ? -1 : (q == 4 ? 1 : q - 6)));
        const bool on = (m >> q) & 1u;
        const uint8_t st = sState[on ? c + off : c];  // own state (UNDEC) when the neighbour does not matter
        dead |= on && st == ST_ALIVE;
        blocked |= on && st == ST_UNDEC;
      }
      if (dead) sState[c] = ST_DEAD;
      else if (!blocked) sState[c] = ST_ALIVE;
      if (dead || !blocked) und &= ~(1ull << j);
      if (!dead && !blocked) alive |= 1ull << j;
    }
    const int pending = und != 0;
    if (pending) sFlag[round & 3] = 1;
    __syncthreads();
    if (sFlag[round & 3] == 0) break;
  }
  __syncthreads();
  if (tid < 4) sFlag[tid] = 0;
  __syncthreads();

  SEL_TP(3);
  // ---- keep the num_features+1 best-ranked survivors (:211-213) ----
  // Radix select on the score bits (positive floats order like their bit patterns), 10 bits a pass over the RANGE the
  // alive keys actually span (key - min: 2^26 for softmax scores between 1/65 and 1, so three passes): every thread holds
  // its <= 16 cells' keys in registers, a pass is one plain LDS atomic per key still in the running into a 1024-bin
  // histogram (range-normalised, so the first pass spreads over the bins), a suffix scan with one bin per thread, and
  // the bucket holding the (num_features+1)-th best becomes the next pass's range.  Afterwards `prefix` is that
  // survivor's key, everything above it is kept, and among equal keys the lowest cell indices win (tie rule).
  // (First version: four 8-bit passes over the raw bits with wave-aggregated atomics, 9 ballots per cell and pass —
  // 46 % of this kernel, 51 k of 110 k cycles at 752x480; a bisection on the key with register-held keys: 29 k.)
  int *sHist = sRowBase + (hc + 16) + 16;  // [1024] + [16] wave totals
  const int ns = __popcll(alive);
  if (ns) atomicAdd(&sCnt[1], ns);
  __syncthreads();
  const int S = sCnt[1];
  SEL_TP(4);
  uint32_t prefix = 0;
  int need = num_features + 1;
  const bool cut = S > need;
  if (cut) {
    constexpr int KREG = 16;                       // cells per thread kept in registers (C <= 16384)
    const bool in_regs = C <= KREG * 1024;         // (BIG on a small frame — the lean form of pipelined calls — reads its scores once)
    uint32_t key[KREG];
    uint32_t kmin = 0xffffffffu, kmax = 0u;
#pragma unroll
    for (int j = 0; j < KREG; ++j) {
      const int c = tid + j * 1024;
      key[j] = (in_regs && ((alive >> j) & 1)) ? __float_as_uint(score_of(c)) : 0u;
      if (key[j]) { kmin = key[j] < kmin ? key[j] : kmin; kmax = key[j] > kmax ? key[j] : kmax; }
    }
    if (!in_regs)
      for (uint64_t rest = alive; rest; rest &= rest - 1) {
        const uint32_t k = __float_as_uint(score_of(tid + (__ffsll((long long)rest) - 1) * 1024));
        kmin = k < kmin ? k : kmin; kmax = k > kmax ? k : kmax;
      }
    int *sAcc = sCnt + 4;
    if (tid == 0) { sAcc[0] = 0x7fffffff; sAcc[1] = 0; }
    __syncthreads();
    if (kmax) { atomicMin(&sAcc[0], (int)kmin); atomicMax(&sAcc[1], (int)kmax); }   // (positive floats: int order = key order)
    __syncthreads();
    SEL_TP(5);
    const uint32_t lo0 = (uint32_t)sAcc[0], range = (uint32_t)sAcc[1] - lo0;
    int ub = 32 - __clz((int)(range | 1u));        // unresolved low bits of (key - lo0); the bucket is [base, base + 2^ub)
    uint32_t base = 0;
    int *sWaveTot = sHist + 1024;
    while (ub > 0) {
      const int sh = ub > 10 ? ub - 10 : 0;
      sHist[tid] = 0;
      __syncthreads();
      if (in_regs) {
#pragma unroll
        for (int j = 0; j < KREG; ++j) {
          const uint32_t r = key[j] - lo0 - base;    // (wraps to something huge for keys below the bucket and for key 0)
          if (key[j] && (r >> ub) == 0) atomicAdd(&sHist[r >> sh], 1);
        }
      } else {
        for (uint64_t rest = alive; rest; rest &= rest - 1) {
          const uint32_t r = __float_as_uint(score_of(tid + (__ffsll((long long)rest) - 1) * 1024)) - lo0 - base;
          if ((r >> ub) == 0) atomicAdd(&sHist[r >> sh], 1);
        }
      }
      __syncthreads();
      const int mine = sHist[tid];                  // suffix sums: `above` = keys of the bucket in higher bins
      int suffix = mine;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_down(suffix, off, 64);
        if (lane + off < 64) suffix += o;
      }
      if (lane == 0) sWaveTot[wave] = suffix;
      __syncthreads();
      int above = suffix - mine;
      for (int w2 = wave + 1; w2 < 16; ++w2) above += sWaveTot[w2];
      if (above < need && above + mine >= need) { sCnt[4] = tid; sCnt[5] = need - above; }
      __syncthreads();
      base += (uint32_t)sCnt[4] << sh;
      need = sCnt[5];
      ub = sh;
    }
    prefix = lo0 + base;
    SEL_TP(6);
    __syncthreads();
    // ties on the threshold key: count them; if more than `need`, rank by index
    if (tid == 0) sCnt[6] = 0;
    __syncthreads();
    for (uint64_t rest = alive; rest; rest &= rest - 1) {
      const int c = tid + (__ffsll((long long)rest) - 1) * 1024;
      if (__float_as_uint(score_of(c)) == prefix) sList[atomicAdd(&sCnt[6], 1)] = (uint16_t)c;
    }
    __syncthreads();
  }
  SEL_TP(7);
  const int nties = cut ? sCnt[6] : 0;
  // keep / border reject, and the cell row's kept count on the way: a kept cell takes the next slot of its row (any
  // order) and parks it where its score was (only the owner thread touches a cell in this phase)
  uint64_t kept = 0;
  for (uint64_t rest = alive; rest; rest &= rest - 1) {
    const int j = __ffsll((long long)rest) - 1;
    const int c = tid + j * 1024;
    bool keep = true;
    if (cut) {
      const uint32_t key = __float_as_uint(score_of(c));
      if (key < prefix) keep = false;
      else if (key == prefix && nties > need) {
        int lower = 0;
        for (int t = 0; t < nties; ++t) lower += sList[t] < c;
        keep = lower < need;
      }
    }
    if (!keep) continue;
    const int cy = row_of(c), cx = c - cy * wc;
    const int k = k_of(c);
    const int x = cx * 8 + (k & 7), y = cy * 8 + (k >> 3);
    // border reject (:222-224)
    if (!(x < SPFE_NMS_BORDER || x >= W - SPFE_NMS_BORDER || y < SPFE_NMS_BORDER ||
          y >= H - SPFE_NMS_BORDER)) {
      sState[c] = ST_KEPT;
      slotp[c] = atomicAdd(&sRow[cy], 1);
      kept |= 1ull << j;
    }
  }
  __syncthreads();

  SEL_TP(8);
  // ---- raster order (y outer, x inner) (:220-238) and occ_grid (:227-228) ----
  // The order is (cell row, dy, cx).  Exclusive prefix over the cell rows' counts; the kept cells are laid out grouped by
  // cell row (slot order inside a row); one thread per keypoint then ranks its (dy, cx) inside its row's group — a dozen
  // entries on a 1000-keypoint frame, at most wc.  (Was: one wavefront per cell row, 2 x 8 ballots per 64 cells — a
  // quarter of this kernel's cycles.)
  if (wave == 0) {  // a wave scan per 64 rows
    int carry = 0;
    for (int r0 = 0; r0 < hc; r0 += 64) {
      const int r = r0 + lane;
      const int v = r < hc ? sRow[r] : 0;
      int incl = v;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
      }
      if (r < hc) sRowBase[r] = carry + incl - v;
      carry += __shfl(incl, 63, 64);
    }
    if (lane == 0) {
      sCnt[3] = carry;
      hdr[0] = carry;     // K
      hdr[1] = sCnt[2];   // n_candidates
      hdr[2] = 0;         // status
      hdr[3] = S;         // NMS survivors before the cut (diagnostic)
    }
  }
  __syncthreads();
  SEL_TP(9);
  for (uint64_t rest = kept; rest; rest &= rest - 1) {   // (the tie list in sList is no longer needed)
    const int c = tid + (__ffsll((long long)rest) - 1) * 1024;
    sList[sRowBase[row_of(c)] + slotp[c]] = (uint16_t)c;
  }
  __syncthreads();
  SEL_TP(10);
  const int K = sCnt[3];
  for (int t = tid; t < K; t += 1024) {
    const int c = sList[t];
    const int cy = row_of(c), cx = c - cy * wc;
    const int k = k_of(c);
    const int mykey = ((k >> 3) << 16) | cx;
    const int g0 = sRowBase[cy], g1 = g0 + sRow[cy];
    int idx = g0;
    for (int u = g0; u < g1; ++u) {
      const int cu = sList[u];
      idx += (((k_of(cu) >> 3) << 16) | (cu - cy * wc)) < mykey;
    }
    kp_xy[2 * idx] = (float)(cx * 8 + (k & 7));
    kp_xy[2 * idx + 1] = (float)(cy * 8 + (k >> 3));
    kp_cell[idx] = c;
    slotp[c] = idx;
    if (f.db_list) {   // the coarse cells this keypoint's descriptor reads (desc_keypoint's taps, same arithmetic)
      const DescTaps tp = desc_taps((float)(cx * 8 + (k & 7)), (float)(cy * 8 + (k >> 3)), H, W);
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) {
        const int txx = tp.x0 + (t4 & 1), tyy = tp.y0 + (t4 >> 1);
        if (txx < 0 || txx >= wc || tyy < 0 || tyy >= hc) continue;
        const int cell = tyy * wc + txx;
        atomicOr(&sBits[cell >> 5], 1u << (cell & 31));
      }
    }
  }
  __syncthreads();
  if (f.db_list) {
    // the marked cells, in cell order, appended to the batch's list: two bitmap words per thread, a workgroup scan of their
    // counts, one global atomic per frame for the base (the frames' order in the list does not matter: a cell's output row
    // depends on its own input row only)
    const unsigned b0 = sBits[2 * tid], b1 = sBits[2 * tid + 1];
    const int cnt = __popc(b0) + __popc(b1);
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(incl, off, 64);
      if (lane >= off) incl += o;
    }
    int *sTot = reinterpret_cast<int *>(sBits + 2048);
    if (lane == 63) sTot[wave] = incl;
    __syncthreads();
    int wbase = 0, total = 0;
#pragma unroll
    for (int w2 = 0; w2 < 16; ++w2) {
      const int v = sTot[w2];
      wbase += w2 < wave ? v : 0;
      total += v;
    }
    if (tid == 0) sTot[16] = atomicAdd(f.db_total, total);
    __syncthreads();
    int pos = sTot[16] + wbase + incl - cnt;
    for (unsigned rest = b0; rest; rest &= rest - 1) f.db_list[pos++] = b * C + 64 * tid + (__ffs((int)rest) - 1);
    for (unsigned rest = b1; rest; rest &= rest - 1) f.db_list[pos++] = b * C + 64 * tid + 32 + (__ffs((int)rest) - 1);
  }
#pragma unroll 4
  for (int c = tid; c < C; c += 1024)
    occ[c] = sState[c] == ST_KEPT ? (int16_t)slotp[c] : (int16_t)-1;
#ifdef SPFE_SELECT_PROBE
  SEL_TP(11);
  if (tid == 0 && b == 0)
    printf("SELECT C=%d S=%d | load %llu mask %llu nms %llu count %llu keys+range %llu passes %llu ties %llu keep+rowcnt %llu rowprefix %llu place %llu rank+occ %llu | total %llu\n",
           C, S, tp[1] - tp[0], tp[2] - tp[1], tp[3] - tp[2], tp[4] - tp[3], tp[5] - tp[4], tp[6] - tp[5], tp[7] - tp[6], tp[8] - tp[7],
           tp[9] - tp[8], tp[10] - tp[9], tp[11] - tp[10], tp[11] - tp[0]);
#endif
}

// ---------------------------------------------------------------------------
// Frames of more than 65,535 cells (3840x2160 = 129,600; the reference accepts any multiple of 8, sp_extractor.cpp:70).
// select_kernel's 16-bit cell indices, its 64 cells per thread and its 2 bytes of LDS a cell end there.  This form is the same
// algorithm — parallel rounds to the greedy NMS's fixed point, radix cut at the (num_features + 1)-th best survivor with the
// lowest-index tie rule, border reject, raster numbering by (cell row, dy, cx), occ_grid, the descriptor head's cell list —
// with EVERYTHING per cell in global scratch (state, slot, list: FrameBufs::sel_state / sel_slot / sel_list32; the
// neighbour masks are read where nms_mask_kernel wrote them) and 64 NW cells per thread.  One workgroup per frame reads its
// own writes back through its CU's L1 (workgroup scope), with the same barriers as select_kernel.  Built for coverage, not
// for speed: a 3840x2160 frame's convolutions take ~10 ms, this kernel a fraction of one.  SPFE_SELECT_HUGE=1 runs it on
// frames of any size (tests).
// (The reference's `inds` are 16 bit (:177,187,229): a candidate of sorted rank >= 65,536 would be emitted with the position
// and descriptor of the candidate 65,536 ranks above it.  It cannot happen for num_features <= 10,000: a survivor's 9x9 window
// touches at most 2 x 2 cells, so the first 4 (num_features + 1) <= 40,004 candidates in sorted order hold num_features + 1
// survivors and the loop breaks (:211-213) before rank 65,536 is reached.)
// ---------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(1024) void select_huge_kernel(FrameBufs f, RecordLayout rl, int H, int W, int num_features) {
  const int wc = W >> 3, hc = H >> 3, C = hc * wc;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  int *sRow = reinterpret_cast<int *>(smem);          // [hc + 16] kept cells per cell row
  int *sRowBase = sRow + (hc + 16);                    // [hc + 16] exclusive prefix
  int *sCnt = sRowBase + (hc + 16);                    // [16]
  int *sHist = sCnt + 16;                              // [1024] + [16] wave totals
  unsigned *sBits = reinterpret_cast<unsigned *>(sHist + 1024 + 16);   // [nwords] cells the descriptor head must compute, + [17] scan scratch
  const int nwords = (C + 31) >> 5;

  const float *gscore = f.cell_score + (size_t)b * C;
  const uint8_t *gk = f.cell_k + (size_t)b * C;
  const uint8_t *gmask = f.cell_mask + (size_t)b * C;
  uint8_t *gstate = f.sel_state + (size_t)b * C;
  int *slotp = f.sel_slot + (size_t)b * C;
  int *gList = f.sel_list32 + (size_t)b * C;
  uint8_t *rec = f.records + (size_t)b * rl.bytes;
  int *hdr = reinterpret_cast<int *>(rec + rl.off_hdr);
  float *kp_xy = reinterpret_cast<float *>(rec + rl.off_xy);
  int16_t *occ = reinterpret_cast<int16_t *>(rec + rl.off_occ);
  int *kp_cell = f.kp_cell + (size_t)b * rl.kmax;
  // cell of bit j of word w of this thread's masks
  auto cell_of = [&](int w, int j) -> int { return tid + (w * 64 + j) * 1024; };

  if (tid < 16) sCnt[tid] = 0;
  for (int i = tid; i < hc + 1; i += 1024) sRow[i] = 0;
  if (f.db_list)
    for (int i = tid; i < nwords + 17; i += 1024) sBits[i] = 0;
  __syncthreads();
  uint64_t und[NW], alive[NW], kept[NW];
  int ncand = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    und[w] = alive[w] = kept[w] = 0;
    for (int j = 0; j < 64; ++j) {
      const int c = cell_of(w, j);
      if (c >= C) break;
      const float sc = gscore[c];
      const uint8_t m = gmask[c];
      gstate[c] = sc > 0.0f ? (m ? ST_UNDEC : ST_ALIVE) : ST_NONE;
      und[w] |= (uint64_t)(sc > 0.0f && m) << j;
      alive[w] |= (uint64_t)(sc > 0.0f && !m) << j;
      ncand += sc > 0.0f;
    }
  }
  if (ncand) atomicAdd(&sCnt[2], ncand);
  __syncthreads();
  // ---- NMS fixed point (select_kernel's rounds; the states live in global memory) ----
  int *sFlag = sCnt + 8;   // [4]
  for (int round = 0; round < C; ++round) {
    if (tid == 0) sFlag[(round + 2) & 3] = 0;
    bool pending = false;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      for (uint64_t rest = und[w]; rest;) {
        const int j = __ffsll((long long)rest) - 1;
        rest &= rest - 1;
        const int c = cell_of(w, j);
        const unsigned m = gmask[c];
        bool dead = false, blocked = false;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int off = (q < 3 ? -wc : (q < 5 ? 0 : wc)) + (q < 3 ? q - 1 : (q == 3 ? -1 : (q == 4 ? 1 : q - 6)));
          const bool on = (m >> q) & 1u;
          const uint8_t st = __atomic_load_n(&gstate[on ? c + off : c], __ATOMIC_RELAXED);
          dead |= on && st == ST_ALIVE;
          blocked |= on && st == ST_UNDEC;
        }
        if (dead) __atomic_store_n(&gstate[c], (uint8_t)ST_DEAD, __ATOMIC_RELAXED);
        else if (!blocked) __atomic_store_n(&gstate[c], (uint8_t)ST_ALIVE, __ATOMIC_RELAXED);
        if (dead || !blocked) und[w] &= ~(1ull << j);
        if (!dead && !blocked) alive[w] |= 1ull << j;
      }
      pending |= und[w] != 0;
    }
    if (pending) sFlag[round & 3] = 1;
    __syncthreads();
    if (sFlag[round & 3] == 0) break;
  }
  __syncthreads();
  // ---- keep the num_features + 1 best-ranked survivors (:211-213): radix select over the keys' range, as select_kernel ----
  int ns = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) ns += __popcll(alive[w]);
  if (ns) atomicAdd(&sCnt[1], ns);
  __syncthreads();
  const int S = sCnt[1];
  uint32_t prefix = 0;
  int need = num_features + 1;
  const bool cut = S > need;
  if (cut) {
    uint32_t kmin = 0xffffffffu, kmax = 0u;
#pragma unroll
    for (int w = 0; w < NW; ++w)
      for (uint64_t rest = alive[w]; rest; rest &= rest - 1) {
        const uint32_t k = __float_as_uint(gscore[cell_of(w, __ffsll((long long)rest) - 1)]);
        kmin = k < kmin ? k : kmin; kmax = k > kmax ? k : kmax;
      }
    int *sAcc = sCnt + 4;
    if (tid == 0) { sAcc[0] = 0x7fffffff; sAcc[1] = 0; }
    __syncthreads();
    if (kmax) { atomicMin(&sAcc[0], (int)kmin); atomicMax(&sAcc[1], (int)kmax); }
    __syncthreads();
    const uint32_t lo0 = (uint32_t)sAcc[0], range = (uint32_t)sAcc[1] - lo0;
    int ub = 32 - __clz((int)(range | 1u));
    uint32_t base = 0;
    int *sWaveTot = sHist + 1024;
    while (ub > 0) {
      const int sh = ub > 10 ? ub - 10 : 0;
      sHist[tid] = 0;
      __syncthreads();
#pragma unroll
      for (int w = 0; w < NW; ++w)
        for (uint64_t rest = alive[w]; rest; rest &= rest - 1) {
          const uint32_t r = __float_as_uint(gscore[cell_of(w, __ffsll((long long)rest) - 1)]) - lo0 - base;
          if ((r >> ub) == 0) atomicAdd(&sHist[r >> sh], 1);
        }
      __syncthreads();
      const int mine = sHist[tid];
      int suffix = mine;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_down(suffix, off, 64);
        if (lane + off < 64) suffix += o;
      }
      if (lane == 0) sWaveTot[wave] = suffix;
      __syncthreads();
      int above = suffix - mine;
      for (int w2 = wave + 1; w2 < 16; ++w2) above += sWaveTot[w2];
      if (above < need && above + mine >= need) { sCnt[4] = tid; sCnt[5] = need - above; }
      __syncthreads();
      base += (uint32_t)sCnt[4] << sh;
      need = sCnt[5];
      ub = sh;
      __syncthreads();   // (sCnt[4..5] are read above and rewritten by the next pass)
    }
    prefix = lo0 + base;
    __syncthreads();
    if (tid == 0) sCnt[6] = 0;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NW; ++w)
      for (uint64_t rest = alive[w]; rest; rest &= rest - 1) {
        const int c = cell_of(w, __ffsll((long long)rest) - 1);
        if (__float_as_uint(gscore[c]) == prefix) gList[atomicAdd(&sCnt[6], 1)] = c;
      }
    __syncthreads();
  }
  const int nties = cut ? sCnt[6] : 0;
  // keep / border reject; a kept cell takes the next slot of its cell row
#pragma unroll
  for (int w = 0; w < NW; ++w)
    for (uint64_t rest = alive[w]; rest; rest &= rest - 1) {
      const int j = __ffsll((long long)rest) - 1;
      const int c = cell_of(w, j);
      bool keep = true;
      if (cut) {
        const uint32_t key = __float_as_uint(gscore[c]);
        if (key < prefix) keep = false;
        else if (key == prefix && nties > need) {
          int lower = 0;
          for (int t = 0; t < nties; ++t) lower += gList[t] < c;
          keep = lower < need;
        }
      }
      if (!keep) continue;
      const int cy = c / wc, cx = c - cy * wc;
      const int k = gk[c];
      const int x = cx * 8 + (k & 7), y = cy * 8 + (k >> 3);
      if (!(x < SPFE_NMS_BORDER || x >= W - SPFE_NMS_BORDER || y < SPFE_NMS_BORDER || y >= H - SPFE_NMS_BORDER)) {   // :222-224
        gstate[c] = ST_KEPT;
        slotp[c] = atomicAdd(&sRow[cy], 1);
        kept[w] |= 1ull << j;
      }
    }
  __syncthreads();   // (also: every thread is done with the tie list before the layout below reuses gList)
  // ---- raster order (y outer, x inner) (:220-238) and occ_grid (:227-228): (cell row, dy, cx) ----
  if (wave == 0) {
    int carry = 0;
    for (int r0 = 0; r0 < hc; r0 += 64) {
      const int r = r0 + lane;
      const int v = r < hc ? sRow[r] : 0;
      int incl = v;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
      }
      if (r < hc) sRowBase[r] = carry + incl - v;
      carry += __shfl(incl, 63, 64);
    }
    if (lane == 0) {
      sCnt[3] = carry;
      hdr[0] = carry;     // K
      hdr[1] = sCnt[2];   // n_candidates
      hdr[2] = 0;         // status
      hdr[3] = S;         // NMS survivors before the cut (diagnostic)
    }
  }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < NW; ++w)
    for (uint64_t rest = kept[w]; rest; rest &= rest - 1) {
      const int c = cell_of(w, __ffsll((long long)rest) - 1);
      gList[sRowBase[c / wc] + slotp[c]] = c;
    }
  __syncthreads();
  const int K = sCnt[3];
  for (int t = tid; t < K; t += 1024) {
    const int c = gList[t];
    const int cy = c / wc, cx = c - cy * wc;
    const int k = gk[c];
    const int mykey = ((k >> 3) << 20) | cx;
    const int g0 = sRowBase[cy], g1 = g0 + sRow[cy];
    int idx = g0;
    for (int u = g0; u < g1; ++u) {
      const int cu = gList[u];
      idx += ((((int)gk[cu] >> 3) << 20) | (cu - cy * wc)) < mykey;
    }
    kp_xy[2 * idx] = (float)(cx * 8 + (k & 7));
    kp_xy[2 * idx + 1] = (float)(cy * 8 + (k >> 3));
    kp_cell[idx] = c;
    slotp[c] = idx;   // (only this thread reads or writes slotp[c] from here on: each kept cell is one list entry)
    if (f.db_list) {
      const DescTaps tp = desc_taps((float)(cx * 8 + (k & 7)), (float)(cy * 8 + (k >> 3)), H, W);
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) {
        const int txx = tp.x0 + (t4 & 1), tyy = tp.y0 + (t4 >> 1);
        if (txx < 0 || txx >= wc || tyy < 0 || tyy >= hc) continue;
        const int cell = tyy * wc + txx;
        atomicOr(&sBits[cell >> 5], 1u << (cell & 31));
      }
    }
  }
  __syncthreads();
  if (f.db_list) {
    // the marked cells, in cell order, appended to the batch's list: thread t owns words [t wpt, (t + 1) wpt)
    const int wpt = (nwords + 1023) >> 10;
    int cnt = 0;
    for (int i = 0; i < wpt; ++i) {
      const int wi = tid * wpt + i;
      cnt += wi < nwords ? __popc(sBits[wi]) : 0;
    }
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(incl, off, 64);
      if (lane >= off) incl += o;
    }
    int *sTot = reinterpret_cast<int *>(sBits + nwords);
    if (lane == 63) sTot[wave] = incl;
    __syncthreads();
    int wbase = 0, total = 0;
#pragma unroll
    for (int w2 = 0; w2 < 16; ++w2) {
      const int v = sTot[w2];
      wbase += w2 < wave ? v : 0;
      total += v;
    }
    if (tid == 0) sTot[16] = atomicAdd(f.db_total, total);
    __syncthreads();
    int pos = sTot[16] + wbase + incl - cnt;
    for (int i = 0; i < wpt; ++i) {
      const int wi = tid * wpt + i;
      if (wi >= nwords) break;
      for (unsigned rest = sBits[wi]; rest; rest &= rest - 1) f.db_list[pos++] = b * C + 32 * wi + (__ffs((int)rest) - 1);
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += 1024)
    occ[c] = gstate[c] == ST_KEPT ? (int16_t)slotp[c] : (int16_t)-1;
}
constexpr int SELECT_HUGE_NW = 4;   // 64 x 4 cells per thread: 262,144 cells
size_t select_huge_lds_bytes(int H, int W) {
  const size_t C = (size_t)(H / 8) * (W / 8);
  return ((size_t)(H / 8) + 16) * 4 * 2 + 16 * 4 + (1024 + 16) * 4 + ((C + 31) / 32 + 17) * 4;
}
size_t select_huge_max_cells() { return (size_t)64 * SELECT_HUGE_NW * 1024 - 1; }

hipError_t launch_select(const FrameBufs &f, const RecordLayout &r, int B, int H, int W,
                         int num_features, hipStream_t s, const CovScratch *with_heat_norm, int kmax_hn, bool lean,
                         hipEvent_t done, hipEvent_t heat_done) {
  if (f.sel_huge) {   // frames of more than 65,535 cells (or SPFE_SELECT_HUGE=1): everything per cell in global scratch
    const size_t lds = select_huge_lds_bytes(H, W);
    if (lds > 160 * 1024 || (size_t)(H / 8) * (W / 8) > select_huge_max_cells() || !f.sel_state || !f.sel_slot || !f.sel_list32)
      return hipErrorInvalidValue;
    auto k = select_huge_kernel<SELECT_HUGE_NW>;
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
    }
    if (with_heat_norm) {
      const int nmask = ((H / 8) * (W / 8) + 255) / 256;
      const int hb = !f.heat_inv && !f.heat && !with_heat_norm->reset_maps ? 1 : (int)(((size_t)H * W / 4 + 255) / 256);   // (one block when there is nothing to write per pixel: the scale / shift only)
      if (heat_done) hipExtLaunchKernelGGL(mask_and_heat_norm_kernel, dim3(nmask + (hb < 128 ? hb : 128), B), dim3(256), 0, s, nullptr, heat_done, 0, f, H, W,
                                           tail_parts(H, W), *with_heat_norm, kmax_hn, nmask);
      else hipLaunchKernelGGL(mask_and_heat_norm_kernel, dim3(nmask + (hb < 128 ? hb : 128), B), dim3(256), 0, s, f, H, W, tail_parts(H, W),
                         *with_heat_norm, kmax_hn, nmask);
    } else {
      hipLaunchKernelGGL(nms_mask_kernel, dim3(((H / 8) * (W / 8) + 255) / 256, B), dim3(256), 0, s, f, H / 8, W / 8);
    }
    if (done) hipExtLaunchKernelGGL(k, dim3(B), dim3(1024), (unsigned)lds, s, nullptr, done, 0, f, r, H, W, num_features);
    else hipLaunchKernelGGL(k, dim3(B), dim3(1024), lds, s, f, r, H, W, num_features);
    return hipGetLastError();
  }
  const bool big = select_big(H, W) || (lean && f.sel_slot && f.sel_list);
  const size_t lds = select_lds_bytes(H, W, big);
  if (lds > 160 * 1024 || (size_t)(H / 8) * (W / 8) > select_max_cells()) return hipErrorInvalidValue;
  if (big && (!f.sel_slot || !f.sel_list)) return hipErrorInvalidValue;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(big ? reinterpret_cast<const void *>(select_kernel<true>)
                                           : reinterpret_cast<const void *>(select_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  if (with_heat_norm) {
    const int nmask = ((H / 8) * (W / 8) + 255) / 256;
      const int hb = !f.heat_inv && !f.heat && !with_heat_norm->reset_maps ? 1 : (int)(((size_t)H * W / 4 + 255) / 256);   // (one block when there is nothing to write per pixel: the scale / shift only)
    if (heat_done) hipExtLaunchKernelGGL(mask_and_heat_norm_kernel, dim3(nmask + (hb < 128 ? hb : 128), B), dim3(256), 0, s, nullptr, heat_done, 0, f, H, W,
                                         tail_parts(H, W), *with_heat_norm, kmax_hn, nmask);
    else hipLaunchKernelGGL(mask_and_heat_norm_kernel, dim3(nmask + (hb < 128 ? hb : 128), B), dim3(256), 0, s, f, H, W, tail_parts(H, W),
                       *with_heat_norm, kmax_hn, nmask);
  } else {
    hipLaunchKernelGGL(nms_mask_kernel, dim3(((H / 8) * (W / 8) + 255) / 256, B), dim3(256), 0, s, f, H / 8, W / 8);
  }
  // `done`: the event rides on the selection's own dispatch packet (its completion signal) instead of a marker packet behind
  // it — a hipEventRecord between two kernels of a latency chain costs the stream ~7 us (spfe_schedule.hip, enqueue_post)
  if (done) {
    if (big) hipExtLaunchKernelGGL(select_kernel<true>, dim3(B), dim3(1024), (unsigned)lds, s, nullptr, done, 0, f, r, H, W, num_features);
    else hipExtLaunchKernelGGL(select_kernel<false>, dim3(B), dim3(1024), (unsigned)lds, s, nullptr, done, 0, f, r, H, W, num_features);
  } else if (big) hipLaunchKernelGGL(select_kernel<true>, dim3(B), dim3(1024), lds, s, f, r, H, W, num_features);
  else hipLaunchKernelGGL(select_kernel<false>, dim3(B), dim3(1024), lds, s, f, r, H, W, num_features);
  return hipGetLastError();
}


__global__ __launch_bounds__(256) void desc_kernel(FrameBufs f, RecordLayout rl, int H, int W) {
  desc_keypoint(f, rl, H, W, blockIdx.y, blockIdx.x * 4 + (threadIdx.x >> 6), threadIdx.x & 63);
}

hipError_t launch_desc(const FrameBufs &f, const RecordLayout &r, int B, int H, int W, hipStream_t s) {
  hipLaunchKernelGGL(desc_kernel, dim3((r.kmax + 3) / 4, B), dim3(256), 0, s, f, r, H, W);
  return hipGetLastError();
}

// exact-math probe: device spfe_expf / spfe_logf on arbitrary inputs
__global__ void math_probe_kernel(const float *in, float *oe, float *ol, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    oe[i] = spfe_expf(in[i]);
    ol[i] = spfe_logf(in[i] < 0.0f ? -in[i] : in[i]);
  }
}
hipError_t launch_math_probe(const float *in, float *out_exp, float *out_log, int n, hipStream_t s) {
  hipLaunchKernelGGL(math_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, s, in, out_exp, out_log, n);
  return hipGetLastError();
}

}  // namespace spfe
