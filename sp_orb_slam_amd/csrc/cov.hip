// cov.hip — computeCovariance on the GPU, exact.
//
// Reference: /root/reference/orb_slam2/src/cv/sp_extractor.cpp:252-340.  For each
// keypoint IN EMITTED ORDER a FIFO breadth-first walk runs down the heat_inv hill
// (neighbours left, up, right, down; taken iff not yet popped by ANY keypoint,
// value > 0, value < current), then cov = sum (s_i / sum s) * delta_i^2 clamped to
// >= 1.  The shared visited mask makes the loop sequential over keypoints.
//
// Parallel form used here (every walk itself stays a sequential FIFO run by one
// thread, so pop order, duplicate pops and float accumulation order are the
// reference's):
//   A. cov_walk_kernel: every keypoint j walks ALONE (sees only its own visits)
//      and claims each popped pixel with atomicMin(claim[p], j).  -> region iso(j)
//   B. cov_classify_kernel: j is CLEAN when no pixel of iso(j) except its start
//      was claimed by a lower index.  Blocking only ever shrinks a walk, so
//      seq(i) is a subset of iso(i): nothing an earlier keypoint really visited can
//      touch iso(j), and the lone walk IS the sequential result.  Clean keypoints
//      are final and stamp done[p] = min(done[p], j); the others go on the
//      frame's dirty list.
//   C. cov_components_kernel: two keypoints interact only if their lone regions
//      share a pixel, and every pixel links all its claimants to its lowest
//      claimant, so the connected components of {(claim[p], j) : p in iso(j)} are
//      closed under interaction and own disjoint pixel sets.  Each component is
//      replayed by ONE thread in ascending keypoint order against `done` (blocked
//      iff a lower FINAL keypoint or an earlier member of the component popped the
//      pixel) — exactly the sequential loop restricted to that component.  No
//      rounds, no barriers; the components of a frame run side by side.  For a
//      trained detector (regions of a few pixels) there are no dirty keypoints.
// The result equals the sequential algorithm exactly (same pixels, same
// multiplicities, same order), not approximately.
//
// Latency notes (the walks are dependent-load chains, one thread each): the FIFO
// and the keypoint's own visited set live in LDS (a 32x32-pixel bitmap window
// around the start; pixels outside it are looked up in the pop list), so a pop
// costs ONE round of independent global loads and no store or atomic sits in the
// loop (on CDNA4 stores share vmcnt with loads: a store per pop would put a full
// write round trip on the critical path).  Claims and stamps are issued in bulk
// after the walk.
#include "spfe_kernels.h"

namespace spfe {

#define COV_INF 0x7f7f7f7f
#define COV_LCAP 96      // FIFO entries kept in LDS per thread
#define COV_WIN 16       // own-visited bitmap covers dx,dy in [-16, 15]

__device__ __forceinline__ int ld_agent(const int *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct WalkMem {
  int *lq;        // LDS [COV_LCAP] pixel ids
  float *lqv;     // LDS [COV_LCAP] values
  uint32_t *bm;   // LDS [32] own-visited bitmap rows
  int *gq;        // global spill / final list [qcap]
  float *gqv;
  int qcap;
};

__device__ __forceinline__ int fifo_id(const WalkMem &m, int i) { return i < COV_LCAP ? m.lq[i] : m.gq[i]; }
__device__ __forceinline__ float fifo_val(const WalkMem &m, int i) { return i < COV_LCAP ? m.lqv[i] : m.gqv[i]; }

// One FIFO walk from (x0, y0) for keypoint j.
// LONE: only the keypoint's own pops block.  !LONE (replay): additionally every
// pixel with done[p] < j (final lower keypoints, earlier members of the component).
// Returns the number of pops (entries of the FIFO), or -1 when it outgrew qcap.
template <bool LONE>
__device__ int walk(const float *__restrict__ hinv, int W, int H, int x0, int y0, int j, const WalkMem &m,
                    const int *done) {
#pragma unroll
  for (int i = 0; i < 32; ++i) m.bm[i] = 0;
  int head = 0, tail = 1;
  int id = y0 * W + x0;
  float here = hinv[id];
  m.lq[0] = id;
  m.lqv[0] = here;
  while (true) {
    const int y = id / W, x = id - y * W;
    {  // visited at POP (:285)
      const int dx = x - x0 + COV_WIN, dy = y - y0 + COV_WIN;
      if ((unsigned)dx < 32u && (unsigned)dy < 32u) m.bm[dy] |= 1u << dx;
    }
    ++head;
    const bool ok[4] = {x - 1 > 0, y - 1 > 0, x + 1 < W, y + 1 < H};  // :302-313
    const int nx[4] = {x - 1, x, x + 1, x}, ny[4] = {y, y - 1, y, y + 1};
    int nid[4];
    float v[4];
    int dn[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) nid[t] = ok[t] ? ny[t] * W + nx[t] : id;
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] = hinv[nid[t]];
    if (!LONE) {
      // plain (cached) loads: during the replay a component's pixels are touched by
      // its one worker thread only; stamps of finished keypoints came from the
      // previous kernel.  Agent-scope (sc1) loads would bypass the XCD's L2 and
      // cost a fabric round trip per pop.
#pragma unroll
      for (int t = 0; t < 4; ++t) dn[t] = done[nid[t]];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {  // left, up, right, down
      if (!ok[t] || !(v[t] > 0.0f && v[t] < here)) continue;
      bool blocked = !LONE && dn[t] < j;
      if (!blocked) {
        const int dx = nx[t] - x0 + COV_WIN, dy = ny[t] - y0 + COV_WIN;
        if ((unsigned)dx < 32u && (unsigned)dy < 32u) {
          blocked = (m.bm[dy] >> dx) & 1u;
        } else {  // outside the bitmap window: search the pops so far
          for (int u = 0; u < head && !blocked; ++u) blocked = fifo_id(m, u) == nid[t];
        }
      }
      if (blocked) continue;
      if (tail >= m.qcap) return -1;
      if (tail < COV_LCAP) { m.lq[tail] = nid[t]; m.lqv[tail] = v[t]; }
      else { m.gq[tail] = nid[t]; m.gqv[tail] = v[t]; }
      ++tail;
    }
    if (head >= tail) break;
    id = fifo_id(m, head);
    here = fifo_val(m, head);
  }
  return tail;
}

// publish the LDS part of the pop list to global memory (later phases read it)
__device__ __forceinline__ void flush_list(const WalkMem &m, int n) {
  const int k = n < COV_LCAP ? n : COV_LCAP;
  for (int i = 0; i < k; ++i) { m.gq[i] = m.lq[i]; m.gqv[i] = m.lqv[i]; }
}

// second moments over the popped sequence, in pop order (:316-333)
__device__ void moments(int W, const WalkMem &m, int n, int x0, int y0, float *cov2, float *cov2_inv) {
  float sum = 0.0f;
  for (int i = 0; i < n; ++i) sum += fifo_val(m, i);
  float cx = 0.0f, cy = 0.0f;
  for (int i = 0; i < n; ++i) {
    const int id = fifo_id(m, i);
    const int y = id / W, x = id - y * W;
    const float wgt = fifo_val(m, i) / sum;
    const float dx = (float)x - (float)x0, dy = (float)y - (float)y0;
    cx += wgt * (dx * dx);
    cy += wgt * (dy * dy);
  }
  cx = cx < 1.0f ? 1.0f : cx;
  cy = cy < 1.0f ? 1.0f : cy;
  cov2[0] = cx;
  cov2[1] = cy;
  cov2_inv[0] = 1.0f / cx;
  cov2_inv[1] = 1.0f / cy;
}

struct CovFrame {
  const float *kp_xy;
  float *cov2, *cinv;
  int *hdr;
  const float *hinv;
  int *claim, *done, *queues, *npop, *dirty, *ndirty;
  float *qvals;
  int K;
};

__device__ __forceinline__ CovFrame cov_frame(const FrameBufs &f, const RecordLayout &rl, const CovScratch &cs,
                                              int b, int H, int W) {
  CovFrame c;
  uint8_t *rec = f.records + (size_t)b * rl.bytes;
  c.hdr = reinterpret_cast<int *>(rec + rl.off_hdr);
  c.K = c.hdr[0];
  c.kp_xy = reinterpret_cast<const float *>(rec + rl.off_xy);
  c.cov2 = reinterpret_cast<float *>(rec + rl.off_cov);
  c.cinv = reinterpret_cast<float *>(rec + rl.off_cinv);
  c.hinv = f.heat_inv + (size_t)b * H * W;
  c.claim = cs.claim + (size_t)b * H * W;
  c.done = cs.done + (size_t)b * H * W;
  c.queues = cs.queue + (size_t)b * rl.kmax * cs.qcap;
  c.qvals = cs.qval + (size_t)b * rl.kmax * cs.qcap;
  c.npop = cs.npop + (size_t)b * rl.kmax;
  c.dirty = cs.dirty + (size_t)b * rl.kmax;
  c.ndirty = cs.ndirty + b;
  return c;
}

// ---- A: lone walks.  64-thread workgroups spread over the CUs: a wave's accesses
// are fully divergent, so one CU's memory pipeline cannot feed many walks. ----
#define WALK_THREADS 64
__global__ __launch_bounds__(WALK_THREADS) void cov_walk_kernel(FrameBufs f, RecordLayout rl, CovScratch cs,
                                                                int H, int W) {
  __shared__ int s_q[WALK_THREADS * COV_LCAP];
  __shared__ float s_qv[WALK_THREADS * COV_LCAP];
  __shared__ uint32_t s_bm[WALK_THREADS * 32];
  const int b = blockIdx.y, tid = threadIdx.x, j = blockIdx.x * WALK_THREADS + tid;
  const CovFrame c = cov_frame(f, rl, cs, b, H, W);
  if (j >= c.K) return;
  WalkMem m{s_q + tid * COV_LCAP, s_qv + tid * COV_LCAP, s_bm + tid * 32, c.queues + (size_t)j * cs.qcap,
            c.qvals + (size_t)j * cs.qcap, cs.qcap};
  const int x0 = (int)c.kp_xy[2 * j], y0 = (int)c.kp_xy[2 * j + 1];
  const int n = walk<true>(c.hinv, W, H, x0, y0, j, m, nullptr);
  c.npop[j] = n;
  if (n < 0) { atomicOr(&c.hdr[2], 1); return; }  // region outgrew the queue: report, do not guess
  flush_list(m, n);
  // tentative moments: final if the keypoint turns out clean (classify decides)
  moments(W, m, n, x0, y0, c.cov2 + 2 * j, c.cinv + 2 * j);
  for (int i = 0; i < n; ++i) atomicMin(&c.claim[fifo_id(m, i)], j);
}

// ---- B: clean keypoints are final (their moments are already in the record);
// the rest go on the frame's dirty list ----
__global__ __launch_bounds__(64) void cov_classify_kernel(FrameBufs f, RecordLayout rl, CovScratch cs,
                                                          int H, int W) {
  const int b = blockIdx.y, j = blockIdx.x * 64 + threadIdx.x;
  const CovFrame c = cov_frame(f, rl, cs, b, H, W);
  if (j >= c.K || (c.hdr[2] & 1)) return;
  const int *q = c.queues + (size_t)j * cs.qcap;
  const int n = c.npop[j];
  int bad = 0;
  for (int i = 1; i < n; ++i) bad |= ld_agent(&c.claim[q[i]]) < j;  // independent loads, no early exit
  if (!bad) {
    for (int i = 0; i < n; ++i) atomicMin(&c.done[q[i]], j);
  } else {
    c.dirty[atomicAdd(c.ndirty, 1)] = j;
  }
}

// ---- C: components (see the header comment) ----
#define COMP_THREADS 128

__device__ __forceinline__ int uf_find(volatile int *parent, int x) {
  while (true) {
    const int p = parent[x];
    if (p == x) return x;
    x = p;
  }
}

__global__ __launch_bounds__(COMP_THREADS) void cov_components_kernel(FrameBufs f, RecordLayout rl,
                                                                      CovScratch cs, int H, int W) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const CovFrame c = cov_frame(f, rl, cs, b, H, W);
  const int nd = *c.ndirty;
  const int K = c.K;
  if (nd == 0 || (c.hdr[2] & 1)) return;
  extern __shared__ __attribute__((aligned(16))) int smem_i[];
  int *parent = smem_i;          // [K] union-find forest over keypoint indices
  int *leader = smem_i + K;      // [K] lowest DIRTY member of the component rooted here
  int *nxt = smem_i + 2 * K;     // [K] next dirty member (ascending) of the same component
  const int kpad = (3 * K + 3) & ~3;
  int *s_q = smem_i + kpad;
  float *s_qv = reinterpret_cast<float *>(s_q + COMP_THREADS * COV_LCAP);
  uint32_t *s_bm = reinterpret_cast<uint32_t *>(s_qv + COMP_THREADS * COV_LCAP);
  unsigned long long *dbg = cs.dbg ? cs.dbg + (size_t)b * 16 : nullptr;
  if (dbg && tid == 0) { dbg[0] = wall_clock64(); dbg[8] = nd; }
  for (int j = tid; j < K; j += COMP_THREADS) { parent[j] = j; leader[j] = COV_INF; nxt[j] = -1; }
  __syncthreads();
  // union every dirty keypoint with the lowest claimant of each of its pixels
  for (int d = tid; d < nd; d += COMP_THREADS) {
    const int j = c.dirty[d];
    const int *q = c.queues + (size_t)j * cs.qcap;
    const int n = c.npop[j];
    for (int i = 1; i < n; ++i) {
      int a = c.claim[q[i]];  // written by the previous kernels: plain load
      int bb = j;
      if (a >= j) continue;
      while (true) {  // hook the larger root under the smaller one
        a = uf_find(parent, a);
        bb = uf_find(parent, bb);
        if (a == bb) break;
        const int hi = a > bb ? a : bb, lo = a > bb ? bb : a;
        const int old = atomicMin(&parent[hi], lo);
        if (old == hi) break;
        a = old;
        bb = lo;
      }
    }
  }
  __syncthreads();
  if (dbg && tid == 0) dbg[1] = wall_clock64();
  for (int j = tid; j < K; j += COMP_THREADS) parent[j] = uf_find(parent, j);  // flatten (roots are fixed now)
  __syncthreads();
  for (int d = tid; d < nd; d += COMP_THREADS) {
    const int j = c.dirty[d];
    atomicMin(&leader[parent[j]], j);
  }
  __syncthreads();
  if (dbg && tid == 0) dbg[2] = wall_clock64();
  // next dirty member of the same component, ascending index
  for (int d = tid; d < nd; d += COMP_THREADS) {
    const int j = c.dirty[d];
    const int root = parent[j];
    int best = COV_INF;
    for (int e = 0; e < nd; ++e) {
      const int i = c.dirty[e];
      if (i > j && i < best && parent[i] == root) best = i;
    }
    nxt[j] = best == COV_INF ? -1 : best;
  }
  __syncthreads();
  if (dbg && tid == 0) dbg[3] = wall_clock64();
  // one thread per component replays its dirty members in order
  for (int d = tid; d < nd; d += COMP_THREADS) {
    int j = c.dirty[d];
    if (leader[parent[j]] != j) continue;
    unsigned long long tw = 0, tm = 0, ts = 0, pops = 0, chain = 0;
    while (j >= 0) {
      const unsigned long long t0 = dbg ? wall_clock64() : 0;
      WalkMem m{s_q + tid * COV_LCAP, s_qv + tid * COV_LCAP, s_bm + tid * 32, c.queues + (size_t)j * cs.qcap,
                c.qvals + (size_t)j * cs.qcap, cs.qcap};
      const int x0 = (int)c.kp_xy[2 * j], y0 = (int)c.kp_xy[2 * j + 1];
      // a replay's pop list is a subsequence of the lone walk's: it cannot overflow
      const int n = walk<false>(c.hinv, W, H, x0, y0, j, m, c.done);
      if (n < 0) { atomicOr(&c.hdr[2], 1); break; }
      const unsigned long long t1 = dbg ? wall_clock64() : 0;
      moments(W, m, n, x0, y0, c.cov2 + 2 * j, c.cinv + 2 * j);
      const unsigned long long t2 = dbg ? wall_clock64() : 0;
      // stamp before the next member starts (same thread, plain stores then plain
      // loads of the same addresses: coherent within the CU)
      for (int i = 0; i < n; ++i) {
        const int p = fifo_id(m, i);
        if (c.done[p] > j) c.done[p] = j;
      }
      if (dbg) { const unsigned long long t3 = wall_clock64(); tw += t1 - t0; tm += t2 - t1; ts += t3 - t2; pops += n; ++chain; }
      j = nxt[j];
    }
    if (dbg) { atomicMax(&dbg[9], tw); atomicMax(&dbg[10], tm); atomicMax(&dbg[11], ts); atomicMax(&dbg[5], chain);
               atomicMax(&dbg[6], pops); atomicAdd(&dbg[7], 1ull); }
  }
  __syncthreads();
  if (dbg && tid == 0) dbg[4] = wall_clock64();
}

size_t cov_components_lds(int kmax) {
  return (((size_t)kmax * 3 + 3) & ~(size_t)3) * sizeof(int) + (size_t)COMP_THREADS * (COV_LCAP * 8 + 32 * 4);
}

hipError_t launch_cov(const FrameBufs &f, const RecordLayout &r, const CovScratch &cs, int B, int H, int W,
                      hipStream_t s) {
  hipError_t e = hipMemsetAsync(cs.claim, 0x7f, (size_t)B * H * W * 4, s);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(cs.done, 0x7f, (size_t)B * H * W * 4, s);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(cs.ndirty, 0, (size_t)B * 4, s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(cov_walk_kernel, dim3((r.kmax + WALK_THREADS - 1) / WALK_THREADS, B), dim3(WALK_THREADS), 0,
                     s, f, r, cs, H, W);
  hipLaunchKernelGGL(cov_classify_kernel, dim3((r.kmax + 63) / 64, B), dim3(64), 0, s, f, r, cs, H, W);
  const size_t lds = cov_components_lds(r.kmax);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  e = hipFuncSetAttribute(reinterpret_cast<const void *>(cov_components_kernel),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(cov_components_kernel, dim3(B), dim3(COMP_THREADS), lds, s, f, r, cs, H, W);
  return hipGetLastError();
}

}  // namespace spfe
