// cov.hip — computeCovariance on the GPU, exact.
//
// Reference: /root/reference/orb_slam2/src/cv/sp_extractor.cpp:252-340.  For each
// keypoint IN EMITTED ORDER a FIFO breadth-first walk runs down the heat_inv hill
// (neighbours left, up, right, down; taken iff not yet popped by ANY keypoint,
// value > 0, value < current), then cov = sum (s_i / sum s) * delta_i^2 clamped to
// >= 1.  The shared visited mask makes the loop sequential over keypoints.
//
// Parallel form used here (one 1024-thread workgroup per frame, one thread per
// keypoint; every walk itself stays a sequential FIFO so pop order, duplicates and
// float accumulation order are the reference's):
//   A. every keypoint j walks ALONE (sees only its own visits) and claims each
//      popped pixel with atomicMin(claim[p], j).            -> region iso(j)
//   B. j is CLEAN when no pixel of iso(j) except its start was claimed by a lower
//      index: blocking only ever shrinks a walk, so seq(i) is a subset of iso(i)
//      and nothing an earlier keypoint really visited can touch iso(j); the lone
//      walk IS the sequential result.  Clean keypoints are final and stamp
//      done[p] = min(done[p], j).
//   C. the others are resolved in rounds: a non-final j whose region meets no
//      region of a non-final lower index re-walks against `done` (blocked iff
//      done[p] <= j: final lower keypoints and its own pops), becomes final and
//      stamps.  The lowest non-final index always qualifies, so the loop ends;
//      rounds = longest chain of overlapping keypoints (a handful).
// The result equals the sequential algorithm exactly (same pixels, same
// multiplicities, same order), not approximately.
#include "spfe_kernels.h"

namespace spfe {

#define COV_INF 0x7f7f7f7f

__device__ __forceinline__ int ld_agent(const int *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One FIFO walk.  Every pop does atomicMin(map[p], j).
// ISO (lone walk, map = claim): a pixel counts as visited iff THIS keypoint
//   popped it.  claim[p] == j: yes.  claim[p] > j: no (a pop by j would have
//   lowered it to <= j).  claim[p] < j: a lower keypoint holds the claim, so look
//   the pixel up in the own pop list q[0..head) (only happens where regions
//   overlap; keeps the walk exact and finite there).
// !ISO (re-walk, map = done): blocked iff done[p] <= j (final lower keypoints and
//   own pops).
// Returns the number of pops (= entries of q), or -1 on queue overflow.
template <bool ISO>
__device__ int walk(const float *__restrict__ hinv, int W, int H, int start, int j, int *q, int qcap,
                    int *map) {
  int head = 0, tail = 0;
  q[tail++] = start;
  while (head < tail) {
    const int id = q[head++];
    const int y = id / W, x = id - y * W;
    atomicMin(&map[id], j);
    const float here = hinv[id];
#define COV_VISIT(nid_)                                                  \
  do {                                                                   \
    const int nid = (nid_);                                              \
    const float v = hinv[nid];                                           \
    if (v > 0.0f && v < here) {                                          \
      const int m = ld_agent(&map[nid]);                                 \
      bool blocked = ISO ? (m == j) : (m <= j);                          \
      if (ISO && m < j) {                                                \
        for (int t = 0; t < head && !blocked; ++t) blocked = q[t] == nid; \
      }                                                                  \
      if (!blocked) {                                                    \
        if (tail >= qcap) return -1;                                     \
        q[tail++] = nid;                                                 \
      }                                                                  \
    }                                                                    \
  } while (0)
    if (x - 1 > 0) COV_VISIT(id - 1);
    if (y - 1 > 0) COV_VISIT(id - W);
    if (x + 1 < W) COV_VISIT(id + 1);
    if (y + 1 < H) COV_VISIT(id + W);
#undef COV_VISIT
  }
  return tail;
}

// second moments over the popped sequence, in pop order (:316-333)
__device__ void moments(const float *__restrict__ hinv, int W, const int *q, int n, int x0, int y0,
                        float *cov2, float *cov2_inv) {
  float sum = 0.0f;
  for (int i = 0; i < n; ++i) sum += hinv[q[i]];
  float cx = 0.0f, cy = 0.0f;
  for (int i = 0; i < n; ++i) {
    const int id = q[i];
    const int y = id / W, x = id - y * W;
    const float wgt = hinv[id] / sum;
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

__global__ __launch_bounds__(1024) void cov_kernel(FrameBufs f, RecordLayout rl, CovScratch cs, int H,
                                                    int W) {
  const int b = blockIdx.x, tid = threadIdx.x;
  uint8_t *rec = f.records + (size_t)b * rl.bytes;
  int *hdr = reinterpret_cast<int *>(rec + rl.off_hdr);
  const int K = hdr[0];
  const float *kp_xy = reinterpret_cast<const float *>(rec + rl.off_xy);
  float *cov2 = reinterpret_cast<float *>(rec + rl.off_cov);
  float *cinv = reinterpret_cast<float *>(rec + rl.off_cinv);
  const float *hinv = f.heat_inv + (size_t)b * H * W;
  int *claim = cs.claim + (size_t)b * H * W;
  int *done = cs.done + (size_t)b * H * W;
  int *queues = cs.queue + (size_t)b * rl.kmax * cs.qcap;
  int *npop = cs.npop + (size_t)b * rl.kmax;
  uint8_t *fin = cs.final_flag + (size_t)b * rl.kmax;
  __shared__ int s_pending, s_overflow;
  if (tid == 0) { s_pending = 0; s_overflow = 0; }
  __syncthreads();

  // ---- A: lone walks ----
  for (int j = tid; j < K; j += 1024) {
    const int x0 = (int)kp_xy[2 * j], y0 = (int)kp_xy[2 * j + 1];
    int *q = queues + (size_t)j * cs.qcap;
    const int n = walk<true>(hinv, W, H, y0 * W + x0, j, q, cs.qcap, claim);
    npop[j] = n;
    fin[j] = 0;
    if (n < 0) s_overflow = 1;
  }
  __syncthreads();
  if (s_overflow) {  // a region outgrew the per-keypoint queue: report, do not guess
    if (tid == 0) hdr[2] |= 1;
    return;
  }
  // ---- B: clean keypoints are final ----
  for (int j = tid; j < K; j += 1024) {
    const int *q = queues + (size_t)j * cs.qcap;
    const int n = npop[j];
    bool clean = true;
    for (int i = 1; i < n && clean; ++i) clean = ld_agent(&claim[q[i]]) >= j;
    if (clean) {
      const int x0 = (int)kp_xy[2 * j], y0 = (int)kp_xy[2 * j + 1];
      moments(hinv, W, q, n, x0, y0, cov2 + 2 * j, cinv + 2 * j);
      for (int i = 0; i < n; ++i) atomicMin(&done[q[i]], j);
      fin[j] = 1;
    } else {
      s_pending = 1;
    }
  }
  __syncthreads();
  // ---- C: resolve overlapping keypoints in rounds ----
  for (int round = 0; round < 1 << 20; ++round) {
    if (!s_pending) break;
    __syncthreads();
    if (tid == 0) s_pending = 0;
    // claim := lowest NON-FINAL index whose lone region holds the pixel
    for (int j = tid; j < K; j += 1024)
      if (!fin[j]) {
        const int *q = queues + (size_t)j * cs.qcap;
        for (int i = 0; i < npop[j]; ++i) __hip_atomic_store(&claim[q[i]], COV_INF, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    __syncthreads();
    for (int j = tid; j < K; j += 1024)
      if (!fin[j]) {
        const int *q = queues + (size_t)j * cs.qcap;
        for (int i = 0; i < npop[j]; ++i) atomicMin(&claim[q[i]], j);
      }
    __syncthreads();
    for (int j = tid; j < K; j += 1024) {
      if (fin[j]) continue;
      int *q = queues + (size_t)j * cs.qcap;
      const int n = npop[j];
      bool ready = true;
      for (int i = 1; i < n && ready; ++i) ready = ld_agent(&claim[q[i]]) >= j;
      if (!ready) { s_pending = 1; continue; }
      const int x0 = (int)kp_xy[2 * j], y0 = (int)kp_xy[2 * j + 1];
      // the re-walk overwrites this keypoint's own list; it is final afterwards
      const int m = walk<false>(hinv, W, H, y0 * W + x0, j, q, cs.qcap, done);
      // the re-walk's pop list is a subsequence of the lone walk's, so m <= n
      if (m < 0) { hdr[2] |= 1; fin[j] = 1; continue; }
      npop[j] = m;
      moments(hinv, W, q, m, x0, y0, cov2 + 2 * j, cinv + 2 * j);
      fin[j] = 1;
    }
    __syncthreads();
  }
}

hipError_t launch_cov(const FrameBufs &f, const RecordLayout &r, const CovScratch &cs, int B, int H, int W,
                      hipStream_t s) {
  hipError_t e = hipMemsetAsync(cs.claim, 0x7f, (size_t)B * H * W * 4, s);
  if (e != hipSuccess) return e;
  e = hipMemsetAsync(cs.done, 0x7f, (size_t)B * H * W * 4, s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(cov_kernel, dim3(B), dim3(1024), 0, s, f, r, cs, H, W);
  return hipGetLastError();
}

}  // namespace spfe
