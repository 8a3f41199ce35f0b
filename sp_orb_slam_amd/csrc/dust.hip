// dust.hip — direct "dust" alignment on the GPU.
//
// Replaces Optimizer::PoseOptimizationDust(Frame*, const vector<MapPoint*>&, vector<bool>&)
// (/root/reference/orb_slam2/src/mapping/optimizer_dust.cpp:170-294): a 6-DoF Levenberg-Marquardt over
// <= a few hundred map points, each edge (g2o::EdgeSE3ProjectDustOnlyPose,
// src/optimization/types_dust_tracking.cpp:37-140) sampling the extractor's dense_dust map bilinearly at the
// point's projection (intrinsics fx/8, (cx-3.5)/8) — Huber(0.9), 40 iterations.  The map is already in HBM
// (it is part of the frame's record), so the pose refinement runs where the data is and only the 4x4 pose,
// the inlier flags and the projections leave.
//
// The problem is tiny and sequential (up to 40 outer iterations x up to 10 trial steps, each needing global
// sums): ONE workgroup per solve, and what is optimised is the LATENCY of one trial step.
//
//   * A thread owns two map points.  Two kinds of phase, each ending in ONE workgroup barrier: a trial's
//     computeActiveErrors (error per edge, chi2 = one quantity through the tree) and an iteration's buildSystem
//     (Jacobian per edge, the 27 entries of H's lower triangle and b through the tree).  A Levenberg iteration
//     of this problem rejects more trials than it accepts (43 trials for 17 iterations on the bench scene), so
//     the Jacobians are formed once per iteration, not speculatively per trial.  g2o evaluates the errors again
//     at the start of the next iteration; after an accepted trial that is the same pose, the same sticky level
//     flags and the same tree — it is skipped (only a NaN gain ratio continues after a rejected trial: then it
//     is run).
//   * The sums follow the header's fixed-shape tree (slot = thread; halving tree per wavefront; the four
//     wavefronts in order).  Within a wavefront the 27 sums are a reduce-scatter butterfly: at the level with
//     partner lane ^ m a lane keeps one half of its values and sends the other, so the six levels cost 16 + 8 +
//     4 + 2 + 1 + 1 = 32 exchanges instead of 6 x 27, and lane l ends up holding the wavefront's total of
//     quantity l >> 1.  One LDS store per even lane, the barrier, and every thread adds the four wavefronts'
//     partials in order.
//   * The serial part.  The gain ratio and the lambda update are computed by every thread (uniform values, no
//     broadcast).  The 6x6 L D L^T solve + exponential map — ~1300 instructions, the longest stretch of a trial —
//     is run by the four wavefronts for FOUR lambdas at once: the lambda of the trial after a rejection is known
//     beforehand (lambda *= ni, ni *= 2), so wavefront w computes the step the w-th trial from now takes if all
//     before it are rejected; the trials then only read their pose from LDS.
//
// Rounds 1-4 summed edge by edge, one lane per sum (17 iterations x 32 us = 544 us for 160 points, slower than
// the CPU oracle's 270 us); this form: see bench.py's dust_alignment leg.  -DSPFE_DUST_PROBE prints cycle counts
// of the phases (map load, edge evaluation, reduce-scatter, barrier + totals, serial part) for workgroup 0.
#include "../../include/spfe_dust_math.h"
#include "spfe_kernels.h"

#ifdef SPFE_DUST_PROBE
#include <cstdio>
#endif

namespace spfe {

namespace {
constexpr int DUST_THREADS = 256;
constexpr int DUST_PPT = DUST_MAX_POINTS / DUST_THREADS;  // points per thread
constexpr int NSUM = SPFE_DUST_NSUM;
constexpr size_t DUST_LDS_FIXED = (2 * 4 * 32 + 4 * 16) * sizeof(double);  // the wavefronts' partial sums, two sets | four candidate steps
static_assert(DUST_THREADS == SPFE_DUST_SLOTS, "the contract's slot is the thread");
static_assert(NSUM <= 32, "reduce-scatter over 32 values");

__device__ __forceinline__ double xchg(double v, int m) { return __shfl_xor(v, m, 64); }
// lane L's value (L a constant) as a wavefront-uniform scalar
__device__ __forceinline__ double bcast(double v, int L) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), L), __builtin_amdgcn_readlane(__double2loint(v), L));
}

// one reduce-scatter level over CNT live values: keep one half, send the other to lane ^ m, add what arrives
template <int CNT>
__device__ __forceinline__ void rs_level(double (&v)[32], int m, bool hi) {
#pragma unroll
  for (int i = 0; i < CNT / 2; ++i) {
    const double keep = hi ? v[i + CNT / 2] : v[i];
    const double send = hi ? v[i] : v[i + CNT / 2];
    v[i] = keep + xchg(send, m);
  }
}

#ifdef SPFE_DUST_PROBE
#define PROBE_T(x) const unsigned long long x = __builtin_readcyclecounter()
#define PROBE_ADD(acc, t0, t1) (acc) += (t1) - (t0)
#else
#define PROBE_T(x)
#define PROBE_ADD(acc, t0, t1)
#endif
}  // namespace

__global__ __launch_bounds__(DUST_THREADS) void dust_align_kernel(DustArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_d[];
  {  // this workgroup's frame: independent solves side by side (the batch path: one record per frame)
    const size_t f = blockIdx.x;
    a.dust = reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.dust) + f * a.dust_stride);
    a.pts = reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.pts) + f * a.pts_stride);
    a.Tcw_in = reinterpret_cast<const float *>(reinterpret_cast<const char *>(a.Tcw_in) + f * a.pose_stride);
    a.Tcw_out = reinterpret_cast<float *>(reinterpret_cast<char *>(a.Tcw_out) + f * a.out_stride);
    a.inlier = a.inlier + f * a.out_stride;
    a.uv = reinterpret_cast<float *>(reinterpret_cast<char *>(a.uv) + f * a.out_stride);
    a.counts = reinterpret_cast<int *>(reinterpret_cast<char *>(a.counts) + f * a.out_stride);
    if (a.n_dev) a.n = min(max(a.n_dev[f], 0), DUST_MAX_POINTS);
  }
  // LDS: partial sums [2 sets][4 wavefronts][32] | candidate steps [4][16] | dust map
  double *s_part = reinterpret_cast<double *>(smem_d);
  double *s_cand = s_part + 2 * 4 * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = a.n, hc = a.hc, wc = a.wc;
  const float *gdust = a.dust;
#ifdef SPFE_DUST_PROBE
  unsigned long long c_load = 0, c_edge = 0, c_rs = 0, c_bar = 0, c_serial = 0, c_bedge = 0, c_brs = 0, c_bbar = 0;
  int n_eval = 0, n_build = 0;
#endif
  PROBE_T(p0);
  // the dust map in LDS when it fits (up to ~40 k cells: 1920x1080 does); larger frames read it where it is, through the L2
  const float *s_dust = gdust;
  if (a.map_in_lds) {
    float *sd = reinterpret_cast<float *>(smem_d + DUST_LDS_FIXED);
    const int cells = hc * wc;
    if ((reinterpret_cast<uintptr_t>(gdust) & 15) == 0) {
      const float4 *g4 = reinterpret_cast<const float4 *>(gdust);
      float4 *s4 = reinterpret_cast<float4 *>(sd);
      for (int i = tid; i < cells / 4; i += DUST_THREADS) s4[i] = g4[i];
      for (int i = (cells & ~3) + tid; i < cells; i += DUST_THREADS) sd[i] = gdust[i];
    } else {
      for (int i = tid; i < cells; i += DUST_THREADS) sd[i] = gdust[i];
    }
    s_dust = sd;
  }

  const double fx = (double)(a.fx / 8.0f), fy = (double)(a.fy / 8.0f);            // optimizer_dust.cpp:223-224
  const double cx = ((double)a.cx - 3.5) / 8.0f, cy = ((double)a.cy - 3.5) / 8.0f;  // :225-226
  const double delta = a.delta;
  double Xw[DUST_PPT][3];
  spfe_dust_edge ed[DUST_PPT];
#pragma unroll
  for (int k = 0; k < DUST_PPT; ++k) {
    const int i = tid + k * DUST_THREADS;
    ed[k].err = 0.0; ed[k].u = 0.0f; ed[k].v = 0.0f; ed[k].level = 0;
    for (int c = 0; c < 3; ++c) Xw[k][c] = i < n ? (double)a.pts[3 * i + c] : 0.0;
  }
  spfe_se3 T;   // every thread carries the pose and the Levenberg state: uniform values
  {
    float Tin[16];
    for (int k = 0; k < 16; ++k) Tin[k] = a.Tcw_in[k];
    spfe_se3_from_f32(Tin, &T);
  }
  __syncthreads();
  PROBE_T(p1);
  PROBE_ADD(c_load, p0, p1);

  int set = 0;
  // computeActiveErrors at Te + activeRobustChi2: one quantity through the contract's tree
  auto errors_and_chi = [&](const spfe_se3 &Te) -> double {
    PROBE_T(e0);
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < DUST_PPT; ++k) {
      const int i = tid + k * DUST_THREADS;
      if (i < n) {
        spfe_dust_error(&Te, Xw[k], fx, fy, cx, cy, s_dust, wc, hc, &ed[k]);
        double rho[3];
        spfe_huber(ed[k].err * ed[k].err, delta, rho);
        v += rho[0];
      }
    }
    PROBE_T(e1);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = v + xchg(v, m);
    double *part = s_part + set * 128;
    if (lane == 0) part[wave * 32] = v;
    PROBE_T(e2);
    __syncthreads();
    const double chi = ((part[0] + part[32]) + part[64]) + part[96];
    set ^= 1;   // the next phase writes the other set: a wavefront still reading this one is not overtaken
    PROBE_T(e3);
#ifdef SPFE_DUST_PROBE
    c_edge += e1 - e0; c_rs += e2 - e1; c_bar += e3 - e2; ++n_eval;
#endif
    return chi;
  };
  // buildSystem at Te (the pose of the last errors_and_chi): linearizeOplus + the contract's sums of H (lower triangle) and b
  auto build = [&](const spfe_se3 &Te, double (&H)[36], double (&b)[6]) {
    PROBE_T(e0);
    double v[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) v[j] = 0.0;
#pragma unroll
    for (int k = 0; k < DUST_PPT; ++k) {
      const int i = tid + k * DUST_THREADS;
      if (i < n) {
        double J[6], q[NSUM];
        spfe_dust_jacobian(&Te, Xw[k], fx, fy, cx, cy, s_dust, wc, hc, ed[k].level, J);
        spfe_dust_terms(ed[k].err, J, delta, q);
#pragma unroll
        for (int j = 1; j < NSUM; ++j) v[j] += q[j];
      }
    }
    PROBE_T(e1);
    rs_level<32>(v, 32, lane & 32);
    rs_level<16>(v, 16, lane & 16);
    rs_level<8>(v, 8, lane & 8);
    rs_level<4>(v, 4, lane & 4);
    rs_level<2>(v, 2, lane & 2);
    v[0] = v[0] + xchg(v[0], 1);
    double *part = s_part + set * 128;
    if (!(lane & 1)) part[wave * 32 + (lane >> 1)] = v[0];
    PROBE_T(e2);
    __syncthreads();
    // lane l adds the four wavefronts' partials of quantity l in order; the totals reach every lane as scalars (readlane)
    const double mine = ((part[lane & 31] + part[32 + (lane & 31)]) + part[64 + (lane & 31)]) + part[96 + (lane & 31)];
    double tot[NSUM], chi_unused;
    tot[0] = 0.0;
#pragma unroll
    for (int j = 1; j < NSUM; ++j) tot[j] = bcast(mine, j);
    spfe_dust_unpack(tot, &chi_unused, H, b);
    set ^= 1;
    PROBE_T(e3);
#ifdef SPFE_DUST_PROBE
    c_bedge += e1 - e0; c_brs += e2 - e1; c_bbar += e3 - e2; ++n_build;
#endif
  };

  spfe_lm lm;
  lm.lambda = 0.0; lm.ni = 2.0;
  int it_done = 0;
  bool fresh = false;        // the edges hold the errors at T, currentChi their chi2 (the last trial was accepted)
  double currentChi = 0.0;
  bool go = a.max_iterations > 0 && n > 0;   // no edges: optimize() has nothing active, the pose is echoed
  for (int it = 0; it < a.max_iterations && go; ++it) {
    // g2o evaluates the errors again at the start of an iteration; after an accepted trial that is the same pose, the same
    // sticky level flags, the same values and the same tree — currentChi already holds those bits
    if (!fresh) currentChi = errors_and_chi(T);
    double H[36], b[6];
    build(T, H, b);
    PROBE_T(s0);
    if (it == 0) {
      double maxDiagonal = 0;
      for (int j = 0; j < 6; ++j) maxDiagonal = fabs(H[j * 6 + j]) > maxDiagonal ? fabs(H[j * 6 + j]) : maxDiagonal;
      lm.lambda = SPFE_LM_TAU * maxDiagonal;
      lm.ni = 2;
    }
    PROBE_T(s1);
    PROBE_ADD(c_serial, s0, s1);
    double rho = 0;
    int qmax = 0;
    do {
      // The trial steps of an iteration share H and b and differ in lambda only, and the lambda of the trial after a
      // rejection is known beforehand (lambda *= ni, ni *= 2: spfe_lm_judge).  The solve and the exponential map are the
      // longest serial stretch of a trial (~1300 instructions), so wavefront w computes the step the w-th trial from now
      // would take if every trial before it is rejected — four candidate poses for the time of one.
      if ((qmax & 3) == 0) {
        PROBE_T(t0);
        double lam = lm.lambda, ni = lm.ni;
#pragma unroll
        for (int r = 0; r < 3; ++r)
          if (r < wave) { lam *= ni; ni *= 2; }
        double xc[6];
        spfe_se3 Tc = T;
        const int okc = spfe_solve6(H, lam, b, xc);
        if (okc) spfe_se3_oplus(&Tc, xc);
        if (lane == 0) {
          double *c = s_cand + wave * 16;
#pragma unroll
          for (int j = 0; j < 6; ++j) c[j] = xc[j];
#pragma unroll
          for (int j = 0; j < 4; ++j) c[6 + j] = Tc.q[j];
#pragma unroll
          for (int j = 0; j < 3; ++j) c[10 + j] = Tc.t[j];
          c[13] = okc ? 1.0 : 0.0;
        }
        __syncthreads();
        PROBE_T(t1);
        PROBE_ADD(c_serial, t0, t1);
      }
      PROBE_T(t4);
      double x[6];
      spfe_se3 Tt;
      const double *c = s_cand + (qmax & 3) * 16;
#pragma unroll
      for (int j = 0; j < 6; ++j) x[j] = c[j];
#pragma unroll
      for (int j = 0; j < 4; ++j) Tt.q[j] = c[6 + j];
#pragma unroll
      for (int j = 0; j < 3; ++j) Tt.t[j] = c[10 + j];
      const bool ok2 = c[13] != 0.0;     // a failed solve leaves the pose where it is: the errors are evaluated there
      PROBE_T(t5);
      PROBE_ADD(c_serial, t4, t5);
      const double chiT = errors_and_chi(Tt);
      PROBE_T(t2);
      const double tempChi = ok2 ? chiT : 1.7976931348623157e308;
      fresh = spfe_lm_judge(&lm, currentChi, tempChi, x, b, &rho) != 0;
      if (fresh) { currentChi = tempChi; T = Tt; }                    // discardTop; otherwise pop: T was never moved
      qmax++;
      PROBE_T(t3);
      PROBE_ADD(c_serial, t2, t3);
    } while (rho < 0 && qmax < SPFE_LM_MAX_TRIALS);
    it_done++;
    if (qmax == SPFE_LM_MAX_TRIALS || rho == 0) go = false;           // Terminate
  }
  // the edges hold the errors of the last evaluation (optimizer_dust.cpp:258-270)
  int my_out = 0;
#pragma unroll
  for (int k = 0; k < DUST_PPT; ++k) {
    const int i = tid + k * DUST_THREADS;
    if (i < n) {
      const int out = ed[k].level == 1 || ed[k].err * ed[k].err > a.inlier_chi2;
      a.inlier[i] = out ? 0 : 1;
      a.uv[2 * i] = ed[k].u;
      a.uv[2 * i + 1] = ed[k].v;
      my_out += out;
    }
  }
  // n_inlier: integer count, order irrelevant
  int *s_cnt = reinterpret_cast<int *>(s_part);
  __syncthreads();
  if (tid == 0) s_cnt[0] = 0;
  __syncthreads();
  if (my_out) atomicAdd(&s_cnt[0], my_out);
  __syncthreads();
  if (tid == 0) {
    float Tout[16];
    spfe_se3_to_f32(&T, Tout);
    for (int k = 0; k < 16; ++k) a.Tcw_out[k] = Tout[k];
    a.counts[0] = n - s_cnt[0];
    a.counts[1] = it_done;
#ifdef SPFE_DUST_PROBE
    if (blockIdx.x == 0)
      printf("dust probe: n %d iterations %d | cycles: load %llu | %d error phases: edges %llu butterfly %llu barrier+total %llu | "
             "%d build phases: edges %llu reduce-scatter %llu barrier+totals %llu | serial %llu\n", n, it_done, c_load, n_eval, c_edge,
             c_rs, c_bar, n_build, c_bedge, c_brs, c_bbar, c_serial);
#endif
  }
}

size_t dust_lds_bytes(int hc, int wc) {
  const size_t with_map = DUST_LDS_FIXED + (size_t)hc * wc * sizeof(float);
  return with_map <= 160 * 1024 ? with_map : DUST_LDS_FIXED;
}

hipError_t launch_dust_align(const DustArgs &a0, hipStream_t s) {
  DustArgs a = a0;
  if (a.n < 0 || a.n > DUST_MAX_POINTS) return hipErrorInvalidValue;
  const size_t lds = dust_lds_bytes(a.hc, a.wc);
  a.map_in_lds = lds > DUST_LDS_FIXED ? 1 : 0;
  static bool attr_done[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(dust_align_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) attr_done[dev] = true;
  }
  hipLaunchKernelGGL(dust_align_kernel, dim3(a.nframes > 0 ? a.nframes : 1), dim3(DUST_THREADS), lds, s, a);
  return hipGetLastError();
}

}  // namespace spfe
