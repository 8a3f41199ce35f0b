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
// The problem is tiny and sequential (40 outer iterations x up to 10 trial steps, each needing a global
// sum): ONE workgroup.  Per phase, a thread owns a map point (errors, Jacobians: the double-precision
// formulas of include/spfe_dust_math.h, shared with the CPU oracle); the sums that g2o forms edge by edge
// — chi2 and the 21 + 6 entries of the pose block's normal equations — are accumulated IN THE SAME ORDER by
// one lane each (28 lanes in parallel, the per-edge terms staged in LDS), so the accept / reject decisions of
// the Levenberg loop see the same bits as the sequential CPU statement (up to the device's sin / cos / sqrt
// in the exponential map); lane 0 runs the 6x6 solve and the lambda logic.  Latency, not throughput: the
// whole solve is a few hundred microseconds with the dust map in LDS.
#include "../../include/spfe_dust_math.h"
#include "spfe_kernels.h"

namespace spfe {

namespace {
constexpr int DUST_THREADS = 256;
constexpr int DUST_PPT = DUST_MAX_POINTS / DUST_THREADS;  // points per thread

struct Shared {
  spfe_se3 T, saved;
  double H[36], b[6], x[6];
  double currentChi, tempChi, rho;
  spfe_lm lm;
  int ok2, accept, cont_trials, cont_iters, qmax, it_done;
};
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
  // LDS: control block | per-point err, rho0, w (rho1) | per-point J[6] | dust map
  Shared *sh = reinterpret_cast<Shared *>(smem_d);
  double *s_err = reinterpret_cast<double *>(smem_d + 1024);
  double *s_rho0 = s_err + DUST_MAX_POINTS;
  double *s_w = s_rho0 + DUST_MAX_POINTS;
  double *s_J = s_w + DUST_MAX_POINTS;                       // [n][6]
  const int tid = threadIdx.x, n = a.n, hc = a.hc, wc = a.wc;
  const float *gdust = a.dust;
  // the dust map in LDS when it fits beside the per-point arrays (up to ~31 k cells); larger frames (1920x1080: 32,400) read it
  // where it is, through the L2
  const float *s_dust = gdust;
  if (a.map_in_lds) {
    float *sd = reinterpret_cast<float *>(s_J + 6 * DUST_MAX_POINTS);
    for (int i = tid; i < hc * wc; i += DUST_THREADS) sd[i] = gdust[i];
    s_dust = sd;
  }

  const double fx = (double)(a.fx / 8.0f), fy = (double)(a.fy / 8.0f);            // optimizer_dust.cpp:223-224
  const double cx = ((double)a.cx - 3.5) / 8.0f, cy = ((double)a.cy - 3.5) / 8.0f;  // :225-226
  double Xw[DUST_PPT][3];
  spfe_dust_edge ed[DUST_PPT];
#pragma unroll
  for (int k = 0; k < DUST_PPT; ++k) {
    const int i = tid + k * DUST_THREADS;
    ed[k].err = 0.0; ed[k].u = 0.0f; ed[k].v = 0.0f; ed[k].level = 0;
    for (int c = 0; c < 3; ++c) Xw[k][c] = i < n ? (double)a.pts[3 * i + c] : 0.0;
  }
  if (tid == 0) {
    float Tin[16];
    for (int k = 0; k < 16; ++k) Tin[k] = a.Tcw_in[k];
    spfe_se3_from_f32(Tin, &sh->T);
    sh->lm.lambda = 0.0; sh->lm.ni = 2.0;
    sh->it_done = 0;
    sh->cont_iters = a.max_iterations > 0 && n > 0;   // no edges: optimize() has nothing active, the pose is echoed
  }
  __syncthreads();

  // computeActiveErrors + the per-edge Huber terms, then activeRobustChi2 summed in edge order by lane 0
  auto errors_and_chi = [&](double *chi_out) {
    const spfe_se3 T = sh->T;
#pragma unroll
    for (int k = 0; k < DUST_PPT; ++k) {
      const int i = tid + k * DUST_THREADS;
      if (i < n) {
        spfe_dust_error(&T, Xw[k], fx, fy, cx, cy, s_dust, wc, hc, &ed[k]);
        double rho[3];
        spfe_huber(ed[k].err * ed[k].err, a.delta, rho);
        s_err[i] = ed[k].err; s_rho0[i] = rho[0]; s_w[i] = rho[1];
      }
    }
    __syncthreads();
    if (tid == 0) {   // (a lane-order readlane fold was tried: SGPR round trips make it slower than these pipelined LDS reads)
      double chi = 0.0;
      for (int i = 0; i < n; ++i) chi += s_rho0[i];
      *chi_out = chi;
    }
    __syncthreads();
  };

  for (int it = 0; it < a.max_iterations; ++it) {
    if (!sh->cont_iters) break;          // uniform: written before the last barrier
    errors_and_chi(&sh->currentChi);
    // buildSystem: linearizeOplus per edge, then the quadratic form summed in edge order, one lane per entry
    {
      const spfe_se3 T = sh->T;
#pragma unroll
      for (int k = 0; k < DUST_PPT; ++k) {
        const int i = tid + k * DUST_THREADS;
        if (i < n) {
          double J[6];
          spfe_dust_jacobian(&T, Xw[k], fx, fy, cx, cy, s_dust, wc, hc, ed[k].level, J);
          for (int c = 0; c < 6; ++c) s_J[i * 6 + c] = J[c];
        }
      }
    }
    __syncthreads();
    if (tid < 36) {
      const int j = tid / 6, k = tid % 6;
      double acc = 0.0;
      for (int i = 0; i < n; ++i) acc += (s_J[i * 6 + j] * s_w[i]) * s_J[i * 6 + k];
      sh->H[tid] = acc;
    } else if (tid >= 64 && tid < 70) {
      const int j = tid - 64;
      double acc = 0.0;
      for (int i = 0; i < n; ++i) acc -= (s_w[i] * s_J[i * 6 + j]) * s_err[i];
      sh->b[j] = acc;
    }
    __syncthreads();
    if (tid == 0) {
      if (it == 0) {
        double maxDiagonal = 0;
        for (int j = 0; j < 6; ++j) maxDiagonal = fabs(sh->H[j * 6 + j]) > maxDiagonal ? fabs(sh->H[j * 6 + j]) : maxDiagonal;
        sh->lm.lambda = SPFE_LM_TAU * maxDiagonal;
        sh->lm.ni = 2;
      }
      sh->qmax = 0;
      sh->cont_trials = 1;
    }
    __syncthreads();
    while (sh->cont_trials) {
      if (tid == 0) {
        sh->saved = sh->T;                                            // push
        for (int j = 0; j < 6; ++j) sh->x[j] = 0.0;
        sh->ok2 = spfe_solve6(sh->H, sh->lm.lambda, sh->b, sh->x);
        if (sh->ok2) spfe_se3_oplus(&sh->T, sh->x);
      }
      __syncthreads();
      errors_and_chi(&sh->tempChi);
      if (tid == 0) {
        double tempChi = sh->ok2 ? sh->tempChi : 1.7976931348623157e308;
        double rho;
        if (spfe_lm_judge(&sh->lm, sh->currentChi, tempChi, sh->x, sh->b, &rho)) sh->currentChi = tempChi;
        else sh->T = sh->saved;                                       // pop
        sh->rho = rho;
        sh->qmax++;
        sh->cont_trials = (rho < 0 && sh->qmax < SPFE_LM_MAX_TRIALS) ? 1 : 0;
        if (!sh->cont_trials) {
          sh->it_done++;
          if (sh->qmax == SPFE_LM_MAX_TRIALS || rho == 0) sh->cont_iters = 0;   // Terminate
        }
      }
      __syncthreads();
    }
  }
  __syncthreads();
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
  int *s_cnt = reinterpret_cast<int *>(s_err);
  __syncthreads();
  if (tid == 0) s_cnt[0] = 0;
  __syncthreads();
  if (my_out) atomicAdd(&s_cnt[0], my_out);
  __syncthreads();
  if (tid == 0) {
    float Tout[16];
    spfe_se3_to_f32(&sh->T, Tout);
    for (int k = 0; k < 16; ++k) a.Tcw_out[k] = Tout[k];
    a.counts[0] = n - s_cnt[0];
    a.counts[1] = sh->it_done;
  }
}

size_t dust_lds_bytes(int hc, int wc) {
  const size_t fixed = 1024 + (size_t)DUST_MAX_POINTS * 9 * sizeof(double), with_map = fixed + (size_t)hc * wc * sizeof(float);
  return with_map <= 160 * 1024 ? with_map : fixed;
}

hipError_t launch_dust_align(const DustArgs &a0, hipStream_t s) {
  DustArgs a = a0;
  if (a.n < 0 || a.n > DUST_MAX_POINTS) return hipErrorInvalidValue;
  static_assert(sizeof(Shared) <= 1024, "control block");
  const size_t lds = dust_lds_bytes(a.hc, a.wc);
  a.map_in_lds = lds > 1024 + (size_t)DUST_MAX_POINTS * 9 * sizeof(double) ? 1 : 0;
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
