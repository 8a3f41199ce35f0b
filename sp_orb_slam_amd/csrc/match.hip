// match.hip — brute-force L2 descriptor matching with OpenCV's cross-check rule, on the GPU.
//
// the step right after extraction.  Replaces
//   cv::BFMatcher::create(cv::NORM_L2, /*crossCheck=*/true)->match(desc_query, matches)
// as called by SPMatcher::SearchByBruteForce
// (/root/reference/orb_slam2/src/cv/sp_matcher.cpp:1642-1674; distance =
// SPMatcher::DescriptorDistance :1636-1640 = L2 norm of the difference), so that two frames'
// descriptors can be matched while their records are still in HBM.
//
// Arithmetic contract (oracle_match_bruteforce restates the same):
//   dist(a, b) = sqrtf(s_255),  s_k = fmaf(a[k] - b[k], a[k] - b[k], s_{k-1}),  s_{-1} = 0
// — a sequential chain in k, which is what a thread that owns a (row, column) pair
// computes anyway; no |a|^2 + |b|^2 - 2ab expansion (that is not the reference's arithmetic
// and loses the small distances).  "Nearest" = smallest dist, lowest index on ties
// (OpenCV's strict `<` scans).
//
//   match_nn_kernel      rows x cols distance tiles (64 x 64 per workgroup, 4 x 4 per lane,
//                        operands staged through LDS 64 dimensions at a time); every row's
//                        nearest column is folded with a 64-bit atomic min of
//                        (dist bits << 32 | column) — dist >= 0, so the bit pattern orders
//                        like the value and the low word breaks ties toward the lower index.
//   match_resolve_kernel cross-check as OpenCV's batchDistance does it: rows = train,
//                        cols = query; each train row votes for its nearest query, each
//                        query keeps the closest train that voted for it (lowest train index
//                        on ties) — again one 64-bit atomic min.
//   match_emit_kernel    unpack to int32 train index (-1 = no match) + float distance.
#include <float.h>

#include "spfe_kernels.h"

namespace spfe {

namespace {
constexpr int M_TILE = 64;        // rows / columns per workgroup
constexpr int M_KC = 64;          // descriptor dimensions staged per pass
constexpr int M_PITCH = M_KC + 4; // floats per staged row: 272 B keeps float4 alignment, spreads banks
constexpr int M_DIM = 256;
constexpr unsigned long long M_NONE = ~0ull;

__device__ __forceinline__ int side_count(const MatchSide &s, int pair) {
  const int n = *reinterpret_cast<const int *>(s.base + (size_t)pair * s.stride + s.off_cnt);
  return n < 0 ? 0 : (n > s.cap ? s.cap : n);
}
__device__ __forceinline__ const float *side_desc(const MatchSide &s, int pair) {
  return reinterpret_cast<const float *>(s.base + (size_t)pair * s.stride + s.off_desc);
}
// four consecutive descriptor elements from element index e of a row block: f32 rows, or bf16 rows widened (exact)
__device__ __forceinline__ float4 desc4(const float *rows, size_t e, int bf16) {
  if (!bf16) return *reinterpret_cast<const float4 *>(rows + e);
  const uint2 p = *reinterpret_cast<const uint2 *>(reinterpret_cast<const unsigned short *>(rows) + e);
  return make_float4(__uint_as_float(p.x << 16), __uint_as_float(p.x & 0xffff0000u), __uint_as_float(p.y << 16),
                     __uint_as_float(p.y & 0xffff0000u));
}
}  // namespace

// excl (may be null): per row, only candidates with a (dist, column) key ABOVE excl[row] compete — with excl =
// the nearest neighbours of a first pass this finds the second nearest (k = 2 of cv::DescriptorMatcher::knnMatch).
__global__ __launch_bounds__(256) void match_nn_kernel(MatchSide rows, MatchSide cols,
                                                       unsigned long long *__restrict__ best,
                                                       const unsigned long long *__restrict__ excl) {
  const int pair = blockIdx.z;
  const int nr = side_count(rows, pair), nc = side_count(cols, pair);
  const int r0 = blockIdx.y * M_TILE, c0 = blockIdx.x * M_TILE;
  if (r0 >= nr || c0 >= nc) return;
  const float *R = side_desc(rows, pair), *Cd = side_desc(cols, pair);

  __shared__ __attribute__((aligned(16))) float sR[M_TILE * M_PITCH];
  __shared__ __attribute__((aligned(16))) float sC[M_TILE * M_PITCH];
  const int tid = threadIdx.x;
  const int ty = tid >> 4, tx = tid & 15;

  // two columns per packed register: v_pk_add_f32 / v_pk_fma_f32 do two IEEE f32 lanes per
  // issue slot (bitwise the scalar ops), which is what this VALU-bound loop is made of
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 acc2[4][2];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 2; ++c) acc2[r][c] = (f32x2){0.0f, 0.0f};

  for (int k0 = 0; k0 < M_DIM; k0 += M_KC) {
    if (k0) __syncthreads();
#pragma unroll
    for (int i = tid; i < M_TILE * (M_KC / 4); i += 256) {
      const int row = i >> 4, k4 = i & 15;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f), u = v;
      if (r0 + row < nr) v = desc4(R, (size_t)(r0 + row) * M_DIM + k0 + k4 * 4, rows.desc_bf16);
      if (c0 + row < nc) u = desc4(Cd, (size_t)(c0 + row) * M_DIM + k0 + k4 * 4, cols.desc_bf16);
      *reinterpret_cast<float4 *>(&sR[row * M_PITCH + k4 * 4]) = v;
      *reinterpret_cast<float4 *>(&sC[row * M_PITCH + k4 * 4]) = u;
    }
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < M_KC; k += 4) {
      float4 a[4], b[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) a[r] = *reinterpret_cast<const float4 *>(&sR[(ty * 4 + r) * M_PITCH + k]);
#pragma unroll
      for (int c = 0; c < 4; ++c) b[c] = *reinterpret_cast<const float4 *>(&sC[(tx + 16 * c) * M_PITCH + k]);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          f32x2 d;
          d = (f32x2){a[r].x, a[r].x} - (f32x2){b[2 * c].x, b[2 * c + 1].x};
          acc2[r][c] = __builtin_elementwise_fma(d, d, acc2[r][c]);
          d = (f32x2){a[r].y, a[r].y} - (f32x2){b[2 * c].y, b[2 * c + 1].y};
          acc2[r][c] = __builtin_elementwise_fma(d, d, acc2[r][c]);
          d = (f32x2){a[r].z, a[r].z} - (f32x2){b[2 * c].z, b[2 * c + 1].z};
          acc2[r][c] = __builtin_elementwise_fma(d, d, acc2[r][c]);
          d = (f32x2){a[r].w, a[r].w} - (f32x2){b[2 * c].w, b[2 * c + 1].w};
          acc2[r][c] = __builtin_elementwise_fma(d, d, acc2[r][c]);
        }
    }
  }
  float acc[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[r][c] = acc2[r][c >> 1][c & 1];

#pragma unroll
  for (int r = 0; r < 4; ++r) {
    unsigned long long p = M_NONE;
    const int row_r = r0 + ty * 4 + r;
    const unsigned long long floor_key = (excl && row_r < nr) ? excl[(size_t)pair * rows.cap + row_r] : 0ull;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int col = c0 + tx + 16 * c;
      const float dist = __builtin_sqrtf(acc[r][c]);  // correctly rounded (v_sqrt_f32 + the fma fix-up), unlike __fsqrt_rn
      if (col < nc && dist < FLT_MAX) {  // NaN / inf distances are never "nearer" (OpenCV: d < FLT_MAX start)
        const unsigned long long cand = ((unsigned long long)__float_as_uint(dist) << 32) | (unsigned)col;
        if (!excl || (floor_key != M_NONE && cand > floor_key)) p = cand < p ? cand : p;
      }
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) {  // the 16 lanes sharing this row
      const unsigned lo = __shfl_xor((unsigned)p, off), hi = __shfl_xor((unsigned)(p >> 32), off);
      const unsigned long long o = ((unsigned long long)hi << 32) | lo;
      p = o < p ? o : p;
    }
    const int row = r0 + ty * 4 + r;
    if (tx == 0 && row < nr && p != M_NONE) atomicMin(&best[(size_t)pair * rows.cap + row], p);
  }
}

// best_t: [pairs][cap_t] (dist, query) per train row; best_q: [pairs][cap_q] (dist, train) per query
__global__ __launch_bounds__(256) void match_resolve_kernel(const unsigned long long *__restrict__ best_t,
                                                            int cap_t, unsigned long long *__restrict__ best_q,
                                                            int cap_q) {
  const int t = blockIdx.x * 256 + threadIdx.x, pair = blockIdx.y;
  if (t >= cap_t) return;
  const unsigned long long p = best_t[(size_t)pair * cap_t + t];
  if (p == M_NONE) return;
  const unsigned q = (unsigned)p;
  atomicMin(&best_q[(size_t)pair * cap_q + q], (p & 0xffffffff00000000ull) | (unsigned)t);
}

__global__ __launch_bounds__(256) void match_emit_kernel(const unsigned long long *__restrict__ best, int cap,
                                                         uint8_t *__restrict__ out, size_t out_stride) {
  const int q = blockIdx.x * 256 + threadIdx.x, pair = blockIdx.y;
  if (q >= cap) return;
  const unsigned long long p = best[(size_t)pair * cap + q];
  int32_t *idx = reinterpret_cast<int32_t *>(out + (size_t)pair * out_stride);
  float *dist = reinterpret_cast<float *>(out + (size_t)pair * out_stride + (size_t)cap * 4);
  idx[q] = p == M_NONE ? -1 : (int32_t)(unsigned)p;
  dist[q] = p == M_NONE ? FLT_MAX : __uint_as_float((unsigned)(p >> 32));
}

hipError_t launch_match(const MatchSide &query, const MatchSide &train, int pairs, bool cross_check,
                        unsigned long long *best_t, unsigned long long *best_q, uint8_t *out, size_t out_stride,
                        hipStream_t s) {
  hipError_t e;
  if ((e = hipMemsetAsync(best_q, 0xff, (size_t)pairs * query.cap * 8, s)) != hipSuccess) return e;
  if (cross_check) {
    if ((e = hipMemsetAsync(best_t, 0xff, (size_t)pairs * train.cap * 8, s)) != hipSuccess) return e;
    dim3 g((query.cap + M_TILE - 1) / M_TILE, (train.cap + M_TILE - 1) / M_TILE, pairs);
    hipLaunchKernelGGL(match_nn_kernel, g, dim3(256), 0, s, train, query, best_t, (const unsigned long long *)nullptr);
    hipLaunchKernelGGL(match_resolve_kernel, dim3((train.cap + 255) / 256, pairs), dim3(256), 0, s, best_t,
                       train.cap, best_q, query.cap);
  } else {
    dim3 g((train.cap + M_TILE - 1) / M_TILE, (query.cap + M_TILE - 1) / M_TILE, pairs);
    hipLaunchKernelGGL(match_nn_kernel, g, dim3(256), 0, s, query, train, best_q, (const unsigned long long *)nullptr);
  }
  hipLaunchKernelGGL(match_emit_kernel, dim3((query.cap + 255) / 256, pairs), dim3(256), 0, s, best_q, query.cap,
                     out, out_stride);
  return hipGetLastError();
}

// k = 2 nearest train rows per query (no cross-check): two passes of the distance kernel, the second restricted
// to keys above the first's.  out: [first: idx[cap] | dist[cap]] [second: idx[cap] | dist[cap]] per pair.
// best1 / best2: [pairs][query.cap] scratch.
hipError_t launch_match_knn2(const MatchSide &query, const MatchSide &train, int pairs, unsigned long long *best1,
                             unsigned long long *best2, uint8_t *out, size_t out_stride, hipStream_t s) {
  hipError_t e;
  if ((e = hipMemsetAsync(best1, 0xff, (size_t)pairs * query.cap * 8, s)) != hipSuccess) return e;
  if ((e = hipMemsetAsync(best2, 0xff, (size_t)pairs * query.cap * 8, s)) != hipSuccess) return e;
  dim3 g((train.cap + M_TILE - 1) / M_TILE, (query.cap + M_TILE - 1) / M_TILE, pairs);
  hipLaunchKernelGGL(match_nn_kernel, g, dim3(256), 0, s, query, train, best1, (const unsigned long long *)nullptr);
  hipLaunchKernelGGL(match_nn_kernel, g, dim3(256), 0, s, query, train, best2, (const unsigned long long *)best1);
  const dim3 ge((query.cap + 255) / 256, pairs);
  hipLaunchKernelGGL(match_emit_kernel, ge, dim3(256), 0, s, best1, query.cap, out, out_stride);
  hipLaunchKernelGGL(match_emit_kernel, ge, dim3(256), 0, s, best2, query.cap, out + (size_t)query.cap * 8, out_stride);
  return hipGetLastError();
}

}  // namespace spfe

// ---------------------------------------------------------------------------
// Patch-wise association of projected map points (tracker_dust.cpp:113-172): every map point looks at
// the 2 x 2 cells at its projected dust-map position, takes the keypoint (one per cell, occ_grid) whose
// descriptor is nearest and nearer than max_dist (0.75), and REMOVES it from the grid — the map points
// are served in order, so an earlier one can take a later one's best candidate.
//
//   patch_dist_kernel     one wave per map point: the (up to) four candidate keypoints and their
//                         distances.  dist = (float)sqrt(S), S = sum of (double)(a_k - b_k)^2 (cv::norm
//                         accumulates in double) — lane l adds its dims 4l..4l+3 in order, then the
//                         64-lane butterfly, the order oracle_match_patches restates.
//   patch_resolve_kernel  the greedy order as a fixed point, one workgroup: a map point is final as
//                         soon as it is the earliest unresolved claimant of every candidate it could
//                         still take; it then takes its nearest free candidate (first one on ties,
//                         cells in the reference's (du, dv) loop order).
// ---------------------------------------------------------------------------
namespace spfe {

__global__ __launch_bounds__(256) void patch_dist_kernel(PatchArgs a, int *__restrict__ cand_idx,
                                                         float *__restrict__ cand_dist) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= a.n_points) return;
  const float uf = a.mp_uv[2 * i], vf = a.mp_uv[2 * i + 1];
  const float fu = __builtin_floorf(uf), fv = __builtin_floorf(vf);
  // positions that do not floor into the grid have no candidates (the reference does not check)
  bool ok = fu >= 0.0f && fv >= 0.0f && fu < (float)a.wc && fv < (float)a.hc;
  if (a.in_view && !a.in_view[i]) ok = false;                 // `if (!mp->in_view ...) continue;`
  if (a.gate_ptr && *a.gate_ptr < a.gate_min) ok = false;     // `if (n_inlier < th_ninlier) return false;`
  const int u = ok ? (int)fu : 0, v = ok ? (int)fv : 0;
  const int K = a.k_ptr ? *a.k_ptr : a.k_imm;
  const float4 m4 = *reinterpret_cast<const float4 *>(a.mp_desc + (size_t)i * 256 + lane * 4);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int uu = u + (c >> 1), vv = v + (c & 1);  // du outer, dv inner (tracker_dust.cpp:126-127)
    int idx = -1;
    if (ok && uu < a.wc && vv < a.hc) idx = a.occ[vv * a.wc + uu];
    if (idx >= K) idx = -1;
    float dist = 3.0e38f;
    if (idx >= 0) {  // wave-uniform
      const float4 k4 = desc4(a.kp_desc, (size_t)idx * 256 + lane * 4, a.kp_desc_bf16);
      const float d0 = m4.x - k4.x, d1 = m4.y - k4.y, d2 = m4.z - k4.z, d3 = m4.w - k4.w;
      double s = (double)d0 * (double)d0;
      s = s + (double)d1 * (double)d1;
      s = s + (double)d2 * (double)d2;
      s = s + (double)d3 * (double)d3;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) s = s + __shfl_xor(s, off, 64);
      dist = (float)__builtin_sqrt(s);
    }
    if (lane == 0) {
      cand_idx[i * 4 + c] = idx;
      cand_dist[i * 4 + c] = dist;
    }
  }
}

__global__ __launch_bounds__(1024) void patch_resolve_kernel(const int *__restrict__ cand_idx,
                                                             const float *__restrict__ cand_dist, int n_points,
                                                             int kcap, float max_dist, int32_t *__restrict__ out) {
  extern __shared__ int sm_p[];
  int *claim = sm_p;                                             // [kcap] earliest unresolved claimant
  uint8_t *taken = reinterpret_cast<uint8_t *>(claim + kcap);    // [kcap]
  __shared__ int pending;
  const int tid = threadIdx.x;
  for (int k = tid; k < kcap; k += 1024) taken[k] = 0;
  constexpr int PER = 4;  // map points per thread (n_points <= 4096)
  int ci[PER][4];
  float cd[PER][4];
  bool done[PER];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int i = tid + q * 1024;
    done[q] = i >= n_points;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      ci[q][c] = -1;
      cd[q][c] = 3.0e38f;
      if (i < n_points) {
        const int idx = cand_idx[i * 4 + c];
        const float d = cand_dist[i * 4 + c];
        if (idx >= 0 && idx < kcap && d < max_dist) { ci[q][c] = idx; cd[q][c] = d; }  // others can never match
      }
    }
  }
  for (int round = 0; round < 8192; ++round) {
    for (int k = tid; k < kcap; k += 1024) claim[k] = 0x7fffffff;
    if (tid == 0) pending = 0;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PER; ++q)
      if (!done[q])
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (ci[q][c] >= 0 && !taken[ci[q][c]]) atomicMin(&claim[ci[q][c]], tid + q * 1024);
    __syncthreads();
    bool fin[PER];
    int pick[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      fin[q] = false;
      pick[q] = -1;
      if (done[q]) continue;
      const int i = tid + q * 1024;
      bool first = true;
      float bd = max_dist;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int k = ci[q][c];
        if (k < 0 || taken[k]) continue;
        first &= claim[k] == i;
        if (cd[q][c] < bd) { bd = cd[q][c]; pick[q] = k; }
      }
      fin[q] = first;
    }
    __syncthreads();  // every decision read `taken` before anybody writes it
    bool mine = false;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      if (done[q]) continue;
      if (fin[q]) {
        out[tid + q * 1024] = pick[q];
        if (pick[q] >= 0) taken[pick[q]] = 1;
        done[q] = true;
      } else {
        mine = true;
      }
    }
    if (mine) pending = 1;
    __syncthreads();
    if (!pending) break;
    __syncthreads();
  }
}

hipError_t launch_match_patches(const PatchArgs &a, int kcap, float max_dist, int *cand_idx, float *cand_dist,
                                int32_t *out, hipStream_t s) {
  if (a.n_points <= 0) return hipSuccess;
  if (a.n_points > 4096) return hipErrorInvalidValue;
  hipLaunchKernelGGL(patch_dist_kernel, dim3((a.n_points + 3) / 4), dim3(256), 0, s, a, cand_idx, cand_dist);
  const size_t lds = (size_t)kcap * 5 + 16;
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  if (lds > 48 * 1024) {   // beyond the default dynamic-LDS limit (kcap > ~9800): raise it
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(patch_resolve_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(patch_resolve_kernel, dim3(1), dim3(1024), lds, s, cand_idx, cand_dist, a.n_points, kcap,
                     max_dist, out);
  return hipGetLastError();
}

}  // namespace spfe
