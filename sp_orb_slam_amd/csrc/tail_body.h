// tail_body.h — the detector tail of ONE 8x8 cell (sp_extractor.cpp:105-131), shared by tail_kernel (tail_select.hip: logits
// from HBM) and pbtail_f32_kernel (pbtail_f32.hip: logits straight from convPb's accumulators).  Float steps follow
// include/spfe_exact_math.h: every integer decision is bit-identical to the CPU oracle given identical logits.
#pragma once
#include "spfe_kernels.h"
#include "../../include/spfe_exact_math.h"

namespace spfe {

#define TAIL_CELLS_PER_WG 32   // cells per workgroup of either kernel = one min/max partial (tail_parts)

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_max64(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float o = __shfl_xor(v, off, 64);
    v = o > v ? o : v;
  }
  return v;
}
__device__ __forceinline__ float wave_min64(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float o = __shfl_xor(v, off, 64);
    v = o < v ? o : v;
  }
  return v;
}

// Four lanes (a DPP quad) per 8x8 cell, 16 cells per wave: lane q of the quad owns position
// channels 16q .. 16q+15 (= pixel rows 2q, 2q+1 of the cell) in registers, so the soft-max, the
// arg-max and the log-heat need two quad exchanges per reduction instead of the six dependent
// ds_bpermute round trips per reduction of a wave-per-cell form.
// The sum of the 64 exponentials follows the butterfly of spfe_sum64_host level by level — pairs
// 32 apart (quad lane ^ 2), 16 apart (quad lane ^ 1), then 8, 4, 2, 1 inside the lane — so it is
// bit-identical.
__device__ __forceinline__ float quad_xor1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
}
__device__ __forceinline__ float quad_xor2(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
}
__device__ __forceinline__ int quad_xor1i(int v) { return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true); }
__device__ __forceinline__ int quad_xor2i(int v) { return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true); }

// One cell, run by the four lanes of its quad (q = lane & 3; whole quads are active or idle together: DPP reads inactive
// lanes as 0 with bound_ctrl).  row: the cell's 65 logits (LDS).  Writes the cell's 8x8 block of the pixel-shuffled log-heat,
// its dust values, score and arg-max; folds the block's log-heat into lmin / lmax.
__device__ __forceinline__ void tail_cell(const float *row, int q, int cell, int wc, int W, float *heat_log, float *semi_dust,
                                          float *dense_dust, float *cell_score, uint8_t *cell_k, float &lmin, float &lmax) {
  const int cy = cell / wc, cx = cell - cy * wc;
  float v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = row[q * 16 + k];
  const float vd = row[64];
  float m = vd;
#pragma unroll
  for (int k = 0; k < 16; ++k) m = v[k] > m ? v[k] : m;
  { const float o = quad_xor2(m); m = o > m ? o : m; }
  { const float o = quad_xor1(m); m = o > m ? o : m; }
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = spfe_expf(v[k] - m);
  const float ed = spfe_expf(vd - m);
  float t[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) t[k] = v[k] + quad_xor2(v[k]);   // channels 32 apart
#pragma unroll
  for (int k = 0; k < 16; ++k) t[k] = t[k] + quad_xor1(t[k]);   // 16 apart
#pragma unroll
  for (int j = 0; j < 8; ++j) t[j] = t[j] + t[j + 8];
#pragma unroll
  for (int j = 0; j < 4; ++j) t[j] = t[j] + t[j + 4];
  t[0] = t[0] + t[2];
  t[1] = t[1] + t[3];
  const float total = (t[0] + t[1]) + ed;
  // arg-max over the 64 position channels, lowest index on ties (:112)
  float bv = -1.0f;
  int bi = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    v[k] = v[k] / total;
    if (v[k] > bv) { bv = v[k]; bi = q * 16 + k; }
  }
  {
    const float ov = quad_xor2(bv);
    const int oi = quad_xor2i(bi);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  {
    const float ov = quad_xor1(bv);
    const int oi = quad_xor1i(bi);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  float *hl = heat_log + (size_t)(cy * 8 + 2 * q) * W + cx * 8;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    float L[8];
#pragma unroll
    for (int dx = 0; dx < 8; ++dx) {
      const float p = v[r * 8 + dx];
      L[dx] = spfe_logf(p < SPFE_HEAT_FLOOR ? SPFE_HEAT_FLOOR : p);
      lmin = L[dx] < lmin ? L[dx] : lmin;
      lmax = L[dx] > lmax ? L[dx] : lmax;
    }
    *reinterpret_cast<float4 *>(hl + (size_t)r * W) = make_float4(L[0], L[1], L[2], L[3]);
    *reinterpret_cast<float4 *>(hl + (size_t)r * W + 4) = make_float4(L[4], L[5], L[6], L[7]);
  }
  if (q == 0) {
    semi_dust[cell] = vd;
    dense_dust[cell] = ed / total;
    cell_score[cell] = bv >= SPFE_SCORE_THRESH ? bv : 0.0f;
    cell_k[cell] = (uint8_t)bi;
  }
}

}  // namespace spfe
