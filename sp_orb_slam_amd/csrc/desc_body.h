// desc_body.h — descriptor sampling of ONE emitted keypoint by one wavefront (sp_extractor.cpp:102-103, :134-148), shared by
// desc_kernel (tail_select.hip) and the covariance replay launch (cov.hip), which carries the sampling as extra wavefronts in
// synchronous calls: it is needed by the finished record only, not by the covariance chain.
#pragma once
#include "spfe_kernels.h"
#include "../../include/spfe_exact_math.h"

namespace spfe {

__device__ __forceinline__ float desc_wave_sum64(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off, 64);
  return v;
}

// ---------------------------------------------------------------------------
// Descriptors for the emitted keypoints: one wavefront per keypoint, lane l owns
// channels 4l..4l+3 (one float4 of the NHWC coarse map per tap: 1 KiB coalesced).
// ---------------------------------------------------------------------------
__device__ __forceinline__ float sum256_wave(float4 sq) {
  float s = sq.x;
  s = s + sq.y;
  s = s + sq.z;
  s = s + sq.w;
  return desc_wave_sum64(s);
}

// The bilinear taps of a keypoint at (x, y) on the coarse map: the top-left cell (x0, y0) — may be -1 / past the last
// row or column: those taps are the zero padding — and the weights.  ONE definition for the sampling below and for the
// selection kernel's list of the cells the descriptor head has to compute (tail_select.hip, "sparse convDb").
struct DescTaps {
  int x0, y0;
  float tw[4];
};
__device__ __forceinline__ DescTaps desc_taps(float x, float y, int H, int W) {
  const int wc = W >> 3, hc = H >> 3;
  // :137-138 with ATen-CUDA scalar division (x * float(1/(w/2))), then the
  // align_corners un-normalisation of grid_sampler
  const float inv_hw = (float)(1.0 / (double)(float)(W / 2.0));
  const float inv_hh = (float)(1.0 / (double)(float)(H / 2.0));
  const float gx = x * inv_hw - 1.0f;
  const float gy = y * inv_hh - 1.0f;
  const float ix = ((gx + 1.0f) / 2.0f) * (float)(wc - 1);
  const float iy = ((gy + 1.0f) / 2.0f) * (float)(hc - 1);
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  DescTaps t;
  t.x0 = (int)fx0; t.y0 = (int)fy0;
  const float wx1 = ix - fx0, wy1 = iy - fy0;
  const float wx0 = (fx0 + 1.0f) - ix, wy0 = (fy0 + 1.0f) - iy;
  t.tw[0] = wx0 * wy0; t.tw[1] = wx1 * wy0; t.tw[2] = wx0 * wy1; t.tw[3] = wx1 * wy1;
  return t;
}

__device__ __forceinline__ void desc_keypoint(const FrameBufs &f, const RecordLayout &rl, int H, int W, int b, int i, int lane) {
  const int wc = W >> 3, hc = H >> 3, C = hc * wc;
  uint8_t *rec = f.records + (size_t)b * rl.bytes;
  const int K = reinterpret_cast<const int *>(rec + rl.off_hdr)[0];
  if (i >= K) return;
  const float *kp_xy = reinterpret_cast<const float *>(rec + rl.off_xy);
  const float x = kp_xy[2 * i], y = kp_xy[2 * i + 1];
  const DescTaps taps = desc_taps(x, y, H, W);
  const int x0 = taps.x0, y0 = taps.y0;
  const float (&tw)[4] = taps.tw;
  const float *coarse = f.coarse + (size_t)b * C * SPFE_DESC_DIM;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int txx = x0 + (t & 1), tyy = y0 + (t >> 1);
    if (txx < 0 || txx >= wc || tyy < 0 || tyy >= hc) continue;  // zeros padding
    const float4 v = *reinterpret_cast<const float4 *>(coarse + ((size_t)tyy * wc + txx) * SPFE_DESC_DIM + lane * 4);
    float4 sq;
    sq.x = v.x * v.x; sq.y = v.y * v.y; sq.z = v.z * v.z; sq.w = v.w * v.w;
    const float nrm = sqrtf(sum256_wave(sq));
    acc.x = acc.x + (v.x / nrm) * tw[t];
    acc.y = acc.y + (v.y / nrm) * tw[t];
    acc.z = acc.z + (v.z / nrm) * tw[t];
    acc.w = acc.w + (v.w / nrm) * tw[t];
  }
  float4 sq;
  sq.x = acc.x * acc.x; sq.y = acc.y * acc.y; sq.z = acc.z * acc.z; sq.w = acc.w * acc.w;
  const float nrm = sqrtf(sum256_wave(sq));
  float4 o;
  o.x = acc.x / nrm; o.y = acc.y / nrm; o.z = acc.z / nrm; o.w = acc.w / nrm;
  if (rl.desc_bf16) {   // SPFE_FLAG_DESC_BF16: the same descriptor, rounded to nearest even
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    uint2 pk;
    pk.x = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){o.x, o.y}, bf16x2_));
    pk.y = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){o.z, o.w}, bf16x2_));
    unsigned short *d16 = reinterpret_cast<unsigned short *>(rec + rl.off_desc);
    *reinterpret_cast<uint2 *>(d16 + (size_t)i * SPFE_DESC_DIM + lane * 4) = pk;
  } else {
    float *desc = reinterpret_cast<float *>(rec + rl.off_desc);
    *reinterpret_cast<float4 *>(desc + (size_t)i * SPFE_DESC_DIM + lane * 4) = o;
  }
  if (lane == 0) {
    float *resp = reinterpret_cast<float *>(rec + rl.off_resp);
    // :271 response = heat_inv(y, x): the log heat through to_heat's multiply + add (cov.hip hinv_of: the bits of the map)
    const float t = f.heat_log[(size_t)b * H * W + (size_t)(int)y * W + (int)x] * f.heat_consts[(size_t)b * 4 + 2];
    resp[i] = t + f.heat_consts[(size_t)b * 4 + 3];
  }
}


}  // namespace spfe
