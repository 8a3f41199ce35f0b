/*
 * spfe_exact_math.h — the arithmetic CONTRACT of the SuperPoint front-end.
 *
 * Every floating-point step of the extractor that can change an integer
 * decision (arg-max, ">= 0.007", NMS priority, BFS "<" compares) is defined
 * here as a fixed sequence of IEEE-754 binary32 operations (+, -, *, /, sqrt,
 * fma) so that a gfx950 kernel and a host C function produce the SAME BITS.
 * The header is included by the HIP kernels (sp_orb_slam_amd/csrc) and by the
 * CPU oracle (oracle/spfe_oracle.c); both must be compiled with
 * -ffp-contract=off and without fast-math.  Nothing here comes from the
 * reference: the reference delegates these steps to libtorch/cuDNN/OpenCV,
 * whose operation order is unspecified; this file pins
 * one order.
 *
 * Reference call sites the functions stand for
 * (orb_slam2/src/cv/sp_extractor.cpp):
 *   spfe_expf      softmax numerator            :105
 *   spfe_logf      log(clamp(p, 0.001))         :129-130
 *   spfe_sum64     channel reductions           :102,105,148
 *   SPFE_LAYERS    conv plan                    :16-43, 81-100
 */
#ifndef SPFE_EXACT_MATH_H
#define SPFE_EXACT_MATH_H

#include <stdint.h>
#include <string.h>
#include <math.h>

#if defined(__HIPCC__)
#define SPFE_HD __host__ __device__ __forceinline__
#else
#define SPFE_HD static inline
#endif

#if defined(__clang__)
#pragma STDC FP_CONTRACT OFF
#endif

#define SPFE_CELL 8            /* sp_extractor.cpp:354 (cell_size) */
#define SPFE_DESC_DIM 256      /* sp_extractor.cpp:21 (d1) */
#define SPFE_SEMI_CH 65        /* sp_extractor.cpp:40 */
#define SPFE_SCORE_THRESH 0.007f /* sp_extractor.cpp:122 */
#define SPFE_HEAT_FLOOR 0.001f   /* sp_extractor.cpp:129 */
#define SPFE_NMS_DIST 4          /* sp_extractor.cpp:502 */
#define SPFE_NMS_BORDER 8        /* sp_extractor.cpp:502 */

/* ------------------------------------------------------------------------- */
/* Layer plan (sp_extractor.cpp:16-43 channel plan, :81-100 activation plan). */
/* kc = input-channel chunk of the implicit-GEMM K loop.  The accumulation     */
/* order of every conv output is                                              */
/*   acc = +0; for chunk: for tap(ky*3+kx): for c in chunk: acc=fma(x,w,acc)   */
/*   out = acc + bias; (relu)                                                  */
/* which is exactly what a chain of v_mfma_f32_32x32x2_f32 computes when the   */
/* K index is fed in that order (MFMA f32 == k-ordered fmaf chain, bitwise).   */
/* ------------------------------------------------------------------------- */
typedef struct {
  const char *name;
  int cin, cout, ksize, relu, pool, kc;
} spfe_layer_t;

#define SPFE_NUM_LAYERS 12
static const spfe_layer_t SPFE_LAYERS[SPFE_NUM_LAYERS] = {
    {"conv1a", 1, 64, 3, 1, 0, 1},     {"conv1b", 64, 64, 3, 1, 1, 16},
    {"conv2a", 64, 64, 3, 1, 0, 16},   {"conv2b", 64, 64, 3, 1, 1, 16},
    {"conv3a", 64, 128, 3, 1, 0, 16},  {"conv3b", 128, 128, 3, 1, 1, 16},
    {"conv4a", 128, 128, 3, 1, 0, 16}, {"conv4b", 128, 128, 3, 1, 0, 16},
    {"convPa", 128, 256, 3, 1, 0, 16}, {"convPb", 256, 65, 1, 0, 0, 16},
    {"convDa", 128, 256, 3, 1, 0, 16}, {"convDb", 256, 256, 1, 0, 0, 16},
};

SPFE_HD float spfe_bits_to_float(uint32_t u) {
  float f;
#if defined(__HIP_DEVICE_COMPILE__)
  f = __uint_as_float(u);
#else
  memcpy(&f, &u, 4);
#endif
  return f;
}
SPFE_HD uint32_t spfe_float_to_bits(float f) {
  uint32_t u;
#if defined(__HIP_DEVICE_COMPILE__)
  u = __float_as_uint(f);
#else
  memcpy(&u, &f, 4);
#endif
  return u;
}

/* u8 pixel -> float, sp_extractor.cpp:388: convertTo(CV_32F, 1.f/255.f) is a
 * MULTIPLY by the float constant 1.f/255.f. */
SPFE_HD float spfe_pixel_to_float(uint8_t p) { return (float)p * (1.0f / 255.0f); }

/* exp(x) for x <= 0 (softmax numerators).  Cephes-style range reduction and a
 * degree-5 minimax polynomial, all in binary32 with explicit fma. x < -86
 * returns 0 (the true value is < 5e-38; keeps every intermediate normal). */
SPFE_HD float spfe_expf(float x) {
  if (x < -86.0f) return 0.0f;
  const float t = x * 1.44269504088896341f;
  const float n = (t + 12582912.0f) - 12582912.0f; /* round-to-nearest-even */
  float r = fmaf(n, -0.693359375f, x);
  r = fmaf(n, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = fmaf(p, r, 1.3981999507e-3f);
  p = fmaf(p, r, 8.3334519073e-3f);
  p = fmaf(p, r, 4.1665795894e-2f);
  p = fmaf(p, r, 1.6666665459e-1f);
  p = fmaf(p, r, 5.0000001201e-1f);
  const float r2 = r * r;
  const float y = fmaf(p, r2, r) + 1.0f;
  const int ni = (int)n;
  return y * spfe_bits_to_float((uint32_t)(ni + 127) << 23);
}

/* log(x) for normal positive x (used on [0.001, 1]).  The fdlibm/msun e_logf
 * scheme: x = 2^k * (1+f), s = f/(2+f), even polynomial in s. Strict
 * left-to-right evaluation, no contraction. */
SPFE_HD float spfe_logf(float x) {
  uint32_t ix = spfe_float_to_bits(x);
  ix += 0x3f800000u - 0x3f3504f3u;
  const int k = (int)(ix >> 23) - 0x7f;
  ix = (ix & 0x007fffffu) + 0x3f3504f3u;
  const float m = spfe_bits_to_float(ix);
  const float f = m - 1.0f;
  const float s = f / (2.0f + f);
  const float z = s * s;
  const float w = z * z;
  const float t1 = w * (0.40000972152f + w * 0.24279078841f);
  const float t2 = z * (0.66666662693f + w * 0.28498786688f);
  const float R = t2 + t1;
  const float hfsq = (0.5f * f) * f;
  const float dk = (float)k;
  float res = s * (hfsq + R);
  res = res + dk * 9.0580006145e-06f;
  res = res - hfsq;
  res = res + f;
  res = res + dk * 6.9313812256e-01f;
  return res;
}

/* 64-lane butterfly sum: the order a wavefront's xor-shuffle reduction uses
 * (offsets 32,16,8,4,2,1).  Addition is commutative, so all 64 lanes end with
 * the same value; v[0] is returned.  Host form (the device form lives in the
 * kernels and uses __shfl_xor with the same offsets). */
SPFE_HD float spfe_sum64_host(const float *v64) {
  float v[64];
  for (int i = 0; i < 64; ++i) v[i] = v64[i];
  for (int off = 32; off >= 1; off >>= 1) {
    float nv[64];
    for (int i = 0; i < 64; ++i) nv[i] = v[i] + v[i ^ off];
    for (int i = 0; i < 64; ++i) v[i] = nv[i];
  }
  return v[0];
}

/* Sum of 256 values as a wavefront computes it: lane l owns elements
 * 4l..4l+3, adds them left to right, then the butterfly above. */
SPFE_HD float spfe_sum256_host(const float *v256) {
  float part[64];
  for (int l = 0; l < 64; ++l) {
    float s = v256[4 * l];
    s = s + v256[4 * l + 1];
    s = s + v256[4 * l + 2];
    s = s + v256[4 * l + 3];
    part[l] = s;
  }
  return spfe_sum64_host(part);
}

/* Candidate priority used by the score sort (sp_extractor.cpp:489-498) and by
 * NMS: descending score, ties broken by ascending candidate (= cell) index.
 * The reference's cv::sortIdx tie order is unspecified; this is the rule the
 * build defines.  Scores are positive finite
 * floats, so their bit patterns order like the values. */
SPFE_HD int spfe_ranks_before(float score_a, int idx_a, float score_b, int idx_b) {
  return (score_a > score_b) || (score_a == score_b && idx_a < idx_b);
}

#endif /* SPFE_EXACT_MATH_H */
