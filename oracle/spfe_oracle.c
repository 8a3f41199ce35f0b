/*
 * spfe_oracle.c — CPU ORACLE for the SuperPoint extraction path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under sp_orb_slam_amd/ or include/ links,
 * imports or calls this file; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg do, and there only as the checker.
 *
 * PARITY UNPINNED: the reference (HyHuang1995/sp_orb_slam) ships no tests,
 * golden vectors or weights for this path and cannot be built or run here
 * (libtorch-CUDA hard-wired, OpenCV/Eigen/ROS absent — SURVEY.md §8c).  This
 * file is a plain-C restatement of the reference algorithm, function by
 * function, each citing the reference lines it follows; it is cross-checked in
 * the build container against the same ATen op sequence executed by PyTorch-CPU
 * (tests/golden/make_golden.py -> tests/golden/*.npz).
 *
 * All citations are to /root/reference/orb_slam2/src/cv/sp_extractor.cpp
 * unless stated otherwise.  Arithmetic order for every float step is the one
 * fixed in include/spfe_exact_math.h (the reference leaves it to
 * libtorch/cuDNN/OpenCV).
 *
 * Data layouts (ours, not the reference's NCHW): activations are NHWC float,
 * i.e. [y][x][c]; `semi` is [hc][wc][65]; `coarse` is [hc][wc][256].
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <math.h>

#include "../include/spfe_exact_math.h"

#define EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* weight blob: for each layer in SPFE_LAYERS order: weight OIHW, then bias —  */
/* the register_module order of sp_extractor.cpp:46-62.                        */
/* ------------------------------------------------------------------------- */
static size_t layer_weight_count(int l) {
  const spfe_layer_t *L = &SPFE_LAYERS[l];
  return (size_t)L->cout * L->cin * L->ksize * L->ksize;
}
EXPORT size_t oracle_weight_offset(int l) {
  size_t off = 0;
  for (int i = 0; i < l; ++i) off += layer_weight_count(i) + SPFE_LAYERS[i].cout;
  return off;
}
EXPORT size_t oracle_bias_offset(int l) { return oracle_weight_offset(l) + layer_weight_count(l); }
EXPORT size_t oracle_num_params(void) { return oracle_weight_offset(SPFE_NUM_LAYERS); }

/* ------------------------------------------------------------------------- */
/* One conv layer (+bias, +ReLU): torch::nn::Conv2d cross-correlation, stride 1,*/
/* zero padding ksize/2 (:27-43), relu(:81-99).  in: [H][W][cin], out:          */
/* [H][W][cout].  Accumulation order = spfe_exact_math.h layer plan.            */
/* ------------------------------------------------------------------------- */
/* bf16 mode of the build (SPFE_PRECISION_BF16, BASELINE configs[3]): round-to-nearest-even
 * to bfloat16, kept in a float. */
static float bf16_round(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  u &= 0xffff0000u;
  memcpy(&f, &u, 4);
  return f;
}

static void conv_layer_ex(int l, const float *blob, const float *in, int H, int W, float *out, int w_bf16,
                          int out_bf16);
static void conv_layer(int l, const float *blob, const float *in, int H, int W, float *out) {
  conv_layer_ex(l, blob, in, H, W, out, 0, 0);
}

static void conv_layer_ex(int l, const float *blob, const float *in, int H, int W, float *out, int w_bf16,
                          int out_bf16) {
  const spfe_layer_t *L = &SPFE_LAYERS[l];
  const int cin = L->cin, cout = L->cout, ks = L->ksize, pad = ks / 2, taps = ks * ks;
  const int kc = L->kc < cin ? L->kc : cin;
  const int nchunk = cin / kc;
  const float *w = blob + oracle_weight_offset(l); /* [cout][cin][ky][kx] */
  const float *b = blob + oracle_bias_offset(l);
  const int Hp = H + 2 * pad, Wp = W + 2 * pad;
  const int coutp = (cout + 63) & ~63; /* pad to 64 for the vector loop */

  /* zero-padded input copy */
  float *inp = (float *)calloc((size_t)Hp * Wp * cin, sizeof(float));
  for (int y = 0; y < H; ++y)
    memcpy(inp + ((size_t)(y + pad) * Wp + pad) * cin, in + (size_t)y * W * cin,
           (size_t)W * cin * sizeof(float));
  /* weights in K order: wt[chunk][tap][c][coutp] */
  float *wt = (float *)calloc((size_t)nchunk * taps * kc * coutp, sizeof(float));
  for (int ch = 0; ch < nchunk; ++ch)
    for (int t = 0; t < taps; ++t)
      for (int c = 0; c < kc; ++c)
        for (int co = 0; co < cout; ++co)
          wt[(((size_t)ch * taps + t) * kc + c) * coutp + co] =
              w_bf16 ? bf16_round(w[((size_t)co * cin + ch * kc + c) * taps + t])
                     : w[((size_t)co * cin + ch * kc + c) * taps + t];

#pragma omp parallel for schedule(dynamic, 1)
  for (int y = 0; y < H; ++y) {
    float acc[64];
    for (int x = 0; x < W; ++x) {
      for (int cb = 0; cb < coutp; cb += 64) {
        for (int i = 0; i < 64; ++i) acc[i] = 0.0f;
        for (int ch = 0; ch < nchunk; ++ch)
          for (int t = 0; t < taps; ++t) {
            const int ky = t / ks, kx = t % ks;
            const float *px = inp + ((size_t)(y + ky) * Wp + (x + kx)) * cin + ch * kc;
            const float *wrow = wt + (((size_t)ch * taps + t) * kc) * coutp + cb;
            for (int c = 0; c < kc; ++c) {
              const float xv = px[c];
              const float *wr = wrow + (size_t)c * coutp;
              for (int i = 0; i < 64; ++i) acc[i] = __builtin_fmaf(xv, wr[i], acc[i]);
            }
          }
        const int nco = (cout - cb) < 64 ? (cout - cb) : 64;
        float *o = out + ((size_t)y * W + x) * cout + cb;
        for (int i = 0; i < nco; ++i) {
          float v = acc[i] + b[cb + i];
          if (L->relu) v = v > 0.0f ? v : 0.0f;
          o[i] = out_bf16 ? bf16_round(v) : v;
        }
      }
    }
  }
  free(inp);
  free(wt);
}

/* torch::max_pool2d(x, 2, 2)  (:83,87,91).  in [H][W][c] -> out [H/2][W/2][c] */
static void maxpool2(const float *in, int H, int W, int c, float *out) {
  const int Ho = H / 2, Wo = W / 2;
  for (int y = 0; y < Ho; ++y)
    for (int x = 0; x < Wo; ++x)
      for (int k = 0; k < c; ++k) {
        float a = in[((size_t)(2 * y) * W + 2 * x) * c + k];
        float b = in[((size_t)(2 * y) * W + 2 * x + 1) * c + k];
        float d = in[((size_t)(2 * y + 1) * W + 2 * x) * c + k];
        float e = in[((size_t)(2 * y + 1) * W + 2 * x + 1) * c + k];
        float m = a > b ? a : b;
        float n = d > e ? d : e;
        out[((size_t)y * Wo + x) * c + k] = m > n ? m : n;
      }
}

/*
 * SPFrontend::forward, network part (:81-100) + input conversion (:388).
 * img: u8 [H][W]; outputs semi [hc][wc][65], coarse [hc][wc][256] (raw, not yet
 * normalised).  `feat` (optional) receives conv4b output [hc][wc][128].
 */
EXPORT int oracle_network(const float *blob, const uint8_t *img, int H, int W, float *semi,
                          float *coarse, float *feat) {
  if (H % 8 || W % 8 || H <= 0 || W <= 0) return -1;
  size_t maxel = (size_t)H * W * 64;
  float *a = (float *)malloc(maxel * sizeof(float));
  float *b = (float *)malloc(maxel * sizeof(float));
  float *x0 = (float *)malloc((size_t)H * W * sizeof(float));
  for (size_t i = 0; i < (size_t)H * W; ++i) x0[i] = spfe_pixel_to_float(img[i]);
  int h = H, w = W;
  conv_layer(0, blob, x0, h, w, a); /* conv1a */
  conv_layer(1, blob, a, h, w, b);  /* conv1b */
  maxpool2(b, h, w, 64, a);
  h /= 2, w /= 2;
  conv_layer(2, blob, a, h, w, b); /* conv2a */
  conv_layer(3, blob, b, h, w, a); /* conv2b */
  maxpool2(a, h, w, 64, b);
  h /= 2, w /= 2;
  conv_layer(4, blob, b, h, w, a); /* conv3a */
  conv_layer(5, blob, a, h, w, b); /* conv3b */
  maxpool2(b, h, w, 128, a);
  h /= 2, w /= 2;
  conv_layer(6, blob, a, h, w, b); /* conv4a */
  conv_layer(7, blob, b, h, w, a); /* conv4b -> a */
  if (feat) memcpy(feat, a, (size_t)h * w * 128 * sizeof(float));
  float *cPa = (float *)malloc((size_t)h * w * 256 * sizeof(float));
  conv_layer(8, blob, a, h, w, cPa);     /* convPa + relu (:96) */
  conv_layer(9, blob, cPa, h, w, semi);  /* convPb (:97) */
  conv_layer(10, blob, a, h, w, cPa);    /* convDa + relu (:99) */
  conv_layer(11, blob, cPa, h, w, coarse); /* convDb (:100) */
  free(cPa);
  free(a);
  free(b);
  free(x0);
  return 0;
}

/*
 * The network in the build's bf16 mode: conv1a as conv1a_bf16() below (bf16 weights with the 1/255 folded in, exact u8
 * inputs, the bias as the accumulator's start, output rounded to bf16);
 * conv1b..conv4b with bf16 weights and activations (f32 accumulate, bias, ReLU, pool; outputs
 * rounded); convPa and convDa with bf16 weights and bf16 output; the 1x1 heads convPb (the detector
 * logits) and convDb with bf16 weights, f32 accumulate and f32 output; everything after the two
 * heads in f32.  (The MFMA's accumulation order differs from this loop's, so the GPU is
 * compared with a tolerance in this mode, not bitwise.)
 */
/* conv1a of the bf16 mode (sp_orb_slam_amd/csrc/conv1a_mfma.h, round 6 form): the u8 pixels enter the product unscaled (exact
 * in bf16); the 1/255 of convertTo (:388) is folded into the weight before its rounding to bf16 (one f32 multiply by
 * float(1/255), then RNE — the library's host packing does the same); the bias is the accumulator's initial value:
 *   a0[c] = bf16( max( b[c] + sum_t float(u8_t) * bf16(w[c][t] * (1/255)), 0 ) ),  f32 accumulate, zero padding.
 * (Rounds 2 - 5: bf16(w) and fmaf(sum, 1/255, b) per value.  Both are bf16 quantisations of the same f32 layer.) */
static void conv1a_bf16(const float *blob, const uint8_t *img, int H, int W, float *out) {
  const float *w = blob + oracle_weight_offset(0); /* [64][1][3][3] */
  const float *b = blob + oracle_bias_offset(0);
  float wb[64 * 9];
  for (int i = 0; i < 64 * 9; ++i) wb[i] = bf16_round(w[i] * (1.0f / 255.0f));
#pragma omp parallel for schedule(static)
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      float px[9];
      for (int t = 0; t < 9; ++t) {
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        px[t] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? (float)img[(size_t)yy * W + xx] : 0.0f;
      }
      float *o = out + ((size_t)y * W + x) * 64;
      for (int c = 0; c < 64; ++c) {
        float s = b[c];
        for (int t = 0; t < 9; ++t) s = __builtin_fmaf(px[t], wb[c * 9 + t], s);
        s = s > 0.0f ? s : 0.0f;
        o[c] = bf16_round(s);
      }
    }
}

EXPORT int oracle_network_bf16(const float *blob, const uint8_t *img, int H, int W, float *semi,
                               float *coarse) {
  if (H % 8 || W % 8 || H <= 0 || W <= 0) return -1;
  size_t maxel = (size_t)H * W * 64;
  float *a = (float *)malloc(maxel * sizeof(float));
  float *b = (float *)malloc(maxel * sizeof(float));
  float *x0 = (float *)malloc((size_t)H * W * sizeof(float));
  for (size_t i = 0; i < (size_t)H * W; ++i) x0[i] = spfe_pixel_to_float(img[i]);
  int h = H, w = W;
  conv1a_bf16(blob, img, h, w, a);
  conv_layer_ex(1, blob, a, h, w, b, 1, 1);
  maxpool2(b, h, w, 64, a);
  h /= 2, w /= 2;
  conv_layer_ex(2, blob, a, h, w, b, 1, 1);
  conv_layer_ex(3, blob, b, h, w, a, 1, 1);
  maxpool2(a, h, w, 64, b);
  h /= 2, w /= 2;
  conv_layer_ex(4, blob, b, h, w, a, 1, 1);
  conv_layer_ex(5, blob, a, h, w, b, 1, 1);
  maxpool2(b, h, w, 128, a);
  h /= 2, w /= 2;
  conv_layer_ex(6, blob, a, h, w, b, 1, 1);
  conv_layer_ex(7, blob, b, h, w, a, 1, 1);
  float *cPa = (float *)malloc((size_t)h * w * 256 * sizeof(float));
  conv_layer_ex(8, blob, a, h, w, cPa, 1, 1);        /* convPa: bf16 weights, bf16 output */
  conv_layer_ex(9, blob, cPa, h, w, semi, 1, 0);     /* convPb: bf16 weights, f32 output */
  conv_layer_ex(10, blob, a, h, w, cPa, 1, 1);       /* convDa: bf16 weights, bf16 output */
  conv_layer_ex(11, blob, cPa, h, w, coarse, 1, 0);  /* convDb: bf16 weights, f32 output */
  free(cPa);
  free(a);
  free(b);
  free(x0);
  return 0;
}

/*
 * Detector tail (:105-131): channel softmax over 65 logits per cell, dustbin
 * slices, per-cell max/arg-max over the 64 non-dust channels (lowest index on
 * ties, ATen semantics), pixel coordinates through the `grid` convention of
 * :64-73 (channel k <-> dy=k/8, dx=k%8), threshold >= 0.007, row-major
 * compaction, and the log-heat map with pixel_shuffle(8).
 *
 * cand_*: capacity hc*wc.  Returns N (number of candidates).
 * heat_log: [H][W].  dense_dust / semi_dust: [hc][wc].
 */
EXPORT int oracle_tail(const float *semi, int H, int W, float *dense_dust, float *semi_dust,
                       float *heat_log, float *cand_x, float *cand_y, float *cand_score,
                       int *cand_cell) {
  const int hc = H / 8, wc = W / 8;
  int n = 0;
  for (int cy = 0; cy < hc; ++cy)
    for (int cx = 0; cx < wc; ++cx) {
      const float *s = semi + ((size_t)cy * wc + cx) * SPFE_SEMI_CH;
      float m = s[0];
      for (int k = 1; k < SPFE_SEMI_CH; ++k) m = s[k] > m ? s[k] : m;
      float e[64];
      for (int k = 0; k < 64; ++k) e[k] = spfe_expf(s[k] - m);
      const float ed = spfe_expf(s[64] - m);
      const float total = spfe_sum64_host(e) + ed;
      semi_dust[cy * wc + cx] = s[64];       /* :106 */
      dense_dust[cy * wc + cx] = ed / total; /* :107 */
      float best = -1.0f;
      int bi = 0;
      for (int k = 0; k < 64; ++k) {
        const float p = e[k] / total; /* :105 */
        if (p > best) { best = p; bi = k; } /* strict > : lowest index wins ties (:112) */
        const float pc = p < SPFE_HEAT_FLOOR ? SPFE_HEAT_FLOOR : p; /* :129 */
        heat_log[(size_t)(cy * 8 + k / 8) * W + cx * 8 + k % 8] = spfe_logf(pc); /* :130-131 */
      }
      if (best >= SPFE_SCORE_THRESH) { /* :122 */
        cand_x[n] = (float)(cx * 8 + bi % 8); /* :64-73,118 */
        cand_y[n] = (float)(cy * 8 + bi / 8);
        cand_score[n] = best;
        cand_cell[n] = cy * wc + cx;
        ++n;
      }
    }
  return n;
}

/*
 * Descriptor sampling (:102-103, :134-148).  coarse [hc][wc][256] RAW; each tap
 * is first divided by its cell's L2 norm (:102-103, no epsilon), then bilinear
 * grid_sample with align_corners=true and zero padding, then the sampled
 * 256-vector is L2-normalised (:148).  Coordinates follow the literal steps
 * c = x / (w/2) - 1 (:137-138), ix = ((c + 1) / 2) * (wc - 1) (ATen
 * grid_sampler_unnormalize, align_corners).
 * out desc [n][256].
 */
static float cell_norm(const float *v) {
  float sq[256];
  for (int i = 0; i < 256; ++i) sq[i] = v[i] * v[i];
  return sqrtf(spfe_sum256_host(sq));
}
EXPORT void oracle_sample_desc(const float *coarse, int H, int W, const float *xs, const float *ys,
                               int n, float *desc) {
  const int hc = H / 8, wc = W / 8;
  /* x_s.div(w / 2.0) - 1.0 (:137-138).  ATen-CUDA (the only device the reference
   * runs on, :348-351) evaluates tensor / scalar as tensor * float(1.0 / scalar). */
  const float inv_hw = (float)(1.0 / (double)(float)(W / 2.0));
  const float inv_hh = (float)(1.0 / (double)(float)(H / 2.0));
  for (int i = 0; i < n; ++i) {
    const float gx = xs[i] * inv_hw - 1.0f;
    const float gy = ys[i] * inv_hh - 1.0f;
    const float ix = ((gx + 1.0f) / 2.0f) * (float)(wc - 1);
    const float iy = ((gy + 1.0f) / 2.0f) * (float)(hc - 1);
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fx0, wy1 = iy - fy0; /* weight of the +1 neighbours */
    const float wx0 = (fx0 + 1.0f) - ix, wy0 = (fy0 + 1.0f) - iy;
    const float w_nw = wx0 * wy0, w_ne = wx1 * wy0, w_sw = wx0 * wy1, w_se = wx1 * wy1;
    const int tx[4] = {x0, x1, x0, x1}, ty[4] = {y0, y0, y1, y1};
    const float tw[4] = {w_nw, w_ne, w_sw, w_se};
    float acc[256];
    for (int c = 0; c < 256; ++c) acc[c] = 0.0f;
    for (int t = 0; t < 4; ++t) {
      if (tx[t] < 0 || tx[t] >= wc || ty[t] < 0 || ty[t] >= hc) continue; /* zeros padding */
      const float *v = coarse + ((size_t)ty[t] * wc + tx[t]) * 256;
      const float nrm = cell_norm(v);
      for (int c = 0; c < 256; ++c) acc[c] = acc[c] + (v[c] / nrm) * tw[t];
    }
    float sq[256];
    for (int c = 0; c < 256; ++c) sq[c] = acc[c] * acc[c];
    const float nrm = sqrtf(spfe_sum256_host(sq));
    for (int c = 0; c < 256; ++c) desc[(size_t)i * 256 + c] = acc[c] / nrm;
  }
}

/*
 * to_heat (:461-474): img = -heat_log; (min,max) by minMaxLoc (doubles); the two
 * cv::MatExpr results evaluate as ONE affine map per pixel:
 *   heat     = L * (-1/(max-min)) + (-min/(max-min))
 *   heat_inv = L * ( 1/(max-min)) + ( max/(max-min))
 * with the scale/shift formed in double ( x * (1.0/d) ) and rounded to float,
 * then a float multiply followed by a float add.
 */
EXPORT void oracle_heat(const float *heat_log, int H, int W, float *heat, float *heat_inv,
                        double *minmax_out) {
  const size_t n = (size_t)H * W;
  float mn = -heat_log[0], mx = -heat_log[0];
  for (size_t i = 1; i < n; ++i) {
    const float m = -heat_log[i];
    mn = m < mn ? m : mn;
    mx = m > mx ? m : mx;
  }
  const double dmin = (double)mn, dmax = (double)mx;
  const double inv = 1.0 / (dmax - dmin);
  const float a_h = (float)(-inv), b_h = (float)(-dmin * inv);
  const float a_i = (float)(inv), b_i = (float)(dmax * inv);
  for (size_t i = 0; i < n; ++i) {
    const float L = heat_log[i];
    if (heat) heat[i] = L * a_h + b_h;
    if (heat_inv) heat_inv[i] = L * a_i + b_i;
  }
  if (minmax_out) { minmax_out[0] = dmin; minmax_out[1] = dmax; }
}

/*
 * Descending score sort (:489-498), tie rule of spfe_ranks_before.
 * order[i] = index (into the candidate arrays) of the i-th best candidate.
 */
static const float *g_sort_score;
static int cmp_rank(const void *pa, const void *pb) {
  const int a = *(const int *)pa, b = *(const int *)pb;
  if (spfe_ranks_before(g_sort_score[a], a, g_sort_score[b], b)) return -1;
  if (spfe_ranks_before(g_sort_score[b], b, g_sort_score[a], a)) return 1;
  return 0;
}
EXPORT void oracle_sort(const float *score, int n, int *order) {
  for (int i = 0; i < n; ++i) order[i] = i;
  g_sort_score = score;
  qsort(order, n, sizeof(int), cmp_rank); /* keys are unique (index tie-break): order is total */
}

/*
 * nms() (:161-250), literal: byte grid W x H padded by dist_thresh, greedy 9x9
 * suppression in sorted order over ALL candidates, stop after the
 * (num_features+1)-th survivor, border reject, raster scan output, occ_grid.
 * px,py: candidate coordinates ALREADY in sorted order (n entries).
 * Outputs: kp_x, kp_y (integer valued), kp_src (index into the sorted list),
 * occ_grid int16 [H/8][W/8] (-1 = empty).  Returns K.
 */
EXPORT int oracle_nms(const float *px, const float *py, int n, int num_features, int border,
                      int dist, int W, int H, float *kp_x, float *kp_y, int *kp_src,
                      int16_t *occ_grid) {
  const int GW = W + 2 * dist, GH = H + 2 * dist;
  unsigned char *grid = (unsigned char *)calloc((size_t)GW * GH, 1);
  unsigned short *inds = (unsigned short *)calloc((size_t)W * H, sizeof(unsigned short));
  for (int i = 0; i < (H / 8) * (W / 8); ++i) occ_grid[i] = -1; /* :178 */
  for (int i = 0; i < n; ++i) { /* :183-189 */
    const int uu = (int)px[i], vv = (int)py[i];
    grid[(size_t)(vv + dist) * GW + uu + dist] = 1;
    inds[(size_t)vv * W + uu] = (unsigned short)i;
  }
  int n_feature = 0;
  for (int i = 0; i < n; ++i) { /* :195-214 */
    const int uu = (int)px[i] + dist, vv = (int)py[i] + dist;
    if (grid[(size_t)vv * GW + uu] != 1) continue;
    for (int k = -dist; k < dist + 1; ++k)
      for (int j = -dist; j < dist + 1; ++j) {
        if (j == 0 && k == 0) continue;
        grid[(size_t)(vv + k) * GW + uu + j] = 0;
      }
    grid[(size_t)vv * GW + uu] = 2;
    n_feature++;
    if (n_feature > num_features) break;
  }
  int n_pts = 0;
  for (int v = 0; v < H + dist; ++v) /* :220-238 */
    for (int u = 0; u < W + dist; ++u) {
      if (u - dist >= W - border || u - dist < border || v - dist >= H - border ||
          v - dist < border)
        continue;
      if (grid[(size_t)v * GW + u] == 2) {
        occ_grid[((v - dist) / 8) * (W / 8) + (u - dist) / 8] = (int16_t)n_pts;
        const int sel = inds[(size_t)(v - dist) * W + (u - dist)];
        kp_x[n_pts] = (float)(int)px[sel];
        kp_y[n_pts] = (float)(int)py[sel];
        kp_src[n_pts] = sel;
        n_pts++;
      }
    }
  free(grid);
  free(inds);
  return n_pts;
}

/*
 * computeCovariance() (:252-340), literal: FIFO BFS per keypoint in emitted
 * order down the heat_inv hill, ONE visited mask shared by all keypoints,
 * visited set at pop, neighbour order left/up/right/down with bounds
 * xx>0, yy>0, xx<w, yy<h, accept iff unvisited && value>0 && value<current.
 * cov = sum (s_i / sum s) * delta_i^2, clamp >= 1, cov_inv = 1/cov.
 * response[i] = heat_inv(y,x) (:271).
 */
EXPORT void oracle_covariance(const float *heat_inv, int H, int W, const float *kp_x,
                              const float *kp_y, int K, float *cov2, float *cov2_inv,
                              float *response) {
  unsigned char *fresh = (unsigned char *)malloc((size_t)H * W);
  memset(fresh, 1, (size_t)H * W);
  size_t cap = 1024, qcap = 1024;
  float *dx2 = (float *)malloc(cap * sizeof(float));
  float *dy2 = (float *)malloc(cap * sizeof(float));
  float *sc = (float *)malloc(cap * sizeof(float));
  int *q = (int *)malloc(qcap * sizeof(int));
  for (int i = 0; i < K; ++i) {
    const int uu = (int)kp_x[i], vv = (int)kp_y[i];
    response[i] = heat_inv[(size_t)vv * W + uu];
    size_t cnt = 0, qh = 0, qt = 0;
    q[qt++] = vv * W + uu;
    while (qh < qt) {
      const int cur = q[qh++];
      const int u = cur % W, v = cur / W;
      fresh[cur] = 0;
      if (cnt == cap) {
        cap *= 2;
        dx2 = (float *)realloc(dx2, cap * sizeof(float));
        dy2 = (float *)realloc(dy2, cap * sizeof(float));
        sc = (float *)realloc(sc, cap * sizeof(float));
      }
      const float fdx = (float)u - (float)uu, fdy = (float)v - (float)vv;
      dx2[cnt] = fdx * fdx;
      dy2[cnt] = fdy * fdy;
      const float centroid = heat_inv[cur];
      sc[cnt] = centroid;
      cnt++;
      if (qt + 4 > qcap) {
        qcap *= 2;
        q = (int *)realloc(q, qcap * sizeof(int));
      }
#define CHECK_UV(u_, v_)                                                      \
  do {                                                                        \
    const int id_ = (v_) * W + (u_);                                          \
    const float hv_ = heat_inv[id_];                                          \
    if (fresh[id_] && hv_ > 0.0f && hv_ < centroid) q[qt++] = id_;            \
  } while (0)
      if (u - 1 > 0) CHECK_UV(u - 1, v);
      if (v - 1 > 0) CHECK_UV(u, v - 1);
      if (u + 1 < W) CHECK_UV(u + 1, v);
      if (v + 1 < H) CHECK_UV(u, v + 1);
#undef CHECK_UV
    }
    float sum = 0.0f;
    for (size_t j = 0; j < cnt; ++j) sum += sc[j];
    float cx = 0.0f, cy = 0.0f;
    for (size_t j = 0; j < cnt; ++j) {
      const float wgt = sc[j] / sum;
      cx += wgt * dx2[j];
      cy += wgt * dy2[j];
    }
    if (cx < 1.0f) cx = 1.0f;
    if (cy < 1.0f) cy = 1.0f;
    cov2[2 * i] = cx;
    cov2[2 * i + 1] = cy;
    cov2_inv[2 * i] = 1.0f / cx;
    cov2_inv[2 * i + 1] = 1.0f / cy;
  }
  free(fresh);
  free(dx2);
  free(dy2);
  free(sc);
  free(q);
}

/*
 * SPExtractor::operator() (:361-514) end to end, given semi/coarse.
 * Buffers sized by the caller: kp_* and cov* and response >= num_features+1,
 * desc >= (num_features+1)*256, occ_grid hc*wc, dense_dust/semi_dust hc*wc,
 * heat/heat_inv H*W.  Returns K (>=0) or <0 on error.
 */
EXPORT int oracle_postprocess(const float *semi, const float *coarse, int H, int W,
                              int num_features, float *kp_x, float *kp_y, float *response,
                              float *desc, float *cov2, float *cov2_inv, int16_t *occ_grid,
                              float *dense_dust, float *semi_dust, float *heat, float *heat_inv,
                              int *n_candidates) {
  const int C = (H / 8) * (W / 8);
  float *heat_log = (float *)malloc((size_t)H * W * sizeof(float));
  float *cx = (float *)malloc(C * sizeof(float)), *cy = (float *)malloc(C * sizeof(float));
  float *cs = (float *)malloc(C * sizeof(float));
  int *cc = (int *)malloc(C * sizeof(int));
  const int N = oracle_tail(semi, H, W, dense_dust, semi_dust, heat_log, cx, cy, cs, cc);
  if (n_candidates) *n_candidates = N;
  /* forward() samples descriptors for ALL candidates (:134-148) */
  float *desc_all = (float *)malloc((size_t)(N > 0 ? N : 1) * 256 * sizeof(float));
  oracle_sample_desc(coarse, H, W, cx, cy, N, desc_all);
  float *hinv_local = heat_inv ? heat_inv : (float *)malloc((size_t)H * W * sizeof(float));
  oracle_heat(heat_log, H, W, heat, hinv_local, NULL); /* :461-474 */
  int *order = (int *)malloc((N > 0 ? N : 1) * sizeof(int));
  oracle_sort(cs, N, order); /* :489-498 */
  float *sx = (float *)malloc((N > 0 ? N : 1) * sizeof(float));
  float *sy = (float *)malloc((N > 0 ? N : 1) * sizeof(float));
  for (int i = 0; i < N; ++i) { sx[i] = cx[order[i]]; sy[i] = cy[order[i]]; }
  int *src = (int *)malloc((size_t)(num_features + 2) * sizeof(int));
  const int K = oracle_nms(sx, sy, N, num_features, SPFE_NMS_BORDER, SPFE_NMS_DIST, W, H, kp_x,
                           kp_y, src, occ_grid); /* :502 */
  for (int i = 0; i < K; ++i) /* :240-249 */
    memcpy(desc + (size_t)i * 256, desc_all + (size_t)order[src[i]] * 256, 256 * sizeof(float));
  oracle_covariance(hinv_local, H, W, kp_x, kp_y, K, cov2, cov2_inv, response); /* :508 */
  if (!heat_inv) free(hinv_local);
  free(heat_log); free(cx); free(cy); free(cs); free(cc);
  free(desc_all); free(order); free(sx); free(sy); free(src);
  return K;
}

/* Full path: u8 image -> everything (network + postprocess). */
EXPORT int oracle_extract(const float *blob, const uint8_t *img, int H, int W, int num_features,
                          float *kp_x, float *kp_y, float *response, float *desc, float *cov2,
                          float *cov2_inv, int16_t *occ_grid, float *dense_dust,
                          float *semi_dust, float *heat, float *heat_inv, int *n_candidates) {
  if (!img) return -2;
  if (H % 8 || W % 8 || H <= 0 || W <= 0) return -1;
  const int C = (H / 8) * (W / 8);
  float *semi = (float *)malloc((size_t)C * SPFE_SEMI_CH * sizeof(float));
  float *coarse = (float *)malloc((size_t)C * 256 * sizeof(float));
  int rc = oracle_network(blob, img, H, W, semi, coarse, NULL);
  if (rc == 0)
    rc = oracle_postprocess(semi, coarse, H, W, num_features, kp_x, kp_y, response, desc, cov2,
                            cov2_inv, occ_grid, dense_dust, semi_dust, heat, heat_inv,
                            n_candidates);
  free(semi);
  free(coarse);
  return rc;
}

/*
 * SURVEY.md §8(f) rank 1 — descriptor matching.  Restates
 *   cv::BFMatcher::create(cv::NORM_L2, crossCheck)->add(train); ->match(query, matches)
 * as called by SPMatcher::SearchByBruteForce (orb_slam2/src/cv/sp_matcher.cpp:1642-1674), with
 * SPMatcher::DescriptorDistance (:1636-1640) = L2 norm of the difference.
 *
 * OpenCV (3.x, `find_package(OpenCV 3.0)` CMakeLists.txt:14; not vendored, not installed here —
 * PARITY UNPINNED, published algorithm restated): BFMatcher::knnMatchImpl(k = 1) calls
 * cv::batchDistance(query, train, dist, CV_32F, nidx, NORM_L2, 1, mask, 0, crossCheck), whose L2
 * distance is sqrt(sum (a-b)^2) in float and whose scans keep the first minimum (strict `<`).
 * With crossCheck it computes, for every TRAIN row, its nearest QUERY row, and then lets each
 * query keep the closest train row that chose it (strict `<` over ascending train index);
 * queries nobody chose get no DMatch.  Without crossCheck: the nearest train row per query.
 * OpenCV's summation order inside normL2Sqr_ depends on its SIMD build; this restatement fixes
 * the sequential fused chain of include/spfe.h (spfe_match).
 * Outputs: train_idx[nq] (-1 = no DMatch for that query), distance[nq] (FLT_MAX when -1).
 */
static float match_dist(const float *a, const float *b) {
  float s = 0.0f;
  for (int k = 0; k < 256; ++k) {
    const float d = a[k] - b[k];
    s = fmaf(d, d, s);
  }
  return sqrtf(s);
}

EXPORT void oracle_match_bruteforce(const float *query, int nq, const float *train, int nt, int cross_check,
                                    int32_t *train_idx, float *distance) {
  for (int i = 0; i < nq; ++i) {
    train_idx[i] = -1;
    distance[i] = FLT_MAX;
  }
  if (nq <= 0 || nt <= 0) return;
  if (!cross_check) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < nq; ++i) {
      float best = FLT_MAX;
      int bi = -1;
      for (int j = 0; j < nt; ++j) {
        const float d = match_dist(query + (size_t)i * 256, train + (size_t)j * 256);
        if (d < best) {
          best = d;
          bi = j;
        }
      }
      train_idx[i] = bi;
      distance[i] = best;
    }
    return;
  }
  float *tdist = (float *)malloc((size_t)nt * sizeof(float));
  int *tidx = (int *)malloc((size_t)nt * sizeof(int));
#pragma omp parallel for schedule(static)
  for (int j = 0; j < nt; ++j) {  /* nearest query of every train row */
    float best = FLT_MAX;
    int bi = -1;
    for (int i = 0; i < nq; ++i) {
      const float d = match_dist(train + (size_t)j * 256, query + (size_t)i * 256);
      if (d < best) {
        best = d;
        bi = i;
      }
    }
    tdist[j] = best;
    tidx[j] = bi;
  }
  for (int j = 0; j < nt; ++j) {  /* each query keeps the closest train row that chose it */
    const int i = tidx[j];
    if (i >= 0 && tdist[j] < distance[i]) {
      distance[i] = tdist[j];
      train_idx[i] = j;
    }
  }
  free(tdist);
  free(tidx);
}

/*
 * Patch-wise association of projected map points: the loop of Tracker::trackFrameDustKFLocal,
 * orb_slam2/src/tracking/tracker_dust.cpp:113-172, literally: occ_grid is cloned (:104); for every map
 * point in order (the caller has dropped `!in_view || isBad()` ones, :115-116) u = floor(dust_proj_u),
 * v = floor(dust_proj_v) (:118-119); best_dist starts at 0.75f (:122, here max_dist); du outer, dv inner
 * (:126-127); idx = occ_grid(v+dv, u+du) (:130), -1 = empty; dist = SPMatcher::DescriptorDistance =
 * (float)cv::norm(a, b, NORM_L2) (sp_matcher.cpp:1636-1640); strict `<` keeps the first minimum
 * (:134); a match clears its cell (:167).  The reference reads the grid without a bounds check;
 * here cells outside it are empty.
 * cv::norm accumulates the squared float differences in double; its order inside a row is an
 * OpenCV build detail — fixed here as: lane l of 64 adds dims 4l..4l+3 in order, then the butterfly
 * of spfe_sum64_host (what one wavefront computes).  kp_idx[i] = matched keypoint or -1.
 */
static float patch_dist(const float *a, const float *b) {
  double v[64];
  for (int l = 0; l < 64; ++l) {
    double s = 0.0;
    for (int q = 0; q < 4; ++q) {
      const float d = a[4 * l + q] - b[4 * l + q];
      if (q == 0) s = (double)d * (double)d;
      else s = s + (double)d * (double)d;
    }
    v[l] = s;
  }
  for (int off = 32; off >= 1; off >>= 1) {
    double nv[64];
    for (int i = 0; i < 64; ++i) nv[i] = v[i] + v[i ^ off];
    memcpy(v, nv, sizeof(v));
  }
  return (float)sqrt(v[0]);
}

EXPORT void oracle_match_patches(const float *mp_desc, const float *mp_uv, int n_points, const int16_t *occ_grid,
                                 int hc, int wc, const float *kp_desc, int n_keypoints, float max_dist,
                                 int32_t *kp_idx) {
  int16_t *occ = (int16_t *)malloc((size_t)hc * wc * sizeof(int16_t));
  memcpy(occ, occ_grid, (size_t)hc * wc * sizeof(int16_t));
  for (int i = 0; i < n_points; ++i) {
    kp_idx[i] = -1;
    const float fu = floorf(mp_uv[2 * i]), fv = floorf(mp_uv[2 * i + 1]);
    if (!(fu >= 0.0f && fv >= 0.0f && fu < (float)wc && fv < (float)hc)) continue;
    const int u = (int)fu, v = (int)fv;
    int best = -1, bu = 0, bv = 0;
    float bd = max_dist;
    for (int du = 0; du < 2; ++du)
      for (int dv = 0; dv < 2; ++dv) {
        const int uu = u + du, vv = v + dv;
        if (uu >= wc || vv >= hc) continue;
        const int idx = occ[vv * wc + uu];
        if (idx < 0 || idx >= n_keypoints) continue;
        const float d = patch_dist(mp_desc + (size_t)i * 256, kp_desc + (size_t)idx * 256);
        if (d < bd) {
          bd = d;
          best = idx;
          bu = uu;
          bv = vv;
        }
      }
    if (best != -1) {
      kp_idx[i] = best;
      occ[bv * wc + bu] = -1;
    }
  }
  free(occ);
}

/*
 * SURVEY.md §8(f) rank 2 — input staging.  Restates, for ONE frame, the host OpenCV sequence the
 * reference runs in front of the extractor:
 *   cv::remap(mono, mono, m1, m2, cv::INTER_LINEAR)      orb_slam2/src/io/data_loader.cc:519-521
 *   mono(cv::Rect(0, 0, camera::width, camera::height))  orb_slam2/src/system.cpp:160-161
 *   cvtColor(.., CV_BGR2GRAY | CV_RGB2GRAY | ..A2GRAY)    orb_slam2/src/tracking/mono_tracker.cpp:18-28
 * OpenCV 3.x (not vendored, not installed — PARITY UNPINNED; published algorithm restated):
 *  - cv::remap with CV_32FC1 maps, INTER_LINEAR, 8-bit source: the maps are converted to fixed
 *    point with INTER_BITS = 5: sx = cvRound(mx * 32), sy = cvRound(my * 32) (round half to even),
 *    integer part saturate_cast<short>(s >> 5), fraction s & 31; remapBilinear with the 15-bit
 *    weight table BilinearTab_i (initInterTab2D: w = saturate_cast<short>(wy * wx * 32768), and
 *    when the four do not sum to 32768 — only the (0,0) entry, whose 32768 saturates to 32767 —
 *    the difference is added at table index [ksize/2][ksize/2] = the LAST tap, giving
 *    {32767, 0, 0, 1}); D = saturate_cast<uchar>((sum + (1 << 14)) >> 15); BORDER_CONSTANT with
 *    value 0: taps outside the source contribute 0.
 *  - cvtColor 8-bit to gray (3.0 - 3.4.1, yuv_shift = 14): (B*1868 + G*9617 + R*4899 + 8192) >> 14;
 *    alpha ignored.  (Later OpenCV uses 15-bit coefficients; identical for gray-valued BGR input
 *    such as EuRoC's, since both sets sum to one.)
 * src: [src_h] rows of src_stride bytes, `channels` interleaved u8; map_x/map_y: [src_h][src_w]
 * or NULL; gray: [H][W].  Returns 0, or -1 on bad geometry.
 */
EXPORT int oracle_stage_input(const uint8_t *src, int src_h, int src_w, int src_stride, int channels, int rgb,
                              const float *map_x, const float *map_y, int H, int W, uint8_t *gray) {
  if (H > src_h || W > src_w || (channels != 1 && channels != 3 && channels != 4)) return -1;
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      int ch[4] = {0, 0, 0, 0};
      if (map_x) {
        const int sx = (int)lrintf(map_x[(size_t)y * src_w + x] * 32.0f);
        const int sy = (int)lrintf(map_y[(size_t)y * src_w + x] * 32.0f);
        int ix = sx >> 5, iy = sy >> 5;
        ix = ix < -32768 ? -32768 : (ix > 32767 ? 32767 : ix);
        iy = iy < -32768 ? -32768 : (iy > 32767 ? 32767 : iy);
        const int fx = sx & 31, fy = sy & 31;
        int w[4] = {(32 - fy) * (32 - fx) * 32, (32 - fy) * fx * 32, fy * (32 - fx) * 32, fy * fx * 32};
        if (fx == 0 && fy == 0) {
          w[0] = 32767;
          w[3] = 1;
        }
        for (int k = 0; k < channels; ++k) {
          int sum = 0;
          for (int t = 0; t < 4; ++t) {
            const int px = ix + (t & 1), py = iy + (t >> 1);
            const int v = (px >= 0 && px < src_w && py >= 0 && py < src_h)
                              ? src[(size_t)py * src_stride + (size_t)px * channels + k]
                              : 0;
            sum += v * w[t];
          }
          int v = (sum + (1 << 14)) >> 15;
          ch[k] = v < 0 ? 0 : (v > 255 ? 255 : v);
        }
      } else {
        for (int k = 0; k < channels; ++k) ch[k] = src[(size_t)y * src_stride + (size_t)x * channels + k];
      }
      int g;
      if (channels == 1) {
        g = ch[0];
      } else {
        const int bl = rgb ? ch[2] : ch[0], rd = rgb ? ch[0] : ch[2];
        g = (bl * 1868 + ch[1] * 9617 + rd * 4899 + (1 << 13)) >> 14;
      }
      gray[(size_t)y * W + x] = (uint8_t)g;
    }
  return 0;
}

/* OpenMP team size for the timing legs (bench.py cpu_baseline, tools/cpu_baseline.py): the loops
 * scale to a few dozen threads, far fewer than a 256-thread host offers. */
#ifdef _OPENMP
#include <omp.h>
EXPORT void oracle_set_num_threads(int n) { if (n > 0) omp_set_num_threads(n); }
EXPORT int oracle_get_max_threads(void) { return omp_get_max_threads(); }
#else
EXPORT void oracle_set_num_threads(int n) { (void)n; }
EXPORT int oracle_get_max_threads(void) { return 1; }
#endif

/* exact-math probes so GPU tests can compare device bits with host bits */
EXPORT float oracle_expf(float x) { return spfe_expf(x); }
EXPORT float oracle_logf(float x) { return spfe_logf(x); }
EXPORT float oracle_sum256(const float *v) { return spfe_sum256_host(v); }

/* ---- direct "dust" alignment -------------------------------------------------
 * Optimizer::PoseOptimizationDust(Frame*, const vector<MapPoint*>&, vector<bool>&)
 *   /root/reference/orb_slam2/src/mapping/optimizer_dust.cpp:170-294
 * with g2o::EdgeSE3ProjectDustOnlyPose (src/optimization/types_dust_tracking.cpp:37-140) and g2o's
 * Levenberg-Marquardt driver (SparseOptimizer::optimize(40), OptimizationAlgorithmLevenberg::solve,
 * BlockSolver_6_3 + LinearSolverDense on the single 6x6 pose block, RobustKernelHuber(0.9)).  g2o is a
 * catkin dependency that is NOT in /root/reference: PARITY UNPINNED — its published algorithm is restated
 * (arithmetic in include/spfe_dust_math.h, the control flow here).  The edges are visited in g2o's order; their
 * contributions to chi2 / H / b are summed by the fixed-shape tree of include/spfe_dust_math.h (round 5: the shape the
 * kernel evaluates in log time; before, a strict edge-order chain — neither is pinned by anything in /root/reference).
 *
 * dust [hc][wc] = Frame::dust_ (dense_dust_ of the extractor); pts [n][3] = MapPoint::GetWorldPos() floats;
 * Tcw_in / Tcw_out = Frame::mTcw, CV_32F 4x4 row-major; fx..cy = Frame::fx.. (floats, full resolution: the
 * /8 and -3.5 of :223-226 are applied here).  Outputs per map point: inlier (is_visible / in_view, :262-266),
 * uv = dust_proj_u / dust_proj_v (:267-268; only meaningful for inliers).  Returns n_inlier (:258-270). */
#include "../include/spfe_dust_math.h"

/* the fixed-shape tree of include/spfe_dust_math.h over quantities q0 .. q0 + nq - 1 of the per-edge terms */
static void dust_tree_sums(const double *terms /* [n][28] */, int n, int q0, int nq, double *out) {
  for (int q = q0; q < q0 + nq; ++q) {
    double slot[SPFE_DUST_SLOTS];
    for (int t = 0; t < SPFE_DUST_SLOTS; ++t) {
      slot[t] = 0.0;
      for (int i = t; i < n; i += SPFE_DUST_SLOTS) slot[t] += terms[(size_t)i * SPFE_DUST_NSUM + q];
    }
    out[q] = spfe_dust_tree_total(slot);
  }
}

static double dust_errors(const spfe_se3 *T, const float *pts, int n, double fx, double fy, double cx, double cy,
                          const float *dust, int wc, int hc, double delta, spfe_dust_edge *ed, double *terms) {
  /* computeActiveErrors + activeRobustChi2: rho[0] of every edge, summed by the contract's tree */
  const double J0[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const double Xw[3] = {(double)pts[3 * i], (double)pts[3 * i + 1], (double)pts[3 * i + 2]};
    spfe_dust_error(T, Xw, fx, fy, cx, cy, dust, wc, hc, &ed[i]);
    spfe_dust_terms(ed[i].err, J0, delta, terms + (size_t)i * SPFE_DUST_NSUM);
  }
  double tot[SPFE_DUST_NSUM];
  dust_tree_sums(terms, n, 0, 1, tot);
  return tot[0];
}

static int align_dust_core(const float *dust, int hc, int wc, const float *pts, int n, const float *Tcw_in,
                           float fxf, float fyf, float cxf, float cyf, int max_iterations, double delta,
                           double inlier_chi2, float *Tcw_out, uint8_t *inlier, float *uv, int *iterations,
                           double *pose64) {
  const double fx = (double)(fxf / 8.0f), fy = (double)(fyf / 8.0f);           /* :223-224 */
  const double cx = ((double)cxf - 3.5) / 8.0f, cy = ((double)cyf - 3.5) / 8.0f; /* :225-226 */
  spfe_se3 T;
  spfe_se3_from_f32(Tcw_in, &T);
  spfe_dust_edge *ed = (spfe_dust_edge *)calloc((size_t)(n > 0 ? n : 1), sizeof(spfe_dust_edge));
  double *terms = (double *)calloc((size_t)(n > 0 ? n : 1) * SPFE_DUST_NSUM, sizeof(double));
  spfe_lm lm = {0.0, 2.0};
  int it_done = 0, ok = 1;
  for (int it = 0; it < max_iterations && ok && n > 0; ++it) {   /* no edges: nothing active, the pose is echoed */
    double currentChi = dust_errors(&T, pts, n, fx, fy, cx, cy, dust, wc, hc, delta, ed, terms);
    /* buildSystem: linearizeOplus + constructQuadraticForm of every edge, summed by the contract's tree */
    double H[36], b[6], tot[SPFE_DUST_NSUM], chi_again;
    for (int i = 0; i < n; ++i) {
      const double Xw[3] = {(double)pts[3 * i], (double)pts[3 * i + 1], (double)pts[3 * i + 2]};
      double J[6];
      spfe_dust_jacobian(&T, Xw, fx, fy, cx, cy, dust, wc, hc, ed[i].level, J);
      spfe_dust_terms(ed[i].err, J, delta, terms + (size_t)i * SPFE_DUST_NSUM);
    }
    dust_tree_sums(terms, n, 0, SPFE_DUST_NSUM, tot);
    spfe_dust_unpack(tot, &chi_again, H, b);
    (void)chi_again;   /* q[0] of the build pass is the same rho0: the same tree, the same bits as currentChi */
    if (it == 0) {
      double maxDiagonal = 0;
      for (int j = 0; j < 6; ++j) maxDiagonal = fabs(H[j * 6 + j]) > maxDiagonal ? fabs(H[j * 6 + j]) : maxDiagonal;
      lm.lambda = SPFE_LM_TAU * maxDiagonal;
      lm.ni = 2;
    }
    double rho = 0;
    int qmax = 0;
    do {
      const spfe_se3 saved = T;                               /* _optimizer->push() */
      double x[6] = {0, 0, 0, 0, 0, 0};
      const int ok2 = spfe_solve6(H, lm.lambda, b, x);
      if (ok2) spfe_se3_oplus(&T, x);
      double tempChi = dust_errors(&T, pts, n, fx, fy, cx, cy, dust, wc, hc, delta, ed, terms);
      if (!ok2) tempChi = DBL_MAX;
      if (spfe_lm_judge(&lm, currentChi, tempChi, x, b, &rho)) currentChi = tempChi;   /* discardTop */
      else T = saved;                                                                    /* pop */
      qmax++;
    } while (rho < 0 && qmax < SPFE_LM_MAX_TRIALS);
    it_done++;
    if (qmax == SPFE_LM_MAX_TRIALS || rho == 0) ok = 0;       /* Terminate */
  }
  /* the edges hold the errors of the LAST evaluation (a rejected trial's, if the run ended on one) */
  int n_inlier = n;
  for (int i = 0; i < n; ++i) {
    const int out = ed[i].level == 1 || ed[i].err * ed[i].err > inlier_chi2;
    if (inlier) inlier[i] = out ? 0 : 1;
    if (uv) { uv[2 * i] = ed[i].u; uv[2 * i + 1] = ed[i].v; }
    if (out) n_inlier--;
  }
  spfe_se3_to_f32(&T, Tcw_out);
  if (pose64) {   /* SE3Quat::to_homogeneous_matrix() before the cast to float */
    double R[9];
    spfe_quat_to_rot(T.q, R);
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) pose64[r * 4 + c] = R[r * 3 + c];
      pose64[r * 4 + 3] = T.t[r];
    }
    pose64[12] = pose64[13] = pose64[14] = 0.0;
    pose64[15] = 1.0;
  }
  if (iterations) *iterations = it_done;
  free(ed);
  free(terms);
  return n_inlier;
}

EXPORT int oracle_align_dust(const float *dust, int hc, int wc, const float *pts, int n, const float *Tcw_in,
                             float fxf, float fyf, float cxf, float cyf, int max_iterations, double delta,
                             double inlier_chi2, float *Tcw_out, uint8_t *inlier, float *uv, int *iterations) {
  return align_dust_core(dust, hc, wc, pts, n, Tcw_in, fxf, fyf, cxf, cyf, max_iterations, delta, inlier_chi2, Tcw_out,
                         inlier, uv, iterations, NULL);
}

/* The same, also returning the pose in double precision (4x4 row-major, what Converter::toCvMat rounds to float):
 * lets tests/test_dust_golden.py hold the solve to the independent f64 fixtures at 1e-9 instead of float ulps. */
EXPORT int oracle_align_dust_pose64(const float *dust, int hc, int wc, const float *pts, int n, const float *Tcw_in,
                                    float fxf, float fyf, float cxf, float cyf, int max_iterations, double delta,
                                    double inlier_chi2, float *Tcw_out, uint8_t *inlier, float *uv, int *iterations,
                                    double *pose64) {
  return align_dust_core(dust, hc, wc, pts, n, Tcw_in, fxf, fyf, cxf, cyf, max_iterations, delta, inlier_chi2, Tcw_out,
                         inlier, uv, iterations, pose64);
}

/* One edge at the pose Tcw (float 4x4, converted as toSE3Quat does), optionally moved first by oplus(update):
 * computeError then linearizeOplus (types_dust_tracking.cpp:64-140).  Outputs: err, level, J[6] (1x6, rotation
 * columns first), uv[2] (u_, v_).  Test hook: tests/test_dust_golden.py checks spfe_dust_error / spfe_dust_jacobian /
 * spfe_se3_oplus of include/spfe_dust_math.h against the independent fixtures and against numeric derivatives. */
EXPORT void oracle_dust_edge(const float *dust, int hc, int wc, const float *Xw_f, const float *Tcw, const double *update,
                             float fxf, float fyf, float cxf, float cyf, double *err, int *level, double *J, float *uv,
                             double *pose64) {
  const double fx = (double)(fxf / 8.0f), fy = (double)(fyf / 8.0f);
  const double cx = ((double)cxf - 3.5) / 8.0f, cy = ((double)cyf - 3.5) / 8.0f;
  spfe_se3 T;
  spfe_se3_from_f32(Tcw, &T);
  if (update) spfe_se3_oplus(&T, update);
  const double Xw[3] = {(double)Xw_f[0], (double)Xw_f[1], (double)Xw_f[2]};
  spfe_dust_edge e = {0.0, 0.0f, 0.0f, 0};
  spfe_dust_error(&T, Xw, fx, fy, cx, cy, dust, wc, hc, &e);
  spfe_dust_jacobian(&T, Xw, fx, fy, cx, cy, dust, wc, hc, e.level, J);
  *err = e.err;
  *level = e.level;
  uv[0] = e.u;
  uv[1] = e.v;
  if (pose64) {
    double R[9];
    spfe_quat_to_rot(T.q, R);
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) pose64[r * 4 + c] = R[r * 3 + c];
      pose64[r * 4 + 3] = T.t[r];
    }
    pose64[12] = pose64[13] = pose64[14] = 0.0;
    pose64[15] = 1.0;
  }
}

/* k = 2 nearest train rows per query — the EXACT search that the reference's
 * flann->knnMatch(query, matches, 2) approximates (keyframe.cpp:447-470, sp_matcher.cpp:195-215, :264-280;
 * cv::FlannBasedMatcher with randomised kd-trees: no bit-parity target).  OpenCV's brute-force k-NN
 * (batchDistance, K = 2) is the statement followed: a sorted insertion with strict `<`, so the earlier train
 * row stays in front on ties; NaN / infinite distances never enter.  train_idx / distance: [nq][2]. */
EXPORT void oracle_match_knn2(const float *query, int nq, const float *train, int nt, int32_t *train_idx,
                              float *distance) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < nq; ++i) {
    float d0 = FLT_MAX, d1 = FLT_MAX;
    int i0 = -1, i1 = -1;
    for (int j = 0; j < nt; ++j) {
      const float d = match_dist(query + (size_t)i * 256, train + (size_t)j * 256);
      if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = j; }
      else if (d < d1) { d1 = d; i1 = j; }
    }
    train_idx[2 * i] = i0; train_idx[2 * i + 1] = i1;
    distance[2 * i] = d0; distance[2 * i + 1] = d1;
  }
}
