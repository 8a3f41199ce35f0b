// conv_bf16.hip — bf16-input / f32-accumulate 3x3 convolutions for gfx950 (MI355X):
// the "bf16 conv path with fp32 NMS" of BASELINE.json configs[3].
//
// Same job as conv_f32.hip (SPFrontend::forward, /root/reference/orb_slam2/src/cv/
// sp_extractor.cpp:81-100) at 16x the matrix rate: v_mfma_f32_32x32x16_bf16
// (2.5 PFLOP/s dense).  Activations live in HBM as NHWC bf16 (half the bytes),
// accumulation, bias, ReLU and max-pool are f32, outputs are rounded to bf16
// (round-to-nearest-even) — except convPa/convDa, which write f32 so that the
// 1x1 heads, the detector tail, NMS, descriptors and covariance stay f32.
//
// Structure = the persistent, double-buffered, everything-in-the-MFMA-shadow
// pipeline of conv_f32.hip, with what bf16 changes:
//  * K chunk = 32 channels; a K step = (tap, 16 channels) = MT x NT MFMAs of 32
//    cycles, so the side work is sliced even finer (one load, one LDS write per step);
//  * the MFMA fragments want 8 consecutive channels per lane, which IS the NHWC
//    order: the halo tile sits in LDS pixel-major ([row][col][32 ch]) and staging is
//    one ds_write_b128 per 16-byte global piece — no transposition;
//  * pixel / weight rows are padded from 64 to 80 bytes: 16 lanes reading
//    consecutive pixels at the same channel offset then hit 16 different 16-byte
//    bank slots (5 is coprime to 16) — conflict-free ds_read_b128 without an XOR
//    swizzle, and every fragment address is base + immediate.
#include <utility>

#include "spfe_kernels.h"

namespace spfe {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define SPFE_OOB 0x80000000u

__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f) {
  unsigned u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);  // finite inputs only (activations after ReLU / conv sums)
  return (unsigned short)(u >> 16);
}

constexpr int BKC = 32;       // channels per K chunk
constexpr int BPITCH = 80;    // bytes per pixel / weight row in LDS and in the packed slabs (64 + 16 pad)

template <int TH>
struct GeoB {
  static constexpr int ROWS = TH + 2, COLS = 34;
  static constexpr int A_BYTES = ROWS * COLS * BPITCH;
  static constexpr int W_BYTES = 12 * 256 * 16;  // 9 * 64 * BPITCH = 46080, padded to whole 256-thread passes
  static constexpr int BUF_BYTES = A_BYTES + W_BYTES;
};

template <int NITER, int NWITER>
struct PipeB {
  i32x4 va[NITER], vw[NWITER];
  unsigned dst[NITER];     // LDS byte offset of each input piece (dummy slot for unused pieces)
  unsigned voff[NITER];    // byte offset inside the input frame, or SPFE_OOB
  unsigned woff[NWITER];   // byte offset inside the weight slab (== LDS offset)
  __amdgpu_buffer_rsrc_t rin, rw;
  const char *aBase, *bBase;  // this stage's operands (LDS)
  char *nA, *nW;              // the other LDS buffer
};

template <int NT>
struct EpiB {
  __amdgpu_buffer_rsrc_t rout;
  unsigned obase[NT];
  float bias[NT];
  int xlim, ylim;
  unsigned rowstep, pixstep;
};

template <int MT, int NT, bool POOL, bool OUT_F32, int E>
__device__ __forceinline__ void epi_store_b(const EpiB<NT> &e, const f32x16 (&acc)[MT][NT]) {
  constexpr int NEPI_ = POOL ? NT * 8 : MT * NT * 16;
  if constexpr (E >= NEPI_) {
    return;
  } else {
    float v;
    unsigned off;
    if constexpr (!POOL) {
      constexpr int j = E / (MT * 16), i = (E / 16) % MT, r = E % 16;
      constexpr int xr = (r & 3) + 8 * (r >> 2);
      v = acc[i][j][r] + e.bias[j];
      v = v > 0.0f ? v : 0.0f;
      off = (xr < e.xlim && i < e.ylim) ? e.obase[j] + i * e.rowstep + xr * e.pixstep : SPFE_OOB;
    } else {
      constexpr int j = E / 8, r = 2 * (E % 8);
      constexpr int xr = (r & 3) + 8 * (r >> 2);
      float v00 = acc[0][j][r] + e.bias[j], v01 = acc[0][j][r + 1] + e.bias[j];
      float v10 = acc[1][j][r] + e.bias[j], v11 = acc[1][j][r + 1] + e.bias[j];
      v00 = v00 > 0.0f ? v00 : 0.0f;
      v01 = v01 > 0.0f ? v01 : 0.0f;
      v10 = v10 > 0.0f ? v10 : 0.0f;
      v11 = v11 > 0.0f ? v11 : 0.0f;
      const float m0 = v00 > v01 ? v00 : v01;
      const float m1 = v10 > v11 ? v10 : v11;
      v = m0 > m1 ? m0 : m1;
      off = (xr < e.xlim && 0 < e.ylim) ? e.obase[j] + (xr >> 1) * e.pixstep : SPFE_OOB;
    }
    if constexpr (OUT_F32) __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(v), e.rout, off, 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b16((short)f32_to_bf16_rne(v), e.rout, off, 0, 0);
  }
}

template <int STEP, int NSTEP, bool FIRST, int MT, int NT, int NITER, int NWITER, bool POOL, bool OUT_F32>
__device__ __forceinline__ void k_steps_b(bf16x8 (&a)[2][MT], bf16x8 (&bb)[2][NT], f32x16 (&acc)[MT][NT],
                                          const f32x16 (&accPrev)[MT][NT], PipeB<NITER, NWITER> &c,
                                          const EpiB<NT> &e) {
  if constexpr (STEP < NSTEP) {
    constexpr int NLD = NITER + NWITER;
    constexpr int NEPI = POOL ? NT * 8 : MT * NT * 16;
    constexpr int EPS = (NEPI + (NSTEP - 2)) / (NSTEP - 1);
    constexpr int cur = STEP & 1, nxt = cur ^ 1;
    // loads of the next stage: one per step from step 0; LDS writes: one per step, NLD steps
    // later (the stage has NSTEP >= NLD steps; a write for load k sits at step k + NSTEP - NLD ... )
    constexpr int M = MT * NT;
#pragma unroll
    for (int m = 0; m < M; ++m) {
      if (m == 0) {  // (1) fragments of the next step
        if constexpr (STEP + 1 < NSTEP) {
          constexpr int tap = (STEP + 1) / 2, kk = (STEP + 1) % 2;
          constexpr int dy = tap / 3, dx = tap % 3;
#pragma unroll
          for (int i = 0; i < MT; ++i)
            a[nxt][i] = *reinterpret_cast<const bf16x8 *>(c.aBase + ((i + dy) * 34 + dx) * BPITCH + kk * 32);
#pragma unroll
          for (int j = 0; j < NT; ++j)
            bb[nxt][j] = *reinterpret_cast<const bf16x8 *>(c.bBase + (tap * 64 + j * 32) * BPITCH + kk * 32);
        }
      }
      if (m == 1 % M) {  // (2) global loads of the next stage: two per step in the first steps
        if constexpr (STEP * 2 < NLD) {
          constexpr int it = STEP * 2;
          if constexpr (it < NITER) c.va[it] = __builtin_amdgcn_raw_buffer_load_b128(c.rin, c.voff[it], 0, 0);
          else c.vw[it - NITER] = __builtin_amdgcn_raw_buffer_load_b128(c.rw, c.woff[it - NITER], 0, 0);
        }
        if constexpr (STEP * 2 + 1 < NLD) {
          constexpr int it = STEP * 2 + 1;
          if constexpr (it < NITER) c.va[it] = __builtin_amdgcn_raw_buffer_load_b128(c.rin, c.voff[it], 0, 0);
          else c.vw[it - NITER] = __builtin_amdgcn_raw_buffer_load_b128(c.rw, c.woff[it - NITER], 0, 0);
        }
      }
      if (m == 2 % M) {  // (3) a slice of the previous tile's epilogue
        if constexpr (FIRST && STEP >= 1) {
          constexpr int e0 = (STEP - 1) * EPS;
          [&]<int... Qs>(std::integer_sequence<int, Qs...>) {
            (epi_store_b<MT, NT, POOL, OUT_F32, e0 + Qs>(e, accPrev), ...);
          }(std::make_integer_sequence<int, EPS>{});
        }
      }
      if (m == 3 % M) {  // (4) staged pieces into the other LDS buffer: two per step in the last steps
        constexpr int W0 = NSTEP - (NLD + 1) / 2;
        if constexpr (STEP >= W0) {
          constexpr int it0 = (STEP - W0) * 2;
          if constexpr (it0 < NLD) {
            if constexpr (it0 < NITER) *reinterpret_cast<i32x4 *>(c.nA + c.dst[it0]) = c.va[it0];
            else *reinterpret_cast<i32x4 *>(c.nW + c.woff[it0 - NITER]) = c.vw[it0 - NITER];
          }
          if constexpr (it0 + 1 < NLD) {
            constexpr int it1 = it0 + 1;
            if constexpr (it1 < NITER) *reinterpret_cast<i32x4 *>(c.nA + c.dst[it1]) = c.va[it1];
            else *reinterpret_cast<i32x4 *>(c.nW + c.woff[it1 - NITER]) = c.vw[it1 - NITER];
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      {
        const int i = m / NT, j = m % NT;
        if constexpr (FIRST && STEP == 0) {
          f32x16 z;
#pragma unroll
          for (int r = 0; r < 16; ++r) z[r] = 0.0f;
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][i], bb[cur][j], z, 0, 0, 0);
        } else {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][i], bb[cur][j], acc[i][j], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    k_steps_b<STEP + 1, NSTEP, FIRST, MT, NT, NITER, NWITER, POOL, OUT_F32>(a, bb, acc, accPrev, c, e);
  }
}

// in: NHWC bf16 [B][H][W][in_stride]; wpack: bf16 slabs [nblk][chunk][tap][64 n][BPITCH bytes];
// out: NHWC bf16 (or f32 when OUT_F32) — strides in ConvParams are in ELEMENTS.
template <int CIN, bool POOL, bool OUT_F32>
__global__ __launch_bounds__(256, 1) void conv_bf16_kernel(ConvParams p) {
  constexpr int WM = 4, MT = 2, NT = 2, TH = WM * MT;
  using G = GeoB<TH>;
  constexpr int NCHUNK = CIN / BKC;
  constexpr int NITEM = G::ROWS * G::COLS * 4;  // 16-byte pieces of the halo tile (4 per pixel)
  constexpr int NITER = (NITEM + 255) / 256;
  constexpr int NW16 = G::W_BYTES / 16;
  constexpr int NWITER = (NW16 + 255) / 256;
  constexpr int NSTEP = 9 * (BKC / 16);
  static_assert((NITER + NWITER + 1) / 2 <= NSTEP / 2, "staging does not fit the K steps");

  extern __shared__ __attribute__((aligned(16))) char smem_b[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int wm = wave;
  const int H = p.H, W = p.W;

  const int total = p.nblk * p.tiles_x * p.tiles_y * p.B;
  const int xcd = blockIdx.x & 7, gi = blockIdx.x >> 3, gper = gridDim.x >> 3;
  const int lo = (int)((long)total * xcd / 8), hi_w = (int)((long)total * (xcd + 1) / 8);
  int w = lo + gi;
  if (w >= hi_w) return;

  int i_nb, i_tx, i_ty, i_b;
  {
    int t = w;
    i_nb = t % p.nblk; t /= p.nblk;
    i_tx = t % p.tiles_x; t /= p.tiles_x;
    i_ty = t % p.tiles_y; i_b = t / p.tiles_y;
  }
  int d_nb, d_tx, d_ty, d_b;
  {
    int t = gper;
    d_nb = t % p.nblk; t /= p.nblk;
    d_tx = t % p.tiles_x; t /= p.tiles_x;
    d_ty = t % p.tiles_y; d_b = t / p.tiles_y;
  }

  const unsigned in_pix_bytes = (unsigned)p.in_stride * 2u;
  const unsigned frame_in_bytes = (unsigned)H * W * in_pix_bytes;
  const int Ho = POOL ? H >> 1 : H, Wo = POOL ? W >> 1 : W;
  constexpr unsigned OEL = OUT_F32 ? 4u : 2u;
  const unsigned out_pix_bytes = (unsigned)p.out_stride * OEL;
  const unsigned frame_out_bytes = (unsigned)Ho * Wo * out_pix_bytes;

  PipeB<NITER, NWITER> c;
  int prow[NITER], pcol[NITER];
  unsigned pqb[NITER];
#pragma unroll
  for (int it = 0; it < NITER; ++it) {
    const int i = tid + it * 256;
    const int qq = i & 3, pix = i >> 2;
    prow[it] = i < NITEM ? pix / G::COLS - 1 : (1 << 20);
    pcol[it] = pix % G::COLS - 1;
    pqb[it] = qq * 16;
    // unused piece: the 16 pad bytes of some pixel (never read)
    c.dst[it] = i < NITEM ? (unsigned)(pix * BPITCH + qq * 16) : (unsigned)((tid % (G::ROWS * G::COLS)) * BPITCH + 64);
  }
  static_assert(NW16 % 256 == 0, "padded weight slab is a whole number of passes");
#pragma unroll
  for (int it = 0; it < NWITER; ++it) c.woff[it] = (unsigned)(tid + it * 256) * 16u;

  auto aim_tile = [&](int tx, int ty) {
#pragma unroll
    for (int it = 0; it < NITER; ++it) {
      const int gy = ty * TH + prow[it], gx = tx * 32 + pcol[it];
      c.voff[it] = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
                       ? (unsigned)(gy * W + gx) * in_pix_bytes + pqb[it]
                       : SPFE_OOB;
    }
  };
  auto aim_stage = [&](int nb, int b, int chunk, bool valid) {
    const char *base = reinterpret_cast<const char *>(p.in) +
                       ((size_t)b * H * W * p.in_stride + p.in_choff + chunk * BKC) * 2;
    c.rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, valid ? frame_in_bytes : 0u, 0x00020000);
    const char *wb = reinterpret_cast<const char *>(p.wpack) + ((size_t)nb * NCHUNK + chunk) * G::W_BYTES;
    c.rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(wb), 0, valid ? (unsigned)G::W_BYTES : 0u, 0x00020000);
  };

  // prologue: first stage straight into buffer 0
  aim_tile(i_tx, i_ty);
  aim_stage(i_nb, i_b, 0, true);
#pragma unroll
  for (int it = 0; it < NITER; ++it) c.va[it] = __builtin_amdgcn_raw_buffer_load_b128(c.rin, c.voff[it], 0, 0);
#pragma unroll
  for (int it = 0; it < NWITER; ++it) c.vw[it] = __builtin_amdgcn_raw_buffer_load_b128(c.rw, c.woff[it], 0, 0);
#pragma unroll
  for (int it = 0; it < NITER; ++it) *reinterpret_cast<i32x4 *>(smem_b + c.dst[it]) = c.va[it];
#pragma unroll
  for (int it = 0; it < NWITER; ++it) *reinterpret_cast<i32x4 *>(smem_b + G::A_BYTES + c.woff[it]) = c.vw[it];
  __syncthreads();

  f32x16 accA[MT][NT], accB[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { accA[i][j][r] = 0.0f; accB[i][j][r] = 0.0f; }

  int buf = 0;
  EpiB<NT> epi, epi_next;
  epi.rout = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, 0u, 0x00020000);
#pragma unroll
  for (int j = 0; j < NT; ++j) { epi.obase[j] = SPFE_OOB; epi.bias[j] = 0.0f; }
  epi.xlim = 0; epi.ylim = 0; epi.rowstep = 0; epi.pixstep = out_pix_bytes;
  bool more = true;

  auto aim_epi = [&](EpiB<NT> &e, int nb, int tx, int ty, int b) {
    char *obase = reinterpret_cast<char *>(p.out) + ((size_t)b * Ho * Wo * p.out_stride + p.out_choff) * OEL;
    e.rout = __builtin_amdgcn_make_buffer_rsrc(obase, 0, frame_out_bytes, 0x00020000);
    const int y0 = ty * TH + wm * MT, x0 = tx * 32 + 4 * hi;
    e.xlim = W - x0;
    e.ylim = H - y0;
    e.rowstep = (unsigned)Wo * out_pix_bytes;
    e.pixstep = out_pix_bytes;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int co = nb * 64 + j * 32 + l31;
      const unsigned pix = POOL ? (unsigned)((y0 >> 1) * Wo + (x0 >> 1)) : (unsigned)(y0 * W + x0);
      e.obase[j] = co < p.cout_real ? pix * out_pix_bytes + (unsigned)co * OEL : SPFE_OOB;
      e.bias[j] = p.bias[co];
    }
  };

  auto run_tile = [&](f32x16(&acc)[MT][NT], const f32x16(&accPrev)[MT][NT]) {
    int n_nb = i_nb + d_nb, n_tx = i_tx + d_tx, n_ty = i_ty + d_ty, n_b = i_b + d_b;
    if (n_nb >= p.nblk) { n_nb -= p.nblk; ++n_tx; }
    if (n_tx >= p.tiles_x) { n_tx -= p.tiles_x; ++n_ty; }
    if (n_ty >= p.tiles_y) { n_ty -= p.tiles_y; ++n_b; }
    const bool have_next_item = w + gper < hi_w;
    aim_epi(epi_next, i_nb, i_tx, i_ty, i_b);
#pragma unroll 1
    for (int chunk = 0; chunk < NCHUNK; ++chunk) {
      const bool last = chunk == NCHUNK - 1;
      if (!last) {
        aim_stage(i_nb, i_b, chunk + 1, true);
      } else {
        aim_tile(n_tx, n_ty);
        aim_stage(n_nb, n_b, 0, have_next_item);
      }
      const char *cA = smem_b + buf * G::BUF_BYTES;
      c.nA = smem_b + (buf ^ 1) * G::BUF_BYTES;
      c.nW = c.nA + G::A_BYTES;
      c.aBase = cA + ((wm * MT) * 34 + l31) * BPITCH + hi * 16;
      c.bBase = cA + G::A_BYTES + l31 * BPITCH + hi * 16;
      bf16x8 a[2][MT], bb[2][NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) a[0][i] = *reinterpret_cast<const bf16x8 *>(c.aBase + (i * 34) * BPITCH);
#pragma unroll
      for (int j = 0; j < NT; ++j) bb[0][j] = *reinterpret_cast<const bf16x8 *>(c.bBase + (j * 32) * BPITCH);
      if (chunk == 0) k_steps_b<0, NSTEP, true, MT, NT, NITER, NWITER, POOL, OUT_F32>(a, bb, acc, accPrev, c, epi);
      else k_steps_b<0, NSTEP, false, MT, NT, NITER, NWITER, POOL, OUT_F32>(a, bb, acc, accPrev, c, epi);
      __syncthreads();
      buf ^= 1;
    }
    epi = epi_next;
    more = have_next_item;
    w += gper;
    i_nb = n_nb; i_tx = n_tx; i_ty = n_ty; i_b = n_b;
  };

  bool lastA = true;
  while (true) {
    run_tile(accA, accB);
    lastA = true;
    if (!more) break;
    run_tile(accB, accA);
    lastA = false;
    if (!more) break;
  }
  {
    constexpr int NEPI = POOL ? NT * 8 : MT * NT * 16;
    auto flush = [&](const f32x16(&acc)[MT][NT]) {
      [&]<int... E>(std::integer_sequence<int, E...>) {
        (epi_store_b<MT, NT, POOL, OUT_F32, E>(epi, acc), ...);
      }(std::make_integer_sequence<int, NEPI>{});
    };
    if (lastA) flush(accA); else flush(accB);
  }
}

template <int CIN, bool POOL, bool OUT_F32>
static hipError_t launch_b(const ConvParams &p, hipStream_t s) {
  using G = GeoB<8>;
  constexpr size_t lds = 2 * (size_t)G::BUF_BYTES;
  static_assert(lds <= 160 * 1024, "double buffer must fit the 160 KB LDS");
  auto k = conv_bf16_kernel<CIN, POOL, OUT_F32>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  int grid = p.num_cus > 0 ? p.num_cus : 256;
  grid &= ~7;
  if (grid < 8) grid = 8;
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, s, p);
  return hipGetLastError();
}

size_t conv_bf16_slab_bytes() { return GeoB<8>::W_BYTES; }

hipError_t launch_conv_bf16(const ConvParams &p, int cin, bool pool, bool out_f32, hipStream_t s) {
  if (cin == 64 && pool && !out_f32) return launch_b<64, true, false>(p, s);
  if (cin == 64 && !pool && !out_f32) return launch_b<64, false, false>(p, s);
  if (cin == 128 && pool && !out_f32) return launch_b<128, true, false>(p, s);
  if (cin == 128 && !pool && !out_f32) return launch_b<128, false, false>(p, s);
  if (cin == 128 && !pool && out_f32) return launch_b<128, false, true>(p, s);
  return hipErrorInvalidValue;
}

// conv1a for the bf16 path: same arithmetic as conv1a_kernel (f32 VALU, K = 9), output
// rounded to bf16: 16 lanes per pixel, 4 channels (8 bytes) each.
__global__ __launch_bounds__(256) void conv1a_bf16_kernel(const uint8_t *__restrict__ img,
                                                          const float *__restrict__ w9x64,
                                                          const float *__restrict__ b64,
                                                          unsigned short *__restrict__ out, int B, int H,
                                                          int W, int tiles_x, int tiles_y) {
  constexpr int TH = 8, TW = 32;
  __shared__ float sI[(TH + 2) * (TW + 2)];
  const int tid = threadIdx.x;
  int wg = blockIdx.x;
  const int tx = wg % tiles_x;
  wg /= tiles_x;
  const int ty = wg % tiles_y;
  const int b = wg / tiles_y;
  const int tx0 = tx * TW, ty0 = ty * TH;
  const uint8_t *ib = img + (size_t)b * H * W;
  for (int i = tid; i < (TH + 2) * (TW + 2); i += 256) {
    const int row = i / (TW + 2), col = i % (TW + 2);
    const int gy = ty0 + row - 1, gx = tx0 + col - 1;
    float v = 0.0f;
    if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
      v = (float)ib[(size_t)gy * W + gx] * (1.0f / 255.0f);
    sI[i] = v;
  }
  const int c4 = tid & 15;
  float4 w[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) w[t] = *reinterpret_cast<const float4 *>(w9x64 + t * 64 + c4 * 4);
  const float4 bias = *reinterpret_cast<const float4 *>(b64 + c4 * 4);
  __syncthreads();
  const int psub = tid >> 4;
#pragma unroll 4
  for (int it = 0; it < 16; ++it) {
    const int pix = it * 16 + psub;
    const int row = pix >> 5, col = pix & 31;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float x = sI[(row + t / 3) * (TW + 2) + col + t % 3];
      a.x = fmaf(x, w[t].x, a.x);
      a.y = fmaf(x, w[t].y, a.y);
      a.z = fmaf(x, w[t].z, a.z);
      a.w = fmaf(x, w[t].w, a.w);
    }
    a.x += bias.x; a.y += bias.y; a.z += bias.z; a.w += bias.w;
    a.x = a.x > 0.f ? a.x : 0.f;
    a.y = a.y > 0.f ? a.y : 0.f;
    a.z = a.z > 0.f ? a.z : 0.f;
    a.w = a.w > 0.f ? a.w : 0.f;
    const int gy = ty0 + row, gx = tx0 + col;
    if (gy < H && gx < W) {
      uint2 o;
      o.x = (unsigned)f32_to_bf16_rne(a.x) | ((unsigned)f32_to_bf16_rne(a.y) << 16);
      o.y = (unsigned)f32_to_bf16_rne(a.z) | ((unsigned)f32_to_bf16_rne(a.w) << 16);
      *reinterpret_cast<uint2 *>(out + (((size_t)b * H + gy) * W + gx) * 64 + c4 * 4) = o;
    }
  }
}

hipError_t launch_conv1a_bf16(const uint8_t *img, const float *w9x64, const float *b64, void *out, int B, int H,
                              int W, hipStream_t s) {
  const int tiles_x = (W + 31) / 32, tiles_y = (H + 7) / 8;
  hipLaunchKernelGGL(conv1a_bf16_kernel, dim3(tiles_x * tiles_y * B), dim3(256), 0, s, img, w9x64, b64,
                     reinterpret_cast<unsigned short *>(out), B, H, W, tiles_x, tiles_y);
  return hipGetLastError();
}

}  // namespace spfe
