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
//  * the MFMA is issued as (weights, pixels): D[cout][pixel], so a lane owns ONE pixel and its 16
//    accumulator registers are output channels 4 at a time — the epilogue is packed f32 ops
//    (v_pk_add_f32 bias, v_pk_max_f32 / v_max3_f32 ReLU + pool, the pool's horizontal neighbour one
//    DPP lane away), v_cvt_pk_bf16_f32, and 8-byte (bf16) / 16-byte (f32) stores of 4 channels:
//    ~100 instructions per tile instead of ~700 scalar ones, cut into sub-items of 3-8
//    instructions that fit the 32-cycle MFMA shadows;
//  * staging is `buffer_load_dwordx4 ... lds` (gfx950): the next stage goes from HBM/L2 straight
//    into the other LDS buffer, lane l of a wave writing LDS[M0 + 16 l] — no staging registers,
//    no ds_write, and the loads have a whole stage (72 MFMAs x 32 cycles) to land instead of the
//    half stage a register round trip leaves at this MFMA rate.  Out-of-range buffer offsets
//    write zeros (tools/microbench/lds_direct_probe.hip), which is the conv's zero padding;
//  * pixel / weight rows are padded from 64 to 80 bytes: 16 lanes reading
//    consecutive pixels at the same channel offset then hit 16 different 16-byte
//    bank slots (5 is coprime to 16) — conflict-free ds_read_b128 without an XOR
//    swizzle, and every fragment address is base + immediate.
#include <utility>

#include "conv1a_mfma.h"
#include "spfe_kernels.h"

namespace spfe {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define SPFE_OOB 0x80000000u
#ifndef SPFE_A_AUX
#define SPFE_A_AUX 0  // cache policy of the halo-tile passes (2 = nt)
#endif

__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f) {
  unsigned u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);  // finite inputs only (activations after ReLU / conv sums)
  return (unsigned short)(u >> 16);
}

constexpr int BKC = 32;       // channels per K chunk
constexpr int BPITCH = 64;    // bytes per pixel / weight row in LDS and in the packed slabs: 32 channels, no padding;
                              // piece g (16 B) of column / row c sits in slot g ^ ((c >> 2) & 3)

template <int TH>
struct GeoB {
  static constexpr int ROWS = TH + 2, COLS = 34;
  static constexpr int A_PIECES = ROWS * COLS * (BPITCH / 16);  // 16-byte pieces
  static constexpr int A_BYTES = (A_PIECES + 255) / 256 * 256 * 16;  // whole 256-lane passes
  static constexpr int W_BYTES = 9 * 64 * BPITCH;  // 36,864 = 9 whole 256-thread passes
  static constexpr int BUF_BYTES = A_BYTES + W_BYTES;
};

typedef __attribute__((address_space(3))) void lds_void;

template <int NITER, int NWITER>
struct PipeB {
  unsigned voff[NITER];    // byte offset of each input piece inside the input frame, or SPFE_OOB
  unsigned woff;           // tid * 16: byte offset inside the weight slab (== LDS offset) of pass 0
  __amdgpu_buffer_rsrc_t rin, rw;
  const char *aBase, *bBase;  // this stage's operands (LDS): buffer bases; + the per-lane offsets below + immediates
  unsigned aofs[3][2], bofs[2];   // [dx][16-channel group]: this lane's pixel column / weight row and its swizzled piece
  char *nA, *nW;              // the other LDS buffer, at this wave's 1 KiB slot of pass 0

  // one direct-to-LDS pass: 256 lanes x 16 bytes; pass IT lands at +4096 * IT
  template <int IT>
  __device__ __forceinline__ void dma() const {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (IT < NITER) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_void *)(nA + IT * 4096), 16, voff[IT], 0, 0, SPFE_A_AUX);
    } else if constexpr (IT < NITER + NWITER) {
      constexpr int W = IT - NITER;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void *)(nW + W * 4096), 16, woff, W * 4096, 0, 0);
    }
#endif
  }
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Accumulator layout after mfma(pixels, weights): lane = (output channel, hi = lane >> 5), register r <-> pixel column
// 8*(r>>2) + 4*hi + (r&3) of the wave's row i.  The packed weights put the block's EVEN channels in accumulator tile
// j = 0 and the ODD ones in tile j = 1 (row m of tile j <-> channel 2 m + j), so a lane holds the adjacent channels
// 2 l31, 2 l31 + 1 of every pixel it has: one v_cvt_pk_bf16_f32 packs them and one dword store per register writes, for
// the 32 lanes of each half, the 64 consecutive bf16 channels (128 B) of ONE pixel — as in conv_bf16_ws.hip.  (The
// transposed form this kernel had — a lane owns a pixel, 8-byte pieces scattered over 32-64 cache lines per store —
// cost convPa|Da 0.024 of its 0.153 ms, and its per-register bias took 64 VGPRs that the 16-row tiles need.)
// Columns past the image (ragged last tile, images narrower than 32) are predicated per store: wlim = W - x0 - 4 hi.
template <int MT>
struct EpiB {
  __amdgpu_buffer_rsrc_t rout;
  unsigned rowoff[MT];  // byte offset of (row i [pool: pooled row i], column x0 + 4 hi [pool: its half], this lane's
                        // channel pair) in the output frame, or OOB
  unsigned pitch;       // bytes per output pixel
  int wlim;             // valid pixel columns of this lane's half: register column cc = 8 (r >> 2) + (r & 3) < wlim
  float bias[2];        // this lane's even / odd channel
};
struct EpiHold {};      // (kept for the signatures)

__device__ __forceinline__ float bmax_nc(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float bmax3_nc(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float brelu_nc(float a) {
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(a));
  return r;
}
__device__ __forceinline__ unsigned bpack2(float v0, float v1) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){v0, v1}, bf16x2));
}

// Sub-item E of the epilogue.  pool: (MT / 2) * 8 items — pooled row ip = E / 8, g = (E / 2) % 4, h2 = E % 2: one pooled
// pixel, both channels.  no pool: MT * 16 items — row i = E / 16, g = (E / 4) % 4, m = E % 4: one pixel.
// Bias last (max(a + b, c + b) == max(a, c) + b exactly): the bits of conv_bf16_ws.hip's epilogue.
template <int MT, int NT, bool POOL, bool OUT_F32, int E>
__device__ __forceinline__ void epi_item(const EpiB<MT> &e, EpiHold &, const f32x16 (&acc)[MT][NT]) {
  static_assert(NT == 2 && !OUT_F32, "bf16 outputs, channel pairs across the two accumulator tiles");
  constexpr int NEPI_ = POOL ? (MT / 2) * 8 : MT * 16;
  if constexpr (E < NEPI_) {
    if constexpr (POOL) {
      constexpr int ip = E / 8, g = (E / 2) % 4, h2 = E % 2, r = 4 * g + 2 * h2, i0 = 2 * ip;
      float v[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        v[j] = bmax3_nc(acc[i0][j][r], acc[i0][j][r + 1], bmax_nc(acc[i0 + 1][j][r], acc[i0 + 1][j][r + 1]));
        v[j] = brelu_nc(v[j] + e.bias[j]);
      }
      constexpr int cc = 8 * g + 2 * h2;   // first of the two input columns of this pooled pixel (widths are even)
      const unsigned off = cc < e.wlim ? e.rowoff[ip] : SPFE_OOB;
      __builtin_amdgcn_raw_buffer_store_b32(bpack2(v[0], v[1]), e.rout, off, (unsigned)(4 * g + h2) * e.pitch, 0);
    } else {
      constexpr int i = E / 16, g = (E / 4) % 4, m = E % 4, cc = 8 * g + m;
      const float v0 = brelu_nc(acc[i][0][4 * g + m] + e.bias[0]), v1 = brelu_nc(acc[i][1][4 * g + m] + e.bias[1]);
      const unsigned off = cc < e.wlim ? e.rowoff[i] : SPFE_OOB;
      __builtin_amdgcn_raw_buffer_store_b32(bpack2(v0, v1), e.rout, off, (unsigned)cc * e.pitch, 0);
    }
  }
}

// Everything the side work in the MFMA shadows needs to prepare the next stage / tile / epilogue.
template <int NITER, int NT>
struct CtlB {
  // per-lane geometry of the staging pieces (fixed for the kernel)
  int prc[NITER];   // halo (row << 8 | column) of this lane's piece in pass it, or a row far outside
  int pslot;        // ... and its 16-byte slot in the pixel: tid & 3
  // work items: current, next, and the per-step increment (workgroups stride through an XCD-local range)
  int i_nb, i_tx, i_ty, i_b, n_nb, n_tx, n_ty, n_b, d_nb, d_tx, d_ty, d_b;
  int w, gper, hi_w;
  bool have_next;
  // dynamic order (streamed-weight layers, ConvParams::tile_ctr set): items after a workgroup's first come from a per-XCD
  // atomic counter, so a workgroup that shares its CU with the side-stream kernels of the previous batch (SPFE_FLAG_ASYNC_COV)
  // takes fewer items instead of making the whole launch wait for its static share (conv3b at 1280x720: 0.128 ms alone,
  // 0.200 ms beside the covariance kernels).  lo = first item of this XCD; w_n = the next item; slot = the LDS word
  // through which wave 0 hands the fetched counter value to the other waves.
  bool dyn;
  int lo, w_n;
  float rcp_nblk, rcp_tx, rcp_ty;
  const __attribute__((address_space(3))) int *slot;
  int chunk;  // K chunk the current stage computes
  // constants
  int H, W, Ho, Wo, wm, l31, hi;
  unsigned in_pix_bytes, frame_in_bytes, out_pix_bytes, frame_out_bytes;
};

template <int CIN, int TH, int NITER, int NWITER>
__device__ __forceinline__ void aim_stage_b(const ConvParams &p, PipeB<NITER, NWITER> &c, unsigned frame_in_bytes,
                                            int nb, int b, int chunk, bool valid) {
  using G = GeoB<TH>;
  const char *base = reinterpret_cast<const char *>(p.in) +
                     ((size_t)b * p.H * p.W * p.in_stride + p.in_choff + chunk * BKC) * 2;
  c.rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, valid ? frame_in_bytes : 0u, 0x00020000);
  const char *wb = reinterpret_cast<const char *>(p.wpack) + ((size_t)nb * (CIN / BKC) + chunk) * G::W_BYTES;
  c.rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(wb), 0, valid ? (unsigned)G::W_BYTES : 0u, 0x00020000);
}

template <int TH, int NITER, int NWITER, int NT, int IT>
__device__ __forceinline__ void aim_piece_b(PipeB<NITER, NWITER> &c, const CtlB<NITER, NT> &t, int tx, int ty) {
  if constexpr (IT < NITER) {
    const int col = t.prc[IT] & 0xff;
    const int gy = ty * TH + (t.prc[IT] >> 8) - 1, gx = tx * 32 + col - 1;
    c.voff[IT] = ((unsigned)gy < (unsigned)t.H && (unsigned)gx < (unsigned)t.W)
                     ? (unsigned)(gy * t.W + gx) * t.in_pix_bytes + (unsigned)((t.pslot ^ ((col >> 2) & 3)) * 16)
                     : SPFE_OOB;
  }
}

// q / d for q < 2^20 with rcp = 1.0f / d: (q + 0.5) / d is at least 0.5 / d away from an integer, i.e. 2^-21 relative,
// four times the rounding error of the two float operations
__device__ __forceinline__ int udiv_small(int q, float rcp) { return (int)(((float)q + 0.5f) * rcp); }

template <int NITER, int NT>
__device__ __forceinline__ void next_item_b(const ConvParams &p, CtlB<NITER, NT> &t) {
  if (t.dyn) {
    const int wn = t.lo + t.gper + __builtin_amdgcn_readfirstlane(*t.slot);
    t.w_n = wn;
    t.have_next = wn < t.hi_w;
    int q = wn;
    int d = __builtin_amdgcn_readfirstlane(udiv_small(q, t.rcp_nblk));
    t.n_nb = q - d * p.nblk; q = d;
    d = __builtin_amdgcn_readfirstlane(udiv_small(q, t.rcp_tx));
    t.n_tx = q - d * p.tiles_x; q = d;
    d = __builtin_amdgcn_readfirstlane(udiv_small(q, t.rcp_ty));
    t.n_ty = q - d * p.tiles_y; t.n_b = d;
    return;
  }
  int n_nb = t.i_nb + t.d_nb, n_tx = t.i_tx + t.d_tx, n_ty = t.i_ty + t.d_ty, n_b = t.i_b + t.d_b;
  if (n_nb >= p.nblk) { n_nb -= p.nblk; ++n_tx; }
  if (n_tx >= p.tiles_x) { n_tx -= p.tiles_x; ++n_ty; }
  if (n_ty >= p.tiles_y) { n_ty -= p.tiles_y; ++n_b; }
  t.n_nb = n_nb; t.n_tx = n_tx; t.n_ty = n_ty; t.n_b = n_b;
  t.w_n = t.w + t.gper;
  t.have_next = t.w_n < t.hi_w;
}

// epilogue context of the CURRENT tile (used one tile later)
template <int TH, int MT, int NT, bool POOL, bool OUT_F32, int NITER>
__device__ __forceinline__ void aim_epi_geom_b(const ConvParams &p, const CtlB<NITER, NT> &t, EpiB<MT> &e) {
  char *obase = reinterpret_cast<char *>(p.out) + ((size_t)t.i_b * t.Ho * t.Wo * p.out_stride + p.out_choff) * 2;
  e.rout = __builtin_amdgcn_make_buffer_rsrc(obase, 0, t.frame_out_bytes, 0x00020000);
  e.pitch = t.out_pix_bytes;
  const int y0 = t.i_ty * TH + t.wm * MT, x0 = t.i_tx * 32;
  e.wlim = t.W - x0 - 4 * t.hi;
  const unsigned chan = (unsigned)(t.i_nb * 64 + 2 * t.l31) * 2u;
  if constexpr (POOL) {
#pragma unroll
    for (int ip = 0; ip < MT / 2; ++ip)
      e.rowoff[ip] = y0 + 2 * ip < t.H ? (unsigned)(((y0 >> 1) + ip) * t.Wo + (x0 >> 1) + 2 * t.hi) * t.out_pix_bytes + chan : SPFE_OOB;
  } else {
#pragma unroll
    for (int i = 0; i < MT; ++i)
      e.rowoff[i] = y0 + i < t.H ? (unsigned)((y0 + i) * t.W + x0 + 4 * t.hi) * t.out_pix_bytes + chan : SPFE_OOB;
  }
}
template <int MT, int NT, int NITER>
__device__ __forceinline__ void aim_epi_bias_b(const ConvParams &p, const CtlB<NITER, NT> &t, EpiB<MT> &e) {
  const float2 b2 = *reinterpret_cast<const float2 *>(p.bias + t.i_nb * 64 + 2 * t.l31);
  e.bias[0] = b2.x;
  e.bias[1] = b2.y;
}

// One stage = NSTEP K steps of MT x NT MFMAs; all side work sits in the MFMA shadows:
//   m0: operand fragments of the next K step (LDS -> registers)
//   m1: steps 0..: the next stage, HBM/L2 -> the other LDS buffer (two passes per step); later steps:
//       what the FOLLOWING stage's passes will need, i.e. the stage two ahead (PREP 0: this tile's
//       chunk + 2; PREP 1, second-last stage: the next work item, its first descriptors and piece
//       offsets; PREP 2, last stage: the next item's chunk 1, and this tile's epilogue context)
//   m2, m3: sub-items of the PREVIOUS tile's epilogue (first stage of a tile only)
template <int STEP, int NSTEP, bool FIRST, int PREP, int CIN, int TH, int MT, int NT, int NITER, int NWITER,
          bool POOL, bool OUT_F32>
__device__ __forceinline__ void k_steps_b(const ConvParams &p, bf16x8 (&a)[2][MT + 2], bf16x8 (&bb)[3][NT],
                                          f32x16 (&acc)[MT][NT], const f32x16 (&accPrev)[MT][NT],
                                          PipeB<NITER, NWITER> &c, CtlB<NITER, NT> &t, EpiB<MT> &eMine,
                                          const EpiB<MT> &ePrev, EpiHold &hold) {
  if constexpr (STEP < NSTEP) {
    constexpr int NEPI = POOL ? (MT / 2) * 8 : MT * 16;        // epilogue sub-items of the previous tile
    constexpr int ES = 8;                                      // ... spread over steps 1..ES,
    constexpr int HALF = (NEPI + 2 * ES - 1) / (2 * ES);       // HALF of them in each of two shadows
    constexpr int NLD = NITER + NWITER;
    constexpr int DPS = NLD > 14 ? 3 : NLD > 10 ? 2 : 1;   // LDS-direct passes per step
    constexpr int S0 = (NLD + DPS - 1) / DPS;    // first step without passes
    static_assert(S0 + 1 + NITER <= NSTEP, "side work does not fit the stage");
    // operand fragments are requested two K steps ahead (a step's MFMAs take 4 x 32 cycles, less than an LDS round trip
    // under load): weights in a ring of three steps, halo rows per K group (below)
    constexpr int cur = STEP % 3, nxt = (STEP + 2) % 3;
    constexpr int M = MT * NT;
#pragma unroll
    for (int m = 0; m < M; ++m) {
      if (m == 0) {
        if constexpr (STEP + 2 < NSTEP) {
          // K order inside the 32-channel chunk: dx -> 16-channel group -> dy (conv_bf16_ws.hip shares fragments across
          // the three vertical taps; the order is the same here so that the two kernels stay bit-identical)
          constexpr int s2 = STEP + 2, dx = s2 / 6, kk = (s2 % 6) / 3, dy = s2 % 3, tap = dy * 3 + dx;
          // halo rows of K group g = step / 3 live in a[g % 2][0 .. MT + 1]: rows 0 .. MT - 1 are read for dy = 0,
          // row MT - 1 + dy for the two taps below — the three vertical taps share them
          constexpr int gp = (s2 / 3) % 2;
          if constexpr (dy == 0) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
              a[gp][i] = *reinterpret_cast<const bf16x8 *>(c.aBase + c.aofs[dx][kk] + (i * 34) * BPITCH);
          } else {
            a[gp][MT - 1 + dy] = *reinterpret_cast<const bf16x8 *>(c.aBase + c.aofs[dx][kk] + ((MT - 1 + dy) * 34) * BPITCH);
          }
#pragma unroll
          for (int j = 0; j < NT; ++j)
            bb[nxt][j] = *reinterpret_cast<const bf16x8 *>(c.bBase + c.bofs[kk] + (tap * 64 + j * 32) * BPITCH);
        }
        if constexpr (PREP == 2 && STEP == 0) aim_epi_bias_b<MT, NT, NITER>(p, t, eMine);  // this tile's bias (early: a vmcnt load too)
      }
      // (16-row tiles: 8 MFMAs per step, the passes one per gap at m = 1, 3, 5)
      if constexpr (M == 8 && STEP < S0) {
        if (m == 3) { if constexpr (DPS >= 2) c.template dma<STEP * DPS + 1>(); }
        if (m == 5) { if constexpr (DPS >= 3) c.template dma<STEP * DPS + 2>(); }
      }
      if (m == 1 % M) {
        if constexpr (STEP < S0) {
          c.template dma<STEP * DPS>();
          if constexpr (M != 8 && DPS >= 2) c.template dma<STEP * DPS + 1>();
          if constexpr (M != 8 && DPS >= 3) c.template dma<STEP * DPS + 2>();
        } else if constexpr (PREP == 0) {
          if constexpr (STEP == S0) aim_stage_b<CIN, TH>(p, c, t.frame_in_bytes, t.i_nb, t.i_b, t.chunk + 2, true);
        } else if constexpr (PREP == 1) {
          if constexpr (STEP == S0) {
            next_item_b(p, t);
            aim_stage_b<CIN, TH>(p, c, t.frame_in_bytes, t.n_nb, t.n_b, 0, t.have_next);
          }
          if constexpr (STEP > S0) aim_piece_b<TH, NITER, NWITER, NT, STEP - S0 - 1>(c, t, t.n_tx, t.n_ty);
        } else {
          if constexpr (STEP == S0) aim_stage_b<CIN, TH>(p, c, t.frame_in_bytes, t.n_nb, t.n_b, 1, t.have_next);
          if constexpr (STEP == S0 + 1) aim_epi_geom_b<TH, MT, NT, POOL, OUT_F32>(p, t, eMine);
        }
      }
      // The previous tile's epilogue goes EARLY in the stage (steps 1..ES): the stage ends with
      // vmcnt(0) for the LDS-direct loads, and stores issued late would be waited for as well.
      if (m == 2 % M) {
        if constexpr (FIRST && STEP >= 1 && STEP <= ES) {
          [&]<int... Q>(std::integer_sequence<int, Q...>) {
            (epi_item<MT, NT, POOL, OUT_F32, (STEP - 1) * 2 * HALF + Q>(ePrev, hold, accPrev), ...);
          }(std::make_integer_sequence<int, HALF>{});
        }
      }
      if (m == (M == 8 ? 6 : 3 % M)) {
        if constexpr (FIRST && STEP >= 1 && STEP <= ES) {
          [&]<int... Q>(std::integer_sequence<int, Q...>) {
            (epi_item<MT, NT, POOL, OUT_F32, (STEP - 1) * 2 * HALF + HALF + Q>(ePrev, hold, accPrev), ...);
          }(std::make_integer_sequence<int, HALF>{});
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      {
        const int i = m / NT, j = m % NT;
        if constexpr (FIRST && STEP == 0) {
          f32x16 z;
#pragma unroll
          for (int r = 0; r < 16; ++r) z[r] = 0.0f;
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(STEP / 3) % 2][i + STEP % 3], bb[cur][j], z, 0, 0, 0);
        } else {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(STEP / 3) % 2][i + STEP % 3], bb[cur][j], acc[i][j], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    k_steps_b<STEP + 1, NSTEP, FIRST, PREP, CIN, TH, MT, NT, NITER, NWITER, POOL, OUT_F32>(p, a, bb, acc, accPrev, c, t,
                                                                                         eMine, ePrev, hold);
  }
}

// in: NHWC bf16 [B][H][W][in_stride]; wpack: bf16 slabs [nblk][chunk][tap][64 n][BPITCH bytes];
// out: NHWC bf16 (or f32 when OUT_F32) — strides in ConvParams are in ELEMENTS.
// MT = output rows per wave: 2 (8-row tiles) or 4 (16-row tiles: a stage's weight chunk feeds twice the MFMAs — the
// streamed-weight layers run at the chip's LDS-DMA fill rate, and 19 passes per 144 MFMAs per wave beat 15 per 72)
template <int CIN, bool POOL, bool OUT_F32, int MT>
__global__ __launch_bounds__(256, 1) void conv_bf16_kernel(ConvParams p) {
  constexpr int WM = 4, NT = 2, TH = WM * MT;
  using G = GeoB<TH>;
  constexpr int NCHUNK = CIN / BKC;
  static_assert(NCHUNK >= 2, "first and last stage of a tile are different stages");
  constexpr int NITEM = G::A_PIECES;  // 16-byte pieces of the halo tile (5 per pixel: 4 data + the pad)
  constexpr int NITER = (NITEM + 255) / 256;
  constexpr int NW16 = G::W_BYTES / 16;
  constexpr int NWPASS = NW16 / 256;
  // Cin = 64: both K chunks of the 64-channel weight block (2 x 48 KB) stay in LDS for the whole
  // kernel and only the halo tiles stream (7 passes per stage instead of 19): at this MFMA rate
  // re-fetching the weights for every tile is what the L2 cannot feed.  Cin = 128 (4 chunks =
  // 192 KB) streams weights with the tile, double buffered.
  constexpr bool RESW = NCHUNK == 2;
  constexpr int NWITER = RESW ? 0 : NWPASS;
  constexpr int NSTEP = 9 * (BKC / 16);
  static_assert(NW16 % 256 == 0, "padded weight slab is a whole number of passes");

  extern __shared__ __attribute__((aligned(16))) char smem_b[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int total = p.nblk * p.tiles_x * p.tiles_y * p.B;
  const int xcd = blockIdx.x & 7, gi = blockIdx.x >> 3;
  const int lo = (int)((long)total * xcd / 8);

  CtlB<NITER, NT> t;
  t.gper = gridDim.x >> 3;
  t.hi_w = (int)((long)total * (xcd + 1) / 8);
  t.w = lo + gi;
  if (t.w >= t.hi_w) return;
  t.H = p.H; t.W = p.W;
  t.Ho = POOL ? p.H >> 1 : p.H;
  t.Wo = POOL ? p.W >> 1 : p.W;
  t.wm = wave; t.l31 = lane & 31; t.hi = lane >> 5;
  constexpr unsigned OEL = OUT_F32 ? 4u : 2u;
  t.in_pix_bytes = (unsigned)p.in_stride * 2u;
  t.frame_in_bytes = (unsigned)p.H * p.W * t.in_pix_bytes;
  t.out_pix_bytes = (unsigned)p.out_stride * OEL;
  t.frame_out_bytes = (unsigned)t.Ho * t.Wo * t.out_pix_bytes;
  {
    int q = t.w;
    t.i_nb = q % p.nblk; q /= p.nblk;
    t.i_tx = q % p.tiles_x; q /= p.tiles_x;
    t.i_ty = q % p.tiles_y; t.i_b = q / p.tiles_y;
    q = t.gper;
    t.d_nb = q % p.nblk; q /= p.nblk;
    t.d_tx = q % p.tiles_x; q /= p.tiles_x;
    t.d_ty = q % p.tiles_y; t.d_b = q / p.tiles_y;
  }
  t.n_nb = t.n_tx = t.n_ty = t.n_b = 0;
  t.have_next = false;
  t.chunk = 0;
  // dynamic order only where weights stream with the tile (a resident 64-channel block would have to be re-loaded
  // whenever the queue hands out another block) and the index arithmetic of udiv_small is exact
  t.dyn = !RESW && p.tile_ctr != nullptr && total < (1 << 20);
  t.lo = lo; t.w_n = 0;
  t.rcp_nblk = 1.0f / (float)p.nblk; t.rcp_tx = 1.0f / (float)p.tiles_x; t.rcp_ty = 1.0f / (float)p.tiles_y;
  t.slot = (const __attribute__((address_space(3))) int *)(smem_b + 2 * G::BUF_BYTES);
  int *const qctr = p.tile_ctr ? p.tile_ctr + xcd : nullptr;
  int fetched = 0;   // (lane 0 of wave 0: the counter value its atomic returned)
#pragma unroll
  for (int it = 0; it < NITER; ++it) {
    const int i = tid + it * 256;
    const int slot = i % 4, pix = i / 4, col = pix % G::COLS;
    // pieces past the tile read out of range (-> zeros); the source piece is the slot's un-swizzled index
    t.prc[it] = ((i < NITEM ? pix / G::COLS : (1 << 16)) << 8) | col;
    (void)slot;
  }
  t.pslot = tid & 3;   // (256 lanes per pass: a lane keeps its slot)

  PipeB<NITER, NWITER> c;
  c.woff = (unsigned)tid * 16u;
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int col = t.l31 + dx;
      c.aofs[dx][kk] = (unsigned)(((t.wm * MT) * 34 + col) * BPITCH + (((2 * kk + t.hi) ^ ((col >> 2) & 3)) * 16));
    }
    c.bofs[kk] = (unsigned)(t.l31 * BPITCH + (((2 * kk + t.hi) ^ ((t.l31 >> 2) & 3)) * 16));
  }
  const unsigned wave_slot = (unsigned)wave * 1024u;

  // LDS map.  streaming weights: [A0 | W0 | A1 | W1]; resident weights: [W chunk 0 | W chunk 1 | A0 | A1]
  auto lds_a = [&](int b) -> char * { return smem_b + (RESW ? 2 * G::W_BYTES + b * G::A_BYTES : b * G::BUF_BYTES); };
  auto lds_w = [&](int b_or_chunk) -> char * {
    return smem_b + (RESW ? b_or_chunk * G::W_BYTES : b_or_chunk * G::BUF_BYTES + G::A_BYTES);
  };
  auto load_resident_weights = [&](int nb) {  // both chunks of block nb, all passes, this wave's slots
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int ch = 0; ch < NCHUNK; ++ch) {
      const char *wb = reinterpret_cast<const char *>(p.wpack) + ((size_t)nb * NCHUNK + ch) * G::W_BYTES;
      const __amdgpu_buffer_rsrc_t rw =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(wb), 0, (unsigned)G::W_BYTES, 0x00020000);
      [&]<int... PS>(std::integer_sequence<int, PS...>) {
        (__builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void *)(lds_w(ch) + wave_slot + PS * 4096), 16, c.woff,
                                                  PS * 4096, 0, 0),
         ...);
      }(std::make_integer_sequence<int, NWPASS>{});
    }
#endif
  };

  // prologue: first stage straight into buffer 0 (and the resident weights)
  [&]<int... IT>(std::integer_sequence<int, IT...>) {
    (aim_piece_b<TH, NITER, NWITER, NT, IT>(c, t, t.i_tx, t.i_ty), ...);
  }(std::make_integer_sequence<int, NITER>{});
  aim_stage_b<CIN, TH>(p, c, t.frame_in_bytes, t.i_nb, t.i_b, 0, true);
  c.nA = lds_a(0) + wave_slot;
  c.nW = lds_w(0) + wave_slot;
  if constexpr (RESW) load_resident_weights(t.i_nb);
  [&]<int... IT>(std::integer_sequence<int, IT...>) {
    (c.template dma<IT>(), ...);
  }(std::make_integer_sequence<int, NITER + NWITER>{});
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the direct-to-LDS loads have landed
  __syncthreads();
  aim_stage_b<CIN, TH>(p, c, t.frame_in_bytes, t.i_nb, t.i_b, 1, true);  // what the first stage's passes load

  f32x16 accA[MT][NT], accB[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { accA[i][j][r] = 0.0f; accB[i][j][r] = 0.0f; }

  EpiB<MT> epiA, epiB;
  EpiHold hold;
  epiA.rout = epiB.rout = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, 0u, 0x00020000);  // nothing to store yet
#pragma unroll
  for (int i = 0; i < MT; ++i) epiA.rowoff[i] = epiB.rowoff[i] = SPFE_OOB;
  epiA.pitch = epiB.pitch = 0u;
  epiA.wlim = epiB.wlim = 0;
  epiA.bias[0] = epiA.bias[1] = epiB.bias[0] = epiB.bias[1] = 0.0f;

  int buf = 0;
  bf16x8 a[2][MT + 2], bb[3][NT];
  auto begin_stage = [&]() {  // operand bases of the stage in `buf`, DMA targets in the other buffer, first fragments
    c.nA = lds_a(buf ^ 1) + wave_slot;
    c.nW = lds_w(buf ^ 1) + wave_slot;  // (unused with resident weights)
    c.aBase = lds_a(buf);
    c.bBase = lds_w(RESW ? t.chunk : buf);
    // K steps 0 and 1: taps (dy = 0, dx = 0) and (1, 0), channels 0-15: halo rows 0 .. MT of group 0
#pragma unroll
    for (int i = 0; i <= MT; ++i) a[0][i] = *reinterpret_cast<const bf16x8 *>(c.aBase + c.aofs[0][0] + (i * 34) * BPITCH);
#pragma unroll
    for (int st = 0; st < 2; ++st) {
#pragma unroll
      for (int j = 0; j < NT; ++j) bb[st][j] = *reinterpret_cast<const bf16x8 *>(c.bBase + c.bofs[0] + (st * 3 * 64 + j * 32) * BPITCH);
    }
  };
  auto end_stage = [&]() {
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the next stage is in LDS (and this tile's stores are out)
    __syncthreads();
    buf ^= 1;
  };
  // first stage of a tile, dynamic order: wave 0 asked the queue at the start of the stage; the value is back with the
  // stage's loads and goes to the other waves through LDS under the same barrier
  auto end_stage_publish = [&]() {
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if (t.dyn && tid == 0) *(__attribute__((address_space(3))) int *)(smem_b + 2 * G::BUF_BYTES) = fetched;
    __syncthreads();
    buf ^= 1;
  };

  auto run_tile = [&](f32x16(&acc)[MT][NT], const f32x16(&accPrev)[MT][NT], EpiB<MT> &eMine, const EpiB<MT> &ePrev) {
    t.chunk = 0;
    if constexpr (!RESW) {
      // (compiled with the atomic optimiser off, see the Makefile: the plain instruction, waited for where it is used)
      if (t.dyn && tid == 0) fetched = atomicAdd(qctr, 1);
    }
    begin_stage();
    k_steps_b<0, NSTEP, true, NCHUNK == 2 ? 1 : 0, CIN, TH, MT, NT, NITER, NWITER, POOL, OUT_F32>(
        p, a, bb, acc, accPrev, c, t, eMine, ePrev, hold);
    if constexpr (!RESW) end_stage_publish(); else end_stage();
    if constexpr (NCHUNK > 2) {
#pragma unroll 1
      for (int ch = 1; ch < NCHUNK - 2; ++ch) {
        t.chunk = ch;
        begin_stage();
        k_steps_b<0, NSTEP, false, 0, CIN, TH, MT, NT, NITER, NWITER, POOL, OUT_F32>(p, a, bb, acc, accPrev, c, t,
                                                                                    eMine, ePrev, hold);
        end_stage();
      }
      t.chunk = NCHUNK - 2;
      begin_stage();
      k_steps_b<0, NSTEP, false, 1, CIN, TH, MT, NT, NITER, NWITER, POOL, OUT_F32>(p, a, bb, acc, accPrev, c, t, eMine,
                                                                                  ePrev, hold);
      end_stage();
    }
    t.chunk = NCHUNK - 1;
    begin_stage();
    k_steps_b<0, NSTEP, false, 2, CIN, TH, MT, NT, NITER, NWITER, POOL, OUT_F32>(p, a, bb, acc, accPrev, c, t, eMine,
                                                                                ePrev, hold);
    end_stage();
    if constexpr (RESW) {
      // another 64-channel block next (only when the workgroup stride is not a multiple of nblk): every
      // wave is past its last read of the resident weights (end_stage barrier), so replace them now
      if (t.have_next && t.n_nb != t.i_nb) {
        load_resident_weights(t.n_nb);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
      }
    }
    t.w = t.w_n;
    t.i_nb = t.n_nb; t.i_tx = t.n_tx; t.i_ty = t.n_ty; t.i_b = t.n_b;
  };

  bool lastA = true;
  while (true) {
    run_tile(accA, accB, epiA, epiB);
    lastA = true;
    if (!t.have_next) break;
    run_tile(accB, accA, epiB, epiA);
    lastA = false;
    if (!t.have_next) break;
  }
  {
    constexpr int NEPI = POOL ? (MT / 2) * 8 : MT * 16;
    auto flush = [&](const f32x16(&acc)[MT][NT], const EpiB<MT> &e) {
      [&]<int... E>(std::integer_sequence<int, E...>) {
        (epi_item<MT, NT, POOL, OUT_F32, E>(e, hold, acc), ...);
      }(std::make_integer_sequence<int, NEPI>{});
    };
    if (lastA) flush(accA, epiA); else flush(accB, epiB);
  }
}

template <int CIN, bool POOL, bool OUT_F32, int MT = 2>
static hipError_t launch_b(const ConvParams &p, hipStream_t s) {
  using G = GeoB<4 * MT>;
  constexpr size_t lds = 2 * (size_t)G::BUF_BYTES + 16;   // + the queue slot
  static_assert(lds <= 160 * 1024, "double buffer must fit the 160 KB LDS");
  auto k = conv_bf16_kernel<CIN, POOL, OUT_F32, MT>;
  static bool attr_done[64] = {};  // per instantiation and device: one process may hold handles on several GPUs
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) attr_done[dev] = true;
  }
  int grid = p.num_cus > 0 ? p.num_cus : 256;
  grid &= ~7;
  if (grid < 8) grid = 8;
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, s, p);
  return hipGetLastError();
}

size_t conv_bf16_slab_bytes() { return GeoB<8>::W_BYTES; }

// tile_rows: 8, or 12 (cin = 128, no pool) / 16 (cin = 128): p.tiles_y must count tiles of that height
hipError_t launch_conv_bf16(const ConvParams &p, int cin, bool pool, bool out_f32, hipStream_t s, int tile_rows) {
  if (tile_rows == 16) {
    if (cin == 128 && pool && !out_f32) return launch_b<128, true, false, 4>(p, s);
    if (cin == 128 && !pool && !out_f32) return launch_b<128, false, false, 4>(p, s);
    return hipErrorInvalidValue;
  }
  if (tile_rows == 12) {
    if (cin == 128 && !pool && !out_f32) return launch_b<128, false, false, 3>(p, s);
    return hipErrorInvalidValue;
  }
  if (cin == 64 && pool && !out_f32) return launch_b<64, true, false>(p, s);
  if (cin == 64 && !pool && !out_f32) return launch_b<64, false, false>(p, s);
  if (cin == 128 && pool && !out_f32) return launch_b<128, true, false>(p, s);
  if (cin == 128 && !pool && !out_f32) return launch_b<128, false, false>(p, s);
  return hipErrorInvalidValue;
}

// conv1a for the bf16 path, stand-alone (launches the wave-specialised conv1b does not take — it computes conv1a
// itself otherwise): the arithmetic of conv1a_mfma.h, 2 MFMAs per 32 pixels of an image row.  One workgroup =
// an 8 x 32 pixel tile, one wavefront = two of its rows; the tile's u8 pixels (+ 1 border) sit in LDS as bf16.
__global__ __launch_bounds__(256) void conv1a_bf16_kernel(const uint8_t *__restrict__ img, const void *__restrict__ wtab,
                                                          const float *__restrict__ b64,
                                                          unsigned short *__restrict__ out, int B, int H,
                                                          int W, int tiles_x, int tiles_y) {
  constexpr int TH = 8, TW = 32;
  __shared__ unsigned short sP[(TH + 2) * c1a::PATCH_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
    unsigned v = 0;
    if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) v = ib[(size_t)gy * W + gx];
    sP[row * c1a::PATCH_PITCH + col] = c1a::u8_to_bf16(v);
  }
  c1a::bf16x8 wA[2];
  float bias[2][16];
  c1a::load_constants(wtab, b64, lane, wA, bias);
  __syncthreads();
  const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const int row = wave * 2 + rr;
    const c1a::bf16x8 px = c1a::pixel_operand(
        (c1a::lds_u16 *)sP + row * c1a::PATCH_PITCH + l31, hi);
    c1a::f32x16 acc[2];
    c1a::product(wA, px, bias, acc);
    const int gy = ty0 + row, gx = tx0 + l31;
    if (gy < H && gx < W) {
      unsigned short *o = out + (((size_t)b * H + gy) * W + gx) * 64 + 8 * hi;   // (the lane's pieces: 4 j + 2 rr + hi, conv1a_mfma.h)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          const c1a::u32x4 v = c1a::finish8(acc[j], rr);
          *reinterpret_cast<uint4 *>(o + 32 * j + 16 * rr) = make_uint4(v.x, v.y, v.z, v.w);
        }
    }
  }
}

// wtab: the bf16 operand table of conv1a_mfma.h ([2][64 lanes][8 bf16]); b64: bias
hipError_t launch_conv1a_bf16(const uint8_t *img, const void *wtab, const float *b64, void *out, int B, int H,
                              int W, hipStream_t s) {
  const int tiles_x = (W + 31) / 32, tiles_y = (H + 7) / 8;
  hipLaunchKernelGGL(conv1a_bf16_kernel, dim3(tiles_x * tiles_y * B), dim3(256), 0, s, img, wtab, b64,
                     reinterpret_cast<unsigned short *>(out), B, H, W, tiles_x, tiles_y);
  return hipGetLastError();
}

}  // namespace spfe
