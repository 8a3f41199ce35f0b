// head_bf16.hip — the two 1x1 head convolutions of the bf16 mode (no ReLU; /root/reference/orb_slam2/src/cv/
// sp_extractor.cpp:96-100): convDb (256 -> 256) and convPb (256 -> 65), plain GEMMs
//   out[P][COUT] (f32) = in[P][256 of the 512 head channels] (bf16) x W^T (bf16) + bias,  P = frames x cells,
// on v_mfma_f32_32x32x16_bf16.
//
// These are HBM-bound (convDb at 1280x720 x 8: 59 MB in, 118 MB of f32 out = 0.03 ms at 6 TB/s, against 0.01 ms of
// matrix time), so the design removes everything that is not the pixel stream:
//   * WEIGHTS LIVE IN REGISTERS: a wave owns 64 (convDb) or 32 (convPb) output channels for the whole kernel — 16 K steps
//     x 2 tiles x 4 VGPRs = 128 registers, loaded once from a table packed in fragment order.  (The first version
//     streamed all 128 KB of weights through LDS for every 64 pixels: 230 MB of L2 -> LDS traffic per launch, 0.068 ms.)
//   * persistent workgroups (two per CU, so one computes while the other waits for memory) walk 32-pixel tiles; a
//     tile's 16 KB come HBM -> LDS with LDS-direct loads into a double buffer, XOR-swizzled by the pixel on the
//     source side (conflict-free 16-byte fragment reads), and are read as the A operand by all four waves;
//   * mfma(pixels, weights): a lane owns output channels, and the even / odd channels of a wave's block sit in its two
//     accumulator tiles, so one 8-byte store per register writes 256 contiguous bytes of a pixel's row (convPb: 4-byte
//     stores, 128-byte runs); a tile's stores go out while the next tile computes.
// K order: the 16 MFMA steps ascend through the input channels, as in the first version: same bits.
#include <algorithm>
#include <cstring>
#include <utility>

#include "spfe_kernels.h"

namespace spfe {

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2v __attribute__((__vector_size__(2 * sizeof(unsigned))));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) char lds_char;
constexpr int HT = 32;                 // pixels per tile
constexpr int HT_BYTES = HT * 512;     // 256 input channels (bf16) per pixel
constexpr int H_KSTEPS = 16;
constexpr int H_WG_PER_CU = 2;
}  // namespace

// in: [npix][IN_STRIDE] bf16, the head reads channels [in_choff, in_choff + 256); out: [npix][COUT] f32
// wpack: [wave 4][tile NTW][K step 16][lane 64][8 bf16] (head_bf16_pack_weights)
// GATHER (the descriptor head of the product path, "sparse convDb"): the pixels are the `*total` cells of `list` (global cell
// indices b * C + cell, written by select_kernel: the cells some emitted keypoint's bilinear taps read); pixel p of the
// walk reads row list[p] of `in` and writes row list[p] of `out` — the dense map's layout, only the rows anybody reads.  A
// pixel's result does not depend on which other pixels share its tile, so those rows hold the dense kernel's bits.
template <int COUT, int IN_STRIDE, bool GATHER>
__global__ __launch_bounds__(256, H_WG_PER_CU) void head1x1_bf16_kernel(const unsigned short *__restrict__ in, int in_choff,
                                                                        const unsigned char *__restrict__ wpack,
                                                                        const float *__restrict__ bias,
                                                                        float *__restrict__ out, int npix,
                                                                        const int *__restrict__ list, const int *__restrict__ total) {
  constexpr int NTW = COUT == 256 ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) char sm_h[];
  lds_char *const lds = (lds_char *)sm_h;
  int *const sIdx = reinterpret_cast<int *>(sm_h + 2 * HT_BYTES);   // GATHER: [4][32] cell indices of the tiles in flight
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int nwalk = GATHER ? __builtin_amdgcn_readfirstlane(*total) : npix;   // pixels this launch walks
  const int ntiles = (nwalk + HT - 1) / HT;

  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned short *>(in) + in_choff, 0, (unsigned)((size_t)npix * IN_STRIDE * 2 - (size_t)in_choff * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rout =
      __builtin_amdgcn_make_buffer_rsrc(out, 0, (unsigned)((size_t)npix * COUT * 4), 0x00020000);
  (void)rin;   // (the host pass of hipcc does not see the uses below)

  // this wave's weights, for the whole kernel
  bf16x8 wreg[NTW][H_KSTEPS];
#pragma unroll
  for (int j = 0; j < NTW; ++j)
#pragma unroll
    for (int kk = 0; kk < H_KSTEPS; ++kk)
      wreg[j][kk] = __builtin_bit_cast(bf16x8, reinterpret_cast<const u32x4 *>(wpack)[((wave * NTW + j) * H_KSTEPS + kk) * 64 + lane]);
  // output channel(s) of this lane: convDb 64 wave + 2 l31 + j; convPb 32 wave + l31
  const int co = COUT == 256 ? wave * 64 + 2 * l31 : wave * 32 + l31;
  float bv[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) bv[j] = co + j < COUT ? bias[co + j] : 0.0f;
  const bool lane_out = co < COUT;

  // a tile's 1024 16-byte pieces = 16 LDS-direct passes, 4 per wave: pass p, lane l -> LDS piece q = 64 p + l = (pixel
  // q >> 5, slot q & 31), which holds the pixel's piece slot ^ (pixel & 31)
  unsigned dsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = (4 * i + wave) * 64 + lane, px = q >> 5, slot = q & 31;
    dsrc[i] = (unsigned)px * (unsigned)(IN_STRIDE * 2) + (unsigned)((slot ^ (px & 31)) * 16);
  }
  // GATHER: pass i of this wave carries the tile's pixels 2 (4 i + wave) + hi; their cell indices are fetched one tile ahead
  int gidx[4] = {-1, -1, -1, -1};
  auto load_idx = [&](int tile) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = tile * HT + 2 * (4 * i + wave) + hi;
      gidx[i] = p < nwalk ? list[p] : -1;
    }
  };
  auto dma = [&](int tile, int buf, int ring) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned base = (unsigned)tile * (unsigned)(HT * IN_STRIDE * 2);   // (past the last pixel: out of range -> zeros)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned src = base + dsrc[i];
      if constexpr (GATHER) {
        const unsigned inrow = (unsigned)((((4 * i + wave) * 64 + lane) & 31) ^ ((2 * (4 * i + wave) + hi) & 31)) * 16u;
        src = gidx[i] < 0 ? 0x80000000u : (unsigned)gidx[i] * (unsigned)(IN_STRIDE * 2) + inrow;
        if (l31 == 0) sIdx[ring * HT + 2 * (4 * i + wave) + hi] = gidx[i];
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_void *)(lds + buf * HT_BYTES + (4 * i + wave) * 1024), 16, src, 0, 0, 0);
    }
#endif
  };
  // A fragment of K step kk: piece 2 kk + hi of pixel l31
  unsigned aoff[H_KSTEPS];
#pragma unroll
  for (int kk = 0; kk < H_KSTEPS; ++kk) aoff[kk] = (unsigned)(l31 * 512 + (((2 * kk + hi) ^ l31) & 31) * 16);

  // D[pixel][channel]: register r = pixel (r & 3) + 8 (r >> 2) + 4 hi of the tile
  auto store_tile = [&](const f32x16 (&acc)[NTW], int tile, int ring) {
    const unsigned base = (unsigned)tile * (unsigned)(HT * COUT * 4) + (unsigned)co * 4u;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      unsigned off = lane_out ? base + (unsigned)(((r & 3) + 8 * (r >> 2) + 4 * hi) * COUT * 4) : 0x80000000u;
      if constexpr (GATHER) {
        const int cell = sIdx[ring * HT + (r & 3) + 8 * (r >> 2) + 4 * hi];
        off = lane_out && cell >= 0 ? (unsigned)cell * (unsigned)(COUT * 4) + (unsigned)co * 4u : 0x80000000u;
      }
      if constexpr (NTW == 2) {
        const f32x2 v = {acc[0][r] + bv[0], acc[1][r] + bv[1]};
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2v, v), rout, off, 0, 0);
      } else {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[0][r] + bv[0]), rout, off, 0, 0);
      }
    }
  };

  f32x16 accA[NTW], accB[NTW];
  int tile = blockIdx.x, prev = -1;
  int it = 0;   // tiles this workgroup has started; tile number `it` keeps its cell indices in ring slot it & 3
  if (tile < ntiles) {
    if constexpr (GATHER) load_idx(tile);
    dma(tile, 0, 0);
    if constexpr (GATHER) load_idx(tile + (int)gridDim.x);
  }
  int buf = 0;
  auto run = [&](f32x16 (&acc)[NTW], const f32x16 (&accPrev)[NTW]) {
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this tile has landed (and the stores of the tile before the previous one are out)
    __syncthreads();                      // ... for every wave; and every wave is done reading the other buffer
    const int nxt = tile + (int)gridDim.x;
    if (nxt < ntiles) {
      dma(nxt, buf ^ 1, (it + 1) & 3);
      if constexpr (GATHER) load_idx(nxt + (int)gridDim.x);
    }
    if (prev >= 0) store_tile(accPrev, prev, (it - 1) & 3);   // the previous tile's outputs leave while this one computes
    lds_char *const a0 = lds + buf * HT_BYTES;
    bf16x8 a[3];
    a[0] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8 *>(a0 + aoff[0]);
    a[1] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8 *>(a0 + aoff[1]);
#pragma unroll
    for (int kk = 0; kk < H_KSTEPS; ++kk) {
      // (pinned: left alone, the scheduler sinks the read to its first use)
      if (kk + 2 < H_KSTEPS) a[(kk + 2) % 3] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8 *>(a0 + aoff[kk + 2]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        if (kk == 0) {
          f32x16 z;
#pragma unroll
          for (int r = 0; r < 16; ++r) z[r] = 0.0f;
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kk % 3], wreg[j][kk], z, 0, 0, 0);
        } else {
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kk % 3], wreg[j][kk], acc[j], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    prev = tile;
    tile = nxt;
    buf ^= 1;
    ++it;
  };
  bool lastA = true;
  while (tile < ntiles) {
    run(accA, accB);
    lastA = true;
    if (tile >= ntiles) break;
    run(accB, accA);
    lastA = false;
  }
  if (prev >= 0) {
    if constexpr (GATHER) __syncthreads();   // (a workgroup with ONE tile: its indices were written just before the loop)
    if (lastA) store_tile(accA, prev, (it - 1) & 3); else store_tile(accB, prev, (it - 1) & 3);
  }
}

size_t head_bf16_weight_bytes(int cout) { return (size_t)4 * (cout == 256 ? 2 : 1) * H_KSTEPS * 64 * 16; }

// Wb: [cout][256] bf16 bit patterns -> the fragment-order table the kernel's waves load once
void head_bf16_pack_weights(const unsigned short *Wb, int cout, unsigned char *dst) {
  const int ntw = cout == 256 ? 2 : 1;
  memset(dst, 0, head_bf16_weight_bytes(cout));
  for (int w = 0; w < 4; ++w)
    for (int j = 0; j < ntw; ++j)
      for (int kk = 0; kk < H_KSTEPS; ++kk)
        for (int ln = 0; ln < 64; ++ln) {
          const int l31 = ln & 31, hi = ln >> 5;
          const int co = cout == 256 ? w * 64 + 2 * l31 + j : w * 32 + l31;
          if (co >= cout) continue;
          unsigned char *o = dst + ((((size_t)w * ntw + j) * H_KSTEPS + kk) * 64 + ln) * 16;
          memcpy(o, Wb + (size_t)co * 256 + 16 * kk + 8 * hi, 16);
        }
}

template <int COUT, bool GATHER>
static hipError_t launch_head(const void *in_bf16, int in_choff, const void *wpack, const float *bias, float *out, int npix,
                              const int *list, const int *total, int max_walk, int num_cus, int tiles_per_wg, hipStream_t s) {
  constexpr size_t lds = 2 * (size_t)HT_BYTES + (GATHER ? 4 * HT * sizeof(int) : 0);
  auto k = head1x1_bf16_kernel<COUT, 512, GATHER>;
  if (npix <= 0 || max_walk <= 0) return hipSuccess;
  const int ntiles = (max_walk + HT - 1) / HT;
  int grid = (num_cus > 0 ? num_cus : 256) * H_WG_PER_CU;
  if (grid > ntiles) grid = ntiles;
  if (GATHER && tiles_per_wg > 1) grid = std::max(1, std::min(grid, (ntiles + tiles_per_wg - 1) / tiles_per_wg));
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, s, reinterpret_cast<const unsigned short *>(in_bf16), in_choff,
                     reinterpret_cast<const unsigned char *>(wpack), bias, out, npix, list, total);
  return hipGetLastError();
}

// in_bf16: [npix][512] = ReLU(convPa) | ReLU(convDa); cout 256: the descriptor head on channels 256..511,
// cout 65: the detector head on channels 0..255
hipError_t launch_head1x1_bf16(const void *in_bf16, const void *wpack, const float *bias, float *out, int npix, int cout,
                               hipStream_t s) {
  if (cout == 256) return launch_head<256, false>(in_bf16, 256, wpack, bias, out, npix, nullptr, nullptr, npix, 0, 0, s);
  if (cout == 65) return launch_head<65, false>(in_bf16, 0, wpack, bias, out, npix, nullptr, nullptr, npix, 0, 0, s);
  return hipErrorInvalidValue;
}

// The descriptor head on the `*total` (<= max_total) rows that `list` names, of the npix rows of in_bf16 / out.
hipError_t launch_head1x1_bf16_gather(const void *in_bf16, const void *wpack, const float *bias, float *out, int npix,
                                      const int *list, const int *total, int max_total, int tiles_per_wg, hipStream_t s) {
  if (!list || !total) return hipErrorInvalidValue;
  // the kernel forms a listed row's byte offset (row index x 1024) in 32 bits, with 0x80000000 as its out-of-range marker
  if ((long long)npix * 1024 >= (1ll << 31)) return hipErrorInvalidValue;
  return launch_head<256, true>(in_bf16, 256, wpack, bias, out, npix, list, total, max_total, 0, tiles_per_wg, s);
}

}  // namespace spfe
