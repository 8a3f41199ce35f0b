// head_f32.hip — the two 1x1 head convolutions of the f32 mode (no ReLU; /root/reference/orb_slam2/src/cv/
// sp_extractor.cpp:96-100): convDb (256 -> 256) and convPb (256 -> 65),
//   out[P][COUT] = in[P][256 of the 512 head channels] x W^T + bias,  P = frames x cells,
// on v_mfma_f32_32x32x2_f32, BIT-IDENTICAL to the generic convolution kernel they replace (conv_f32.hip ran them as
// K = 256 "convolutions" at 24 % / 50 % of the f32 MFMA peak): the arithmetic contract of include/spfe_exact_math.h
// makes a 1x1 layer acc = +0; for k = 0 .. 255: acc = fmaf(x[k], w[k], acc); out = acc + bias, and the MFMA is a
// k-ordered fmaf chain, so 128 steps (k = 2 s, 2 s + 1) in ascending order are that chain.
//
// Same design as head_bf16.hip: the WEIGHTS LIVE IN REGISTERS — a wave owns 32 output channels for the whole kernel,
// 128 K steps = 128 VGPRs; eight waves per workgroup (two per SIMD: each other's MFMAs fill the issue gaps that the
// operand selects leave — with one wave per SIMD and 256 weight registers, half of them parked in AccVGPRs, the
// copies and selects between the 64-cycle MFMAs cost a third of the matrix time) — persistent workgroups, one per CU, walk 32-pixel tiles whose 32 KB come HBM -> LDS with LDS-direct loads into a double buffer,
// XOR-swizzled by the pixel on the source side (conflict-free 16-byte reads: one read feeds two K steps), and a lane's
// outputs leave as 4-byte stores in 128-byte runs.  These layers are matrix-bound (7.4 GFLOP per
// eight 752x480 frames = 47 us at the f32 peak).
// STATUS: opt-in (SPFE_F32_HEADS=1).  Measured (rocprofv3, 752x480 x 8): convDb 63 us, convPb 38.5 us — against 72 and 35.5 us
// for the generic kernel (the first version, four waves with 256 weight registers each, half of them in AccVGPRs: 66 + 38): no gain worth a second code path by default; kept, tested for bit-identity, as the record of VERDICT
// round 1 item 9 (whose "<= 0.05 ms" is the layers' roofline itself).
#include <algorithm>
#include <cstring>
#include <utility>

#include "spfe_kernels.h"

namespace spfe {

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2v __attribute__((__vector_size__(2 * sizeof(unsigned))));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) char lds_char;
constexpr int FT = 32;                  // pixels per tile
constexpr int FT_BYTES = FT * 1024;     // 256 input channels (f32) per pixel
constexpr int F_KSTEPS = 128;           // K = 256, two per MFMA
}  // namespace

// in: [npix][IN_STRIDE] f32, the head reads channels [in_choff, in_choff + 256); out: [npix][COUT] f32
// wpack: [wave 8][K step 128 / 4][lane 64][4] f32 (head_f32_pack_weights)
// GATHER: as in head_bf16.hip — the pixels are the `*total` cells of `list` (select_kernel's list of the cells the descriptor
// sampling reads), row list[p] of `in` -> row list[p] of `out`, the dense kernel's bits in the rows anybody reads.
template <int COUT, int IN_STRIDE, bool GATHER>
__global__ __launch_bounds__(512, 1) void head1x1_f32_kernel(const float *__restrict__ in, int in_choff,
                                                             const float *__restrict__ wpack, const float *__restrict__ bias,
                                                             float *__restrict__ out, int npix,
                                                             const int *__restrict__ list, const int *__restrict__ total) {
  constexpr int NTW = 1;
  extern __shared__ __attribute__((aligned(16))) char sm_f[];
  lds_char *const lds = (lds_char *)sm_f;
  int *const sIdx = reinterpret_cast<int *>(sm_f + 2 * FT_BYTES);   // GATHER: [4][32] cell indices of the tiles in flight
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int nwalk = GATHER ? __builtin_amdgcn_readfirstlane(*total) : npix;   // pixels this launch walks
  const int ntiles = (nwalk + FT - 1) / FT;

  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(in) + in_choff, 0, (unsigned)((size_t)npix * IN_STRIDE * 4 - (size_t)in_choff * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rout =
      __builtin_amdgcn_make_buffer_rsrc(out, 0, (unsigned)((size_t)npix * COUT * 4), 0x00020000);
  (void)rin;   // (the host pass of hipcc does not see the uses below)

  // this wave's weights, for the whole kernel: B operand of step s = W[channel of lane & 31][k = 2 s + hi]
  float wreg[NTW][F_KSTEPS];
#pragma unroll
  for (int j = 0; j < NTW; ++j)
#pragma unroll
    for (int s4 = 0; s4 < F_KSTEPS / 4; ++s4) {
      const f32x4 v = reinterpret_cast<const f32x4 *>(wpack)[((wave * NTW + j) * (F_KSTEPS / 4) + s4) * 64 + lane];
      wreg[j][4 * s4] = v.x; wreg[j][4 * s4 + 1] = v.y; wreg[j][4 * s4 + 2] = v.z; wreg[j][4 * s4 + 3] = v.w;
    }
  // output channel of this lane
  const int co = wave * 32 + l31;
  float bv[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) bv[j] = co + j < COUT ? bias[co + j] : 0.0f;
  const bool lane_out = co < COUT;
  const bool wave_active = wave * 32 < COUT;   // (convPb: waves 3..7 only help with the loads)

  // a tile's 2048 16-byte pieces = 32 LDS-direct passes, 4 per wave: pass p, lane l -> LDS piece q = 64 p + l = (pixel
  // q >> 6, slot q & 63), which holds the pixel's piece slot ^ (pixel & 15)
  unsigned dsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = (8 * i + wave) * 64 + lane, px = q >> 6, slot = q & 63;
    dsrc[i] = (unsigned)px * (unsigned)(IN_STRIDE * 4) + (unsigned)((slot ^ (px & 15)) * 16);
  }
  // GATHER: pass i of this wave carries the tile's pixel 8 i + wave; the cell indices are fetched one tile ahead
  int gidx[4] = {-1, -1, -1, -1};
  auto load_idx = [&](int tile) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = tile * FT + 8 * i + wave;
      gidx[i] = p < nwalk ? list[p] : -1;
    }
  };
  auto dma = [&](int tile, int buf, int ring) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned base = (unsigned)tile * (unsigned)(FT * IN_STRIDE * 4);   // (past the last pixel: out of range -> zeros)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned src = base + dsrc[i];
      if constexpr (GATHER) {
        const unsigned inrow = (unsigned)(lane ^ ((8 * i + wave) & 15)) * 16u;
        src = gidx[i] < 0 ? 0x80000000u : (unsigned)gidx[i] * (unsigned)(IN_STRIDE * 4) + inrow;
        if (lane == 0) sIdx[ring * FT + 8 * i + wave] = gidx[i];
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_void *)(lds + buf * FT_BYTES + (8 * i + wave) * 1024), 16, src, 0, 0, 0);
    }
#endif
  };
  const unsigned arow = (unsigned)(l31 * 1024);
  const unsigned akey = (unsigned)(l31 & 15);

  // D[pixel][channel]: register r = pixel (r & 3) + 8 (r >> 2) + 4 hi of the tile
  auto store_tile = [&](const f32x16 (&acc)[NTW], int tile, int ring) {
    const unsigned base = (unsigned)tile * (unsigned)(FT * COUT * 4) + (unsigned)co * 4u;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      unsigned off = lane_out ? base + (unsigned)(((r & 3) + 8 * (r >> 2) + 4 * hi) * COUT * 4) : 0x80000000u;
      if constexpr (GATHER) {
        const int cell = sIdx[ring * FT + (r & 3) + 8 * (r >> 2) + 4 * hi];
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
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this tile has landed
    __syncthreads();                      // ... for every wave; and every wave is done reading the other buffer
    const int nxt = tile + (int)gridDim.x;
    if (nxt < ntiles) {
      dma(nxt, buf ^ 1, (it + 1) & 3);
      if constexpr (GATHER) load_idx(nxt + (int)gridDim.x);
    }
    if (prev >= 0 && wave_active) store_tile(accPrev, prev, (it - 1) & 3);   // the previous tile's outputs leave while this one computes
    lds_char *const a0 = lds + buf * FT_BYTES + arow;
    if (wave_active) {
    // piece m = channels 4 m .. 4 m + 3 of this lane's pixel: K steps 2 m (dwords 0 | 1 by hi) and 2 m + 1 (dwords 2 | 3)
    f32x4 pc[3];
    auto rd = [&](int m) -> f32x4 {
      return *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>(a0 + (((unsigned)m ^ akey) * 16u));
    };
    pc[0] = rd(0);
    pc[1] = rd(1);
#pragma unroll
    for (int m = 0; m < F_KSTEPS / 2; ++m) {
      // (pinned: left alone, the scheduler sinks the read to its first use and every 4 MFMAs wait for an LDS round trip)
      if (m + 2 < F_KSTEPS / 2) pc[(m + 2) % 3] = rd(m + 2);
      __builtin_amdgcn_sched_barrier(0);
      const f32x4 v = pc[m % 3];
      const float a_even = hi ? v.y : v.x, a_odd = hi ? v.w : v.z;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int s = 2 * m + h;
        const float av = h ? a_odd : a_even;
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
          if (s == 0) {
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.0f;
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wreg[j][s], z, 0, 0, 0);
          } else {
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wreg[j][s], acc[j], 0, 0, 0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
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
  if (prev >= 0 && wave_active) { if (lastA) store_tile(accA, prev, (it - 1) & 3); else store_tile(accB, prev, (it - 1) & 3); }
}

size_t head_f32_weight_bytes(int) { return (size_t)8 * F_KSTEPS * 64 * 4; }

// W: [cout][256] f32 -> the fragment-order table the kernel's waves load once:
// [wave][s / 4][lane][s % 4] = W[32 wave + (lane & 31)][2 s + (lane >> 5)]
void head_f32_pack_weights(const float *W, int cout, float *dst) {
  memset(dst, 0, head_f32_weight_bytes(cout));
  for (int w = 0; w < 8; ++w)
    for (int s = 0; s < F_KSTEPS; ++s)
      for (int ln = 0; ln < 64; ++ln) {
        const int l31 = ln & 31, hi = ln >> 5, co = w * 32 + l31;
        if (co >= cout) continue;
        dst[(((size_t)w * (F_KSTEPS / 4) + s / 4) * 64 + ln) * 4 + s % 4] = W[(size_t)co * 256 + 2 * s + hi];
      }
}

template <int COUT, bool GATHER>
static hipError_t launch_head_f32(const float *in, int in_choff, const float *wpack, const float *bias, float *out, int npix,
                                  const int *list, const int *total, int max_walk, int num_cus, int tiles_per_wg, hipStream_t s) {
  constexpr size_t lds = 2 * (size_t)FT_BYTES + (GATHER ? 4 * FT * sizeof(int) : 0);
  auto k = head1x1_f32_kernel<COUT, 512, GATHER>;
  if (npix <= 0 || max_walk <= 0) return hipSuccess;
  static bool attr_done[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) attr_done[dev] = true;
  }
  const int ntiles = (max_walk + FT - 1) / FT;
  int grid = num_cus > 0 ? num_cus : 256;
  if (grid > ntiles) grid = ntiles;
  if (GATHER && tiles_per_wg > 1) grid = std::max(1, std::min(grid, (ntiles + tiles_per_wg - 1) / tiles_per_wg));
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, s, in, in_choff, wpack, bias, out, npix, list, total);
  return hipGetLastError();
}

// in: [npix][512] f32 = ReLU(convPa) | ReLU(convDa); cout 256: the descriptor head on channels 256..511,
// cout 65: the detector head on channels 0..255
hipError_t launch_head1x1_f32(const float *in, const float *wpack, const float *bias, float *out, int npix, int cout,
                              hipStream_t s) {
  if (cout == 256) return launch_head_f32<256, false>(in, 256, wpack, bias, out, npix, nullptr, nullptr, npix, 0, 0, s);
  if (cout == 65) return launch_head_f32<65, false>(in, 0, wpack, bias, out, npix, nullptr, nullptr, npix, 0, 0, s);
  return hipErrorInvalidValue;
}

// The descriptor head on the `*total` (<= max_total) rows that `list` names, of the npix rows of in / out.
hipError_t launch_head1x1_f32_gather(const float *in, const float *wpack, const float *bias, float *out, int npix,
                                     const int *list, const int *total, int max_total, int tiles_per_wg, hipStream_t s) {
  if (!list || !total) return hipErrorInvalidValue;
  // the kernel forms a listed row's byte offset (row index x 2048) in 32 bits, with 0x80000000 as its out-of-range marker
  if ((long long)npix * 2048 >= (1ll << 31)) return hipErrorInvalidValue;
  return launch_head_f32<256, true>(in, 256, wpack, bias, out, npix, list, total, max_total, 0, tiles_per_wg, s);
}

}  // namespace spfe
