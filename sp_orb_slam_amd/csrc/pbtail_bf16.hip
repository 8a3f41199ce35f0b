// pbtail_bf16.hip — bf16 mode: the detector head's 1x1 convolution convPb (256 -> 65, no ReLU) and the detector tail in ONE
// kernel (/root/reference/orb_slam2/src/cv/sp_extractor.cpp:97 `semi = convPb(relu(convPa(x)))`, :105-131 softmax / dust
// slices / arg-max / threshold / log-heat + pixel_shuffle): the bf16 counterpart of pbtail_f32.hip.
//
// Why.  As two launches on the launch stream — head1x1_bf16_kernel<65> (16.9 us per eight 1280x720 frames) and tail_kernel
// (25.7 us, which reads the 30 MB of logits back) — the detector head costs 42.6 us of a 1.05 ms step, most of it two kernel
// start-ups and one HBM round trip of the logits.  Here a workgroup owns 32 cells of one frame:
//   * its wavefronts bring the cells' 32 x 256 bf16 activations (16 KB) HBM -> LDS with LDS-direct loads (head_bf16.hip's
//     swizzle: conflict-free 16-byte fragment reads);
//   * three 32-channel tiles on v_mfma_f32_32x32x16_bf16 — 16 K steps each, weights in registers, loaded while the
//     activations are in flight; the third tile holds the dustbin channel alone (the MFMA's internal summation order is the
//     hardware's, so the channel stays on the MFMA: same bits as head1x1_bf16_kernel<65>);
//   * the f32 logits go to HBM (`semi`: spfe_debug_read, tests) AND into the LDS the activations held, and two wavefronts
//     run the tail on them, 16 cells each — tail_body.h, the code tail_kernel runs.
// Two forms (template NW).  NW = 4, for launches of one round of workgroups (a single frame): the shortest chain — four
// wavefronts share the load, three run a tile each, two the tail.  NW = 2, for full launches: wavefront 1 runs the dustbin
// tile behind its own on the same registers and BOTH wavefronts run the tail.  Measured (1280x720 x 8, rocprofv3): the
// two launches 17.3 + 26.0 us; NW = 4 41.5 us; NW = 2 38.5 us.  The ablations of the NW = 4 form said why fusing buys so
// little there: an empty kernel of that shape is 10.5 us of dispatch, the head part alone 21.6 us, the tail part alone 33 us
// with 4 tail wavefronts per SIMD against tail_kernel's 26 with 8 — the tail is ~1500 VALU instructions per lane (33 IEEE
// divisions, 17 spfe_expf, 16 spfe_logf: ~22 us of VALU issue for the chip), workgroups of one round run their memory phase
// and their VALU phase in lockstep, and what bounds the residency is LDS (16 KB of activations per 32 cells).  Whole path:
// 1280x720 x 8 +0.5 %, 752x480 x 8 +1.8 %, single frames -3 ... -11 us (752x480: 0.300 -> 0.289 ms).
// K order (16 ascending MFMA steps from a zero accumulator) and `acc + bias` are head_bf16.hip's, a cell's result does not
// depend on which cells share its tile: `semi` is bit-identical to the two-launch path's and everything behind it unchanged
// (tests/test_gpu_bf16.py::test_bf16_convPb_inside_the_tail_launch_is_bit_identical).
#include <cstdlib>
#include <cstring>

#include "spfe_kernels.h"
#include "tail_body.h"

namespace spfe {

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) char lds_char;
constexpr int PT = TAIL_CELLS_PER_WG;   // 32 cells per workgroup
constexpr int PT_BYTES = PT * 512;      // 256 input channels (bf16) per cell
constexpr int P_KSTEPS = 16;            // K = 256, sixteen per MFMA
constexpr int IN_STRIDE = 512;          // head activations: [cell][ReLU(convPa) 256 | ReLU(convDa) 256] bf16
}  // namespace

// head: [B * C][512] bf16; wpack: head_bf16_pack_weights(convPb, 65) ([wave 4][K step 16][lane 64][8 bf16]; waves 0 .. 2 are
// read); bias: [>= 65]; zero_ints / nzero: the bf16 convolutions' tile-queue counters, cleared for the NEXT call (every
// convolution of this call is behind this launch in stream order)
template <int NW>
__global__ __launch_bounds__(64 * NW) void pbtail_bf16_kernel(const unsigned short *__restrict__ head,
                                                          const unsigned char *__restrict__ wpack,
                                                          const float *__restrict__ bias, float *__restrict__ semi_out,
                                                          FrameBufs f, RecordLayout rl, int H, int W, int nparts, int b0,
                                                          int *zero_ints, int nzero, int zstride) {
  // (the logits take the activations' place once both wavefronts have read their operands: 16 KB per workgroup, nine per CU)
  __shared__ __attribute__((aligned(16))) char sA[PT_BYTES];
  static_assert(PT * SPFE_SEMI_CH * 4 <= PT_BYTES, "the logits reuse the activation tile");
  float *const sm = reinterpret_cast<float *>(sA);
  __shared__ float smin[2], smax[2];
  if (zero_ints && blockIdx.x == 0 && blockIdx.y == 0)
    for (int i = threadIdx.x; i < nzero; i += 64 * NW) zero_ints[(i >> 5) * zstride + (i & 31)] = 0;   // runs of 32, zstride apart
  const int wc = W >> 3, hc = H >> 3, C = hc * wc;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int b = blockIdx.y + b0;   // (b0: the first frame of this launch's part of the batch)
  const int cell0 = blockIdx.x * PT;                      // first cell of this workgroup, inside frame b
  const int ncell = C - cell0 < PT ? C - cell0 : PT;      // (>= 1 by the grid)
  lds_char *const lds = (lds_char *)sA;

  // this frame's rows only: rows past the frame's last cell read as zeros (their outputs are never stored)
  const unsigned short *frame_in = head + (size_t)b * C * IN_STRIDE;
  const __amdgpu_buffer_rsrc_t rin =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(frame_in), 0, (unsigned)((size_t)C * IN_STRIDE * 2), 0x00020000);
  (void)rin;
#if defined(__HIP_DEVICE_COMPILE__)
  // the tile's 1024 16-byte pieces = 16 LDS-direct passes, 16 / NW per wave: pass p, lane l -> LDS piece q = 64 p + l = (cell
  // q >> 5, slot q & 31), which holds the cell's piece slot ^ (cell & 31)
#pragma unroll
  for (int i = 0; i < 16 / NW; ++i) {
    const int p = NW * i + wave;
    const int q = p * 64 + lane, px = q >> 5, slot = q & 31;
    const unsigned src = (unsigned)(cell0 + px) * (unsigned)(IN_STRIDE * 2) + (unsigned)((slot ^ (px & 31)) * 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_void *)(lds + p * 1024), 16, src, 0, 0, 0);
  }
#endif
  // weights while the activations are in flight: B operand of K step kk of channel tile t = W[channel 32 t + l31][16 kk + 8 hi ..]
  // (table index = head_bf16.hip's wave index = channel tile)
  bf16x8 wreg[P_KSTEPS];
  auto load_w = [&](int t) {
#pragma unroll
    for (int kk = 0; kk < P_KSTEPS; ++kk)
      wreg[kk] = __builtin_bit_cast(bf16x8, reinterpret_cast<const u32x4 *>(wpack)[(t * P_KSTEPS + kk) * 64 + lane]);
  };
  if (wave < 3) load_w(wave);
  const int co = wave * 32 + l31;
  const float bv = co < 64 ? bias[co] : 0.0f, bdust = bias[64];
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's pieces have landed
  __syncthreads();                      // ... and the other wave's

  float *semi_g = semi_out + ((size_t)b * C + cell0) * SPFE_SEMI_CH;
  // A fragment of K step kk: piece 2 kk + hi of cell l31
  lds_char *const a0 = lds + (unsigned)(l31 * 512);
  auto rd = [&](int kk) -> bf16x8 {
    return *reinterpret_cast<const __attribute__((address_space(3))) bf16x8 *>(a0 + (((unsigned)(2 * kk + hi) ^ (unsigned)l31) & 31u) * 16u);
  };
  auto tile = [&]() -> f32x16 {
    f32x16 acc;
    bf16x8 a[3];
    a[0] = rd(0);
    a[1] = rd(1);
#pragma unroll
    for (int kk = 0; kk < P_KSTEPS; ++kk) {
      // (pinned: left alone, the scheduler sinks the read to its first use)
      if (kk + 2 < P_KSTEPS) a[(kk + 2) % 3] = rd(kk + 2);
      __builtin_amdgcn_sched_barrier(0);
      if (kk == 0) {
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.0f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], wreg[0], z, 0, 0, 0);
      } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kk % 3], wreg[kk], acc, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    return acc;
  };
  f32x16 acc = {}, accd = {};
  if (wave < 2) acc = tile();   // channels 32 wave + l31
  // the dustbin tile (channel 64 = its lanes 0 and 32): wavefront 2's own, or wavefront 1's second on the same registers
  constexpr int DW = NW == 4 ? 2 : 1;
  if (wave == DW) {
    if constexpr (NW == 2) load_w(2);
    accd = tile();
  }
  __syncthreads();   // every wavefront has read its operands: the tile becomes the logits' place
  // D[cell][channel]: register r = cell (r & 3) + 8 (r >> 2) + 4 hi of the tile, channel 32 tile + l31
  if (wave < 2) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int p = (r & 3) + 8 * (r >> 2) + 4 * hi;
      const float o = acc[r] + bv;
      sm[p * SPFE_SEMI_CH + co] = o;
      if (p < ncell) semi_g[(size_t)p * SPFE_SEMI_CH + co] = o;
    }
  }
  if (wave == DW && l31 == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int p = (r & 3) + 8 * (r >> 2) + 4 * hi;
      const float o = accd[r] + bdust;
      sm[p * SPFE_SEMI_CH + 64] = o;
      if (p < ncell) semi_g[(size_t)p * SPFE_SEMI_CH + 64] = o;
    }
  }
  __syncthreads();

  // ---- the tail on the 32 cells: two wavefronts, a DPP quad per cell.  NW = 2: both (the tail is ~1500 VALU instructions
  // per lane in dependent chains: in a full chip's launch every resident wavefront has to carry some).  NW = 4: wavefronts
  // 2, 3 of even workgroups and 0, 1 of odd ones (all four SIMDs of a CU) ----
  const int tw = NW == 2 ? wave : ((blockIdx.x & 1) ? wave : wave - 2);
  if (tw >= 0 && tw < 2) {
    const int q = lane & 3, lc = 16 * tw + (lane >> 2);
    float lmin = 0.0f, lmax = -1e30f;   // log-heat is <= 0
    uint8_t *rec = f.records + (size_t)b * rl.bytes;
    if (lc < ncell)
      tail_cell(&sm[lc * SPFE_SEMI_CH], q, cell0 + lc, wc, W, f.heat_log + (size_t)b * H * W,
                reinterpret_cast<float *>(rec + rl.off_sd), reinterpret_cast<float *>(rec + rl.off_dd),
                f.cell_score + (size_t)b * C, f.cell_k + (size_t)b * C, lmin, lmax);
    lmin = wave_min64(lmin);
    lmax = wave_max64(lmax);
    if (lane == 0) { smin[tw] = lmin; smax[tw] = lmax; }
  }
  __syncthreads();
  if (tid == 0) {
    float *part = reinterpret_cast<float *>(f.minmax) + ((size_t)b * nparts + blockIdx.x) * 2;
    part[0] = smin[1] < smin[0] ? smin[1] : smin[0];
    part[1] = smax[1] > smax[0] ? smax[1] : smax[0];
  }
}

// head: the bf16 head activations [B * C][512]; wpack / bias: convPb's (see the kernel); semi: [B][C][65] f32
hipError_t launch_pbtail_bf16(const void *head, const void *wpack, const float *bias, float *semi, const FrameBufs &f,
                              const RecordLayout &r, int B, int H, int W, hipStream_t s, int b0, int *zero_ints, int nzero,
                              int zstride, int force) {
  const int nparts = tail_parts(H, W);
  if ((size_t)(H / 8) * (W / 8) * IN_STRIDE * 2 >= ((size_t)1 << 32)) return hipErrorInvalidValue;   // (32-bit SRD offsets inside a frame)
  // few frames (one round of workgroups): four wavefronts share the load and the three channel tiles — the shorter chain
  // (single-frame calls: both forms within the +-4 us noise of the call's p50, 3 ... 8 us below the two launches); full
  // launches: two wavefronts, both on the tail (1280x720 x 8: 38.5 us against 41.5; the two launches it replaces: 17.3 +
  // 26.0).  force = 2 | 4 (SPFE_PBTAIL=2|4, tests) takes that form
  const bool four = force ? force == 4 : (long)nparts * B <= 1024;
  if (four)
    hipLaunchKernelGGL(pbtail_bf16_kernel<4>, dim3(nparts, B), dim3(256), 0, s, reinterpret_cast<const unsigned short *>(head),
                       reinterpret_cast<const unsigned char *>(wpack), bias, semi, f, r, H, W, nparts, b0, zero_ints, nzero, zstride);
  else
    hipLaunchKernelGGL(pbtail_bf16_kernel<2>, dim3(nparts, B), dim3(128), 0, s, reinterpret_cast<const unsigned short *>(head),
                       reinterpret_cast<const unsigned char *>(wpack), bias, semi, f, r, H, W, nparts, b0, zero_ints, nzero, zstride);
  return hipGetLastError();
}

}  // namespace spfe
