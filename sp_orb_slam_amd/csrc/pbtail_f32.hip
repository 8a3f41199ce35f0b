// pbtail_f32.hip — f32 mode: the detector head's 1x1 convolution convPb (256 -> 65, no ReLU) and the detector tail in ONE
// kernel (/root/reference/orb_slam2/src/cv/sp_extractor.cpp:97 `semi = convPb(relu(convPa(x)))`, :105-131 softmax / dust
// slices / arg-max / threshold / log-heat + pixel_shuffle).
//
// Why.  As a launch of the generic convolution kernel convPb ran at 0.25 of the f32 MFMA peak (65 output channels are two
// 32-channel tiles + ONE channel that cost a third tile; K = 256 is 128 MFMAs behind a full persistent-kernel start-up), and
// the tail then read the 65 logits per cell back from HBM: 39 + 12.6 us per eight 752x480 frames, 15.5 + 7.8 us of a single
// frame's 0.75 ms.  Here a workgroup owns 32 cells of one frame:
//   * all four wavefronts bring the cells' 32 x 256 activations HBM -> LDS with LDS-direct loads (head_f32.hip's swizzle:
//     a lane's 16-byte reads are conflict-free);
//   * wavefronts 0 and 1 run the two full 32-channel tiles on v_mfma_f32_32x32x2_f32 with their weights in REGISTERS (128 K
//     steps = 128 VGPRs, loaded while the activations are in flight);
//   * wavefront 2 computes the dustbin channel (channel 64) as a plain fmaf chain on the VALU — lane = cell, 256 steps in
//     ascending k: the arithmetic contract's chain itself (include/spfe_exact_math.h; the MFMA is bitwise that chain), so
//     the third MFMA tile is gone;
//   * the logits go to HBM (`semi`: spfe_debug_read, tests) AND to LDS, and after one barrier wavefronts 2 and 3 run the
//     tail on them — tail_body.h, the code tail_kernel runs: same bits.
// One launch instead of two, no third MFMA tile, no logits read back.  K order, accumulation from +0 and `acc + bias` are
// the contract's, so `semi` is bit-identical to the generic kernel's and everything behind it is unchanged.
#include <cstring>

#include "spfe_kernels.h"
#include "tail_body.h"

namespace spfe {

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) char lds_char;
constexpr int PT = TAIL_CELLS_PER_WG;   // 32 cells per workgroup
constexpr int PT_BYTES = PT * 1024;     // 256 input channels (f32) per cell
constexpr int P_KSTEPS = 128;           // K = 256, two per MFMA
constexpr int IN_STRIDE = 512;          // head activations: [cell][ReLU(convPa) 256 | ReLU(convDa) 256]
}  // namespace

// head: [B * C][512] f32; wpack: head_f32_pack_weights(convPb, 65) ([wave][s / 4][lane][4]; waves 0, 1 are read);
// wdust: convPb's row 64, [256]; bias: [>= 65]
__global__ __launch_bounds__(256, 2) void pbtail_f32_kernel(const float *__restrict__ head, const float *__restrict__ wpack,
                                                            const float *__restrict__ wdust, const float *__restrict__ bias,
                                                            float *__restrict__ semi_out, FrameBufs f, RecordLayout rl, int H,
                                                            int W, int nparts, int b0) {
  __shared__ __attribute__((aligned(16))) char sA[PT_BYTES];
  __shared__ __attribute__((aligned(16))) float sW[256];
  __shared__ float sm[PT * SPFE_SEMI_CH];
  __shared__ float smin[2], smax[2];
  const int wc = W >> 3, hc = H >> 3, C = hc * wc;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int b = blockIdx.y + b0;   // (b0: the first frame of this launch's part of the batch)
  const int cell0 = blockIdx.x * PT;                      // first cell of this workgroup, inside frame b
  const int ncell = C - cell0 < PT ? C - cell0 : PT;      // (>= 1 by the grid)
  lds_char *const lds = (lds_char *)sA;

  // this frame's rows only: rows past the frame's last cell read as zeros (their outputs are never stored)
  const float *frame_in = head + (size_t)b * C * IN_STRIDE;
  const __amdgpu_buffer_rsrc_t rin =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(frame_in), 0, (unsigned)((size_t)C * IN_STRIDE * 4), 0x00020000);
  (void)rin;
#if defined(__HIP_DEVICE_COMPILE__)
  // the tile's 2048 16-byte pieces = 32 LDS-direct passes, 8 per wave: pass p, lane l -> LDS piece q = 64 p + l = (cell
  // q >> 6, slot q & 63), which holds the cell's piece slot ^ (cell & 15)
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int p = 4 * i + wave, slot = lane;
    const unsigned src = (unsigned)(cell0 + p) * (unsigned)(IN_STRIDE * 4) + (unsigned)((slot ^ (p & 15)) * 16);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_void *)(lds + p * 1024), 16, src, 0, 0, 0);
  }
#endif
  // weights while the activations are in flight: B operand of step s = W[channel 32 wave + (lane & 31)][k = 2 s + hi]
  float wreg[P_KSTEPS];
  float bv = 0.0f;
  if (wave < 2) {
#pragma unroll
    for (int s4 = 0; s4 < P_KSTEPS / 4; ++s4) {
      const f32x4 v = reinterpret_cast<const f32x4 *>(wpack)[(wave * (P_KSTEPS / 4) + s4) * 64 + lane];
      wreg[4 * s4] = v.x; wreg[4 * s4 + 1] = v.y; wreg[4 * s4 + 2] = v.z; wreg[4 * s4 + 3] = v.w;
    }
    bv = bias[wave * 32 + l31];
  } else if (wave == 2) {
    reinterpret_cast<f32x4 *>(sW)[lane] = reinterpret_cast<const f32x4 *>(wdust)[lane];
    bv = bias[64];
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's pieces have landed
  __syncthreads();                      // ... and everybody's

  float *semi_g = semi_out + ((size_t)b * C + cell0) * SPFE_SEMI_CH;
  lds_char *const a0 = lds + (unsigned)(l31 * 1024);
  const unsigned akey = (unsigned)(l31 & 15);
  auto rd = [&](int m) -> f32x4 {   // piece m = channels 4 m .. 4 m + 3 of this lane's cell
    return *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>(a0 + (((unsigned)m ^ akey) * 16u));
  };
  if (wave < 2) {
    f32x16 acc;
    f32x4 pc[3];
    pc[0] = rd(0);
    pc[1] = rd(1);
#pragma unroll
    for (int m = 0; m < P_KSTEPS / 2; ++m) {
      // (pinned: left alone, the scheduler sinks the read to its first use and every MFMA pair waits for an LDS round trip)
      if (m + 2 < P_KSTEPS / 2) pc[(m + 2) % 3] = rd(m + 2);
      __builtin_amdgcn_sched_barrier(0);
      const f32x4 v = pc[m % 3];
      const float a_even = hi ? v.y : v.x, a_odd = hi ? v.w : v.z;   // K steps 2 m (k = 4 m + hi) and 2 m + 1 (k = 4 m + 2 + hi)
      if (m == 0) {
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.0f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_even, wreg[0], z, 0, 0, 0);
      } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_even, wreg[2 * m], acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_odd, wreg[2 * m + 1], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // D[cell][channel]: register r = cell (r & 3) + 8 (r >> 2) + 4 hi of the tile, channel 32 wave + l31
    const int co = wave * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int p = (r & 3) + 8 * (r >> 2) + 4 * hi;
      const float o = acc[r] + bv;
      sm[p * SPFE_SEMI_CH + co] = o;
      if (p < ncell) semi_g[(size_t)p * SPFE_SEMI_CH + co] = o;
    }
  } else if (wave == 2) {
    // the dustbin logit of cell l31 (both lane halves run the same chain: identical addresses, broadcast reads)
    float acc = 0.0f;
    const f32x4 *w4 = reinterpret_cast<const f32x4 *>(sW);
#pragma unroll 8
    for (int m = 0; m < 64; ++m) {
      const f32x4 v = rd(m), w = w4[m];
      acc = fmaf(v.x, w.x, acc);
      acc = fmaf(v.y, w.y, acc);
      acc = fmaf(v.z, w.z, acc);
      acc = fmaf(v.w, w.w, acc);
    }
    const float o = acc + bv;
    if (hi == 0) {
      sm[l31 * SPFE_SEMI_CH + 64] = o;
      if (l31 < ncell) semi_g[(size_t)l31 * SPFE_SEMI_CH + 64] = o;
    }
  }
  __syncthreads();

  // ---- the tail on the 32 cells: wavefronts 2 and 3 (the SIMDs that carried no MFMAs; a co-resident workgroup's matrix
  // wavefronts overlap with them), a DPP quad per cell ----
  if (wave >= 2) {
    const int q = lane & 3, lc = 16 * (wave - 2) + (lane >> 2);
    float lmin = 0.0f, lmax = -1e30f;   // log-heat is <= 0
    uint8_t *rec = f.records + (size_t)b * rl.bytes;
    if (lc < ncell)
      tail_cell(&sm[lc * SPFE_SEMI_CH], q, cell0 + lc, wc, W, f.heat_log + (size_t)b * H * W,
                reinterpret_cast<float *>(rec + rl.off_sd), reinterpret_cast<float *>(rec + rl.off_dd),
                f.cell_score + (size_t)b * C, f.cell_k + (size_t)b * C, lmin, lmax);
    lmin = wave_min64(lmin);
    lmax = wave_max64(lmax);
    if (lane == 0) { smin[wave - 2] = lmin; smax[wave - 2] = lmax; }
  }
  __syncthreads();
  if (tid == 0) {
    float *part = reinterpret_cast<float *>(f.minmax) + ((size_t)b * nparts + blockIdx.x) * 2;
    part[0] = smin[1] < smin[0] ? smin[1] : smin[0];
    part[1] = smax[1] > smax[0] ? smax[1] : smax[0];
  }
}

// head: the f32 head activations [B * C][512]; wpack / wdust / bias: convPb's (see the kernel); semi: [B][C][65]
hipError_t launch_pbtail_f32(const float *head, const float *wpack, const float *wdust, const float *bias, float *semi,
                             const FrameBufs &f, const RecordLayout &r, int B, int H, int W, hipStream_t s, int b0) {
  const int nparts = tail_parts(H, W);
  if ((size_t)(H / 8) * (W / 8) * IN_STRIDE * 4 >= ((size_t)1 << 32)) return hipErrorInvalidValue;   // (32-bit SRD offsets inside a frame)
  hipLaunchKernelGGL(pbtail_f32_kernel, dim3(nparts, B), dim3(256), 0, s, head, wpack, wdust, bias, semi, f, r, H, W, nparts, b0);
  return hipGetLastError();
}

}  // namespace spfe
