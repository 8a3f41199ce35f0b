// conv_f32_kc.hip — exact-f32 3x3 convolutions (bias, ReLU, no pool) of a SINGLE FRAME's low-resolution layers on
// v_mfma_f32_16x16x4_f32: the "K-chain" kernel (/root/reference/orb_slam2/src/cv/sp_extractor.cpp:88-99: conv3a, conv4a,
// conv4b, convPa [| convDa] — the reference's one call shape is one frame per operator(), :361-514).
//
// Why another kernel.  Bit-exactness pins every output to ONE sequential fmaf chain over K = 9 x Cin (include/
// spfe_exact_math.h); K cannot be split across wavefronts.  On v_mfma_f32_32x32x2_f32 a 32 x 32 output tile is 576 dependent
// MFMAs of 64 cycles for K = 1152: 36.9 k cycles ~ 17 us that no amount of parallelism shortens, and conv4a of one
// 752x480 frame is only 705 such tiles for 1024 SIMDs — conv_f32.hip's 2-row tiles run it in 23 us with a third of the
// matrix pipes idle.  v_mfma_f32_16x16x4_f32 has the same throughput (64 flop / clk / SIMD) but advances a chain by FOUR k per
// 32-cycle issue (40 cycles dependent), and it is bitwise the same ascending-k fmaf chain (tools/microbench/
// mfma_chain_probe.hip: 0 mismatches of 51,200): the layer becomes 2,880 chains of 288 steps, 11.25 per CU — every
// SIMD busy, 3 chains interleaved per wavefront (96 cycles between dependent MFMAs: the 40-cycle latency is hidden).
// What this does NOT do is beat the layer's roofline: 1.66 GFLOP are 10.6 us at the f32 MFMA peak however they are cut
// (DESIGN.md 4.1 had estimated "~5 us" from the chain latency alone — that is the floor of ONE chain, not of the layer).
//
// Shape.  A workgroup = (frame, strip of R image rows, 16 output channels); R = 2 when two rows are <= 12 pixel groups of
// 16 (752x480 / 8: 94 px = 6 groups), else 1.  Its <= 12 chains (R x groups) are dealt to 4 wavefronts, 3 each.  Per K chunk
// of 16 input channels the halo strip (1 KB blocks of 16 consecutive halo pixels x the chunk's 4 channel quads, as the
// LDS-direct loads deliver them: an A operand — 16 px x 4 channels — is 64 dwords 16 bytes apart, a 2-way bank conflict)
// and the chunk's 144 x 16 weights ([tap][k][16 ch]: a B operand is 64 consecutive floats) sit in LDS, double buffered; the
// next chunk travels L2 -> LDS (buffer_load ... lds) under the current chunk's 108 MFMAs per wavefront, one block per MFMA
// gap; one barrier per chunk.  One B read feeds three MFMAs.
// K order: chunk -> tap -> channel (4 per MFMA, ascending): the contract's.  Accumulation starts from C = 0; out = max(acc
// + bias, 0).  Same bits as conv_f32.hip (tests/test_gpu_parity.py::test_f32_k_chain_kernel_on_mfma_16x16x4_is_bit_identical).
//
// STATUS: opt-in (SPFE_KC=<layer mask>), NOT the default.  Measured on single frames (round 4; stage events, which read ~5 us
// high; same-box A/B against conv_f32.hip's 2-row tiles):
//   752x480   conv4a / conv4b 27.0 -> 29.2 us   convPa 45.4 -> 40.6 us (p50 of the call: +-0)   conv3a 39.6 -> 62.9 us
//   640x480   convPa 45.3 -> 40.2 us (p50 0.681 -> 0.674 ms)          1280x720   convPa 80 -> 108 us
// What the ablation builds said on the way (each run on the GPU box):
//   1. the matrix loop does what the arithmetic promised: with the staging removed the 2,880 chains of conv4a take 11 us
//      (the layer's roofline is 10.6) — once every operand read sits ALONE in an MFMA gap, two K steps ahead (four reads in
//      one gap: +16 cycles per MFMA; reads issued where they are used: 3x slower);
//   2. staging through registers in the SAME wavefront (10 buffer loads + 43 ds_writes per chunk) doubles the kernel — with or
//      without memory traffic (every load out of range: same time): a VMEM or LDS-store instruction between MFMAs holds a lone
//      wavefront's in-order stream for 100+ cycles (MI355X_MICROARCH.md prices an LDS-DMA piece at 60 - 185);
//   3. four producer wavefronts beside four consumers (512-thread workgroups) removed that — and that build spent ~19 us
//      before its FIRST barrier completed (workgroups that returned before it: 5 us; one barrier or nine: the same 24 us; 1 KB
//      or 70 KB of LDS: the same).  NOT a property of 512-thread workgroups as such: a stand-alone probe (tools/microbench/
//      barrier_probe.hip: 256 / 512 / 1024 threads, 16 ... 180 live registers, with and without 70 KB of dynamic LDS, half the
//      wavefronts leaving first, a producer / consumer role split with LDS and global traffic) runs every case in the same
//      6 us.  The cause in that build was not found; the form was dropped for the one below;
//   4. hence this form: 256 threads, LDS-direct staging (13 instead of 53 staging instructions per wavefront and chunk, no
//      VGPR round trip).  It still pays ~1 us per chunk for them, which a second co-resident workgroup hides (convPa: 480
//      workgroups, two per CU: faster) and a lone one does not (conv4a: 240 workgroups: slower).
// So bit-exact f32 at batch 1 is bounded by staging issue cost, not by the chain latency the 16x16x4 form removes.
#include <cstdlib>
#include <cstring>

#include "spfe_kernels.h"

namespace spfe {
namespace kc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;
constexpr int KC = 16, TAPS = 9, NCO = 16;           // K chunk (input channels), taps, output channels per workgroup
constexpr int WCHUNK = TAPS * KC * NCO;              // floats of a chunk's weight slab
constexpr int MAXG = 12, MAXCH = 3;                  // pixel groups per strip, chains per wavefront
constexpr int A_PASSES = 10, W_PASSES = 3;           // LDS-direct blocks (1 KB) per wavefront and chunk: halo <= 37 blocks (3 x 194 px at R = 1, G = 12; 25 at R = 2, G = 6), weights 9

struct Params {
  const float *in;       // [B][H][W][in_stride], channels [0, cin)
  const float *wpack;    // [cout / 16][cin / 16][tap 9][k 16][n 16]
  const float *bias;     // [cout]
  float *out;            // [B][H][W][out_stride] at out_choff
  int in_stride, out_stride, out_choff;
  int B, H, W, cin, cout;
  int R, G;              // rows per strip, pixel groups of 16 per row
};

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) char lds_char;

__global__ __launch_bounds__(256, 2) void conv_f32_kc_kernel(Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem_kc[];
  lds_char *const lds = (lds_char *)smem_kc;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = lane & 15, kk = lane >> 4;
  const int H = p.H, W = p.W, R = p.R, G = p.G;
  const int nstrip = (H + R - 1) / R, ncg = p.cout / NCO;
  // workgroup -> (channel group fastest: the strips' halos are shared in L2 by neighbouring workgroups)
  int wg = blockIdx.x;
  const int cg = wg % ncg; wg /= ncg;
  const int strip = wg % nstrip;
  const int b = wg / nstrip;
  const int y0 = strip * R;
  const int nchunk = p.cin / KC;
  const int hrows = R + 2, hcols = 16 * G + 2;
  const int npix = hrows * hcols, nblk = (npix + 15) >> 4;   // halo blocks of 16 pixels = 1 KB per K chunk
  const int HBYTES = nblk * 1024, BUFB = HBYTES + WCHUNK * 4;
  const unsigned pixb = (unsigned)p.in_stride * 4u;
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float *>(p.in + (size_t)b * H * W * p.in_stride), 0, (unsigned)((size_t)H * W * pixb), 0x00020000);
  const float *wbase = p.wpack + (size_t)cg * nchunk * WCHUNK;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(wbase), 0, (unsigned)((size_t)nchunk * WCHUNK * 4), 0x00020000);
  (void)rin; (void)rw;   // (the host pass of hipcc does not see the uses below)

  // ---- staging: LDS-direct.  Halo block blk = 16 consecutive pixels of the flattened halo strip x the chunk's 4 channel
  // quads: lane (quad q = lane >> 4, pixel pp = lane & 15) fetches the 16 bytes (pixel, quad) and they land at
  // blk * 1024 + q * 256 + pp * 16 — no VGPR round trip, no ds_write.  Wavefront w issues blocks w, w + 4, ...; a chunk's
  // weight slab is 9 more 1 KB blocks, already in LDS order in memory.
  unsigned aoff[A_PASSES];   // byte offset of this lane's (pixel, quad) in the frame, or OOB (padding, past the strip)
#pragma unroll
  for (int it = 0; it < A_PASSES; ++it) {
    const int blk = wave + 4 * it;
    const int P = blk * 16 + px;
    const int hr = P / hcols, hcx = P - hr * hcols;
    const int gy = y0 + hr - 1, gx = hcx - 1;
    const bool in = blk < nblk && P < npix && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
    aoff[it] = in ? (unsigned)(gy * W + gx) * pixb + (unsigned)kk * 16u : OOB;
  }
  auto dma_a = [&](int it, int chunk, int buf) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int blk = wave + 4 * it;
    if (blk < nblk)   // (wave-uniform)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_void *)(lds + buf * BUFB + blk * 1024), 16,
                                               aoff[it] == OOB || chunk >= nchunk ? OOB : aoff[it] + (unsigned)chunk * (KC * 4), 0, 0, 0);
#endif
  };
  auto dma_w = [&](int it, int chunk, int buf) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int blk = wave + 4 * it;
    if (blk < WCHUNK * 4 / 1024)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void *)(lds + buf * BUFB + HBYTES + blk * 1024), 16,
                                               chunk < nchunk ? (unsigned)(chunk * WCHUNK * 4 + blk * 1024 + lane * 16) : OOB, 0, 0, 0);
#endif
  };

  // ---- this wavefront's chains: chain c = wave + 4 j -> (row c / G, pixel group c % G); per (chain, tap) the byte address
  // of this lane's A element inside a halo buffer: pixel P -> (P >> 4) * 1024 + (P & 15) * 16, + kk * 4 (+ k4 * 256 per K step)
  const int nchain = R * G;
  bool have[MAXCH];
  unsigned aaddr[MAXCH][TAPS];
#pragma unroll
  for (int j = 0; j < MAXCH; ++j) {
    const int c = wave + 4 * j;
    have[j] = c < nchain;                        // (wave-uniform)
    const int cc = have[j] ? c : 0;
    const int row = cc / G, grp = cc - row * G;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      const int P = (row + t / 3) * hcols + 16 * grp + px + t % 3;
      aaddr[j][t] = (unsigned)((P >> 4) * 1024 + (P & 15) * 16 + kk * 4);
    }
  }
  const unsigned waddr = (unsigned)(HBYTES + (kk * NCO + px) * 4);   // B operand: [tap][k][n] -> ((tap * 16 + 4 k4 + kk) * 16 + n) * 4

#pragma unroll
  for (int it = 0; it < A_PASSES; ++it) dma_a(it, 0, 0);
#pragma unroll
  for (int it = 0; it < W_PASSES; ++it) dma_w(it, 0, 0);
  f32x4 acc[MAXCH];
#pragma unroll
  for (int j = 0; j < MAXCH; ++j) acc[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};   // the contract's chain starts from +0
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wavefront's blocks of chunk 0 have landed
  __syncthreads();
#pragma unroll 1
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    const int cur = chunk & 1;
    const unsigned boff = (unsigned)(cur * BUFB);
    // 36 K steps (tap, channel quad); the operands of step s + 2 — one B and three A values — are read while the MFMAs of
    // step s issue, ONE read per MFMA gap (pinned: left alone, the scheduler sinks every read to its first use and each MFMA
    // waits out an LDS round trip; four reads in one gap overran the 32-cycle shadow); the next chunk's LDS-direct blocks go
    // out one per gap from gap 2 on
    float av[3][MAXCH], bw[3];
    auto rd = [&](int st, int slice) {
      const int slot = st % 3;
      const int tap = st >> 2, k4 = st & 3;
      if (slice == 0)
        bw[slot] = *reinterpret_cast<const __attribute__((address_space(3))) float *>(lds + boff + waddr + (tap * KC + 4 * k4) * NCO * 4);
      av[slot][slice] = *reinterpret_cast<const __attribute__((address_space(3))) float *>(lds + boff + aaddr[slice][tap] + k4 * 256);
    };
#pragma unroll
    for (int j = 0; j < MAXCH; ++j) { rd(0, j); rd(1, j); }
#pragma unroll
    for (int st = 0; st < TAPS * 4; ++st) {
      // (a wavefront with fewer than three chains runs the spare ones on chain 0's operands and drops them: the workgroup
      // lasts as long as its fullest wavefront anyway, and the loop keeps no branch between its MFMAs)
#pragma unroll
      for (int j = 0; j < MAXCH; ++j) {
        if (st + 2 < TAPS * 4) rd(st + 2, j);
        const int g = 3 * st + j;
        if (g >= 2 && g < 2 + A_PASSES) dma_a(g - 2, chunk + 1, cur ^ 1);
        else if (g >= 2 + A_PASSES && g < 2 + A_PASSES + W_PASSES) dma_w(g - 2 - A_PASSES, chunk + 1, cur ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[st % 3][j], bw[st % 3], acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the next chunk's blocks (issued ~100 MFMAs ago) have landed
    __syncthreads();                      // ... for every wavefront; and everybody is done reading this buffer
  }

  // ---- epilogue: D[pixel 4 kk + r][channel px] of each chain ----
  const int co = cg * NCO + px;
  const float bias = p.bias[co];
  float *outb = p.out + (size_t)b * H * W * p.out_stride + p.out_choff + co;
#pragma unroll
  for (int j = 0; j < MAXCH; ++j) {
    if (!have[j]) continue;
    const int c = wave + 4 * j;
    const int row = c / G, grp = c - row * G;
    const int y = y0 + row;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int x = 16 * grp + 4 * kk + r;
      float v = acc[j][r] + bias;
      v = v > 0.0f ? v : 0.0f;
      if (y < H && x < W) outb[((size_t)y * W + x) * p.out_stride] = v;
    }
  }
}

}  // namespace kc

size_t conv_f32_kc_weight_bytes(int cin, int cout) { return (size_t)cin * cout * 9 * 4; }

// W: [cout][cin][9] f32 (OIHW; several layers may be concatenated along cout by the caller) -> [cout / 16][cin / 16][tap][k 16][n 16]
void conv_f32_kc_pack_weights(const float *W, int cin, int cout, float *dst) {
  const int nchunk = cin / 16;
  for (int co = 0; co < cout; ++co)
    for (int ci = 0; ci < cin; ++ci)
      for (int t = 0; t < 9; ++t)
        dst[((((size_t)(co / 16) * nchunk + ci / 16) * 9 + t) * 16 + ci % 16) * 16 + co % 16] = W[((size_t)co * cin + ci) * 9 + t];
}

bool conv_f32_kc_supports(int H, int W, int cin, int cout) {
  const int G = (W + 15) / 16;
  return G <= kc::MAXG && cin % 16 == 0 && cout % 16 == 0 && H >= 1 && W >= 1;
}

// p.in / p.out / strides / B / H / W as for launch_conv_f32 (3x3, pad 1, bias, ReLU, no pool); wpack = conv_f32_kc_pack_weights,
// bias = plain [cout] channel order; cout = the channels this launch computes (a multiple of 16)
hipError_t launch_conv_f32_kc(const ConvParams &cp, int cin, int cout, const float *wpack, const float *bias, hipStream_t s) {
  if (!conv_f32_kc_supports(cp.H, cp.W, cin, cout)) return hipErrorInvalidValue;
  kc::Params p{};
  p.in = cp.in; p.wpack = wpack; p.bias = bias; p.out = cp.out;
  p.in_stride = cp.in_stride; p.out_stride = cp.out_stride; p.out_choff = cp.out_choff;
  p.B = cp.B; p.H = cp.H; p.W = cp.W; p.cin = cin; p.cout = cout;
  p.G = (cp.W + 15) / 16;
  p.R = 2 * p.G <= kc::MAXG && cp.H >= 2 ? 2 : 1;
  const int npix = (p.R + 2) * (16 * p.G + 2), nblk = (npix + 15) / 16;
  if (nblk > 4 * kc::A_PASSES) return hipErrorInvalidValue;
  if ((size_t)cp.H * cp.W * cp.in_stride * 4 >= ((size_t)1 << 31)) return hipErrorInvalidValue;   // 32-bit offsets, OOB marker
  const size_t lds = 2 * ((size_t)nblk * 1024 + kc::WCHUNK * 4);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  auto k = kc::conv_f32_kc_kernel;
  static bool attr_done[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) attr_done[dev] = true;
  }
  const int nstrip = (cp.H + p.R - 1) / p.R;
  hipLaunchKernelGGL(k, dim3((unsigned)(cp.B * nstrip * (cout / kc::NCO))), dim3(256), lds, s, p);
  return hipGetLastError();
}

}  // namespace spfe
