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
// of 16 input channels the halo strip ([16 ch][R + 2 rows][16 G + 2 px], channel-major: an A operand — 16 px x 4 channels —
// is 4 runs of 16 consecutive floats, plane stride = 16 mod 32 banks: conflict-free) and the chunk's 144 x 16 weights
// ([tap][k][16 ch]: a B operand is 64 consecutive floats) sit in LDS, double buffered; the next chunk travels global ->
// registers -> LDS under the current chunk's 108 MFMAs per wavefront; one barrier per chunk.  One B read feeds three MFMAs.
// K order: chunk -> tap -> channel (4 per MFMA, ascending): the contract's.  Accumulation starts from C = 0; out = max(acc
// + bias, 0).  Same bits as conv_f32.hip (tests/test_gpu_parity.py::test_f32_k_chain_kernel_on_mfma_16x16x4_is_bit_identical).
//
// STATUS: opt-in (SPFE_KC=<layer mask>), NOT the default — measured on a single 752x480 frame (rocprofv3, round 4):
//   conv4a / conv4b   conv_f32.hip 2-row tiles 22.8 us   this kernel 26.6 us      convPa 42 -> 40 us      conv3a 35 -> 62 us
// What the measurements say (each an ablation build run on the GPU box):
//   1. the matrix loop itself does what the arithmetic promised: with the staging removed the 2,880 chains of conv4a take 11 us
//      (the layer's roofline is 10.6) — once every operand read sits ALONE in an MFMA gap, two K steps ahead (four reads in
//      one gap: 16 extra cycles per MFMA; reads issued where they are used: 3x slower);
//   2. staging in the SAME wavefront doubles it (35 us) with or without memory traffic (all loads out of range: same time): a
//      buffer_load or a staging ds_write between MFMAs holds a lone wavefront's in-order stream for 100+ cycles each
//      (MI355X_MICROARCH.md prices an LDS-DMA piece at 60 - 185), 56 of them per 108 MFMAs;
//   3. so the staging moved to four producer wavefronts (this file) — and a 512-thread workgroup pays ~19 us before its FIRST
//      barrier completes on this stack (workgroups that return before the barrier: 5 us; one barrier or nine: the same 24 us;
//      1 KB or 70 KB of LDS: the same; the round-2 note on conv_bf16_ws.hip's "8 us more start-up" is the same effect), which
//      is more than the kernel saves on a layer that lasts 23 us.
// The next step would be LDS-direct staging in a 256-thread workgroup (no VGPR round trip, no ds_write: 13 instead of 56
// staging instructions per chunk); not built.
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
constexpr int A_PIECES = 10, W_PIECES = 3;           // float4 pieces per thread and chunk (halo: 4 x 98 px x 4 quads at R = 2, G = 6; 3 x 194 x 4 at R = 1, G = 12)

struct Params {
  const float *in;       // [B][H][W][in_stride], channels [0, cin)
  const float *wpack;    // [cout / 16][cin / 16][tap 9][k 16][n 16]
  const float *bias;     // [cout]
  float *out;            // [B][H][W][out_stride] at out_choff
  int in_stride, out_stride, out_choff;
  int B, H, W, cin, cout;
  int R, G;              // rows per strip, pixel groups of 16 per row
  int rowp, plane;       // LDS row pitch (16 G + 2) and channel-plane pitch (floats; = 16 mod 32)
};

__global__ __launch_bounds__(512, 2) void conv_f32_kc_kernel(Params p) {
  extern __shared__ __attribute__((aligned(16))) float smem_kc[];
  // 8 wavefronts, two per SIMD: 0..3 CONSUMERS (operand reads + MFMAs, nothing else), 4..7 PRODUCERS (the next chunk:
  // global loads -> registers -> LDS).  In ONE instruction stream the 13 loads and 43 staging stores of a chunk cost the lone
  // wavefront of a SIMD as much issue time as its 108 MFMAs (measured: 35 us against 16 without them, memory traffic or not —
  // a VMEM / LDS-store instruction between MFMAs holds the in-order stream for 100+ cycles); in a second wavefront they run
  // beside the matrix stream.
  const int tid = threadIdx.x & 255, lane = tid & 63;
  const bool producer = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) != 0;
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
  const int ABUF = KC * p.plane;                 // floats of a halo buffer
  const int BUF = ABUF + WCHUNK + 4;             // (+ a spare float4: the dummy destination of unused weight pieces)

  if (producer) {
    const unsigned pixb = (unsigned)p.in_stride * 4u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.in + (size_t)b * H * W * p.in_stride), 0, (unsigned)((size_t)H * W * pixb), 0x00020000);
    const float *wbase = p.wpack + (size_t)cg * nchunk * WCHUNK;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(wbase), 0, (unsigned)((size_t)nchunk * WCHUNK * 4), 0x00020000);
    // staging geometry: halo pieces (a float4 = 4 channels of one halo pixel) and weight pieces of a chunk.  16 consecutive
    // lanes = 16 consecutive halo pixels of one channel quad, the four quads in the wavefront's four lane groups: a load
    // instruction touches 16 x 64 contiguous bytes, a staging store 16 consecutive banks per plane
    const int hrows = R + 2, hcols = 16 * G + 2;
    const int npix = hrows * hcols;
    unsigned aoff[A_PIECES];   // byte offset of the piece's pixel in the frame (+ quad), or OOB
    int adst[A_PIECES];        // float index in the halo buffer (quad's first plane); unused piece: the planes' last pad floats
#pragma unroll
    for (int it = 0; it < A_PIECES; ++it) {
      const int i = tid + it * 256;
      const int q = (i >> 4) & 3, pix = (i >> 6) * 16 + (i & 15);
      const int hr = pix / hcols, hcx = pix - hr * hcols;
      const int gy = y0 + hr - 1, gx = hcx - 1;
      const bool used = pix < npix;
      const bool in = used && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
      aoff[it] = in ? (unsigned)(gy * W + gx) * pixb + (unsigned)q * 16u : OOB;
      adst[it] = used ? (4 * q) * p.plane + hr * p.rowp + hcx : p.plane - 1;
    }
    // two register sets: chunk c + 1 is in flight (global -> registers) while chunk c goes registers -> LDS, so a store waits
    // for loads issued a whole chunk earlier (with one set the producers' round trip — 2 us under load — was the chunk time)
    f32x4 va[2][A_PIECES], vw[2][W_PIECES];
    auto issue = [&](int chunk, f32x4 (&a)[A_PIECES], f32x4 (&w)[W_PIECES]) {
      const bool ok = chunk < nchunk;   // (past the end: out-of-range loads, no branch — the compiler keeps count of what is in flight)
#pragma unroll
      for (int it = 0; it < A_PIECES; ++it) {
        const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rin, aoff[it] == OOB || !ok ? OOB : aoff[it] + (unsigned)chunk * (KC * 4), 0, 0);
        a[it] = __builtin_bit_cast(f32x4, v);
      }
#pragma unroll
      for (int it = 0; it < W_PIECES; ++it) {
        const int i = tid + it * 256;
        const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rw, i < WCHUNK / 4 && ok ? (unsigned)(chunk * WCHUNK + 4 * i) * 4u : OOB, 0, 0);
        w[it] = __builtin_bit_cast(f32x4, v);
      }
    };
    auto stage = [&](int chunk, const f32x4 (&a)[A_PIECES], const f32x4 (&w)[W_PIECES]) {
      float *buf = smem_kc + (chunk & 1) * BUF;   // (last read by the consumers during chunk - 2: they are past barrier chunk - 1)
#pragma unroll
      for (int it = 0; it < A_PIECES; ++it) {
        float *d = buf + adst[it];
        d[0] = a[it].x; d[p.plane] = a[it].y; d[2 * p.plane] = a[it].z; d[3 * p.plane] = a[it].w;
      }
#pragma unroll
      for (int it = 0; it < W_PIECES; ++it) {
        const int i = tid + it * 256;
        reinterpret_cast<f32x4 *>(buf + ABUF)[i < WCHUNK / 4 ? i : WCHUNK / 4] = w[it];
      }
    };
    issue(0, va[0], vw[0]);
    for (int chunk = 0; chunk < nchunk; chunk += 2) {
      issue(chunk + 1, va[1], vw[1]);
      stage(chunk, va[0], vw[0]);
      __syncthreads();   // barrier `chunk`: chunk is in LDS
      if (chunk + 1 < nchunk) {   // (uniform)
        issue(chunk + 2, va[0], vw[0]);
        stage(chunk + 1, va[1], vw[1]);
        __syncthreads();
      }
    }
    __syncthreads();     // (the consumers' last barrier)
    return;
  }

  // ---- consumers.  This wavefront's chains: chain c = wave + 4 j -> (row c / G, pixel group c % G) ----
  const int nchain = R * G;
  int abase[MAXCH];          // float index of (row, 16 grp + px) of the chain's top-left tap in plane kk
  bool have[MAXCH];
#pragma unroll
  for (int j = 0; j < MAXCH; ++j) {
    const int c = wave + 4 * j;
    have[j] = c < nchain;                        // (wave-uniform)
    const int cc = have[j] ? c : 0;
    const int row = cc / G, grp = cc - row * G;
    abase[j] = kk * p.plane + row * p.rowp + 16 * grp + px;
  }
  const int wlane = kk * NCO + px;               // B operand: [tap][k][n] -> (tap * 16 + 4 k4 + kk) * 16 + n
  f32x4 acc[MAXCH];
#pragma unroll
  for (int j = 0; j < MAXCH; ++j) acc[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};   // the contract's chain starts from +0
  __syncthreads();           // barrier 0: chunk 0 is in LDS
#pragma unroll 1
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    const float *bufA = smem_kc + (chunk & 1) * BUF;
    const float *bufW = bufA + ABUF;
    // 36 K steps (tap, channel quad); the operands of step s + 2 — one B and three A values — are read while the MFMAs of
    // step s issue, one read per MFMA gap (pinned: left alone, the scheduler sinks every read to its first use and each MFMA
    // waits out an LDS round trip; four reads in ONE gap overran the 32-cycle shadow)
    float av[3][MAXCH], bw[3];
    auto rd = [&](int st, int slice) {
      const int slot = st % 3;
      const int tap = st >> 2, k4 = st & 3;
      const int dy = tap / 3, dx = tap - 3 * dy;
      const int ao = (4 * k4) * p.plane + dy * p.rowp + dx;
      if (slice == 0) bw[slot] = bufW[(tap * KC + 4 * k4) * NCO + wlane];
      av[slot][slice] = bufA[abase[slice] + ao];
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
        __builtin_amdgcn_sched_barrier(0);
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[st % 3][j], bw[st % 3], acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();         // barrier chunk + 1: chunk + 1 is in LDS, and this buffer may be refilled (by chunk + 2)
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
  p.rowp = 16 * p.G + 2;
  const int raw = (p.R + 2) * p.rowp;
  p.plane = raw + ((16 - raw % 32) + 32) % 32;   // = 16 mod 32: the four channel planes of an A operand on distinct bank halves
  if (p.plane == raw) p.plane += 32;             // (at least one pad float per plane: the dummy destination of unused staging pieces)
  if ((((p.R + 2) * p.rowp + 15) / 16) * 64 > kc::A_PIECES * 256) return hipErrorInvalidValue;
  if ((size_t)cp.H * cp.W * cp.in_stride * 4 >= ((size_t)1 << 31)) return hipErrorInvalidValue;   // 32-bit offsets, OOB marker
  const size_t lds = 2 * ((size_t)kc::KC * p.plane + kc::WCHUNK + 4) * sizeof(float);
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
  hipLaunchKernelGGL(k, dim3((unsigned)(cp.B * nstrip * (cout / kc::NCO))), dim3(512), lds, s, p);
  return hipGetLastError();
}

}  // namespace spfe
