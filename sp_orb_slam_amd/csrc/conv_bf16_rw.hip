// conv_bf16_rw.hip — bf16 3x3 convolution for the Cin = 128 layers of the bf16 mode (conv3b, conv4a, conv4b, convPa|Da of
// SPFrontend::forward, /root/reference/orb_slam2/src/cv/sp_extractor.cpp:88-100; BASELINE.json configs[3]) with the
// WEIGHTS RESIDENT IN REGISTERS.
//
// Why.  A 64-output-channel block of a Cin = 128 layer has 147 KB of weights: they do not fit in LDS beside two halo
// buffers, so conv_bf16.hip streams a 36.9 KB weight chunk with every 32-channel stage of every tile and those layers run
// at the chip's LDS-DMA fill rate (0.34-0.44 of the bf16 MFMA peak; ~19 B of L2 -> LDS traffic per MFMA cycle and CU).
// A CU's register file is three times its LDS (4 SIMDs x 128 KB): with ONE wavefront per SIMD a wavefront may hold 512
// registers, and the 73,728 bytes of weights of 32 output channels (K = 9 x 128) are 288 of them.  So here
//   * a workgroup is 4 wavefronts, one per SIMD; wavefront w owns output channels 32 w .. 32 w + 31 of the workgroup's
//     128-channel group for the whole kernel — its B operands (72 K steps x 4 registers) are loaded once and never move;
//   * all four wavefronts work on the SAME pixel tile (MT rows x 32 pixels), so the only thing LDS holds is the halo tile
//     with all 128 input channels (272-byte pixel pitch: 256 + 16 of padding, which makes every fragment address one
//     register + an immediate and the 16 lanes of a ds_read_b128 group hit 16 distinct bank slots);
//   * per tile and CU 55 KB come in from L2 for 4 x 288 MFMAs (5.9 B per MFMA cycle, a third of before) and nothing
//     is re-fetched per output-channel block: the halo is read once per 128 output channels;
//   * the three vertical taps of a (dx, 16-channel group) share their MT + 2 halo-row fragments: 0.5 LDS fragment reads
//     per MFMA (conv_bf16_ws.hip: 0.83; the LDS operand traffic is what costs that kernel its clock).
// Everything else is the persistent, everything-in-the-MFMA-shadow pipeline of the other two bf16 kernels: dynamic tile
// queue per (XCD, channel group), LDS-direct halo loads of the NEXT tile into the other buffer, the PREVIOUS tile's
// epilogue out of a second accumulator set, one barrier per tile placed one K group before the tile's end so that the next
// tile's first fragments are already in registers when its first MFMA issues.
//
// Arithmetic: v_mfma_f32_32x32x16_bf16 as mfma(pixels, weights), K order 32-channel chunk -> dx -> 16-channel group -> dy,
// k slot 8 hi + j <-> channel 32 chunk + 16 group + 8 hi + j, f32 accumulate, 2x2 max before the bias, bias, ReLU, RNE to
// bf16 — exactly conv_bf16.hip's sequence, so the two kernels are bit-identical and which one runs is a launch-size
// decision (tests/test_gpu_bf16.py::test_bf16_rw_kernel_equals_streamed_weight_kernel).
#include <utility>

#include "spfe_kernels.h"

namespace spfe {
namespace rw {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(3))) bf16x8 lds_frag;

constexpr unsigned OOB = 0x80000000u;
constexpr int COLS = 34;
constexpr int PITCH = 272;                       // bytes per halo pixel in LDS: 128 channels + one 16-byte pad piece
constexpr int PPP = PITCH / 16;                  // 17 pieces per pixel
constexpr int ROW_BYTES = COLS * PITCH;          // 9248
constexpr int NSTEP = 72;                        // K steps per tile: 4 chunks x 3 dx x 2 groups x 3 dy
constexpr int NGROUP = NSTEP / 3;                // 24 (chunk, dx, group) triples: the three vertical taps share fragments
constexpr int NW_AGPR = 64;                      // weight fragments kept in AGPRs (4 registers each): all 256 of them
constexpr int W_WAVE_BYTES = NSTEP * 64 * 16;    // 73,728: this wavefront's B operands, [step][lane][8 bf16]

template <int MT>
struct Geo {
  static constexpr int ROWS = MT + 2;
  static constexpr int PIECES = ROWS * COLS * PPP;
  static constexpr int INSTR = (PIECES + 63) / 64;        // wave-level LDS-direct passes per halo tile
  static constexpr int IT = (INSTR + 3) / 4;              // ... per wavefront
  static constexpr int BUF_BYTES = IT * 4 * 1024;         // whole rounds of four passes: the surplus pieces land in the tail (zeros)
  static constexpr int LDS_GEO = 2 * BUF_BYTES;           // per pass and lane: {source offset for tile (0, 0), halo (row << 8 | column)}
  static constexpr int LDS_SLOT = LDS_GEO + IT * 256 * 8; // the queue slot (one int) + padding
  static constexpr int LDS_TOTAL = LDS_SLOT + 64;
};

__device__ __forceinline__ void wg_barrier() { asm volatile("s_barrier" ::: "memory"); }

#ifdef RW_PROBE
// probe builds (tools/microbench/conv_rw_probe.hip): cycle counters of wavefront 0, summed over workgroups
// [0] loop cycles  [1] ... of which from the end-of-tile wait to the barrier's release  [2] tiles  [3] 100 MHz ticks  [4] prologue cycles
__device__ unsigned long long rw_dbg[8];
#define RW_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define RW_ACC(i, v) do { if (wave == 0 && lane == 0) atomicAdd(&rw_dbg[i], (unsigned long long)(v)); } while (0)
#else
#define RW_T(var)
#define RW_ACC(i, v)
#endif
#ifndef RW_ABLATE
#define RW_ABLATE 0   // probe builds: 1 = no halo passes in the loop, 2 = no MFMAs, 3 = no epilogue, 4 = no fragment reads
#endif

__device__ __forceinline__ float max_nc(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float max3_nc(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float relu_nc(float a) {
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(a));
  return r;
}
__device__ __forceinline__ unsigned pack2(float v0, float v1) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){v0, v1}, bf16x2));
}
// q / d for q < 2^20 with rcp = 1.0f / d (conv_bf16.hip: exact)
__device__ __forceinline__ int udiv_small(int q, float rcp) { return (int)(((float)q + 0.5f) * rcp); }

__device__ __forceinline__ int tile_x0(int tx, int W) {
  const int x0 = tx * 32;
  return x0 + 32 > W ? W - 32 : x0;   // the last tile of a ragged row ends at the image edge (recomputes a few columns)
}

struct KStep {
  int chunk, dx, k2, dy, tap, piece;
};
__host__ __device__ constexpr KStep kstep_of(int s) {
  const int chunk = s / 18, r = s % 18, dx = r / 6, k2 = (r % 6) / 3, dy = r % 3;
  return KStep{chunk, dx, k2, dy, dy * 3 + dx, chunk * 4 + k2 * 2};
}

// The epilogue of one tile.  After mfma(pixels, weights) a lane holds ONE output channel (32 w + l31) of 16 pixels per
// accumulator: register r <-> pixel column 8 (r >> 2) + 4 hi + (r & 3) of row i.  bf16 outputs are 2 bytes, so two
// neighbouring lanes trade halves: v_cvt_pk packs this lane's channel of pixels (m, m + 1), one DPP quad swap fetches the
// neighbour's pair, v_perm keeps [own.lo, nb.lo] on even lanes — channels (l31, l31 + 1) of pixel m — and [nb.hi, own.hi]
// on odd lanes — channels (l31 - 1, l31) of pixel m + 1; the dword store then writes, per half wavefront, two whole
// 64-byte runs (32 channels of two pixels).  pool: the same with the two pooled pixels of a register quad.
struct Epi {
  __amdgpu_buffer_rsrc_t rout;
  unsigned off0;        // byte offset of this lane's dword at (first output row of the tile [pool: pooled row], register column 0)
  unsigned pitch;       // bytes per output pixel
  unsigned rowpitch;    // bytes per output row; rows past the frame fall behind the buffer's num_records and are dropped
};

template <int MT, bool POOL>
struct EpiN {
  static constexpr int N = POOL ? (MT / 2) * 4 : MT * 8;   // items (one dword store each)
};

template <int MT, bool POOL, int E>
__device__ __forceinline__ void epi_item(const Epi &e, float bias, unsigned sel, const f32x16 (&acc)[MT]) {
  if constexpr (E >= 0 && E < EpiN<MT, POOL>::N) {
    float v0, v1;
    unsigned row, col;
    if constexpr (POOL) {
      constexpr int ip = E / 4, g = E % 4, i0 = 2 * ip, r = 4 * g;
      v0 = max3_nc(acc[i0][r], acc[i0][r + 1], max_nc(acc[i0 + 1][r], acc[i0 + 1][r + 1]));
      v1 = max3_nc(acc[i0][r + 2], acc[i0][r + 3], max_nc(acc[i0 + 1][r + 2], acc[i0 + 1][r + 3]));
      row = ip;
      col = 4 * g;         // pooled columns 4 g + 2 hi + {0, 1}
    } else {
      constexpr int i = E / 8, q = E % 8, r = 2 * q;
      v0 = acc[i][r];
      v1 = acc[i][r + 1];
      row = i;
      col = 8 * (r >> 2) + (r & 3);
    }
    v0 = relu_nc(v0 + bias);
    v1 = relu_nc(v1 + bias);
    const unsigned pk = pack2(v0, v1);
    const unsigned nb = (unsigned)__builtin_amdgcn_mov_dpp((int)pk, 0xB1, 0xF, 0xF, true);   // quad_perm [1, 0, 3, 2]
    const unsigned o = __builtin_amdgcn_perm(nb, pk, sel);
    __builtin_amdgcn_raw_buffer_store_b32(o, e.rout, e.off0 + row * e.rowpitch, col * e.pitch, 0);
  }
}

// The same, cut into micro-steps of <= 6 instructions for the MFMA gaps (a wavefront alone on its SIMD hides ~5 issue
// slots per 32-cycle MFMA; a whole item in one gap stalls the matrix pipe for ~90 cycles).  no pool: item E = U / 2 in two
// steps; pool: item E = U / 3 in three.
struct EpiTmp {
  float v0, v1, t0, t1;
  unsigned pk, nb;
};
template <int MT, bool POOL>
struct EpiU {
  static constexpr int PER = 4;
  static constexpr int N = EpiN<MT, POOL>::N * PER;
};
template <int MT, bool POOL, int U>
__device__ __forceinline__ void epi_micro(const Epi &e, float bias, unsigned sel, const f32x16 (&acc)[MT], EpiTmp &t) {
  if constexpr (U >= 0 && U < EpiU<MT, POOL>::N) {
    constexpr int E = U / 4, ph = U % 4;
    constexpr int row = POOL ? E / 4 : E / 8;
    constexpr int col = POOL ? 4 * (E % 4) : 8 * ((2 * (E % 8)) >> 2) + ((2 * (E % 8)) & 3);
    if constexpr (ph == 0) {
      if constexpr (POOL) {
        constexpr int i0 = 2 * (E / 4), r = 4 * (E % 4);
        t.t0 = max_nc(acc[i0 + 1][r], acc[i0 + 1][r + 1]);
        t.t1 = max_nc(acc[i0 + 1][r + 2], acc[i0 + 1][r + 3]);
        t.v0 = max3_nc(acc[i0][r], acc[i0][r + 1], t.t0);
        t.v1 = max3_nc(acc[i0][r + 2], acc[i0][r + 3], t.t1);
      } else {
        constexpr int i = E / 8, r = 2 * (E % 8);
        t.v0 = acc[i][r];
        t.v1 = acc[i][r + 1];
      }
    } else if constexpr (ph == 1) {
      t.v0 = relu_nc(t.v0 + bias);
      t.v1 = relu_nc(t.v1 + bias);
    } else if constexpr (ph == 2) {
      t.pk = pack2(t.v0, t.v1);
    } else {
      t.nb = (unsigned)__builtin_amdgcn_mov_dpp((int)t.pk, 0xB1, 0xF, 0xF, true);
      const unsigned o = __builtin_amdgcn_perm(t.nb, t.pk, sel);
      __builtin_amdgcn_raw_buffer_store_b32(o, e.rout, e.off0 + (unsigned)row * e.rowpitch, (unsigned)col * e.pitch, 0);
    }
  }
}

struct TileDesc {
  int b, ty, tx, valid;
};

// in: NHWC bf16 [B][H][W][in_stride]; wpack: [ncg][wave 4][step 72][lane 64][8 bf16] (pack_layer_bf16_rw);
// bias: f32 [ncg * 128]; out: NHWC bf16; p.nblk = ncg (128-channel groups); p.tile_ctr: ncg * 8 counters, zero on entry.
template <int MT, bool POOL>
__global__ __launch_bounds__(256, 1) void conv_bf16_rw_kernel(ConvParams p) {
  using G = Geo<MT>;
  extern __shared__ __attribute__((aligned(16))) char smem_rw[];
  lds_char *const lds = (lds_char *)smem_rw;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int xcd = blockIdx.x & 7;
  const int ncg = p.nblk;
  const int cg = (int)(blockIdx.x >> 3) % ncg;
  const int gi = (int)(blockIdx.x >> 3) / ncg, gsize = (int)(gridDim.x >> 3) / ncg;
  const int per_cg = p.tiles_x * p.tiles_y * p.B;
  const int t_lo = (int)((long)per_cg * xcd / 8), t_cnt = (int)((long)per_cg * (xcd + 1) / 8) - t_lo;
  if (gi >= t_cnt) return;   // (whole workgroup: nothing to do)
  RW_T(kernel0);
  const int H = p.H, W = p.W;
  const int Ho = POOL ? H >> 1 : H, Wo = POOL ? W >> 1 : W;
  const unsigned in_pix_bytes = (unsigned)p.in_stride * 2u;
  const unsigned frame_in_bytes = (unsigned)H * W * in_pix_bytes;
  const unsigned out_pix_bytes = (unsigned)p.out_stride * 2u;
  const unsigned frame_out_bytes = (unsigned)Ho * Wo * out_pix_bytes;
  const float rcp_tx = 1.0f / (float)p.tiles_x, rcp_ty = 1.0f / (float)p.tiles_y;
  int *const ctr = p.tile_ctr + cg * 8 + xcd;

  auto decode = [&](int idx) -> TileDesc {   // idx: index inside this XCD's range, or >= t_cnt
    TileDesc d;
    d.valid = idx < t_cnt;
    int q = t_lo + (d.valid ? idx : 0);
    int dv = __builtin_amdgcn_readfirstlane(udiv_small(q, rcp_tx));
    d.tx = q - dv * p.tiles_x;
    q = dv;
    dv = __builtin_amdgcn_readfirstlane(udiv_small(q, rcp_ty));
    d.ty = q - dv * p.tiles_y;
    d.b = dv;
    return d;
  };

  const float bias = p.bias[cg * 128 + wave * 32 + l31];
  const unsigned sel = (l31 & 1) ? 0x03020706u : 0x05040100u;

  // ---- halo staging: pass `it` of this wavefront covers LDS pieces q = (it * 4 + wave) * 64 + lane; piece q belongs to halo
  // pixel q / 17 (row pix / 34, column pix % 34), 16-byte piece q % 17 (16 = the pad).  What a pass needs from that is fixed
  // for the kernel: it sits in an LDS table (registers are what this kernel has none of) — per pass and lane the source
  // offset the piece has in tile (0, 0) and its halo (row, column) for the border test; pad / surplus pieces carry an
  // offset that stays out of range whatever is added.
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  auto geo_of = [&](int it) -> u32x2 {
    const unsigned q = (unsigned)((it * 4 + wave) * 64 + lane);
    const unsigned pix = (q * 3856u) >> 16, piece = q - pix * 17u;   // q / 17, q % 17 (q < 4096)
    const unsigned r = (pix * 1928u) >> 16, c = pix - r * 34u;       // pix / 34, pix % 34 (pix < 260)
    const bool real = q < (unsigned)G::PIECES && piece < 16u;
    u32x2 g;
    g.x = real ? (unsigned)(((int)r - 1) * W + ((int)c - 1)) * in_pix_bytes + piece * 16u : OOB;
    g.y = real ? (r << 8) | c : 0xffff00u;
    return g;
  };
  lds_char *const geo_base = lds + G::LDS_GEO + tid * 8;
  struct TileGeo {   // scalars of the tile whose halo is being fetched
    __amdgpu_buffer_rsrc_t rin;
    unsigned base;   // byte offset of the tile's pixel (0, 0) in its frame
    int y0m1, x0m1;  // halo origin
    bool interior;   // no halo pixel outside the frame: no per-piece test
  };
  auto aim_dma = [&](const TileDesc &d) -> TileGeo {
    TileGeo t;
    const char *base = reinterpret_cast<const char *>(p.in) + ((size_t)d.b * H * W * p.in_stride + p.in_choff) * 2;
    t.rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, d.valid ? frame_in_bytes : 0u, 0x00020000);
    const int y0 = d.ty * MT, x0 = tile_x0(d.tx, W);
    t.base = (unsigned)(y0 * W + x0) * in_pix_bytes;
    t.y0m1 = y0 - 1;
    t.x0m1 = x0 - 1;
    t.interior = y0 >= 1 && y0 + MT + 1 <= H && x0 >= 1 && x0 + 33 <= W;
    return t;
  };
  u32x2 dgeo[2] = {{OOB, 0xffff00u}, {OOB, 0xffff00u}};
  unsigned dvoff = OOB;
  (void)dvoff;   // (the host pass of hipcc does not see the uses below)
  // phase 0: the table entry of the NEXT pass (a pass reads its own three gaps ahead: no wait on the LDS queue);
  // 1: where this lane's 16 bytes come from; 2: the pass itself.  dma_first: the entry of pass 0.
  auto dma_first = [&]() { dgeo[0] = *reinterpret_cast<const __attribute__((address_space(3))) u32x2 *>(geo_base); };
  auto dma = [&](const TileGeo &t, int buf, int it, int phase) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (phase == 0) {
      if (it + 1 < G::IT)
        dgeo[(it + 1) & 1] = *reinterpret_cast<const __attribute__((address_space(3))) u32x2 *>(geo_base + (it + 1) * 2048);
    } else if (phase == 1) {
      const u32x2 g = dgeo[it & 1];
      const unsigned off = g.x + t.base;
      if (t.interior) {
        dvoff = off;
      } else {
        const int gy = t.y0m1 + (int)(g.y >> 8), gx = t.x0m1 + (int)(g.y & 0xffu);
        dvoff = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? off : OOB;
      }
    } else {
      int wave_now = wave;
      asm volatile("" : "+s"(wave_now));   // (opaque: keeps 2 x IT LDS addresses from being hoisted into SGPRs)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(t.rin, (lds_void *)(lds + buf * G::BUF_BYTES + (it * 4 + wave_now) * 1024), 16, dvoff,
                                               0, 0, 0);
    }
#endif
  };

  // ---- operand fragment addresses: one register per buffer + immediates
  lds_char *const abase[2] = {lds + l31 * PITCH + hi * 16, lds + G::BUF_BYTES + l31 * PITCH + hi * 16};
  auto frag = [&](int buf, int row, int dx, int piece) -> bf16x8 {
    return *reinterpret_cast<lds_frag *>(abase[buf] + row * ROW_BYTES + dx * PITCH + piece * 16);
  };

  auto aim_epi = [&](const TileDesc &d, Epi &e) {
    char *obase = reinterpret_cast<char *>(p.out) + ((size_t)d.b * Ho * Wo * p.out_stride + p.out_choff) * 2;
    e.rout = __builtin_amdgcn_make_buffer_rsrc(obase, 0, d.valid ? frame_out_bytes : 0u, 0x00020000);
    e.pitch = out_pix_bytes;
    const int y0 = d.ty * MT, x0 = tile_x0(d.tx, W);
    // this lane's dword: even lanes channels (l31, l31 + 1) of the first pixel of a pair, odd lanes (l31 - 1, l31) of the second
    const unsigned lane_part = (unsigned)(cg * 128 + wave * 32 + (l31 & ~1)) * 2u + (unsigned)(l31 & 1) * out_pix_bytes;
    // (rows at or past H: their offsets are >= frame_out_bytes, the buffer's range check drops the stores)
    if constexpr (POOL) e.off0 = (unsigned)((y0 >> 1) * Wo + (x0 >> 1) + 2 * hi) * out_pix_bytes + lane_part;
    else e.off0 = (unsigned)(y0 * W + x0 + 4 * hi) * out_pix_bytes + lane_part;
  };

  // ---- prologue: tiles gi and gi + gsize of this queue are pre-assigned; the queue hands out the ones after those
  TileDesc cur = decode(gi), nxt = decode(gi + gsize);
  {
    TileGeo tg0 = aim_dma(cur);
    tg0.interior = false;
#pragma unroll
    for (int it = 0; it < G::IT; ++it) {
      dgeo[it & 1] = geo_of(it);
      *reinterpret_cast<__attribute__((address_space(3))) u32x2 *>(geo_base + it * 2048) = dgeo[it & 1];
      dma(tg0, 0, it, 1);
      dma(tg0, 0, it, 2);
    }
  }
  TileGeo tgeo = aim_dma(nxt);
  // ---- this wavefront's weights: 72 fragments of 8 bf16 per lane, in registers for the whole kernel
  bf16x8 wreg[NSTEP];
  {
    const char *wb = reinterpret_cast<const char *>(p.wpack) + ((size_t)(cg * 4 + wave) * NSTEP * 64 + lane) * 16;
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) wreg[s] = *reinterpret_cast<const bf16x8 *>(wb + (size_t)s * 1024);
    // Register classes.  A wavefront alone on its SIMD has 256 VGPRs + 256 AGPRs; VALU instructions only see the former,
    // MFMA operands may come from either.  Left alone, the allocator fills the VGPRs first and uses AGPRs as spill space
    // (v_accvgpr_read before every use).  Pinned: NW_AGPR fragments live in AGPRs and are read from there by the MFMAs, the
    // rest in VGPRs beside the accumulators (which the epilogue's VALU code reads) and the pixel fragments.
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      if (s < NW_AGPR) asm volatile("" : "+a"(wreg[s]));
      else asm volatile("" : "+v"(wreg[s]));
    }
  }

  f32x16 accA[MT], accB[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { accA[i][r] = 0.0f; accB[i][r] = 0.0f; }
  Epi epiA, epiB;
  epiA.rout = epiB.rout = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, 0u, 0x00020000);   // nothing to store yet
  epiA.off0 = epiB.off0 = OOB;
  epiA.pitch = epiB.pitch = out_pix_bytes;
  epiA.rowpitch = epiB.rowpitch = (unsigned)Wo * out_pix_bytes;
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): weights in registers, tile 0 in LDS
  wg_barrier();
  bf16x8 a[2][MT + 2];
#pragma unroll
  for (int r = 0; r < MT + 2; ++r) a[0][r] = frag(0, r, 0, 0);   // group 0 of the first tile
  int fetched = 0;   // (lane 0 of wavefront 0: what its atomic returned)

  // One tile: 24 K groups x 3 vertical taps x MT MFMAs = NGAP gaps, each with <= ~5 instructions of side work:
  //   every group, gaps 0 .. MT + 1   the MT + 2 halo-row fragments of the NEXT group (the next tile's group 0 at the end)
  //   gap 2                          wavefront 0 asks the tile queue for the tile after next
  //   gaps E0 .. E0 + NU - 1         the PREVIOUS tile's epilogue, one micro-step each (early: the end-of-tile wait covers stores)
  //   gaps D0 .. D0 + 3 IT - 1       the NEXT tile's halo -> the other buffer, three micro-steps per pass
  //   group 21                       this tile's epilogue geometry (used one tile later)
  //   group 23, gap 0                the barrier — one group early: this wavefront holds every fragment of the tile (group
  //                                  23's were read during group 22), the next halo and the queue slot are written
  //   group 23                       the queue slot -> the descriptor of the tile after next
  EpiTmp et = {0.0f, 0.0f, 0.0f, 0.0f, 0u, 0u};
#ifdef RW_PROBE
  unsigned long long bar_cycles = 0, ntiles = 0;
#endif
  TileDesc nxt2 = nxt;
  auto run_tile = [&]<int BUF>(std::integral_constant<int, BUF>, f32x16(&acc)[MT], const f32x16(&accPrev)[MT], Epi &eMine,
                               const Epi &ePrev) {
    constexpr int NQ = 3 * MT, NGAP = NGROUP * NQ;
    constexpr int E0 = NQ, NU = EpiU<MT, POOL>::N, D0 = E0 + NU;
    static_assert(D0 + 3 * G::IT <= (NGROUP - 3) * NQ, "side work does not fit the tile");
    int slot_val = 0;
    [&]<int... QI>(std::integer_sequence<int, QI...>) {
      (
          [&] {
            constexpr int Q = QI, g = Q / NQ, q = Q % NQ, gp = g % 2, dy = q / MT, i = q % MT;
            if constexpr (g == NGROUP - 1 && q == 0) {
              if (wave == 0 && lane == 0) *reinterpret_cast<__attribute__((address_space(3))) int *>(lds + G::LDS_SLOT) = fetched;
              RW_T(b0);
              __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0)
              wg_barrier();
              RW_T(b1);
#ifdef RW_PROBE
              bar_cycles += b1 - b0;
#endif
            }
            if constexpr (q < MT + 2 && RW_ABLATE != 4) {
              if constexpr (g + 1 < NGROUP) {
                constexpr KStep kn = kstep_of(3 * (g + 1));
                a[gp ^ 1][q] = frag(BUF, q, kn.dx, kn.piece);
              } else {
                a[gp ^ 1][q] = frag(BUF ^ 1, q, 0, 0);
              }
            }
            if constexpr (Q == 2) {
              if (wave == 0 && lane == 0) fetched = atomicAdd(ctr, 1);
            }
            if constexpr (Q >= E0 && Q < E0 + NU && RW_ABLATE != 3) epi_micro<MT, POOL, Q - E0>(ePrev, bias, sel, accPrev, et);
            if constexpr (Q == D0 - 2) dma_first();
            if constexpr (Q >= D0 && Q < D0 + 3 * G::IT && RW_ABLATE != 1) dma(tgeo, BUF ^ 1, (Q - D0) / 3, (Q - D0) % 3);
            if constexpr (g == NGROUP - 3 && q == MT + 2) aim_epi(cur, eMine);
            if constexpr (g == NGROUP - 1 && q == MT + 2)
              slot_val = *reinterpret_cast<const __attribute__((address_space(3))) int *>(lds + G::LDS_SLOT);
            if constexpr (g == NGROUP - 1 && q == NQ - 2) nxt2 = decode(__builtin_amdgcn_readfirstlane(slot_val) + 2 * gsize);
            if constexpr (g == NGROUP - 1 && q == NQ - 1) tgeo = aim_dma(nxt2);   // (used by the next tile's passes)
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (RW_ABLATE == 2) {
            } else if constexpr (Q < MT) {   // the tile's first MFMA of each accumulator takes C = 0
              f32x16 z;
#pragma unroll
              for (int r = 0; r < 16; ++r) z[r] = 0.0f;
              acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[gp][i + dy], wreg[3 * g + dy], z, 0, 0, 0);
            } else {
              acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[gp][i + dy], wreg[3 * g + dy], acc[i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
          }(),
          ...);
    }(std::make_integer_sequence<int, NGAP>{});
    cur = nxt;
    nxt = nxt2;
#ifdef RW_PROBE
    ++ntiles;
#endif
  };

  bool lastA = true;
#ifdef RW_PROBE
  RW_T(loop0);
  const unsigned long long wall0 = wall_clock64();
  RW_ACC(4, loop0 - kernel0);
#endif
  while (true) {
    run_tile(std::integral_constant<int, 0>{}, accA, accB, epiA, epiB);
    lastA = true;
    if (!cur.valid) break;
    run_tile(std::integral_constant<int, 1>{}, accB, accA, epiB, epiA);
    lastA = false;
    if (!cur.valid) break;
  }
#ifdef RW_PROBE
  {
    RW_T(loop1);
    RW_ACC(0, loop1 - loop0); RW_ACC(1, bar_cycles); RW_ACC(2, ntiles); RW_ACC(3, wall_clock64() - wall0);
  }
#endif
  {
    constexpr int NEPI = EpiN<MT, POOL>::N;
    auto flush = [&](const f32x16(&acc)[MT], const Epi &e) {
      [&]<int... E>(std::integer_sequence<int, E...>) {
        (epi_item<MT, POOL, E>(e, bias, sel, acc), ...);
      }(std::make_integer_sequence<int, NEPI>{});
    };
    if (lastA) flush(accA, epiA); else flush(accB, epiB);
  }
}

template <int MT, bool POOL>
static hipError_t launch(const ConvParams &p, hipStream_t s) {
  using G = Geo<MT>;
  static_assert(G::LDS_TOTAL <= 160 * 1024, "two halo buffers must fit the 160 KB LDS");
  auto k = conv_bf16_rw_kernel<MT, POOL>;
  static bool attr_done[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_TOTAL);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) attr_done[dev] = true;
  }
  int grid = p.num_cus > 0 ? p.num_cus : 256;
  grid -= grid % (8 * p.nblk);   // a multiple of 8 XCDs x the channel groups
  if (grid < 8 * p.nblk) grid = 8 * p.nblk;
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), G::LDS_TOTAL, s, p);
  return hipGetLastError();
}

}  // namespace rw

size_t conv_bf16_rw_weight_bytes() { return 4 * (size_t)rw::W_WAVE_BYTES; }   // per 128-channel group

// cin = 128; p.nblk = 128-channel output groups (1 .. 4); p.tiles_y = ceil(H / tile_rows); tile_rows 4, 2, or 3 (layers without a pool)
hipError_t launch_conv_bf16_rw(const ConvParams &p, bool pool, int tile_rows, hipStream_t s) {
  if (!p.tile_ctr || p.nblk < 1 || p.nblk > 4 || p.W < 32 || (p.W & 1) || (pool && (p.H & 1))) return hipErrorInvalidValue;
  if ((long)p.tiles_x * p.tiles_y * p.B >= (1 << 20)) return hipErrorInvalidValue;
  if (tile_rows == 4) return pool ? rw::launch<4, true>(p, s) : rw::launch<4, false>(p, s);
  if (tile_rows == 3 && !pool) return rw::launch<3, false>(p, s);
  if (tile_rows == 2) return pool ? rw::launch<2, true>(p, s) : rw::launch<2, false>(p, s);
  return hipErrorInvalidValue;
}

// [ncg][wave 4][step 72][lane 64][8]: element j of lane (l31, hi) at step (chunk, dx, k2, dy) =
// W[cout = 128 cg + 32 wave + l31][cin = 32 chunk + 16 k2 + 8 hi + j][tap = 3 dy + dx], rounded to bf16 by the caller
void conv_bf16_rw_pack_weights(const unsigned short *Wb /* [cout][128][9] bf16 */, int cout, unsigned char *dst) {
  for (int co = 0; co < cout; ++co) {
    const int cg = co / 128, wave = (co % 128) / 32, l31 = co % 32;
    for (int s = 0; s < rw::NSTEP; ++s) {
      const rw::KStep k = rw::kstep_of(s);
      for (int hi = 0; hi < 2; ++hi)
        for (int j = 0; j < 8; ++j) {
          const int ci = 32 * k.chunk + 16 * k.k2 + 8 * hi + j;
          const unsigned short v = Wb[((size_t)co * 128 + ci) * 9 + k.tap];
          unsigned char *q = dst + ((((size_t)(cg * 4 + wave) * rw::NSTEP + s) * 64 + (hi * 32 + l31)) * 8 + j) * 2;
          q[0] = (unsigned char)(v & 0xff);
          q[1] = (unsigned char)(v >> 8);
        }
    }
  }
}

}  // namespace spfe
