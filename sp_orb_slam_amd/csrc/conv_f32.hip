// conv_f32.hip — exact-f32 implicit-GEMM convolutions for gfx950 (MI355X).
//
// Replaces the cuDNN convolutions libtorch runs for SPFrontend::forward
// (/root/reference/orb_slam2/src/cv/sp_extractor.cpp:81-100): 3x3 (pad 1) and
// 1x1 convs with bias, fused ReLU and fused 2x2 max-pool (:83,87,91).
//
// Design (MI355X-first, not a cuDNN translation):
//  * GEMM view: M = pixels of a TH x 32 spatial tile, N = 64 output channels per
//    workgroup, K = taps x input channels, fed to v_mfma_f32_32x32x2_f32
//    (157 TF/s dense peak, bit-exact k-ordered fma chain) so results equal the
//    CPU oracle BITWISE when K is walked in the order fixed by
//    include/spfe_exact_math.h: chunk of KC channels -> tap -> channel.
//  * LDS holds the input halo tile channel-major ([KC][rows][cols]): the 32
//    lanes of an MFMA A operand read 32 consecutive floats (conflict free) and a
//    filter tap is just an address offset; the 9 x KC x 64 weight slab ([n-tile][tap][k][32]) of the
//    chunk sits next to it (~60 KB), double buffered (120 KB): one persistent
//    workgroup per CU streams (tile, chunk) stages back to back.
//  * 64-wide wavefronts: each wave owns MT x NT 32x32 accumulator tiles
//    (two image rows -> the 2x2 pool is done in registers in the epilogue).
//  * workgroup -> tile mapping is XCD-aware: the 8 XCDs each get a contiguous
//    run of tiles so neighbouring tiles (shared halos, shared weight slabs) hit
//    the same L2.
#include <type_traits>
#include <utility>

#include "spfe_kernels.h"
#include "../../include/spfe_exact_math.h"

namespace spfe {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KS, int TH>
struct Geo {
  static constexpr int HALO = KS / 2;
  static constexpr int ROWS = TH + 2 * HALO;
  static constexpr int COLS = 32 + 2 * HALO;
  static constexpr int ROWP = (KS == 3) ? 36 : 32;
  static constexpr int PLANE_RAW = ROWS * ROWP;
  // PLANE % 8 == 2: the four channel planes a float4 staging store touches land
  // on banks 8 apart
  static constexpr int PLANE = PLANE_RAW + ((2 - PLANE_RAW % 8) + 8) % 8;
};

// ---------------------------------------------------------------------------
// Persistent, double-buffered implicit-GEMM convolution with everything but the
// MFMAs riding in the MFMA shadow.
//
//  * grid = one workgroup per CU (4 waves, one per SIMD, up to 512 registers
//    each); a workgroup walks an XCD-local run of (frame, tile, 64-channel block)
//    work items.
//  * the pipeline stages (tile, K chunk) form one continuous stream.  During the
//    72 K steps of stage s (4 MFMAs = 256 matrix-pipe cycles each) the same wave
//    also, a few instructions per step:
//       - reads the MFMA operands of the next step (LDS, one step ahead),
//       - issues the global loads of stage s+1           (steps 1 .. NLD),
//       - stores the PREVIOUS tile's outputs (bias/ReLU/pool epilogue) out of the
//         other accumulator set                           (steps 1 .. NEG, chunk 0),
//       - writes the loaded stage s+1 into the other LDS buffer (last NLD steps).
//    One barrier per stage; the first MFMA of a tile takes C = 0, so nothing is
//    zeroed or copied between tiles.
//    A wave issues in order, so work only hides if it is sliced this finely:
//    a 400-instruction epilogue in one piece idles the matrix pipe for ~3k cycles.
// ---------------------------------------------------------------------------
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define SPFE_OOB 0x80000000u  // byte offset beyond any buffer: loads return 0, stores are dropped

// Buffer (SRD) addressing: global traffic goes through buffer_load/buffer_store
// with hardware bounds checking, so zero padding at the image border, ragged
// tiles and the "no next stage" case cost no compare / exec-mask / select
// instructions in the MFMA shadow — an out-of-image piece simply carries the
// offset SPFE_OOB, a missing stage a resource with 0 records.
template <int NITER, int NWITER>
struct Pipe {
  float4 va[NITER];   // staged input pieces of the next stage (4 channels of one pixel each)
  float4 vw[NWITER];  // staged weight pieces
  int dst[NITER];     // LDS float offset of each input piece (a dummy slot for unused pieces)
  unsigned voff[NITER];  // byte offset of the piece inside the frame, or SPFE_OOB
  unsigned woff[NWITER]; // byte offset inside the weight slab
  __amdgpu_buffer_rsrc_t rin, rw;  // next stage: input frame (+channel offset), weight slab
  // this stage's operands in LDS.  at[t] = channel pair t of the halo tile: every A
  // read is then base + an 8-bit ds_read2 offset; the weight slab is stored
  // [n-tile][tap][k][32] so both B values of a step sit a multiple of 256 B apart
  // (one ds_read2st64_b32, no address arithmetic in the MFMA shadow).
  const float *at[8];
  const float *bBase;
  float *nA, *nW;                  // the other LDS buffer
};

template <int NT>
struct EpiCtx {  // the tile whose outputs are being stored
  __amdgpu_buffer_rsrc_t rout;  // output frame (+channel offset); 0 records = nothing to store
  unsigned obase[NT];           // per lane: byte offset of (tile origin row of this wave, x = tx0 + 4*hi, channel), or OOB
  float bias[NT];
  int xlim;                     // W - tx0 - 4*hi : columns left in the image for this lane
  int ylim;                     // H - (ty0 + wm*MT) : rows left for this wave
  unsigned rowstep, pixstep;    // bytes per output row / pixel
};

template <int MT, int NT, bool POOL, bool RELU, int E>
__device__ __forceinline__ void epi_store(const EpiCtx<NT> &e, const f32x16 (&acc)[MT][NT]) {
  // C layout of the 32x32 MFMA: column (N) = lane&31, row (M) = (r&3)+8*(r>>2)+4*(lane>>5)
  constexpr int NEPI_ = POOL ? (MT / 2) * NT * 8 : MT * NT * 16;
  if constexpr (E >= NEPI_) {
    return;
  } else if constexpr (!POOL) {
    constexpr int j = E / (MT * 16), i = (E / 16) % MT, r = E % 16;
    constexpr int xr = (r & 3) + 8 * (r >> 2);
    float v = acc[i][j][r] + e.bias[j];
    if (RELU) v = v > 0.0f ? v : 0.0f;
    const unsigned off = (xr < e.xlim && i < e.ylim) ? e.obase[j] + i * e.rowstep + xr * e.pixstep : SPFE_OOB;
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(v), e.rout, off, 0, 0);
  } else {
    constexpr int ip = E / (NT * 8), j = (E / 8) % NT, r = 2 * (E % 8), i0 = 2 * ip;   // pooled row ip = rows 2 ip, 2 ip + 1 of this wave
    constexpr int xr = (r & 3) + 8 * (r >> 2);
    float v00 = acc[i0][j][r] + e.bias[j], v01 = acc[i0][j][r + 1] + e.bias[j];
    float v10 = acc[i0 + 1][j][r] + e.bias[j], v11 = acc[i0 + 1][j][r + 1] + e.bias[j];
    if (RELU) {
      v00 = v00 > 0.0f ? v00 : 0.0f;
      v01 = v01 > 0.0f ? v01 : 0.0f;
      v10 = v10 > 0.0f ? v10 : 0.0f;
      v11 = v11 > 0.0f ? v11 : 0.0f;
    }
    const float m0 = v00 > v01 ? v00 : v01;
    const float m1 = v10 > v11 ? v10 : v11;
    const float v = m0 > m1 ? m0 : m1;
    const unsigned off = (xr < e.xlim && i0 < e.ylim) ? e.obase[j] + ip * e.rowstep + (xr >> 1) * e.pixstep : SPFE_OOB;
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(v), e.rout, off, 0, 0);
  }
}

// ---------------------------------------------------------------------------
// conv1a fused into conv1b (LAYER tag 2).  conv1a (u8 -> x 1/255 -> 3x3 conv 1->64, bias, ReLU,
// sp_extractor.cpp:81,388) is K = 9 VALU work whose only real cost as its own kernel is writing —
// and conv1b then re-reading — 92 MB of f32 activations per frame.  Fused, the workgroup keeps the
// 12 x 36 patch of the image under its halo tile in LDS and each lane COMPUTES the pieces of the
// next stage it used to load: 4 channels of one halo pixel = 4 chains of 9 fmaf in tap order,
// + bias, ReLU — the arithmetic of conv1a_kernel, so the result is bit-identical — sliced into the
// MFMA shadows (6 pieces x 6 sub-steps per stage).  Halo pixels outside the image are conv1b's
// zero padding, not conv1a evaluated there.
// ---------------------------------------------------------------------------
constexpr int F1_PROWS = 12, F1_PCOLS = 36, F1_PATCH = F1_PROWS * F1_PCOLS;  // floats per patch buffer

template <int NITER>
struct Fuse1a {
  float4 w[9], bias;        // taps / bias of this lane's 4 channels in the chunk being produced
  float px[9];              // 3x3 patch of the piece in flight
  float4 o;                 // its 4 outputs
  unsigned pb[NITER];       // patch float index (top-left of the 3x3) of each piece; ~0u: unused piece
  const float *patch;       // LDS patch of the tile whose stage is being produced
  float *patch_next;        // LDS patch buffer being filled for the tile after this one
  const float *wsrc;        // conv1a taps of the chunk being produced: w1a + chunk*16 + qq*4
  const float *bsrc;
  __amdgpu_buffer_rsrc_t rimg;  // frame of the tile after this one (0 records: none)
  unsigned poff[2];         // byte offset of this lane's two patch pixels in that frame, or OOB
  unsigned pidx[2];         // their float index in the patch buffer (>= F1_PATCH: none)
  unsigned pval[2];         // loaded bytes
  bool fill;                // this stage fills patch_next
};
struct NoFuse {};

constexpr int F1_SUBS = 10;  // sub-steps per piece: patch reads, 4 channels x 2 halves, LDS writes
template <int NITER, int IT, int SUB, int PLANE, int NWITER>
__device__ __forceinline__ void fuse_piece(Fuse1a<NITER> &z, Pipe<NITER, NWITER> &c) {
  if constexpr (SUB == 0) {
    const unsigned b = z.pb[IT] == ~0u ? 0u : z.pb[IT];
#pragma unroll
    for (int t = 0; t < 9; ++t) z.px[t] = z.patch[b + (t / 3) * F1_PCOLS + t % 3];
  } else if constexpr (SUB <= 8) {
    constexpr int j = (SUB - 1) / 2, half = (SUB - 1) % 2;
    float acc = half == 0 ? 0.0f : (j == 0 ? z.o.x : (j == 1 ? z.o.y : (j == 2 ? z.o.z : z.o.w)));
#pragma unroll
    for (int t = half * 5; t < (half == 0 ? 5 : 9); ++t) {
      const float wt = j == 0 ? z.w[t].x : (j == 1 ? z.w[t].y : (j == 2 ? z.w[t].z : z.w[t].w));
      acc = __builtin_fmaf(z.px[t], wt, acc);
    }
    if constexpr (half == 1) {
      const float bj = j == 0 ? z.bias.x : (j == 1 ? z.bias.y : (j == 2 ? z.bias.z : z.bias.w));
      acc = acc + bj;
      acc = acc > 0.0f ? acc : 0.0f;
      if (c.voff[IT] == SPFE_OOB) acc = 0.0f;  // outside the image: conv1b's zero padding
    }
    if constexpr (j == 0) z.o.x = acc;
    else if constexpr (j == 1) z.o.y = acc;
    else if constexpr (j == 2) z.o.z = acc;
    else z.o.w = acc;
  } else {
    float *d = c.nA + c.dst[IT];
    d[0] = z.o.x;
    d[PLANE] = z.o.y;
    d[2 * PLANE] = z.o.z;
    d[3 * PLANE] = z.o.w;
  }
}

template <int STEP, int NSTEP, bool FIRST, int KC, int KS, int MT, int NT, int PLANE, int ROWP, int NITER,
          int NWITER, bool POOL, bool RELU, bool FUSE, class FZ>
__device__ __forceinline__ void k_steps(float (&a)[2][MT], float (&bb)[2][NT], f32x16 (&acc)[MT][NT],
                                        const f32x16 (&accPrev)[MT][NT], Pipe<NITER, NWITER> &c,
                                        const EpiCtx<NT> &e, FZ &fz) {
  if constexpr (STEP < NSTEP) {
    constexpr int NLDA = FUSE ? 0 : NITER;  // fused: the input pieces are computed, not loaded
    constexpr int NLD = NLDA + NWITER;
    constexpr int L0 = 1, W0 = NSTEP - NLD - 1;
    constexpr int NEPI = POOL ? (MT / 2) * NT * 8 : MT * NT * 16;       // stores per wave per tile
    constexpr int EPS = (NEPI + (NSTEP - 3)) / (NSTEP - 2);  // stores per step
    constexpr int cur = STEP & 1, nxt = cur ^ 1;
    static_assert(L0 + NLD <= W0, "loads and LDS writes of a stage must not overlap");
    // The four kinds of side work are dealt out over the MT*NT gaps between this
    // step's MFMAs: a wave issues in order, so only what sits BETWEEN two MFMAs
    // runs in the 64-cycle shadow of the first.
    constexpr int M = MT * NT;
#pragma unroll
    for (int m = 0; m < M; ++m) {
      // (1) operands of the next step
      if (m == 0) {
        if constexpr (STEP + 1 < NSTEP) {
          constexpr int tap = (STEP + 1) / (KC / 2), t = (STEP + 1) % (KC / 2);
          constexpr int dy = tap / KS, dx = tap % KS;
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            if constexpr (KC / 2 <= 8) a[nxt][i] = c.at[t][(i + dy) * ROWP + dx];
            else a[nxt][i] = c.at[0][(2 * t) * PLANE + (i + dy) * ROWP + dx];
          }
#pragma unroll
          for (int j = 0; j < NT; ++j) bb[nxt][j] = c.bBase[j * (KS * KS * KC * 32) + (tap * KC + 2 * t) * 32];
        }
      }
      // (2) one global load of the next stage
      if (m == 1 % M) {
        if constexpr (STEP >= L0 && STEP - L0 < NLD) {
          constexpr int it = STEP - L0;
          if constexpr (it < NLDA) {
            const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(c.rin, c.voff[it], 0, 0);
            c.va[it] = make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
          } else {
            constexpr int wi = it - NLDA;
            const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(c.rw, c.woff[wi], 0, 0);
            c.vw[wi] = make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
          }
        }
      }
      // (3) a slice of the previous tile's epilogue
      if (m == 2 % M) {
        if constexpr (FIRST && STEP >= 1) {
          constexpr int e0 = (STEP - 1) * EPS;
          [&]<int... Qs>(std::integer_sequence<int, Qs...>) {
            (epi_store<MT, NT, POOL, RELU, e0 + Qs>(e, accPrev), ...);
          }(std::make_integer_sequence<int, EPS>{});
        }
      }
      // (4) one staged piece of the next stage into the other LDS buffer
      if (m == 3 % M) {
        if constexpr (STEP >= W0 && STEP - W0 < NLD) {
          constexpr int it = STEP - W0;
          if constexpr (it < NLDA) {
            float *d = c.nA + c.dst[it];
            d[0] = c.va[it].x;
            d[PLANE] = c.va[it].y;
            d[2 * PLANE] = c.va[it].z;
            d[3 * PLANE] = c.va[it].w;
          } else {
            constexpr int wi = it - NLDA;
            *reinterpret_cast<float4 *>(reinterpret_cast<char *>(c.nW) + c.woff[wi]) = c.vw[wi];
          }
        }
      }
      // (5) fused conv1a: taps of the chunk being produced (steps 2..11, gap 3), its NITER pieces
      //     (10 sub-steps of <= 6 instructions each from step 12, gap 1 — free once the slab loads are
      //     out), the next tile's image patch (load at 50, LDS at 58, gap 3)
      if (m == 1 % M) {
        if constexpr (FUSE) {
          constexpr int F0 = 12;
          static_assert(L0 + NLD <= F0 && F0 + F1_SUBS * NITER <= NSTEP, "fused conv1a schedule does not fit the stage");
          if constexpr (STEP >= F0 && STEP < F0 + F1_SUBS * NITER)
            fuse_piece<NITER, (STEP - F0) / F1_SUBS, (STEP - F0) % F1_SUBS, PLANE, NWITER>(fz, c);
        }
      }
      if (m == 3 % M) {
        if constexpr (FUSE) {
          static_assert(58 < W0, "fused conv1a schedule does not fit the stage");
          if constexpr (STEP >= 2 && STEP < 11) fz.w[STEP - 2] = *reinterpret_cast<const float4 *>(fz.wsrc + (STEP - 2) * 64);
          if constexpr (STEP == 11) fz.bias = *reinterpret_cast<const float4 *>(fz.bsrc);
          if constexpr (STEP == 50) {
            if (fz.fill) {
              fz.pval[0] = __builtin_amdgcn_raw_buffer_load_b8(fz.rimg, fz.poff[0], 0, 0);
              fz.pval[1] = __builtin_amdgcn_raw_buffer_load_b8(fz.rimg, fz.poff[1], 0, 0);
            }
          }
          if constexpr (STEP == 58) {
            if (fz.fill) {
              if (fz.pidx[0] < F1_PATCH) fz.patch_next[fz.pidx[0]] = spfe_pixel_to_float((uint8_t)fz.pval[0]);
              if (fz.pidx[1] < F1_PATCH) fz.patch_next[fz.pidx[1]] = spfe_pixel_to_float((uint8_t)fz.pval[1]);
            }
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
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], bb[cur][j], z, 0, 0, 0);
        } else {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], bb[cur][j], acc[i][j], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    k_steps<STEP + 1, NSTEP, FIRST, KC, KS, MT, NT, PLANE, ROWP, NITER, NWITER, POOL, RELU, FUSE, FZ>(a, bb, acc, accPrev, c, e, fz);
  }
}

// LAYER only names the instantiation (conv1b gets its own symbol so profiles can
// tell the dominant launch from the other layers that share its shape).
template <int LAYER, int CIN, int KS, int KC, int WM, int WN, int MT, int NT, bool POOL, bool RELU>
__global__ __launch_bounds__(64 * WM * WN, 1) void conv_f32_kernel(ConvParams p) {
  constexpr int TH = WM * MT;
  using G = Geo<KS, TH>;
  constexpr int TAPS = KS * KS;
  constexpr int NCHUNK = CIN / KC;
  constexpr int PLANE = G::PLANE;
  constexpr int ROWP = G::ROWP;
  constexpr int BUF = KC * PLANE + TAPS * KC * 64;  // floats per LDS buffer
  // 4 waves = one per SIMD; 8 waves = two per SIMD on a 16-row tile: while one wave of a SIMD does
  // its side work (operand reads, staging, epilogue — it issues in order, so that work costs matrix
  // time when the wave is alone) the other one's MFMAs keep the matrix pipe busy.
  static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves per workgroup");
  constexpr int NTHR = 64 * WM * WN;
  static_assert(WN * NT == 2, "64 output channels per workgroup");
  static_assert(!POOL || MT % 2 == 0, "pooling needs row pairs per wave");
  static_assert((KC * PLANE) % 4 == 0 && BUF % 4 == 0, "weight slabs must stay 16B aligned");
  constexpr int Q = KC / 4;  // float4 per pixel per chunk
  constexpr int NITEM = G::ROWS * G::COLS * Q;
  constexpr int NITER = (NITEM + NTHR - 1) / NTHR;
  constexpr int NW4 = TAPS * KC * 16;  // float4 in the weight slab
  constexpr int NWITER = (NW4 + NTHR - 1) / NTHR;  // a ragged last pass re-stages pieces of the first (same data, same place)
  constexpr int NSTEP = TAPS * (KC / 2);
  constexpr int SLAB_BYTES = TAPS * KC * 64 * 4;

  constexpr bool FUSE = LAYER == 2;  // conv1a computed in place of the input loads (conv1b only)
  static_assert(!FUSE || (CIN == 64 && KS == 3 && KC == 16 && TH == 8), "fused conv1a is conv1b's first operand");
  using FZ = typename std::conditional<FUSE, Fuse1a<NITER>, NoFuse>::type;

  extern __shared__ __attribute__((aligned(16))) float smem[];  // 2 buffers (+ 2 image patches when fused)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int wm = wave % WM, wn = wave / WM;
  const int H = p.H, W = p.W;

  // XCD-local work range: workgroup g runs on XCD g % 8 (observed placement; only
  // speed depends on it).  Each XCD owns one contiguous eighth of the work list
  // (64-channel blocks of a tile adjacent), its workgroups interleave inside it.
  const int total = p.nblk * p.tiles_x * p.tiles_y * p.B;
  const int first = p.item_lo, count = (p.item_hi > 0 ? p.item_hi : total) - first;   // this launch's part of the list
  const int xcd = blockIdx.x & 7, gi = blockIdx.x >> 3, gper = gridDim.x >> 3;
  const int lo = first + (int)((long)count * xcd / 8), hi_w = first + (int)((long)count * (xcd + 1) / 8);
  int w = lo + gi;
  if (w >= hi_w) return;

  // work item = (nb, tx, ty, b); advanced incrementally by gper (no divisions in the loop)
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

  const unsigned in_pix_bytes = (unsigned)p.in_stride * 4u;
  const unsigned frame_in_bytes = (unsigned)H * W * in_pix_bytes;
  const int Ho = POOL ? H >> 1 : H, Wo = POOL ? W >> 1 : W;
  const unsigned out_pix_bytes = (unsigned)p.out_stride * 4u;
  const unsigned frame_out_bytes = (unsigned)Ho * Wo * out_pix_bytes;

  Pipe<NITER, NWITER> c;
  int prow[NITER], pcol[NITER];
  unsigned pqb[NITER];
#pragma unroll
  for (int it = 0; it < NITER; ++it) {
    const int i = tid + it * NTHR;
    const int qq = i % Q, pix = i / Q;
    prow[it] = i < NITEM ? pix / G::COLS - G::HALO : (1 << 20);  // unused piece: never inside the image
    pcol[it] = pix % G::COLS - G::HALO;
    pqb[it] = qq * 16;
    // an unused piece (tid + it*256 >= NITEM) is written where nothing is ever read: a pad
    // column of the halo rows (planes 0..3), or the pad floats that end each plane
    c.dst[it] = i < NITEM ? (qq * 4) * PLANE + (pix / G::COLS) * ROWP + pix % G::COLS
                          : (ROWP > G::COLS ? (tid % G::ROWS) * ROWP + G::COLS : G::PLANE_RAW);
  }
#pragma unroll
  for (int it = 0; it < NWITER; ++it) c.woff[it] = ((tid + it * NTHR) % NW4) * 16;

  // per-thread frame offsets of the input pieces of tile (tx, ty)
  auto aim_tile = [&](int tx, int ty) {
#pragma unroll
    for (int it = 0; it < NITER; ++it) {
      const int gy = ty * TH + prow[it], gx = tx * 32 + pcol[it];
      c.voff[it] = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
                       ? (unsigned)(gy * W + gx) * in_pix_bytes + pqb[it]
                       : SPFE_OOB;
    }
  };
  // next stage = (frame b, 64-channel block nb, chunk); valid == false: a stage that does not exist
  auto aim_stage = [&](int nb, int b, int chunk, bool valid) {
    const float *base = p.in + (size_t)b * H * W * p.in_stride + p.in_choff + chunk * KC;
    c.rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, valid ? frame_in_bytes : 0u, 0x00020000);
    const float *wb = p.wpack + ((size_t)nb * NCHUNK + chunk) * (TAPS * KC * 64);
    c.rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(wb), 0, valid ? (unsigned)SLAB_BYTES : 0u, 0x00020000);
  };

  // prologue: stage (first item, chunk 0) straight into buffer 0
  aim_tile(i_tx, i_ty);
  aim_stage(i_nb, i_b, 0, true);
  FZ fz;
  [[maybe_unused]] float *sP = smem + 2 * BUF;  // fused: two 12 x 36 image patches
  [[maybe_unused]] int pcur = 0;
  [[maybe_unused]] int php[2] = {0, 0}, pwp[2] = {0, 0};
  if constexpr (FUSE) {
    const int qq = tid % Q;
#pragma unroll
    for (int it = 0; it < NITER; ++it) {
      const int i = tid + it * 256, pix = i / Q;
      fz.pb[it] = i < NITEM ? (unsigned)((pix / G::COLS) * F1_PCOLS + pix % G::COLS) : ~0u;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = tid + k * 256;
      fz.pidx[k] = i < F1_PATCH ? (unsigned)i : 0xffffu;
      php[k] = i / F1_PCOLS;
      pwp[k] = i % F1_PCOLS;
      fz.poff[k] = SPFE_OOB;
      fz.pval[k] = 0u;
    }
    fz.fill = false;
    fz.patch = sP;
    fz.patch_next = sP + F1_PATCH;
    fz.rimg = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(p.img), 0, 0u, 0x00020000);
    fz.wsrc = p.w1a + qq * 4;
    fz.bsrc = p.b1a + qq * 4;
    // the first tile's patch, then its chunk-0 pieces, computed in the open
    {
      const uint8_t *ib = p.img + (size_t)i_b * H * W;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int gy = i_ty * TH - 2 + php[k], gx = i_tx * 32 - 2 + pwp[k];
        const bool in = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        if (fz.pidx[k] < F1_PATCH) sP[fz.pidx[k]] = in ? spfe_pixel_to_float(ib[(size_t)gy * W + gx]) : 0.0f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 9; ++t) fz.w[t] = *reinterpret_cast<const float4 *>(fz.wsrc + t * 64);
    fz.bias = *reinterpret_cast<const float4 *>(fz.bsrc);
    c.nA = smem;
    [&]<int... I>(std::integer_sequence<int, I...>) {
      (fuse_piece<NITER, I / F1_SUBS, I % F1_SUBS, PLANE, NWITER>(fz, c), ...);
    }(std::make_integer_sequence<int, F1_SUBS * NITER>{});
  } else {
#pragma unroll
    for (int it = 0; it < NITER; ++it) {
      const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(c.rin, c.voff[it], 0, 0);
      c.va[it] = make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
    }
  }
#pragma unroll
  for (int it = 0; it < NWITER; ++it) {
    const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(c.rw, c.woff[it], 0, 0);
    c.vw[it] = make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
  }
  if constexpr (!FUSE) {
#pragma unroll
    for (int it = 0; it < NITER; ++it) {
      float *d = smem + c.dst[it];
      d[0] = c.va[it].x;
      d[PLANE] = c.va[it].y;
      d[2 * PLANE] = c.va[it].z;
      d[3 * PLANE] = c.va[it].w;
    }
  }
#pragma unroll
  for (int it = 0; it < NWITER; ++it)
    *reinterpret_cast<float4 *>(reinterpret_cast<char *>(smem + KC * PLANE) + c.woff[it]) = c.vw[it];
  __syncthreads();

  f32x16 accA[MT][NT], accB[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { accA[i][j][r] = 0.0f; accB[i][j][r] = 0.0f; }

  int buf = 0;
  EpiCtx<NT> epi;
  epi.rout = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, 0u, 0x00020000);  // nothing to store yet
#pragma unroll
  for (int j = 0; j < NT; ++j) { epi.obase[j] = SPFE_OOB; epi.bias[j] = 0.0f; }
  epi.xlim = 0; epi.ylim = 0; epi.rowstep = 0; epi.pixstep = out_pix_bytes;
  bool more = true;

  // describe tile (nb, tx, ty, b) for the epilogue; the bias is loaded one tile ahead
  auto aim_epi = [&](EpiCtx<NT> &e, int nb, int tx, int ty, int b) {
    float *obase = p.out + (size_t)b * Ho * Wo * p.out_stride + p.out_choff;
    e.rout = __builtin_amdgcn_make_buffer_rsrc(obase, 0, frame_out_bytes, 0x00020000);
    const int y0 = ty * TH + wm * MT, x0 = tx * 32 + 4 * hi;
    e.xlim = W - x0;
    e.ylim = H - y0;
    e.rowstep = (unsigned)Wo * out_pix_bytes;
    e.pixstep = out_pix_bytes;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int co = nb * 64 + (wn * NT + j) * 32 + l31;
      const unsigned pix = POOL ? (unsigned)((y0 >> 1) * Wo + (x0 >> 1)) : (unsigned)(y0 * W + x0);
      e.obase[j] = co < p.cout_real ? pix * out_pix_bytes + (unsigned)co * 4u : SPFE_OOB;
      e.bias[j] = p.bias[co];
    }
  };

  // one tile: NCHUNK stages accumulating into `acc`, while the previous tile's
  // outputs (in `accPrev`, described by `epi`) are stored during the first stage
  EpiCtx<NT> epi_next;
  auto run_tile = [&](f32x16(&acc)[MT][NT], const f32x16(&accPrev)[MT][NT]) {
    // the item after this one
    int n_nb = i_nb + d_nb, n_tx = i_tx + d_tx, n_ty = i_ty + d_ty, n_b = i_b + d_b;
    if (n_nb >= p.nblk) { n_nb -= p.nblk; ++n_tx; }
    if (n_tx >= p.tiles_x) { n_tx -= p.tiles_x; ++n_ty; }
    if (n_ty >= p.tiles_y) { n_ty -= p.tiles_y; ++n_b; }
    const bool have_next_item = w + gper < hi_w;
    aim_epi(epi_next, i_nb, i_tx, i_ty, i_b);  // this tile's epilogue context (bias in flight for a whole tile)
#pragma unroll 1
    for (int chunk = 0; chunk < NCHUNK; ++chunk) {
      const bool last = chunk == NCHUNK - 1;
      if (!last) {
        aim_stage(i_nb, i_b, chunk + 1, true);
      } else {
        aim_tile(n_tx, n_ty);
        aim_stage(n_nb, n_b, 0, have_next_item);
      }
      if constexpr (FUSE) {
        const int produced = last ? 0 : chunk + 1;   // chunk of the stage being produced
        const int qq = tid % Q;
        fz.wsrc = p.w1a + produced * KC + qq * 4;
        fz.bsrc = p.b1a + produced * KC + qq * 4;
        fz.patch = sP + (last ? pcur ^ 1 : pcur) * F1_PATCH;  // the last stage produces the NEXT tile's first
        fz.patch_next = sP + (pcur ^ 1) * F1_PATCH;
        fz.fill = chunk == NCHUNK - 2 && have_next_item;      // one stage earlier its patch is fetched
        if (fz.fill) {
          fz.rimg = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(p.img) + (size_t)n_b * H * W, 0,
                                                      (unsigned)(H * W), 0x00020000);
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int gy = n_ty * TH - 2 + php[k], gx = n_tx * 32 - 2 + pwp[k];
            fz.poff[k] = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? (unsigned)(gy * W + gx) : SPFE_OOB;
          }
        }
      }
      float *cA = smem + buf * BUF;
      c.nA = smem + (buf ^ 1) * BUF;
      c.nW = c.nA + KC * PLANE;
#pragma unroll
      for (int t = 0; t < (KC / 2 <= 8 ? KC / 2 : 1); ++t)
        c.at[t] = cA + (2 * t + hi) * PLANE + (wm * MT) * ROWP + l31;
      c.bBase = cA + KC * PLANE + (wn * NT) * (TAPS * KC * 32) + hi * 32 + l31;
      float a[2][MT], bb[2][NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) a[0][i] = c.at[0][i * ROWP];
#pragma unroll
      for (int j = 0; j < NT; ++j) bb[0][j] = c.bBase[j * (TAPS * KC * 32)];
      if (chunk == 0)
        k_steps<0, NSTEP, true, KC, KS, MT, NT, PLANE, ROWP, NITER, NWITER, POOL, RELU, FUSE, FZ>(a, bb, acc, accPrev, c, epi, fz);
      else
        k_steps<0, NSTEP, false, KC, KS, MT, NT, PLANE, ROWP, NITER, NWITER, POOL, RELU, FUSE, FZ>(a, bb, acc, accPrev, c, epi, fz);
      __syncthreads();  // the other buffer is complete, this one is free
      buf ^= 1;
    }
    epi = epi_next;  // this tile's outputs are stored while the next tile starts
    if constexpr (FUSE) pcur ^= 1;
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
  // the last tile's outputs: nothing left to hide them under
  {
    constexpr int NEPI = POOL ? (MT / 2) * NT * 8 : MT * NT * 16;
    auto flush = [&](const f32x16(&acc)[MT][NT]) {
      [&]<int... E>(std::integer_sequence<int, E...>) {
        (epi_store<MT, NT, POOL, RELU, E>(epi, acc), ...);
      }(std::make_integer_sequence<int, NEPI>{});
    };
    if (lastA) flush(accA); else flush(accB);
  }
}

template <int LAYER, int CIN, int KS, int KC, int WM, int WN, int MT, int NT, bool POOL, bool RELU>
static hipError_t launch_one(const ConvParams &p, hipStream_t s) {
  constexpr int TH = WM * MT;
  using G = Geo<KS, TH>;
  constexpr size_t lds = (2 * (size_t)(KC * G::PLANE + KS * KS * KC * 64) + (LAYER == 2 ? 2 * F1_PATCH : 0)) * sizeof(float);
  static_assert(lds <= 160 * 1024, "double buffer must fit the 160 KB LDS");
  auto k = conv_f32_kernel<LAYER, CIN, KS, KC, WM, WN, MT, NT, POOL, RELU>;
  static bool attr_done[64] = {};  // per instantiation and device: one process may hold handles on several GPUs
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) attr_done[dev] = true;
  }
  const int total = p.nblk * p.tiles_x * p.tiles_y * p.B;
  int grid = p.num_cus > 0 ? p.num_cus : 256;  // one persistent workgroup per CU
  grid &= ~7;
  if (grid < 8) grid = 8;
  (void)total;
  hipLaunchKernelGGL(k, dim3(grid), dim3(64 * WM * WN), lds, s, p);
  return hipGetLastError();
}

int conv_kc(int ksize) { return ksize == 3 ? 16 : 64; }
int conv_tile_rows(int tile_mode) { return tile_mode == 1 ? 4 : (tile_mode == 2 || tile_mode == 4 ? 16 : (tile_mode == 3 ? 2 : 8)); }

hipError_t launch_conv_f32(const ConvParams &p, int cin, int ksize, bool pool, bool relu,
                           int tile_mode, int layer_tag, hipStream_t s) {
  const bool small_tile = tile_mode == 1;
  if (tile_mode == 3 && ksize == 3 && relu && !pool) {  // 2-row tiles (2x2 waves, one row x 32 channels each): single frames
    if (cin == 64) return launch_one<0, 64, 3, 16, 2, 2, 1, 1, false, true>(p, s);   // on the low-resolution layers, where
    if (cin == 128) return launch_one<0, 128, 3, 16, 2, 2, 1, 1, false, true>(p, s);  // a 4-row item per CU leaves CUs idle
    return hipErrorInvalidValue;
  }
  if (tile_mode == 4 && ksize == 3 && relu && pool && cin == 64 && layer_tag == 1)   // conv1b, 16-row tiles of 4 waves x 4 rows
    return launch_one<1, 64, 3, 16, 4, 1, 4, 2, true, true>(p, s);
  if (tile_mode == 2 && ksize == 3 && relu) {  // 16-row tiles, 8 waves (two per SIMD)
    if (layer_tag == 1 && cin == 64 && pool) return launch_one<1, 64, 3, 16, 8, 1, 2, 2, true, true>(p, s);  // conv1b
    if (cin == 64 && pool) return launch_one<0, 64, 3, 16, 8, 1, 2, 2, true, true>(p, s);
    if (cin == 64 && !pool) return launch_one<0, 64, 3, 16, 8, 1, 2, 2, false, true>(p, s);
    if (cin == 128 && pool) return launch_one<0, 128, 3, 16, 8, 1, 2, 2, true, true>(p, s);
    if (cin == 128 && !pool) return launch_one<0, 128, 3, 16, 8, 1, 2, 2, false, true>(p, s);
    return hipErrorInvalidValue;
  }
  if (layer_tag == 1 && cin == 64 && ksize == 3 && pool && relu && !small_tile)
    return launch_one<1, 64, 3, 16, 4, 1, 2, 2, true, true>(p, s);  // conv1b
  if (layer_tag == 2 && cin == 64 && ksize == 3 && pool && relu && !small_tile && p.img && p.w1a && p.b1a)
    return launch_one<2, 64, 3, 16, 4, 1, 2, 2, true, true>(p, s);  // conv1a + conv1b fused
#define SPFE_CONV(CIN_, KS_, KC_, WM_, WN_, MT_, NT_, POOL_, RELU_)                       \
  if (cin == CIN_ && ksize == KS_ && pool == POOL_ && relu == RELU_ &&                    \
      small_tile == (WM_ * MT_ == 4))                                                     \
    return launch_one<0, CIN_, KS_, KC_, WM_, WN_, MT_, NT_, POOL_, RELU_>(p, s);
  // 8x32-pixel tiles: 4 waves stacked in M, each 2 rows x 64 channels
  SPFE_CONV(64, 3, 16, 4, 1, 2, 2, true, true)
  SPFE_CONV(64, 3, 16, 4, 1, 2, 2, false, true)
  SPFE_CONV(128, 3, 16, 4, 1, 2, 2, true, true)
  SPFE_CONV(128, 3, 16, 4, 1, 2, 2, false, true)
  // 4x32-pixel tiles: 2x2 waves, each 2 rows x 32 channels
  SPFE_CONV(64, 3, 16, 2, 2, 2, 1, true, true)
  SPFE_CONV(64, 3, 16, 2, 2, 2, 1, false, true)
  SPFE_CONV(128, 3, 16, 2, 2, 2, 1, true, true)
  SPFE_CONV(128, 3, 16, 2, 2, 2, 1, false, true)
  // 1x1 heads (convPb, convDb): K = 256 channels in chunks of 64
  SPFE_CONV(256, 1, 64, 2, 2, 2, 1, false, false)
#undef SPFE_CONV
  return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------
// conv1a: u8 -> f32 * (1/255) (sp_extractor.cpp:388) -> 3x3 conv 1->64, bias, ReLU (:81).
// K = 9 is too small for MFMA; a VALU kernel whose only real cost is the NHWC
// store (256 B per pixel, written as whole 1 KiB wave stores).
// 16 lanes per pixel (one float4 of channels each), 4 pixels per wave step.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv1a_kernel(const uint8_t *__restrict__ img,
                                                     const float *__restrict__ w9x64,
                                                     const float *__restrict__ b64,
                                                     float *__restrict__ out, int B, int H, int W,
                                                     int tiles_x, int tiles_y) {
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
  const int c4 = tid & 15;  // channels 4*c4 .. 4*c4+3
  float4 w[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) w[t] = *reinterpret_cast<const float4 *>(w9x64 + t * 64 + c4 * 4);
  const float4 bias = *reinterpret_cast<const float4 *>(b64 + c4 * 4);
  __syncthreads();
  const int psub = tid >> 4;  // 0..15
#pragma unroll 4
  for (int it = 0; it < 16; ++it) {
    const int pix = it * 16 + psub;  // 0..255, row-major in the 8x32 tile
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
    a.x += bias.x;
    a.y += bias.y;
    a.z += bias.z;
    a.w += bias.w;
    a.x = a.x > 0.f ? a.x : 0.f;
    a.y = a.y > 0.f ? a.y : 0.f;
    a.z = a.z > 0.f ? a.z : 0.f;
    a.w = a.w > 0.f ? a.w : 0.f;
    const int gy = ty0 + row, gx = tx0 + col;
    if (gy < H && gx < W)
      *reinterpret_cast<float4 *>(out + (((size_t)b * H + gy) * W + gx) * 64 + c4 * 4) = a;
  }
}

// 2x2 / 2 max-pool of an NHWC f32 activation (a float4 of channels per lane).  Used when a POOLED layer of a single frame runs
// as un-pooled 2-row tiles (spfe_schedule.hip): bias, ReLU and the maximum commute exactly (all monotonic), so pooling the stored
// ReLU outputs gives the bits of the fused epilogue.
__global__ __launch_bounds__(256) void pool2x2_f32_kernel(const float4 *__restrict__ in, float4 *__restrict__ out, int B, int Ho,
                                                          int Wo, int c4) {
  const size_t n = (size_t)B * Ho * Wo * c4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % c4);
    size_t t = i / c4;
    const int x = (int)(t % Wo);
    t /= Wo;
    const int y = (int)(t % Ho), b = (int)(t / Ho);
    const size_t W = (size_t)Wo * 2, base = (((size_t)b * Ho * 2 + 2 * y) * W + 2 * x) * c4 + c;
    const float4 a = in[base], bq = in[base + c4], cq = in[base + W * c4], d = in[base + W * c4 + c4];
    float4 o;
    o.x = fmaxf(fmaxf(a.x, bq.x), fmaxf(cq.x, d.x));
    o.y = fmaxf(fmaxf(a.y, bq.y), fmaxf(cq.y, d.y));
    o.z = fmaxf(fmaxf(a.z, bq.z), fmaxf(cq.z, d.z));
    o.w = fmaxf(fmaxf(a.w, bq.w), fmaxf(cq.w, d.w));
    out[i] = o;
  }
}
hipError_t launch_pool2x2_f32(const float *in, float *out, int B, int H, int W, int C, hipStream_t s) {
  if ((H & 1) || (W & 1) || (C & 3)) return hipErrorInvalidValue;
  const size_t n = (size_t)B * (H / 2) * (W / 2) * (C / 4);
  const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(pool2x2_f32_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<const float4 *>(in),
                     reinterpret_cast<float4 *>(out), B, H / 2, W / 2, C / 4);
  return hipGetLastError();
}

hipError_t launch_conv1a(const uint8_t *img, const float *w9x64, const float *b64, float *out, int B,
                         int H, int W, hipStream_t s) {
  const int tiles_x = (W + 31) / 32, tiles_y = (H + 7) / 8;
  hipLaunchKernelGGL(conv1a_kernel, dim3(tiles_x * tiles_y * B), dim3(256), 0, s, img, w9x64, b64,
                     out, B, H, W, tiles_x, tiles_y);
  return hipGetLastError();
}

}  // namespace spfe
