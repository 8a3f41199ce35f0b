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
//    filter tap is just an address offset; the 9 x KC x 64 weight slab of the
//    chunk sits next to it.  ~60 KB per workgroup -> 2 workgroups per CU, so one
//    stages while the other issues MFMAs.
//  * 64-wide wavefronts: each wave owns MT x NT 32x32 accumulator tiles
//    (two image rows -> the 2x2 pool is done in registers in the epilogue).
//  * workgroup -> tile mapping is XCD-aware: the 8 XCDs each get a contiguous
//    run of tiles so neighbouring tiles (shared halos, shared weight slabs) hit
//    the same L2.
#include "spfe_kernels.h"

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

// One K step = one (tap, channel pair): MT x NT MFMAs on operand set STEP & 1, while
// the operands of step STEP + 1 are being read into the other set.  Compile-time
// recursion keeps every LDS offset an immediate.
template <int STEP, int NSTEP, int KC, int KS, int MT, int NT, int PLANE, int ROWP>
__device__ __forceinline__ void mfma_steps(float (&a)[2][MT], float (&bb)[2][NT], f32x16 (&acc)[MT][NT],
                                           const float *aBase, const float *bBase) {
  if constexpr (STEP < NSTEP) {
    constexpr int cur = STEP & 1, nxt = cur ^ 1;
    if constexpr (STEP + 1 < NSTEP) {
      constexpr int tap = (STEP + 1) / (KC / 2), t = (STEP + 1) % (KC / 2);
      constexpr int dy = tap / KS, dx = tap % KS;
#pragma unroll
      for (int i = 0; i < MT; ++i) a[nxt][i] = aBase[(2 * t) * PLANE + (i + dy) * ROWP + dx];
#pragma unroll
      for (int j = 0; j < NT; ++j) bb[nxt][j] = bBase[(tap * KC + 2 * t) * 64 + j * 32];
    }
    // keep the reads of step+1 ABOVE the MFMAs of this step (the machine scheduler
    // otherwise sinks them to just before their use and exposes the LDS latency)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], bb[cur][j], acc[i][j], 0, 0, 0);
    mfma_steps<STEP + 1, NSTEP, KC, KS, MT, NT, PLANE, ROWP>(a, bb, acc, aBase, bBase);
  }
}

template <int CIN, int KS, int KC, int WM, int WN, int MT, int NT, bool POOL, bool RELU>
__global__ __launch_bounds__(256, 2) void conv_f32_kernel(ConvParams p) {
  constexpr int TH = WM * MT;
  using G = Geo<KS, TH>;
  constexpr int TAPS = KS * KS;
  constexpr int NCHUNK = CIN / KC;
  constexpr int PLANE = G::PLANE;
  constexpr int ROWP = G::ROWP;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  static_assert(WN * NT == 2, "64 output channels per workgroup");
  static_assert(!POOL || MT == 2, "pooling needs two rows per wave");
  static_assert((KC * PLANE) % 4 == 0, "weight slab must stay 16B aligned");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *sA = smem;               // [KC][PLANE]
  float *sW = smem + KC * PLANE;  // [TAPS][KC][64]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int wm = wave % WM, wn = wave / WM;

  // XCD-aware bijective remap (blocks are dealt round-robin to the 8 XCDs)
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int nb = wg % p.nblk;
  wg /= p.nblk;
  const int tx = wg % p.tiles_x;
  wg /= p.tiles_x;
  const int ty = wg % p.tiles_y;
  const int b = wg / p.tiles_y;
  const int tx0 = tx * 32, ty0 = ty * TH;
  const int H = p.H, W = p.W;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const float *inb = p.in + (size_t)b * H * W * p.in_stride + p.in_choff;
  const float *aBase = sA + hi * PLANE + (wm * MT) * ROWP + l31;
  const float *bBase = sW + hi * 64 + (wn * NT) * 32 + l31;

  constexpr int Q = KC / 4;  // float4 per pixel per chunk
  constexpr int NITEM = G::ROWS * G::COLS * Q;
  constexpr int NITER = (NITEM + 255) / 256;
  constexpr int NW4 = TAPS * KC * 16;  // float4 in the weight slab

  // Register-staged software pipeline: the global loads of chunk c+1 are issued
  // before the MFMA loop of chunk c and only written to LDS after it, so their
  // latency hides under ~18k cycles of matrix work instead of standing between two
  // barriers.
  constexpr int NWITER = (NW4 + 255) / 256;
  float4 va[NITER], vw[NWITER];
  int dst[NITER];
#pragma unroll
  for (int it = 0; it < NITER; ++it) {  // per-thread staging slots are the same for every chunk
    const int i = tid + it * 256;
    const int qq = i % Q, pix = i / Q;
    const int row = pix / G::COLS, col = pix % G::COLS;
    dst[it] = (qq * 4) * PLANE + row * ROWP + col;
  }
#define SPFE_LOAD_CHUNK(CHUNK_)                                                                       \
  do {                                                                                                \
    _Pragma("unroll") for (int it = 0; it < NITER; ++it) {                                            \
      const int i = tid + it * 256;                                                                   \
      const int qq = i % Q, pix = i / Q;                                                              \
      const int row = pix / G::COLS, col = pix % G::COLS;                                             \
      const int gy = ty0 + row - G::HALO, gx = tx0 + col - G::HALO;                                   \
      va[it] = make_float4(0.f, 0.f, 0.f, 0.f);                                                       \
      if (i < NITEM && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)                      \
        va[it] = *reinterpret_cast<const float4 *>(inb + ((size_t)gy * W + gx) * p.in_stride +       \
                                                   (CHUNK_) * KC + qq * 4);                           \
    }                                                                                                 \
    const float4 *ws_ = reinterpret_cast<const float4 *>(                                             \
        p.wpack + ((size_t)nb * NCHUNK + (CHUNK_)) * (TAPS * KC * 64));                               \
    _Pragma("unroll") for (int it = 0; it < NWITER; ++it) {                                           \
      const int i = tid + it * 256;                                                                   \
      vw[it] = (i < NW4) ? ws_[i] : make_float4(0.f, 0.f, 0.f, 0.f);                                  \
    }                                                                                                 \
  } while (0)
  SPFE_LOAD_CHUNK(0);

#pragma unroll 1
  for (int chunk = 0; chunk < NCHUNK; ++chunk) {
    if (chunk) __syncthreads();  // every wave is done reading the previous chunk's tiles
#pragma unroll
    for (int it = 0; it < NITER; ++it) {
      const int i = tid + it * 256;
      if (i < NITEM) {
        float *d = sA + dst[it];
        d[0] = va[it].x;
        d[PLANE] = va[it].y;
        d[2 * PLANE] = va[it].z;
        d[3 * PLANE] = va[it].w;
      }
    }
    {
      float4 *wd = reinterpret_cast<float4 *>(sW);
#pragma unroll
      for (int it = 0; it < NWITER; ++it) {
        const int i = tid + it * 256;
        if (i < NW4) wd[i] = vw[it];
      }
    }
    __syncthreads();
    if (chunk + 1 < NCHUNK) SPFE_LOAD_CHUNK(chunk + 1);  // in flight during the MFMA loop below

    // ---- MFMA over (tap, channel pair) in the contract's K order ----
    // Operand reads are software pipelined one step ahead into a second register
    // set: issued back to back with the MFMAs that consume the previous set, the
    // ~100-cycle LDS latency hides under 4 x 64 cycles of matrix work.
    constexpr int NSTEP = TAPS * (KC / 2);
    float a[2][MT], bb[2][NT];
#define SPFE_LOAD_OPS(SET_, STEP_)                                                         \
  do {                                                                                      \
    constexpr int tap_ = (STEP_) / (KC / 2), t_ = (STEP_) % (KC / 2);                       \
    constexpr int dy_ = tap_ / KS, dx_ = tap_ % KS;                                         \
    _Pragma("unroll") for (int i = 0; i < MT; ++i)                                          \
        a[SET_][i] = aBase[(2 * t_) * PLANE + (i + dy_) * ROWP + dx_];                      \
    _Pragma("unroll") for (int j = 0; j < NT; ++j)                                          \
        bb[SET_][j] = bBase[(tap_ * KC + 2 * t_) * 64 + j * 32];                            \
  } while (0)
    SPFE_LOAD_OPS(0, 0);
    mfma_steps<0, NSTEP, KC, KS, MT, NT, PLANE, ROWP>(a, bb, acc, aBase, bBase);
#undef SPFE_LOAD_OPS
  }

  // ---- epilogue: bias, ReLU, optional 2x2 max-pool, NHWC store ----
  // C layout of the 32x32 MFMA: column (N) = lane&31, row (M) = (r&3)+8*(r>>2)+4*(lane>>5)
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int co = nb * 64 + (wn * NT + j) * 32 + l31;
    const float bias = p.bias[co];
    const bool cok = co < p.cout_real;
    if constexpr (!POOL) {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int y = ty0 + wm * MT + i;
        float *orow = p.out + ((size_t)b * H + y) * W * p.out_stride + p.out_choff + co;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int x = tx0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          float v = acc[i][j][r] + bias;
          if (RELU) v = v > 0.0f ? v : 0.0f;
          if (cok && y < H && x < W) orow[(size_t)x * p.out_stride] = v;
        }
      }
    } else {
      const int y = ty0 + wm * MT;
      const int Ho = H >> 1, Wo = W >> 1;
      float *orow = p.out + ((size_t)b * Ho + (y >> 1)) * Wo * p.out_stride + p.out_choff + co;
#pragma unroll
      for (int rp = 0; rp < 8; ++rp) {
        const int r = 2 * rp;
        const int x = tx0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        float v00 = acc[0][j][r] + bias, v01 = acc[0][j][r + 1] + bias;
        float v10 = acc[1][j][r] + bias, v11 = acc[1][j][r + 1] + bias;
        if (RELU) {
          v00 = v00 > 0.0f ? v00 : 0.0f;
          v01 = v01 > 0.0f ? v01 : 0.0f;
          v10 = v10 > 0.0f ? v10 : 0.0f;
          v11 = v11 > 0.0f ? v11 : 0.0f;
        }
        const float m0 = v00 > v01 ? v00 : v01;
        const float m1 = v10 > v11 ? v10 : v11;
        const float v = m0 > m1 ? m0 : m1;
        if (cok && y < H && x < W) orow[(size_t)(x >> 1) * p.out_stride] = v;
      }
    }
  }
}

template <int CIN, int KS, int KC, int WM, int WN, int MT, int NT, bool POOL, bool RELU>
static hipError_t launch_one(const ConvParams &p, hipStream_t s) {
  constexpr int TH = WM * MT;
  using G = Geo<KS, TH>;
  constexpr size_t lds = (size_t)(KC * G::PLANE + KS * KS * KC * 64) * sizeof(float);
  const int grid = p.nblk * p.tiles_x * p.tiles_y * p.B;
  auto k = conv_f32_kernel<CIN, KS, KC, WM, WN, MT, NT, POOL, RELU>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, s, p);
  return hipGetLastError();
}

int conv_kc(int ksize) { return ksize == 3 ? 16 : 64; }
int conv_tile_rows(bool small_tile) { return small_tile ? 4 : 8; }

hipError_t launch_conv_f32(const ConvParams &p, int cin, int ksize, bool pool, bool relu,
                           bool small_tile, hipStream_t s) {
#define SPFE_CONV(CIN_, KS_, KC_, WM_, WN_, MT_, NT_, POOL_, RELU_)                       \
  if (cin == CIN_ && ksize == KS_ && pool == POOL_ && relu == RELU_ &&                    \
      small_tile == (WM_ * MT_ == 4))                                                     \
    return launch_one<CIN_, KS_, KC_, WM_, WN_, MT_, NT_, POOL_, RELU_>(p, s);
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

hipError_t launch_conv1a(const uint8_t *img, const float *w9x64, const float *b64, float *out, int B,
                         int H, int W, hipStream_t s) {
  const int tiles_x = (W + 31) / 32, tiles_y = (H + 7) / 8;
  hipLaunchKernelGGL(conv1a_kernel, dim3(tiles_x * tiles_y * B), dim3(256), 0, s, img, w9x64, b64,
                     out, B, H, W, tiles_x, tiles_y);
  return hipGetLastError();
}

}  // namespace spfe
