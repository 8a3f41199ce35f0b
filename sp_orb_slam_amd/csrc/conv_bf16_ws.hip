// conv_bf16_ws.hip — wave-specialised bf16 3x3 convolution for the Cin = 64 layers of the bf16 mode
// (conv1b, conv2a, conv2b, conv3a of SPFrontend::forward, /root/reference/orb_slam2/src/cv/
// sp_extractor.cpp:82-88; BASELINE.json configs[3] "bf16 conv path").  Same arithmetic as
// conv_bf16.hip's kernel (v_mfma_f32_32x32x16_bf16, K order chunk(32 ch) -> tap -> 16 channels, f32
// accumulate / bias / ReLU / 2x2 max-pool, RNE to bf16), so the two are bit-identical; what changes is
// who does what:
//
//   * 512-thread workgroups, one per CU: waves 0-3 are CONSUMERS (one per SIMD: MFMAs, operand
//     fragment reads, the previous tile's epilogue in the MFMA shadows — nothing else), waves 4-7 are
//     PRODUCERS (one per SIMD: tile scheduling, address generation and the `buffer_load ... lds`
//     passes that bring the next halo tile HBM/L2 -> LDS).  An LDS-direct pass costs 60-180 issue
//     cycles in the stream that carries it (MI355X_MICROARCH.md, per-instruction constants) — two to
//     five MFMA slots — and its vmcnt(0) + barrier parked the single-role wave of conv_bf16.hip for
//     ~20 % of its cycles (profiles/r01g_pmc_bf16_conv1b.txt: SQ_WAIT_ANY).  In a separate wave that
//     cost lands beside the MFMA stream, not inside it.
//   * one stage per tile: a halo pixel keeps all 64 channels (128 B) in LDS, so a tile is ONE
//     barrier and 144 MFMAs per consumer wave (4608 matrix cycles) — the time an HBM round trip of
//     the next tile's loads gets to hide under, twice what a 32-channel stage offered.
//   * no padding: pixels and weight rows are 128 B, XOR-swizzled in 16-byte pieces
//     (piece j of halo column c sits in slot j ^ halo_swz(c), halo_swz(c) = ((c >> 1) & 7) ^ ((c & 1) << 2); the swizzle is applied by the
//     producer's choice of source address — free with LDS-direct loads — and by the packed weight
//     layout).  A halo row is 34 * 128 B = 17 * 256 B, so the bank slot of a fragment depends on the
//     column only: the 16 lanes of a ds_read_b128 group read 16 distinct columns mod 16 = 16 distinct
//     16-byte bank slots (conflict-free), and every fragment address is a per-kernel register
//     (3 dx x 4 k-groups) + immediate.  2 halo buffers (2 x 43,520 B) + the resident 64 x 576
//     weight block (73,728 B) = 160,768 B of the 160 KiB.
//   * dynamic tile queue: the producer fetches tile indices from a per-(XCD, channel-block) counter
//     two tiles ahead, so a workgroup that starts late or shares its CU's issue slots with the
//     side-stream kernels of the previous batch (SPFE_FLAG_ASYNC_COV) simply takes fewer tiles —
//     a static split made the slowest workgroup the kernel's duration.
//   * mfma(pixels, weights): a lane owns an output channel, so stores are 64-byte channel runs and the bias is
//     one register per accumulator tile.
#include <utility>

#include "conv1a_mfma.h"
#include "spfe_kernels.h"

namespace spfe {
namespace ws {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(3))) bf16x8 lds_frag;

constexpr unsigned OOB = 0x80000000u;
constexpr int TH = 8, ROWS = TH + 2, COLS = 34;
constexpr int ROW_BYTES = COLS * 128;                  // 4352 = 17 * 256
constexpr int HALO_BYTES = ROWS * ROW_BYTES;           // 43,520
constexpr int HALO_PIECES = HALO_BYTES / 16;           // 2720
constexpr int HALO_INSTR = (HALO_PIECES + 63) / 64;    // 43 wave-level LDS-direct passes
constexpr int HALO_IT = (HALO_INSTR + 3) / 4;          // 11 per producer wave
constexpr int W_BYTES = 9 * 64 * 128;                  // 73,728: [tap][cout][64 cin], swizzled
[[maybe_unused]] constexpr int W_INSTR = W_BYTES / 1024;                // 72
constexpr int LDS_W = 2 * HALO_BYTES;                  // 87,040
constexpr int LDS_SLOT = LDS_W + W_BYTES;              // 160,768: NSLOT tile descriptors of 16 B
constexpr int NSLOT = 4;                               // descriptors run three tiles ahead of the tile being multiplied (see the producers)
constexpr int LDS_PATCH = LDS_SLOT + 64;               // TAG 2: per producer wave a 4 x 40 bf16 patch of the frame (320 B)
constexpr int LDS_TOTAL = LDS_PATCH + 4 * 320;         // 162,112 of the 163,840 bytes
constexpr int NSTEP = 36;                              // K steps per tile: 2 chunks x 9 taps x 2

#ifdef WS_PROBE_TIMING
// probe builds (tools/microbench/conv_ws_probe.hip): cycle counters summed over workgroups
// [0] consumer wave 0 loop cycles  [1] ... of which at the end-of-tile barrier  [2] tiles
// [4] producer wave 4 loop cycles  [5] ... issuing passes  [6] ... waiting vmcnt(0)  [7] ... at the barrier
__device__ unsigned long long ws_dbg[10];
#define WS_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define WS_ACC(i, v) do { if (lane == 0) atomicAdd(&ws_dbg[i], (unsigned long long)(v)); } while (0)
#else
#define WS_T(var)
#define WS_ACC(i, v)
#endif
#ifndef WS_PRIO
#define WS_PRIO 1
#endif
#ifndef WS_EPI_MICRO
#define WS_EPI_MICRO -1  // probe builds: force the epilogue form (1 = single instructions in every gap, 0 = one item per burst)
#endif
#ifndef WS_READ_SPREAD
#define WS_READ_SPREAD -1
#endif
#ifndef WS_EPI_PKRELU
#define WS_EPI_PKRELU 1  // pool, burst form: the epilogue's ReLU on the rounded channel pair (one v_pk_max_i16) instead of one v_max_f32 per
                         // value (round 6: +0.3 %; an item split into two half bursts over two K steps measured +-0 and is gone)
#endif
#ifndef WS_EPI_GAP
#define WS_EPI_GAP 2     // the gap (0..3) of a K step that carries the epilogue item
#endif
#ifndef WS_C1A_STORE
#define WS_C1A_STORE 1   // TAG 2, how a producer lane stores its four 16-byte pieces of a halo pixel: 1 = ds_write_b128, 2 = two conflict-free ds_write_b64 (see make_halo)
#endif
#ifndef WS_CLAIM
#define WS_CLAIM 1       // TAG 2: tiles a workgroup takes from its queue per atomic, asked for one tile ahead of their use
#endif
#ifndef WS_ABLATE
#define WS_ABLATE 0  // probe builds: 1 = producers issue no halo passes in the loop, 2 = consumers issue no MFMAs,
                     // 3 = every tile loads the same halo (cache hits only), 4 = no fragment reads, 5 = no stores
#endif

// s_barrier without the vmcnt(0) the fence of a full workgroup sync would add: consumers must not wait
// for their epilogue stores, producers wait explicitly for what the barrier publishes
__device__ __forceinline__ void wg_barrier() { asm volatile("s_barrier" ::: "memory"); }

__device__ __forceinline__ bf16x8 lds_read(lds_char *p, int imm) {
  return *reinterpret_cast<lds_frag *>(p + imm);
}

// Slot of logical 16-byte piece j of halo column c inside the pixel's 128 bytes: j ^ halo_swz(c).  (c >> 1) & 7 is what makes
// the consumers' ds_read_b128 conflict-free (the 16 lanes of a read group hold 8 even and 8 odd columns; a column pair shares
// (c >> 1) and sits 128 B apart: the two parities use the two halves of the 64 read banks, and within a parity the 8 values
// are distinct for every tap offset).  Round 6 adds the (c & 1) << 2 term, which keeps that property — within a parity it is
// a constant xor — and makes the PRODUCERS' stores of whole pieces (TAG 2: ds_write_b128, 8 consecutive columns per store
// group, 32 store banks = 128 B) hit 8 distinct slots instead of 4 slots twice: the fused conv1a's halo stores were 2-way
// bank conflicts, 19 % of this kernel's LDS cycles in round 5 (profiles/r05_pmc_bf16_720p.txt).
__device__ __forceinline__ constexpr int halo_swz(int c) { return ((c >> 1) & 7) ^ ((c & 1) << 2); }

// left edge of tile column tx: the last column of a ragged row ends at the image edge instead of overhanging it
__device__ __forceinline__ int tile_x0(int tx, int W) {
  const int x0 = tx * 32;
  return x0 + 32 > W ? W - 32 : x0;
}

struct Epi {
  __amdgpu_buffer_rsrc_t rout;
  unsigned rowoff[2];  // byte offset of (row i, column x0 + 4 hi, this lane's channel) in the output frame (pool: [0]), or OOB
  unsigned pitch;      // bytes per output pixel
};
// Accumulator layout after mfma(pixels, weights): lane = (output channel, hi = lane >> 5), register r <-> pixel
// column 8*(r>>2) + 4*hi + (r&3) of the wave's row i.  The packed weights put the block's EVEN channels in accumulator
// tile j = 0 and the ODD ones in tile j = 1 (row m of tile j <-> channel 2 m + j), so a lane holds the adjacent channels
// 2 l31, 2 l31 + 1 of every pixel it has: one v_cvt_pk_bf16_f32 packs them and one dword store per register writes, for
// the 32 lanes of each half, the 64 consecutive bf16 channels (128 B) of ONE pixel — half the conversions and half the
// store instructions of a channel-per-tile layout (2-byte stores), which is what bounds the layers without a pool.
// (The transposed form of conv_bf16.hip — a lane owns a pixel — scatters 8-byte pieces over 32-64 cache lines per store.)
// Bias (one value per lane and accumulator tile), ReLU, the 2x2 max (max(a + b, c + b) == max(a, c) + b exactly:
// rounding is monotonic), RNE to bf16: bit-identical to conv_bf16.hip's epilogue.
// Item G: pool: 8 items (g = G / 2, h2 = G % 2: one pooled pixel, both channels); no pool: 16 items (i = G / 8,
// g = (G / 2) % 4, two of the four registers 4g..4g+3 each).
// No column predicates: when the width is not a multiple of 32, the last tile of a row is shifted left to end at
// the image edge (x0 = W - 32) and recomputes a few columns of its neighbour — identical values, written twice.
// max without the canonicalising v_max(x, x) that fmaxf's sNaN rule puts in front of every MFMA result (the
// instruction itself: v_max_f32 returns the non-NaN operand, as fmaxf does)
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
// RNE, then the ReLU on the rounded pair as 16-bit integers (max(bf16(v), +0) == bf16(max(v, 0)) bit for bit: conv1a_mfma.h)
typedef short i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2_relu(float v0, float v1) {
  const i16x2 r = __builtin_bit_cast(i16x2, __builtin_convertvector((f32x2){v0, v1}, bf16x2));
  const i16x2 z = {0, 0};
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(r, z));
}

template <bool POOL, int G>
__device__ __forceinline__ void epi_item_c(const Epi &e, const float (&bias)[2], const f32x16 (&acc)[2][2]) {
  constexpr int NITEM = POOL ? 8 : 16;
  if constexpr (G >= 0 && G < NITEM) {
    if constexpr (POOL) {
      constexpr int g = G / 2, h2 = G % 2, r = 4 * g + 2 * h2;
      float v[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        v[j] = max3_nc(acc[0][j][r], acc[0][j][r + 1], max_nc(acc[1][j][r], acc[1][j][r + 1]));
        v[j] = WS_EPI_PKRELU ? v[j] + bias[j] : relu_nc(v[j] + bias[j]);
      }
      __builtin_amdgcn_raw_buffer_store_b32(WS_EPI_PKRELU ? pack2_relu(v[0], v[1]) : pack2(v[0], v[1]), e.rout, e.rowoff[0], (unsigned)(4 * g + h2) * e.pitch, 0);
    } else {
      constexpr int i = G / 8, g = (G / 2) % 4, mh = G % 2;
#pragma unroll
      for (int mm = 0; mm < 2; ++mm) {
        const int m = 2 * mh + mm;
        const float v0 = relu_nc(acc[i][0][4 * g + m] + bias[0]), v1 = relu_nc(acc[i][1][4 * g + m] + bias[1]);
        __builtin_amdgcn_raw_buffer_store_b32(pack2(v0, v1), e.rout, e.rowoff[i], (unsigned)(8 * g + m) * e.pitch, 0);
      }
    }
  }
}

struct TileDesc {
  int b, ty, tx, valid;
};

// the consumer's per-tile side work that needs runtime state, placed at fixed K steps (see k_steps)
template <typename Aim>
struct TileHooks {
  Aim &aim;
  lds_char *lds;
  TileDesc dc, dn;     // this tile, the next one
  Epi *eMine;
  i32x4 raw;
  int t;
  unsigned long long bar;
  template <int S, int M>
  __device__ __forceinline__ void at() {
    if constexpr (S == 3 && M == 1) aim(dc, *eMine);
    if constexpr (S == 24 && M == 1)
      raw = *reinterpret_cast<const __attribute__((address_space(3))) i32x4 *>(lds + LDS_SLOT + ((t + 1) % NSLOT) * 16);
    if constexpr (S == 28 && M == 1) {
      dn.b = __builtin_amdgcn_readfirstlane(raw.x);
      dn.ty = __builtin_amdgcn_readfirstlane(raw.y);
      dn.tx = __builtin_amdgcn_readfirstlane(raw.z);
      dn.valid = __builtin_amdgcn_readfirstlane(raw.w);
    }
  }
};

// The previous tile's epilogue as single instructions in the MFMA gaps of K steps 2..33 instead of bursts: a burst
// delays the MFMA behind it by its whole issue time.  Same operations on the same values as epi_item_c, which still
// flushes the last tile.
// pool: pooled pixel vI = 0..7 (g = vI / 2, h2 = vI % 2) takes the 16 gaps of steps 2 + 4 vI .. 5 + 4 vI (10 used).
// no pool: register u = 0..31 (i = u / 16, g = (u / 4) % 4, m = u % 4) takes the 4 gaps of step 2 + u.
struct EpiTmp {
  float v[2], t[2];
  unsigned pk;
};
template <bool POOL, int S, int M>
__device__ __forceinline__ void epi_micro(const Epi &e, const float (&bias)[2], const f32x16 (&acc)[2][2], EpiTmp &t) {
  if constexpr (WS_ABLATE != 5 && S >= 2 && S < 34) {
    if constexpr (POOL) {
      constexpr int vI = (S - 2) / 4, k = ((S - 2) % 4) * 4 + M;
      constexpr int g = vI / 2, h2 = vI % 2, r = 4 * g + 2 * h2;
      if constexpr (k == 0) t.t[0] = max_nc(acc[1][0][r], acc[1][0][r + 1]);
      if constexpr (k == 1) t.t[1] = max_nc(acc[1][1][r], acc[1][1][r + 1]);
      if constexpr (k == 2) t.v[0] = max3_nc(acc[0][0][r], acc[0][0][r + 1], t.t[0]);
      if constexpr (k == 3) t.v[1] = max3_nc(acc[0][1][r], acc[0][1][r + 1], t.t[1]);
      if constexpr (k == 4) t.v[0] = t.v[0] + bias[0];
      if constexpr (k == 5) t.v[1] = t.v[1] + bias[1];
      if constexpr (k == 6) t.v[0] = relu_nc(t.v[0]);
      if constexpr (k == 7) t.v[1] = relu_nc(t.v[1]);
      if constexpr (k == 8) t.pk = pack2(t.v[0], t.v[1]);
      if constexpr (k == 9) __builtin_amdgcn_raw_buffer_store_b32(t.pk, e.rout, e.rowoff[0], (unsigned)(4 * g + h2) * e.pitch, 0);
    } else {
      constexpr int u = S - 2, i = u / 16, g = (u / 4) % 4, m = u % 4;
      if constexpr (M == 0) { t.v[0] = acc[i][0][4 * g + m] + bias[0]; t.v[1] = acc[i][1][4 * g + m] + bias[1]; }
      if constexpr (M == 1) { t.v[0] = relu_nc(t.v[0]); t.v[1] = relu_nc(t.v[1]); }
      if constexpr (M == 2) t.pk = pack2(t.v[0], t.v[1]);
      if constexpr (M == 3) __builtin_amdgcn_raw_buffer_store_b32(t.pk, e.rout, e.rowoff[i], (unsigned)(8 * g + m) * e.pitch, 0);
    }
  }
}

// One consumer K step: MT x NT = 2 x 2 MFMAs; in their shadows the operand fragments of step S + 2 (ring of three) and
// the previous tile's epilogue.  The ring runs THROUGH the tile boundary: steps 34 and 35 prefetch steps 0 and 1 of the
// next tile from the other halo buffer, so the matrix pipe never waits for a tile's first fragments (the ~400-cycle head
// of every tile before).  That moves the end-of-tile barrier to the start of step 34: by then this wave has issued — and
// waits for — its last reads of this tile's buffer (step 35's, issued at step 33), which is all the barrier has to say to the
// producers ("the buffer is free") and all it has to hear from them ("the next halo is in LDS").
// hooks.at<S, M>() — the caller's side work that needs runtime state (next tile's descriptor, epilogue geometry).
// K order (both bf16 kernels, so that they stay bit-identical): 32-channel chunk -> dx -> 16-channel group -> dy, i.e. the three
// VERTICAL taps of a column are consecutive steps.  Step (group g = (chunk, dx, k-group), dy) multiplies halo rows i + dy
// (i = 0, 1: the wave's two output rows) with the weights of tap (dy, dx): the four halo rows of a group are read ONCE
// (2 + 1 + 1 fragments over its three steps) instead of 2 per step — 120 LDS fragment reads per tile instead of 144.  The
// LDS operand traffic is what costs this kernel its clock.
struct KStep {
  int chunk, dx, k2, dy, kk, tap, hiw, g;
};
__device__ constexpr KStep kstep_of(int s) {
  const int chunk = s / 18, r = s % 18, dx = r / 6, k2 = (r % 6) / 3, dy = r % 3;
  return KStep{chunk, dx, k2, dy, chunk * 2 + k2, dy * 3 + dx, dy * 3 + dx >= 7 ? 1 : 0, s / 3};
}
// the fragments step T (0..35 of this tile; 36, 37 = steps 0, 1 of the next one, other halo buffer) needs and no earlier
// step of its group has read: halo rows 0, 1 at dy = 0, row dy + 1 after that; the step's two weight fragments
template <int T, int BUF, int PART>
__device__ __forceinline__ void read_frags(bf16x8 (&ah)[2][4], bf16x8 (&w)[3][2], lds_char *const (&aptr)[3][4],
                                           lds_char *const (&wptr)[2][4]) {
  constexpr KStep k = kstep_of(T % NSTEP);
  constexpr int buf = T < NSTEP ? BUF : BUF ^ 1, gp = (T / 3) % 2, ws = T % 3;
  // PART: 0..3 = one fragment each (spread over the gaps), -1 = all of them
  if constexpr (PART == 0 || PART < 0) {
    if constexpr (k.dy == 0) ah[gp][0] = lds_read(aptr[k.dx][k.kk], buf * HALO_BYTES + 0 * ROW_BYTES);
    else ah[gp][k.dy + 1] = lds_read(aptr[k.dx][k.kk], buf * HALO_BYTES + (k.dy + 1) * ROW_BYTES);
  }
  if constexpr (PART == 1 || PART < 0) w[ws][0] = lds_read(wptr[k.hiw][k.kk], (k.tap - 7 * k.hiw) * 8192 + 0 * 4096);
  if constexpr (PART == 2 || PART < 0) w[ws][1] = lds_read(wptr[k.hiw][k.kk], (k.tap - 7 * k.hiw) * 8192 + 1 * 4096);
  if constexpr (PART == 3 || PART < 0) {
    if constexpr (k.dy == 0) ah[gp][1] = lds_read(aptr[k.dx][k.kk], buf * HALO_BYTES + 1 * ROW_BYTES);
  }
}

template <int S, bool POOL, int BUF, bool MICRO, typename Hooks>
__device__ __forceinline__ void k_steps(bf16x8 (&ah)[2][4], bf16x8 (&w)[3][2], f32x16 (&acc)[2][2],
                                        const f32x16 (&accPrev)[2][2], const float (&bias)[2],
                                        lds_char *const (&aptr)[3][4], lds_char *const (&wptr)[2][4],
                                        const Epi &ePrev, EpiTmp &et, Hooks &hooks) {
  if constexpr (S < NSTEP) {
    constexpr KStep ks = kstep_of(S);
    constexpr int gp = ks.g % 2, cur = S % 3;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (m == 0) {
        if constexpr (S == NSTEP - 2) {
          __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's last reads of the tile's halo buffer are back
          wg_barrier();                        // end of tile: the buffer is free, the next halo has landed
        }
      }
      if constexpr (WS_ABLATE != 4) {
        // the operand fragments of step S + 2: one per gap where the epilogue is spread too (+1 %), else all behind the
        // first MFMA (probe builds: WS_READ_SPREAD forces either)
        // (round 6: one fragment read per gap in the burst form too — with the producers at ~270 VALU instructions a tile the
        // gaps have room again: conv1b with conv1a inside +1.0 %)
        constexpr bool SPREAD = WS_READ_SPREAD >= 0 ? WS_READ_SPREAD != 0 : true;
        if constexpr (SPREAD) {
          if (m == 0) read_frags<S + 2, BUF, 0>(ah, w, aptr, wptr);
          if (m == 1) read_frags<S + 2, BUF, 1>(ah, w, aptr, wptr);
          if (m == 2) read_frags<S + 2, BUF, 2>(ah, w, aptr, wptr);
          if (m == 3) read_frags<S + 2, BUF, 3>(ah, w, aptr, wptr);
        } else {
          if (m == 0) read_frags<S + 2, BUF, -1>(ah, w, aptr, wptr);
        }
      }
      if constexpr (MICRO) {
      if (m == 0) { epi_micro<POOL, S, 0>(ePrev, bias, accPrev, et); hooks.template at<S, 0>(); }
      if (m == 1) { epi_micro<POOL, S, 1>(ePrev, bias, accPrev, et); hooks.template at<S, 1>(); }
      if (m == 2) { epi_micro<POOL, S, 2>(ePrev, bias, accPrev, et); hooks.template at<S, 2>(); }
      if (m == 3) { epi_micro<POOL, S, 3>(ePrev, bias, accPrev, et); hooks.template at<S, 3>(); }
      } else {
      // one epilogue item (12-16 instructions) in one gap every second / fourth step
      if (m == WS_EPI_GAP) {
        if constexpr (POOL) {  // 8 items: one every fourth step
          if constexpr (S >= 2 && S % 4 == 2 && WS_ABLATE != 5) epi_item_c<POOL, (S - 2) / 4>(ePrev, bias, accPrev);
        } else {               // 16 items: one every second step
          if constexpr (S >= 2 && S % 2 == 0 && WS_ABLATE != 5) epi_item_c<POOL, (S - 2) / 2>(ePrev, bias, accPrev);
        }
        hooks.template at<S, 1>();
      }
      }
      __builtin_amdgcn_sched_barrier(0);
      {
        const int i = m / 2, j = m % 2;
        if constexpr (S == 0) {
          f32x16 z;
#pragma unroll
          for (int r = 0; r < 16; ++r) z[r] = 0.0f;
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[gp][i + ks.dy], w[cur][j], z, 0, 0, 0);
        } else {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[gp][i + ks.dy], w[cur][j], acc[i][j], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    k_steps<S + 1, POOL, BUF, MICRO>(ah, w, acc, accPrev, bias, aptr, wptr, ePrev, et, hooks);
  }
}


// in: NHWC bf16 [B][H][W][in_stride]; wpack: [nblk][tap 9][row 64][8 pieces, piece g at slot g ^ ((row >> 1) & 7)][8 bf16],
// row 32 j + m <-> output channel 2 m + j of the block (even channels first: see the epilogue);
// out: NHWC bf16.  p.tile_ctr: nblk * 8 counters, zeroed before the launch.
// TAG 1 only names the instantiation (conv1b, the dominant kernel): profiler rows of conv1b and conv2b — same
// template arguments otherwise, same grid — stay apart.
// TAG 2 = conv1b with conv1a INSIDE (sp_extractor.cpp:81-82 as one kernel): the producer waves do not load the
// halo tile, they compute it — conv1a of the u8 frame (p.img; p.w1a = the bf16 operand table, p.b1a = the bias) as
// 2 MFMAs per 32 pixels, the arithmetic of conv1a_mfma.h that the stand-alone conv1a_bf16_kernel shares (bit-identical
// activations) — and write it into the swizzled LDS layout the consumers read.  The 64-channel activation of the
// first layer (118 MB per 1280x720 frame) never exists in HBM: conv1a's launch, its writes and conv1b's reads
// of them are gone; what remains of the first two layers' HBM traffic is the u8 frame in and conv1b's pooled
// output.  (As VALU code — 9 taps x 64 channels of f32 FMAs per pixel — the producers needed ~950 instructions per tile
// and wave, more issue slots than the consumers' MFMA stream leaves: 12.3 k cycles per tile instead of 5.9 k.)
template <bool POOL, int TAG>
__global__ __launch_bounds__(512) void conv_bf16_ws_kernel(ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem_ws[];
  lds_char *const lds = (lds_char *)smem_ws;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = blockIdx.x & 7;
  const int nb = (int)(blockIdx.x >> 3) % p.nblk;
  const int gi = (int)(blockIdx.x >> 3) / p.nblk, gsize = (int)(gridDim.x >> 3) / p.nblk;  // this workgroup in its queue's group
  const int per_nb = p.tiles_x * p.tiles_y * p.B;
  const int t_lo = (int)((long)per_nb * xcd / 8), t_cnt = (int)((long)per_nb * (xcd + 1) / 8) - t_lo;
  const int Ho = POOL ? p.H >> 1 : p.H, Wo = POOL ? p.W >> 1 : p.W;
  const unsigned in_pix_bytes = (unsigned)p.in_stride * 2u;
  [[maybe_unused]] const unsigned frame_in_bytes = (unsigned)p.H * p.W * in_pix_bytes;
  const unsigned out_pix_bytes = (unsigned)p.out_stride * 2u;
  const unsigned frame_out_bytes = (unsigned)Ho * Wo * out_pix_bytes;

  auto read_slot = [&](int k) -> TileDesc {
    const i32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) i32x4 *>(lds + LDS_SLOT + k * 16);
    TileDesc d;
    d.b = __builtin_amdgcn_readfirstlane(v.x);
    d.ty = __builtin_amdgcn_readfirstlane(v.y);
    d.tx = __builtin_amdgcn_readfirstlane(v.z);
    d.valid = __builtin_amdgcn_readfirstlane(v.w);
    return d;
  };

  if (wave >= 4) {
    // ------------------------------------------------------------------ producers
    const int pw = wave - 4;
    int *ctr = p.tile_ctr + nb * 8 + xcd;
    // The queue: one counter per (XCD, channel block), shared by the ~32 workgroups of that group; a workgroup takes WS_CLAIM
    // consecutive tiles per atomic.  An atomic's round trip is ~2.5 us under load — as long as a tile: the probe's timing build
    // (round 6) showed a tile of the fused kernel taking 2.86 us with the consumers issuing no MFMA at all, the queue wave
    // waiting for its answer; with the request ONE TILE AHEAD of its use (TAG 2, below) the same build makes a tile in 1.6 us.
    // Layers whose halo comes by LDS-direct loads (TAG 0 / 1) ask per tile, behind their loads: the answer comes back under
    // the loads' own wait, and they are memory-bound — larger claims measured -2 % there (coarser balance).
    int q_base = 0, q_left = 0;
    [[maybe_unused]] int raw_pend = 0;
    [[maybe_unused]] bool pend = false;
    auto queue_index = [&](int k) -> int {   // tile number k (>= NSLOT - 1) of this workgroup -> the tile, or -1   (WS_ABLATE 6: no queue)
      const int v = k * gsize + gi;
      return v < t_cnt ? t_lo + v : -1;
    };
    (void)queue_index;
    auto fetch = [&]() -> int {  // next tile of this (XCD, block) queue, or -1   (TAG 0 / 1: one atomic per tile)
      int v = 0;
      if (lane == 0) v = atomicAdd(ctr, 1);
      v = __builtin_amdgcn_readfirstlane(v) + (NSLOT - 1) * gsize;   // the first three tiles of every workgroup are pre-assigned
      return v < t_cnt ? t_lo + v : -1;
    };
    auto publish = [&](int k, int tile) {  // decode + write descriptor k (wave 4 only)
      i32x4 d = {0, 0, 0, 0};
      if (tile >= 0) {
        int q = tile;
        d.z = q % p.tiles_x; q /= p.tiles_x;
        d.y = q % p.tiles_y; d.x = q / p.tiles_y;
        d.w = 1;
      }
      if (lane == 0) *reinterpret_cast<__attribute__((address_space(3))) i32x4 *>(lds + LDS_SLOT + k * 16) = d;
    };
    // per-lane geometry of this wave's halo pieces: pass `it` covers LDS pieces (it * 4 + pw) * 64 + lane
    int prow[HALO_IT], pcol[HALO_IT];
    unsigned pj16[HALO_IT];
#pragma unroll
    for (int it = 0; it < HALO_IT; ++it) {
      const int q = (it * 4 + pw) * 64 + lane;
      const int r = q / (COLS * 8), rem = q % (COLS * 8), c = rem >> 3, slot = rem & 7;
      prow[it] = q < HALO_PIECES ? r - 1 : (1 << 20);
      pcol[it] = c - 1;
      pj16[it] = (unsigned)(slot ^ halo_swz(c)) * 16u;
    }
    auto load_halo = [&](const TileDesc &d, int buf) {
#if defined(__HIP_DEVICE_COMPILE__)
      const char *base = reinterpret_cast<const char *>(p.in) + ((size_t)(WS_ABLATE == 3 ? 0 : d.b) * p.H * p.W * p.in_stride + p.in_choff) * 2;
      const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, frame_in_bytes, 0x00020000);
      const int y0 = WS_ABLATE == 3 ? 8 : d.ty * TH, x0 = WS_ABLATE == 3 ? 32 : tile_x0(d.tx, p.W);
#pragma unroll
      for (int it = 0; it < HALO_IT; ++it) {
        const int k = it * 4 + pw;
        if (k < HALO_INSTR) {
          const int gy = y0 + prow[it], gx = x0 + pcol[it];
          const unsigned voff = ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
                                    ? (unsigned)(gy * p.W + gx) * in_pix_bytes + pj16[it]
                                    : OOB;
          if (k * 64 + lane < HALO_PIECES)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_void *)(lds + buf * HALO_BYTES + k * 1024), 16, voff, 0, 0, 0);
        }
      }
#endif
    };

    // ---- TAG 2: conv1a computed here, as a K = 16 matrix product (conv1a_mfma.h).  The 340 halo pixels are 11 groups
    // of 32 (linear index P = 34 r + c); this wave makes groups pw, pw + 4, pw + 8.  A group's u8 pixels (4 rows x 40
    // columns: it can straddle two halo rows) go through a wave-private bf16 patch in LDS, one group at a time.
    constexpr int NGRP = (ROWS * COLS + 31) / 32;   // 11
    [[maybe_unused]] bf16x8 wA[2];
    [[maybe_unused]] float bias1[2][16];
    [[maybe_unused]] unsigned tap0[3], hdst[3][4], pdst[3];
    [[maybe_unused]] bool podd[3];
    [[maybe_unused]] int hrc[3], prel[3];
    if constexpr (TAG == 2) {
      c1a::load_constants(p.w1a, p.b1a, lane, wA, bias1);
      const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
      for (int gi = 0; gi < 3; ++gi) {
        const int g = pw + 4 * gi, P = 32 * g + l31, r = P / COLS, c = P % COLS, r0 = (32 * g) / COLS;
        tap0[gi] = (unsigned)(((r - r0) * c1a::PATCH_PITCH + c + 2) * 2);   // patch column 0 <-> image column x0 - 4
        hrc[gi] = (g < NGRP && P < ROWS * COLS) ? (int)(((unsigned)(r - 1) << 16) | ((unsigned)(c - 1) & 0xffffu)) : (int)0x80000000;
        // the lane's four whole pieces of its pixel's 128-byte row (conv1a_mfma.h): piece 4 j + 2 rr + hi, at its swizzled slot,
        // one ds_write_b128 each (WS_C1A_STORE 1, the default: half the store instructions of the 8-byte form and no selects).
        // WS_C1A_STORE 2 = the bank-conflict-free form: two ds_write_b64 per piece — FIRST the low 8 bytes from the even pixels
        // and the high 8 bytes from the odd ones, then the other halves (pixels c and c + 1 share a slot and sit 128 B apart, the
        // same banks of a 32-bank store; with different halves the 16 lanes of a store group cover all 32 banks once).  Measured
        // (round 6, 1280x720 x 8): SQ_LDS_BANK_CONFLICT 19 % -> 4.8 % of the LDS cycles, LDS-active cycles -16 %, and the kernel
        // 1.6 % SLOWER (392.4 -> 398.7 us): its 16 selects + 4 address flips per group cost the producers' issue slots more than
        // the conflicts cost the LDS.  hdst = the address of the (first) store.
        podd[gi] = (c & 1) != 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          hdst[gi][k] = (unsigned)(P * 128 + (((2 * k + hi) ^ halo_swz(c)) * 16) + (WS_C1A_STORE == 2 ? 8 * (c & 1) : 0));
        // the group's patch: 4 rows x 40 bytes starting at image column x0 - 4 (4-byte aligned; widths are multiples of
        // 8, so a dword is entirely inside the frame or entirely outside): lane < 40 loads one dword = 4 pixels
        const int prow = lane / 10, pdw = lane % 10;
        pdst[gi] = (unsigned)((prow * c1a::PATCH_PITCH + 4 * pdw) * 2);
        prel[gi] = (g < NGRP && lane < 40) ? (int)(((unsigned)(r0 - 2 + prow) << 16) | ((unsigned)(4 * pdw - 4) & 0xffffu)) : (int)0x80000000;
      }
    }
    // The u8 pixels a tile's groups need (one dword = 4 pixels per lane and group): issued ONE TILE AHEAD of their use — the
    // probe's timing build had shown the producers, not the MFMA stream, to be this kernel's critical path (round 6: 4,800
    // cycles of a 6,170-cycle tile in make_halo at 1.66 GHz, and 2.86 us a tile even with the consumers issuing no MFMA at
    // all): every tile began with this load's round trip — ~1 us from L2 / HBM — in front of the matrix products it feeds.
    auto load_patch = [&](const TileDesc &d, unsigned (&px)[3]) {
      // the frame as a buffer: out-of-range offsets read 0 — conv1a's own zero padding — and nothing is conditional,
      // so the tile's patch bytes leave as ONE burst of loads
      const __amdgpu_buffer_rsrc_t rimg = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<uint8_t *>(p.img) + (size_t)d.b * p.H * p.W, 0, (unsigned)(p.H * p.W), 0x00020000);
      const int y0 = d.ty * TH, x0 = tile_x0(d.tx, p.W);
#pragma unroll
      for (int gi = 0; gi < 3; ++gi) {
        const int gy = y0 + (prel[gi] >> 16), gx = x0 + (int)(short)(prel[gi] & 0xffff);
        const bool in = d.valid && prel[gi] != (int)0x80000000 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        px[gi] = __builtin_amdgcn_raw_buffer_load_b32(rimg, in ? (unsigned)(gy * p.W + gx) : OOB, 0, 0);
      }
    };
    auto make_halo = [&](const TileDesc &d, int buf, const unsigned (&px)[3]) {
      const int y0 = d.ty * TH, x0 = tile_x0(d.tx, p.W);
      lds_char *patch = lds + LDS_PATCH + pw * 320;
      const int hi = lane >> 5;
      // border tiles only: some halo pixel lies outside the frame (conv1b's zero padding)
      const bool border = y0 == 0 || x0 == 0 || y0 + TH + 1 > p.H || x0 + 33 > p.W;
#pragma unroll
      for (int gi = 0; gi < 3; ++gi) {
        if (pw + 4 * gi >= NGRP) break;   // (wave-uniform)
        if (prel[gi] != (int)0x80000000) {
          u32x2 pb;   // 4 pixels -> 4 bf16
          pb.x = c1a::u8_to_bf16(px[gi] & 0xffu) | ((unsigned)c1a::u8_to_bf16((px[gi] >> 8) & 0xffu) << 16);
          pb.y = c1a::u8_to_bf16((px[gi] >> 16) & 0xffu) | ((unsigned)c1a::u8_to_bf16(px[gi] >> 24) << 16);
          *reinterpret_cast<__attribute__((address_space(3))) u32x2 *>(patch + pdst[gi]) = pb;
        }
        const bf16x8 pxop = c1a::pixel_operand(reinterpret_cast<c1a::lds_u16 *>(patch + tap0[gi]), hi);
        f32x16 acc1[2];
        c1a::product(wA, pxop, bias1, acc1);
        if (hrc[gi] != (int)0x80000000) {
          [[maybe_unused]] const bool odd = podd[gi];
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
              const u32x4 v = c1a::finish8(acc1[j], rr);
              const unsigned a = hdst[gi][2 * j + rr];
              if constexpr (WS_C1A_STORE == 2) {
                *reinterpret_cast<__attribute__((address_space(3))) u32x2 *>(lds + buf * HALO_BYTES + a) =
                    odd ? (u32x2){v.z, v.w} : (u32x2){v.x, v.y};
                *reinterpret_cast<__attribute__((address_space(3))) u32x2 *>(lds + buf * HALO_BYTES + (a ^ 8u)) =
                    odd ? (u32x2){v.x, v.y} : (u32x2){v.z, v.w};
              } else {
                *reinterpret_cast<__attribute__((address_space(3))) u32x4 *>(lds + buf * HALO_BYTES + a) = v;
              }
            }
          if (border) {   // (wave-uniform branch: interior tiles carry nothing of this)
            // conv1b's own zero padding: halo pixels outside the frame are zeros, not conv1a evaluated out there — written over
            // what the lane has just stored (LDS operations of a wave are in order)
            const int hy = y0 + (hrc[gi] >> 16), hx = x0 + (int)(short)(hrc[gi] & 0xffff);
            if (!((unsigned)hy < (unsigned)p.H && (unsigned)hx < (unsigned)p.W)) {
#pragma unroll
              for (int k = 0; k < 4; ++k)
                *reinterpret_cast<__attribute__((address_space(3))) u32x4 *>(lds + buf * HALO_BYTES + (hdst[gi][k] & ~8u)) = (u32x4){0u, 0u, 0u, 0u};
            }
          }
        }
      }
    };
    // The resident weight block first: nothing it needs has to be fetched or decided.
    {
#if defined(__HIP_DEVICE_COMPILE__)
      const char *wb = reinterpret_cast<const char *>(p.wpack) + (size_t)nb * W_BYTES;
      const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(wb), 0, (unsigned)W_BYTES, 0x00020000);
      if (gi < t_cnt) {
#pragma unroll
        for (int it = 0; it < W_INSTR / 4; ++it) {
          const int k = it * 4 + pw;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void *)(lds + LDS_W + k * 1024), 16, (unsigned)(k * 1024 + lane * 16), 0, 0, 0);
        }
      }
#endif
    }
    // the first three tiles of every workgroup are fixed (its index in the queue's group, + the group size, + twice that): no
    // atomic round trips before the first loads; the queue hands out the tiles after those, THREE tiles ahead of the tile the
    // consumers multiply (a ring of NSLOT descriptors): tile t + 1's halo is made while tile t is multiplied, tile t + 2's u8
    // pixels are in flight meanwhile (TAG 2), and the queue's answer for tile t + 3 has that long to come back.
    // the wave that keeps the tile queue: with conv1a inside (TAG 2), wave 7 — it makes two groups of halo pixels per tile,
    // the others three, so the atomic's round trip (1-2 us under load) hides in its slack instead of adding to the
    // longest producer
    constexpr int QW = TAG == 2 ? 3 : 0;
    if (pw == QW) {
#pragma unroll
      for (int k = 0; k < NSLOT - 1; ++k) publish(k, gi + k * gsize < t_cnt ? t_lo + gi + k * gsize : -1);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    wg_barrier();                        // barrier #0: the first descriptors are published
    TileDesc cur = read_slot(0);
    [[maybe_unused]] unsigned pxA[3] = {0u, 0u, 0u}, pxB[3] = {0u, 0u, 0u};
    if constexpr (TAG == 2) {
      load_patch(cur, pxB);
      load_patch(read_slot(1), pxA);     // (an invalid descriptor loads zeros nobody uses)
      if (cur.valid) make_halo(cur, 0, pxB);
    } else {
      if (cur.valid) load_halo(cur, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): (TAG 2) the halo's LDS stores
    wg_barrier();                        // barrier #1: weights and tile 0 are in LDS
    int t = 0;
    if constexpr (TAG == 2 && WS_ABLATE != 6) {   // the first claim (tile NSLOT - 1 ...), taken at the start of the first iteration
      if (pw == QW && cur.valid) { if (lane == 0) raw_pend = atomicAdd(ctr, WS_CLAIM); pend = true; }
    }
#ifdef WS_PROBE_TIMING
    unsigned long long pt_issue = 0, pt_wait = 0, pt_bar = 0;
    WS_T(pt_begin);
#endif
    while (cur.valid) {
      WS_T(p0);
      const TileDesc nxt = read_slot((t + 1) % NSLOT);
      const bool more = read_slot((t + 2) % NSLOT).valid != 0;   // the queue is asked as long as the last tile it gave was one
      int i3 = -1;
      if constexpr (TAG == 2) {
        // the claim asked for a tile ago has come back by now (taken BEFORE this tile's loads are issued: the wait for it must
        // not cover them) ...
        if (pw == QW && WS_ABLATE != 6 && q_left == 0 && pend) { q_base = __builtin_amdgcn_readfirstlane(raw_pend); q_left = WS_CLAIM; pend = false; }
        load_patch(read_slot((t + 2) % NSLOT), pxB);   // tile t + 2's pixels: a whole tile to land
        // ... and the next one goes out at once (one claim in flight)
        if (pw == QW && WS_ABLATE != 6 && more && !pend && q_left <= 1) { if (lane == 0) raw_pend = atomicAdd(ctr, WS_CLAIM); pend = true; }
        if (nxt.valid) make_halo(nxt, (t + 1) & 1, pxA);
        if (pw == QW && more) {
          if constexpr (WS_ABLATE == 6) i3 = queue_index(t + NSLOT - 1);
          else if (q_left > 0) {
            const int v = q_base + (WS_CLAIM - q_left) + (NSLOT - 1) * gsize;
            --q_left;
            i3 = v < t_cnt ? t_lo + v : -1;
          }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) pxA[k] = pxB[k];
      } else if (WS_ABLATE != 1 && nxt.valid) {
        load_halo(nxt, (t + 1) & 1);
      }
      WS_T(p1);
      if (TAG != 2 && pw == QW && more) i3 = fetch();  // behind the passes: its round trip hides under theirs
      if constexpr (TAG != 2) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the next tile has landed (TAG 2: its stores are LDS stores, below)
      WS_T(p2);
      if (pw == QW) publish((t + NSLOT - 1) % NSLOT, i3);
      __builtin_amdgcn_s_waitcnt(0xC07F);
      wg_barrier();                      // end of tile t
      WS_T(p3);
#ifdef WS_PROBE_TIMING
      pt_issue += p1 - p0; pt_wait += p2 - p1; pt_bar += p3 - p2;
#endif
      cur = nxt;
      ++t;
    }
#ifdef WS_PROBE_TIMING
    if (pw == 0) {
      WS_T(pt_end);
      WS_ACC(4, pt_end - pt_begin); WS_ACC(5, pt_issue); WS_ACC(6, pt_wait); WS_ACC(7, pt_bar);
    }
#endif
    return;
  }

  // -------------------------------------------------------------------- consumers
  __builtin_amdgcn_s_setprio(WS_PRIO);
  const int l31 = lane & 31, hi = lane >> 5, wm = wave;
  lds_char *aptr[3][4], *wptr[2][4];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int c = dx + l31;
      aptr[dx][kk] = lds + (wm * 2) * ROW_BYTES + c * 128 + (((kk * 2 + hi) ^ halo_swz(c)) * 16);
    }
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    wptr[0][kk] = lds + LDS_W + l31 * 128 + (((kk * 2 + hi) ^ ((l31 >> 1) & 7)) * 16);
    wptr[1][kk] = wptr[0][kk] + 7 * 8192;
  }
  const float bias[2] = {p.bias[nb * 64 + 2 * l31], p.bias[nb * 64 + 2 * l31 + 1]};   // this lane's channel of each accumulator tile (even | odd)

  f32x16 accA[2][2], accB[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { accA[i][j][r] = 0.0f; accB[i][j][r] = 0.0f; }
  Epi epiA, epiB;
  epiA.rout = epiB.rout = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, 0u, 0x00020000);  // nothing to store yet
  epiA.rowoff[0] = epiA.rowoff[1] = epiB.rowoff[0] = epiB.rowoff[1] = OOB;
  epiA.pitch = epiB.pitch = out_pix_bytes;
  bf16x8 ah[2][4], w[3][2];   // halo-row fragments of two K groups, weight fragments of three steps

  auto aim_epi = [&](const TileDesc &d, Epi &e) {
    char *obase = reinterpret_cast<char *>(p.out) + ((size_t)d.b * Ho * Wo * p.out_stride + p.out_choff) * 2;
    e.rout = __builtin_amdgcn_make_buffer_rsrc(obase, 0, frame_out_bytes, 0x00020000);
    const int y0 = d.ty * TH + wm * 2;
    // this lane's channel pair nb*64 + 2 l31 (+ j per accumulator tile) of the pixel column x0 + 4 hi (+ the register's)
    const int x0 = tile_x0(d.tx, p.W);
    const unsigned ch = (unsigned)(nb * 64 + 2 * l31) * 2u;
    if constexpr (POOL) {
      e.rowoff[0] = y0 < p.H ? (unsigned)((y0 >> 1) * Wo + (x0 >> 1) + 2 * hi) * out_pix_bytes + ch : OOB;
      e.rowoff[1] = OOB;
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        e.rowoff[i] = y0 + i < p.H ? (unsigned)((y0 + i) * p.W + x0 + 4 * hi) * out_pix_bytes + ch : OOB;
    }
  };

  wg_barrier();  // barrier #0
  wg_barrier();  // barrier #1: weights and tile 0 are in LDS
  int t = 0;
  bool lastA = true, any = false;
#ifdef WS_PROBE_TIMING
  unsigned long long ct_head = 0;
  WS_T(ct_begin);
  const unsigned long long wc_begin = wall_clock64();
#endif
  // Per tile, in the MFMA shadows: the epilogue geometry of THIS tile (used while the next one computes), the
  // descriptor of the NEXT tile (published by the producers a tile ahead), and — inside k_steps — the barrier and the
  // next tile's first fragments.
  TileHooks<decltype(aim_epi)> hooks{aim_epi, lds, {}, {}, nullptr, {}, 0, 0ull};
  // The epilogue as single instructions spread over all MFMA gaps wins where the producers only issue LDS-direct passes
  // (-4 % conv1b-shaped, -9 % without a pool: a burst of 12-16 instructions delays the MFMA behind it); with conv1a in the
  // producers their VALU stream already sits in those gaps and the burst form is the faster one (+1.5 % otherwise).
  // (wall-clock A/B of non-instrumented builds, tools/microbench/run_probe12.sh)
  constexpr bool EPI_MICRO = WS_EPI_MICRO >= 0 ? WS_EPI_MICRO != 0 : TAG != 2;
  EpiTmp et;
  et.v[0] = et.v[1] = et.t[0] = et.t[1] = 0.0f; et.pk = 0u;

  auto run_tile = [&]<int BUF>(std::integral_constant<int, BUF>, f32x16(&acc)[2][2], const f32x16(&accPrev)[2][2], Epi &eMine,
                               const Epi &ePrev) -> bool {
    WS_T(ch0);
    hooks.eMine = &eMine;
    hooks.t = t;
    if (WS_ABLATE != 2) k_steps<0, POOL, BUF, EPI_MICRO>(ah, w, acc, accPrev, bias, aptr, wptr, ePrev, et, hooks);
    else { hooks.template at<3, 1>(); hooks.template at<24, 1>(); hooks.template at<28, 1>(); __builtin_amdgcn_s_waitcnt(0xC07F); wg_barrier(); }
    WS_T(c1);
#ifdef WS_PROBE_TIMING
    ct_head += c1 - ch0;   // (whole tile, for the probe's per-tile figure)
#endif
    ++t;
    hooks.dc = hooks.dn;
    return hooks.dn.valid != 0;
  };
  hooks.dc = read_slot(0);
  if (hooks.dc.valid) {
    // K steps 0 and 1 of the first tile (every later tile's come from steps 34 / 35 of its predecessor)
    read_frags<0, 0, -1>(ah, w, aptr, wptr);
    read_frags<1, 0, -1>(ah, w, aptr, wptr);
    while (true) {
      any = true;
      lastA = true;
      if (!run_tile(std::integral_constant<int, 0>{}, accA, accB, epiA, epiB)) break;
      lastA = false;
      if (!run_tile(std::integral_constant<int, 1>{}, accB, accA, epiB, epiA)) break;
    }
  }
#ifdef WS_PROBE_TIMING
  if (wm == 0) {
    WS_T(ct_end);
    WS_ACC(0, ct_end - ct_begin); WS_ACC(1, hooks.bar); WS_ACC(2, t); WS_ACC(8, ct_head);
    WS_ACC(3, wall_clock64() - wc_begin);  // constant-rate counter (100 MHz): [0] / [3] = shader clock
  }
#endif
  if (any) {
    constexpr int NEPI = POOL ? 8 : 16;
    auto flush = [&](const f32x16(&acc)[2][2], const Epi &e) {
      [&]<int... E>(std::integer_sequence<int, E...>) {
        (epi_item_c<POOL, E>(e, bias, acc), ...);
      }(std::make_integer_sequence<int, NEPI>{});
    };
    if (lastA) flush(accA, epiA); else flush(accB, epiB);
  }
}

template <bool POOL, int TAG>
static hipError_t launch(const ConvParams &p, hipStream_t s) {
  static_assert(LDS_TOTAL <= 160 * 1024, "halo double buffer + resident weights must fit the 160 KB LDS");
  auto k = conv_bf16_ws_kernel<POOL, TAG>;
  static bool attr_done[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) attr_done[dev] = true;
  }
  int grid = p.num_cus > 0 ? p.num_cus : 256;
  grid &= ~15;  // a multiple of 8 XCDs x (up to) 2 channel blocks
  if (grid < 16) grid = 16;
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), LDS_TOTAL, s, p);
  return hipGetLastError();
}

}  // namespace ws

size_t conv_bf16_ws_weight_bytes() { return ws::W_BYTES; }

// cin = 64 only; p.tile_ctr: >= nblk * 8 ints, zero on entry (the kernel leaves them non-zero)
hipError_t launch_conv_bf16_ws(const ConvParams &p, bool pool, int layer_tag, hipStream_t s) {
  if (!p.tile_ctr || p.nblk < 1 || p.nblk > 2 || p.W < 32 || (p.W & 1)) return hipErrorInvalidValue;
  if (layer_tag == 2) {   // conv1a fused: the u8 frames, conv1a's taps and bias instead of an input activation
    if (!pool || p.nblk != 1 || !p.img || !p.w1a || !p.b1a) return hipErrorInvalidValue;
    return ws::launch<true, 2>(p, s);
  }
  if (layer_tag == 1 && pool) return ws::launch<true, 1>(p, s);
  return pool ? ws::launch<true, 0>(p, s) : ws::launch<false, 0>(p, s);
}

}  // namespace spfe
