// da_gather_bf16.hip — convDa (3x3, 128 -> 256, ReLU; /root/reference/orb_slam2/src/cv/sp_extractor.cpp:99) of the bf16
// mode ON THE LISTED CELLS ONLY.  The descriptor branch (convDa -> convDb -> bilinear sampling, :99-103, :134-148) is read
// at the emitted keypoints' four taps and nowhere else, so behind the selection it runs as gathered GEMMs over
// select_kernel's cell list (FrameBufs::db_list): this kernel writes ReLU(convDa) into channels 256..511 of the listed
// rows of the head activations, head_bf16.hip's gathered convDb reads exactly those rows.  At 1280x720 and 1000 keypoints
// the list is 19 % of the frame.
//
// Shape: out[cell][256] = sum over 9 taps x 128 channels.  A workgroup is 4 wavefronts, one per SIMD, and owns one
// 128-channel half of convDa for the whole kernel: wavefront w keeps the B operands of its 32 output channels (72 K steps
// x 4 registers, conv_bf16_rw.hip's table and register classes) in registers.  All four work on the same tile of 32 listed
// cells, whose im2col rows — 9 taps x 32 cells x 256 bytes — come L2 -> LDS with LDS-direct loads into a double buffer
// (XOR-swizzled by the cell on the source side: conflict-free 16-byte fragment reads); taps outside the frame are
// out-of-range loads = the zero padding.  The previous tile's outputs leave (16 two-byte stores per lane: 64-byte runs)
// while this tile computes.
//
// Measured (1280x720 x 8, 22 k listed cells = 690 tiles x 2 groups on 256 workgroups): 25 us, of which the MFMA loops are
// 6 x 1.1 us — they run at the LDS's full read rate (four wavefronts x one 1 KB fragment per 32-cycle MFMA = 128 B / clk),
// so the LDS-direct loads of the next tile get no LDS write slots beside them and a tile costs its loads' round trip PLUS
// its MFMAs (probe: without the loads 21 us, without the MFMAs 19 us).  Used for synchronous calls (spfe_schedule.hip).
//
// Arithmetic: conv_bf16_rw.hip's — mfma(cells, weights), v_mfma_f32_32x32x16_bf16, K order 32-channel chunk -> dx ->
// 16-channel group -> dy, f32 accumulate, bias, ReLU, RNE to bf16 — so a listed row holds the bits the dense launch writes
// there (tests/test_gpu_sparse_db.py).
#include <algorithm>

#include "spfe_kernels.h"

namespace spfe {
namespace dag {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(3))) bf16x8 lds_frag;

constexpr unsigned OOB = 0x80000000u;
constexpr int NSTEP = 72;                      // conv_bf16_rw.hip: 4 chunks x 3 dx x 2 groups x 3 dy
constexpr int NW_AGPR = 64;                    // weight fragments kept in AGPRs (4 registers each)
constexpr int CELLS = 32;                      // listed cells per tile
constexpr int TAP_BYTES = CELLS * 256;         // one tap's rows: 128 channels bf16 per cell
constexpr int BUF_BYTES = 9 * TAP_BYTES;       // 73,728
[[maybe_unused]] constexpr int PASSES = BUF_BYTES / 1024 / 4;   // 18 LDS-direct passes per wavefront and tile
constexpr int LDS_IDX = 2 * BUF_BYTES;         // [4][32] cell indices of the tiles in flight
constexpr int LDS_TOTAL = LDS_IDX + 4 * CELLS * 4;

struct KStep {
  int tap, piece;
};
__host__ __device__ constexpr KStep kstep_of(int s) {   // (conv_bf16_rw.hip's order)
  const int chunk = s / 18, r = s % 18, dx = r / 6, k2 = (r % 6) / 3, dy = r % 3;
  return KStep{dy * 3 + dx, chunk * 4 + k2 * 2};
}

__device__ __forceinline__ float relu_nc(float a) {
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(a));
  return r;
}

// feat: [B][hc][wc][128] bf16 (conv4b's output); wpack: conv_bf16_rw_pack_weights of convPa|Da (4 groups of 128 output
// channels; convDa = groups 2, 3); bias: [512]; out: [B * hc * wc][512] bf16, this kernel writes channels 256..511
__global__ __launch_bounds__(256, 1) void da_gather_bf16_kernel(const unsigned short *__restrict__ feat,
                                                                const unsigned char *__restrict__ wpack,
                                                                const float *__restrict__ bias, unsigned short *__restrict__ out,
                                                                const int *__restrict__ list, const int *__restrict__ total,
                                                                int B, int hc, int wc) {
  extern __shared__ __attribute__((aligned(16))) char smem_dag[];
  lds_char *const lds = (lds_char *)smem_dag;
  int *const sIdx = reinterpret_cast<int *>(smem_dag + LDS_IDX);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  // workgroups b and b + 8 — same XCD (round-robin by index), so same L2 — take the same tiles for the two 128-channel
  // groups of convDa: the second one's im2col rows are L2 hits
  const int cg = 2 + (int)((blockIdx.x >> 3) & 1);
  const int wg = (int)(blockIdx.x & 7) + 8 * (int)(blockIdx.x >> 4), nwg = (int)(gridDim.x >> 1);
  const int nwalk = __builtin_amdgcn_readfirstlane(*total);
  const int ntiles = (nwalk + CELLS - 1) / CELLS;
  if (wg >= ntiles) return;
  const int C = hc * wc;
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(feat), 0, (unsigned)((size_t)B * C * 256), 0x00020000);
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(out, 0, (unsigned)((size_t)B * C * 1024), 0x00020000);
  (void)rin;   // (the host pass of hipcc does not see the uses below)
  const float bv = bias[cg * 128 + wave * 32 + l31];
  const unsigned out_ch_bytes = (unsigned)(cg * 128 + wave * 32 + l31) * 2u;

  // ---- im2col staging.  Pass i of this wavefront covers LDS pieces q = (4 i + wave) * 64 + lane of the tile: tap q / 512,
  // cell (q % 512) / 16, slot q % 16 — so a thread serves two cells (4 wave + lane / 16, + 16 for odd passes), all nine taps
  // (tap i / 2); slot s of a cell's row holds its 16-byte piece s ^ (cell & 15).
  int gidx[2] = {-1, -1};   // the two cells of the tile about to be staged
  auto load_idx = [&](int tile) {
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
      const int p = tile * CELLS + 16 * c2 + 4 * wave + (lane >> 4);
      gidx[c2] = p < nwalk ? list[p] : -1;
    }
  };
  auto dma = [&](int buf, int ring) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned src0[2];
    int cy[2], cx[2];
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
      const int g = gidx[c2] < 0 ? 0 : gidx[c2];
      const int cell = 16 * c2 + 4 * wave + (lane >> 4);
      const int rem = g % C;
      cy[c2] = gidx[c2] < 0 ? -4 : rem / wc;     // (no cell: every tap is outside)
      cx[c2] = rem - (rem / wc) * wc;
      src0[c2] = (unsigned)g * 256u + (unsigned)(((lane & 15) ^ (cell & 15)) * 16);
      if ((lane & 15) == 0) sIdx[ring * CELLS + cell] = gidx[c2];
    }
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
      const int c2 = i & 1, tap = i >> 1, dy = tap / 3 - 1, dx = tap % 3 - 1;
      const bool ok = (unsigned)(cy[c2] + dy) < (unsigned)hc && (unsigned)(cx[c2] + dx) < (unsigned)wc;
      const unsigned src = ok ? src0[c2] + (unsigned)((dy * wc + dx) * 256) : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_void *)(lds + buf * BUF_BYTES + (4 * i + wave) * 1024), 16, src, 0, 0, 0);
    }
#endif
  };

  // ---- this wavefront's weights: 72 fragments of 8 bf16 per lane, in registers for the whole kernel (conv_bf16_rw.hip)
  bf16x8 wreg[NSTEP];
  {
    const char *wb = reinterpret_cast<const char *>(wpack) + ((size_t)(cg * 4 + wave) * NSTEP * 64 + lane) * 16;
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) wreg[s] = *reinterpret_cast<const bf16x8 *>(wb + (size_t)s * 1024);
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      if (s < NW_AGPR) asm volatile("" : "+a"(wreg[s]));
      else asm volatile("" : "+v"(wreg[s]));
    }
  }

  // D[cell][channel]: register r = cell (r & 3) + 8 (r >> 2) + 4 hi of the tile, this lane's channel
  auto store_tile = [&](const f32x16 &acc, int ring) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cell = sIdx[ring * CELLS + (r & 3) + 8 * (r >> 2) + 4 * hi];
      const float v = relu_nc(acc[r] + bv);
      const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){v, v}, bf16x2));
      const unsigned off = cell >= 0 ? (unsigned)cell * 1024u + out_ch_bytes : OOB;
      __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(pk & 0xffffu), rout, off, 0, 0);
    }
  };

  const unsigned lane_row = (unsigned)(l31 * 256);
  const unsigned xs = (unsigned)((l31 & 15) << 4);
  unsigned poff[8];   // byte offset, inside this lane's cell row, of its fragment of (chunk, 16-channel group) q: piece 2 q + hi
#pragma unroll
  for (int q = 0; q < 8; ++q) poff[q] = (unsigned)((2 * q + hi) << 4) ^ xs;
  f32x16 accA, accB;
  int tile = wg, it = 0;
  bool have_prev = false;
  load_idx(tile);
  dma(0, 0);
  load_idx(tile + nwg);
  int buf = 0;
  auto run = [&](f32x16 &acc, const f32x16 &accPrev) {
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this tile has landed (and the previous tile's stores are out)
    __syncthreads();                      // ... for every wavefront; and every wavefront is done reading the other buffer
    const int nxt = tile + nwg;
    if (nxt < ntiles) {
      dma(buf ^ 1, (it + 1) & 3);
      load_idx(nxt + nwg);
    }
    if (have_prev) store_tile(accPrev, (it - 1) & 3);   // the previous tile's outputs leave while this one computes
    lds_char *const a0 = lds + buf * BUF_BYTES + lane_row;
    bf16x8 a[3];
    auto rd = [&](int s) -> bf16x8 {
      const KStep k = kstep_of(s);
      return *reinterpret_cast<lds_frag *>(a0 + k.tap * TAP_BYTES + poff[k.piece >> 1]);
    };
    a[0] = rd(0);
    a[1] = rd(1);
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      if (s + 2 < NSTEP) a[(s + 2) % 3] = rd(s + 2);
      __builtin_amdgcn_sched_barrier(0);
      if (s == 0) {
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.0f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s % 3], wreg[s], z, 0, 0, 0);
      } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s % 3], wreg[s], acc, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    have_prev = true;
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
  if (have_prev) {
    if (lastA) store_tile(accA, (it - 1) & 3); else store_tile(accB, (it - 1) & 3);
  }
}

}  // namespace dag

// convDa on the *total (<= max_total) cells of `list`: feat = conv4b's output [B][hc][wc][128] bf16, wpack = convPa|Da in
// conv_bf16_rw_pack_weights order, out = the head activations [B * hc * wc][512] bf16 (channels 256..511 of the listed rows)
hipError_t launch_da_gather_bf16(const void *feat, const void *wpack, const float *bias, void *out, const int *list,
                                 const int *total, int max_total, int B, int hc, int wc, int num_cus, hipStream_t s) {
  if (!feat || !wpack || !bias || !out || !list || !total) return hipErrorInvalidValue;
  if (max_total <= 0) return hipSuccess;
  if ((size_t)B * hc * wc * 1024 >= ((size_t)1 << 31)) return hipErrorInvalidValue;   // 32-bit buffer offsets, OOB marker
  auto k = dag::da_gather_bf16_kernel;
  static bool attr_done[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, dag::LDS_TOTAL);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) attr_done[dev] = true;
  }
  const int ntiles = (max_total + dag::CELLS - 1) / dag::CELLS;
  int grid = num_cus > 0 ? num_cus : 256;
  grid = std::max(16, std::min(grid & ~15, 16 * ((ntiles + 7) / 8)));   // 8 XCDs x two channel groups x the workgroups that walk the tiles
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), dag::LDS_TOTAL, s, reinterpret_cast<const unsigned short *>(feat),
                     reinterpret_cast<const unsigned char *>(wpack), bias, reinterpret_cast<unsigned short *>(out), list, total, B, hc, wc);
  return hipGetLastError();
}

}  // namespace spfe
