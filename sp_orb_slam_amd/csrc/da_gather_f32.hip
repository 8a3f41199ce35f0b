// da_gather_f32.hip — convDa (3x3, 128 -> 256, ReLU; /root/reference/orb_slam2/src/cv/sp_extractor.cpp:99) of the f32
// mode ON THE LISTED CELLS ONLY: the f32 counterpart of da_gather_bf16.hip.  Behind the selection the descriptor branch
// (convDa -> convDb -> sampling) runs over select_kernel's cell list (FrameBufs::db_list): this kernel writes ReLU(convDa)
// into channels 256..511 of the listed rows of the head activations, head_f32.hip's gathered convDb reads those rows.
//
// BIT-IDENTICAL to conv_f32.hip on those rows: the arithmetic contract (include/spfe_exact_math.h) makes an output
// acc = +0; for chunk of 16 channels, tap = 3 dy + dx, channel pair: acc = fmaf(x, w, acc) twice; out = max(acc + bias, 0),
// and v_mfma_f32_32x32x2_f32 is that k-ordered chain.  Taps outside the frame are out-of-range loads = zeros, multiplied
// and added like the dense kernel's zero padding.
//
// Shape.  A work item is 64 listed cells x 64 output channels; a workgroup is 4 wavefronts = 2 cell halves x 2 channel
// halves, one 32x32 accumulator each, 576 MFMAs per item.  K runs in 24 stages (chunk, dy): the stage's three taps of
// A — [tap][4 channel quads][64 cells][4 floats], gathered — and of W — [tap][quad][64 channels][4] from a table packed in
// that order — come L2 -> LDS with LDS-direct loads into a ring of two stages, six 16-byte pieces per thread and stage; one
// barrier per stage (24 MFMAs per wavefront).  A lane's piece holds the operands of two K steps (channels hi and 2 + hi of its
// quad): one ds_read2_b32.  The ring runs on across the workgroup's items.
//
// What bounds it.  One workgroup alone on a CU runs the stage loop at about half the MFMA rate (a barrier per 24 MFMAs, the
// LDS-direct issue on a SIMD with no second wavefront to fill the gaps, at the clock a short burst gets; NOT the accumulator
// dependency: tools/microbench/mfma_chain_probe.hip — one dependent chain of v_mfma_f32_32x32x2_f32 issues every 64 cycles
// like four independent ones): a single frame's list (152 items, one per CU) takes 34 us whatever the prefetch depth.
// Throughput comes from other wavefronts on the same SIMD: 49 KB of LDS per workgroup = three per CU; 752x480 x 8 (1204
// items): 162 us with one workgroup per CU, 122 with two, 114 with three.
#include <algorithm>
#include <cstring>

#include "spfe_kernels.h"

namespace spfe {
namespace dagf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) float lds_f1;

constexpr unsigned OOB = 0x80000000u;
constexpr int CELLS = 64, CH = 64;
constexpr int NSTAGE = 24;                     // per item: 8 chunks of 16 channels x 3 tap rows
constexpr int PART_BYTES = 3 * 4 * 1024;       // A (or W) of one stage: [3 taps][4 quads][64][16 bytes]
constexpr int STAGE_BYTES = 2 * PART_BYTES;    // 24,576
constexpr int RING = 2;                        // 49 KB + indices: three workgroups per CU (see above)
constexpr int LDS_IDX = RING * STAGE_BYTES;    // [4][64] cell indices of the items in flight
constexpr int LDS_TOTAL = LDS_IDX + 4 * CELLS * 4;

// feat: [B][hc][wc][128] f32 (conv4b's output); wpack: da_gather_f32_pack_weights; bias: convDa's 256; out: [B * hc * wc][512]
// f32 head activations, channels 256..511 written
__global__ __launch_bounds__(256, 3) void da_gather_f32_kernel(const float *__restrict__ feat, const float *__restrict__ wpack,
                                                               const float *__restrict__ bias, float *__restrict__ out,
                                                               const int *__restrict__ list, const int *__restrict__ total,
                                                               int B, int hc, int wc) {
  extern __shared__ __attribute__((aligned(16))) char smem_dagf[];
  lds_char *const lds = (lds_char *)smem_dagf;
  int *const sIdx = reinterpret_cast<int *>(smem_dagf + LDS_IDX);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wm = wave & 1, wn = wave >> 1;   // this wavefront's cell half / channel half of the item
  const int nwalk = __builtin_amdgcn_readfirstlane(*total);
  const int nitems = ((nwalk + CELLS - 1) / CELLS) * 4;   // item = tile * 4 + 64-channel block
  if ((int)blockIdx.x >= nitems) return;
  const int C = hc * wc;
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(feat), 0, (unsigned)((size_t)B * C * 512), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(wpack), 0, (unsigned)(4 * 8 * 9 * 4 * 64 * 16), 0x00020000);
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(out, 0, (unsigned)((size_t)B * C * 2048), 0x00020000);
  (void)rin; (void)rw;   // (the host pass of hipcc does not see the uses below)

  // ---- the issue side of the ring: stage (item, st) -> ring slot.  A thread loads, for each of the stage's three taps, the
  // 16-byte piece (quad = wave, cell = lane) of A and (quad = wave, channel = lane) of W.  The cell indices of item number n
  // of this workgroup sit in sIdx[n & 3]; those of item n + 1 come in with item n's first stage (an LDS-direct load like
  // the others: a register load here would make the compiler drain the ring where its value is used).
  const int G = (int)gridDim.x;
  int is_item = (int)blockIdx.x, is_n = 0, is_st = 0;   // next stage to issue
  unsigned is_base = OOB;                                            // this thread's cell of the item being issued: byte offset of its row + quad
  int is_cy = -4, is_cx = 0;
  const __amdgpu_buffer_rsrc_t rlist = __builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(list), 0, (unsigned)nwalk * 4u, 0x00020000);
  (void)rlist;
  auto aim = [&](int item, int g) {   // geometry of this thread's cell of the item about to be issued
    const bool valid = item < nitems && (item >> 2) * CELLS + lane < nwalk;
    const int gg = valid ? g : 0;
    const int rem = gg % C;
    is_cy = valid ? rem / wc : -4;     // (no cell: every tap is outside)
    is_cx = rem - (rem / wc) * wc;
    is_base = (unsigned)gg * 512u + (unsigned)wave * 16u;
  };
  auto issue = [&](const int is_slot) {   // one stage, into ring slot is_slot (a constant at every call site: the compiler
                                          // must see that these loads do not touch the slot being read)
#if defined(__HIP_DEVICE_COMPILE__)
    if (is_item < nitems) {
      const int chunk = is_st / 3, dy = is_st % 3 - 1;
      const unsigned wsrc = (unsigned)(((((is_item & 3) * 8 + chunk) * 9 + (dy + 1) * 3) * 4 + wave) * 64 + lane) * 16u;
      lds_char *const dst = lds + is_slot * STAGE_BYTES + wave * 1024;
#pragma unroll
      for (int t3 = 0; t3 < 3; ++t3) {
        const int dx = t3 - 1;
        const bool ok = (unsigned)(is_cy + dy) < (unsigned)hc && (unsigned)(is_cx + dx) < (unsigned)wc;
        const unsigned asrc = ok ? is_base + (unsigned)((dy * wc + dx) * 512 + chunk * 64) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_void *)(dst + t3 * 4096), 16, asrc, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void *)(dst + PART_BYTES + t3 * 4096), 16, wsrc + (unsigned)(t3 * 4 * 64 * 16), 0, 0, 0);
      }
    } else {   // past the last item: keep the count of loads per stage (the waits below count them)
#pragma unroll
      for (int t3 = 0; t3 < 6; ++t3)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_void *)(lds + is_slot * STAGE_BYTES + wave * 1024 + t3 * 4096), 16, OOB, 0, 0, 0);
    }
    if (is_st == 0)   // the cell indices of the workgroup's NEXT item (every wavefront: same data, same place, same load count)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rlist, (lds_void *)(lds + LDS_IDX + ((is_n + 1) & 3) * CELLS * 4), 4,
                                               (unsigned)(((is_item + G) >> 2) * CELLS + lane) * 4u, 0, 0, 0);
#endif
    if (++is_st == NSTAGE) {   // on to the workgroup's next item (its indices came in 24 stages ago)
      is_st = 0;
      is_item += G;
      ++is_n;
      aim(is_item, sIdx[(is_n & 3) * CELLS + lane]);
    }
  };

  // operands of this lane inside a stage part: A piece (quad q, cell 32 wm + l31), W piece (quad q, channel 32 wn + l31)
  const unsigned a_lane = (unsigned)((32 * wm + l31) * 16 + hi * 4), w_lane = (unsigned)(PART_BYTES + (32 * wn + l31) * 16 + hi * 4);

  float bias4[4];   // this lane's bias in each of the four 64-channel blocks (loaded here: a load inside the loop drains the ring)
#pragma unroll
  for (int b4 = 0; b4 < 4; ++b4) bias4[b4] = bias[b4 * CH + 32 * wn + l31];
  int item = (int)blockIdx.x, n = 0;
  {
    const int p0 = (item >> 2) * CELLS + lane;
    const int g0 = p0 < nwalk ? list[p0] : 0;
    if (wave == 0) sIdx[lane] = g0;
    aim(item, g0);
  }
  issue(0);
  while (item < nitems) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll 1
    for (int st4 = 0; st4 < NSTAGE / RING; ++st4) {
#pragma unroll
      for (int slot = 0; slot < RING; ++slot) {   // (24 stages = 12 turns of the ring: an item starts at slot 0)
        // the stage issued one stage ago must have landed
        __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0), expcnt / lgkmcnt untouched
        asm volatile("s_barrier" ::: "memory");     // ... for every wavefront; and everybody is done with the slot refilled next
                                                   // (not __syncthreads(): its fence would wait for ALL loads in flight)
        issue((slot + 1) % RING);
        lds_char *const sb = lds + slot * STAGE_BYTES;
        // 12 operand pairs (tap, quad) per stage, read two pairs ahead of their MFMAs (pinned: left alone, the scheduler sinks
        // every read to its first use and the matrix pipe waits out an LDS round trip per pair)
        // (a lane's 16-byte piece holds channels e0..e3 of its quad; its K steps need e[hi] and e[2 + hi]: two dwords 8 bytes
        // apart, one ds_read2_b32 — no selects between the MFMAs)
        float a0v[3], a1v[3], w0v[3], w1v[3];
        auto rd = [&](int i) {
          const lds_f1 *ap = reinterpret_cast<const lds_f1 *>(sb + (i / 4) * 4096 + (i % 4) * 1024 + a_lane);
          const lds_f1 *wp = reinterpret_cast<const lds_f1 *>(sb + (i / 4) * 4096 + (i % 4) * 1024 + w_lane);
          a0v[i % 3] = ap[0]; a1v[i % 3] = ap[2];
          w0v[i % 3] = wp[0]; w1v[i % 3] = wp[2];
        };
        rd(0);
        rd(1);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          if (i + 2 < 12) rd(i + 2);
          __builtin_amdgcn_sched_barrier(0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0v[i % 3], w0v[i % 3], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v[i % 3], w1v[i % 3], acc, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    // D[cell][channel]: register r = cell (r & 3) + 8 (r >> 2) + 4 hi of this wavefront's 32, lane = channel
    const int co = 256 + (item & 3) * CH + 32 * wn + l31;
    const int blk = item & 3;
    const float bv = blk == 0 ? bias4[0] : (blk == 1 ? bias4[1] : (blk == 2 ? bias4[2] : bias4[3]));
    const int p_base = (item >> 2) * CELLS + 32 * wm;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = (r & 3) + 8 * (r >> 2) + 4 * hi;
      const int cell = sIdx[(n & 3) * CELLS + 32 * wm + j];
      float v = acc[r] + bv;
      v = v > 0.0f ? v : 0.0f;
      const unsigned off = p_base + j < nwalk ? (unsigned)cell * 2048u + (unsigned)co * 4u : OOB;
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rout, off, 0, 0);
    }
    item += G;
    ++n;
  }
}

}  // namespace dagf

size_t da_gather_f32_weight_bytes() { return (size_t)4 * 8 * 9 * 4 * 64 * 16; }

// W: convDa's [256][128][9] f32 (OIHW) -> [block 4][chunk 8][tap 9][quad 4][channel 64][4]:
// element e = W[64 block + channel][16 chunk + 4 quad + e][tap]
void da_gather_f32_pack_weights(const float *W, float *dst) {
  for (int blk = 0; blk < 4; ++blk)
    for (int chunk = 0; chunk < 8; ++chunk)
      for (int tap = 0; tap < 9; ++tap)
        for (int q = 0; q < 4; ++q)
          for (int ch = 0; ch < 64; ++ch)
            for (int e = 0; e < 4; ++e)
              dst[((((((size_t)blk * 8 + chunk) * 9 + tap) * 4 + q) * 64 + ch) * 4) + e] =
                  W[((size_t)(64 * blk + ch) * 128 + 16 * chunk + 4 * q + e) * 9 + tap];
}

// convDa on the *total (<= max_total) cells of `list`: feat = conv4b's output [B][hc][wc][128] f32, out = the head
// activations [B * hc * wc][512] f32 (channels 256..511 of the listed rows), bias = convDa's 256 values
hipError_t launch_da_gather_f32(const float *feat, const float *wpack, const float *bias, float *out, const int *list,
                                const int *total, int max_total, int B, int hc, int wc, int num_cus, hipStream_t s) {
  if (!feat || !wpack || !bias || !out || !list || !total) return hipErrorInvalidValue;
  if (max_total <= 0) return hipSuccess;
  if ((size_t)B * hc * wc * 2048 >= ((size_t)1 << 31)) return hipErrorInvalidValue;   // 32-bit buffer offsets, OOB marker
  auto k = dagf::da_gather_f32_kernel;
  static bool attr_done[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, dagf::LDS_TOTAL);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) attr_done[dev] = true;
  }
  const int nitems = ((max_total + dagf::CELLS - 1) / dagf::CELLS) * 4;
  int grid = 3 * (num_cus > 0 ? num_cus : 256);   // three workgroups per CU
  grid = std::max(1, std::min(grid, nitems));
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), dagf::LDS_TOTAL, s, feat, wpack, bias, out, list, total, B, hc, wc);
  return hipGetLastError();
}

}  // namespace spfe
