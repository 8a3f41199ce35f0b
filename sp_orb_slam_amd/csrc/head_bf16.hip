// head_bf16.hip — the two 1x1 head convolutions of the bf16 mode (no ReLU; /root/reference/orb_slam2/src/cv/
// sp_extractor.cpp:96-100): convDb (256 -> 256) and convPb (256 -> 65), plain GEMMs
//   out[P][COUT] (f32) = in[P][256 of the 512 head channels] (bf16) x W^T (bf16) + bias,  P = frames x cells,
// on v_mfma_f32_32x32x16_bf16.  As f32 kernels they were 70-100 us each of a ~1.7 ms bf16 step at 1280x720.
//
// One workgroup = 64 pixels x all 256 output channels: the pixel tile (all 256 input channels, 40 KB
// with the 80-byte row pitch of conv_bf16.hip) is loaded once and stays; the weights stream through a
// double buffer one 64-channel block (40 KB) at a time, everything with LDS-direct buffer loads.
// Wave w computes the 32 x 32 block (pixels 32*(w&1).., channels 32*(w>>1)..) of each block: 16 MFMAs.
// Pixels are the A operand, so lanes are channels and every store is whole 128-byte pixel rows.
// convPb's 65 outputs are two blocks (the second holds the dustbin channel and 63 zero rows).
#include <utility>

#include "spfe_kernels.h"

namespace spfe {

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
constexpr int HP = 80;                       // bytes per (row, 32-channel chunk) in LDS and in the packed weights
constexpr int H_TILE = 64;                   // pixels per workgroup
constexpr int H_BLOCK = 8 * 64 * HP;         // one operand block: 8 chunks x 64 rows x 80 B = 40960
constexpr int H_PASSES = H_BLOCK / 16 / 256; // LDS-direct passes per block (10)
constexpr unsigned H_OOB = 0x80000000u;
}  // namespace

// in: [npix][IN_STRIDE] bf16, the head reads channels [in_choff, in_choff + 256); out: [npix][COUT] f32
template <int NBLK, int COUT, int IN_STRIDE>
__global__ __launch_bounds__(256, 1) void head1x1_bf16_kernel(const unsigned short *__restrict__ in, int in_choff,
                                                              const unsigned char *__restrict__ wpack,
                                                              const float *__restrict__ bias,
                                                              float *__restrict__ out, int npix) {
  extern __shared__ __attribute__((aligned(16))) char sm_h[];
  char *sA = sm_h, *sW0 = sm_h + H_BLOCK, *sW1 = sm_h + 2 * H_BLOCK;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int p0 = blockIdx.x * H_TILE;
  const unsigned wslot = (unsigned)wave * 1024u;

  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned short *>(in) + in_choff, 0, (unsigned)((size_t)npix * IN_STRIDE * 2 - (size_t)in_choff * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rout =
      __builtin_amdgcn_make_buffer_rsrc(out, 0, (unsigned)((size_t)npix * COUT * 4), 0x00020000);

  // pixel tile: piece i -> (chunk, row, q): LDS offset 16 i; global (p0 + row) * 512 + chunk * 64 + q * 16
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int ps = 0; ps < H_PASSES; ++ps) {
    const int i = tid + ps * 256;
    const int q = i % 5, row = (i / 5) % 64, chunk = i / 320;
    const unsigned voff = q < 4 ? (unsigned)(p0 + row) * (unsigned)(IN_STRIDE * 2) + (unsigned)chunk * 64u + (unsigned)q * 16u : H_OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_void *)(sA + ps * 4096 + wslot), 16, voff, 0, 0, 0);
  }
  auto load_w = [&](int nb, char *dst) {
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char *>(wpack) + (size_t)nb * H_BLOCK, 0, (unsigned)H_BLOCK, 0x00020000);
#pragma unroll
    for (int ps = 0; ps < H_PASSES; ++ps)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void *)(dst + ps * 4096 + wslot), 16, (unsigned)tid * 16u,
                                               ps * 4096, 0, 0);
  };
  load_w(0, sW0);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();

  const int pr = (wave & 1) * 32, cr = (wave >> 1) * 32;
  const char *aBase = sA + (pr + l31) * HP + hi * 16;
#pragma unroll 1
  for (int nb = 0; nb < NBLK; ++nb) {
    char *wcur = (nb & 1) ? sW1 : sW0;
    if (nb + 1 < NBLK) load_w(nb + 1, (nb & 1) ? sW0 : sW1);
    const char *bBase = wcur + (cr + l31) * HP + hi * 16;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const bf16x8 a = *reinterpret_cast<const bf16x8 *>(aBase + c * 64 * HP + kk * 32);
        const bf16x8 b = *reinterpret_cast<const bf16x8 *>(bBase + c * 64 * HP + kk * 32);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
      }
    // D[pixel][channel]: lane & 31 = channel, register r = pixel (r&3) + 8*(r>>2) + 4*hi
    const int co = nb * 64 + cr + l31;
    const float bv = co < COUT ? bias[co] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int p = p0 + pr + (r & 3) + 8 * (r >> 2) + 4 * hi;
      const unsigned off = co < COUT ? (unsigned)p * (unsigned)(COUT * 4) + (unsigned)co * 4u : H_OOB;   // (pixels past npix: out of range too)
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[r] + bv), rout, off, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // the next block of weights has landed
    __syncthreads();
  }
#endif
}

size_t head_bf16_weight_bytes(int cout) { return (size_t)((cout + 63) / 64) * H_BLOCK; }

template <int NBLK, int COUT>
static hipError_t launch_head(const void *in_bf16, int in_choff, const void *wpack, const float *bias, float *out, int npix,
                              hipStream_t s) {
  constexpr size_t lds = 3 * (size_t)H_BLOCK;
  auto k = head1x1_bf16_kernel<NBLK, COUT, 512>;
  static bool attr_done[64] = {};  // per instantiation and device: one process may hold handles on several GPUs
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) attr_done[dev] = true;
  }
  if (npix <= 0) return hipSuccess;
  hipLaunchKernelGGL(k, dim3((npix + H_TILE - 1) / H_TILE), dim3(256), lds, s, reinterpret_cast<const unsigned short *>(in_bf16),
                     in_choff, reinterpret_cast<const unsigned char *>(wpack), bias, out, npix);
  return hipGetLastError();
}

// in_bf16: [npix][512] = ReLU(convPa) | ReLU(convDa); cout 256: the descriptor head on channels 256..511,
// cout 65: the detector head on channels 0..255
hipError_t launch_head1x1_bf16(const void *in_bf16, const void *wpack, const float *bias, float *out, int npix, int cout,
                               hipStream_t s) {
  if (cout == 256) return launch_head<4, 256>(in_bf16, 256, wpack, bias, out, npix, s);
  if (cout == 65) return launch_head<2, 65>(in_bf16, 0, wpack, bias, out, npix, s);
  return hipErrorInvalidValue;
}

}  // namespace spfe
