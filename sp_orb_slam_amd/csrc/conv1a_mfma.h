// conv1a_mfma.h — the first convolution of the bf16 mode (conv1a, 1 -> 64 channels, 3 x 3;
// /root/reference/orb_slam2/src/cv/sp_extractor.cpp:81 after the input conversion :388) as a K = 16 matrix
// product on v_mfma_f32_32x32x16_bf16, shared by the two places that compute it so that they produce the same bits:
//   * conv_bf16_ws.hip, TAG 2: the producer waves of conv1b make conv1b's halo tile instead of loading it;
//   * conv_bf16.hip, conv1a_bf16_kernel: the stand-alone layer (launches too small for the wave-specialised kernel).
//
// Definition (the oracle's conv1a_bf16 restates it; round 6 form):
//   s[c]  = bias[c] + sum over the 9 taps of  float(u8 pixel) * bf16(w[c][tap] * (1/255))      (f32 accumulate; products exact)
//   a0[c] = bf16( max( s[c], 0 ) )
// i.e. the 1/255 of convertTo(CV_32F, 1.f/255.f) is folded into the weights BEFORE their rounding to bf16 (one f32 multiply by
// float(1/255), then RNE: the host packing and the oracle do the same), the matrix operands stay exact — u8 values are
// integers below 2^8 (exact in bf16), an 8-bit x 8-bit significand product is exact in f32 — and the bias is the MFMA's C
// operand, the accumulator's initial value.  What is left per output value is the rounding and the ReLU.  (Rounds 2 - 5 rounded
// w itself and applied fmaf(sum, 1/255, bias) per value: two more VALU instructions per pair of values in the producers of the
// fused conv1b, whose issue slots are what that kernel is short of — conv_bf16_ws.hip.  Both are bf16 quantisations of the
// same f32 layer with the same relative weight error; the mode's tolerances and its flip / margin reports were re-made.)
// As VALU code (9 taps x 64 channels of f32 FMAs per pixel) this layer cost 0.27 ms per eight 1280x720 frames on its own,
// and ~950 instructions per tile and wave when fused — more issue slots than the MFMA stream beside it leaves; as a
// matrix product it is 2 MFMAs per 32 pixels.
//
// Operands of one 32-pixel group (lane = (l31 = lane & 31, hi = lane >> 5)):
//   A (weights, per channel tile j): row m = l31 <-> channel 32 j + row_channel(m), k = 8 hi + e <-> tap k (zero for k >= 9):
//     the packed table `wtab` [2][64 lanes][8 bf16] built by the host;
//   B (pixels): column n = l31 <-> the group's pixel l31, k = 8 hi + e <-> its tap k, read from a bf16 patch in LDS;
//   D[row][pixel]: the lane owns one pixel, register r <-> row 8 (r >> 2) + 4 hi + (r & 3).
// Which channel a row computes is free (the rows are independent dot products), and round 6 chose it so that a lane ends up
// with WHOLE 16-byte pieces of the NHWC activation: row_channel swaps bits 2 and 3 of the row index, so register r of lane
// (l31, hi) holds channel 32 j + 16 (r >> 3) + 8 hi + (r & 7) — registers 8 rr .. 8 rr + 7 are the eight consecutive channels
// of piece 4 j + 2 rr + hi of the lane's pixel.  (Before, a lane held the low or the high 8 bytes of every piece according to
// hi; the 16 lanes of a ds_write_b64 group then all wrote the same half of their pieces: 2-way bank conflicts on every halo
// write of the fused conv1b, 19 % of its LDS cycles.  Now the even pixels of a group write the low half of their piece while
// the odd ones write the high half, and the other way round in a second store: conv_bf16_ws.hip, make_halo.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace spfe {
namespace c1a {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) unsigned short lds_u16;

constexpr int PATCH_PITCH = 40;   // bf16 elements per patch row (36 used)

// the bf16 bit pattern of an integer 0..255 (exact: 8 significant bits)
__device__ __forceinline__ unsigned short u8_to_bf16(unsigned v) { return (unsigned short)(__float_as_uint((float)v) >> 16); }

// B operand of a lane: `tap0` = LDS address of its pixel's tap (0, 0) in a bf16 patch of pitch PATCH_PITCH
__device__ __forceinline__ bf16x8 pixel_operand(lds_u16 *tap0, int hi) {
  unsigned t[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) t[k] = tap0[(k / 3) * PATCH_PITCH + k % 3];
  u32x4 d;
  d.x = hi ? t[8] : (t[0] | (t[1] << 16));
  d.y = hi ? 0u : (t[2] | (t[3] << 16));
  d.z = hi ? 0u : (t[4] | (t[5] << 16));
  d.w = hi ? 0u : (t[6] | (t[7] << 16));
  return __builtin_bit_cast(bf16x8, d);
}

// the two accumulator tiles of a 32-pixel group; `bias` = the accumulators' initial values (register r of tile j <-> the
// lane's channel 32 j + 16 (r >> 3) + 8 hi + (r & 7), load_constants)
__device__ __forceinline__ void product(const bf16x8 (&wA)[2], bf16x8 px, const float (&bias)[2][16], f32x16 (&acc)[2]) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    f32x16 c;
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = bias[j][r];
    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wA[j], px, c, 0, 0, 0);
  }
}

// row m of a channel tile <-> channel 32 j + row_channel(m): bits 2 and 3 of m swapped (host packing: spfe_pack.hip)
__host__ __device__ constexpr int row_channel(int m) { return (m & ~12) | ((m & 4) << 1) | ((m & 8) >> 1); }

// epilogue of one (channel tile j, rr = r >> 3): channels 32 j + 16 rr + 8 hi + 0..7 of the lane's pixel = piece 4 j + 2 rr + hi
// of its 128-byte NHWC row, as 16 bytes of bf16
typedef short i16x2 __attribute__((ext_vector_type(2)));
// epilogue of one (channel tile j, rr = r >> 3): channels 32 j + 16 rr + 8 hi + 0..7 of the lane's pixel = piece 4 j + 2 rr + hi
// of its 128-byte NHWC row, as 16 bytes of bf16: RNE, then the ReLU on the rounded pair as ONE v_pk_max_i16 against 0 — a
// negative float is a negative int16 in its upper half and rounds to a negative (or -0) bf16, RNE is monotonic and keeps the
// sign: max(bf16(v), +0) as 16-bit integers == bf16(max(v, 0)) bit for bit.
__device__ __forceinline__ u32x4 finish8(const f32x16 &acc, int rr) {
  unsigned o[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const i16x2 r = __builtin_bit_cast(i16x2, __builtin_convertvector((f32x2){acc[8 * rr + 2 * m], acc[8 * rr + 2 * m + 1]}, bf16x2));
    const i16x2 z = {0, 0};
    o[m] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(r, z));
  }
  return (u32x4){o[0], o[1], o[2], o[3]};
}

// per-lane constants: A operands from the host table, the lane's 2 x 16 biases (register r of tile j <-> channel
// 32 j + 16 (r >> 3) + 8 hi + (r & 7))
__device__ __forceinline__ void load_constants(const void *wtab, const float *b64, int lane, bf16x8 (&wA)[2], float (&bias)[2][16]) {
  const int hi = lane >> 5;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    wA[j] = __builtin_bit_cast(bf16x8, reinterpret_cast<const u32x4 *>(wtab)[j * 64 + lane]);
#pragma unroll
    for (int r = 0; r < 16; ++r) bias[j][r] = b64[32 * j + 16 * (r >> 3) + 8 * hi + (r & 7)];
  }
}

}  // namespace c1a
}  // namespace spfe
