// conv_rw_probe.hip — stand-alone timing of the register-resident-weights bf16 convolution
// (sp_orb_slam_amd/csrc/conv_bf16_rw.hip) on random data, with the in-kernel cycle counters of its RW_PROBE build.
// Build (from the repo root; the kernel needs its accumulators in VGPRs):
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -ffp-contract=off -mllvm -amdgpu-atomic-optimizer-strategy=None \
//         -mllvm -amdgpu-mfma-vgpr-form -DRW_PROBE [-DRW_ABLATE=n] -Iinclude -Isp_orb_slam_amd/csrc \
//         tools/microbench/conv_rw_probe.hip -o tools/microbench/bin/conv_rw_probe
// Run: conv_rw_probe H W B cout(128|512) pool(0|1) tile_rows(4|2) iters
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../sp_orb_slam_amd/csrc/conv_bf16_rw.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

int main(int argc, char **argv) {
  const int H = argc > 1 ? atoi(argv[1]) : 180, W = argc > 2 ? atoi(argv[2]) : 320, B = argc > 3 ? atoi(argv[3]) : 8;
  const int cout = argc > 4 ? atoi(argv[4]) : 128, pool = argc > 5 ? atoi(argv[5]) : 0, tr = argc > 6 ? atoi(argv[6]) : 4;
  const int iters = argc > 7 ? atoi(argv[7]) : 50;
  const int ncg = cout / 128, Ho = pool ? H / 2 : H, Wo = pool ? W / 2 : W;
  std::vector<unsigned short> in((size_t)B * H * W * 128);
  uint32_t x = 12345u;
  for (auto &v : in) { x = x * 1664525u + 1013904223u; v = (x & 0x80000000u) ? 0 : (unsigned short)(0x3e00u + ((x >> 8) & 0x1ffu)); }
  std::vector<unsigned short> wb((size_t)cout * 128 * 9);
  for (auto &v : wb) { x = x * 1664525u + 1013904223u; v = (unsigned short)(((x >> 31) << 15) | 0x3c00u | ((x >> 8) & 0x1ffu)); }
  std::vector<unsigned char> wp((size_t)ncg * spfe::conv_bf16_rw_weight_bytes());
  spfe::conv_bf16_rw_pack_weights(wb.data(), cout, wp.data());
  std::vector<float> bias(cout, 0.01f);
  void *d_in, *d_w, *d_out;
  float *d_b;
  int *d_ctr;
  CK(hipMalloc(&d_in, in.size() * 2));
  CK(hipMalloc(&d_w, wp.size()));
  CK(hipMalloc(&d_out, (size_t)B * Ho * Wo * cout * 2));
  CK(hipMalloc(&d_b, cout * 4));
  CK(hipMalloc(&d_ctr, 64 * 4));
  CK(hipMemcpy(d_in, in.data(), in.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_w, wp.data(), wp.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(d_b, bias.data(), cout * 4, hipMemcpyHostToDevice));
  spfe::ConvParams p{};
  p.in = (const float *)d_in; p.in_stride = 128; p.in_choff = 0;
  p.wpack = (const float *)d_w; p.bias = d_b;
  p.out = (float *)d_out; p.out_stride = cout; p.out_choff = 0; p.cout_real = cout;
  p.B = B; p.H = H; p.W = W;
  p.tiles_x = (W + 31) / 32; p.tiles_y = (H + tr - 1) / tr; p.nblk = ncg;
  p.num_cus = 256; p.tile_ctr = d_ctr;
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 5; ++w) { CK(hipMemsetAsync(d_ctr, 0, 64 * 4, s)); CK(spfe::launch_conv_bf16_rw(p, pool, tr, s)); }
  CK(hipStreamSynchronize(s));
#ifdef RW_PROBE
  unsigned long long zero[8] = {};
  CK(hipMemcpyToSymbol(HIP_SYMBOL(spfe::rw::rw_dbg), zero, sizeof(zero)));
#endif
  float tot = 0;
  for (int i = 0; i < iters; ++i) {
    CK(hipMemsetAsync(d_ctr, 0, 64 * 4, s));
    CK(hipEventRecord(e0, s));
    CK(spfe::launch_conv_bf16_rw(p, pool, tr, s));
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    tot += ms;
  }
  const double us = tot / iters * 1e3, flop = 2.0 * B * H * W * 128.0 * cout * 9;
  printf("conv_rw %dx%d B=%d cout=%d pool=%d rows=%d: %.2f us  %.1f TFLOP/s  %.3f of 2.5 PF\n", W, H, B, cout, pool, tr, us,
         flop / us * 1e-6, flop / us * 1e-6 / 2500.0);
#ifdef RW_PROBE
  unsigned long long d[8];
  CK(hipMemcpyFromSymbol(d, HIP_SYMBOL(spfe::rw::rw_dbg), sizeof(d)));
  const double tiles = (double)d[2];
  printf("  per tile: %.0f cycles (MFMA floor %d), %.0f of them from the end-of-tile wait to the barrier's release; prologue %.0f cycles per workgroup; "
         "shader clock %.2f GHz; tiles per launch %.0f\n",
         d[0] / tiles, 288 * 32 * tr / 4, d[1] / tiles, (double)d[4] / (iters * 256.0), d[0] / (d[3] * 10.0), tiles / iters);
#endif
  return 0;
}
