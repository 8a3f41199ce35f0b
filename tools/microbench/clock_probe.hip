// clock_probe.hip — what shader clock does the chip sustain under a dense bf16 MFMA stream?
// One wave per SIMD on every CU issues back-to-back v_mfma_f32_32x32x16_bf16 (4 independent
// accumulators) on (a) zero operands, (b) random operands; prints s_memtime ticks / wall_clock64
// (100 MHz) = the shader clock, and the TFLOP/s.  Build: hipcc --offload-arch=gfx950 -O3 clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void mfma_loop(const uint4 *src, float *sink, unsigned long long *clk, int iters) {
  const uint4 va = src[threadIdx.x & 63], vb = src[64 + (threadIdx.x & 63)];
  const bf16x8 a = __builtin_bit_cast(bf16x8, va), b = __builtin_bit_cast(bf16x8, vb);
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, b, c3, 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
  float s = 0;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  sink[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { atomicAdd(&clk[0], t1 - t0); atomicAdd(&clk[1], w1 - w0); }
}
int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  uint4 *src; float *sink; unsigned long long *clk;
  hipMalloc(&src, 128 * 16); hipMalloc(&sink, 256 * 256 * 4); hipMalloc(&clk, 16);
  for (int mode = 0; mode < 3; ++mode) {
    std::vector<unsigned> h(512);
    for (auto &v : h) {
      if (mode == 0) v = 0;
      else if (mode == 1) v = 0x3f803f80u;                          // all ones (bf16 1.0)
      else { unsigned e = 0x3f00 + (rand() & 0xff); unsigned f = 0xbf00 + (rand() & 0xff); v = e | (f << 16); }  // random mantissas, mixed signs
    }
    hipMemcpy(src, h.data(), 2048, hipMemcpyHostToDevice);
    hipMemset(clk, 0, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop, dim3(256), dim3(256), 0, 0, src, sink, clk, 1000);
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop, dim3(256), dim3(256), 0, 0, src, sink, clk, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
    const double flop = 256.0 * 4 * iters * 4 * 32768.0;
    printf("mode %d (%s): %.3f ms, %.0f TFLOP/s, shader clock %.3f GHz, cycles per MFMA %.2f\n", mode,
           mode == 0 ? "zeros" : mode == 1 ? "ones" : "random", ms, flop / ms * 1e-9, (double)c[0] / c[1] * 0.1,
           (double)c[0] / 256.0 / (1000.0 + iters) / 4);
  }
  return 0;
}
