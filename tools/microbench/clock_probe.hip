// clock_probe.hip — what shader clock does the chip sustain under a dense bf16 MFMA stream?
// One wave per SIMD on every CU issues back-to-back v_mfma_f32_32x32x16_bf16 (4 independent
// accumulators) on (a) zero operands, (b) random operands; prints s_memtime ticks / wall_clock64
// (100 MHz) = the shader clock, and the TFLOP/s.  Build: hipcc --offload-arch=gfx950 -O3 clock_probe.hip
// `clock_probe --json [device]` (bench.py, round 5): random operands only, bf16 (v_mfma_f32_32x32x16_bf16) AND f32
// (v_mfma_f32_32x32x2_f32), ~50 ms each after a warm-up launch, one JSON line: the MFMA rate the part SUSTAINS per dtype
// — what a kernel's fraction of the nominal-clock peak has to be read against on this box, at this moment.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
typedef float f32x16v __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_loop_f32(const float *src, float *sink, unsigned long long *clk, int iters) {
  const float a = src[threadIdx.x & 63], b = src[64 + (threadIdx.x & 63)];
  f32x16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, c3, 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
  float s = 0;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  sink[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { atomicAdd(&clk[0], t1 - t0); atomicAdd(&clk[1], w1 - w0); }
}
// --json: {"bf16": {"tflops", "ghz", "ms"}, "f32": {...}, "cus": n}
static int json_mode(int dev) {
  if (hipSetDevice(dev) != hipSuccess) { printf("{\"error\": \"hipSetDevice(%d)\"}\n", dev); return 1; }
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, dev);
  const int cus = prop.multiProcessorCount;
  uint4 *src; float *sink; unsigned long long *clk;
  hipMalloc(&src, 128 * 16); hipMalloc(&sink, (size_t)cus * 256 * 4); hipMalloc(&clk, 16);
  std::vector<unsigned> h(512);
  srand(1);
  double tf[2], ghz[2], msv[2];
  for (int t = 0; t < 2; ++t) {
    if (t == 0) for (auto &v : h) { unsigned e = 0x3f00 + (rand() & 0xff); unsigned f = 0xbf00 + (rand() & 0xff); v = e | (f << 16); }
    else for (auto &v : h) { const float x = (float)(rand() & 0xffff) / 65536.0f - 0.5f; v = __builtin_bit_cast(unsigned, x); }
    hipMemcpy(src, h.data(), 2048, hipMemcpyHostToDevice);
    const int iters = t == 0 ? 800000 : 400000;   // ~50 ms at 32 / 64 cycles per MFMA
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pass = 0; pass < 2; ++pass) {   // pass 0: warm-up (code load, clocks ramp), same length
      hipMemset(clk, 0, 16);
      hipEventRecord(e0);
      if (t == 0) hipLaunchKernelGGL(mfma_loop, dim3(cus), dim3(256), 0, 0, src, sink, clk, iters);
      else hipLaunchKernelGGL(mfma_loop_f32, dim3(cus), dim3(256), 0, 0, reinterpret_cast<const float *>(src), sink, clk, iters);
      hipEventRecord(e1);
      hipDeviceSynchronize();
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
    const double flop = (double)cus * 4 * iters * 4 * (t == 0 ? 32768.0 : 4096.0);
    tf[t] = flop / ms * 1e-9; ghz[t] = (double)c[0] / c[1] * 0.1; msv[t] = ms;
  }
  printf("{\"bf16\": {\"tflops\": %.1f, \"ghz\": %.3f, \"ms\": %.2f}, \"f32\": {\"tflops\": %.2f, \"ghz\": %.3f, \"ms\": %.2f}, \"cus\": %d, "
         "\"what\": \"bare MFMA loop on random operands, one wavefront per SIMD on every CU, operands in registers\"}\n",
         tf[0], ghz[0], msv[0], tf[1], ghz[1], msv[1], cus);
  return 0;
}
int main(int argc, char **argv) {
  if (argc > 1 && !strcmp(argv[1], "--json")) return json_mode(argc > 2 ? atoi(argv[2]) : 0);
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
