// Micro-benchmark: cycles per v_mfma_f32_32x32x2_f32 for one wave per SIMD, with
// and without LDS operand reads between the MFMAs.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_f32_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void probe(float *out, long long *cyc, int iters) {
  __shared__ float lds[16384];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (float)(i & 7);
  __syncthreads();
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  float x0 = threadIdx.x, x1 = 1.f, y0 = 2.f, y1 = 3.f;
  const float *p = lds + (threadIdx.x & 63);
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (MODE >= 1) {  // operand reads for the NEXT step, issued before this step's MFMAs
        x0 = p[(s * 72) & 8191]; x1 = p[((s * 72) & 8191) + 36];
        y0 = p[8192 + s * 128]; y1 = p[8192 + s * 128 + 32];
        __builtin_amdgcn_sched_barrier(0);
      }
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y0, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y1, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y0, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y1, a3, 0, 0, 0);
      if (MODE >= 1) __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = clock64();
  float r = 0;
  for (int i = 0; i < 16; ++i) r += a0[i] + a1[i] + a2[i] + a3[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int WAVES>
void run(const char *name, int blocks) {
  float *out; long long *cyc;
  hipMalloc(&out, blocks * 64 * WAVES * 4); hipMalloc(&cyc, blocks * 8);
  const int iters = 2000;
  hipLaunchKernelGGL((probe<MODE, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<MODE, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[4]; hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
  const double nm = (double)iters * 64;  // MFMAs per wave
  printf("%-34s blocks %4d waves/blk %d: %.2f cycles/MFMA/wave, %.3f ms, %.1f TFLOP/s\n", name, blocks, WAVES,
         h[0] / nm, ms, 2.0 * 32 * 32 * 2 * nm * WAVES * blocks / (ms * 1e-3) / 1e12);
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0, 4>("mfma only, 1 wave/SIMD", 256);
  run<1, 4>("mfma + lds operand reads, 1 w/SIMD", 256);
  run<0, 8>("mfma only, 2 waves/SIMD", 256);
  run<1, 8>("mfma + lds operand reads, 2 w/SIMD", 256);
  return 0;
}
