// mfma_chain_probe.hip — how fast does ONE dependent accumulation chain run on the f32 MFMA shapes, and are the shapes
// with K = 4 k-ordered fmaf chains like v_mfma_f32_32x32x2_f32 (include/spfe_exact_math.h)?
//   part 1: cycles per instruction for 1, 2, 4 independent chains of 32x32x2, 16x16x4, 4x4x1 (one wave per SIMD)
//   part 2: D = A x B + C with random operands against the host's fmaf chain in ascending k, bitwise
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off mfma_chain_probe.hip -o bin/mfma_chain_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int NCH>
__global__ __launch_bounds__(256) void chain(const float *src, float *sink, unsigned long long *clk, int iters) {
  const float a = src[threadIdx.x & 63], b = src[64 + (threadIdx.x & 63)];
  f32x16 c32[4] = {};
  f32x4 c16[4] = {};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        if constexpr (SHAPE == 0) c32[ch] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c32[ch], 0, 0, 0);
        if constexpr (SHAPE == 1) c16[ch] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c16[ch], 0, 0, 0);
        if constexpr (SHAPE == 2) c16[ch] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c16[ch], 0, 0, 0);
      }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int ch = 0; ch < 4; ++ch) { for (int r = 0; r < 16; ++r) s += c32[ch][r]; for (int r = 0; r < 4; ++r) s += c16[ch][r]; }
  sink[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

// one 16x16x4 product: A[16][4], B[4][16], C[16][16]; lane l: A[l % 16][l / 16], B[l / 16][l % 16]; D reg r of lane l:
// row 4 (l / 16) + r, column l % 16
__global__ void one_16x16x4(const float *A, const float *B, const float *C, float *D) {
  const int l = threadIdx.x;
  f32x4 c;
  for (int r = 0; r < 4; ++r) c[r] = C[(4 * (l / 16) + r) * 16 + l % 16];
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(l % 16) * 4 + l / 16], B[(l / 16) * 16 + l % 16], c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * (l / 16) + r) * 16 + l % 16] = c[r];
}

int main() {
  float *src, *sink; unsigned long long *clk;
  hipMalloc(&src, 128 * 4); hipMalloc(&sink, 256 * 256 * 4); hipMalloc(&clk, 16);
  std::vector<float> h(128);
  for (auto &v : h) v = (float)(rand() % 2000 - 1000) / 1024.0f;
  hipMemcpy(src, h.data(), 512, hipMemcpyHostToDevice);
  const int iters = 2000;
  auto run = [&](auto k, const char *name, int nch) {
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, src, sink, clk, 100);
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, src, sink, clk, iters);
    hipDeviceSynchronize();
    unsigned long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    // (s_memtime ticks = shader cycles here: a 16-pass v_mfma_f32_32x32x2_f32 reads 64)
    printf("%-10s %d chain(s): %.2f cycles per MFMA\n", name, nch, (double)c / (iters * 8.0 * nch));
  };
  run(chain<0, 1>, "32x32x2", 1); run(chain<0, 2>, "32x32x2", 2); run(chain<0, 4>, "32x32x2", 4);
  run(chain<1, 1>, "16x16x4", 1); run(chain<1, 2>, "16x16x4", 2); run(chain<1, 4>, "16x16x4", 4);
  run(chain<2, 1>, "4x4x1", 1); run(chain<2, 2>, "4x4x1", 2); run(chain<2, 4>, "4x4x1", 4);

  // part 2: exactness of 16x16x4
  std::vector<float> A(64), B(64), C(256), D(256);
  int bad_asc = 0, bad_desc = 0;
  float *dA, *dB, *dC, *dD;
  hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dC, 1024); hipMalloc(&dD, 1024);
  for (int trial = 0; trial < 200; ++trial) {
    for (auto &v : A) v = ldexpf((float)(rand() % 20000 - 10000) / 7.0f, rand() % 20 - 10);
    for (auto &v : B) v = ldexpf((float)(rand() % 20000 - 10000) / 3.0f, rand() % 20 - 10);
    for (auto &v : C) v = ldexpf((float)(rand() % 20000 - 10000) / 11.0f, rand() % 24 - 12);
    hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(one_16x16x4, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        float up = C[i * 16 + j], dn = C[i * 16 + j];
        for (int k = 0; k < 4; ++k) up = fmaf(A[i * 4 + k], B[k * 16 + j], up);
        for (int k = 3; k >= 0; --k) dn = fmaf(A[i * 4 + k], B[k * 16 + j], dn);
        unsigned u, d, g;
        memcpy(&u, &up, 4); memcpy(&d, &dn, 4); memcpy(&g, &D[i * 16 + j], 4);
        bad_asc += u != g; bad_desc += d != g;
      }
  }
  printf("16x16x4 vs fmaf chain k = 0..3: %d mismatches of %d; vs k = 3..0: %d\n", bad_asc, 200 * 256, bad_desc);
  return 0;
}
