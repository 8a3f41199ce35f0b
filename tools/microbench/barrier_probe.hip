// barrier_probe — how long does a workgroup's FIRST barrier take, by workgroup size and register footprint?
// Round 4 found that conv_f32_kc.hip's 512-thread form (193 VGPRs) spent ~19 us before its first barrier completed while
// workgroups that returned before the barrier finished in 5 us (HISTORY.md, "Round 4").  This probe isolates it: a kernel that
// does nothing but `nbar` barriers, with NREG registers kept live, for 256 / 512 / 1024 threads.
// build: hipcc --offload-arch=gfx950 -O3 tools/microbench/barrier_probe.hip -o tools/microbench/bin/barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int NT, int NREG>
__global__ __launch_bounds__(NT) void k(float *out, int nbar, int early) {
  float r[NREG];
#pragma unroll
  for (int i = 0; i < NREG; ++i) r[i] = (float)(threadIdx.x + i);
#pragma unroll
  for (int i = 0; i < NREG; ++i) asm volatile("" : "+v"(r[i]));
  if (early) { out[blockIdx.x * NT + threadIdx.x] = r[0]; return; }
  for (int b = 0; b < nbar; ++b) __syncthreads();
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < NREG; ++i) { asm volatile("" : "+v"(r[i])); s += r[i]; }
  out[blockIdx.x * NT + threadIdx.x] = s;
}

template <int NT, int NREG>
void run(float *d, int grid, int lds) {
  for (int early = 1; early >= 0; --early)
    for (int nbar : {1, 9}) {
      if (early && nbar == 9) continue;
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<NT, NREG>), dim3(grid), dim3(NT), lds, 0, d, nbar, early);
      hipDeviceSynchronize();
      float best = 1e9f;
      for (int rep = 0; rep < 20; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<NT, NREG>), dim3(grid), dim3(NT), lds, 0, d, nbar, early);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
      }
      printf("threads %4d  live regs %3d  grid %4d  lds %6d  %s: %.1f us\n", NT, NREG, grid, lds,
             early ? "return before the barrier" : (nbar == 1 ? "1 barrier                " : "9 barriers               "), best * 1e3f);
    }
}

// variants that add, one at a time, what conv_f32_kc.hip's producer / consumer build had and the kernel above has not
struct P { const float *in; float *out; int a, b, c, d, e, f, g, h; };
template <int MODE>
__global__ __launch_bounds__(512, 2) void k2(P p) {
  extern __shared__ __attribute__((aligned(16))) float sm2[];
  float r[150];
#pragma unroll
  for (int i = 0; i < 150; ++i) r[i] = (float)(threadIdx.x + i);
#pragma unroll
  for (int i = 0; i < 150; ++i) asm volatile("" : "+v"(r[i]));
  const bool producer = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) != 0;
  if (MODE == 1 && producer) return;                       // half the wavefronts leave before the barrier
  if (MODE == 2 && producer) {                             // role split: producers write LDS, then the barrier
    sm2[threadIdx.x & 255] = r[3];
    __syncthreads();
    return;
  }
  if (MODE == 3 && producer) {                             // ... producers load from global first
    sm2[threadIdx.x & 255] = p.in[(blockIdx.x * 256 + (threadIdx.x & 255)) & 0xffff];
    __syncthreads();
    return;
  }
  __syncthreads();
  float s = MODE >= 2 ? sm2[threadIdx.x & 255] : 0.0f;
#pragma unroll
  for (int i = 0; i < 150; ++i) { asm volatile("" : "+v"(r[i])); s += r[i]; }
  p.out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int MODE>
void run2(float *d, int grid, int lds, bool attr) {
  if (attr) hipFuncSetAttribute(reinterpret_cast<const void *>(k2<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  P p{d + (1 << 20), d, 1, 2, 3, 4, 5, 6, 7, 8};
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k2<MODE>), dim3(grid), dim3(512), lds, 0, p);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 20; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k2<MODE>), dim3(grid), dim3(512), lds, 0, p);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  printf("k2 mode %d (0 plain, 1 half the waves exit first, 2 role split + LDS, 3 + global load)  grid %d  lds %6d  attr %d: %.1f us\n", MODE, grid, lds, (int)attr, best * 1e3f);
}

int main(int argc, char **argv) {
  const int grid = argc > 1 ? atoi(argv[1]) : 240;
  float *d;
  hipMalloc(&d, (size_t)4096 * 1024 * 4);
  run<256, 16>(d, grid, 0);  run<256, 120>(d, grid, 0);  run<256, 180>(d, grid, 0);
  run<512, 16>(d, grid, 0);  run<512, 60>(d, grid, 0);   run<512, 120>(d, grid, 0);  run<512, 180>(d, grid, 0);
  run<1024, 16>(d, grid, 0); run<1024, 60>(d, grid, 0);  run<1024, 100>(d, grid, 0);
  run<512, 180>(d, grid, 65536);
  run<512, 180>(d, 64, 0);
  run2<0>(d, grid, 1024, false); run2<0>(d, grid, 69664, true);
  run2<1>(d, grid, 69664, true); run2<2>(d, grid, 69664, true); run2<3>(d, grid, 69664, true);
  hipFree(d);
  return 0;
}
