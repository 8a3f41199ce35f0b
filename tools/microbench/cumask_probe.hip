// cumask_probe.hip — does hipExtStreamCreateWithCUMask confine a stream's workgroups on this stack, and how do mask
// bits map to (XCC, SE, CU)?   hipcc --offload-arch=gfx950 -O2 tools/microbench/cumask_probe.hip -o tools/microbench/bin/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
__global__ void who(unsigned *out, int spin) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  unsigned long long t0 = clock64();
  while (clock64() - t0 < (unsigned long long)spin) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
static int run(hipStream_t s, const char *name, unsigned *d, std::vector<unsigned> &h, int n) {
  hipLaunchKernelGGL(who, dim3(n), dim3(64), 0, s, d, 20000);
  CK(hipStreamSynchronize(s));
  CK(hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost));
  std::set<unsigned> cus;
  int per_xcc[16] = {};
  for (int i = 0; i < n; ++i) {
    const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;   // gfx9 HW_ID: cu_id[11:8] sh_id[12] se_id[15:13]
    const unsigned key = (xcc << 16) | (se << 8) | (sh << 4) | cu;
    if (cus.insert(key).second) per_xcc[xcc]++;
  }
  printf("%s: %zu distinct CUs; per XCC:", name, cus.size());
  for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
  printf("\n");
  return 0;
}
int main() {
  const int n = 8192;
  unsigned *d;
  CK(hipMalloc(&d, n * 8));
  std::vector<unsigned> h(2 * n);
  hipStream_t s0, s1, s2;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  unsigned m240[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x0000ffffu};
  unsigned m16[8] = {0, 0, 0, 0, 0, 0, 0, 0xffff0000u};
  hipError_t e1 = hipExtStreamCreateWithCUMask(&s1, 8, m240), e2 = hipExtStreamCreateWithCUMask(&s2, 8, m16);
  printf("create masked streams: %s / %s\n", hipGetErrorString(e1), hipGetErrorString(e2));
  if (run(s0, "unmasked", d, h, n)) return 2;
  if (e1 == hipSuccess && run(s1, "mask bits 0..239", d, h, n)) return 2;
  if (e2 == hipSuccess && run(s2, "mask bits 240..255", d, h, n)) return 2;
  return 0;
}
