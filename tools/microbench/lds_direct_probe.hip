// lds_direct_probe.hip — semantics probe for `buffer_load_dwordx4 ... lds` on gfx950:
// (1) lane l of a wave writes LDS[M0 + 16*l .. +16); (2) an out-of-range buffer offset writes
// zeros (what the conv halo padding relies on); (3) pre-existing LDS content is overwritten.
//   hipcc --offload-arch=gfx950 -O3 -o lds_direct_probe lds_direct_probe.hip && ./lds_direct_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
__global__ void probe(const unsigned* g, unsigned nbytes, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned sm[4 * 256];
  for (int i = threadIdx.x; i < 1024; i += 256) sm[i] = 0xdeadbeefu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(g), 0, nbytes, 0x00020000);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // lane reads global piece (63 - lane) of its wave's 1 KiB; every 5th lane is out of range
  unsigned voff = (unsigned)(wave * 1024 + (63 - lane) * 16);
  if (lane % 5 == 4) voff = 0x80000000u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)((char*)sm + wave * 1024), 16, voff, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 256) out[i] = sm[i];
}
int main() {
  std::vector<unsigned> h(1024), o(1024);
  for (int i = 0; i < 1024; ++i) h[i] = 1000 + i;
  unsigned *dg, *dout;
  hipMalloc(&dg, 4096); hipMalloc(&dout, 4096);
  hipMemcpy(dg, h.data(), 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, dg, 4096u, dout);
  hipMemcpy(o.data(), dout, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int w = 0; w < 4; ++w)
    for (int l = 0; l < 64; ++l)
      for (int c = 0; c < 4; ++c) {
        const unsigned got = o[w * 256 + l * 4 + c];
        const unsigned want = (l % 5 == 4) ? 0u : 1000 + w * 256 + (63 - l) * 4 + c;
        if (got != want && bad++ < 8) printf("wave %d lane %d c %d: got %u want %u\n", w, l, c, got, want);
      }
  printf("lds_direct_probe: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
  return bad != 0;
}
