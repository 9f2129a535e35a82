// Probe (GPU box): semantics of global_load_lds_dwordx4 on gfx950 as used by the BANDED kernel:
// lane L's 16 bytes must land at (M0 base) + 16*L; several DMAs in flight; vmcnt(N) ordering.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned char* p, uint4* out, int nt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const unsigned wbase = __builtin_amdgcn_groupstaticsize() + __builtin_amdgcn_readfirstlane(tid >> 6) * 1024u;
  unsigned keep;
  for (int b = 0; b < 3; ++b) {
    const unsigned char* g = p + (size_t)b * 4096 + 16 * tid;
    const unsigned base = wbase + b * 4096u;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(base) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
  out[tid] = *(uint4*)(smem + 16 * ((tid + 65) & 255));
  asm volatile("s_waitcnt vmcnt(1)\n\ts_barrier" ::: "memory");
  out[256 + tid] = *(uint4*)(smem + 4096 + 16 * ((tid + 65) & 255));
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  out[512 + tid] = *(uint4*)(smem + 8192 + 16 * ((tid + 65) & 255));
}
int main() {
  const int N = 3 * 4096;
  std::vector<unsigned char> h(N);
  for (int i = 0; i < N; ++i) h[i] = (unsigned char)((i * 7 + (i >> 8)) & 0xff);
  unsigned char* d; uint4* o;
  hipMalloc(&d, N); hipMalloc(&o, N);
  hipMemcpy(d, h.data(), N, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 3 * 4096, 0, d, o, 0);
  std::vector<unsigned char> r(N);
  hipMemcpy(r.data(), o, N, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int b = 0; b < 3; ++b)
    for (int t = 0; t < 256; ++t)
      for (int j = 0; j < 16; ++j) {
        const int src = b * 4096 + 16 * ((t + 65) & 255) + j;
        if (r[b * 4096 + 16 * t + j] != h[src]) ++bad;
      }
  printf("dma_probe bad=%d\n", bad);
  return bad != 0;
}
