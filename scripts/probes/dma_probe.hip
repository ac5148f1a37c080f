// Probe of buffer_load_dwordx4 ... lds (LDS-DMA) semantics on gfx950: where the bytes land (M0, instruction offset,
// lane order), what an out-of-range lane writes, and whether the scalar offset moves the LDS side.
//   hipcc --offload-arch=gfx950 -O2 -o scripts/bin/dma_probe scripts/probes/dma_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4_t make_rsrc(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  u32x4_t r;
  r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  r[2] = __builtin_amdgcn_readfirstlane(bytes);
  r[3] = 0x00020000u;
  return r;
}

template <int IMM>
__device__ __forceinline__ void dma16(u32x4_t rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen offset:%5 lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff), "i"(IMM) : "memory");
}

__global__ void k(const unsigned* src, unsigned bytes, unsigned* out, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned lds[2048];   // 8 KB
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 2048; i += blockDim.x) lds[i] = 0xAAAA0000u + i;
  __syncthreads();
  const u32x4_t r = make_rsrc(src, bytes);
  const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned*)lds);
  unsigned voff = lane * 16;
  if (mode == 1 && lane >= 32) voff = 0x80000000u;          // out of range lanes
  if (mode == 4) voff = ((lane ^ 1) * 16);                  // permuted source, linear destination
  if (tid < 64) {
    if (mode == 0 || mode == 1 || mode == 4) dma16<0>(r, base + 1024, voff, 0);
    if (mode == 2) dma16<256>(r, base + 1024, voff, 0);       // instruction offset 256
    if (mode == 3) dma16<0>(r, base + 1024, voff, 512);       // scalar offset 512
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = tid; i < 2048; i += blockDim.x) out[i] = lds[i];
}

int main() {
  const int n = 4096;
  std::vector<unsigned> h(n);
  for (int i = 0; i < n; ++i) h[i] = i;
  unsigned *d, *o;
  hipMalloc(&d, n * 4); hipMalloc(&o, 2048 * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  std::vector<unsigned> r(2048);
  const char* names[] = {"plain", "lanes>=32 out of range", "inst offset 256", "soffset 512", "source permuted (lane^1)"};
  for (int mode = 0; mode < 5; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, 2048u, o, mode);
    hipMemcpy(r.data(), o, 2048 * 4, hipMemcpyDeviceToHost);
    printf("mode %d (%s): ", mode, names[mode]);
    // describe: for each dword of LDS that changed from the fill pattern, print runs
    int first = -1;
    for (int i = 0; i <= 2048; ++i) {
      const bool ch = i < 2048 && r[i] != 0xAAAA0000u + i;
      if (ch && first < 0) first = i;
      if (!ch && first >= 0) {
        printf("[lds dwords %d..%d <- src %u..%u] ", first, i - 1, r[first], r[i - 1]);
        first = -1;
      }
    }
    printf("\n   lds[256..263] = ");
    for (int i = 256; i < 264; ++i) printf("%x ", r[i]);
    printf(" lds[384..391] = ");
    for (int i = 384; i < 392; ++i) printf("%x ", r[i]);
    printf("\n");
  }
  return 0;
}
