// Development probe: does the instruction offset of global_load_lds_dwordx4 move the LDS destination, the global
// source, or both?   hipcc --offload-arch=gfx950 -O3 tools/probe_dma_offset.hip -o tools/bin/probe_dma_offset
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)

__global__ void __launch_bounds__(64) probe(const uint32_t* src, uint32_t* out) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[4096];    // 16 KiB
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  const uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)lds);
  const uint32_t voff = threadIdx.x * 16;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:2048\n\ts_waitcnt vmcnt(0)"
               :: "v"(voff), "s"(src), "s"(base + 4096) : "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 4096; i += 64) out[i] = lds[i];
}

int main() {
  uint32_t *src, *out;
  CK(hipMalloc(&src, 1 << 20)); CK(hipMalloc(&out, 16384));
  static uint32_t h[1 << 18];
  for (int i = 0; i < (1 << 18); ++i) h[i] = i * 4;               // value = byte offset
  CK(hipMemcpy(src, h, 1 << 20, hipMemcpyHostToDevice));
  probe<<<1, 64>>>(src, out);
  CK(hipDeviceSynchronize());
  static uint32_t r[4096];
  CK(hipMemcpy(r, out, 16384, hipMemcpyDeviceToHost));
  int first = -1, last = -1;
  for (int i = 0; i < 4096; ++i) if (r[i] != 0xdeadbeefu) { if (first < 0) first = i; last = i; }
  printf("M0 = base + 4096, voffset = lane*16, offset:2048\n");
  printf("LDS bytes written: [%d, %d); first value (global byte offset read) = %u\n", first * 4, last * 4 + 4, first >= 0 ? r[first] : 0);
  return 0;
}
