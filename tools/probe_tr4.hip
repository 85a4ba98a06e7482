#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(const uint32_t* in, uint32_t* out, int mode) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = in[i];
  __syncthreads();
  uint32_t a = (uint32_t)(uintptr_t)lds;   // LDS byte address base
  uint32_t addr = a + (mode == 0 ? threadIdx.x * 8 : mode == 1 ? 0 : (threadIdx.x & 15) * 8 + (threadIdx.x >> 4) * 128);
  uint64_t v;
  asm volatile("ds_read_b64_tr_b4 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[threadIdx.x * 2] = (uint32_t)v;
  out[threadIdx.x * 2 + 1] = (uint32_t)(v >> 32);
}
__global__ void probe8(const uint32_t* in, uint32_t* out, int mode) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = in[i];
  __syncthreads();
  uint32_t a = (uint32_t)(uintptr_t)lds;
  uint32_t addr = a + (mode == 0 ? threadIdx.x * 8 : 0);
  uint64_t v;
  asm volatile("ds_read_b64_tr_b8 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[threadIdx.x * 2] = (uint32_t)v;
  out[threadIdx.x * 2 + 1] = (uint32_t)(v >> 32);
}
int main() {
  uint32_t h[1024];
  // nibble k of the LDS image (nibble index = 8*dword + pos) holds a tag: we use two passes to recover the 16-bit nibble index
  uint32_t *din, *dout; hipMalloc(&din, 4096); hipMalloc(&dout, 64 * 8);
  uint32_t res[4][128];
  for (int mode = 0; mode < 1; ++mode) {
    for (int pass = 0; pass < 4; ++pass) {      // pass p: nibble = (index >> 4p) & 15
      for (int w = 0; w < 1024; ++w) { uint32_t x = 0; for (int p = 0; p < 8; ++p) { uint32_t idx = w * 8 + p; x |= ((idx >> (4 * pass)) & 15u) << (4 * p); } h[w] = x; }
      hipMemcpy(din, h, 4096, hipMemcpyHostToDevice);
      probe<<<1, 64>>>(din, dout, mode); hipDeviceSynchronize();
      hipMemcpy(res[pass], dout, 512, hipMemcpyDeviceToHost);
    }
    printf("tr_b4 mode %d: lane: source nibble index of each of the 16 result nibbles\n", mode);
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d:", l);
      for (int e = 0; e < 16; ++e) {
        uint32_t idx = 0;
        for (int pass = 0; pass < 4; ++pass) { uint32_t wv = res[pass][l * 2 + (e >> 3)]; idx |= ((wv >> (4 * (e & 7))) & 15u) << (4 * pass); }
        printf(" %4u", idx);
      }
      printf("\n");
    }
  }
  // tr_b8: byte index recovered in 2 passes
  uint32_t r8[2][128];
  for (int pass = 0; pass < 2; ++pass) {
    for (int w = 0; w < 1024; ++w) { uint32_t x = 0; for (int p = 0; p < 4; ++p) { uint32_t idx = w * 4 + p; x |= ((idx >> (8 * pass)) & 255u) << (8 * p); } h[w] = x; }
    hipMemcpy(din, h, 4096, hipMemcpyHostToDevice);
    probe8<<<1, 64>>>(din, dout, 0); hipDeviceSynchronize();
    hipMemcpy(r8[pass], dout, 512, hipMemcpyDeviceToHost);
  }
  printf("tr_b8: lane: source byte index of each of the 8 result bytes\n");
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int e = 0; e < 8; ++e) { uint32_t idx = 0; for (int pass = 0; pass < 2; ++pass) idx |= ((r8[pass][l * 2 + (e >> 2)] >> (8 * (e & 3))) & 255u) << (8 * pass); printf(" %4u", idx); }
    printf("\n");
  }
  return 0;
}
