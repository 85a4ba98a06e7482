// Does v_cvt_pk_f32_fp8 turn the byte patterns 0x00..0x0f (an int4 code in the low nibble) into n * 2^-k exactly
// (fp8 e4m3 denormals + first binade are one linear ramp)?  Prints the 16 values and the word-select behaviour.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void probe(const uint32_t* in, float* out) {
  const uint32_t w = in[threadIdx.x];
  f2 lo = __builtin_amdgcn_cvt_pk_f32_fp8(w, false);     // bytes 0,1
  f2 hi = __builtin_amdgcn_cvt_pk_f32_fp8(w, true);      // bytes 2,3
  out[threadIdx.x * 4 + 0] = lo.x; out[threadIdx.x * 4 + 1] = lo.y;
  out[threadIdx.x * 4 + 2] = hi.x; out[threadIdx.x * 4 + 3] = hi.y;
}
int main() {
  uint32_t h[64]; float o[256];
  for (int i = 0; i < 64; ++i) h[i] = (uint32_t)(i & 15) | ((uint32_t)((i + 1) & 15) << 8) | ((uint32_t)((i + 2) & 15) << 16) | ((uint32_t)((i + 3) & 15) << 24);
  uint32_t* din; float* dout;
  hipMalloc(&din, 256); hipMalloc(&dout, 1024);
  hipMemcpy(din, h, 256, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(din, dout); hipDeviceSynchronize();
  hipMemcpy(o, dout, 1024, hipMemcpyDeviceToHost);
  for (int i = 0; i < 16; ++i)
    printf("n=%2d: byte0 -> %.10g (x512 = %g)  byte1(n+1) -> %.10g  byte2(n+2) -> %.10g  byte3(n+3) -> %.10g\n", i, o[i * 4], o[i * 4] * 512.0,
           o[i * 4 + 1], o[i * 4 + 2], o[i * 4 + 3]);
  return 0;
}
