// Development probe: how many VALU instructions fit beside one v_mfma_f32_32x32x16_bf16 on a gfx950 SIMD, with one and
// with two waves per SIMD?  Each wave loops over { 1 MFMA (4 independent accumulators in rotation) + N independent
// v_fma_f32 }; prints shader cycles (s_memtime) per MFMA per wave.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_mix.hip -o tools/bin/mfma_valu_mix
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int N, int KIND, bool MFMA = true>
__global__ void __launch_bounds__(1024) mix(float* out, long long* cyc, int iters) {
  f32x16_t acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = threadIdx.x * 0.001f + r;
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.01f + i); b[i] = (__bf16)(1.0f + i * 0.5f); }
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x + i * 0.25f;
  const float c1 = 1.0001f, c2 = 0.0003f;
  typedef __attribute__((ext_vector_type(2))) float f2;
  f2 vp[8]; for (int i = 0; i < 8; ++i) vp[i] = f2{v[i], v[i + 8]};
  const f2 cp1 = {c1, c1}, cp2 = {c2, c2};
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (MFMA) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < N; ++j) {
        float& x = v[(m * N + j) % 16];
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2));
        else if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
        else if (KIND == 2) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2));
        else if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(vp[(m * N + j) % 8]) : "v"(cp1), "v"(cp2));
        else if (KIND == 4) asm volatile("v_cvt_pk_f32_fp8 %0, %1" : "=v"(vp[(m * N + j) % 8]) : "v"(x));
        else if (KIND == 5) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(x) : "v"(vp[(m * N + j) % 8].x), "v"(vp[(m * N + j) % 8].y));
        else if (KIND == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(vp[(m * N + j) % 8]) : "v"(cp1));
        else if (KIND == 7) asm volatile("v_and_b32 %0, 0x0f0f0f0f, %0" : "+v"(x));
        else if (KIND == 8) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2));
        else if (KIND == 9) asm volatile("v_lshrrev_b32 %0, 4, %0" : "+v"(x));
        else asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(vp[(m * N + j) % 8]) : "v"(cp1));
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float sum = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) sum += acc[i][r];
  for (int i = 0; i < 16; ++i) sum += v[i];
  for (int i = 0; i < 8; ++i) sum += vp[i].x + vp[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int N, int KIND, bool MFMA = true>
int run(const char* kind, int waves, float* out, long long* cyc) {
  const int iters = 2000;
  mix<N, KIND, MFMA><<<1, 64 * waves>>>(out, cyc, iters);
  CK(hipDeviceSynchronize());
  long long h[16];
  CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
  double mx = 0;
  for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
  if (MFMA) printf("%-10s waves/WG %2d  VALU per MFMA %2d : %7.1f cycles per MFMA (slowest wave)\n", kind, waves, N, mx / (iters * 4.0));
  else printf("%-10s waves/WG %2d  no MFMA, %2d VALU per group : %6.2f cycles per VALU per wave, %5.2f per VALU per SIMD\n", kind, waves, N,
              mx / (iters * 4.0 * N), mx / (iters * 4.0 * N) / (waves / 4.0));
  return 0;
}

int main() {
  float* out; long long* cyc;
  CK(hipMalloc(&out, 1024 * 4)); CK(hipMalloc(&cyc, 128));
  for (int waves : {4, 8}) {
    run<0, 0>("v_fma", waves, out, cyc); run<2, 0>("v_fma", waves, out, cyc); run<4, 0>("v_fma", waves, out, cyc);
    run<6, 0>("v_fma", waves, out, cyc); run<8, 0>("v_fma", waves, out, cyc); run<12, 0>("v_fma", waves, out, cyc);
    run<16, 0>("v_fma", waves, out, cyc);
    run<4, 1>("v_exp", waves, out, cyc); run<8, 1>("v_exp", waves, out, cyc);
    run<8, 2>("v_max3", waves, out, cyc);
  }
  for (int waves : {4, 8, 16}) {
    run<8, 0, false>("v_fma", waves, out, cyc);
    run<8, 1, false>("v_exp", waves, out, cyc);
    run<8, 2, false>("v_max3", waves, out, cyc);
    run<8, 3, false>("v_pk_fma", waves, out, cyc);
  }
  for (int waves : {4, 8}) { run<4, 3>("v_pk_fma", waves, out, cyc); run<8, 3>("v_pk_fma", waves, out, cyc); }
  // instruction kinds of the int4 dequantisation (KIVI stage 1)
  for (int waves : {4, 8, 12}) {
    run<8, 4, false>("cvt_pk_f32_fp8", waves, out, cyc);
    run<8, 5, false>("cvt_pk_bf16_f32", waves, out, cyc);
    run<8, 6, false>("v_pk_mul", waves, out, cyc);
    run<8, 10, false>("v_pk_add", waves, out, cyc);
    run<8, 7, false>("v_and", waves, out, cyc);
    run<8, 9, false>("v_lshrrev", waves, out, cyc);
    run<8, 8, false>("v_perm", waves, out, cyc);
  }
  return 0;
}
