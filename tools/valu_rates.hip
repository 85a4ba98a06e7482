// Issue-rate probe for the VALU instructions the dequantising kernels lean on (development tool, gfx950):
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/valu_rates tools/valu_rates.hip && tools/bin/valu_rates
// One wave per SIMD runs N independent copies of one instruction in a loop; cycles per instruction = dt / count.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP16(X) X X X X X X X X X X X X X X X X

template <int OP>
__global__ void __launch_bounds__(256) probe(unsigned long long* out, int iters) {
  float a0 = threadIdx.x * 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b0 = 1.0001f, b1 = 0.9999f;
  typedef __attribute__((ext_vector_type(2))) float f2;
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, q = {b0, b1};
  unsigned u0 = threadIdx.x * 2654435761u, u1 = u0 ^ 0x55aa55aa, u2 = u0 + 77, u3 = u1 + 99;
  const unsigned long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    if (OP == 0) { REP16(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));) }
    if (OP == 1) { REP16(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));) }
    if (OP == 2) { REP16(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));) }
    if (OP == 3) { REP16(asm volatile("v_cvt_f32_ubyte0 %0, %4\n v_cvt_f32_ubyte1 %1, %4\n v_cvt_f32_ubyte2 %2, %4\n v_cvt_f32_ubyte3 %3, %4" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(u0));) }
    if (OP == 4) { REP16(asm volatile("v_cvt_pk_bf16_f32 %0, %4, %5\n v_cvt_pk_bf16_f32 %1, %5, %4\n v_cvt_pk_bf16_f32 %2, %4, %4\n v_cvt_pk_bf16_f32 %3, %5, %5" : "=v"(u0), "=v"(u1), "=v"(u2), "=v"(u3) : "v"(a4), "v"(a5));) }
    if (OP == 5) { REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (OP == 6) { REP16(asm volatile("v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u0), "v"(u1));) }
    if (OP == 7) { REP16(asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u1));) }
    if (OP == 8) { REP16(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));) }
    if (OP == 9) { REP16(asm volatile("v_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n v_max_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n v_max_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n v_max_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
  }
  const unsigned long long t1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  if (a0 + a1 + a2 + a3 + p0.x + p1.y + p2.x + p3.y == 12345.678f || (u0 ^ u1 ^ u2 ^ u3) == 0xdeadbeefu) out[1] = 1;
}

template <int OP>
void run(const char* name, unsigned long long* d) {
  const int iters = 2000;
  probe<OP><<<256, 256>>>(d, iters);
  hipDeviceSynchronize();
  probe<OP><<<256, 256>>>(d, iters);
  hipDeviceSynchronize();
  unsigned long long h[2];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-22s %6.2f cycles per wave-instruction (1 wave per SIMD, 4 independent chains)\n", name, (double)h[0] / (iters * 64.0));
}

int main() {
  unsigned long long* d;
  hipMalloc(&d, 16);
  run<0>("v_mul_f32", d); run<8>("v_fma_f32", d); run<1>("v_pk_mul_f32", d); run<2>("v_pk_add_f32", d);
  run<3>("v_cvt_f32_ubyteN", d); run<4>("v_cvt_pk_bf16_f32", d); run<5>("v_exp_f32", d); run<6>("v_perm_b32", d);
  run<7>("v_and_b32", d); run<9>("v_max_f32_dpp row_ror", d);
  return 0;
}
