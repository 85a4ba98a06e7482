// Development probe (not part of the product): HBM bandwidth of the decode kernel's paged K/V gather when the K
// tile is brought in by LDS-DMA (global_load_lds_dwordx4: 4 token rows x 256 B per wave-instruction, full cache
// lines, so the non-temporal policy applies) and read back from LDS in the MFMA B-operand layout, next to V loaded
// straight to registers (4 rows x 256 B, nt).  All vector-memory instructions of the loop are inline asm with
// hand-counted s_waitcnt (hipcc drains vmcnt(0) before any ds_read while a builtin LDS-DMA is in flight).  No compute.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_kdma.hip -o tools/bin/probe_kdma
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <random>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned int u4v;

template <bool NT>
__device__ __forceinline__ void dma16(const void* g, unsigned lds_addr) {
  unsigned keep;
  if (NT)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_addr) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ u4v ldv_nt(const void* g) {
  u4v r;
  asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(r) : "v"(g) : "memory");
  return r;
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// One workgroup = 4 waves = the 4 kv heads of `tokens_per_wg` tokens (like the decode kernel).  TT tokens per stage,
// S stages per wave.  K: LDS-DMA ring; V (VMODE 1): registers, nt.
template <int TT, int S, bool NT, int VMODE>
__global__ void __launch_bounds__(256) kdma_kernel(const uint4* __restrict__ kc, const uint4* __restrict__ vc,
                                                   const int* __restrict__ slots, int tokens_per_wg, float* out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int* row = slots + (size_t)blockIdx.x * tokens_per_wg;
  int* slot_lds = reinterpret_cast<int*>(lds);                    // [tokens_per_wg]
  for (int i = threadIdx.x; i < tokens_per_wg; i += 256) slot_lds[i] = row[i];
  __syncthreads();
  const unsigned ring = (unsigned)(tokens_per_wg * 4 + w * (S * TT * 256));   // LDS byte address of this wave's ring
  const char* ringp = lds + ring;
  const int n = lane & 15, jq = lane >> 4;
  const int rq = lane >> 4, pc = lane & 15;          // DMA: row-in-instruction, 16-byte position
  constexpr int NI = TT / 4;                          // DMA instructions per stage
  constexpr int NV = VMODE ? TT / 4 : 0;              // V loads per tile
  const int ntiles = tokens_per_wg / TT;
  float acc = 0.f;
  auto issue = [&](int tile, int stage) {
    const uint4* src[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int r = i * 4 + rq;
      // LDS row r, position pc holds chunk pc ^ (r & 15)  (source-side swizzle, lane-linear destination)
      src[i] = kc + (size_t)slot_lds[tile * TT + r] * 64 + w * 16 + (pc ^ (r & 15));
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) dma16<NT>(src[i], ring + stage * (TT * 256) + i * 1024);
  };
#pragma unroll
  for (int s = 0; s < S - 1; ++s) issue(s, s);
  // steady state: per tile the queue sees V(t) x NV, then DMA(t+S-1) x NI
  for (int t = 0; t < ntiles; ++t) {
    u4v vr[NV > 0 ? NV : 1];
    const uint4* vsrc[NV > 0 ? NV : 1];
#pragma unroll
    for (int e = 0; e < NV; ++e) vsrc[e] = vc + (size_t)slot_lds[t * TT + e * 4 + rq] * 64 + w * 16 + pc;
#pragma unroll
    for (int e = 0; e < NV; ++e) vr[e] = ldv_nt(vsrc[e]);
    if (t + S - 1 < ntiles) issue(t + S - 1, (t + S - 1) % S);
    else {
#pragma unroll
      for (int i = 0; i < NI; ++i) asm volatile("s_nop 0");       // keep the count uniform: wait for everything below
    }
    // stage t's DMA is older than: V(t-S+2..t) and DMA(t+1..t+S-1)
    if (t + S - 1 < ntiles) wait_vm<(S - 1) * (NI + NV)>();
    else wait_vm<0>();
    const char* st = ringp + (t % S) * (TT * 256);
#pragma unroll
    for (int g = 0; g < TT / 16; ++g)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int r = g * 16 + n;
        const uint4 k = *reinterpret_cast<const uint4*>(st + r * 256 + (((c * 4 + jq) ^ (r & 15)) << 4));
        acc += __builtin_bit_cast(float, k.x ^ k.y ^ k.z ^ k.w);
      }
    if (NV > 0) {
      if (t + S - 1 < ntiles) wait_vm<NI>(); else wait_vm<0>();
#pragma unroll
      for (int e = 0; e < NV; ++e) {
        asm volatile("" : "+v"(vr[e]));
        acc += __builtin_bit_cast(float, vr[e].x ^ vr[e].y ^ vr[e].z ^ vr[e].w);
      }
    }
    // the stage is overwritten by the DMA issued in the next iteration: all ds_reads must have returned
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (acc == 123.456f) out[0] = acc;
}

// random payload: constant bytes make the same kernels look ~10 % faster (less switching power -> higher clocks)
__global__ void fill_random(uint4* p, size_t n, unsigned seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u ^ seed;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
    p[i] = make_uint4(x, x * 0x9E3779B1u, x ^ 0x85EBCA6Bu, x * 0xC2B2AE35u + 7u);
  }
}

constexpr int NSETS = 6;   // distinct K/V pools cycled per launch: nothing is re-served by the 256 MB MALL

template <int TT, int S, bool NT, int VMODE>
void run(const char* name, uint4* const* kcs, uint4* const* vcs, const int* slots, size_t n_tok, int tpw, float* out) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int n_wg = (int)(n_tok / tpw);
  const size_t shm = 4 * S * TT * 256 + tpw * 4;
  CK(hipFuncSetAttribute((const void*)kdma_kernel<TT, S, NT, VMODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  for (int i = 0; i < 3; ++i) kdma_kernel<TT, S, NT, VMODE><<<n_wg, 256, shm>>>(kcs[i % NSETS], vcs[i % NSETS], slots, tpw, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int iters = 24;
  for (int i = 0; i < iters; ++i) kdma_kernel<TT, S, NT, VMODE><<<n_wg, 256, shm>>>(kcs[i % NSETS], vcs[i % NSETS], slots, tpw, out);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)n_tok * 1024 * (VMODE ? 2 : 1);
  printf("%-48s wg=%5d tok/wg=%4d lds=%3zuK : %8.2f us  %6.3f TB/s\n", name, n_wg, tpw, shm >> 10, ms * 1e3 / iters,
         bytes / (ms * 1e-3 / iters) / 1e12);
}

int main() {
  const size_t n_tok = 64 * 4224;
  const size_t n_slots = n_tok + 4096;
  uint4 *kc[NSETS], *vc[NSETS]; int* slots; float* out;
  for (int i = 0; i < NSETS; ++i) {
    CK(hipMalloc(&kc[i], n_slots * 1024)); CK(hipMalloc(&vc[i], n_slots * 1024));
    fill_random<<<2048, 256>>>(kc[i], n_slots * 64, 11u + i); fill_random<<<2048, 256>>>(vc[i], n_slots * 64, 77u + i);
  }
  CK(hipMalloc(&out, 4));
  std::vector<int> perm(n_slots);
  for (size_t i = 0; i < n_slots; ++i) perm[i] = (int)i;
  std::mt19937 rng(1);
  std::shuffle(perm.begin(), perm.end(), rng);
  CK(hipMalloc(&slots, n_tok * 4));
  CK(hipMemcpy(slots, perm.data(), n_tok * 4, hipMemcpyHostToDevice));
  for (int tpw : {1056, 528, 264}) {
    if (tpw % 32) continue;
    run<32, 2, false, 0>("K only: DMA 32-tok x2", kc, vc, slots, n_tok, tpw, out);
    run<32, 2, true, 0>("K only: DMA 32-tok x2, nt", kc, vc, slots, n_tok, tpw, out);
    run<32, 3, true, 0>("K only: DMA 32-tok x3, nt", kc, vc, slots, n_tok, tpw, out);
    run<32, 4, true, 0>("K only: DMA 32-tok x4, nt", kc, vc, slots, n_tok, tpw, out);
    run<16, 4, true, 0>("K only: DMA 16-tok x4, nt", kc, vc, slots, n_tok, tpw, out);
    run<32, 2, false, 1>("K DMA 32-tok x2 + V regs nt", kc, vc, slots, n_tok, tpw, out);
    run<32, 2, true, 1>("K DMA 32-tok x2 nt + V regs nt", kc, vc, slots, n_tok, tpw, out);
    run<32, 3, true, 1>("K DMA 32-tok x3 nt + V regs nt", kc, vc, slots, n_tok, tpw, out);
    run<32, 4, true, 1>("K DMA 32-tok x4 nt + V regs nt", kc, vc, slots, n_tok, tpw, out);
    run<16, 4, true, 1>("K DMA 16-tok x4 nt + V regs nt", kc, vc, slots, n_tok, tpw, out);
    run<16, 3, true, 1>("K DMA 16-tok x3 nt + V regs nt", kc, vc, slots, n_tok, tpw, out);
    run<16, 2, true, 1>("K DMA 16-tok x2 nt + V regs nt", kc, vc, slots, n_tok, tpw, out);
  }
  return 0;
}
