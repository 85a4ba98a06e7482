// Development probe (not part of the product): what HBM bandwidth do the paged-KV access
// patterns of the decode kernel reach on MI355X, independent of any compute?
//   hipcc --offload-arch=gfx950 -O3 tools/probe_gather.hip -o tools/bin/probe_gather
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <random>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// mode 0: contiguous stream; 1: 1 KiB row per wave-instruction; 2: 4 rows x 256 B (V pattern,
// head segment w of 4 by wave); 3: 16 rows x 64 B x 4 instr (K pattern)
template <int MODE, bool NT, int UNROLL>
__global__ void __launch_bounds__(256) gather_kernel(const uint4* __restrict__ kv, const int* __restrict__ slots,
                                                     int tokens_per_wg, float* out) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wg = blockIdx.x;
  const int* row = slots + (size_t)wg * tokens_per_wg;
  float acc = 0.f;
  typedef __attribute__((ext_vector_type(4))) unsigned int u4v;
  auto ld = [&](const uint4* p) -> uint4 {
    if (NT) return __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u4v*>(p)));
    return *p;
  };
  if (MODE == 0) {
    // each WG streams tokens_per_wg KiB contiguous
    const uint4* base = kv + (size_t)wg * tokens_per_wg * 64;
    for (int t = w * UNROLL; t < tokens_per_wg; t += 4 * UNROLL) {
      uint4 r[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) r[u] = ld(base + (size_t)(t + u) * 64 + lane);
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) acc += __builtin_bit_cast(float, r[u].x ^ r[u].y ^ r[u].z ^ r[u].w);
    }
  } else if (MODE == 1) {
    // wave w takes tokens t with t%4==w; full 1 KiB row per instruction
    for (int t = w * UNROLL; t < tokens_per_wg; t += 4 * UNROLL) {
      uint4 r[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) r[u] = ld(kv + (size_t)row[t + u] * 64 + lane);
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) acc += __builtin_bit_cast(float, r[u].x ^ r[u].y ^ r[u].z ^ r[u].w);
    }
  } else if (MODE == 2) {
    // wave w reads head segment w (256 B) of 4 tokens per instruction
    const int tq = lane >> 4, dc = lane & 15;
    for (int t = 0; t < tokens_per_wg; t += 4 * UNROLL) {
      uint4 r[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) r[u] = ld(kv + (size_t)row[t + u * 4 + tq] * 64 + w * 16 + dc);
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) acc += __builtin_bit_cast(float, r[u].x ^ r[u].y ^ r[u].z ^ r[u].w);
    }
  } else {
    // K pattern: lane (n = lane&15, j = lane>>4), 4 instructions of 16 tokens x 64 B
    const int n = lane & 15, j = lane >> 4;
    for (int t = 0; t < tokens_per_wg; t += 16 * UNROLL) {
      uint4 r[UNROLL][4];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const uint4* p = kv + (size_t)row[t + u * 16 + n] * 64 + w * 16 + j;
#pragma unroll
        for (int c = 0; c < 4; ++c) r[u][c] = ld(p + c * 4);
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc += __builtin_bit_cast(float, r[u][c].x ^ r[u][c].y ^ r[u][c].z ^ r[u][c].w);
    }
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

constexpr int NSETS = 6;   // distinct pools cycled per launch: nothing is re-served by the 256 MB MALL

template <int MODE, bool NT, int UNROLL>
void run(const char* name, uint4* const* kvs, const int* slots, int n_wg, int tpw, float* out, double bytes) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) gather_kernel<MODE, NT, UNROLL><<<n_wg, 256>>>(kvs[i % NSETS], slots, tpw, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int iters = 24;
  for (int i = 0; i < iters; ++i) gather_kernel<MODE, NT, UNROLL><<<n_wg, 256>>>(kvs[i % NSETS], slots, tpw, out);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-46s wg=%5d tok/wg=%4d : %8.2f us  %6.3f TB/s\n", name, n_wg, tpw, ms * 1e3 / iters, bytes / (ms * 1e-3 / iters) / 1e12);
}

int main() {
  const size_t n_slots = 64 * 4224 * 2 + 4096;   // K and V pools of the B=64 bench, as one pool
  uint4* kv[NSETS]; int* slots; float* out;
  for (int i = 0; i < NSETS; ++i) {
    CK(hipMalloc(&kv[i], n_slots * 1024));
    fill_random<<<2048, 256>>>(kv[i], n_slots * 64, 11u + i);
  }
  CK(hipMalloc(&out, 4));
  std::vector<int> perm(n_slots);
  for (size_t i = 0; i < n_slots; ++i) perm[i] = (int)i;
  std::mt19937 rng(1);
  std::shuffle(perm.begin(), perm.end(), rng);
  const size_t n_tok = 64 * 4224 * 2;
  CK(hipMalloc(&slots, n_tok * 4));
  CK(hipMemcpy(slots, perm.data(), n_tok * 4, hipMemcpyHostToDevice));
  const double bytes = (double)n_tok * 1024;
  for (int tpw : {128, 256, 512}) {
    const int n_wg = (int)(n_tok / tpw);
    run<0, false, 8>("contiguous stream, unroll 8", kv, slots, n_wg, tpw, out, bytes);
    run<0, true, 8>("contiguous stream, unroll 8, nt", kv, slots, n_wg, tpw, out, bytes);
    run<1, false, 8>("random 1KiB rows (1 row/instr), unroll 8", kv, slots, n_wg, tpw, out, bytes);
    run<1, true, 8>("random 1KiB rows (1 row/instr), unroll 8, nt", kv, slots, n_wg, tpw, out, bytes);
    run<1, false, 2>("random 1KiB rows (1 row/instr), unroll 2", kv, slots, n_wg, tpw, out, bytes);
    run<2, false, 8>("V pattern 4 rows x 256B, unroll 8", kv, slots, n_wg, tpw, out, bytes);
    run<2, true, 8>("V pattern 4 rows x 256B, unroll 8, nt", kv, slots, n_wg, tpw, out, bytes);
    run<2, false, 4>("V pattern 4 rows x 256B, unroll 4", kv, slots, n_wg, tpw, out, bytes);
    run<3, false, 2>("K pattern 16 rows x 64B x4, unroll 2", kv, slots, n_wg, tpw, out, bytes);
    run<3, true, 2>("K pattern 16 rows x 64B x4, unroll 2, nt", kv, slots, n_wg, tpw, out, bytes);
    run<3, false, 4>("K pattern 16 rows x 64B x4, unroll 4", kv, slots, n_wg, tpw, out, bytes);
  }
  return 0;
}
