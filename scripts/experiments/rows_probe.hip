// Experiment: HBM efficiency of the "rows" access pattern (64 lanes <-> 64 aggregates, 256 B pieces at
// a 4 KiB row stride) versus the linear tile stream, both through global_load_lds nt.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int ROWS>
__global__ void __launch_bounds__(64) probe(const uint4* __restrict__ ev, int64_t n_agg, int L, int aggs_per_task, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  const int64_t a0 = (int64_t)blockIdx.x * aggs_per_task;
  uint32_t acc = 0;
  const int chunks = L / 16;
  if (ROWS) {
    // 64 aggregates per pass; chunk c of every row
    for (int64_t g = a0; g < a0 + aggs_per_task; g += 64) {
      for (int c = 0; c < chunks; ++c) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint4 v = *(const uint4*)(smem + lane * 256 + ((c & 15) * 16));
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int64_t row = g + 4 * q + (lane >> 4);
          const int cc = ROWS == 2 ? (int)((c + (row & 15)) % chunks) : c;  // ROWS == 2: rows desynchronised by up to 15 chunks
          const uint4* src = ev + (row * L + cc * 16 + (lane & 15));
          __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + q * 1024), 16, 0, 2);
        }
      }
    }
  } else {
    const int64_t e0 = a0 * L, e1 = (a0 + aggs_per_task) * L;
    for (int64_t e = e0; e < e1; e += 1024) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      uint4 v = *(const uint4*)(smem + lane * 256);
      acc ^= v.x ^ v.y ^ v.z ^ v.w;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int q = 0; q < 16; ++q)
        __builtin_amdgcn_global_load_lds((gptr_t)(ev + e + q * 64 + lane), (lptr_t)(smem + q * 1024), 16, 0, 2);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
  const int64_t A = 1000000 / 64 * 64; const int L = 256;
  const size_t bytes = (size_t)A * L * 16;
  uint4* d; uint32_t* sink;
  hipMalloc(&d, bytes); hipMalloc(&sink, 4);
  hipMemset(d, 1, bytes);
  const bool random_fill = true;
  if (random_fill) {
    std::vector<uint64_t> h(bytes / 8);
    uint64_t x = 88172645463325252ull;
    for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = x; }
    hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int lds_kb : {16}) {
    for (int rows = 0; rows < 3; ++rows) {
      for (int apt : {64, 128, 256}) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
          hipEventRecord(e0);
          if (rows == 2) hipLaunchKernelGGL(probe<2>, dim3(A / apt), dim3(64), lds_kb * 1024, 0, d, A, L, apt, sink);
          else if (rows) hipLaunchKernelGGL(probe<1>, dim3(A / apt), dim3(64), lds_kb * 1024, 0, d, A, L, apt, sink);
          else hipLaunchKernelGGL(probe<0>, dim3(A / apt), dim3(64), lds_kb * 1024, 0, d, A, L, apt, sink);
          hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("lds=%dKB rows=%d aggs_per_task=%d: %.3f ms  %.1f GB/s\n", lds_kb, rows, apt, best, bytes / best / 1e6);
      }
    }
  }
  return 0;
}
