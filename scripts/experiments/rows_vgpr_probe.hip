// Experiment: the "rows" access pattern (lane l <-> aggregate l of a 64-aggregate group, 128 B per lane per
// tile at a 4 KiB row stride) landed directly in VGPRs (global_load_dwordx4 nt, 8 per tile) with DEPTH tiles
// in flight per wave, versus the LDS-DMA tile stream the fold kernels use (capped by 160 KB LDS per CU).
// Question: does more bytes in flight buy bandwidth beyond ~6 TB/s?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint4 ldnt(const uint4* p) {
  uint4 v;
  v.x = __builtin_nontemporal_load(&p->x); v.y = __builtin_nontemporal_load(&p->y);
  v.z = __builtin_nontemporal_load(&p->z); v.w = __builtin_nontemporal_load(&p->w);
  return v;
}

template <int DEPTH>
__global__ void __launch_bounds__(64) probe(const uint4* __restrict__ ev, int L, int groups_per_wave, uint32_t* sink) {
  const int lane = threadIdx.x;
  const int chunks = L / 8;
  const int64_t g0 = (int64_t)blockIdx.x * groups_per_wave;
  const int n_tiles = groups_per_wave * chunks;
  uint4 buf[DEPTH][8];
  uint32_t acc = 0;
  auto src = [&](int t) {
    const int g = t / chunks, c = t - g * chunks;
    return ev + (((g0 + g) * 64 + lane) * (int64_t)L + c * 8);
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    if (d < n_tiles) {
      const uint4* s = src(d);
#pragma unroll
      for (int q = 0; q < 8; ++q) buf[d][q] = ldnt(s + q);
    }
  for (int t0 = 0; t0 < n_tiles; t0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int t = t0 + d;
      if (t < n_tiles) {
#pragma unroll
        for (int q = 0; q < 8; ++q) acc ^= buf[d][q].x ^ buf[d][q].y ^ buf[d][q].z ^ buf[d][q].w;
        if (t + DEPTH < n_tiles) {
          const uint4* s = src(t + DEPTH);
#pragma unroll
          for (int q = 0; q < 8; ++q) buf[d][q] = ldnt(s + q);
        }
      }
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
  const int64_t A = 1000000 / 64 * 64; const int L = 256;
  const size_t bytes = (size_t)A * L * 16;
  uint4* d; uint32_t* sink;
  hipMalloc(&d, bytes); hipMalloc(&sink, 4);
  {
    std::vector<uint64_t> h(bytes / 8);
    uint64_t x = 88172645463325252ull;
    for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = x; }
    hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int64_t groups = A / 64;
  for (int depth : {1, 2, 3, 4}) {
    for (int gpw : {1, 2, 4, 8}) {
      float best = 1e9;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        const unsigned grid = (unsigned)(groups / gpw);
        if (depth == 1) hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(64), 0, 0, d, L, gpw, sink);
        if (depth == 2) hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(64), 0, 0, d, L, gpw, sink);
        if (depth == 3) hipLaunchKernelGGL(probe<3>, dim3(grid), dim3(64), 0, 0, d, L, gpw, sink);
        if (depth == 4) hipLaunchKernelGGL(probe<4>, dim3(grid), dim3(64), 0, 0, d, L, gpw, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      printf("vgpr-landing depth=%d groups_per_wave=%d: %.3f ms  %.0f GB/s\n", depth, gpw, best, bytes / best / 1e6);
    }
  }
  return 0;
}
