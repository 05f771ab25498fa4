// Experiment: does a second, register-staged tile per wave (32 KiB in flight instead of 16 KiB) raise the read-stream
// ceiling of the LDS-DMA transport?   hipcc --offload-arch=gfx950 -O3 probe2.hip -o probe2 && ./probe2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ntload(const uint4* p) { const v4u v = __builtin_nontemporal_load((const v4u*)p); return make_uint4(v.x, v.y, v.z, v.w); }

template <int MODE>  // 0: DMA only (16 KiB/iter), 1: DMA + 16 register loads (32 KiB/iter), 2: register loads only (16 KiB/iter), 3: 32 reg loads
__global__ void __launch_bounds__(64) probe(const uint4* __restrict__ src, int64_t n_vec, int64_t vec_per_wave, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  const int64_t v0 = (int64_t)blockIdx.x * vec_per_wave;
  int64_t v1 = v0 + vec_per_wave; v1 = v1 < n_vec ? v1 : n_vec;
  uint32_t acc = 0;
  uint4 r[32];
  for (int i = 0; i < 32; ++i) r[i] = make_uint4(0, 0, 0, 0);
  const int64_t step = MODE == 1 ? 2048 : (MODE == 3 ? 2048 : 1024);
  for (int64_t v = v0; v + step <= v1; v += step) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == 0 || MODE == 1) { const uint4 x = *(const uint4*)(smem + lane * 256); acc ^= x.x ^ x.y ^ x.z ^ x.w; }
    if (MODE != 0) {
#pragma unroll
      for (int q = 0; q < (MODE == 3 ? 32 : 16); ++q) acc ^= r[q].x ^ r[q].w;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (MODE == 0 || MODE == 1) {
#pragma unroll
      for (int q = 0; q < 16; ++q)
        __builtin_amdgcn_global_load_lds((gptr_t)(src + v + q * 64 + lane), (lptr_t)(smem + q * 1024), 16, 0, 2);
    }
    if (MODE == 1) {
#pragma unroll
      for (int q = 0; q < 16; ++q) r[q] = ntload(&src[v + 1024 + q * 64 + lane]);
    }
    if (MODE == 2) {
#pragma unroll
      for (int q = 0; q < 16; ++q) r[q] = ntload(&src[v + q * 64 + lane]);
    }
    if (MODE == 3) {
#pragma unroll
      for (int q = 0; q < 32; ++q) r[q] = ntload(&src[v + q * 64 + lane]);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int q = 0; q < 32; ++q) acc ^= r[q].y;
  if (acc == 0x9e3779b9u) sink[0] = acc;
}

int main() {
  const int64_t bytes = 8ll << 30, n_vec = bytes / 16;
  uint4* d; uint32_t* sink;
  hipMalloc(&d, bytes); hipMemset(d, 1, bytes); hipMalloc(&sink, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wpc : {4, 6, 8, 9, 12}) {
    for (int mode = 0; mode < 4; ++mode) {
      const int64_t waves = 256ll * wpc;
      int64_t per = (n_vec / waves) / 2048 * 2048;
      float best = 1e9;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(waves), dim3(64), 16384, 0, d, n_vec, per, sink);
        if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(waves), dim3(64), 16384, 0, d, n_vec, per, sink);
        if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(waves), dim3(64), 16384, 0, d, n_vec, per, sink);
        if (mode == 3) hipLaunchKernelGGL(probe<3>, dim3(waves), dim3(64), 16384, 0, d, n_vec, per, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      printf("waves/CU %2d mode %d (%s): %.3f ms %.0f GB/s\n", wpc, mode,
             mode == 0 ? "LDS-DMA 16K" : mode == 1 ? "DMA 16K + regs 16K" : mode == 2 ? "regs 16K" : "regs 32K", best,
             (double)(per * waves * 16) / best / 1e6);
    }
  }
  return 0;
}
