// Experiment: where do 2 s of `prepare` go on a fresh box?  Times, in a fresh process, the pieces a
// first bind + prepare of the 10 M-aggregate log pays once: device allocations of the log's size (first and second time
// round: is VRAM that was never handed out slower?), the first kernel launch, the first rocPRIM call, hipFree.
//   hipcc --offload-arch=gfx950 -O3 vram_probe.hip -o vram_probe && ./vram_probe
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <chrono>
#include <cstdio>
#include <cstdint>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(uint4* p, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = make_uint4(1, 2, 3, 4);
}
int main() {
  double t0 = now();
  hipFree(0);
  printf("runtime init                      %9.2f ms\n", now() - t0);
  hipStream_t st; hipStreamCreate(&st);
  const int64_t GB = 1ll << 30;
  for (int round = 0; round < 3; ++round) {
    for (int64_t gb : {1ll, 8ll, 74ll}) {
      void* p = nullptr;
      t0 = now();
      hipError_t e = hipMalloc(&p, gb * GB);
      const double t_alloc = now() - t0;
      if (e != hipSuccess) { printf("hipMalloc %lld GB failed: %s\n", (long long)gb, hipGetErrorString(e)); continue; }
      t0 = now();
      hipLaunchKernelGGL(touch, dim3(256 * 8), dim3(256), 0, st, (uint4*)p, gb * GB / 16);
      hipStreamSynchronize(st);
      const double t_touch = now() - t0;
      t0 = now();
      hipLaunchKernelGGL(touch, dim3(256 * 8), dim3(256), 0, st, (uint4*)p, gb * GB / 16);
      hipStreamSynchronize(st);
      const double t_touch2 = now() - t0;
      t0 = now();
      hipFree(p);
      printf("round %d  %3lld GB: hipMalloc %9.2f ms  first write pass %9.2f ms  second %9.2f ms  hipFree %9.2f ms\n", round, (long long)gb, t_alloc, t_touch,
             t_touch2, now() - t0);
    }
  }
  // two live 74 GB allocations, as bench.py holds them (the CSR log + the tile-major copy)
  void *a = nullptr, *b = nullptr;
  t0 = now(); hipMalloc(&a, 74 * GB); double ta = now() - t0;
  t0 = now(); hipMalloc(&b, 74 * GB); double tb = now() - t0;
  printf("two live 74 GB allocations: %9.2f ms, %9.2f ms\n", ta, tb);
  t0 = now();
  hipLaunchKernelGGL(touch, dim3(256 * 8), dim3(256), 0, st, (uint4*)b, 74 * GB / 16);
  hipStreamSynchronize(st);
  printf("first write pass over the second: %9.2f ms\n", now() - t0);
  hipFree(a); hipFree(b);
  // first rocPRIM radix sort of the process (code-object load) vs the second
  const size_t n = 10'000'000;
  uint32_t *k0, *k1, *v0, *v1; hipMalloc(&k0, n * 4); hipMalloc(&k1, n * 4); hipMalloc(&v0, n * 4); hipMalloc(&v1, n * 4);
  hipMemset(k0, 7, n * 4);
  for (int rep = 0; rep < 3; ++rep) {
    t0 = now();
    size_t tb2 = 0; void* tmp = nullptr;
    rocprim::radix_sort_pairs_desc(nullptr, tb2, k0, k1, v0, v1, n, 0u, 16u, st);
    hipMalloc(&tmp, tb2);
    rocprim::radix_sort_pairs_desc(tmp, tb2, k0, k1, v0, v1, n, 0u, 16u, st);
    hipStreamSynchronize(st);
    hipFree(tmp);
    printf("rocPRIM radix_sort_pairs_desc of 10 M pairs, call %d: %9.2f ms (wall, incl. its scratch malloc / free)\n", rep, now() - t0);
  }
  return 0;
}
