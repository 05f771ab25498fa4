// Experiment (second part): a 74 GB hipMalloc that normally returns in 0.3 ms took 6.3 s once
// (vram_probe.hip: the third allocate / free round).  Is it VRAM that was freed moments ago — the kernel driver scrubs
// freed VRAM before handing it out again — and does waiting after the free make the next allocation fast again?
//   hipcc --offload-arch=gfx950 -O3 vram_probe2.hip -o vram_probe2 && ./vram_probe2
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipFree(0);
  size_t fr = 0, tot = 0;
  hipMemGetInfo(&fr, &tot);
  printf("device memory: %.1f GB free of %.1f GB\n", fr / 1e9, tot / 1e9);
  const size_t GB = 1ull << 30;
  // use every byte of fresh VRAM once: three 74 GB blocks live (222 of ~288 GB), then free them all
  void* p[3];
  for (int i = 0; i < 3; ++i) { double t0 = now(); hipMalloc(&p[i], 74 * GB); printf("fresh 74 GB #%d: %9.2f ms\n", i, now() - t0); }
  for (int i = 0; i < 3; ++i) hipFree(p[i]);
  for (int wait_ms : {0, 0, 200, 1000, 3000, 0}) {
    std::this_thread::sleep_for(std::chrono::milliseconds(wait_ms));
    void* a = nullptr; void* b = nullptr; void* c = nullptr;
    double t0 = now();
    hipMalloc(&a, 74 * GB); double ta = now() - t0; t0 = now();
    hipMalloc(&b, 74 * GB); double tb = now() - t0; t0 = now();
    hipMalloc(&c, 74 * GB); double tc = now() - t0;
    printf("after freeing 222 GB and waiting %4d ms: three 74 GB allocations take %9.2f %9.2f %9.2f ms\n", wait_ms, ta, tb, tc);
    t0 = now();
    hipFree(a); hipFree(b); hipFree(c);
    printf("   freeing them: %9.2f ms\n", now() - t0);
  }
  return 0;
}
