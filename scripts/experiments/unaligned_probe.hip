// Experiment: are unaligned 8 / 16-byte global loads and stores usable on gfx950 (AMDHSA enables unaligned access mode)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdint>
struct __attribute__((packed)) U16 { uint64_t a, b; };
__global__ void k(const uint8_t* in, uint8_t* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  U16 v;
  __builtin_memcpy(&v, in + 1 + 17 * i, 16);        // odd addresses
  v.a ^= 0x0101010101010101ull;
  __builtin_memcpy(out + 3 + 17 * i, &v, 16);
}
int main() {
  const int n = 4096;
  uint8_t *h = (uint8_t*)malloc(17 * n + 64), *ho = (uint8_t*)malloc(17 * n + 64), *d, *o;
  for (int i = 0; i < 17 * n + 64; ++i) h[i] = (uint8_t)(i * 7 + 3);
  hipMalloc(&d, 17 * n + 64); hipMalloc(&o, 17 * n + 64);
  hipMemcpy(d, h, 17 * n + 64, hipMemcpyHostToDevice); hipMemset(o, 0, 17 * n + 64);
  hipLaunchKernelGGL(k, dim3(n / 64), dim3(64), 0, 0, d, o, n);
  hipError_t e = hipDeviceSynchronize();
  printf("sync: %s\n", hipGetErrorString(e));
  hipMemcpy(ho, o, 17 * n + 64, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < 16; ++j) {
      uint8_t want = h[1 + 17 * i + j] ^ (j < 8 ? 1 : 0);
      if (ho[3 + 17 * i + j] != want) ++bad;
    }
  printf("mismatches: %d\n", bad);
  return bad != 0;
}
