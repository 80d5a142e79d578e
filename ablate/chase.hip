// Probe: dependent random loads (latency) vs footprint, at the frontier kernel's concurrency.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
__global__ void chase(const uint32_t* __restrict__ next, size_t n_lines, int hops, int lanes_per_wave, uint32_t* out, long long* cyc) {
  const int lane = threadIdx.x & 63;
  if (lane >= lanes_per_wave) return;
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  size_t cur = ((wave * 64 + lane) * 2654435761ull) % n_lines;
  const long long t0 = clock64();
  for (int i = 0; i < hops; ++i) cur = next[cur * 16];   // one 64-byte line per hop
  const long long t1 = clock64();
  out[wave * 64 + lane] = (uint32_t)cur;
  if (lane == 0 && wave == 0) *cyc = t1 - t0;
}
__global__ void fill(uint32_t* next, size_t n_lines, uint64_t seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_lines; i += (size_t)gridDim.x * blockDim.x) {
    uint64_t x = (i + seed) * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    next[i * 16] = (uint32_t)(x % n_lines);
  }
}
int main() {
  const int hops = 64;
  for (double gb : {0.25, 1.0, 8.0, 32.0, 128.0, 240.0}) {
    size_t bytes = (size_t)(gb * (1ull << 30)), n_lines = bytes / 64;
    if (n_lines > 0xFFFFFFFFull) n_lines = 0xFFFFFFFFull;
    uint32_t* next; if (hipMalloc(&next, n_lines * 64) != hipSuccess) { printf("%.1f GB alloc failed\n", gb); continue; }
    uint32_t* out; hipMalloc(&out, 4096 * 4 * 64 * 4); long long* cyc; hipMalloc(&cyc, 8);
    fill<<<4096, 256>>>(next, n_lines, 12345); hipDeviceSynchronize();
    for (int waves : {1, 3072}) for (int lpw : {1, 8, 64}) {
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      const int blocks = waves == 1 ? 1 : waves / 4, threads = waves == 1 ? 64 : 256;
      chase<<<blocks, threads>>>(next, n_lines, hops, lpw, out, cyc); hipDeviceSynchronize();
      hipEventRecord(a); chase<<<blocks, threads>>>(next, n_lines, hops, lpw, out, cyc); hipEventRecord(b); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, a, b); long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      printf("%6.2f GB  waves %4d lanes/wave %2d: %.3f us per hop (wall/hops), wave0 %lld cyc/hop\n", gb, waves, lpw, ms * 1e3 / hops, c / hops);
    }
    hipFree(next); hipFree(out); hipFree(cyc);
  }
  return 0;
}
