// Probe: a hop that waits for K independent random 64-byte lines per lane (3072 waves x 8 lanes): how does the
// time per hop grow with K, and with a store to a random line in front of every hop?  (diagnostic, never shipped)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int K, bool ST>
__global__ void chase(const uint32_t* __restrict__ next, uint32_t* scratch, size_t n_lines, int hops, uint32_t* out) {
  const int lane = threadIdx.x & 63;
  if (lane >= 8) return;
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  size_t cur[K];
#pragma unroll
  for (int k = 0; k < K; ++k) cur[k] = ((wave * 64 + lane) * 2654435761ull + k * 40503ull * 7919ull) % n_lines;
  for (int i = 0; i < hops; ++i) {
    uint32_t v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = next[cur[k] * 16];
    uint32_t mix = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) mix ^= v[k];   // every next address depends on ALL loads of this hop
#pragma unroll
    for (int k = 0; k < K; ++k) cur[k] = (v[k] + (mix & 1) + k * 977u) % n_lines;
    if (ST) scratch[cur[0] * 16 + 1] = mix;
  }
  out[wave * 64 + lane] = (uint32_t)cur[0];
}
__global__ void fill(uint32_t* next, size_t n_lines, uint64_t seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_lines; i += (size_t)gridDim.x * blockDim.x) {
    uint64_t x = (i + seed) * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    next[i * 16] = (uint32_t)(x % n_lines);
  }
}
template <int K, bool ST>
void run(const uint32_t* next, uint32_t* scratch, size_t n_lines, uint32_t* out) {
  const int hops = 40;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  chase<K, ST><<<768, 256>>>(next, scratch, n_lines, hops, out); hipDeviceSynchronize();
  hipEventRecord(a); chase<K, ST><<<768, 256>>>(next, scratch, n_lines, hops, out); hipEventRecord(b); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("K=%2d loads per hop%s: %.3f us per hop\n", K, ST ? " + 1 store" : "          ", ms * 1e3 / hops); fflush(stdout);
}
int main() {
  const size_t n_lines = (size_t)(32ull << 30) / 64;
  uint32_t* next; if (hipMalloc(&next, n_lines * 64) != hipSuccess) return 1;
  uint32_t* out; hipMalloc(&out, 4096 * 64 * 4);
  fill<<<4096, 256>>>(next, n_lines, 12345); hipDeviceSynchronize();
  run<1, false>(next, next, n_lines, out); run<2, false>(next, next, n_lines, out); run<4, false>(next, next, n_lines, out);
  run<8, false>(next, next, n_lines, out); run<14, false>(next, next, n_lines, out); run<30, false>(next, next, n_lines, out);
  run<1, true>(next, next, n_lines, out); run<4, true>(next, next, n_lines, out); run<14, true>(next, next, n_lines, out);
  return 0;
}
