// Probe: a hop whose K loads per lane go to ONE 128-byte region (K pieces of 16 bytes, as the pair line of the blocked
// heap is read) against K loads to K random lines (chase_k.hip), and dependent chains of J single loads (3072 waves x
// 8 lanes): is the frontier kernel's time set by the lines a lane touches, by the load instructions it issues, or
// by the dependent round trips?  (diagnostic, never shipped)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
// MODE 0: K x 16-byte loads from one random 128-byte region per hop; MODE 1: K x 4-byte loads from K random lines;
// MODE 2: K DEPENDENT single 4-byte loads per hop (K round trips, one load each); MODE 3: K x 4-byte loads from ONE line
template <int K, int MODE>
__global__ void chase(const uint32_t* __restrict__ next, size_t n_lines, int hops, uint32_t* out) {
  const int lane = threadIdx.x & 63;
  if (lane >= 8) return;
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  size_t cur = ((wave * 64 + lane) * 2654435761ull) % n_lines;
  for (int i = 0; i < hops; ++i) {
    uint32_t mix = 0;
    if (MODE == 0) {
      const uint4* p = reinterpret_cast<const uint4*>(next + (cur & ~(size_t)1) * 16);
      uint4 v[K];
#pragma unroll
      for (int k = 0; k < K; ++k) v[k] = p[k];
#pragma unroll
      for (int k = 0; k < K; ++k) mix ^= v[k].x ^ v[k].w;
    } else if (MODE == 1) {
      uint32_t v[K];
#pragma unroll
      for (int k = 0; k < K; ++k) v[k] = next[((cur + (size_t)k * 7919u * 40503u) % n_lines) * 16];
#pragma unroll
      for (int k = 0; k < K; ++k) mix ^= v[k];
    } else if (MODE == 2) {
      size_t c = cur;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        mix = next[c * 16];
        c = (mix + k) % n_lines;
      }
    } else {
      const uint32_t* p = next + cur * 16;
      uint32_t v[K];
#pragma unroll
      for (int k = 0; k < K; ++k) v[k] = p[k];
#pragma unroll
      for (int k = 0; k < K; ++k) mix ^= v[k];
    }
    cur = (size_t)(mix + i) % n_lines;
  }
  out[wave * 64 + lane] = (uint32_t)cur;
}
__global__ void fill(uint32_t* next, size_t n_words, uint64_t seed, size_t n_lines) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x) {
    uint64_t x = (i + seed) * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    next[i] = (uint32_t)(x % n_lines);
  }
}
template <int K, int MODE>
void run(const uint32_t* next, size_t n_lines, uint32_t* out, const char* what) {
  const int hops = 40;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  chase<K, MODE><<<768, 256>>>(next, n_lines, hops, out); hipDeviceSynchronize();
  hipEventRecord(a); chase<K, MODE><<<768, 256>>>(next, n_lines, hops, out); hipEventRecord(b); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("K=%2d %-46s: %.3f us per hop\n", K, what, ms * 1e3 / hops); fflush(stdout);
}
int main() {
  const size_t n_lines = (size_t)(8ull << 30) / 64;   // 8 GB of 64-byte lines, every word filled
  uint32_t* next; if (hipMalloc(&next, n_lines * 64) != hipSuccess) return 1;
  uint32_t* out; hipMalloc(&out, 4096 * 64 * 4);
  fill<<<8192, 256>>>(next, n_lines * 16, 12345, n_lines - 2); hipDeviceSynchronize();
  run<1, 0>(next, n_lines, out, "x 16 B from one 128-byte region"); run<2, 0>(next, n_lines, out, "x 16 B from one 128-byte region");
  run<4, 0>(next, n_lines, out, "x 16 B from one 128-byte region"); run<8, 0>(next, n_lines, out, "x 16 B from one 128-byte region");
  run<1, 3>(next, n_lines, out, "x 4 B from one 64-byte line"); run<4, 3>(next, n_lines, out, "x 4 B from one 64-byte line");
  run<8, 3>(next, n_lines, out, "x 4 B from one 64-byte line"); run<16, 3>(next, n_lines, out, "x 4 B from one 64-byte line");
  run<1, 1>(next, n_lines, out, "x 4 B from K random lines"); run<4, 1>(next, n_lines, out, "x 4 B from K random lines");
  run<8, 1>(next, n_lines, out, "x 4 B from K random lines");
  run<2, 2>(next, n_lines, out, "dependent single loads (K round trips)"); run<3, 2>(next, n_lines, out, "dependent single loads (K round trips)");
  run<4, 2>(next, n_lines, out, "dependent single loads (K round trips)"); run<8, 2>(next, n_lines, out, "dependent single loads (K round trips)");
  return 0;
}
