// Probe: does a latency-bound kernel (dependent random loads, 8 lanes per wave) overlap with a bandwidth-bound
// kernel (streaming copy) (a) on two HIP streams, (b) as two roles of one launch?  (diagnostic, never shipped)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ void chase_role(const uint32_t* __restrict__ next, size_t n_lines, int hops, int wave,
                                           uint32_t* out) {
  const int lane = threadIdx.x & 63;
  if (lane >= 8) return;
  size_t cur = (((size_t)wave * 64 + lane) * 2654435761ull) % n_lines;
  for (int i = 0; i < hops; ++i) cur = next[cur * 16];
  out[(size_t)wave * 64 + lane] = (uint32_t)cur;
}
__device__ __forceinline__ void copy_role(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4, int blk,
                                          int nblk) {
  for (size_t i = (size_t)blk * blockDim.x + threadIdx.x; i < n4; i += (size_t)nblk * blockDim.x) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void chase_k(const uint32_t* next, size_t n_lines, int hops, uint32_t* out) {
  chase_role(next, n_lines, hops, blockIdx.x * 4 + (threadIdx.x >> 6), out);
}
__global__ __launch_bounds__(256) void copy_k(const float4* src, float4* dst, size_t n4) {
  copy_role(src, dst, n4, blockIdx.x, gridDim.x);
}
__global__ __launch_bounds__(256) void fused_k(const uint32_t* next, size_t n_lines, int hops, uint32_t* out,
                                               int chase_blocks, const float4* src, float4* dst, size_t n4) {
  if ((int)blockIdx.x < chase_blocks)
    chase_role(next, n_lines, hops, blockIdx.x * 4 + (threadIdx.x >> 6), out);
  else
    copy_role(src, dst, n4, blockIdx.x - chase_blocks, gridDim.x - chase_blocks);
}
__global__ void fill(uint32_t* next, size_t n_lines, uint64_t seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_lines; i += (size_t)gridDim.x * blockDim.x) {
    uint64_t x = (i + seed) * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    next[i * 16] = (uint32_t)(x % n_lines);
  }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main() {
  const size_t n_lines = (size_t)(16ull << 30) / 64;  // 16 GB of lines
  uint32_t* next; CK(hipMalloc(&next, n_lines * 64));
  uint32_t* out; CK(hipMalloc(&out, 8192 * 64 * 4));
  const size_t cbytes = 256ull << 20;  // 256 MB in + 256 MB out per copy launch
  float4 *src, *dst; CK(hipMalloc(&src, cbytes)); CK(hipMalloc(&dst, cbytes));
  CK(hipMemset(src, 1, cbytes));
  fill<<<4096, 256>>>(next, n_lines, 12345); CK(hipDeviceSynchronize());
  hipStream_t s[2];
  CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
  const int hops = 150, chase_blocks = 768, reps = 50;
  const size_t n4 = cbytes / 16;
  for (int copy_blocks : {1024, 2048, 8192}) {
    for (int mode = 0; mode < 5; ++mode) {
      double best = 1e30;
      for (int trial = 0; trial < 3; ++trial) {
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; ++i) {
          if (mode == 0) {
            chase_k<<<chase_blocks, 256, 0, s[0]>>>(next, n_lines, hops, out);
          } else if (mode == 1) {
            copy_k<<<copy_blocks, 256, 0, s[0]>>>(src, dst, n4);
          } else if (mode == 2) {
            chase_k<<<chase_blocks, 256, 0, s[0]>>>(next, n_lines, hops, out);
            copy_k<<<copy_blocks, 256, 0, s[0]>>>(src, dst, n4);
          } else if (mode == 3) {
            chase_k<<<chase_blocks, 256, 0, s[0]>>>(next, n_lines, hops, out);
            copy_k<<<copy_blocks, 256, 0, s[1]>>>(src, dst, n4);
          } else {
            fused_k<<<chase_blocks + copy_blocks, 256, 0, s[0]>>>(next, n_lines, hops, out, chase_blocks, src, dst, n4);
          }
        }
        CK(hipDeviceSynchronize());
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        if (us < best) best = us;
      }
      const char* nm[] = {"chase alone", "copy alone", "chase+copy one stream", "chase | copy two streams", "fused roles one launch"};
      printf("copy_blocks %5d  %-26s %8.1f us per iteration\n", copy_blocks, nm[mode], best);
      fflush(stdout);
    }
  }
  return 0;
}
