// Probe: what rate do random 1 KB row gathers reach on this box, one wave per chain of L rows, R rows in flight,
// at a given occupancy (registers per wave forced by a dummy array)?  (diagnostic, never shipped)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>

template <int R, int WPE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void gather_k(const float4* __restrict__ pool, const int32_t* __restrict__ chain_ptr, const int32_t* __restrict__ chain,
              int n_items, float* __restrict__ out, int dep_hops, const int32_t* __restrict__ hopbuf) {
  const int item = blockIdx.x;
  const int lane = threadIdx.x;
  if (item >= n_items) return;
  int c0 = chain_ptr[item], c1 = chain_ptr[item + 1];
  // optional dependent small hops in front (the item / ancestor-list round trips of the scoring kernel)
  int x = item;
  for (int h = 0; h < dep_hops; ++h) x = hopbuf[((size_t)x * 64 + lane) & 0xFFFFFF] + item;
  c0 += x & 0;
  float4 acc = make_float4(0, 0, 0, 0);
  for (int i = c0; i < c1; i += R) {
    float4 v[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
      const int j = i + u < c1 ? i + u : c1 - 1;
      const int row = chain[j];
      v[u] = pool[(size_t)row * 64 + lane];
    }
#pragma unroll
    for (int u = 0; u < R; ++u) {
      acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
    }
  }
  out[(size_t)item * 64 + lane] = acc.x + acc.y + acc.z + acc.w;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int R, int WPE>
static int run(const char* tag, const float4* pool, const int32_t* cp, const int32_t* ch, int n_items, float* out,
               int dep, const int32_t* hopbuf, double bytes) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  gather_k<R, WPE><<<n_items, 64>>>(pool, cp, ch, n_items, out, dep, hopbuf);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int r = 0; r < 5; ++r) gather_k<R, WPE><<<n_items, 64>>>(pool, cp, ch, n_items, out, dep, hopbuf);
  CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
  printf("%-28s R=%2d waves/SIMD=%d dep_hops=%d: %7.1f us  %.2f TB/s\n", tag, R, WPE, dep, ms * 1e3, bytes / ms / 1e9);
  fflush(stdout);
  return 0;
}

int main() {
  const int n_items = 20800;
  for (double gb : {8.0, 240.0}) {
    const size_t rows = (size_t)(gb * (1ull << 30)) / 1024;
    float4* pool; if (hipMalloc(&pool, rows * 1024) != hipSuccess) { printf("alloc %.0f GB failed\n", gb); continue; }
    CK(hipMemset(pool, 0, rows * 1024));
    int32_t* hopbuf; CK(hipMalloc(&hopbuf, (1 << 24) * 4)); CK(hipMemset(hopbuf, 0, (1 << 24) * 4));
    for (int mode = 0; mode < 3; ++mode) {
      // mode 0: every chain 23 rows, rows fully random; mode 1: geometric-ish lengths (mean 23, long tail), random rows;
      // mode 2: like 1, rows of a chain drawn from a handful of 64-row pages (a search's own pages)
      std::mt19937_64 rng(7);
      std::vector<int32_t> cp(n_items + 1, 0), ch;
      for (int i = 0; i < n_items; ++i) {
        int L = 23;
        if (mode >= 1) { std::exponential_distribution<double> ex(1.0 / 20.0); L = 3 + (int)ex(rng); if (L > 250) L = 250; }
        if (mode == 2) {
          const int np = 1 + L / 8;
          std::vector<size_t> pages(np);
          for (auto& p : pages) p = rng() % (rows / 64);
          for (int k = 0; k < L; ++k) ch.push_back((int32_t)(pages[rng() % np] * 64 + rng() % 64));
        } else {
          for (int k = 0; k < L; ++k) ch.push_back((int32_t)(rng() % rows));
        }
        cp[i + 1] = (int32_t)ch.size();
      }
      int32_t *dcp, *dch; float* out;
      CK(hipMalloc(&dcp, cp.size() * 4)); CK(hipMalloc(&dch, ch.size() * 4)); CK(hipMalloc(&out, (size_t)n_items * 256));
      CK(hipMemcpy(dcp, cp.data(), cp.size() * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(dch, ch.data(), ch.size() * 4, hipMemcpyHostToDevice));
      const double bytes = (double)ch.size() * 1024;
      char tag[64]; snprintf(tag, sizeof tag, "pool %.0f GB mode %d", gb, mode);
      run<8, 4>(tag, pool, dcp, dch, n_items, out, 0, hopbuf, bytes);
      run<8, 4>(tag, pool, dcp, dch, n_items, out, 2, hopbuf, bytes);
      run<8, 6>(tag, pool, dcp, dch, n_items, out, 0, hopbuf, bytes);
      run<8, 6>(tag, pool, dcp, dch, n_items, out, 2, hopbuf, bytes);
      run<8, 8>(tag, pool, dcp, dch, n_items, out, 0, hopbuf, bytes);
      run<8, 8>(tag, pool, dcp, dch, n_items, out, 2, hopbuf, bytes);
      run<4, 8>(tag, pool, dcp, dch, n_items, out, 2, hopbuf, bytes);
      run<16, 4>(tag, pool, dcp, dch, n_items, out, 0, hopbuf, bytes);
      run<16, 4>(tag, pool, dcp, dch, n_items, out, 2, hopbuf, bytes);
      run<16, 2>(tag, pool, dcp, dch, n_items, out, 2, hopbuf, bytes);
      run<8, 2>(tag, pool, dcp, dch, n_items, out, 2, hopbuf, bytes);
      CK(hipFree(dcp)); CK(hipFree(dch)); CK(hipFree(out));
    }
    CK(hipFree(pool)); CK(hipFree(hopbuf));
  }
  return 0;
}
