// Do kernels from two HIP streams overlap on this box?  (diagnostic, not part of the library)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin(long long cycles, int* out) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (out && threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1;
}
int main() {
  hipStream_t s[2];
  hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking);
  hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking);
  int* d;
  hipMalloc(&d, 4);
  const long long cyc = 5000;  // wall_clock64 ticks at 100 MHz -> 50 us
  for (int mode = 0; mode < 3; ++mode) {
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 200; ++i) {
      if (mode == 0) {
        hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s[0], cyc, d);
        hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s[0], cyc, d);
      } else {
        hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s[0], cyc, d);
        hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s[1], cyc, d);
      }
    }
    hipDeviceSynchronize();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    printf("mode %d (%s): %.2f ms for 400 kernels of ~50 us\n", mode, mode == 0 ? "one stream" : "two streams", ms);
  }
  return 0;
}
