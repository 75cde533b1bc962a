// sync_probe.hip - development probe (round 6, rcg_loop_step): what one tiny launch + host wait costs on this runtime when the
// host waits with hipStreamSynchronize against polling a sequence number the kernel stores into coherent pinned host memory.
// hipcc -O3 --offload-arch=gfx950 tools/sync_probe.hip -o build/sync_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

__global__ void k(volatile double* out, double seq) {
  out[1] = seq * 2;
  __threadfence_system();
  out[0] = seq;
}

int main() {
  double* p;
  if (hipHostMalloc(&p, 4096, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) return 1;
  p[0] = 0;
  hipStream_t s;
  (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  for (int mode = 0; mode < 2; ++mode) {
    for (int w = 0; w < 200; ++w) {
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, p, (double)(w + 1));
      (void)hipStreamSynchronize(s);
    }
    const int n = 5000;
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) {
      const double seq = 1000.0 + mode * 100000 + i;
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, p, seq);
      if (mode == 0) {
        (void)hipStreamSynchronize(s);
      } else {
        while (*(volatile double*)p != seq) __builtin_ia32_pause();
      }
    }
    auto t1 = std::chrono::steady_clock::now();
    printf("%s: %.2f us per launch + wait\n", mode == 0 ? "hipStreamSynchronize" : "poll pinned sequence number",
           std::chrono::duration<double, std::micro>(t1 - t0).count() / n);
  }
  return 0;
}
