// event_probe.hip - what does timing a kernel cost the stream it runs on?  (MI355X, tools only)
//
//   hipcc -O3 --offload-arch=gfx950 tools/event_probe.hip -o build/event_probe && build/event_probe
//
// Three ways to run the same back-to-back sequence of {short kernel, long streaming kernel} pairs (the shape of a
// control tick: k_sim 7 us + k_actor_dma 200 us):
//   plain     no events at all                                   -> wall time per pair
//   record    hipEventRecord before and after the long kernel    -> wall per pair, mean / min of the bracket
//   ext       hipExtLaunchKernelGGL(start, stop) on the long one -> wall per pair, mean / min of the kernel's own stamps
// rcg_profile used `record` until round 3: each record is a barrier packet, the bracketed kernel starts on a drained
// GPU and the bracket contains the drain.  `ext` takes the dispatch packet's own start / end stamps - what rocprofv3
// reports - and does not order anything.
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);   \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

__global__ void k_stream(const float4* __restrict__ in, float* __restrict__ out, long n4) {
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float* q = reinterpret_cast<const float*>(in + i);
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(q));
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 12345.678f) out[0] = acc;  // never true: keeps the loads
}

__global__ void k_small(float* p, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0001f + 1.0f;
}

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
  const long bytes = 1345585152L;  // the C2 candidate tensor
  const long n4 = bytes / 16;
  float4* in;
  float *out, *small;
  CK(hipMalloc(&in, bytes));
  CK(hipMalloc(&out, 64));
  CK(hipMalloc(&small, 65536 * 5 * 4));
  CK(hipMemset(in, 0, bytes));
  CK(hipMemset(small, 0, 65536 * 5 * 4));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const int n = 200;
  std::vector<hipEvent_t> ea(n), eb(n);
  for (int i = 0; i < n; ++i) {
    CK(hipEventCreate(&ea[i]));
    CK(hipEventCreate(&eb[i]));
  }
  auto pair = [&](int mode, int i) {
    hipLaunchKernelGGL(k_small, dim3(256), dim3(256), 0, s, small, 65536L);
    if (mode == 1) (void)hipEventRecord(ea[i], s);
    if (mode == 2)
      hipExtLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, s, ea[i], eb[i], 0, (const float4*)in, out, n4);
    else
      hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, s, (const float4*)in, out, n4);
    if (mode == 1) (void)hipEventRecord(eb[i], s);
  };
  for (int i = 0; i < 600; ++i) pair(0, 0);  // clock pre-spin
  CK(hipStreamSynchronize(s));
  const char* names[3] = {"plain", "record", "ext"};
  for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 3; ++mode) {
      for (int i = 0; i < 100; ++i) pair(0, 0);
      hipEvent_t t0, t1;
      CK(hipEventCreate(&t0));
      CK(hipEventCreate(&t1));
      CK(hipEventRecord(t0, s));
      const double w0 = now();
      for (int i = 0; i < n; ++i) pair(mode, i);
      CK(hipEventRecord(t1, s));
      CK(hipStreamSynchronize(s));
      const double w1 = now();
      float span = 0.f;
      CK(hipEventElapsedTime(&span, t0, t1));
      printf("%-6s rep %d: %.2f us per pair in-stream (host wall incl. queued warm-up %.2f)", names[mode], rep,
             span * 1e3 / n, (w1 - w0) * 1e6 / n);
      if (mode) {
        std::vector<float> d(n);
        for (int i = 0; i < n; ++i) CK(hipEventElapsedTime(&d[i], ea[i], eb[i]));
        std::sort(d.begin(), d.end());
        double sum = 0;
        for (float v : d) sum += v;
        printf("; long kernel: mean %.2f  median %.2f  min %.2f  max %.2f us", sum / n * 1e3, d[n / 2] * 1e3,
               d[0] * 1e3, d[n - 1] * 1e3);
      }
      printf("\n");
    }
  return 0;
}
