// trig_probe.hip - development probe: accuracy of gfx950's v_sin_f32 / v_cos_f32 (input in revolutions)
// behind a two-constant reduction, and of the f64 fast path rcg::sincos_fast (rcg_math.hpp), against float64 / long
// double libm.  hipcc -O3 --offload-arch=gfx950 -Ircognita_amd/csrc tools/trig_probe.hip -o /tmp/trig_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#include "rcg_math.hpp"

__global__ void k64(const double* x, double* s, double* c, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) rcg::sincos_fast(x[i], &s[i], &c[i]);
}

__global__ void k(const float* x, float* s, float* c, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  rcg::sincos_hw(x[i], &s[i], &c[i]);  // the shipped function: two-constant reduction + v_sin_f32 / v_cos_f32
}

__global__ void kr(const float* x, float* s, float* c, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  rcg::sincos_r<float>(x[i], &s[i], &c[i]);  // the simulator's form: three-constant reduction + minimax
}

int main() {
  const int n = 1 << 22;
  std::vector<float> hx(n), hs(n), hc(n);
  for (int form = 0; form < 2; ++form)
  for (double range : {3.2, 100.0, 1000.0, 1e4, 1e5, 1e6}) {
    for (int i = 0; i < n; ++i) hx[i] = (float)(((double)rand() / RAND_MAX * 2 - 1) * range);
    float *dx, *ds, *dc;
    hipMalloc(&dx, n * 4);
    hipMalloc(&ds, n * 4);
    hipMalloc(&dc, n * 4);
    hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(form ? kr : k, dim3(n / 256), dim3(256), 0, 0, dx, ds, dc, n);
    hipMemcpy(hs.data(), ds, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hc.data(), dc, n * 4, hipMemcpyDeviceToHost);
    double es = 0, ec = 0;
    for (int i = 0; i < n; ++i) {
      es = fmax(es, fabs((double)hs[i] - sin((double)hx[i])));
      ec = fmax(ec, fabs((double)hc[i] - cos((double)hx[i])));
    }
    printf("%s, range +-%g: max abs err sin %.3e cos %.3e\n", form ? "f32 sincos_r" : "f32 sincos_hw", range, es, ec);
    hipFree(dx);
    hipFree(ds);
    hipFree(dc);
  }
  {  // f64 fast path against long double libm
    std::vector<double> gx(n), gs(n), gc(n);
    for (double range : {3.2, 100.0, 1000.0, 1e5}) {
      for (int i = 0; i < n; ++i) gx[i] = ((double)rand() / RAND_MAX * 2 - 1) * range;
      double *dx, *ds, *dc;
      hipMalloc(&dx, n * 8);
      hipMalloc(&ds, n * 8);
      hipMalloc(&dc, n * 8);
      hipMemcpy(dx, gx.data(), n * 8, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k64, dim3(n / 256), dim3(256), 0, 0, dx, ds, dc, n);
      hipMemcpy(gs.data(), ds, n * 8, hipMemcpyDeviceToHost);
      hipMemcpy(gc.data(), dc, n * 8, hipMemcpyDeviceToHost);
      long double es = 0, ec = 0;
      for (int i = 0; i < n; ++i) {
        es = fmaxl(es, fabsl((long double)gs[i] - sinl((long double)gx[i])));
        ec = fmaxl(ec, fabsl((long double)gc[i] - cosl((long double)gx[i])));
      }
      printf("f64 sincos_fast, range +-%g: max abs err sin %.3Le cos %.3Le\n", range, es, ec);
      hipFree(dx);
      hipFree(ds);
      hipFree(dc);
    }
  }
  return 0;
}
