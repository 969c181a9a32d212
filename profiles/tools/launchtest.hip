// micro-benchmark: how long does it take to get 1024 one-wavefront workgroups (512 registers, 39 KB of LDS each) onto the chip, against 256
// workgroups of four wavefronts?  Each wavefront stamps s_memrealtime at its start and end and spins a fixed dependent FMA chain in between.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <vector>
template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES, 1) k(double* out, unsigned long long* stamps, int iters) {
  __shared__ double lds[WAVES][4896];   // 39 168 B per wavefront
  const int w = threadIdx.x / 64, lane = threadIdx.x % 64;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  lds[w][lane] = lane;
  double a[120];   // forces a large register allocation
#pragma unroll
  for (int i = 0; i < 120; i++) a[i] = 1.0 + 1e-3 * (lane + i);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 120; i++) a[i] = __builtin_fma(a[i], 0.999999, 1e-9 * a[(i + 7) % 120]);
  }
  double s = lds[w][lane];
#pragma unroll
  for (int i = 0; i < 120; i++) s += a[i];
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  const int gw = blockIdx.x * WAVES + w;
  out[gw * 64 + lane] = s;
  if (lane == 0) { stamps[2 * gw] = t0; stamps[2 * gw + 1] = t1; }
}
template <int WAVES> void run(double* out, unsigned long long* stamps, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int nwg = 1024 / WAVES;
  k<WAVES><<<nwg, 64 * WAVES>>>(out, stamps, iters); hipDeviceSynchronize();
  hipEventRecord(e0);
  k<WAVES><<<nwg, 64 * WAVES>>>(out, stamps, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(2048); hipMemcpy(h.data(), stamps, 2048 * 8, hipMemcpyDeviceToHost);
  unsigned long long first = ~0ull, last_start = 0, last_end = 0; double life = 0;
  for (int i = 0; i < 1024; i++) { first = std::min(first, h[2 * i]); last_start = std::max(last_start, h[2 * i]); last_end = std::max(last_end, h[2 * i + 1]); life += (double)(h[2 * i + 1] - h[2 * i]); }
  printf("%d wavefront(s) per workgroup: kernel %.3f ms (events) | first start -> last start %.1f us | first start -> last end %.1f us | mean wavefront life %.1f us (100 MHz ticks)\n",
         WAVES, ms, (last_start - first) / 100.0, (last_end - first) / 100.0, life / 1024 / 100.0);
}
int main() {
  double* out; unsigned long long* stamps; hipMalloc(&out, 1024 * 64 * 8); hipMalloc(&stamps, 2048 * 8);
  for (int iters : {2000, 8000}) { run<1>(out, stamps, iters); run<2>(out, stamps, iters); run<4>(out, stamps, iters); }
  return 0;
}
