// micro-benchmark: cost of a per-wavefront hand-over workspace (batched buffer stores / loads) beside FP64 work, 1024 single-wave workgroups
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
template <int NSLOT, bool MEM>
__global__ void __launch_bounds__(64, 1) k(double* ws, double* out, int passes, int flops) {
  __shared__ double lds_pad[4800];   // 38.4 KB: four workgroups per CU, as the physics kernels
  lds_pad[threadIdx.x] = 0;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(ws + (size_t)blockIdx.x * 64 * NSLOT, 0, NSLOT * 512, 0x00020000);
  unsigned voff = threadIdx.x * 8;
  double a[8];
  for (int i = 0; i < 8; i++) a[i] = 1.0 + 1e-3 * (threadIdx.x + i);
  double acc = 0;
  for (int p = 0; p < passes; p++) {
    for (int ph = 0; ph < 6; ph++) {   // six phases per pass: each loads NSLOT/6 slots in one batch, computes, stores them back
      double v[NSLOT / 6];
      if (MEM) {
#pragma unroll
        for (int s = 0; s < NSLOT / 6; s++) { u2 w = __builtin_amdgcn_raw_buffer_load_b64(r, voff, (ph * (NSLOT / 6) + s) * 512, 0); v[s] = __hiloint2double(w.y, w.x); }
      } else {
#pragma unroll
        for (int s = 0; s < NSLOT / 6; s++) v[s] = a[s & 7];
      }
      for (int f = 0; f < flops; f++) {
#pragma unroll
        for (int i = 0; i < 8; i++) a[i] = __builtin_fma(a[i], 0.999999, 1e-9 * a[(i + 1) & 7]);
      }
#pragma unroll
      for (int s = 0; s < NSLOT / 6; s++) v[s] += a[s & 7];
      if (MEM) {
#pragma unroll
        for (int s = 0; s < NSLOT / 6; s++) { u2 w; w.x = __double2loint(v[s]); w.y = __double2hiint(v[s]); __builtin_amdgcn_raw_buffer_store_b64(w, r, voff, (ph * (NSLOT / 6) + s) * 512, 0); }
      } else {
#pragma unroll
        for (int s = 0; s < NSLOT / 6; s++) acc += v[s];
      }
    }
  }
  for (int i = 0; i < 8; i++) acc += a[i];
  out[blockIdx.x * 64 + threadIdx.x] = acc + lds_pad[threadIdx.x];
}
template <int NSLOT, bool MEM> float run(double* ws, double* out, int nwg, int passes, int flops) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NSLOT, MEM><<<nwg, 64>>>(ws, out, passes, flops);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; i++) k<NSLOT, MEM><<<nwg, 64>>>(ws, out, passes, flops);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5;
}
int main() {
  const int nwg = 1024;
  double *ws, *out; hipMalloc(&ws, (size_t)nwg * 64 * 300 * 8); hipMalloc(&out, nwg * 64 * 8);
  hipMemset(ws, 0, (size_t)nwg * 64 * 300 * 8);
  // flops per phase chosen so that compute-only is ~0.75 ms for 11 passes (the duo kernel's VALU-busy time)
  for (int flops : {300, 340}) {
    printf("flops/phase %d: compute only %.3f ms | +132 slots (66 KB/wave) %.3f ms | +264 slots (132 KB/wave) %.3f ms | +300 slots %.3f ms\n", flops,
           run<132, false>(ws, out, nwg, 11, flops), run<132, true>(ws, out, nwg, 11, flops), run<264, true>(ws, out, nwg, 11, flops), run<300, true>(ws, out, nwg, 11, flops));
  }
  return 0;
}
