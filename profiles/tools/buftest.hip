#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
__global__ void k(double* ws, double* out, int flags) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(ws + (size_t)blockIdx.x * 64 * 16, 0, 16 * 512, flags);
  unsigned voff = threadIdx.x * 8;
  for (int s = 0; s < 16; s++) {
    double v = 1000.0 * blockIdx.x + 10.0 * s + threadIdx.x * 0.001;
    u2 w; w.x = __double2loint(v); w.y = __double2hiint(v);
    __builtin_amdgcn_raw_buffer_store_b64(w, r, voff, s * 512, 0);
  }
  double acc = 0;
  for (int s = 0; s < 16; s++) {
    u2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, s * 512, 0);
    acc += __hiloint2double(v.y, v.x);
  }
  out[blockIdx.x * 64 + threadIdx.x] = acc;
}
int main() {
  double *ws, *out; hipMalloc(&ws, 4 * 64 * 16 * 8); hipMalloc(&out, 4 * 64 * 8);
  for (int flags : {0x00020000, 0x00027000}) {
    hipMemset(ws, 0, 4 * 64 * 16 * 8);
    k<<<4, 64>>>(ws, out, flags);
    double h[256], hw[64 * 16]; hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(hw, ws + 64 * 16, sizeof hw, hipMemcpyDeviceToHost);
    double exp1 = 0; for (int s = 0; s < 16; s++) exp1 += 1000.0 + 10.0 * s + 5 * 0.001;
    printf("flags %x: out[1][5] = %.6f expect %.6f ; ws[1][slot3][lane7] = %.6f expect %.6f\n", flags, h[64 + 5], exp1, hw[3 * 64 + 7], 1000.0 + 30 + 0.007);
  }
  return 0;
}
