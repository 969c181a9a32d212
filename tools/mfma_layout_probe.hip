#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(double* out) {
  int lane = threadIdx.x;
  unsigned a = lane, b = 100 + lane;
  u2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  u2 q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[lane * 8 + 0] = r.x; out[lane * 8 + 1] = r.y; out[lane * 8 + 2] = q.x; out[lane * 8 + 3] = q.y;
  // hypothesis: A operand lane i = A[i%16][i/16], B operand lane i = B[i/16][i%16], D lane i reg j = D[4*(i/16)+j][i%16]
  int row = lane % 16, kk = lane / 16;
  double A = row * 4 + kk + 1;          // A[row][k] = 4 row + k + 1
  double B = (kk + 1) * 100 + row;      // B[k][col] = 100 (k+1) + col
  d4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A, B, acc, 0, 0, 0);
  out[lane * 8 + 4] = acc.x; out[lane * 8 + 5] = acc.y; out[lane * 8 + 6] = acc.z; out[lane * 8 + 7] = acc.w;
}
int main() {
  double* d; hipMalloc(&d, 64 * 8 * 8);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  double h[512]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("permlane32_swap(a=lane,b=100+lane): lane: r.x r.y | permlane16: q.x q.y\n");
  for (int l = 0; l < 64; l += 5) printf("  %2d: %3.0f %3.0f | %3.0f %3.0f\n", l, h[l*8], h[l*8+1], h[l*8+2], h[l*8+3]);
  int bad = 0;
  for (int l = 0; l < 64; l++) for (int j = 0; j < 4; j++) {
    int r = 4 * (l / 16) + j, c = l % 16; double e = 0;
    for (int kk = 0; kk < 4; kk++) e += (r * 4 + kk + 1) * ((kk + 1) * 100.0 + c);
    if (h[l*8+4+j] != e) bad++;
  }
  printf("mfma layout hypothesis mismatches: %d (lane 17 regs: %.0f %.0f %.0f %.0f)\n", bad, h[17*8+4], h[17*8+5], h[17*8+6], h[17*8+7]);
  return 0;
}
