// hwid_census.hip -- which values HW_REG_HW_ID / HW_REG_XCC_ID take on this part (for DuoSlots::claim's first-probe hash, cassie_kernels_duo.hip):
// 512 workgroups of two wavefronts, 78 KB of LDS each (the occupancy of env_step_duo_kernel: two workgroups per CU), every wavefront records its
// raw registers.  Prints per bit of HW_ID whether it varies, the value sets of the candidate fields, and the number of distinct keys.
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/hwid_census profiles/tools/hwid_census.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <map>
#include <set>
#include <vector>
__global__ void __launch_bounds__(128) census(unsigned* out, int spin) {
  __shared__ double pad[9000];
  pad[threadIdx.x] = threadIdx.x;
  const unsigned hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);
  const unsigned xcc = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 20);
  for (int i = 0; i < spin; i++) __builtin_amdgcn_s_sleep(100);
  const int w = blockIdx.x * 2 + (threadIdx.x >> 6);
  if ((threadIdx.x & 63) == 0) { out[2 * w] = hw; out[2 * w + 1] = xcc + (unsigned)(pad[threadIdx.x] * 0); }
}
int main() {
  const int waves = 1024;
  unsigned* d; hipMalloc(&d, waves * 8);
  census<<<waves / 2, 128>>>(d, 2000);
  std::vector<unsigned> h(2 * waves);
  hipMemcpy(h.data(), d, waves * 8, hipMemcpyDeviceToHost);
  unsigned orv = 0, andv = ~0u, xo = 0, xa = ~0u;
  std::set<unsigned long long> keys;
  std::map<int, std::set<unsigned>> f;
  for (int w = 0; w < waves; w++) {
    const unsigned hw = h[2 * w], x = h[2 * w + 1];
    orv |= hw; andv &= hw; xo |= x; xa &= x;
    f[0].insert(hw & 15); f[1].insert((hw >> 4) & 3); f[2].insert((hw >> 6) & 3); f[3].insert((hw >> 8) & 15); f[4].insert((hw >> 12) & 1); f[5].insert((hw >> 13) & 7);
    f[6].insert(x & 15);
    keys.insert(((unsigned long long)(x & 15) << 32) | (hw & 0xFFF0u));
  }
  printf("HW_ID bits that vary: %08x   XCC_ID reg bits that vary: %08x (or %08x)\n", orv & ~andv, xo & ~xa, xo);
  const char* nm[] = {"WAVE_ID[3:0]", "SIMD_ID[5:4]", "PIPE_ID[7:6]", "CU_ID[11:8]", "SH_ID[12]", "SE_ID[15:13]", "XCC_ID[3:0]"};
  for (int k = 0; k < 7; k++) { printf("%-14s:", nm[k]); for (unsigned v : f[k]) printf(" %u", v); printf("\n"); }
  printf("distinct (xcc, hw[15:4]) keys: %zu of %d wavefronts\n", keys.size(), waves);
  for (int w = 0; w < 8; w++) printf("wave %d: hw %08x xcc %08x\n", w, h[2 * w], h[2 * w + 1]);
  return 0;
}
