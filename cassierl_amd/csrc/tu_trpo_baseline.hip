// tu_trpo_baseline.hip -- the baseline side of the TRPO outer loop (include/cassie_trpo.h): rllab's
// LinearFeatureBaseline (trpo_cassie.py:30: `LinearFeatureBaseline(env_spec=env.spec)`) on the batch the vectorised environment
// produces -- features [o, o^2, t/100, (t/100)^2, (t/100)^3, 1], o = clip(obs, -10, 10), a ridge regression of the discounted
// return on them, and the advantages of the batch against its prediction.
//
// As torch operations (cassierl_amd/trpo.py: LinearFeatureBaseline, TRPO.process) this materialises the [N, 56] feature matrix three
// times, twice of them in float64 (235 MB each at N = 524 288), and sends the normal equations through rocBLAS: 1.9 ms of the 4.7 ms
// TRPO update (r04_j).  Here:
//   * returns_adv_kernel: one lane per environment walks its column of the [T][n] batch backwards -- value = features . coeffs (the
//     feature vector lives in registers, never in memory), return-to-go with rllab's path cuts, advantage, and the sums the
//     advantage normalisation needs (fixed-order tree per workgroup);
//   * gram_kernel: Z'Z for Z = [features | y] on the FP64 matrix cores (v_mfma_f64_16x16x4_f64).  The A and B operands of a
//     16x16x4 step are both "lane (f, k) holds column 16 r + f of sample 4 step + k", so every lane EVALUATES the (up to) four
//     entries of Z it feeds -- one float load each -- and the ten upper 16x16 blocks of the 64 x 64 Gram matrix are ten MFMAs per four
//     samples: 524 288 samples = 1.3 M MFMAs = ~35 us of matrix-core time.  One block set of partial sums per wavefront; the caller adds
//     them up in a fixed order (and all-reduces over ranks where it did before).
// Feature arithmetic is float32 exactly as the torch expressions evaluate it (o * o, t * (1 / 100), x * x, x * x * x), promoted to float64.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cassie_trpo.h"
#include "../../include/cassie_vec.h"

namespace cassie_trpo {

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int GRAM_BLOCKS = 512, GRAM_WAVES = 4;   // two wavefronts per SIMD

template <int D> struct Feat {
  static constexpr int NF = 2 * D + 4;          // features
  static constexpr int NC = NF + 1;             // ... and the regression target as one more column of Z
  static constexpr int NB = (NC + 15) / 16;     // 16-column blocks of Z
  static constexpr int NBLK = NB * (NB + 1) / 2;
};

__device__ __forceinline__ float clip10(float x) { return fminf(fmaxf(x, -10.0f), 10.0f); }
// t / 100 as torch evaluates `t.to(float32) / 100.0` on the device: a multiplication by the float32 reciprocal of the scalar
__device__ __forceinline__ float path_clock(long long t) { return (float)t * (1.0f / 100.0f); }

// column F of Z for one sample (obs row `o`, path clock t, target y); compile-time block r, run-time f
template <int D>
__device__ __forceinline__ double zcol(const float* __restrict__ o, float al, double y, int F, bool ok) {
  constexpr int NF = Feat<D>::NF;
  if (!ok) return 0.0;
  if (F < 2 * D) {
    const float v = clip10(o[F < D ? F : F - D]);
    return (double)(F < D ? v : v * v);
  }
  if (F == 2 * D) return (double)al;
  if (F == 2 * D + 1) return (double)(al * al);
  if (F == 2 * D + 2) return (double)(al * al * al);
  if (F == 2 * D + 3) return 1.0;
  if (F == NF) return y;
  return 0.0;
}

template <int D>
__global__ void __launch_bounds__(64 * GRAM_WAVES) gram_kernel(const float* __restrict__ obs, const long long* __restrict__ t, const double* __restrict__ y, int m,
                                                              double* __restrict__ partial) {
  typedef Feat<D> Ft;
  constexpr int NB = Ft::NB;
  const int lane = threadIdx.x & 63, wave = blockIdx.x * GRAM_WAVES + (threadIdx.x >> 6), nwaves = gridDim.x * GRAM_WAVES;
  const int f = lane & 15, k = lane >> 4;
  const int steps = (m + 3) / 4, per = (steps + nwaves - 1) / nwaves;
  const int s0 = wave * per, s1 = s0 + per < steps ? s0 + per : steps;
  v4d acc[Ft::NBLK];
#pragma unroll
  for (int b = 0; b < Ft::NBLK; b++) acc[b] = (v4d){0.0, 0.0, 0.0, 0.0};
  // the entries of Z this lane feeds in one step: NB columns of sample 4 step + k.  Loaded one step AHEAD: with one or two wavefronts
  // per SIMD the global loads of a step would otherwise sit in front of its ten MFMAs (first build: 0.19 ms, all of it load latency).
  auto fetch = [&](int st, double (&z)[NB]) {
    const int s = 4 * st + k;
    const bool ok = st < s1 && s < m;
    const size_t ss = ok ? (size_t)s : 0;
    const float al = path_clock(t[ss]);
    const double yy = y[ss];
#pragma unroll
    for (int r = 0; r < NB; r++) z[r] = zcol<D>(obs + ss * D, al, yy, 16 * r + f, ok);
  };
  double z[NB], zn[NB];
  fetch(s0, z);
  for (int st = s0; st < s1; st++) {
    fetch(st + 1, zn);
    int b = 0;
#pragma unroll
    for (int r = 0; r < NB; r++)
#pragma unroll
      for (int c = r; c < NB; c++, b++) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(z[r], z[c], acc[b], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < NB; r++) z[r] = zn[r];
  }
  // element (i = k + 4 v, j = f) of block b in register v
  double* out = partial + (size_t)wave * Ft::NBLK * 256;
#pragma unroll
  for (int b = 0; b < Ft::NBLK; b++)
#pragma unroll
    for (int v = 0; v < 4; v++) out[b * 256 + (k + 4 * v) * 16 + f] = acc[b][v];
}

// One lane per environment, backwards over the T steps of its column: baseline value of every sample, return-to-go (restarting behind a
// cut path; bootstrapped with last_value), advantage = return - value (gae_lambda = 1), and per workgroup (sum adv, sum adv^2).
template <int D>
__global__ void __launch_bounds__(256) returns_adv_kernel(const float* __restrict__ obs, const long long* __restrict__ t, const double* __restrict__ rew,
                                                         const uint8_t* __restrict__ cut, int T, int n, const double* __restrict__ coeffs,
                                                         const double* __restrict__ last_value, double gamma, double* __restrict__ returns,
                                                         double* __restrict__ adv, double* __restrict__ partial) {
  constexpr int NF = Feat<D>::NF;
  __shared__ double red[2][256];
  const int i = blockIdx.x * 256 + threadIdx.x;
  double s1 = 0.0, s2 = 0.0;
  if (i < n) {
    double run = last_value ? last_value[i] : 0.0;
    for (int tt = T - 1; tt >= 0; tt--) {
      const size_t s = (size_t)tt * n + i;
      double value = 0.0;
      if (coeffs) {
        const float* o = obs + s * D;
        const float al = path_clock(t[s]);
#pragma unroll
        for (int j = 0; j < D; j++) {
          const float v = clip10(o[j]);
          value = fma((double)v, coeffs[j], value);
          value = fma((double)(v * v), coeffs[D + j], value);
        }
        value = fma((double)al, coeffs[2 * D], value);
        value = fma((double)(al * al), coeffs[2 * D + 1], value);
        value = fma((double)(al * al * al), coeffs[2 * D + 2], value);
        value += coeffs[NF - 1];
      }
      run = rew[s] + gamma * run * (cut[s] ? 0.0 : 1.0);
      const double a = run - value;
      returns[s] = run; adv[s] = a;
      s1 += a; s2 += a * a;
    }
  }
  red[0][threadIdx.x] = s1; red[1][threadIdx.x] = s2;
  __syncthreads();
  for (int m = 128; m >= 1; m >>= 1) {
    if ((int)threadIdx.x < m) { red[0][threadIdx.x] += red[0][threadIdx.x + m]; red[1][threadIdx.x] += red[1][threadIdx.x + m]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { partial[2 * blockIdx.x] = red[0][0]; partial[2 * blockIdx.x + 1] = red[1][0]; }
}

template <int D>
__global__ void __launch_bounds__(256) predict_kernel(const float* __restrict__ obs, const long long* __restrict__ t, int m, const double* __restrict__ coeffs,
                                                     double* __restrict__ out) {
  constexpr int NF = Feat<D>::NF;
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= m) return;
  const float* o = obs + (size_t)s * D;
  const float al = path_clock(t[s]);
  double value = 0.0;
#pragma unroll
  for (int j = 0; j < D; j++) {
    const float v = clip10(o[j]);
    value = fma((double)v, coeffs[j], value);
    value = fma((double)(v * v), coeffs[D + j], value);
  }
  value = fma((double)al, coeffs[2 * D], value);
  value = fma((double)(al * al), coeffs[2 * D + 1], value);
  value = fma((double)(al * al * al), coeffs[2 * D + 2], value);
  out[s] = value + coeffs[NF - 1];
}

// (A + reg I) x = b for the baseline's F coefficients, A symmetric positive semi-definite: Cholesky by ONE wavefront with row i of the matrix
// in the registers of lane i (the pivot column travels through v_readlane: no LDS, no barrier), forward substitution by columns, backward
// substitution by wave sums, and LinearFeatureBaseline.fit's retry rule ON THE DEVICE -- if the factorisation meets a non-positive pivot or
// the solution is not finite, the regulariser is multiplied by ten (five tries) -- so that the fit needs no host read-back (as torch
// operations: rocSOLVER's LU + torch.isfinite(...).all() -> bool, 0.3 ms and a synchronisation per iteration; a first version of this
// kernel with the matrix in LDS and serial substitutions took 0.23 ms).
__device__ __forceinline__ double lane_bcast(double x, int src) {   // src: compile-time constant after unrolling -> v_readlane
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), src), __builtin_amdgcn_readlane(__double2loint(x), src));
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
template <int F>
__global__ void __launch_bounds__(64) ridge_solve_kernel(const double* __restrict__ A, const double* __restrict__ b, double reg, double* __restrict__ x_out) {
  const int i = threadIdx.x;
  const bool row = i < F;
  double xs = 0.0;
  for (int attempt = 0; attempt < 5; attempt++, reg *= 10.0) {
    double a[F], diag = 1.0;
#pragma unroll
    for (int k = 0; k < F; k++) a[k] = (row && k <= i) ? A[(size_t)i * F + k] + (k == i ? reg : 0.0) : 0.0;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < F; j++) {
      const double d = lane_bcast(a[j], j);
      bad = bad || !(d > 0.0) || !(d < 1e300);
      const double sd = sqrt(bad ? 1.0 : d), lij = a[j] / sd;   // lanes i > j: L[i][j]; lane j: sqrt(d)
      if (i == j) diag = sd;
      a[j] = lij;
#pragma unroll
      for (int k = j + 1; k < F; k++) { const double lkj = lane_bcast(lij, k); if (i >= k) a[k] -= lij * lkj; }
    }
    // forward: L y = b, by columns (lane r finishes y_r, the lanes below subtract its column)
    double sacc = row ? b[i] : 0.0, y = 0.0;
#pragma unroll
    for (int r = 0; r < F; r++) {
      const double yr = lane_bcast(sacc / diag, r);
      if (i == r) y = yr;
      if (i > r) sacc -= a[r] * yr;
    }
    // backward: L' x = y, x_r = (y_r - sum_{k > r} L[k][r] x_k) / L[r][r]: the sum runs over LANES
    xs = 0.0;
#pragma unroll
    for (int r = F - 1; r >= 0; r--) {
      const double t = wave_sum((row && i > r) ? a[r] * xs : 0.0);
      if (i == r) xs = (y - t) / diag;
    }
    const bool nonfinite = __ballot(row && !(fabs(xs) < 1e300)) != 0ull;
    if (!bad && !nonfinite) break;
  }
  if (row) x_out[i] = xs;   // (as the torch loop: after five tries the last attempt's vector, finite or not)
}

}  // namespace cassie_trpo

extern "C" {

int CassieTrpoBaselineFeatures(int obs_dim) { return (obs_dim == 26 || obs_dim == 17) ? 2 * obs_dim + 4 : 0; }
int CassieTrpoGramRows(void) { return cassie_trpo::GRAM_BLOCKS * cassie_trpo::GRAM_WAVES; }
int CassieTrpoGramRowSize(int obs_dim) {
  return obs_dim == 26 ? cassie_trpo::Feat<26>::NBLK * 256 : (obs_dim == 17 ? cassie_trpo::Feat<17>::NBLK * 256 : 0);
}

int CassieTrpoBaselineGram(const float* obs_dev, const long long* t_dev, const double* y_dev, int m, int obs_dim, double* partial_dev, void* stream) {
  if (!obs_dev || !t_dev || !y_dev || m <= 0 || !partial_dev) return CASSIE_EINVAL;
  const dim3 grid(cassie_trpo::GRAM_BLOCKS), block(64 * cassie_trpo::GRAM_WAVES);
  if (obs_dim == 26) hipLaunchKernelGGL(cassie_trpo::gram_kernel<26>, grid, block, 0, (hipStream_t)stream, obs_dev, t_dev, y_dev, m, partial_dev);
  else if (obs_dim == 17) hipLaunchKernelGGL(cassie_trpo::gram_kernel<17>, grid, block, 0, (hipStream_t)stream, obs_dev, t_dev, y_dev, m, partial_dev);
  else return CASSIE_EINVAL;
  return hipGetLastError() == hipSuccess ? CASSIE_OK : CASSIE_EHIP;
}

int CassieTrpoReturnsAdvantages(const float* obs_dev, const long long* t_dev, const double* rew_dev, const unsigned char* cut_dev, int T, int n, int obs_dim,
                                const double* coeffs_dev, const double* last_value_dev, double gamma, double* returns_dev, double* adv_dev, double* partial_dev,
                                void* stream) {
  if (!obs_dev || !t_dev || !rew_dev || !cut_dev || T <= 0 || n <= 0 || !returns_dev || !adv_dev || !partial_dev) return CASSIE_EINVAL;
  const dim3 grid((n + 255) / 256), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (obs_dim == 26) hipLaunchKernelGGL(cassie_trpo::returns_adv_kernel<26>, grid, block, 0, s, obs_dev, t_dev, rew_dev, cut_dev, T, n, coeffs_dev, last_value_dev, gamma, returns_dev, adv_dev, partial_dev);
  else if (obs_dim == 17) hipLaunchKernelGGL(cassie_trpo::returns_adv_kernel<17>, grid, block, 0, s, obs_dev, t_dev, rew_dev, cut_dev, T, n, coeffs_dev, last_value_dev, gamma, returns_dev, adv_dev, partial_dev);
  else return CASSIE_EINVAL;
  return hipGetLastError() == hipSuccess ? CASSIE_OK : CASSIE_EHIP;
}

int CassieTrpoRidgeSolve(const double* A_dev, const double* b_dev, int F, double reg, double* x_dev, void* stream) {
  if (!A_dev || !b_dev || !x_dev) return CASSIE_EINVAL;
  if (F == 56) hipLaunchKernelGGL(cassie_trpo::ridge_solve_kernel<56>, dim3(1), dim3(64), 0, (hipStream_t)stream, A_dev, b_dev, reg, x_dev);
  else if (F == 38) hipLaunchKernelGGL(cassie_trpo::ridge_solve_kernel<38>, dim3(1), dim3(64), 0, (hipStream_t)stream, A_dev, b_dev, reg, x_dev);
  else return CASSIE_EINVAL;
  return hipGetLastError() == hipSuccess ? CASSIE_OK : CASSIE_EHIP;
}

int CassieTrpoBaselinePredict(const float* obs_dev, const long long* t_dev, int m, int obs_dim, const double* coeffs_dev, double* out_dev, void* stream) {
  if (!obs_dev || !t_dev || m <= 0 || !coeffs_dev || !out_dev) return CASSIE_EINVAL;
  const dim3 grid((m + 255) / 256), block(256);
  if (obs_dim == 26) hipLaunchKernelGGL(cassie_trpo::predict_kernel<26>, grid, block, 0, (hipStream_t)stream, obs_dev, t_dev, m, coeffs_dev, out_dev);
  else if (obs_dim == 17) hipLaunchKernelGGL(cassie_trpo::predict_kernel<17>, grid, block, 0, (hipStream_t)stream, obs_dev, t_dev, m, coeffs_dev, out_dev);
  else return CASSIE_EINVAL;
  return hipGetLastError() == hipSuccess ? CASSIE_OK : CASSIE_EHIP;
}

}  // extern "C"
