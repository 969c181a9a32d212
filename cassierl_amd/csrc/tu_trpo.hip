// tu_trpo.hip -- fused policy kernels of the TRPO outer loop (include/cassie_trpo.h; rllab/envs/trpo_cassie.py:21-42).
//
// One TRPO update evaluates, on the 524 288 samples one rank collects per iteration, the policy gradient and ~11 Fisher-vector
// products of the 26-32-32-6 tanh policy.  As torch operations a product is ~8 skinny GEMMs (K = 32) and a dozen element-wise passes
// over [N, 32] tensors: 0.7 ms each, 7.8 ms per update (r04 profile, tests/prof_trpo_update.py).  Here it is ONE launch: a lane owns
// a sample, the weights (and the direction of the product) come through the scalar unit, the sample's activations are
// recomputed in registers (obs is the only per-sample input read from HBM: 104 B), the directional derivative runs forward, the
// cotangent runs back, and the outer products that make up the parameter gradient are accumulated per wavefront: the 64 samples of a
// tile are staged in LDS and every lane adds them into the ~35 parameters it owns, in registers, over all the tiles of the launch.
// Each wavefront writes one row of partial sums; the caller adds the rows (and all-reduces over ranks where it did before).
// FP32 vector arithmetic on purpose: per sample 6.3 k multiply-adds with K = 26..32 -- far too small for MFMA tiles to pay.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cassie_trpo.h"
#include "../../include/cassie_vec.h"

namespace cassie_trpo {

constexpr int H = 32;         // hidden units (both layers)
constexpr int STG = 36;       // floats per staged sample row: 16-byte aligned rows, bank = (36 lane + k) mod 64
constexpr int WAVES = 2;      // wavefronts per workgroup
constexpr int MAX_BLOCKS = 768;

template <int D, int A> struct Shape {
  static constexpr int DP = (D + 3) / 4 * 4;   // obs row padded to float4s (zeros)
  static constexpr int NP = H * D + H + H * H + H + A * H + A;
  static constexpr int O_W1 = 0, O_B1 = H * D, O_W2 = O_B1 + H, O_B2 = O_W2 + H * H, O_W3 = O_B2 + H, O_B3 = O_W3 + A * H;
};

struct Net { const float *W1, *b1, *W2, *b2, *W3, *b3; };

// The weights are the same for every lane: read through the CONSTANT address space at compile-time offsets they are scalar loads
// (s_load through the scalar cache) and the multiply-adds take them as SGPR operands -- no LDS traffic, no vector registers for the
// 2 x 2118 weights.  (r04 history, 524 288 samples, ms per product: weights in LDS read as broadcasts 1.44 -- the scheduler hoists the
// ~800 reads of a product and spills ~1000 registers whatever barriers are put in --, this form 0.80, the torch operations it would
// replace 0.67: not the default, see TRPO.fused_fisher.)
typedef const __attribute__((address_space(4))) float* cptr;
struct CNet { cptr W1, b1, W2, b2, W3, b3; };
// the same pointer, wave-uniform, but opaque to the optimiser: otherwise all 4236 loads are hoisted out of the tile loop and kept in
// VGPR lanes (a v_readlane per use)
__device__ __forceinline__ cptr reissue(const float* p) {
  uint32_t lo = (uint32_t)(uintptr_t)p, hi = (uint32_t)((uintptr_t)p >> 32);
  asm volatile("" : "+v"(lo), "+v"(hi));
  lo = __builtin_amdgcn_readfirstlane(lo); hi = __builtin_amdgcn_readfirstlane(hi);
  return (cptr)(((uintptr_t)hi << 32) | lo);
}
__device__ __forceinline__ CNet reissue(const Net& n) { return CNet{reissue(n.W1), reissue(n.b1), reissue(n.W2), reissue(n.b2), reissue(n.W3), reissue(n.b3)}; }
// y[j] = bias[j] + sum_i W[j * LD + i] x[i]
template <int NO, int NI, int LD> __device__ __forceinline__ void matvec(cptr W, cptr bias, const float (&x)[NI], float (&y)[NO]) {
#pragma unroll
  for (int j = 0; j < NO; j++) {
    float a = bias ? bias[j] : 0.0f;
#pragma unroll
    for (int i = 0; i < NI; i++) a = __builtin_fmaf(W[j * LD + i], x[i], a);
    y[j] = a;
  }
}
// y[i] = sum_j W[j * LD + i] g[j]   (transposed product)
template <int NJ, int NI, int LD> __device__ __forceinline__ void matvec_t(cptr W, const float (&g)[NJ], float (&y)[NI]) {
#pragma unroll
  for (int i = 0; i < NI; i++) y[i] = 0.0f;
#pragma unroll
  for (int j = 0; j < NJ; j++) {
#pragma unroll
    for (int i = 0; i < NI; i++) y[i] = __builtin_fmaf(W[j * LD + i], g[j], y[i]);
  }
}
template <int N> __device__ __forceinline__ void tanh_all(float (&h)[N]) {
#pragma unroll
  for (int j = 0; j < N; j++) h[j] = tanhf(h[j]);
}

// one sample's vector into its staging row (float4 stores, zero padded)
template <int N> __device__ __forceinline__ void stage_row(float* row, const float (&v)[N]) {
#pragma unroll
  for (int i = 0; i < N; i += 4)
    *reinterpret_cast<float4*>(row + i) = make_float4(v[i], i + 1 < N ? v[i + 1 < N ? i + 1 : 0] : 0.0f, i + 2 < N ? v[i + 2 < N ? i + 2 : 0] : 0.0f, i + 3 < N ? v[i + 3 < N ? i + 3 : 0] : 0.0f);
}

// FVP: w = S J dir (forward mode) per sample; otherwise w comes from memory.  Then J' w, accumulated per wavefront.
template <int D, int A, bool FVP>
__global__ void __launch_bounds__(64 * WAVES, 1) trpo_kernel(const float* __restrict__ obs, int n, Net th, Net dir, const float* __restrict__ prec, float scale,
                                                         const float* __restrict__ wext, float* __restrict__ partial) {
  typedef Shape<D, A> S;
  constexpr int DP = S::DP;
  constexpr int A3 = (A + 1) / 2;   // rows of W3 a lane accumulates
  __shared__ alignas(16) float bufA[WAVES][64][STG];
  __shared__ alignas(16) float bufB[WAVES][64][STG];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  float pr[A];
#pragma unroll
  for (int a = 0; a < A; a++) pr[a] = FVP ? prec[a] * scale : 0.0f;
  // the parameters this lane accumulates: W2[j2][c2 .. c2 + 15], W1[j2][c1 .. c1 + DP/2 - 1], W3[r3 .. r3 + A3 - 1][i3], one bias each
  const int j2 = lane >> 1, c2 = (lane & 1) * 16, c1 = (lane & 1) * (DP / 2), i3 = lane & 31, r3 = (lane >> 5) * A3;
  float acc2[16], acc1[DP / 2], acc3[A3], accb12 = 0.0f, accb3 = 0.0f;
#pragma unroll
  for (int i = 0; i < 16; i++) acc2[i] = 0.0f;
#pragma unroll
  for (int i = 0; i < DP / 2; i++) acc1[i] = 0.0f;
#pragma unroll
  for (int i = 0; i < A3; i++) acc3[i] = 0.0f;
  float (*sa)[STG] = bufA[wave];
  float (*sb)[STG] = bufB[wave];
  const int ntiles = (n + 63) / 64;
  for (int tile = blockIdx.x * WAVES + wave; tile < ntiles; tile += gridDim.x * WAVES) {
    const CNet W = reissue(th), V = FVP ? reissue(dir) : W;
    const int s = tile * 64 + lane;
    const bool valid = s < n;
    float x[DP];
#pragma unroll
    for (int i = 0; i < DP; i++) x[i] = (valid && i < D) ? obs[(size_t)s * D + i] : 0.0f;
    float xin[D];
#pragma unroll
    for (int i = 0; i < D; i++) xin[i] = x[i];

    float h1[H], h2[H], w[A], g2[H], g1[H];
    matvec<H, D, D>(W.W1, W.b1, xin, h1);
    tanh_all(h1);
    matvec<H, H, H>(W.W2, W.b2, h1, h2);
    tanh_all(h2);
    if (FVP) {
      float dh1[H], dh2[H], t[H], dmu[A], t3[A];
      matvec<H, D, D>(V.W1, V.b1, xin, dh1);
#pragma unroll
      for (int j = 0; j < H; j++) dh1[j] *= 1.0f - h1[j] * h1[j];
      matvec<H, H, H>(W.W2, (cptr)nullptr, dh1, dh2);
      matvec<H, H, H>(V.W2, V.b2, h1, t);
#pragma unroll
      for (int j = 0; j < H; j++) dh2[j] = (dh2[j] + t[j]) * (1.0f - h2[j] * h2[j]);
      matvec<A, H, H>(W.W3, (cptr)nullptr, dh2, dmu);
      matvec<A, H, H>(V.W3, V.b3, h2, t3);
#pragma unroll
      for (int a = 0; a < A; a++) w[a] = valid ? (dmu[a] + t3[a]) * pr[a] : 0.0f;
    } else {
#pragma unroll
      for (int a = 0; a < A; a++) w[a] = valid ? wext[(size_t)s * A + a] : 0.0f;
    }
    // reverse mode: g2 = (W3' w) o (1 - h2^2), g1 = (W2' g2) o (1 - h1^2)
#pragma unroll
    for (int i = 0; i < H; i++) {
      float a = 0.0f;
#pragma unroll
      for (int r = 0; r < A; r++) a = __builtin_fmaf(W.W3[r * H + i], w[r], a);
      g2[i] = a * (1.0f - h2[i] * h2[i]);
    }
    matvec_t<H, H, H>(W.W2, g2, g1);
#pragma unroll
    for (int i = 0; i < H; i++) g1[i] *= 1.0f - h1[i] * h1[i];
    // ---- outer products, 64 samples at a time: stage (cotangent, activation) rows, every lane adds into the parameters it owns
    // (a wavefront's LDS accesses complete in order; the fences keep the compiler from moving them across the phases)
    stage_row<H>(sa[lane], g2); stage_row<H>(sb[lane], h1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 2
    for (int k = 0; k < 64; k++) {
      const float g = sa[k][j2];
#pragma unroll
      for (int i = 0; i < 16; i += 4) {
        const float4 h = *reinterpret_cast<const float4*>(&sb[k][c2 + i]);
        acc2[i] = __builtin_fmaf(g, h.x, acc2[i]); acc2[i + 1] = __builtin_fmaf(g, h.y, acc2[i + 1]);
        acc2[i + 2] = __builtin_fmaf(g, h.z, acc2[i + 2]); acc2[i + 3] = __builtin_fmaf(g, h.w, acc2[i + 3]);
      }
      if (lane < 32) accb12 += sa[k][lane];   // b2
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    stage_row<H>(sa[lane], g1); stage_row<DP>(sb[lane], x);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 2
    for (int k = 0; k < 64; k++) {
      const float g = sa[k][j2];
#pragma unroll
      for (int i = 0; i < DP / 2; i += 2) {
        const float2 xv = *reinterpret_cast<const float2*>(&sb[k][c1 + i]);
        acc1[i] = __builtin_fmaf(g, xv.x, acc1[i]); acc1[i + 1] = __builtin_fmaf(g, xv.y, acc1[i + 1]);
      }
      if (lane >= 32) accb12 += sa[k][lane - 32];   // b1
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    stage_row<A>(sa[lane], w); stage_row<H>(sb[lane], h2);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 2
    for (int k = 0; k < 64; k++) {
      const float h = sb[k][i3];
#pragma unroll
      for (int r = 0; r < A3; r++) acc3[r] = __builtin_fmaf(r3 + r < A ? sa[k][r3 + r] : 0.0f, h, acc3[r]);
      if (lane < A) accb3 += sa[k][lane];   // b3
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  // ---- one row of partial sums per wavefront
  float* out = partial + (size_t)(blockIdx.x * WAVES + wave) * S::NP;
#pragma unroll
  for (int i = 0; i < 16; i++) out[S::O_W2 + j2 * H + c2 + i] = acc2[i];
#pragma unroll
  for (int i = 0; i < DP / 2; i++) if (c1 + i < D) out[S::O_W1 + j2 * D + c1 + i] = acc1[i];
#pragma unroll
  for (int r = 0; r < A3; r++) if (r3 + r < A) out[S::O_W3 + (r3 + r) * H + i3] = acc3[r];
  if (lane < 32) out[S::O_B2 + lane] = accb12; else out[S::O_B1 + lane - 32] = accb12;
  if (lane < A) out[S::O_B3 + lane] = accb3;
}

// ---------------------------------------------------------------------------------------------------------------- policy step
// One lane per environment: float32 view of the observation, mean network (weights through the scalar unit), exploration noise,
// rllab's normalize() action map.  Replaces ~20 torch launches per Env.step of the sampler (0.29 ms at 65 536 envs, r03).
template <int D, int A>
__global__ void __launch_bounds__(256) policy_step_kernel(const double* __restrict__ obs, int n, Net th, const float* __restrict__ log_std,
                                                          const float* __restrict__ noise, const double* __restrict__ low, const double* __restrict__ high,
                                                          float* __restrict__ obs32, float* __restrict__ mean, float* __restrict__ act, double* __restrict__ env_act) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const CNet W{(cptr)th.W1, (cptr)th.b1, (cptr)th.W2, (cptr)th.b2, (cptr)th.W3, (cptr)th.b3};
  float x[D], h1[H], h2[H], mu[A];
#pragma unroll
  for (int i = 0; i < D; i++) { x[i] = (float)obs[(size_t)s * D + i]; obs32[(size_t)s * D + i] = x[i]; }
  matvec<H, D, D>(W.W1, W.b1, x, h1);
  tanh_all(h1);
  matvec<H, H, H>(W.W2, W.b2, h1, h2);
  tanh_all(h2);
  matvec<A, H, H>(W.W3, W.b3, h2, mu);
#pragma unroll
  for (int a = 0; a < A; a++) {
    const float v = mu[a] + noise[(size_t)s * A + a] * expf(((cptr)log_std)[a]);
    mean[(size_t)s * A + a] = mu[a];
    act[(size_t)s * A + a] = v;
    const double lo = low[a], hi = high[a];
    double e = lo + ((double)v + 1.0) * 0.5 * (hi - lo);
    e = e < lo ? lo : (e > hi ? hi : e);
    env_act[(size_t)s * A + a] = e;
  }
}

inline int blocks_for(int n) {
  const int tiles = (n + 63) / 64;
  int b = (tiles + WAVES - 1) / WAVES;
  return b < 1 ? 1 : (b > MAX_BLOCKS ? MAX_BLOCKS : b);
}

template <bool FVP>
int launch(const float* obs, int n, int D, int A, const Net& th, const Net& dir, const float* prec, float scale, const float* wext, float* partial, hipStream_t s) {
  if (!obs || n <= 0 || !partial) return CASSIE_EINVAL;
  const dim3 grid(blocks_for(n)), block(64 * WAVES);
  if (D == 26 && A == 6) hipLaunchKernelGGL((trpo_kernel<26, 6, FVP>), grid, block, 0, s, obs, n, th, dir, prec, scale, wext, partial);
  else if (D == 26 && A == 7) hipLaunchKernelGGL((trpo_kernel<26, 7, FVP>), grid, block, 0, s, obs, n, th, dir, prec, scale, wext, partial);
  else if (D == 17 && A == 6) hipLaunchKernelGGL((trpo_kernel<17, 6, FVP>), grid, block, 0, s, obs, n, th, dir, prec, scale, wext, partial);
  else if (D == 17 && A == 7) hipLaunchKernelGGL((trpo_kernel<17, 7, FVP>), grid, block, 0, s, obs, n, th, dir, prec, scale, wext, partial);
  else return CASSIE_EINVAL;
  return hipGetLastError() == hipSuccess ? CASSIE_OK : CASSIE_EHIP;
}

}  // namespace cassie_trpo

extern "C" {

int CassieTrpoParamCount(int obs_dim, int act_dim) {
  if ((obs_dim != 26 && obs_dim != 17) || (act_dim != 6 && act_dim != 7)) return 0;
  return 32 * obs_dim + 32 + 32 * 32 + 32 + act_dim * 32 + act_dim;
}
int CassieTrpoPartialRows(int n_samples) { return cassie_trpo::blocks_for(n_samples) * cassie_trpo::WAVES; }

int CassieTrpoFvp(const float* obs_dev, int n, int obs_dim, int act_dim, const float* W1, const float* b1, const float* W2, const float* b2,
                  const float* W3, const float* b3, const float* dW1, const float* db1, const float* dW2, const float* db2, const float* dW3,
                  const float* db3, const float* prec, float scale, float* partial_dev, void* stream) {
  if (!W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !dW1 || !db1 || !dW2 || !db2 || !dW3 || !db3 || !prec) return CASSIE_EINVAL;
  const cassie_trpo::Net th{W1, b1, W2, b2, W3, b3}, dir{dW1, db1, dW2, db2, dW3, db3};
  return cassie_trpo::launch<true>(obs_dev, n, obs_dim, act_dim, th, dir, prec, scale, nullptr, partial_dev, (hipStream_t)stream);
}

int CassieTrpoVjp(const float* obs_dev, int n, int obs_dim, int act_dim, const float* W1, const float* b1, const float* W2, const float* b2,
                  const float* W3, const float* b3, const float* w_dev, float* partial_dev, void* stream) {
  if (!W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !w_dev) return CASSIE_EINVAL;
  const cassie_trpo::Net th{W1, b1, W2, b2, W3, b3};
  return cassie_trpo::launch<false>(obs_dev, n, obs_dim, act_dim, th, th, nullptr, 0.0f, w_dev, partial_dev, (hipStream_t)stream);
}

int CassieTrpoPolicyStep(const double* obs_dev, int n, int obs_dim, int act_dim, const float* W1, const float* b1, const float* W2,
                         const float* b2, const float* W3, const float* b3, const float* log_std, const float* noise_dev,
                         const double* low_dev, const double* high_dev, float* obs32_dev, float* mean_dev, float* act_dev,
                         double* env_actions_dev, void* stream) {
  if (!obs_dev || n <= 0 || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !log_std || !noise_dev || !low_dev || !high_dev || !obs32_dev || !mean_dev ||
      !act_dev || !env_actions_dev)
    return CASSIE_EINVAL;
  const cassie_trpo::Net th{W1, b1, W2, b2, W3, b3};
  const dim3 grid((n + 255) / 256), block(256);
  hipStream_t s = (hipStream_t)stream;
  using namespace cassie_trpo;
  if (obs_dim == 26 && act_dim == 6) hipLaunchKernelGGL((policy_step_kernel<26, 6>), grid, block, 0, s, obs_dev, n, th, log_std, noise_dev, low_dev, high_dev, obs32_dev, mean_dev, act_dev, env_actions_dev);
  else if (obs_dim == 26 && act_dim == 7) hipLaunchKernelGGL((policy_step_kernel<26, 7>), grid, block, 0, s, obs_dev, n, th, log_std, noise_dev, low_dev, high_dev, obs32_dev, mean_dev, act_dev, env_actions_dev);
  else return CASSIE_EINVAL;
  return hipGetLastError() == hipSuccess ? CASSIE_OK : CASSIE_EHIP;
}

}  // extern "C"
