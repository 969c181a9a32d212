// tu_trpo.hip -- fused policy kernels of the TRPO outer loop (include/cassie_trpo.h; rllab/envs/trpo_cassie.py:21-42).
//
// One TRPO update evaluates, on the 524 288 samples one rank collects per iteration, the policy gradient and ~11 Fisher-vector
// products of the 26-32-32-6 tanh policy.  As torch operations a product is ~8 skinny GEMMs (K = 32) and a dozen element-wise passes
// over [N, 32] tensors: 0.67 ms each, 7.3 ms per update (r04 profile, tools/prof_trpo_update.py).  Here it is ONE launch on the
// matrix cores, exact float32 (v_mfma_f32_32x32x2_f32 = a k-ordered fmaf chain):
//
//   * a wavefront owns a tile of 32 samples; every activation-like quantity X (hidden units x samples, 32 x 32) lives in the
//     accumulator layout -- the sample on the lane (column = lane & 31), the hidden unit in the 16 registers
//     (row r(v, h) = (v & 3) + 8 (v >> 2) + 4 h, h = lane >> 5) -- so element-wise work (tanh, 1 - h^2, the precision) is register-wise;
//   * a layer Y = W X takes X's registers AS ITS B OPERAND with no data movement: k-step v sums over row r(v, h), and the A operand
//     of that step is W[i][r(v, h)] (lane i, half h), read from memory ONCE per wavefront in exactly that order and kept in registers
//     for all tiles (W1, W2, W3, their directions, W2', W3': 110 registers);
//   * the parameter gradients are products over the SAMPLE index (gW2 = G2 H1', ...): both operands are needed with the hidden unit on
//     the lane, i.e. transposed -- one pass through a per-wavefront LDS tile (16 ds_write_b32 + 4 ds_read_b128 per matrix and lane);
//     the observations are read from HBM in both forms (104 B per sample, twice);
//   * the 6-wide output layer is padded to 32 rows (64 of the 174 MFMAs of a tile do 6 useful rows of 32): simpler and exact, and still
//     ~80 us of matrix-core time per product against 670 us for the torch operations;
//   * gradients are accumulated in three accumulator tiles per wavefront over all its tiles; bias gradients are row sums (b1 rides on
//     a column of ones next to the observations).  Each wavefront writes one row of partial sums; the caller adds the rows (and
//     all-reduces over ranks where it did before).
// r04 history: a one-lane-per-sample FP32 vector version (weights through the scalar unit) was parity-green and SLOWER than torch
// (0.80 ms against 0.67; weights in LDS 1.44 ms: the compiler hoists the weight reads and spills) -- replaced by this one.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/cassie_trpo.h"
#include "../../include/cassie_vec.h"

namespace cassie_trpo {

constexpr int H = 32;         // hidden units (both layers)
constexpr int TP = 36;        // floats per row of a transpose tile (16-byte aligned rows; ds_read_b128 of 16 lanes conflict-free)
constexpr int WAVES = 4;      // wavefronts per workgroup: one per SIMD
constexpr int MAX_BLOCKS = 512;   // two workgroups of four wavefronts per CU

template <int D, int A> struct Shape {
  static constexpr int NP = H * D + H + H * H + H + A * H + A;
  static constexpr int O_W1 = 0, O_B1 = H * D, O_W2 = O_B1 + H, O_B2 = O_W2 + H * H, O_W3 = O_B2 + H, O_B3 = O_W3 + A * H;
};

struct Net { const float *W1, *b1, *W2, *b2, *W3, *b3; };

typedef float v16f __attribute__((ext_vector_type(16)));
#define TRPO_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// tanh through the hardware exp2 / rcp: 1 - 2 / (e^2x + 1); absolute error ~1e-7 (the saturated ends are exact: e = inf or 0)
__device__ __forceinline__ float tanh_fast(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}
__device__ __forceinline__ void wave_lds_sync() {   // a wavefront's LDS accesses complete in order; this keeps the compiler from moving them
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// FVP: w = S J dir (forward mode) per sample; otherwise w comes from memory.  Then J' w, accumulated per wavefront.
// The A operands (weights in k-step order, see the header) are the same for every tile and every wavefront: the workgroup lays them
// out ONCE in LDS as [quad of k-steps][lane] float4 images (29 quads, 29.7 KB) and a wavefront reads the four quads of a product right
// before its 16 MFMAs -- conflict-free ds_read_b128, 29 per tile.  Holding them in registers instead (110 per lane, the first build of
// this kernel) costs the second wavefront per SIMD: with two, one wavefront's tanh / transposes run under the other's MFMAs
// (0.134 -> see DESIGN.md section 8 ms per product at 524 288 samples).
enum { Q_W1 = 0, Q_DW1 = 4, Q_W2 = 8, Q_DW2 = 12, Q_W3 = 16, Q_DW3 = 20, Q_W2T = 24, Q_W3T = 28, Q_N = 29 };
template <int D, int A, bool FVP>
__global__ void __launch_bounds__(64 * WAVES, 2) trpo_kernel(const float* __restrict__ obs, int n, Net th, Net dir, const float* __restrict__ prec, float scale,
                                                         const float* __restrict__ wext, float* __restrict__ partial) {
  typedef Shape<D, A> S;
  static_assert(D < 32 && A <= 8, "a column of ones next to the observations; the cotangent rows in registers 0..3 of the two lane halves");
  constexpr int KS1 = (D + 1) / 2;   // k-steps of the first layer (k = 2 s + h)
  __shared__ alignas(16) float tile[WAVES][2][32 * TP];
  __shared__ alignas(16) float sbias[5][32];   // b1, b2, db1, db2, db3 (zero padded)
  __shared__ float4 wimg[Q_N][64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, h = lane >> 5;
  if (tid < 32) {
    sbias[0][tid] = th.b1[tid]; sbias[1][tid] = th.b2[tid];
    sbias[2][tid] = FVP ? dir.b1[tid] : 0.0f; sbias[3][tid] = FVP ? dir.b2[tid] : 0.0f; sbias[4][tid] = (FVP && tid < A) ? dir.b3[tid] : 0.0f;
  }
  // ---- A operands, once per workgroup: element e of quad q = k-step 4 (q mod 4) + e of its matrix, for lane (i = c, h)
  for (int q = wave; q < Q_N; q += WAVES) {
    float v4[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int st = 4 * (q & 3) + e;                       // k-step within the matrix
      const int r = (st & 3) + 8 * (st >> 2) + 4 * h;       // accumulator-order row of that step (layers 2, 3 and the transposes)
      const int k1 = 2 * st + h;                            // first layer: k = 2 s + h
      float x = 0.0f;
      if (q < Q_DW1) x = (st < KS1 && k1 < D) ? th.W1[c * D + k1] : 0.0f;
      else if (q < Q_W2) x = (FVP && st < KS1 && k1 < D) ? dir.W1[c * D + k1] : 0.0f;
      else if (q < Q_DW2) x = th.W2[c * H + r];
      else if (q < Q_W3) x = FVP ? dir.W2[c * H + r] : 0.0f;
      else if (q < Q_DW3) x = c < A ? th.W3[c * H + r] : 0.0f;
      else if (q < Q_W2T) x = (FVP && c < A) ? dir.W3[c * H + r] : 0.0f;
      else if (q < Q_W3T) x = th.W2[r * H + c];
      else x = (e + 4 * h < A) ? th.W3[(e + 4 * h) * H + c] : 0.0f;   // W3': cotangent row a = v + 4 h
      v4[e] = x;
    }
    wimg[q][lane] = make_float4(v4[0], v4[1], v4[2], v4[3]);
  }
  float pr[4];
#pragma unroll
  for (int v = 0; v < 4; v++) { const int a = v + 4 * h; pr[v] = (FVP && a < A) ? prec[a] * scale : 0.0f; }
  __syncthreads();
  auto aop = [&](int q0, float (&a)[16]) {   // the 16 k-steps of a product
#pragma unroll
    for (int q = 0; q < 4; q++) { const float4 w = wimg[q0 + q][lane]; a[4 * q] = w.x; a[4 * q + 1] = w.y; a[4 * q + 2] = w.z; a[4 * q + 3] = w.w; }
  };
  auto bias_tile = [&](int which) {   // C operand: bias[r(v, h)] in register v (rows 8 g + 4 h .. + 3 are one float4)
    v16f z;
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const float4 b = *reinterpret_cast<const float4*>(&sbias[which][8 * g + 4 * h]);
      z[4 * g] = b.x; z[4 * g + 1] = b.y; z[4 * g + 2] = b.z; z[4 * g + 3] = b.w;
    }
    return z;
  };
  float* t0 = tile[wave][0];
  float* t1 = tile[wave][1];
  auto put = [&](float* t, const v16f& x) {   // accumulator layout -> [row][sample] image
#pragma unroll
    for (int v = 0; v < 16; v++) t[((v & 3) + 8 * (v >> 2) + 4 * h) * TP + c] = x[v];
  };
  auto get = [&](const float* t, float (&y)[16]) {   // lane (i = c, h): row i, samples 16 h .. 16 h + 15
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const float4 b = *reinterpret_cast<const float4*>(&t[c * TP + 16 * h + 4 * q]);
      y[4 * q] = b.x; y[4 * q + 1] = b.y; y[4 * q + 2] = b.z; y[4 * q + 3] = b.w;
    }
  };
  v16f gW1, gW2, gW3;
#pragma unroll
  for (int v = 0; v < 16; v++) { gW1[v] = 0.0f; gW2[v] = 0.0f; gW3[v] = 0.0f; }
  float gb2 = 0.0f, gb3 = 0.0f;
  const int ntiles = (n + 31) / 32;
  for (int tl = blockIdx.x * WAVES + wave; tl < ntiles; tl += gridDim.x * WAVES) {
    const int s0 = tl * 32, smp = s0 + c;
    const bool valid = smp < n;
    // observations: as the B operand of the first layer (sample on the lane) and transposed (feature on the lane; column D = ones)
    float xb[KS1], xt[16], aw[16];
#pragma unroll
    for (int s = 0; s < KS1; s++) { const int k = 2 * s + h; xb[s] = (valid && k < D) ? obs[(size_t)smp * D + k] : 0.0f; }
#pragma unroll
    for (int s = 0; s < 16; s++) {
      const int sm = s0 + 16 * h + s;
      xt[s] = c < D ? (sm < n ? obs[(size_t)sm * D + c] : 0.0f) : (c == D ? 1.0f : 0.0f);
    }
    v16f h1 = bias_tile(0);
    aop(Q_W1, aw);
#pragma unroll
    for (int s = 0; s < KS1; s++) h1 = TRPO_MFMA(aw[s], xb[s], h1);
    v16f wt;   // cotangent on the mean, rows a = v + 4 h in registers v = 0 .. 3
    v16f d1;
    if (FVP) {
      d1 = bias_tile(2);
      aop(Q_DW1, aw);
#pragma unroll
      for (int s = 0; s < KS1; s++) d1 = TRPO_MFMA(aw[s], xb[s], d1);
    }
#pragma unroll
    for (int v = 0; v < 16; v++) h1[v] = tanh_fast(h1[v]);
    v16f h2 = bias_tile(1);
    aop(Q_W2, aw);
#pragma unroll
    for (int v = 0; v < 16; v++) h2 = TRPO_MFMA(aw[v], h1[v], h2);
    if (FVP) {
      v16f d2 = bias_tile(3);
#pragma unroll
      for (int v = 0; v < 16; v++) d1[v] *= 1.0f - h1[v] * h1[v];
#pragma unroll
      for (int v = 0; v < 16; v++) d2 = TRPO_MFMA(aw[v], d1[v], d2);   // W2 dH1
      aop(Q_DW2, aw);
#pragma unroll
      for (int v = 0; v < 16; v++) d2 = TRPO_MFMA(aw[v], h1[v], d2);   // + dW2 H1
#pragma unroll
      for (int v = 0; v < 16; v++) h2[v] = tanh_fast(h2[v]);
      v16f dm = bias_tile(4);
      aop(Q_DW3, aw);
#pragma unroll
      for (int v = 0; v < 16; v++) dm = TRPO_MFMA(aw[v], h2[v], dm);
#pragma unroll
      for (int v = 0; v < 16; v++) d2[v] *= 1.0f - h2[v] * h2[v];
      aop(Q_W3, aw);
#pragma unroll
      for (int v = 0; v < 16; v++) dm = TRPO_MFMA(aw[v], d2[v], dm);
#pragma unroll
      for (int v = 0; v < 16; v++) wt[v] = (v < 4 && valid) ? dm[v] * pr[v] : 0.0f;
    } else {
#pragma unroll
      for (int v = 0; v < 16; v++) h2[v] = tanh_fast(h2[v]);
#pragma unroll
      for (int v = 0; v < 16; v++) wt[v] = (v < 4 && valid && v + 4 * h < A) ? wext[(size_t)smp * A + v + 4 * h] : 0.0f;
    }
    // reverse mode: G2 = (W3' w) o (1 - H2^2), G1 = (W2' G2) o (1 - H1^2)
    v16f g2, g1;
#pragma unroll
    for (int v = 0; v < 16; v++) { g2[v] = 0.0f; g1[v] = 0.0f; }
    {
      const float4 w = wimg[Q_W3T][lane];
      g2 = TRPO_MFMA(w.x, wt[0], g2); g2 = TRPO_MFMA(w.y, wt[1], g2); g2 = TRPO_MFMA(w.z, wt[2], g2); g2 = TRPO_MFMA(w.w, wt[3], g2);
    }
#pragma unroll
    for (int v = 0; v < 16; v++) g2[v] *= 1.0f - h2[v] * h2[v];
    aop(Q_W2T, aw);
#pragma unroll
    for (int v = 0; v < 16; v++) g1 = TRPO_MFMA(aw[v], g2[v], g1);
#pragma unroll
    for (int v = 0; v < 16; v++) g1[v] *= 1.0f - h1[v] * h1[v];
    // ---- parameter gradients: products over the sample index, operands transposed through the wavefront's LDS tiles
    float ta[16], tb[16];
    put(t0, g2); put(t1, h1);
    wave_lds_sync();
    get(t0, ta); get(t1, tb);
    wave_lds_sync();
#pragma unroll
    for (int s = 0; s < 16; s++) { gW2 = TRPO_MFMA(ta[s], tb[s], gW2); gb2 += ta[s]; }
    put(t0, g1);
    wave_lds_sync();
    get(t0, ta);
    wave_lds_sync();
#pragma unroll
    for (int s = 0; s < 16; s++) gW1 = TRPO_MFMA(ta[s], xt[s], gW1);
    put(t0, wt); put(t1, h2);
    wave_lds_sync();
    get(t0, ta); get(t1, tb);
    wave_lds_sync();
#pragma unroll
    for (int s = 0; s < 16; s++) { gW3 = TRPO_MFMA(ta[s], tb[s], gW3); gb3 += ta[s]; }
  }
  // ---- one row of partial sums per wavefront: gW[r(v, h)][c] in register v
  float* out = partial + (size_t)(blockIdx.x * WAVES + wave) * S::NP;
#pragma unroll
  for (int v = 0; v < 16; v++) {
    const int r = (v & 3) + 8 * (v >> 2) + 4 * h;
    out[S::O_W2 + r * H + c] = gW2[v];
    if (c < D) out[S::O_W1 + r * D + c] = gW1[v];
    if (c == D) out[S::O_B1 + r] = gW1[v];
    if (r < A) out[S::O_W3 + r * H + c] = gW3[v];
  }
  gb2 += __shfl_xor(gb2, 32, 64); gb3 += __shfl_xor(gb3, 32, 64);
  if (h == 0) out[S::O_B2 + c] = gb2;
  if (h == 0 && c < A) out[S::O_B3 + c] = gb3;
}

// Line search of the TRPO step (rllab's f_loss / f_constraint on the sampled batch): surrogate loss terms -exp(ll_new - ll_old) adv and
// KL(old || new) per sample (GaussianMLPPolicy.log_likelihood / .kl of trpo.py), summed per wavefront in float64.  The forward pass is
// the one of trpo_kernel (same tiles, same operands); the mean comes out with action a = v + 4 h in register v < 4 of lane (sample, h).
template <int D, int A>
__global__ void __launch_bounds__(64 * WAVES, 1) surrogate_kernel(const float* __restrict__ obs, int n, Net th, const float* __restrict__ ls_new,
                                                              const float* __restrict__ ls_old, const float* __restrict__ act, const float* __restrict__ adv,
                                                              const float* __restrict__ old_mean, double* __restrict__ partial) {
  constexpr int KS1 = (D + 1) / 2;
  __shared__ alignas(16) float sbias[3][32];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, h = lane >> 5;
  if (tid < 32) { sbias[0][tid] = th.b1[tid]; sbias[1][tid] = th.b2[tid]; sbias[2][tid] = tid < A ? th.b3[tid] : 0.0f; }
  float aW1[KS1], aW2[16], aW3[16], isn[4], iso[4], dls[4], kden[4], kvar[4];
#pragma unroll
  for (int s = 0; s < KS1; s++) { const int k = 2 * s + h; aW1[s] = k < D ? th.W1[c * D + k] : 0.0f; }
#pragma unroll
  for (int v = 0; v < 16; v++) {
    const int r = (v & 3) + 8 * (v >> 2) + 4 * h;
    aW2[v] = th.W2[c * H + r];
    aW3[v] = c < A ? th.W3[c * H + r] : 0.0f;
  }
#pragma unroll
  for (int v = 0; v < 4; v++) {
    const int a = v + 4 * h;
    const float ln = a < A ? ls_new[a] : 0.0f, lo = a < A ? ls_old[a] : 0.0f;
    const float sn = expf(ln), so = expf(lo);
    isn[v] = 1.0f / sn; iso[v] = 1.0f / so; dls[v] = ln - lo;
    kden[v] = 1.0f / (2.0f * sn * sn + 1e-8f); kvar[v] = so * so - sn * sn;
  }
  __syncthreads();
  auto bias_tile = [&](int which) {
    v16f z;
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const float4 b = *reinterpret_cast<const float4*>(&sbias[which][8 * g + 4 * h]);
      z[4 * g] = b.x; z[4 * g + 1] = b.y; z[4 * g + 2] = b.z; z[4 * g + 3] = b.w;
    }
    return z;
  };
  double accL = 0.0, accK = 0.0;
  const int ntiles = (n + 31) / 32;
  for (int tl = blockIdx.x * WAVES + wave; tl < ntiles; tl += gridDim.x * WAVES) {
    const int smp = tl * 32 + c;
    const bool valid = smp < n;
    float xb[KS1];
#pragma unroll
    for (int s = 0; s < KS1; s++) { const int k = 2 * s + h; xb[s] = (valid && k < D) ? obs[(size_t)smp * D + k] : 0.0f; }
    float ac[4], om[4];
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const bool on = valid && v + 4 * h < A;
      ac[v] = on ? act[(size_t)smp * A + v + 4 * h] : 0.0f; om[v] = on ? old_mean[(size_t)smp * A + v + 4 * h] : 0.0f;
    }
    const float ad = valid ? adv[smp] : 0.0f;
    v16f h1 = bias_tile(0);
#pragma unroll
    for (int s = 0; s < KS1; s++) h1 = TRPO_MFMA(aW1[s], xb[s], h1);
#pragma unroll
    for (int v = 0; v < 16; v++) h1[v] = tanh_fast(h1[v]);
    v16f h2 = bias_tile(1);
#pragma unroll
    for (int v = 0; v < 16; v++) h2 = TRPO_MFMA(aW2[v], h1[v], h2);
#pragma unroll
    for (int v = 0; v < 16; v++) h2[v] = tanh_fast(h2[v]);
    v16f mu = bias_tile(2);
#pragma unroll
    for (int v = 0; v < 16; v++) mu = TRPO_MFMA(aW3[v], h2[v], mu);
    float ll = 0.0f, kl = 0.0f;
#pragma unroll
    for (int v = 0; v < 4; v++) {
      if (v + 4 * h < A) {
        const float zn = (ac[v] - mu[v]) * isn[v], zo = (ac[v] - om[v]) * iso[v], dm = om[v] - mu[v];
        ll += 0.5f * (zo * zo - zn * zn) - dls[v];
        kl += (dm * dm + kvar[v]) * kden[v] + dls[v];
      }
    }
    ll += __shfl_xor(ll, 32, 64); kl += __shfl_xor(kl, 32, 64);
    if (h == 0 && valid) { accL -= (double)(expf(ll) * ad); accK += (double)kl; }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { accL += __shfl_xor(accL, m, 64); accK += __shfl_xor(accK, m, 64); }
  if (lane == 0) { double* out = partial + (size_t)(blockIdx.x * WAVES + wave) * 2; out[0] = accL; out[1] = accK; }
}

// The weights are the same for every lane: read through the CONSTANT address space at compile-time offsets they are scalar loads
// (s_load through the scalar cache) and the multiply-adds take them as SGPR operands -- no LDS traffic, no vector registers.
typedef const __attribute__((address_space(4))) float* cptr;
struct CNet { cptr W1, b1, W2, b2, W3, b3; };
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
template <int N> __device__ __forceinline__ void tanh_all(float (&h)[N]) {
#pragma unroll
  for (int j = 0; j < N; j++) h[j] = tanhf(h[j]);
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

// The same policy step on the matrix cores (r04): a wavefront per tile of 32 environments, the forward pass of trpo_kernel (A operands
// from memory once per wavefront, activations in the accumulator layout), the mean with action a = v + 4 h in register v < 4 of lane
// (environment, h).  tanh through the hardware exp2 / rcp as in the update's kernels, so that the mean the sampler records IS the mean
// the line search evaluates at the same weights.  33 -> see DESIGN.md section 8 us per step of 65 536 environments.
template <int D, int A>
__global__ void __launch_bounds__(64 * WAVES, 2) policy_step_mfma_kernel(const double* __restrict__ obs, int n, Net th, const float* __restrict__ log_std,
                                                                     const float* __restrict__ noise, const double* __restrict__ low, const double* __restrict__ high,
                                                                     float* __restrict__ obs32, float* __restrict__ mean, float* __restrict__ act, double* __restrict__ env_act) {
  constexpr int KS1 = (D + 1) / 2;
  __shared__ alignas(16) float sbias[3][32];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, h = lane >> 5;
  if (tid < 32) { sbias[0][tid] = th.b1[tid]; sbias[1][tid] = th.b2[tid]; sbias[2][tid] = tid < A ? th.b3[tid] : 0.0f; }
  float aW1[KS1], aW2[16], aW3[16], sd[4];
  double lo[4], hi[4];
#pragma unroll
  for (int s = 0; s < KS1; s++) { const int k = 2 * s + h; aW1[s] = k < D ? th.W1[c * D + k] : 0.0f; }
#pragma unroll
  for (int v = 0; v < 16; v++) {
    const int r = (v & 3) + 8 * (v >> 2) + 4 * h;
    aW2[v] = th.W2[c * H + r];
    aW3[v] = c < A ? th.W3[c * H + r] : 0.0f;
  }
#pragma unroll
  for (int v = 0; v < 4; v++) {
    const int a = v + 4 * h;
    sd[v] = a < A ? expf(log_std[a]) : 0.0f;
    lo[v] = a < A ? low[a] : 0.0; hi[v] = a < A ? high[a] : 0.0;
  }
  __syncthreads();
  auto bias_tile = [&](int which) {
    v16f z;
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const float4 b = *reinterpret_cast<const float4*>(&sbias[which][8 * g + 4 * h]);
      z[4 * g] = b.x; z[4 * g + 1] = b.y; z[4 * g + 2] = b.z; z[4 * g + 3] = b.w;
    }
    return z;
  };
  const int ntiles = (n + 31) / 32;
  for (int tl = blockIdx.x * WAVES + wave; tl < ntiles; tl += gridDim.x * WAVES) {
    const int smp = tl * 32 + c;
    const bool valid = smp < n;
    float xb[KS1];
#pragma unroll
    for (int s = 0; s < KS1; s++) {
      const int k = 2 * s + h;
      const bool on = valid && k < D;
      xb[s] = on ? (float)obs[(size_t)smp * D + k] : 0.0f;
      if (on) obs32[(size_t)smp * D + k] = xb[s];
    }
    v16f h1 = bias_tile(0);
#pragma unroll
    for (int s = 0; s < KS1; s++) h1 = TRPO_MFMA(aW1[s], xb[s], h1);
#pragma unroll
    for (int v = 0; v < 16; v++) h1[v] = tanh_fast(h1[v]);
    v16f h2 = bias_tile(1);
#pragma unroll
    for (int v = 0; v < 16; v++) h2 = TRPO_MFMA(aW2[v], h1[v], h2);
#pragma unroll
    for (int v = 0; v < 16; v++) h2[v] = tanh_fast(h2[v]);
    v16f mu = bias_tile(2);
#pragma unroll
    for (int v = 0; v < 16; v++) mu = TRPO_MFMA(aW3[v], h2[v], mu);
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const int a = v + 4 * h;
      if (valid && a < A) {
        const size_t o = (size_t)smp * A + a;
        const float val = mu[v] + noise[o] * sd[v];
        mean[o] = mu[v]; act[o] = val;
        double e = lo[v] + ((double)val + 1.0) * 0.5 * (hi[v] - lo[v]);
        e = e < lo[v] ? lo[v] : (e > hi[v] ? hi[v] : e);
        env_act[o] = e;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------- sampler step
// Per-path clocks and returns of the sampler for one Env.step, one lane per environment (trpo.collect did this with ~20 element-wise
// torch launches per step; rllab's sampler keeps the same quantities on the host).  Episode statistics: one (count, summed return) pair
// per workgroup by a fixed-order tree, so the caller's sum over the rows is the same number on every run (a resumed run logs what the
// uninterrupted one does).
__global__ void __launch_bounds__(256) sampler_step_kernel(const double* __restrict__ rew, const uint8_t* __restrict__ done, int n, long long max_len,
                                                          long long* __restrict__ path_t, double* __restrict__ path_ret, double* __restrict__ rew_row,
                                                          long long* __restrict__ t_row, uint8_t* __restrict__ cut_row, double* __restrict__ partial) {
  __shared__ double red[2][256];
  const int i = blockIdx.x * 256 + threadIdx.x;
  double cnt = 0.0, sum = 0.0;
  if (i < n) {
    const double r = rew[i];
    const long long t = path_t[i];
    rew_row[i] = r; t_row[i] = t;
    const double ret = path_ret[i] + r;
    const bool cut = done[i] != 0 || t + 1 >= max_len;   // rllab truncates paths at max_path_length
    cut_row[i] = cut ? 1 : 0;
    cnt = cut ? 1.0 : 0.0; sum = cut ? ret : 0.0;
    path_ret[i] = cut ? 0.0 : ret;
    path_t[i] = cut ? 0 : t + 1;
  }
  red[0][threadIdx.x] = cnt; red[1][threadIdx.x] = sum;
  __syncthreads();
  for (int m = 128; m >= 1; m >>= 1) {
    if ((int)threadIdx.x < m) { red[0][threadIdx.x] += red[0][threadIdx.x + m]; red[1][threadIdx.x] += red[1][threadIdx.x + m]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { partial[2 * blockIdx.x] = red[0][0]; partial[2 * blockIdx.x + 1] = red[1][0]; }
}

// ---------------------------------------------------------------------------------------------------------------- CG vector step
// What rllab's conjugate gradient (rllab/misc/krylov.py: cg) does between two Fisher-vector products, on the ~2 k-entry parameter
// vector, in one workgroup: F p from the mean network's part (summed over wavefronts and ranks by the caller) + the log-std block +
// the damping, step length, x and r updates, new residual, early exit as a frozen step length (no host read-back), new direction.
// As torch operations: ~15 launches per iteration, a third of what an iteration costs next to the 0.12 ms product.
__device__ __forceinline__ float block_sum_1024(float v, float* red) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.0f;
#pragma unroll
  for (int i = 0; i < 16; i++) t += red[i];   // every thread adds the 16 wavefront sums in the same order
  return t;
}
__global__ void __launch_bounds__(1024) cg_update_kernel(int n, int ls_off, int n_ls, const float* __restrict__ Apm, const float* __restrict__ hls, float reg,
                                                        float tol, float* __restrict__ x, float* __restrict__ r, float* __restrict__ p, float* __restrict__ scal) {
  __shared__ float red[16];
  constexpr int PER = 3;   // n <= 3072
  float pv[PER], ap[PER], rv[PER];
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < PER; k++) {
    const int i = threadIdx.x + 1024 * k;
    pv[k] = 0.0f; ap[k] = 0.0f; rv[k] = 0.0f;
    if (i < n) {
      pv[k] = p[i]; rv[k] = r[i];
      const bool ls = i >= ls_off && i < ls_off + n_ls;
      const float f = ls ? hls[i - ls_off] * pv[k] : Apm[i < ls_off ? i : i - n_ls];
      ap[k] = f + reg * pv[k];
      s += pv[k] * ap[k];
    }
  }
  const float pAp = block_sum_1024(s, red);
  const float rr = scal[0];
  const bool running = scal[1] != 0.0f;
  const float alpha = running ? rr / pAp : 0.0f;
  s = 0.0f;
#pragma unroll
  for (int k = 0; k < PER; k++) {
    const int i = threadIdx.x + 1024 * k;
    if (i < n) { x[i] += alpha * pv[k]; rv[k] -= alpha * ap[k]; r[i] = rv[k]; s += rv[k] * rv[k]; }
  }
  const float rr_new = block_sum_1024(s, red);
  const bool go_on = running && rr_new >= tol;
  const float beta = go_on ? rr_new / rr : 0.0f;
#pragma unroll
  for (int k = 0; k < PER; k++) {
    const int i = threadIdx.x + 1024 * k;
    if (i < n) p[i] = rv[k] + beta * pv[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) { scal[0] = rr_new; scal[1] = go_on ? 1.0f : 0.0f; }
}

inline int blocks_for(int n) {
  const int tiles = (n + 31) / 32;
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

int CassieTrpoSurrogate(const float* obs_dev, int n, int obs_dim, int act_dim, const float* W1, const float* b1, const float* W2, const float* b2,
                        const float* W3, const float* b3, const float* log_std_new, const float* log_std_old, const float* act_dev,
                        const float* adv_dev, const float* old_mean_dev, double* partial_dev, void* stream) {
  if (!obs_dev || n <= 0 || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !log_std_new || !log_std_old || !act_dev || !adv_dev || !old_mean_dev || !partial_dev)
    return CASSIE_EINVAL;
  const cassie_trpo::Net th{W1, b1, W2, b2, W3, b3};
  const dim3 grid(cassie_trpo::blocks_for(n)), block(64 * cassie_trpo::WAVES);
  hipStream_t s = (hipStream_t)stream;
  using namespace cassie_trpo;
  if (obs_dim == 26 && act_dim == 6) hipLaunchKernelGGL((surrogate_kernel<26, 6>), grid, block, 0, s, obs_dev, n, th, log_std_new, log_std_old, act_dev, adv_dev, old_mean_dev, partial_dev);
  else if (obs_dim == 26 && act_dim == 7) hipLaunchKernelGGL((surrogate_kernel<26, 7>), grid, block, 0, s, obs_dev, n, th, log_std_new, log_std_old, act_dev, adv_dev, old_mean_dev, partial_dev);
  else if (obs_dim == 17 && act_dim == 6) hipLaunchKernelGGL((surrogate_kernel<17, 6>), grid, block, 0, s, obs_dev, n, th, log_std_new, log_std_old, act_dev, adv_dev, old_mean_dev, partial_dev);
  else if (obs_dim == 17 && act_dim == 7) hipLaunchKernelGGL((surrogate_kernel<17, 7>), grid, block, 0, s, obs_dev, n, th, log_std_new, log_std_old, act_dev, adv_dev, old_mean_dev, partial_dev);
  else return CASSIE_EINVAL;
  return hipGetLastError() == hipSuccess ? CASSIE_OK : CASSIE_EHIP;
}

int CassieTrpoCgUpdate(int n, int ls_off, int n_ls, const float* Ap_mean_dev, const float* hls_dev, float reg, float tol, float* x_dev, float* r_dev, float* p_dev,
                       float* scal_dev, void* stream) {
  if (n <= 0 || n > 3072 || ls_off < 0 || n_ls < 0 || ls_off + n_ls > n || !Ap_mean_dev || !hls_dev || !x_dev || !r_dev || !p_dev || !scal_dev) return CASSIE_EINVAL;
  hipLaunchKernelGGL(cassie_trpo::cg_update_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, n, ls_off, n_ls, Ap_mean_dev, hls_dev, reg, tol, x_dev, r_dev, p_dev, scal_dev);
  return hipGetLastError() == hipSuccess ? CASSIE_OK : CASSIE_EHIP;
}

int CassieTrpoSamplerRows(int n_envs) { return n_envs > 0 ? (n_envs + 255) / 256 : 0; }

int CassieTrpoSamplerStep(const double* rew_dev, const unsigned char* done_dev, int n, long long max_path_length, long long* path_t_dev, double* path_ret_dev,
                          double* rew_row_dev, long long* t_row_dev, unsigned char* cut_row_dev, double* partial_dev, void* stream) {
  if (!rew_dev || !done_dev || n <= 0 || !path_t_dev || !path_ret_dev || !rew_row_dev || !t_row_dev || !cut_row_dev || !partial_dev) return CASSIE_EINVAL;
  hipLaunchKernelGGL(cassie_trpo::sampler_step_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, rew_dev, done_dev, n, max_path_length, path_t_dev,
                     path_ret_dev, rew_row_dev, t_row_dev, cut_row_dev, partial_dev);
  return hipGetLastError() == hipSuccess ? CASSIE_OK : CASSIE_EHIP;
}

int CassieTrpoPolicyStep(const double* obs_dev, int n, int obs_dim, int act_dim, const float* W1, const float* b1, const float* W2,
                         const float* b2, const float* W3, const float* b3, const float* log_std, const float* noise_dev,
                         const double* low_dev, const double* high_dev, float* obs32_dev, float* mean_dev, float* act_dev,
                         double* env_actions_dev, void* stream) {
  if (!obs_dev || n <= 0 || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !log_std || !noise_dev || !low_dev || !high_dev || !obs32_dev || !mean_dev ||
      !act_dev || !env_actions_dev)
    return CASSIE_EINVAL;
  const cassie_trpo::Net th{W1, b1, W2, b2, W3, b3};
  hipStream_t s = (hipStream_t)stream;
  using namespace cassie_trpo;
  static const bool vector_path = [] { const char* e = getenv("CASSIE_TRPO_POLICY_VALU"); return e && e[0] == '1'; }();   // the one-lane-per-environment kernel (A/B, tests)
  if (vector_path) {
    const dim3 grid((n + 255) / 256), block(256);
    if (obs_dim == 26 && act_dim == 6) hipLaunchKernelGGL((policy_step_kernel<26, 6>), grid, block, 0, s, obs_dev, n, th, log_std, noise_dev, low_dev, high_dev, obs32_dev, mean_dev, act_dev, env_actions_dev);
    else if (obs_dim == 26 && act_dim == 7) hipLaunchKernelGGL((policy_step_kernel<26, 7>), grid, block, 0, s, obs_dev, n, th, log_std, noise_dev, low_dev, high_dev, obs32_dev, mean_dev, act_dev, env_actions_dev);
    else return CASSIE_EINVAL;
    return hipGetLastError() == hipSuccess ? CASSIE_OK : CASSIE_EHIP;
  }
  const dim3 grid(blocks_for(n)), block(64 * WAVES);
  if (obs_dim == 26 && act_dim == 6) hipLaunchKernelGGL((policy_step_mfma_kernel<26, 6>), grid, block, 0, s, obs_dev, n, th, log_std, noise_dev, low_dev, high_dev, obs32_dev, mean_dev, act_dev, env_actions_dev);
  else if (obs_dim == 26 && act_dim == 7) hipLaunchKernelGGL((policy_step_mfma_kernel<26, 7>), grid, block, 0, s, obs_dev, n, th, log_std, noise_dev, low_dev, high_dev, obs32_dev, mean_dev, act_dev, env_actions_dev);
  else return CASSIE_EINVAL;
  return hipGetLastError() == hipSuccess ? CASSIE_OK : CASSIE_EHIP;
}

}  // extern "C"
