/* cassie_trpo.h -- fused policy kernels of the TRPO outer loop: the counterpart of what rllab's TRPO
 * (rllab/envs/trpo_cassie.py:21-42: GaussianMLPPolicy 32x32 tanh, conjugate-gradient optimiser with Fisher-vector products by
 * double backprop through the mean KL) evaluates on the sampled batch, for the batch sizes the vectorised environment produces
 * (65 536 envs x 8 steps = 524 288 samples per rank and iteration).
 *
 * The policy is mean(obs) = W3 tanh(W2 tanh(W1 obs + b1) + b2) + b3 with a state-independent log-std; float32, row-major weights
 * as torch.nn.Linear stores them (W1 [32][obs_dim], W2 [32][32], W3 [act_dim][32]).  Plain device pointers and sizes; every call
 * enqueues on `stream` (a hipStream_t, 0 = default) and returns 0 or a negative CASSIE_E* code (cassie_vec.h).
 *
 * Both entry points write PARTIAL sums, one row per wavefront of the launch: partial [n_rows][CassieTrpoParamCount] float32 with the
 * row layout [gW1 | gb1 | gW2 | gb2 | gW3 | gb3]; the caller adds the rows up (one reduction) -- and all-reduces the result over
 * ranks, exactly where the unfused version did.  CassieTrpoPartialRows() says how many rows a launch over n samples writes.
 */
#ifndef CASSIE_TRPO_H_
#define CASSIE_TRPO_H_

#ifdef __cplusplus
extern "C" {
#endif

/* parameters of the mean network: 32 * obs_dim + 32 + 32 * 32 + 32 + act_dim * 32 + act_dim; 0 for an unsupported shape
 * (supported: obs_dim 26 or 17, act_dim 6 or 7) */
int CassieTrpoParamCount(int obs_dim, int act_dim);
int CassieTrpoPartialRows(int n_samples);

/* Fisher-vector product of the mean network, (1/n) J' S J v: J = d mean / d theta at the CURRENT weights (forward mode per sample
 * along the direction dW1..db3, activations recomputed), S = diag(prec[act_dim]) the precision of the old Gaussian, then reverse
 * mode back to the parameters.  scale is the 1/n (n of the whole job) the caller wants folded in. */
int CassieTrpoFvp(const float* obs_dev, int n, int obs_dim, int act_dim, const float* W1, const float* b1, const float* W2, const float* b2,
                  const float* W3, const float* b3, const float* dW1, const float* db1, const float* dW2, const float* db2, const float* dW3,
                  const float* db3, const float* prec, float scale, float* partial_dev, void* stream);

/* Vector-Jacobian product J' w for per-sample cotangents w [n][act_dim] on the mean (the policy gradient of the surrogate loss:
 * w = d loss / d mean, evaluated by the caller). */
int CassieTrpoVjp(const float* obs_dev, int n, int obs_dim, int act_dim, const float* W1, const float* b1, const float* W2, const float* b2,
                  const float* W3, const float* b3, const float* w_dev, float* partial_dev, void* stream);

/* Line search of the TRPO step on the sampled batch (what rllab's f_loss / f_constraint evaluate per backtrack): for the mean network
 * AT THE GIVEN WEIGHTS and log-std log_std_new [act_dim], against the old Gaussian (old_mean [n][act_dim], log_std_old [act_dim]):
 *   partial[row][0] = sum_s -exp(ll_new(act_s) - ll_old(act_s)) adv_s,   partial[row][1] = sum_s KL(old_s || new_s)
 * (GaussianMLPPolicy.log_likelihood / .kl of cassierl_amd/trpo.py), float64, one row per wavefront (CassieTrpoPartialRows); the
 * caller adds the rows and divides by the job's n. */
int CassieTrpoSurrogate(const float* obs_dev, int n, int obs_dim, int act_dim, const float* W1, const float* b1, const float* W2, const float* b2,
                        const float* W3, const float* b3, const float* log_std_new, const float* log_std_old, const float* act_dev,
                        const float* adv_dev, const float* old_mean_dev, double* partial_dev, void* stream);

/* The vector work of one conjugate-gradient iteration (rllab/misc/krylov.py: cg) between two Fisher-vector products, one launch: with the
 * parameter vector laid out as [.. | log_std block at ls_off, n_ls entries | ..] and Ap_mean [n - n_ls] = the mean network's part of F p
 * (rows of CassieTrpoFvp added up, summed over ranks, in the order of the remaining entries):
 *   Ap = Ap_mean (+ hls o p on the log_std block) + reg p;  alpha = rr / p.Ap;  x += alpha p;  r -= alpha Ap;  rr' = r.r;
 *   p = r + (rr' / rr) p;  scal = {rr', running}: once rr' < tol the step length stays 0 (cg's early exit without a host read-back). */
int CassieTrpoCgUpdate(int n, int ls_off, int n_ls, const float* Ap_mean_dev, const float* hls_dev, float reg, float tol, float* x_dev, float* r_dev, float* p_dev,
                       float* scal_dev, void* stream);

/* One policy step of the sampler for n environments in ONE launch (the counterpart of GaussianMLPPolicy.get_actions + rllab's
 * normalize() wrapper, rllab/envs/trpo_cassie.py:13,21-27): obs float64 [n][obs_dim] as the environment wrote it ->
 *   obs32 [n][obs_dim] (the policy's float32 view, kept for the update), mean [n][act_dim] = mean network, act [n][act_dim] = mean +
 *   noise * exp(log_std) (noise: standard normal numbers of the caller, row stride act_dim), and
 *   env_actions float64 [n][act_dim] = clip(low + (act + 1) / 2 * (high - low), low, high): what CassieVecStep consumes. */
int CassieTrpoPolicyStep(const double* obs_dev, int n, int obs_dim, int act_dim, const float* W1, const float* b1, const float* W2,
                         const float* b2, const float* W3, const float* b3, const float* log_std, const float* noise_dev,
                         const double* low_dev, const double* high_dev, float* obs32_dev, float* mean_dev, float* act_dev,
                         double* env_actions_dev, void* stream);

/* Sampler bookkeeping of one Env.step for n environments in one launch (the per-path clocks and returns rllab's sampler keeps on the
 * host; cassierl_amd/trpo.py: collect): with rew / done as CassieVecStep wrote them,
 *   rew_row[i] = rew[i], t_row[i] = path_t[i]  (this step's rows of the batch),  path_ret[i] += rew[i], path_t[i] += 1,
 *   cut = done[i] || path_t[i] >= max_path_length  (rllab truncates paths there),  cut_row[i] = cut (one byte, 0 / 1),
 *   a cut path adds (1, path_ret[i]) to its workgroup's row of partial [CassieTrpoSamplerRows(n)][2] and restarts: path_ret[i] = path_t[i] = 0.
 * The rows are summed in a fixed order inside the kernel; the caller adds them up. */
int CassieTrpoSamplerRows(int n_envs);
int CassieTrpoSamplerStep(const double* rew_dev, const unsigned char* done_dev, int n, long long max_path_length, long long* path_t_dev, double* path_ret_dev,
                          double* rew_row_dev, long long* t_row_dev, unsigned char* cut_row_dev, double* partial_dev, void* stream);

/* ---- the baseline side: rllab's LinearFeatureBaseline (trpo_cassie.py:30) on the [T][n] batch of the vectorised environment.
 * Features of a sample: [o, o^2, a, a^2, a^3, 1] with o = clip(obs, -10, 10) [obs_dim] and a = path clock / 100, evaluated in float32 as the
 * torch expressions of cassierl_amd/trpo.py evaluate them, used in float64; CassieTrpoBaselineFeatures(obs_dim) = 2 obs_dim + 4 (0: unsupported). */
int CassieTrpoBaselineFeatures(int obs_dim);

/* values[s] = features(obs[s], t[s]) . coeffs  (obs float32 [m][obs_dim], t int64 [m], coeffs float64 [features]) */
int CassieTrpoBaselinePredict(const float* obs_dev, const long long* t_dev, int m, int obs_dim, const double* coeffs_dev, double* out_dev, void* stream);

/* One lane per environment, backwards over the T steps of its column of the batch (sample s = step * n + env):
 *   value = features . coeffs (0 with coeffs = NULL: the first iteration),
 *   returns[s] = rew[s] + gamma * returns[next step] * !cut[s]  (bootstrapped behind the last step with last_value [n], NULL = 0),
 *   adv[s] = returns[s] - value  (gae_lambda = 1),
 * and partial[(n + 255) / 256][2] = per workgroup (sum adv, sum adv^2), fixed order: what the advantage normalisation needs. */
int CassieTrpoReturnsAdvantages(const float* obs_dev, const long long* t_dev, const double* rew_dev, const unsigned char* cut_dev, int T, int n, int obs_dim,
                                const double* coeffs_dev, const double* last_value_dev, double gamma, double* returns_dev, double* adv_dev, double* partial_dev,
                                void* stream);

/* Normal equations of the baseline's ridge regression: Z'Z for Z = [features | y] (m samples, 2 obs_dim + 5 columns padded to a multiple
 * of 16) on the FP64 matrix cores.  partial [CassieTrpoGramRows()][CassieTrpoGramRowSize(obs_dim)]: per wavefront the UPPER 16 x 16 blocks
 * (r <= c, r-major) of the Gram matrix, each row-major; the caller adds the rows up: X'X = Z'Z[:features, :features], X'y = Z'Z[:features, features]. */
/* (A + reg I) x = b, A [F][F] symmetric positive semi-definite (F = CassieTrpoBaselineFeatures(obs_dim): 56 or 38), by Cholesky on the device; a failed factorisation or a non-finite
 * solution retries with ten times the regulariser, five times in all (LinearFeatureBaseline.fit's rule, without a host read-back). */
int CassieTrpoRidgeSolve(const double* A_dev, const double* b_dev, int F, double reg, double* x_dev, void* stream);
int CassieTrpoGramRows(void);
int CassieTrpoGramRowSize(int obs_dim);
int CassieTrpoBaselineGram(const float* obs_dev, const long long* t_dev, const double* y_dev, int m, int obs_dim, double* partial_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif
