/*
 * C restatement of the TensorBNN weight transition -- TEST INFRASTRUCTURE ONLY
 * (oracle; also the "port" CPU baseline bench.py times on the host cores).
 * PARITY UNPINNED: see oracle/tbnn_oracle.py's header; this file follows the
 * same reference lines and is itself checked against that NumPy restatement
 * (tests/test_oracle.py).
 *
 * float32 arithmetic like the reference (dtype=tf.float32 everywhere), OpenMP
 * over row blocks.  Follows:
 *   forward      network.py:141-171, layer.py:266-279, activationFunctions.py:35-63
 *   likelihoods  likelihood.py:69-96, :143-169, :210-237 + BNN_functions.py:7-34
 *   priors       layer.py:166-197, :346-377 + BNN_functions.py:37-57 (Q1, Q2 kept)
 *   HMC          tfp.mcmc.HamiltonianMonteCarlo (un-vendored TFP 0.12.2; call
 *                sites network.py:394-408): bootstrap value_and_grad EVERY
 *                epoch (Q10), half kick, L x (drift, grad, kick), undo half
 *                kick, Metropolis with non-finite -> reject.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define RB 64
#define MAXL 16

typedef struct {
    int nl;
    int in[MAXL], out[MAXL], act[MAXL], prior[MAXL];
    int lik;          /* 0 gaussian, 1 fixed gaussian, 2 bernoulli */
    float fixed_sd;
} onet;

static int net_P(const onet* n) { int p = 0; for (int l = 0; l < n->nl; ++l) p += n->in[l] * n->out[l] + n->out[l]; return p; }
static int net_H(const onet* n) { return 4 * n->nl + (n->lik == 0 ? 1 : 0); }

static inline float actf(float z, int a) {
    switch (a) { case 1: return z > 0.f ? z : 0.f; case 2: return tanhf(z); case 3: return 1.f / (1.f + expf(-z)); case 4: return expf(z); case 5: return z > 0.f ? z : expm1f(z); default: return z; }
}
static inline float dactf(float a, int act) {
    switch (act) { case 1: return a > 0.f ? 1.f : 0.f; case 2: return 1.f - a * a; case 3: return a * (1.f - a); case 4: return a; case 5: return a > 0.f ? 1.f : a + 1.f; default: return 1.f; }
}
static float lik_sigma(const onet* n, const float* eta) {
    float s = n->lik == 0 ? eta[net_H(n) - 1] * eta[net_H(n) - 1] : n->fixed_sd;   /* likelihood.py:88 / :162 */
    if (s < 1e-8f) s = 1e-8f;                                                       /* BNN_functions.py:23-24 */
    if (s > 1e8f) s = 1e8f;
    return s;
}

/* prior log-density and its gradient (added into grad) */
static double prior_logp_grad(const onet* n, const float* th, const float* eta, float* grad) {
    double lp = 0.0;
    int off = 0;
    for (int l = 0; l < n->nl; ++l) {
        for (int part = 0; part < 2; ++part) {
            const int cnt = part ? n->out[l] : n->in[l] * n->out[l];
            const float loc = eta[4 * l + 2 * part], g = eta[4 * l + 2 * part + 1];
            const float sc = g * g;                                    /* layer.py:178,180 (Q3) */
            if (n->prior[l] == 0) {
                const float lb = logf(3.14159265358979323846f * sc);  /* BNN_functions.py:52 */
                for (int e = 0; e < cnt; ++e) {
                    const float z = (th[off + e] - loc) / sc;
                    lp += (double)(logf(1.f + z * z) - lb);            /* :51-55 (Q1) */
                    if (grad) grad[off + e] += 2.f * z / (sc * (1.f + z * z));
                }
            } else {
                float s = sc < 1e-8f ? 1e-8f : (sc > 1e8f ? 1e8f : sc);
                double z2 = 0.0;
                for (int e = 0; e < cnt; ++e) {
                    const float d = (th[off + e] - loc) / s;
                    z2 += (double)(d * d);
                    if (grad) grad[off + e] += -(th[off + e] - loc) / (s * s);
                }
                lp += -0.5 * (2.0 * (double)logf(s) + z2 + 1.8378770664093453);   /* Q2: k = 1 */
            }
            off += cnt;
        }
    }
    return lp;
}

/* value and gradient of the target (network.py:370-392 + autodiff) */
double oracle_logp_grad(const onet* n, const float* th, const float* eta, const float* X, const float* Y,
                        long nrows, float* grad, double* stat_out)
{
    const int P = net_P(n), nl = n->nl, d_in = n->in[0], d_out = n->out[nl - 1];
    int sum = d_in, maxw = d_in, aoff[MAXL + 1], woff[MAXL];
    aoff[0] = 0;
    for (int l = 0, o = 0; l < nl; ++l) { woff[l] = o; o += n->in[l] * n->out[l] + n->out[l]; aoff[l + 1] = sum; sum += n->out[l]; if (n->out[l] > maxw) maxw = n->out[l]; }
    const float sigma = lik_sigma(n, eta), inv_var = 1.f / (sigma * sigma);
    const long nblk = (nrows + RB - 1) / RB;
    int nth = 1;
#ifdef _OPENMP
    nth = omp_get_max_threads();
#endif
    /* per-thread gradient accumulators ACROSS row blocks in double (the 64-row dot products inside a block are fp32
     * like the reference's arithmetic): at n = 1e6 a sequential fp32 sum over 15k blocks would put the oracle's own
     * rounding (~1e-4 of the tensor norm) on top of what the parity tests measure */
    double* gacc = (double*)calloc((size_t)nth * P, sizeof(double));
    double* sacc = (double*)calloc((size_t)nth, sizeof(double));
#pragma omp parallel
    {
        int t = 0;
#ifdef _OPENMP
        t = omp_get_thread_num();
#endif
        float* A = (float*)malloc((size_t)sum * RB * sizeof(float));      /* a_0..a_L, [unit][row] */
        float* DZ = (float*)malloc((size_t)maxw * RB * sizeof(float));
        float* DA = (float*)malloc((size_t)maxw * RB * sizeof(float));
        double* g = gacc + (size_t)t * P;
        double st = 0.0;
#pragma omp for schedule(static)
        for (long b = 0; b < nblk; ++b) {
            const long r0 = b * RB;
            const int rows = (int)((nrows - r0) < RB ? (nrows - r0) : RB);
            for (int k = 0; k < d_in; ++k)
                for (int r = 0; r < RB; ++r) A[k * RB + r] = r < rows ? X[(r0 + r) * d_in + k] : 0.f;
            for (int l = 0; l < nl; ++l) {                                   /* forward */
                const int in = n->in[l], out = n->out[l];
                const float* W = th + woff[l]; const float* bb = W + in * out;
                const float* ain = A + (size_t)aoff[l] * RB; float* aout = A + (size_t)aoff[l + 1] * RB;
                for (int i = 0; i < out; ++i) {
                    float* z = aout + i * RB;
                    for (int r = 0; r < RB; ++r) z[r] = bb[i];
                    for (int k = 0; k < in; ++k) {
                        const float w = W[i * in + k]; const float* a = ain + k * RB;
#pragma omp simd
                        for (int r = 0; r < RB; ++r) z[r] += w * a[r];
                    }
                    for (int r = 0; r < RB; ++r) z[r] = actf(z[r], n->act[l]);
                }
            }
            const float* f = A + (size_t)aoff[nl] * RB;                     /* likelihood */
            for (int i = 0; i < d_out; ++i)
                for (int r = 0; r < RB; ++r) {
                    float da = 0.f;
                    if (r < rows) {
                        const float y = Y[(r0 + r) * d_out + i], fi = f[i * RB + r];
                        if (n->lik == 2) {
                            float p = fi < 1e-8f ? 1e-8f : (fi > 1.f - 1e-7f ? 1.f - 1e-7f : fi);   /* likelihood.py:226-231 */
                            const int inside = fi > 1e-8f && fi < 1.f - 1e-7f;
                            st += (double)((y == 0.f ? 0.f : y * logf(p)) + ((1.f - y) == 0.f ? 0.f : (1.f - y) * log1pf(-p)));
                            da = inside ? (y / p - (1.f - y) / (1.f - p)) : 0.f;
                        } else {
                            const float res = y - fi;
                            st += (double)res * (double)res;
                            da = res * inv_var;
                        }
                    }
                    DA[i * RB + r] = da;
                }
            for (int l = nl - 1; l >= 0; --l) {                              /* backward */
                const int in = n->in[l], out = n->out[l];
                const float* W = th + woff[l];
                const float* ain = A + (size_t)aoff[l] * RB; const float* aout = A + (size_t)aoff[l + 1] * RB;
                double* gW = g + woff[l]; double* gb = gW + in * out;
                for (int i = 0; i < out; ++i)
                    for (int r = 0; r < RB; ++r) DZ[i * RB + r] = DA[i * RB + r] * dactf(aout[i * RB + r], n->act[l]);
                for (int i = 0; i < out; ++i) {
                    const float* dz = DZ + i * RB;
                    float sb = 0.f;
#pragma omp simd reduction(+ : sb)
                    for (int r = 0; r < RB; ++r) sb += dz[r];
                    gb[i] += sb;
                    for (int k = 0; k < in; ++k) {
                        const float* a = ain + k * RB;
                        float s = 0.f;
#pragma omp simd reduction(+ : s)
                        for (int r = 0; r < RB; ++r) s += dz[r] * a[r];
                        gW[i * in + k] += s;
                    }
                }
                if (l > 0) {
                    for (int k = 0; k < in; ++k) {
                        float* d = DA + k * RB;
                        for (int r = 0; r < RB; ++r) d[r] = 0.f;
                    }
                    for (int i = 0; i < out; ++i) {
                        const float* dz = DZ + i * RB;
                        for (int k = 0; k < in; ++k) {
                            const float w = W[i * in + k]; float* d = DA + k * RB;
#pragma omp simd
                            for (int r = 0; r < RB; ++r) d[r] += w * dz[r];
                        }
                    }
                }
            }
        }
        sacc[t] = st;
        free(A); free(DZ); free(DA);
    }
    double stat = 0.0;
    for (int t = 0; t < nth; ++t) stat += sacc[t];
    if (grad)
        for (int j = 0; j < P; ++j) {
            double sj = 0.0;
            for (int t = 0; t < nth; ++t) sj += gacc[(size_t)t * P + j];
            grad[j] = (float)sj;
        }
    free(gacc); free(sacc);
    double lp = prior_logp_grad(n, th, eta, grad);
    if (n->lik == 2) lp += stat;
    else {
        const double nel = (double)nrows * d_out, s = (double)sigma;
        lp += -0.5 * (2.0 * nel * log(s) + stat / (s * s) + nel * 1.8378770664093453);
    }
    if (stat_out) *stat_out = stat;
    return lp;
}

/* the proposal of one transition: q_out = q_L, *lar_out = log accept ratio (non-finite -> -inf) */
void oracle_hmc_propose(const onet* n, const float* theta, const float* eta, const float* X, const float* Y, long nrows,
                        float eps, int L, const float* p0, float* q_out, double* lar_out, double* logp_old, double* logp_new)
{
    const int P = net_P(n);
    float* q = q_out; float* p = (float*)malloc(P * sizeof(float));
    float* g = (float*)malloc(P * sizeof(float));
    memcpy(q, theta, P * sizeof(float));
    const double lp0 = oracle_logp_grad(n, q, eta, X, Y, nrows, g, NULL);      /* bootstrap_results (Q10) */
    double k0 = 0.0, k1 = 0.0, lp = lp0;
    for (int j = 0; j < P; ++j) { k0 += (double)p0[j] * p0[j]; p[j] = p0[j] + 0.5f * eps * g[j]; }
    for (int t = 0; t < L; ++t) {
        for (int j = 0; j < P; ++j) q[j] = q[j] + eps * p[j];
        lp = oracle_logp_grad(n, q, eta, X, Y, nrows, g, NULL);
        for (int j = 0; j < P; ++j) p[j] = p[j] + eps * g[j];
    }
    for (int j = 0; j < P; ++j) { p[j] = p[j] - 0.5f * eps * g[j]; k1 += (double)p[j] * p[j]; }
    double lar = lp - lp0 + 0.5 * k0 - 0.5 * k1;
    if (!isfinite(lar)) lar = -INFINITY;
    if (lar_out) *lar_out = lar;
    if (logp_old) *logp_old = lp0;
    if (logp_new) *logp_new = lp;
    free(p); free(g);
}

/* one transition; returns accepted flag.  theta is updated in place. */
int oracle_hmc_step(const onet* n, float* theta, const float* eta, const float* X, const float* Y, long nrows,
                    float eps, int L, const float* p0, float log_u, double* lar_out, double* logp_old, double* logp_new)
{
    const int P = net_P(n);
    float* q = (float*)malloc(P * sizeof(float));
    double lar = 0.0;
    oracle_hmc_propose(n, theta, eta, X, Y, nrows, eps, L, p0, q, &lar, logp_old, logp_new);
    const int acc = (double)log_u < lar;
    if (acc) memcpy(theta, q, P * sizeof(float));
    if (lar_out) *lar_out = lar;
    free(q);
    return acc;
}

/* thread count of the following calls (the cpu_baseline leg times 1 thread and the best of several counts) */
void oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
