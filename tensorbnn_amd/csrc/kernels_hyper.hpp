// Hyper-parameter HMC transition: the HMC part of InnerStepHyper
// (network.py:414-456) as ONE single-workgroup kernel.
//
// Target (closure network.py:416-440):
//   sum over dense layers of calculateHyperProbs (layer.py:199-242 Cauchy,
//   :379-422 Gaussian): hyper-priors evaluated at the squared value (Q4) +
//   the weight-prior terms as a function of (loc, g);
//   + (mainProbsInHypers, likelihood.py:67) the Gaussian data log-likelihood
//   as a function of sd = eta_last^2.  The prediction does not depend on eta,
//   so the data term uses the sufficient statistic S = sum (y-f)^2 cached by
//   the weight transition: same value, no forward pass.
// Gradient: hand-coded (TF autodiff in the reference).  O(P) per leapfrog
// step: every thread strides the weights, 64-lane shuffle reduction, LDS
// double atomics across the 16 waves.
#pragma once
#include "common.hpp"

#define HYP_THREADS 1024
#define HYP_MAXH (4 * TBNN_MAX_LAYERS + 1)
#define HYP_WS_GRAD 0
enum { HYP_EVAL = 0, HYP_STEP = 1 };

static inline size_t hyper_ws_bytes(const NetDev& nd) { return (size_t)4 * nd.H * sizeof(float); }

__device__ __forceinline__ double mvn1_logp(double x, double loc, double sc) {
    const double z = (x - loc) / sc;
    return -0.5 * z * z - log(sc) - 0.9189385332046727;   // 1/2 log 2pi
}

// value + gradient of the hyper target at e[] (LDS).  All threads call it.
__device__ void hyper_eval(const NetDev& nd, const float* e, const float* __restrict__ q, double S, long n,
                           double* acc /*LDS [2*nl*3]*/, double* val /*LDS*/, float* grad /*LDS [H]*/) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int ng = 2 * nd.nl;
    for (int i = tid; i < ng * 3; i += blockDim.x) acc[i] = 0.0;
    __syncthreads();
    for (int grp = 0; grp < ng; ++grp) {
        const int l = grp >> 1, part = grp & 1;
        const int off = part ? nd.offB[l] : nd.offW[l];
        const int cnt = part ? nd.out[l] : nd.out[l] * nd.in[l];
        const float loc = e[4 * l + 2 * part];
        const float gg = e[4 * l + 2 * part + 1];
        const float scale = gg * gg;                               // layer.py:209-212 (Q3)
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
        if (nd.prior[l] == TBNN_PRIOR_CAUCHY) {
            for (int i = tid; i < cnt; i += blockDim.x) {
                const float z = (q[off + i] - loc) / scale;        // BNN_functions.py:51
                const float w = 2.f * z / (1.f + z * z);
                a0 += (double)logf(1.f + z * z);
                a1 += (double)w;
                a2 += (double)(w * z);
            }
        } else {
            for (int i = tid; i < cnt; i += blockDim.x) {
                const float d = q[off + i] - loc;
                a1 += (double)d;
                a2 += (double)d * (double)d;
            }
        }
        a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
        if (lane == 0) {
            atomicAdd(&acc[grp * 3 + 0], a0);
            atomicAdd(&acc[grp * 3 + 1], a1);
            atomicAdd(&acc[grp * 3 + 2], a2);
        }
    }
    __syncthreads();
    if (tid == 0) {
        double v = 0.0;
        for (int grp = 0; grp < ng; ++grp) {
            const int l = grp >> 1, part = grp & 1;
            const double cnt = part ? nd.out[l] : (double)nd.out[l] * nd.in[l];
            const double loc = e[4 * l + 2 * part];
            const double gg = e[4 * l + 2 * part + 1];
            const double scale = (double)(float)((float)gg * (float)gg);
            const double s0 = acc[grp * 3], s1 = acc[grp * 3 + 1], s2 = acc[grp * 3 + 2];
            double d_loc, d_scale;
            if (nd.prior[l] == TBNN_PRIOR_CAUCHY) {
                // hyper-priors layer.py:136-153, evaluated :221-228
                v += mvn1_logp(loc, 0.0, 0.2) + mvn1_logp(scale, 0.70710678118654757, 0.5);
                v += s0 - cnt * log(3.14159265358979323846 * scale);   // sum cauchyLogProb (Q1)
                d_loc = -s1 / scale - loc / 0.04;
                d_scale = -s2 / scale - cnt / scale - (scale - 0.70710678118654757) / 0.25;
            } else {
                // hyper-priors layer.py:316-334, evaluated :401-408
                v += mvn1_logp(loc, 0.0, 0.1) + mvn1_logp(scale, 1.0, 0.1);
                const double s = fmin(fmax(scale, 1e-8), 1e8);
                const bool clamped = !(scale > 1e-8 && scale < 1e8);
                v += -0.5 * (2.0 * log(s) + s2 / (s * s) + 1.8378770664093453);   // Q2: k = 1
                d_loc = s1 / (s * s) - loc / 0.01;
                d_scale = (clamped ? 0.0 : (-1.0 / s + s2 / (s * s * s))) - (scale - 1.0) / 0.01;
            }
            grad[4 * l + 2 * part] = (float)d_loc;
            grad[4 * l + 2 * part + 1] = (float)(d_scale * 2.0 * gg);
        }
        if (nd.lik == TBNN_LIK_GAUSSIAN) {                         // network.py:435-438
            const double el = e[nd.H - 1];
            const double sr = (double)(float)((float)el * (float)el);
            const double s = fmin(fmax(sr, 1e-8), 1e8);
            const bool clamped = !(sr > 1e-8 && sr < 1e8);
            const double nel = (double)n * nd.d_out;
            v += -0.5 * (2.0 * nel * log(s) + S / (s * s) + nel * 1.8378770664093453);
            const double ds = clamped ? 0.0 : (-nel / s + S / (s * s * s));
            grad[nd.H - 1] = (float)(ds * 2.0 * el);
        }
        *val = v;
    }
    __syncthreads();
}

__global__ __launch_bounds__(HYP_THREADS) void k_hyper(
    NetDev nd, int mode, float eps, int L, float* __restrict__ eta, const float* __restrict__ q, long n,
    const float* __restrict__ p0_inj, const float* __restrict__ logu_inj, uint32_t epoch, uint32_t key0, uint32_t key1,
    const Scal* __restrict__ sc, float* __restrict__ ws, Scal* __restrict__ out)
{
    __shared__ double acc[2 * TBNN_MAX_LAYERS * 3];
    __shared__ double val;
    __shared__ float e[HYP_MAXH], e0[HYP_MAXH], p[HYP_MAXH], g[HYP_MAXH];
    __shared__ double sh_k0, sh_lp0;
    const int tid = threadIdx.x, H = nd.H;
    const double S = sc->stat_cur;
    if (tid < H) { e[tid] = eta[tid]; e0[tid] = eta[tid]; }
    __syncthreads();
    hyper_eval(nd, e, q, S, n, acc, &val, g);
    if (mode == HYP_EVAL) {
        if (tid < H) ws[HYP_WS_GRAD * H + tid] = g[tid];
        if (tid == 0) { Scal o = *sc; o.logp_new = val; *out = o; }
        return;
    }
    if (tid == 0) {
        double k = 0.0;
        for (int j = 0; j < H; ++j) {
            const float v = p0_inj ? p0_inj[j] : philox_normal((uint32_t)j, epoch, PURPOSE_HYPER_MOMENTUM, key0, key1);
            p[j] = v; k += (double)v * (double)v;
        }
        sh_k0 = 0.5 * k; sh_lp0 = val;
    }
    __syncthreads();
    if (tid < H) p[tid] = p[tid] + 0.5f * eps * g[tid];           // half kick
    __syncthreads();
    for (int t = 1; t <= L; ++t) {
        if (tid < H) e[tid] = e[tid] + eps * p[tid];               // drift
        __syncthreads();
        hyper_eval(nd, e, q, S, n, acc, &val, g);
        if (tid < H) p[tid] = p[tid] + eps * g[tid];               // full kick
        __syncthreads();
    }
    if (tid < H) p[tid] = p[tid] - 0.5f * eps * g[tid];           // undo half kick
    __syncthreads();
    if (tid == 0) {
        double k1 = 0.0, d2 = 0.0;
        for (int j = 0; j < H; ++j) {
            k1 += (double)p[j] * (double)p[j];
            const double d = (double)e[j] - (double)e0[j];
            d2 += d * d;
        }
        k1 *= 0.5;
        double lar = val - sh_lp0 + sh_k0 - k1;
        if (!isfinite(lar)) lar = -INFINITY;
        const double lu = logu_inj ? (double)logu_inj[0] : (double)philox_logu(epoch, PURPOSE_HYPER_LOGU, key0, key1);
        const int a = lu < lar ? 1 : 0;
        Scal o = *sc;
        o.logp_cur = sh_lp0; o.logp_new = val; o.k0 = sh_k0; o.k1 = k1; o.lar = lar; o.logu = lu;
        o.d2 = d2; o.sjd = a ? d2 : 0.0; o.accepted = a;
        *out = o;
        acc[0] = (double)a;
    }
    __syncthreads();
    if (acc[0] != 0.0 && tid < H) eta[tid] = e[tid];
}
