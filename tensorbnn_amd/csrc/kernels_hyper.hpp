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
// Gradient: hand-coded (TF autodiff in the reference).  O(P) per leapfrog step.
//
// Work split (L_h + 1 evaluations per launch, so the per-evaluation latency is what counts): wave w of the 16 owns
// a contiguous chunk of theta and keeps its elements in registers for the whole launch (theta does not move during
// a hyper transition); per evaluation it reduces its chunk once per prior group it overlaps (usually one or two of the
// 2*nl groups: W_l, b_l) and writes the three sums to its own LDS row -- no atomics, fixed summation order.  After one
// barrier, thread t < 2*nl finishes group t (hyper-prior + weight-prior terms, d/d loc, d/d g) and thread 2*nl the
// Gaussian data term, each in parallel and each applying the leapfrog kick / drift to the hyper entries it owns:
// two barriers per leapfrog step.  (The first version walked the groups one after the other with 8 x 3 double
// wave reductions + LDS atomics and finished every group on thread 0: 19 us per evaluation at configs[1], 1.9 ms per
// epoch with L_h = 100 -- 40 % of an epoch of network.train with adjustHypers=True.)
#pragma once
#include "common.hpp"

#define HYP_THREADS 1024
#define HYP_MAXH (4 * TBNN_MAX_LAYERS + 1)
#define HYP_WS_GRAD 0
enum { HYP_EVAL = 0, HYP_STEP = 1 };

static inline size_t hyper_ws_bytes(const NetDev& nd) { return (size_t)4 * nd.H * sizeof(float); }

// sc is a literal at every call site: 1/sc and log(sc) fold to constants once this is inlined
__device__ __forceinline__ double mvn1_logp(double x, double loc, double sc) {
    const double z = (x - loc) * (1.0 / sc);
    return -0.5 * z * z - log(sc) - 0.9189385332046727;   // 1/2 log 2pi
}

#define HYP_WAVES (HYP_THREADS / 64)
// diagnostic build (-DTBNN_TILE_STAMPS): shader-clock stamps of leapfrog step 1 (thread 0) in g_tile_stamps[40..]
#ifdef TBNN_TILE_STAMPS
#define HSTAMP(k) do { if (tid == 0 && t == 1) g_tile_stamps[40 + (k)] = clock64(); } while (0)
#else
#define HSTAMP(k) do { } while (0)
#endif
#define HYP_MAXG (2 * TBNN_MAX_LAYERS)
#define HYP_REG 20         // theta elements per lane held in registers (a wave's share <= 1280); larger shares re-read theta

struct HypGroup { int off, cnt, l, part; };
__device__ __forceinline__ HypGroup hyp_group(const NetDev& nd, int grp) {
    HypGroup G; G.l = grp >> 1; G.part = grp & 1;
    G.off = G.part ? nd.offB[G.l] : nd.offW[G.l];
    G.cnt = G.part ? nd.out[G.l] : nd.out[G.l] * nd.in[G.l];
    return G;
}

// 64-lane sum of a double with DPP row shifts / row broadcasts (no LDS round trips: six ds_bpermute pairs in a row
// were the longest chain of an evaluation).  The total lands in lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
    return v + __hiloint2double(hi2, lo2);
}
__device__ __forceinline__ double wave_sum63(double v) {
    v = dpp_add<0x111, 0xf>(v);      // row_shr:1
    v = dpp_add<0x112, 0xf>(v);      // row_shr:2
    v = dpp_add<0x114, 0xf>(v);      // row_shr:4
    v = dpp_add<0x118, 0xf>(v);      // row_shr:8   -> lane 15 of every row: the row's sum
    v = dpp_add<0x142, 0xa>(v);      // row_bcast:15 into rows 1, 3
    v = dpp_add<0x143, 0xc>(v);      // row_bcast:31 into rows 2, 3 -> lane 63: the wave's sum
    return v;
}

// per-wave work, set up once per launch: ONE prior group per wave and a contiguous share [beg, end) of it, so an
// evaluation is a single pass + one reduction per wave (thread 0 deals the 16 waves out: one per group, the rest to
// whichever group has the largest share per wave); a network with more groups than waves folds the tail groups
// into the last wave's passes
struct HypPlan { int grp[HYP_WAVES], beg[HYP_WAVES], end[HYP_WAVES], extra0;       // extra0: first group of the folded tail (ng: none)
                 int w0[HYP_MAXG], w1[HYP_MAXG]; };                                    // waves [w0, w1) hold group g's partial sums
__device__ void hyper_plan(const NetDev& nd, HypPlan* P) {
    const int ng = 2 * nd.nl;
    int nw[HYP_MAXG];
    const int direct = ng <= HYP_WAVES ? ng : HYP_WAVES - 1;           // groups with waves of their own
    for (int g = 0; g < ng; ++g) nw[g] = g < direct ? 1 : 0;
    int left = ng <= HYP_WAVES ? HYP_WAVES - ng : 0;
    while (left > 0) {
        int best = 0; long bl = -1;
        for (int g = 0; g < direct; ++g) { const long ld = ((long)hyp_group(nd, g).cnt + nw[g] - 1) / nw[g]; if (ld > bl) { bl = ld; best = g; } }
        ++nw[best]; --left;
    }
    int w = 0;
    for (int g = 0; g < direct; ++g) {
        const HypGroup G = hyp_group(nd, g);
        P->w0[g] = w; P->w1[g] = w + nw[g];
        for (int k = 0; k < nw[g]; ++k, ++w) {
            P->grp[w] = g;
            P->beg[w] = G.off + (int)(((long)G.cnt * k) / nw[g]);
            P->end[w] = G.off + (int)(((long)G.cnt * (k + 1)) / nw[g]);
        }
    }
    P->extra0 = ng;
    for (int g = direct; g < ng; ++g) { P->w0[g] = HYP_WAVES - 1; P->w1[g] = HYP_WAVES; }
    if (direct < ng) { P->grp[w] = -1; P->beg[w] = P->end[w] = 0; P->extra0 = direct; ++w; }
    for (; w < HYP_WAVES; ++w) { P->grp[w] = -1; P->beg[w] = P->end[w] = 0; }
}

// the three sums of one prior group over this thread's elements; Cauchy: (sum log(1+z^2), sum 2z/(1+z^2), sum 2z^2/(1+z^2)),
// Gaussian: (0, sum d, sum d^2).  One workgroup runs the whole transition: hardware reciprocal / log2 (1 ulp)
// instead of IEEE division and the library logf.
template <class T>
__device__ __forceinline__ void hyp_term(bool cauchy, float qi, float loc, float inv_scale, T& a0, T& a1, T& a2) {
    if (cauchy) {
        const float z = (qi - loc) * inv_scale;                // BNN_functions.py:51
        const float u = 1.f + z * z;
        const float w = 2.f * z * __builtin_amdgcn_rcpf(u);
        a0 += (T)__logf(u);
        a1 += (T)w;
        a2 += (T)(w * z);
    } else {
        const float d = qi - loc;
        a1 += (T)d;
        a2 += (T)d * (T)d;
    }
}

struct HypWave { int grp, beg, end; bool held; float qv[HYP_REG]; };

// one pass over [lo, hi) of group grp from memory -> part[wave][grp*3 + k]
__device__ __forceinline__ void hyper_pass_mem(const NetDev& nd, int grp, int lo, int hi, const float* e, const float* __restrict__ q,
                                               double (*part)[HYP_MAXG * 3]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const HypGroup G = hyp_group(nd, grp);
    const float loc = e[4 * G.l + 2 * G.part];
    const float gg = e[4 * G.l + 2 * G.part + 1];
    const float inv_scale = 1.f / (gg * gg);                       // layer.py:209-212 (Q3)
    const bool cauchy = nd.prior[G.l] == TBNN_PRIOR_CAUCHY;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int i = lo + lane; i < hi; i += 64) hyp_term<double>(cauchy, q[i], loc, inv_scale, a0, a1, a2);
    a0 = wave_sum63(a0); a1 = wave_sum63(a1); a2 = wave_sum63(a2);
    if (lane == 63) { part[wave][grp * 3 + 0] = a0; part[wave][grp * 3 + 1] = a1; part[wave][grp * 3 + 2] = a2; }
}

// this wave's partial sums -> part[wave][grp*3 + k] (lane 63 writes)
__device__ __forceinline__ void hyper_partials(const NetDev& nd, const HypWave& W, int extra0, const float* e, const float* __restrict__ q,
                                               double (*part)[HYP_MAXG * 3]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (W.grp >= 0) {
        if (W.held) {
            const HypGroup G = hyp_group(nd, W.grp);
            const float loc = e[4 * G.l + 2 * G.part];
            const float gg = e[4 * G.l + 2 * G.part + 1];
            const float inv_scale = 1.f / (gg * gg);
            const bool cauchy = nd.prior[G.l] == TBNN_PRIOR_CAUCHY;
            // <= HYP_REG terms per lane: fp32 partial sums of the bounded terms, widened once
            float f0 = 0.f, f1 = 0.f; double a1 = 0.0, a2 = 0.0;
#pragma unroll
            for (int k = 0; k < HYP_REG; ++k)
                if (W.beg + lane + 64 * k < W.end) {
                    if (cauchy) { float t2 = 0.f; hyp_term<float>(true, W.qv[k], loc, inv_scale, f0, f1, t2); a2 += (double)t2; }
                    else { double z0 = 0.0; hyp_term<double>(false, W.qv[k], loc, inv_scale, z0, a1, a2); }
                }
            double a0 = wave_sum63((double)f0);
            a1 = wave_sum63(a1 + (double)f1); a2 = wave_sum63(a2);
            if (lane == 63) { part[wave][W.grp * 3 + 0] = a0; part[wave][W.grp * 3 + 1] = a1; part[wave][W.grp * 3 + 2] = a2; }
        } else {
            hyper_pass_mem(nd, W.grp, W.beg, W.end, e, q, part);
        }
    } else if (wave == HYP_WAVES - 1) {
        for (int g = extra0; g < 2 * nd.nl; ++g) { const HypGroup G = hyp_group(nd, g); hyper_pass_mem(nd, g, G.off, G.off + G.cnt, e, q, part); }
    }
}

// thread t <= ng: finish group t (t < ng) or the Gaussian data term (t == ng): value -> vpart[t], gradient -> grad[]
__device__ __forceinline__ void hyper_finish(const NetDev& nd, int t, const HypPlan& plan, const float* e, double S, long n,
                                             const double (*part)[HYP_MAXG * 3], double* vpart, float* grad) {
    const int ng = 2 * nd.nl;
    if (t < ng) {
        const int l = t >> 1, part_ = t & 1;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
        for (int w = plan.w0[t]; w < plan.w1[t]; ++w) { s0 += part[w][t * 3]; s1 += part[w][t * 3 + 1]; s2 += part[w][t * 3 + 2]; }
        const double cnt = part_ ? nd.out[l] : (double)nd.out[l] * nd.in[l];
        const double loc = e[4 * l + 2 * part_];
        const double gg = e[4 * l + 2 * part_ + 1];
        const double scale = (double)(float)((float)gg * (float)gg);
        double v = 0.0, d_loc, d_scale;
        if (nd.prior[l] == TBNN_PRIOR_CAUCHY) {
            // hyper-priors layer.py:136-153, evaluated :221-228
            v += mvn1_logp(loc, 0.0, 0.2) + mvn1_logp(scale, 0.70710678118654757, 0.5);
            v += s0 - cnt * log(3.14159265358979323846 * scale);   // sum cauchyLogProb (Q1)
            const double inv = 1.0 / scale;                        // one fp64 division per group (serial code: latency counts)
            d_loc = -s1 * inv - loc * 25.0;
            d_scale = -(s2 + cnt) * inv - (scale - 0.70710678118654757) * 4.0;
        } else {
            // hyper-priors layer.py:316-334, evaluated :401-408
            v += mvn1_logp(loc, 0.0, 0.1) + mvn1_logp(scale, 1.0, 0.1);
            const double s = fmin(fmax(scale, 1e-8), 1e8);
            const bool clamped = !(scale > 1e-8 && scale < 1e8);
            v += -0.5 * (2.0 * log(s) + s2 / (s * s) + 1.8378770664093453);   // Q2: k = 1
            d_loc = s1 / (s * s) - loc / 0.01;
            d_scale = (clamped ? 0.0 : (-1.0 / s + s2 / (s * s * s))) - (scale - 1.0) / 0.01;
        }
        grad[4 * l + 2 * part_] = (float)d_loc;
        grad[4 * l + 2 * part_ + 1] = (float)(d_scale * 2.0 * gg);
        vpart[t] = v;
    } else if (t == ng) {
        double v = 0.0;
        if (nd.lik == TBNN_LIK_GAUSSIAN) {                         // network.py:435-438
            const double el = e[nd.H - 1];
            const double sr = (double)(float)((float)el * (float)el);
            const double s = fmin(fmax(sr, 1e-8), 1e8);
            const bool clamped = !(sr > 1e-8 && sr < 1e8);
            const double nel = (double)n * nd.d_out;
            v = -0.5 * (2.0 * nel * log(s) + S / (s * s) + nel * 1.8378770664093453);
            const double ds = clamped ? 0.0 : (-nel / s + S / (s * s * s));
            grad[nd.H - 1] = (float)(ds * 2.0 * el);
        }
        vpart[ng] = v;
    }
}
// hyper entries thread t owns (the ones hyper_finish(t) writes the gradient of): [j0, j1)
__device__ __forceinline__ void hyper_owned(const NetDev& nd, int t, int& j0, int& j1) {
    const int ng = 2 * nd.nl;
    if (t < ng) { j0 = 4 * (t >> 1) + 2 * (t & 1); j1 = j0 + 2; }
    else if (t == ng && nd.lik == TBNN_LIK_GAUSSIAN) { j0 = nd.H - 1; j1 = nd.H; }
    else { j0 = 0; j1 = 0; }
}

__global__ __launch_bounds__(HYP_THREADS) void k_hyper(
    NetDev nd, int mode, float eps, int L, float* __restrict__ eta, const float* __restrict__ q, long n,
    const float* __restrict__ p0_inj, const float* __restrict__ logu_inj, uint32_t epoch, uint32_t key0, uint32_t key1,
    const Scal* __restrict__ sc, float* __restrict__ ws, Scal* __restrict__ out, uint32_t seed_hi = 0,
    const float* __restrict__ eps_each = nullptr)          // per-chain step sizes (tbnn_hyper_step_each: one dual averaging per chain)
{
    if (eps_each) eps = eps_each[blockIdx.x];
    // gridDim.x = chains of a multi-chain handle (one workgroup runs one chain's whole transition): chain c's [H] hypers, [P] weights,
    // record, work space, and its Philox key (seed, chain0 + c) -- key1 = chain_id ^ seed_hi
    if (blockIdx.x) {
        const uint32_t c = blockIdx.x;
        eta += (size_t)c * nd.H; q += (size_t)c * nd.P; sc += c; out += c; ws += (size_t)c * 4 * nd.H;
        key1 = ((key1 ^ seed_hi) + c) ^ seed_hi;
    }
    __shared__ double part[HYP_WAVES][HYP_MAXG * 3];
    __shared__ double vpart[HYP_MAXG + 1];
    __shared__ float e[HYP_MAXH], e0[HYP_MAXH], p[HYP_MAXH], g[HYP_MAXH];
    __shared__ double sh_k0, sh_lp0;
    __shared__ int sh_acc;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, H = nd.H, ng = 2 * nd.nl;
    const double S = sc->stat_cur;
    if (tid < H) { e[tid] = eta[tid]; e0[tid] = eta[tid]; }
    for (int i = tid; i < HYP_WAVES * HYP_MAXG * 3; i += HYP_THREADS) (&part[0][0])[i] = 0.0;
    // deal the waves out (thread 0), then every wave loads its share of theta
    __shared__ HypPlan plan;
    if (tid == 0) hyper_plan(nd, &plan);
    __syncthreads();
    HypWave W;
    W.grp = plan.grp[wave]; W.beg = plan.beg[wave]; W.end = plan.end[wave];
    W.held = W.end - W.beg <= HYP_REG * 64;
#pragma unroll
    for (int k = 0; k < HYP_REG; ++k) {
        const int i = W.beg + lane + 64 * k;
        W.qv[k] = (W.held && i < W.end) ? q[i] : 0.f;
    }
    const int extra0 = plan.extra0;
    int j0, j1;
    hyper_owned(nd, tid, j0, j1);
    auto total = [&]() { double v = 0.0; for (int k = 0; k <= ng; ++k) v += vpart[k]; return v; };
    __syncthreads();
    hyper_partials(nd, W, extra0, e, q, part);
    __syncthreads();
    if (tid <= ng) hyper_finish(nd, tid, plan, e, S, n, part, vpart, g);
    if (mode == HYP_EVAL) {
        __syncthreads();
        if (tid < H) ws[HYP_WS_GRAD * H + tid] = g[tid];
        if (tid == 0) { Scal o = *sc; o.logp_new = total(); *out = o; }
        return;
    }
    if (tid == 0) {
        double k = 0.0;
        for (int j = 0; j < H; ++j) {
            const float v = p0_inj ? p0_inj[j] : philox_normal((uint32_t)j, epoch, PURPOSE_HYPER_MOMENTUM, key0, key1);
            p[j] = v; k += (double)v * (double)v;
        }
        sh_k0 = 0.5 * k;
    }
    __syncthreads();
    if (tid == 0) sh_lp0 = total();
    // half kick + first drift on the entries this thread owns (it wrote their gradient itself)
    for (int j = j0; j < j1; ++j) { p[j] = p[j] + 0.5f * eps * g[j]; if (L >= 1) e[j] = e[j] + eps * p[j]; }
    for (int t = 1; t <= L; ++t) {
        HSTAMP(0);
        __syncthreads();                                           // e[] of this step, vpart consumed
        HSTAMP(1);
        hyper_partials(nd, W, extra0, e, q, part);
        HSTAMP(2);
        __syncthreads();
        HSTAMP(3);
        if (tid <= ng) hyper_finish(nd, tid, plan, e, S, n, part, vpart, g);
        HSTAMP(4);
        for (int j = j0; j < j1; ++j) {
            p[j] = p[j] + eps * g[j];                              // full kick
            if (t < L) e[j] = e[j] + eps * p[j];                   // drift of the next step
            else p[j] = p[j] - 0.5f * eps * g[j];                  // undo half kick
        }
        HSTAMP(5);
    }
    if (L < 1) for (int j = j0; j < j1; ++j) p[j] = p[j] - 0.5f * eps * g[j];
    __syncthreads();
    if (tid == 0) {
        double k1 = 0.0, d2 = 0.0;
        for (int j = 0; j < H; ++j) {
            k1 += (double)p[j] * (double)p[j];
            const double d = (double)e[j] - (double)e0[j];
            d2 += d * d;
        }
        k1 *= 0.5;
        const double val = total();
        double lar = val - sh_lp0 + sh_k0 - k1;
        if (!isfinite(lar)) lar = -INFINITY;
        const double lu = logu_inj ? (double)logu_inj[0] : (double)philox_logu(epoch, PURPOSE_HYPER_LOGU, key0, key1);
        const int a = lu < lar ? 1 : 0;
        Scal o = *sc;
        o.logp_cur = sh_lp0; o.logp_new = val; o.k0 = sh_k0; o.k1 = k1; o.lar = lar; o.logu = lu;
        o.d2 = d2; o.sjd = a ? d2 : 0.0; o.accepted = a;
        *out = o;
        sh_acc = a;
    }
    __syncthreads();
    if (sh_acc != 0 && tid < H) eta[tid] = e[tid];
}

// ---- predictor.trainProbs / reweight (predictor.py:157-273): the sum over the dense layers of calculateHyperProbs
// (layer.py:199-242 Cauchy, :379-422 Gaussian) for m SAVED networks in one launch -- blockIdx.x = network, theta_i and eta_i
// strided; nd.prior[] is the prior family the CALLER wants each layer judged under (reweight loads another architecture).
// The value hyper_finish computes per group, without the data term and without the gradient; sums in double.
__global__ __launch_bounds__(256) void k_hyper_probs(NetDev nd, const float* __restrict__ thetas, long theta_stride,
                                                     const float* __restrict__ etas, long eta_stride, double* __restrict__ out)
{
    __shared__ double red[4];
    const float* q = thetas + (size_t)blockIdx.x * theta_stride;
    const float* e = etas + (size_t)blockIdx.x * eta_stride;
    double total = 0.0;                                          // thread 0's
    for (int grp = 0; grp < 2 * nd.nl; ++grp) {
        const HypGroup G = hyp_group(nd, grp);
        const float loc = e[4 * G.l + 2 * G.part];
        const float gg = e[4 * G.l + 2 * G.part + 1];
        const float inv_scale = 1.f / (gg * gg);                 // layer.py:209-212 (Q3)
        const bool cauchy = nd.prior[G.l] == TBNN_PRIOR_CAUCHY;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
        for (int i = threadIdx.x; i < G.cnt; i += blockDim.x) hyp_term<double>(cauchy, q[G.off + i], loc, inv_scale, a0, a1, a2);
        const double s0 = block_sum(a0, red);
        const double s2 = block_sum(a2, red);
        if (threadIdx.x == 0) {
            const double scale = (double)(float)(gg * gg), cnt = (double)G.cnt;
            if (cauchy) {
                total += mvn1_logp((double)loc, 0.0, 0.2) + mvn1_logp(scale, 0.70710678118654757, 0.5);      // layer.py:221-228
                total += s0 - cnt * log(3.14159265358979323846 * scale);                                     // sum cauchyLogProb (Q1)
            } else {
                total += mvn1_logp((double)loc, 0.0, 0.1) + mvn1_logp(scale, 1.0, 0.1);                      // layer.py:401-408
                const double s = fmin(fmax(scale, 1e-8), 1e8);
                total += -0.5 * (2.0 * log(s) + s2 / (s * s) + 1.8378770664093453);                           // Q2: k = 1
            }
        }
    }
    if (threadIdx.x == 0) out[blockIdx.x] = total;
}
