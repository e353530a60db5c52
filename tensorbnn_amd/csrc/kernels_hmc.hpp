// Leapfrog / energy / Metropolis kernels: everything of one HMC transition that
// is not the fused forward+backward pass.  All of it runs on the device, queued
// on one stream; the host reads one small record per transition.
//
// Restates tfp.mcmc.HamiltonianMonteCarlo (un-vendored; call sites
// network.py:394-408) in the SimpleLeapfrogIntegrator ordering:
//   p = p0 + eps/2 g0 ; L x { q += eps p ; g = grad(q) ; p += eps g } ; p -= eps/2 g
//   lar = logp_L - logp_0 + 1/2|p0|^2 - 1/2|p_L|^2 (non-finite -> -inf); accept iff log u < lar
// and the weight priors layer.py:166-197 / :346-377 with BNN_functions.py:7-57.
#pragma once
#include "common.hpp"
#include "update_ops.hpp"

template <int UC = UPD_COLS, int UG = UPD_GROUPS>
__global__ __launch_bounds__(UC * UG) void k_update(
    NetDev nd, int mode, float eps, const float* __restrict__ eta,
    const float* __restrict__ slabs, int nslab, int pitch,
    const float* __restrict__ q_cur, const float* __restrict__ g_cur,
    float* __restrict__ q, float* __restrict__ p, float* __restrict__ g,
    const int* __restrict__ imgmap, float* __restrict__ qimg, float* __restrict__ gd = nullptr, int img_floats = 0,
    // per-chain step control (tbnn_hmc_step_each): chain blockIdx.y integrates with ITS eps for ITS L steps; t = 0: the opening half
    // kick + drift, t = 1 .. max L: the step index.  (mode UPD_GRAD_ONLY -- the bootstrap evaluation -- is the same for every chain)
    const StepCtl* __restrict__ ctl = nullptr, int t = 0)
{
    if (ctl && mode != UPD_GRAD_ONLY) {
        const StepCtl me = ctl[blockIdx.y];
        eps = me.eps;
        if (t == 0) mode = UPD_FIRST;
        else if (t < me.L) mode = UPD_MID;
        else if (t == me.L) mode = UPD_LAST;
        else return;                                   // this chain's trajectory is complete
    }
    // gridDim.y = chains of a multi-chain handle: [C][P] state arrays, [C][H] hypers, [C][nslab][pitch] slabs, [C][img_floats] images
    if (blockIdx.y) {
        const size_t c = blockIdx.y, cp = c * (size_t)nd.P;
        eta += c * nd.H; slabs += c * (size_t)nslab * pitch;
        q_cur += cp; g_cur += cp; q += cp; p += cp; g += cp;
        if (qimg) qimg += c * (size_t)img_floats;
        if (gd) gd += cp;
    }
    // gd (optional): the DATA term of the gradient alone (the reduced slabs, before the prior is added), stored SIGMA-FREE
    // (times sigma^2 for the Gaussian likelihood): kept next to the state so that a hyper transition, which changes only
    // the prior and the likelihood's sigma, can refresh the cached (log-prob, gradient) of the current state without
    // another pass over the rows (k_refresh_grad_after_hyper) -- and without compounding a rescaling factor over
    // consecutive accepted hyper steps
    __shared__ float4 part[UG][UC];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int c4 = blockIdx.x * UC + tx;                // float4 column
    const int jf = c4 * 4 + ty;                         // ty < 4 selects which of the column's 4 parameters this thread finishes
    const bool fin = ty < 4 && jf < nd.P;
    UpdPre u;
    if (fin) upd_prefetch(u, nd, mode, eta, jf, q_cur, g_cur, q, p, imgmap);
    float4 gs = make_float4(0.f, 0.f, 0.f, 0.f);
    if (mode != UPD_FIRST) {
        part[ty][tx] = upd_column_partial<UG>(slabs, nslab, pitch, c4, ty);
        __syncthreads();
#pragma unroll
        for (int h = UG / 2; h > 0; h >>= 1) {
            if (ty < h) {
                const float4 a = part[ty][tx], b = part[ty + h][tx];
                part[ty][tx] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
            }
            __syncthreads();
        }
        gs = part[0][tx];
    }
    if (!fin) return;
    const float gj = ty == 0 ? gs.x : ty == 1 ? gs.y : ty == 2 ? gs.z : gs.w;
    upd_finish(u, nd, mode, eps, eta, jf, gj, q, p, g, imgmap, qimg, gd);
}

// row-sharded chains: sum the per-workgroup slabs into ONE dense row (the all-reduce operand: doubles when outd is given);
// 64 parameters x 4 slab groups per block, fixed order
__global__ __launch_bounds__(256) void k_slab_reduce(const float* __restrict__ slabs, int nslab, int pitch, int P,
                                                      float* __restrict__ out, double* __restrict__ outd = nullptr) {
    __shared__ float part[4][64];
    const int x = threadIdx.x, y = threadIdx.y, j = blockIdx.x * 64 + x;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (j < P) {
        const float* src = slabs + j;
        int w = y;
        for (; w + 12 < nslab; w += 16) {
            s0 += src[(size_t)w * pitch]; s1 += src[(size_t)(w + 4) * pitch];
            s2 += src[(size_t)(w + 8) * pitch]; s3 += src[(size_t)(w + 12) * pitch];
        }
        for (; w < nslab; w += 4) s0 += src[(size_t)w * pitch];
    }
    part[y][x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (y == 0 && j < P) {
        const float v = (part[0][x] + part[1][x]) + (part[2][x] + part[3][x]);
        if (out) out[j] = v;
        if (outd) outd[j] = (double)v;
    }
}

// row-sharded chains: the operand of the ONE all-reduce per fused pass -- buf[0..P) = the dense gradient row as doubles
// (row == null: k_slab_reduce has already written them), buf[P] = this rank's statistic summed over its workgroups
__global__ __launch_bounds__(256) void k_shard_pack(int P, const float* __restrict__ row, const double* __restrict__ pstat, int nstat,
                                                     double* __restrict__ buf) {
    __shared__ double red[4];
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (row && j < P) buf[j] = (double)row[j];
    if (blockIdx.x == 0) {
        double st = 0.0;
        for (int w = threadIdx.x; w < nstat; w += blockDim.x) st += pstat[w];
        st = block_sum(st, red);
        if (threadIdx.x == 0) buf[P] = st;
    }
}
__global__ __launch_bounds__(256) void k_shard_unpack(int P, const double* __restrict__ buf, float* __restrict__ row,
                                                       double* __restrict__ stat_red) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < P) row[j] = (float)buf[j];
    if (j == 0) stat_red[0] = buf[P];
}

// plain copy (tbnn_get_state: theta into device-mapped pinned host memory)
__global__ __launch_bounds__(256) void k_copy_f32(int n, const float* __restrict__ src, float* __restrict__ dst) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) dst[j] = src[j];
}

// scatter a flat theta into the padded weight image (bootstrap / tbnn_logp_grad)
// blockIdx.y: network of an ensemble (q + y * q_stride -> qimg + y * img_stride; both strides 0 for a single one)
__global__ __launch_bounds__(256) void k_make_image(int P, const float* __restrict__ q,
                                                    const int* __restrict__ imgmap, float* __restrict__ qimg,
                                                    long q_stride = 0, long img_stride = 0) {
    q += (size_t)blockIdx.y * q_stride; qimg += (size_t)blockIdx.y * img_stride;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < P) { const float v = q[j]; qimg[imgmap[j]] = v; const int m1 = imgmap[P + j]; if (m1 >= 0) qimg[m1] = v; }
}

// data-term log-likelihood from the reduced statistic
__device__ __forceinline__ double data_logp(const NetDev& nd, const float* __restrict__ eta, double stat, long n) {
    if (nd.lik == TBNN_LIK_BERNOULLI) return stat;
    // multivariateLogProb with sigma broadcast to [n, d_out] (likelihood.py:92, BNN_functions.py:25-32)
    const double s = (double)lik_sigma(nd, eta);
    const double nel = (double)n * (double)nd.d_out;
    return -0.5 * (2.0 * nel * log(s) + stat / (s * s) + nel * 1.8378770664093453 /* log 2pi */);
}

// weight-prior log-density of the whole theta; every thread returns its
// partial, caller block-reduces.  (layer.py:187-195 / :367-375)
__device__ __forceinline__ double prior_logp_partial(const NetDev& nd, const float* __restrict__ eta,
                                                     const float* __restrict__ q) {
    double acc = 0.0;
    for (int l = 0; l < nd.nl; ++l) {
        for (int part = 0; part < 2; ++part) {
            const int off = part ? nd.offB[l] : nd.offW[l];
            const int cnt = part ? nd.out[l] : nd.out[l] * nd.in[l];
            const float loc = eta[4 * l + 2 * part];
            const float gg = eta[4 * l + 2 * part + 1];
            const float scale = gg * gg;
            if (nd.prior[l] == TBNN_PRIOR_CAUCHY) {
                const float lb = logf(3.14159265358979323846f * scale);   // BNN_functions.py:52
                for (int e = threadIdx.x; e < cnt; e += blockDim.x) {
                    const float z = (q[off + e] - loc) / scale;
                    acc += (double)(logf(1.f + z * z) - lb);               // :51-55 (Q1)
                }
            } else {
                const float s = fminf(fmaxf(scale, 1e-8f), 1e8f);
                for (int e = threadIdx.x; e < cnt; e += blockDim.x) {
                    const float d = (q[off + e] - loc) / s;
                    acc += -0.5 * (double)(d * d);
                }
                // Q2: the normaliser counted once per call (k = size(sigma) = 1)
                if (threadIdx.x == 0) acc += -0.5 * (2.0 * (double)logf(s) + 1.8378770664093453);
            }
        }
    }
    return acc;
}

// Single-workgroup kernel: momentum draw + K0 + log u.
__global__ __launch_bounds__(1024) void k_begin(
    NetDev nd, const float* __restrict__ p0_inj, const float* __restrict__ logu_inj,
    uint32_t epoch, uint32_t key0, uint32_t key1, float* __restrict__ p, Scal* __restrict__ sc, uint32_t seed_hi = 0)
{
    __shared__ double red[16];
    // gridDim.y = chains of a multi-chain handle: chain c draws from the Philox key (seed, chain0 + c) -- key1 = chain_id ^ seed_hi
    if (blockIdx.y) {
        const uint32_t c = blockIdx.y;
        key1 = (((key1 ^ seed_hi) + c) ^ seed_hi);
        p += (size_t)c * nd.P; sc += c;
        if (p0_inj) p0_inj += (size_t)c * nd.P;
        if (logu_inj) logu_inj += c;
    }
    double k = 0.0;
    if (p0_inj) {
        for (int j = threadIdx.x; j < nd.P; j += blockDim.x) { const float v = p0_inj[j]; p[j] = v; k += (double)v * (double)v; }
    } else {
        // one Philox block = four momenta per thread and trip
        for (int b = threadIdx.x; 4 * b < nd.P; b += blockDim.x) {
            float v[4];
            philox_normal4((uint32_t)b, epoch, PURPOSE_MOMENTUM, key0, key1, v);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (4 * b + i < nd.P) { p[4 * b + i] = v[i]; k += (double)v[i] * (double)v[i]; }
        }
    }
    k = block_sum(k, red);
    if (threadIdx.x == 0) {
        sc->k0 = 0.5 * k;
        sc->logu = logu_inj ? (double)logu_inj[0] : (double)philox_logu(epoch, PURPOSE_LOGU, key0, key1);
    }
}

enum { EN_CUR = 0, EN_NEW = 1, EN_TRACE = 2, EN_REFRESH = 3 };   // EN_REFRESH: EN_CUR from the cached statistic sc->stat_cur
// Single-workgroup kernel: total target log-prob (+ kinetic energy and the
// Metropolis decision when which == EN_NEW).
__global__ __launch_bounds__(1024) void k_energy(
    NetDev nd, int which, const float* __restrict__ eta, const float* __restrict__ q,
    const float* __restrict__ p, const float* q_cur,              // (no restrict: the merged commit writes the same array)
    const double* __restrict__ partial_stat, int nslab, long n,
    Scal* __restrict__ sc, double* __restrict__ trace_slot,
    // EN_NEW with commit_out: the transition's end in this one launch -- the record for the host (k_commit_scal) and, when
    // accepted, cur <- proposal (k_commit: q, g, gd); two launches fewer per transition
    Scal* __restrict__ commit_out = nullptr, const float* __restrict__ g = nullptr, float* q_cur_w = nullptr,
    float* __restrict__ g_cur = nullptr, const float* __restrict__ gd = nullptr, float* __restrict__ gd_cur = nullptr,
    // multi-chain EN_REFRESH after a hyper transition: chain blockIdx.y runs only when ITS hyper proposal was accepted
    const Scal* __restrict__ only_if_accepted = nullptr)
{
    __shared__ double red[16];
    __shared__ int s_acc;
    if (only_if_accepted && !only_if_accepted[blockIdx.y].accepted) return;
    // gridDim.y = chains of a multi-chain handle (no trace there: trace_slot is null)
    if (blockIdx.y) {
        const size_t c = blockIdx.y, cp = c * (size_t)nd.P;
        eta += c * nd.H; q += cp; p += cp; q_cur += cp; partial_stat += c * PSTAT_CAP; sc += c;
        if (commit_out) commit_out += c;
        if (g) g += cp;
        if (q_cur_w) q_cur_w += cp;
        if (g_cur) g_cur += cp;
        if (gd) gd += cp;
        if (gd_cur) gd_cur += cp;
    }
    double st = 0.0;
    if (which != EN_REFRESH) {
        for (int w = threadIdx.x; w < nslab; w += blockDim.x) st += partial_stat[w];
        st = block_sum(st, red);
    }
    double pr = prior_logp_partial(nd, eta, q);
    pr = block_sum(pr, red);
    double k1 = 0.0, d2 = 0.0;
    if (which == EN_NEW) {
        for (int j = threadIdx.x; j < nd.P; j += blockDim.x) {
            const double pj = (double)p[j];
            const double dq = (double)q[j] - (double)q_cur[j];
            k1 += pj * pj;
            d2 += dq * dq;
        }
        k1 = block_sum(k1, red);
        d2 = block_sum(d2, red);
    }
    const bool commit = which == EN_NEW && commit_out != nullptr;
    if (threadIdx.x != 0 && !commit) return;
    if (threadIdx.x == 0) {
        if (which == EN_REFRESH) { st = sc->stat_cur; which = EN_CUR; }
        const double lp = pr + data_logp(nd, eta, st, n);
        if (which == EN_TRACE) { *trace_slot = lp; return; }
        if (which == EN_CUR) {
            sc->stat_cur = st; sc->prior_cur = pr; sc->logp_cur = lp;
            if (trace_slot) *trace_slot = lp;
            return;
        }
        sc->stat_new = st; sc->prior_new = pr; sc->logp_new = lp;
        sc->k1 = 0.5 * k1; sc->d2 = d2;
        double lar = lp - sc->logp_cur + sc->k0 - 0.5 * k1;
        if (!isfinite(lar)) lar = -INFINITY;                       // TFP safe_sum / non-finite => reject
        sc->lar = lar;
        const int acc = sc->logu < lar ? 1 : 0;
        sc->accepted = acc;
        sc->sjd = acc ? d2 : 0.0;
        if (trace_slot) *trace_slot = lp;
        if (commit) {                                              // k_commit_scal: the record as the host reads it, then cur <- new
            *commit_out = *sc;
            if (acc) { sc->stat_cur = st; sc->prior_cur = pr; sc->logp_cur = lp; }
            s_acc = acc;
        }
    }
    if (!commit) return;
    __syncthreads();
    if (!s_acc) return;
    for (int j = threadIdx.x; j < nd.P; j += blockDim.x) { q_cur_w[j] = q[j]; g_cur[j] = g[j]; gd_cur[j] = gd[j]; }       // k_commit
}

// After EN_NEW: commit the proposal when accepted.  (scalars are committed by
// k_commit_scal afterwards so that logp_old stays readable for the host.)
__global__ __launch_bounds__(256) void k_commit(
    int P, const Scal* __restrict__ sc, const float* __restrict__ q, const float* __restrict__ g,
    float* __restrict__ q_cur, float* __restrict__ g_cur, const float* __restrict__ gd, float* __restrict__ gd_cur)
{
    const size_t c = blockIdx.y, cp = c * (size_t)P;               // gridDim.y = chains of a multi-chain handle
    if (!sc[c].accepted) return;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < P) { q_cur[cp + j] = q[cp + j]; g_cur[cp + j] = g[cp + j]; gd_cur[cp + j] = gd[cp + j]; }
}

// After an ACCEPTED hyper transition (-> eta): the prediction does not depend on eta, so the data-term gradient at the
// current state only rescales with the likelihood's sigma (Gaussian: 1/sigma^2; fixed-sd / Bernoulli: unchanged) and the
// statistic is the cached one; the prior terms are O(P).  Replaces a whole fused pass over the rows per epoch.  gd_cur is
// the sigma-free data term k_update stored: it is only READ here, so nothing compounds over repeated hyper steps.
__global__ __launch_bounds__(256) void k_refresh_grad_after_hyper(
    NetDev nd, const float* __restrict__ eta, const float* __restrict__ q_cur,
    const float* __restrict__ gd_cur, float* __restrict__ g_cur, const Scal* __restrict__ hyper_rec = nullptr)
{
    // gridDim.y = chains of a multi-chain handle; hyper_rec: the chains' hyper-transition records -- a chain whose proposal was
    // rejected keeps its cached gradient bit for bit
    if (blockIdx.y) {
        const size_t c = blockIdx.y, cp = c * (size_t)nd.P;
        eta += c * nd.H; q_cur += cp; gd_cur += cp; g_cur += cp;
    }
    if (hyper_rec && !hyper_rec[blockIdx.y].accepted) return;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nd.P) return;
    const float sn = lik_sigma(nd, eta);
    const float inv_var = nd.lik == TBNN_LIK_GAUSSIAN ? 1.f / (sn * sn) : 1.f;
    int prior; float loc, scale;
    prior_params(nd, eta, j, prior, loc, scale);
    g_cur[j] = gd_cur[j] * inv_var + prior_grad(prior, loc, scale, q_cur[j]);
}
// copies the record for the host, then rolls cur <- new when accepted
__global__ void k_commit_scal(Scal* __restrict__ sc, Scal* __restrict__ host_copy) {
    if (threadIdx.x != 0) return;
    sc += blockIdx.x; host_copy += blockIdx.x;                     // gridDim.x = chains
    *host_copy = *sc;
    if (sc->accepted) { sc->stat_cur = sc->stat_new; sc->prior_cur = sc->prior_new; sc->logp_cur = sc->logp_new; }
}

// trace slot 0 of a transition that starts from a cached (logp, grad): the chain's own record
__global__ void k_trace_logp_cur(const Scal* __restrict__ sc, double* __restrict__ slot) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *slot = sc->logp_cur;
}

// debug: the chain RNG as tbnn_hmc_step draws it
__global__ void k_debug_draw(uint32_t epoch, uint32_t purpose, uint32_t key0, uint32_t key1, int n,
                             float* __restrict__ normals, float* __restrict__ logu) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) normals[j] = philox_normal((uint32_t)j, epoch, purpose, key0, key1);
    if (j == 0) logu[0] = philox_logu(epoch, purpose + 1u, key0, key1);
}
