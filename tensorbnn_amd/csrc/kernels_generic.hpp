// Generic (runtime-shape) fused forward + likelihood + backward kernel.
//
// Works for any dense architecture / activation / likelihood the C ABI can
// describe.  It is the fallback and the on-device cross-check for the
// shape-specialised MFMA kernel (kernels_fast.hpp); it is plain f32 FMA code:
//   - one thread per data row for the forward chain and the delta chain
//     (weights are wave-uniform -> scalar loads; activations live in a
//     per-workgroup scratch record laid out [unit][row] so every access is a
//     coalesced 256-B wave read);
//   - dW/db: one (i,k) entry per wave at a time, lanes stride the rows of the
//     block, 64-lane shuffle reduction, accumulation into the workgroup's own
//     gradient slab (no atomics -> bitwise reproducible).
//
// Math restated from the reference: dense W@a+b layer.py:278, activations
// activationFunctions.py:36/49/62, Gaussian residual likelihood.py:88-94 +
// BNN_functions.py:23-32, Bernoulli likelihood.py:226-236; reverse mode per
// SURVEY.md A12 (TF autodiff has no source in the tree).
#pragma once
#include "common.hpp"

#define GEN_RB 256   // rows per block iteration == threads per workgroup

// floats of scratch one workgroup needs
static inline size_t generic_scratch_floats(const NetDev& nd) {
    return (size_t)(nd.d_in + nd.sumOut + 3 * nd.maxW) * GEN_RB;
}

__global__ __launch_bounds__(GEN_RB) void k_fwd_bwd_generic(
    NetDev nd, const float* __restrict__ q, const float* __restrict__ eta,
    const float* __restrict__ X, const float* __restrict__ Y, long n,
    float* __restrict__ scratch, size_t scratchPerWG,
    float* __restrict__ partial_grad, int pitch, double* __restrict__ partial_stat)
{
    __shared__ double red[GEN_RB / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* A0 = scratch + (size_t)blockIdx.x * scratchPerWG;     // a_0 = x        [d_in][RB]
    float* ACT = A0 + (size_t)nd.d_in * GEN_RB;                  // a_1..a_L       [sumOut][RB]
    float* DZ = ACT + (size_t)nd.sumOut * GEN_RB;                // dL/dz_l        [maxW][RB]
    float* DA0 = DZ + (size_t)nd.maxW * GEN_RB;                  // dL/da ping
    float* DA1 = DA0 + (size_t)nd.maxW * GEN_RB;                 // dL/da pong
    float* slab = partial_grad + (size_t)blockIdx.x * pitch;

    // zero this workgroup's gradient slab (same ownership pattern as the adds)
    for (int l = 0; l < nd.nl; ++l) {
        const int cnt = nd.out[l] * nd.in[l] + nd.out[l];
        for (int e = wave; e < cnt; e += GEN_RB / 64)
            if (lane == 0) slab[nd.offW[l] + e] = 0.f;
    }

    const float sigma = lik_sigma(nd, eta);
    const float inv_var = 1.f / (sigma * sigma);
    double stat = 0.0;
    const long nblk = (n + GEN_RB - 1) / GEN_RB;

    for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const long R = blk * GEN_RB + tid;
        const bool valid = R < n;
        // stage x (zero for rows past n)
        for (int k = 0; k < nd.d_in; ++k) A0[k * GEN_RB + tid] = valid ? X[R * nd.d_in + k] : 0.f;

        // ---- forward chain: a_l = act(W a_{l-1} + b)
        for (int l = 0; l < nd.nl; ++l) {
            const int in = nd.in[l], out = nd.out[l], act = nd.act[l];
            const float* __restrict__ W = q + nd.offW[l];
            const float* __restrict__ b = q + nd.offB[l];
            const float* ain = (l == 0) ? A0 : ACT + (size_t)nd.actOff[l - 1] * GEN_RB;
            float* aout = ACT + (size_t)nd.actOff[l] * GEN_RB;
            for (int i = 0; i < out; ++i) {
                float z = b[i];
                for (int k = 0; k < in; ++k) z = fmaf(W[i * in + k], ain[k * GEN_RB + tid], z);
                aout[i * GEN_RB + tid] = act_fwd(z, act);
            }
        }

        // ---- likelihood: statistic + dL/df
        {
            const float* f = ACT + (size_t)nd.actOff[nd.nl - 1] * GEN_RB;
            for (int i = 0; i < nd.d_out; ++i) {
                const float fi = f[i * GEN_RB + tid];
                const float y = valid ? Y[R * nd.d_out + i] : 0.f;
                float da = 0.f;
                if (nd.lik == TBNN_LIK_BERNOULLI) {
                    const float p = fminf(fmaxf(fi, 1e-8f), 1.f - 1e-7f);     // likelihood.py:226-231
                    const bool inside = (fi > 1e-8f) && (fi < 1.f - 1e-7f);
                    if (valid) {
                        // tfd.Bernoulli.log_prob = xlogy(y,p) + xlog1py(1-y,-p)
                        const float t1 = (y == 0.f) ? 0.f : y * logf(p);
                        const float t2 = (1.f - y == 0.f) ? 0.f : (1.f - y) * log1pf(-p);
                        stat += (double)(t1 + t2);
                        da = inside ? (y / p - (1.f - y) / (1.f - p)) : 0.f;
                    }
                } else {
                    const float r = y - fi;
                    if (valid) { stat += (double)r * (double)r; da = r * inv_var; }
                }
                DA0[i * GEN_RB + tid] = da;
            }
        }

        // ---- backward chain
        float* cur = DA0;
        float* nxt = DA1;
        for (int l = nd.nl - 1; l >= 0; --l) {
            const int in = nd.in[l], out = nd.out[l], act = nd.act[l];
            const float* __restrict__ W = q + nd.offW[l];
            const float* ain = (l == 0) ? A0 : ACT + (size_t)nd.actOff[l - 1] * GEN_RB;
            const float* aout = ACT + (size_t)nd.actOff[l] * GEN_RB;
            for (int i = 0; i < out; ++i)
                DZ[i * GEN_RB + tid] = cur[i * GEN_RB + tid] * act_bwd(aout[i * GEN_RB + tid], act);
            __syncthreads();
            // dW_l = dz a_{l-1}^T, db_l = sum_rows dz : one entry per wave at a time
            const int nW = out * in, cnt = nW + out;
            for (int e = wave; e < cnt; e += GEN_RB / 64) {
                float s = 0.f;
                if (e < nW) {
                    const int i = e / in, k = e - i * in;
#pragma unroll
                    for (int r = lane; r < GEN_RB; r += 64) s = fmaf(DZ[i * GEN_RB + r], ain[k * GEN_RB + r], s);
                } else {
                    const int i = e - nW;
#pragma unroll
                    for (int r = lane; r < GEN_RB; r += 64) s += DZ[i * GEN_RB + r];
                }
                s = wave_sumf(s);
                if (lane == 0) slab[nd.offW[l] + e] += s;
            }
            // dL/da_{l-1} = W^T dz
            if (l > 0) {
                for (int k = 0; k < in; ++k) {
                    float s = 0.f;
                    for (int i = 0; i < out; ++i) s = fmaf(W[i * in + k], DZ[i * GEN_RB + tid], s);
                    nxt[k * GEN_RB + tid] = s;
                }
            }
            __syncthreads();
            float* t = cur; cur = nxt; nxt = t;
        }
    }
    const double tot = block_sum(stat, red);
    if (tid == 0) partial_stat[blockIdx.x] = tot;
}

// network.predict (network.py:141-171): out[d_out][n]
__global__ __launch_bounds__(GEN_RB) void k_forward_generic(
    NetDev nd, const float* __restrict__ q, const float* __restrict__ X, long n,
    float* __restrict__ scratch, size_t scratchPerWG, float* __restrict__ out)
{
    const int tid = threadIdx.x;
    float* A0 = scratch + (size_t)blockIdx.x * scratchPerWG;
    float* ACT = A0 + (size_t)nd.d_in * GEN_RB;
    const long nblk = (n + GEN_RB - 1) / GEN_RB;
    for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const long R = blk * GEN_RB + tid;
        const bool valid = R < n;
        for (int k = 0; k < nd.d_in; ++k) A0[k * GEN_RB + tid] = valid ? X[R * nd.d_in + k] : 0.f;
        for (int l = 0; l < nd.nl; ++l) {
            const int in = nd.in[l], od = nd.out[l], act = nd.act[l];
            const float* __restrict__ W = q + nd.offW[l];
            const float* __restrict__ b = q + nd.offB[l];
            const float* ain = (l == 0) ? A0 : ACT + (size_t)nd.actOff[l - 1] * GEN_RB;
            float* aout = ACT + (size_t)nd.actOff[l] * GEN_RB;
            for (int i = 0; i < od; ++i) {
                float z = b[i];
                for (int k = 0; k < in; ++k) z = fmaf(W[i * in + k], ain[k * GEN_RB + tid], z);
                aout[i * GEN_RB + tid] = act_fwd(z, act);
            }
        }
        if (valid) {
            const float* f = ACT + (size_t)nd.actOff[nd.nl - 1] * GEN_RB;
            for (int i = 0; i < nd.d_out; ++i) out[(size_t)i * n + R] = f[i * GEN_RB + tid];
        }
    }
}

// metrics.py:30-141 on the device: one pass over the predictions f[d_out][n] and the targets Y[n][d_out];
//   p = f*sd + mean, r = y*sd + mean (metrics.py:36-42), optionally exp() of either (scaleExp; SquaredError does
//   not exponentiate the validation predictions, :44-47 -- the caller passes the flags)
//   part[3b+0] += (p-r)^2, part[3b+1] += 100|(p-r)/r|, part[3b+2] += |r - round(p)|   (round half to even, tf.round)
__global__ __launch_bounds__(256) void k_metrics(const float* __restrict__ f, const float* __restrict__ Y, long n, int d_out,
                                                  float mean, float sd, int exp_pred, int exp_real, double* __restrict__ part) {
    __shared__ double red[8];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    const long tot = n * d_out;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (long)gridDim.x * blockDim.x) {
        const long row = e / d_out; const int o = (int)(e - row * d_out);
        float p = f[(size_t)o * n + row] * sd + mean;
        float r = Y[e] * sd + mean;
        if (exp_pred) p = expf(p);
        if (exp_real) r = expf(r);
        const float d = p - r;
        a0 += (double)(d * d);
        a1 += (double)(fabsf(d / r) * 100.f);
        a2 += (double)fabsf(r - rintf(p));
    }
    a0 = block_sum(a0, red); a1 = block_sum(a1, red); a2 = block_sum(a2, red);
    if (threadIdx.x == 0) { part[3 * blockIdx.x] = a0; part[3 * blockIdx.x + 1] = a1; part[3 * blockIdx.x + 2] = a2; }
}
