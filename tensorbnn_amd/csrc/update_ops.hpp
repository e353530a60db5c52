// The leapfrog update of one parameter block -- slab reduction, prior gradient, kick / drift, image scatter -- as inline
// pieces (k_update in kernels_hmc.hpp; tools/ubench/tail_probe.hip re-uses them for the in-kernel variant that was measured
// and not kept).
// Priors: layer.py:166-197 / :346-377 with BNN_functions.py:7-57; leapfrog ordering: see kernels_hmc.hpp.
#pragma once
#include "common.hpp"

enum { UPD_GRAD_ONLY = 0, UPD_FIRST = 1, UPD_MID = 2, UPD_LAST = 3 };

// which (layer, W-or-b) group a flat parameter index belongs to -> (loc, scale)
__device__ __forceinline__ void prior_params(const NetDev& nd, const float* __restrict__ eta, int j,
                                             int& prior, float& loc, float& scale) {
    int l = 0;
#pragma unroll 1
    for (int m = 1; m < nd.nl; ++m) if (j >= nd.offW[m]) l = m;
    const bool isb = j >= nd.offB[l];
    prior = nd.prior[l];
    loc = eta[4 * l + (isb ? 2 : 0)];
    const float g = eta[4 * l + (isb ? 3 : 1)];
    scale = g * g;                                   // layer.py:178,180 / :358,360 (Q3)
}

// d/dx of the reference's prior log-density (Q1 sign kept)
__device__ __forceinline__ float prior_grad(int prior, float loc, float scale, float x) {
    if (prior == TBNN_PRIOR_CAUCHY) {
        const float z = (x - loc) / scale;           // BNN_functions.py:51
        return 2.f * z / (scale * (1.f + z * z));
    }
    const float s = fminf(fmaxf(scale, 1e-8f), 1e8f);   // BNN_functions.py:23-24
    return -(x - loc) / (s * s);
}

// Reduce the per-workgroup gradient slabs, add the prior gradient, then kick /
// drift.  One thread column = 4 consecutive parameters (float4 slab reads; the
// slab pitch is a multiple of 4), UPD_GROUPS slab groups per block, every thread keeps
// 4 independent 16-B loads in flight; fixed-order LDS tree => deterministic.
// When imgmap != null the new position is also scattered into the padded
// weight image (W_l and, for l >= 1, W_l^T: imgmap[j] / imgmap[P+j]) the
// shape-specialised kernel stages into LDS.
// (geometry measured at configs[1] after the slabs became write-through: 8 x 32: 19.92 k leapfrog steps/s, 16 x 16: 19.94, 4 x 64: 19.57,
// 8 x 16: 19.91, 16 x 32: 20.12, 32 x 32: 20.00, 16 x 64: 19.82, 32 x 16: 19.98)
#ifndef UPD_COLS
#define UPD_COLS 16    // float4 columns per block (64 parameters)
#endif
#ifndef UPD_GROUPS
#define UPD_GROUPS 32  // slab groups per block
#endif
// networks with many parameters (P >= UPD_BIG_P: the tall-fan-in family's 784 -> 20 -> 20 -> 1 has 15.7 k, its k_update reads 16 MB of
// slabs): 32 columns x 16 groups per block -- measured on that shape (round 5): 16 x 32: 33.35 k leapfrog steps/s, 32 x 16: 34.3 k,
// 32 x 32: 34.0 k, 64 x 16: 33.3 k, 64 x 8: 32.4 k, 128 x 8: 30.5 k, 8 x 32: 32.6 k; configs[3] / [4] (12.4 k / 82.8 k): no difference
#define UPD_BIG_P 8192
#define UPD_COLS_BIG 32
#define UPD_GROUPS_BIG 16

// what a finishing thread (one parameter) needs besides the reduced gradient; fetched BEFORE the slab loads: one memory
// round trip instead of two on the critical path of a leapfrog step
struct UpdPre {
    float q_j = 0.f, p_j = 0.f, gc_j = 0.f, qc_j = 0.f, loc = 0.f, scale = 1.f;
    int prior = 0, m0 = -1, m1 = -1;
};
__device__ __forceinline__ void upd_prefetch(UpdPre& u, const NetDev& nd, int mode, const float* __restrict__ eta, int jf,
                                             const float* __restrict__ q_cur, const float* __restrict__ g_cur,
                                             const float* __restrict__ q, const float* __restrict__ p, const int* __restrict__ imgmap) {
    if (mode != UPD_FIRST) { prior_params(nd, eta, jf, u.prior, u.loc, u.scale); u.q_j = q[jf]; }
    if (mode == UPD_FIRST) { u.gc_j = g_cur[jf]; u.qc_j = q_cur[jf]; }
    if (mode != UPD_GRAD_ONLY) u.p_j = p[jf];
    if (imgmap && (mode == UPD_FIRST || mode == UPD_MID)) { u.m0 = imgmap[jf]; u.m1 = imgmap[nd.P + jf]; }
}
// one thread's share of a float4 column: slabs ty, ty + 32, ... in a fixed order, 8 independent 16-B loads in flight
template <int UPD_GROUPS_T = UPD_GROUPS>
__device__ __forceinline__ float4 upd_column_partial(const float* slabs, int nslab, int pitch, int c4, int ty) {
    constexpr int UG = UPD_GROUPS_T;
    float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 s0 = z, s1 = z, s2 = z, s3 = z;
    if (c4 * 4 < pitch) {
        const float4* base = reinterpret_cast<const float4*>(slabs) + c4;
        const int p4 = pitch >> 2;
        int w = ty;
        for (; w + 7 * UG < nslab; w += 8 * UG) {
            float4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = base[(size_t)(w + k * UG) * p4];
#pragma unroll
            for (int k = 0; k < 8; k += 4) {
                s0.x += v[k].x; s0.y += v[k].y; s0.z += v[k].z; s0.w += v[k].w;
                s1.x += v[k + 1].x; s1.y += v[k + 1].y; s1.z += v[k + 1].z; s1.w += v[k + 1].w;
                s2.x += v[k + 2].x; s2.y += v[k + 2].y; s2.z += v[k + 2].z; s2.w += v[k + 2].w;
                s3.x += v[k + 3].x; s3.y += v[k + 3].y; s3.z += v[k + 3].z; s3.w += v[k + 3].w;
            }
        }
        for (; w < nslab; w += UG) {
            const float4 a = base[(size_t)w * p4];
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
        }
    }
    float4 s;
    s.x = (s0.x + s1.x) + (s2.x + s3.x); s.y = (s0.y + s1.y) + (s2.y + s3.y);
    s.z = (s0.z + s1.z) + (s2.z + s3.z); s.w = (s0.w + s1.w) + (s2.w + s3.w);
    return s;
}
// the finishing thread of parameter j: prior gradient, kick / drift, image scatter
__device__ __forceinline__ void upd_finish(const UpdPre& u, const NetDev& nd, int mode, float eps, const float* __restrict__ eta, int j,
                                           float gj, float* __restrict__ q, float* __restrict__ p, float* __restrict__ g,
                                           const int* __restrict__ imgmap, float* __restrict__ qimg, float* __restrict__ gd) {
    // (the gradient and its data term are kept for the state a transition ENDS in -- k_commit copies them, the hyper refresh reads
    // them; between two leapfrog steps nobody reads them: UPD_MID stores neither)
    if (mode != UPD_FIRST && mode != UPD_MID && gd) {
        const float sg = nd.lik == TBNN_LIK_GAUSSIAN ? lik_sigma(nd, eta) : 1.f;
        gd[j] = gj * (sg * sg);
    }
    if (mode != UPD_FIRST) gj += prior_grad(u.prior, u.loc, u.scale, u.q_j);
    if (mode == UPD_GRAD_ONLY) { g[j] = gj; return; }
    if (mode == UPD_FIRST) {
        const float pj = u.p_j + 0.5f * eps * u.gc_j;        // half kick
        const float qj = u.qc_j + eps * pj;                   // drift
        p[j] = pj; q[j] = qj;
        if (imgmap) { qimg[u.m0] = qj; if (u.m1 >= 0) qimg[u.m1] = qj; }
        return;
    }
    float pj = u.p_j + eps * gj;                              // full kick
    if (mode == UPD_MID) {
        const float qj = u.q_j + eps * pj;                    // drift
        p[j] = pj; q[j] = qj;
        if (imgmap) { qimg[u.m0] = qj; if (u.m1 >= 0) qimg[u.m1] = qj; }
    } else {
        g[j] = gj;
        pj = pj - 0.5f * eps * gj;                            // undo half kick
        p[j] = pj;
    }
}

