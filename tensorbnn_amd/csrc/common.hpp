// Shared device/host definitions for libtbnn (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/tbnn.h"

// a fused kernel's hidden activations as ONE int: a TBNN_ACT_* value (every hidden layer the same), or this flag + 3 bits per hidden layer
// (kernels_fast.hpp: Shape::act; fused_ops.hpp)
#define TBNN_ACT_PACKED 0x40000000

#define TBNN_WAVE 64
#define PSTAT_CAP 512                     // entries of the per-workgroup statistic buffer (tbnn_api.hip): >= the grid of every fused pass

// Network descriptor passed BY VALUE as a kernel argument (lives in SGPRs /
// the kernarg segment: every lookup is wave-uniform).
struct NetDev {
    int nl;                       // dense layers
    int in[TBNN_MAX_LAYERS];      // inputDims   (layer.py:110)
    int out[TBNN_MAX_LAYERS];     // outputDims  (layer.py:111)
    int act[TBNN_MAX_LAYERS];     // activation following the layer
    int prior[TBNN_MAX_LAYERS];   // TBNN_PRIOR_*
    int offW[TBNN_MAX_LAYERS];    // offset of W_l in theta
    int offB[TBNN_MAX_LAYERS];    // offset of b_l in theta
    int actOff[TBNN_MAX_LAYERS];  // offset (in floats per row) of a_{l+1} in the per-row activation record
    int P, H;
    int lik;                      // TBNN_LIK_*
    int d_in, d_out;
    int sumOut;                   // sum of out dims = floats per row in the activation record
    int maxW;                     // widest layer (in or out)
    float fixed_sd;
    int reserved_flags;           // bit 0: fast kernel processes one tile at a time (diagnostic)
};

// Gradient-slab stores of the narrow family: WRITE-THROUGH (global_store ... sc1: a relaxed agent-scope atomic store) when the slabs
// are big next to the launch (P >= 2048 parameters x 256 workgroups: configs[1] writes 5.6 MB in a 47-us launch).  A slab is written
// once per launch and read once by the next kernel (k_update).  Left dirty in the L2s it waits for the end of the launch: between two
// kernels this 8-XCD part writes every dirty line back, and that write-back sat on the critical path of every leapfrog step.
// configs[1], bench.py --workload c2: plain stores 18.81 k leapfrog steps/s (fused pass 49.6 us by hipEvent), non-temporal stores
// 19.31 k (47.9 us), write-through 19.72 k (47.1 us).  Small next to long launches (configs[4], the mid-width kernel: 2,788 -> 2,797 steps/s), neutral for small networks (P < 2048:
// plain stores), negative for the layered family's many small dW launches (8 -> 300 -> 300 -> 1: 493 -> 511 us): plain stores there.
// TBNN_SLAB_NT: 0 plain stores everywhere, 1 non-temporal, 2 write-through (default).
#ifndef TBNN_SLAB_NT
#define TBNN_SLAB_NT 2
#endif
template <bool NT>
__device__ __forceinline__ void slab_store(float* ptr, float val) {
    if constexpr (NT && TBNN_SLAB_NT == 2) __hip_atomic_store(ptr, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if constexpr (NT && TBNN_SLAB_NT == 1) __builtin_nontemporal_store(val, ptr);
    else *ptr = val;
}

// 16 bytes of a slab / stored block at once, write-through: one `global_store_dwordx4 ... sc1` (a 4-byte sc1 store is one fabric
// write per lane: ~6x the time per byte of the 16-byte form, MI355X_MICROARCH.md).  The compiler's hazard recognizer does not see
// an inline-asm store as a VMEM store and would not keep a following VALU write off the store-data VGPRs (2 wait states on a
// > 64-bit store): the s_nop inside the statement provides them whatever the register allocation.
typedef float tb_f32x4 __attribute__((ext_vector_type(4)));
template <bool WT>
__device__ __forceinline__ void store16(float* ptr, const tb_f32x4& v) {
    if constexpr (WT && TBNN_SLAB_NT == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(ptr), "v"(v) : "memory");
    else *reinterpret_cast<tb_f32x4*>(ptr) = v;
}

// Several chains in one launch (round 4): a handle created with tbnn_create_multi keeps its C chains' buffers as [C][...]
// arrays and launches the per-chain kernels with gridDim.y = C; blockIdx.y picks the chain.  Strides the kernels cannot derive
// from NetDev travel in this struct (all zero for one chain: blockIdx.y is 0 then anyway).
// Per-chain step control of a multi-chain handle whose chains run at their own (eps, L) (tbnn_hmc_step_each; the reference runs one
// adapter per chain, network.py:221-235, :603-607): the chains advance in lockstep for max_c L_c leapfrog steps; chain c takes its
// closing half kick at step L_c and is skipped afterwards -- k_update returns for it and the blocks of the fused pass exit at once, so
// its gradient slab and statistic stay those of ITS last step.
struct StepCtl { float eps; int L; };
__device__ __forceinline__ bool chain_done(const StepCtl* __restrict__ ctl, int t, int chain) { return ctl && t > ctl[chain].L; }
struct ChainStride { long img, eta, slab; const StepCtl* ctl; int t; };

// Per-chain scalar record kept on the device (doubles: energies are summed
// and differenced in fp64 -- strictly more accurate than the reference's fp32).
struct Scal {
    double stat_cur, prior_cur, logp_cur;   // at the current (accepted) state
    double stat_new, prior_new, logp_new;   // at the proposal
    double k0, k1;                          // kinetic energies
    double lar;                             // log accept ratio
    double logu;                            // log U(0,1)
    double d2;                              // |q_prop - q_cur|^2
    double sjd;                             // d2 if accepted else 0
    int accepted;
    int pad;
};

// Likelihood standard deviation used on the hot path.
// Gaussian: clip(eta_last^2, 1e-8, 1e8) (likelihood.py:88 + BNN_functions.py:23-24);
// FixedGaussian: sd as given (likelihood.py:162), same clip inside multivariateLogProb.
__device__ __forceinline__ float lik_sigma(const NetDev& nd, const float* __restrict__ eta) {
    float s = (nd.lik == TBNN_LIK_GAUSSIAN) ? eta[nd.H - 1] * eta[nd.H - 1] : nd.fixed_sd;
    s = fmaxf(s, 1e-8f);
    s = fminf(s, 1e8f);
    return s;
}

__device__ __forceinline__ float act_fwd(float z, int act) {
    switch (act) {
        case TBNN_ACT_RELU: return fmaxf(z, 0.f);                  // activationFunctions.py:36
        case TBNN_ACT_TANH: return tanhf(z);                       // :62
        case TBNN_ACT_SIGMOID: return 1.f / (1.f + expf(-z));      // :49
        case TBNN_ACT_EXP: return expf(z);                         // :23
        case TBNN_ACT_ELU: return z > 0.f ? z : expm1f(z);         // :75
        default: return z;
    }
}
// derivative expressed through the activation OUTPUT a
__device__ __forceinline__ float act_bwd(float a, int act) {
    switch (act) {
        case TBNN_ACT_RELU: return a > 0.f ? 1.f : 0.f;
        case TBNN_ACT_TANH: return 1.f - a * a;
        case TBNN_ACT_SIGMOID: return a * (1.f - a);
        case TBNN_ACT_EXP: return a;
        case TBNN_ACT_ELU: return a > 0.f ? 1.f : a + 1.f;
        default: return 1.f;
    }
}

// diagnostic build only (-DTBNN_TILE_STAMPS): shader-clock stamps (per translation unit; tbnn_debug_tile_stamps reads
// the copy of tbnn_api.hip)
#ifdef TBNN_TILE_STAMPS
static __device__ unsigned long long g_tile_stamps[64];
#endif

// wave (64-lane) and block reductions
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
// wave_sum's value in LANE 0 ONLY, bit for bit (the same tree: lane i += lane i + o for o = 32 .. 1), without the six dependent
// LDS-crossbar round trips of __shfl_down on a double (~650 cycles with one wave per SIMD): v_permlane32_swap / v_permlane16_swap
// (gfx950) bring the upper half / the odd rows down, row_shl DPP moves do the steps inside a row.  tools/ubench/wsum.hip checks the
// bits against wave_sum on the device.
template <int CTRL>
__device__ __forceinline__ double dpp_shl_add(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return v + __hiloint2double(hi2, lo2);
}
__device__ __forceinline__ double wave_sum_lane0(double v) {
    {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);      // [1]: lanes 0..31 <- lanes 32..63
        const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v += __hiloint2double((int)b[1], (int)a[1]);
    }
    {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);      // [1]: rows 0, 2 <- rows 1, 3
        const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v += __hiloint2double((int)b[1], (int)a[1]);
    }
    v = dpp_shl_add<0x108>(v);       // row_shl:8: lane i <- lane i + 8 of its row
    v = dpp_shl_add<0x104>(v);
    v = dpp_shl_add<0x102>(v);
    v = dpp_shl_add<0x101>(v);
    return v;
}
__device__ __forceinline__ float wave_sumf(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
// sum over the whole block; result valid in thread 0.  red: >= blockDim/64 doubles of LDS.
__device__ __forceinline__ double block_sum(double v, double* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    double t = 0.0;
    if (w == 0) {
        t = lane < nw ? red[lane] : 0.0;
        t = wave_sum(t);
    }
    return t;
}

// Philox4x32-10 chain RNG (replaces tf.random.set_seed(50), network.py:562).
// key = (seed, chain_id); counter = (block, epoch, purpose, 0).
struct Philox4 { uint32_t v[4]; };
__host__ __device__ __forceinline__ uint32_t tb_mulhi(uint32_t a, uint32_t b) {
    return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
}
__host__ __device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                           uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        const uint32_t p0h = tb_mulhi(0xD2511F53u, c0), p0l = 0xD2511F53u * c0;
        const uint32_t p1h = tb_mulhi(0xCD9E8D57u, c2), p1l = 0xCD9E8D57u * c2;
        const uint32_t n0 = p1h ^ c1 ^ k0, n1 = p1l, n2 = p0h ^ c3 ^ k1, n3 = p0l;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    Philox4 o; o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}
__host__ __device__ __forceinline__ float u01(uint32_t x) {   // (0,1)
    return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f);
}
// j-th standard normal of (epoch, purpose): block j/4, Box-Muller pair (j%4)/2
__device__ __forceinline__ float philox_normal(uint32_t j, uint32_t epoch, uint32_t purpose, uint32_t k0, uint32_t k1) {
    const Philox4 r = philox4x32_10(j >> 2, epoch, purpose, 0u, k0, k1);
    const int h = (j >> 1) & 1;
    const float u1 = u01(r.v[2 * h]), u2 = u01(r.v[2 * h + 1]);
    const float rad = sqrtf(-2.f * logf(u1));
    const float ang = 6.28318530717958647692f * u2;
    return (j & 1) ? rad * sinf(ang) : rad * cosf(ang);
}
// the four normals of Philox block b = j / 4 at once (the same arithmetic as philox_normal, one block cipher instead of four)
__device__ __forceinline__ void philox_normal4(uint32_t b, uint32_t epoch, uint32_t purpose, uint32_t k0, uint32_t k1, float (&out)[4]) {
    const Philox4 r = philox4x32_10(b, epoch, purpose, 0u, k0, k1);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float u1 = u01(r.v[2 * h]), u2 = u01(r.v[2 * h + 1]);
        const float rad = sqrtf(-2.f * logf(u1));
        const float ang = 6.28318530717958647692f * u2;
        out[2 * h] = rad * cosf(ang);
        out[2 * h + 1] = rad * sinf(ang);
    }
}
__device__ __forceinline__ float philox_logu(uint32_t epoch, uint32_t purpose, uint32_t k0, uint32_t k1) {
    const Philox4 r = philox4x32_10(0u, epoch, purpose, 0u, k0, k1);
    return logf(u01(r.v[0]));
}

#define PURPOSE_MOMENTUM 0u
#define PURPOSE_LOGU 1u
#define PURPOSE_HYPER_MOMENTUM 2u
#define PURPOSE_HYPER_LOGU 3u
