// The LAYERED family: forward + likelihood + backward for ANY dense architecture the C ABI can describe, on the f32 MFMA,
// with run-time shapes -- what a network runs on when no shape-specialised fused kernel covers it (fan-in above 32, hidden
// widths above 256, more than two outputs, ... -- e.g. the reference's own MNIST example, 784 -> 20 -> 20 -> 1,
// docs/ClassificationExample.md) and no compiler is around to instantiate one (tensorbnn_amd/jit.py).  It replaces the
// thread-per-row scalar kernel (kernels_generic.hpp: ~0.1-0.7 % of the MFMA peak) as the fallback; that one stays as the
// on-device cross-check (TBNN_KERNEL_GENERIC).
//
// Layer by layer, activations through HBM (the fused families keep them in registers; this one trades that for generality):
//   k_lay_pack_x   once per data set: X[n][d_in] -> a_0 blocks
//   k_lay_gemm<0>  per layer, forward:  a_{l+1} = act_l(W_l [a_l, 1])            (the bias rides as the weight of a constant-1 slot)
//   k_lay_lik      likelihood: statistic + dL/dz of the last layer
//   k_lay_gemm<1>  per layer, backward: dz_{l-1} = (W_l^T dz_l) * act_{l-1}'(a_l)
//   k_lay_dw       every layer's dW_l = dz_l^T [a_l, 1] over a workgroup's rows -> one gradient slab per workgroup
//                  (k_update reduces the slabs, as for every other family: deterministic)
// Storage: 1-KB blocks [16 rows][16 slots] row-major, [row tile][slot tile]; a_l carries a constant-1 slot behind its last
// unit.  The GEMM reads both MFMA operands with ONE 16-byte load per lane and k-group (lane (i, g): W[16t + i][16kg + 4g ..],
// a[row i][16kg + 4g ..]: the four values are the lane's k-steps -- the k order inside a k-group is permuted the same way
// for both operands) and stores a result tile with one 16-byte store; k_lay_dw reads a block lane-linearly (float 64 s + lane =
// row 4s + g, slot i: exactly the operand of k-step s).  Weights come from the padded image k_update maintains
// (W_l row-major [out slots][in slots + 1], and W_l^T for the delta chain), straight from L2.
//
// Reference math: layer.py:278 (W@a+b), activationFunctions.py:23-75, likelihood.py:88-94 + BNN_functions.py:23-32,
// likelihood.py:226-236; reverse mode SURVEY A12; the path: network.py:394-408.
#pragma once
#include <vector>
#include <algorithm>
#include "kernels_fast.hpp"

struct LayPlan {
    int nl;
    int TK[TBNN_MAX_LAYERS];    // slot tiles of a_l (units + the ones slot): ceil((in + 1) / 16)
    int TM[TBNN_MAX_LAYERS];    // unit tiles of z_l / dz_l: ceil(out / 16)
    int TO[TBNN_MAX_LAYERS];    // tiles the forward GEMM of layer l writes: TK[l + 1] (units + ones slot), last layer TM[l]
    int wOff[TBNN_MAX_LAYERS];  // image: W_l   [16 TO[l]][16 TK[l]]
    int tOff[TBNN_MAX_LAYERS];  // image: W_l^T [16 TM[l - 1]][16 TM[l]]   (l >= 1)
    int img_floats;
    // per data set (ntiles row tiles): float offsets of the block arrays in the activation store
    long ntiles;
    long aOff[TBNN_MAX_LAYERS + 1];   // a_l, l = 0 .. nl (a_nl = the network output f)
    long dOff[TBNN_MAX_LAYERS];       // dz_l
    long store_floats;
    int NS;                           // gradient slabs = row ranges of k_lay_dw (grid.x)
    int NY;                           // k_lay_dw grid.y: splits the tile-block list
    int NLK;                          // workgroups of k_lay_lik
    int dw_items;                     // tile blocks over all layers
    // the fused tail (k_lay_tail): layers l0 .. nl-1 (all narrow enough) + likelihood + their delta chain in ONE launch
    int l0;                           // nl: no tail
    int TT;                           // its compile-time tile bound: 2 or 4
    int GT;                           // its grid; NP = entries of the statistic buffer in use = max(NS, GT)
    int NP;
    int tail;                         // this data set runs the tail (set per data set: lay_plan_rows)
    // round 5: a last layer of ONE output tile (<= 16 outputs) behind a layer too wide for the tail above (up to 8 tiles = 127 units + ones
    // slot): its forward GEMM, the likelihood and the first backward GEMM as one launch (k_lay_last) -- three ~5-us launches otherwise
    int last_ok;                      // the shape allows it (lay_plan_shape)
    int last;                         // this data set runs it (lay_plan_rows)
};

// a result tile's 16 bytes per lane.  Every block this family stores is read by a LATER kernel, and between two kernels this part
// writes every dirty L2 line back: stored write-through (sc1) nothing waits dirty for a launch's end (784-20-20-1: 56.4 -> 55.0 us per
// leapfrog step, 100-50-50-1 at 1e5 rows: 153.4 -> 149.6; non-temporal stores instead: 56.4 -> 56.8 / 158.5 -- the reader then misses
// L2 AND the memory-side cache).  LAY_ST_POLICY=0: plain stores.
#ifndef LAY_ST_POLICY
#define LAY_ST_POLICY 2
#endif
__device__ __forceinline__ void lay_block_store(float* p, const f32x4& v) {
#if LAY_ST_POLICY == 2
    // (the s_nop: wait states between this > 64-bit store and a VALU write of its data registers, which the compiler's hazard
    // recognizer does not insert behind inline asm)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
#else
    *reinterpret_cast<f32x4*>(p) = v;
#endif
}
#define LAY_RSRC_FLAGS 0x00020000           // raw buffer resources (as kernels_wide.hpp)
static inline int lay_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

static inline void lay_plan_shape(const NetDev& nd, LayPlan& p) {
    p.nl = nd.nl;
    int o = 0;
    for (int l = 0; l < nd.nl; ++l) { p.TK[l] = lay_cdiv(nd.in[l] + 1, 16); p.TM[l] = lay_cdiv(nd.out[l], 16); }
    for (int l = 0; l < nd.nl; ++l) p.TO[l] = l + 1 < nd.nl ? p.TK[l + 1] : p.TM[l];
    for (int l = 0; l < nd.nl; ++l) { p.wOff[l] = o; o += 256 * p.TO[l] * p.TK[l]; }
    for (int l = 0; l < nd.nl; ++l) { p.tOff[l] = o; if (l >= 1) o += 256 * p.TM[l - 1] * p.TM[l]; }
    p.img_floats = (o + 3) & ~3;
    p.ntiles = 0; p.store_floats = 0; p.NS = 1; p.NY = 1; p.NLK = 1; p.dw_items = 0;
    // the longest suffix of layers whose tile counts all fit the tail kernel's register arrays (<= 4 tiles = 63 units + ones slot)
    p.l0 = nd.nl; p.TT = 2;
    while (p.l0 >= 1 && p.TK[p.l0 - 1] <= 4 && p.TO[p.l0 - 1] <= 4 && p.TM[p.l0 - 1] <= 4) --p.l0;
    for (int l = p.l0; l < nd.nl; ++l) if (p.TK[l] > 2 || p.TO[l] > 2 || p.TM[l] > 2 || (l >= 1 && p.TM[l - 1] > 2)) p.TT = 4;
    if (const char* e = getenv("TBNN_LAY_TAIL")) if (atoi(e) == 0) p.l0 = nd.nl;       // A/B runs: one launch per layer and direction
    p.GT = 1; p.NP = 1; p.tail = 0;
    p.last = 0;
    p.last_ok = p.l0 == nd.nl && nd.nl >= 2 && p.TM[nd.nl - 1] == 1 && p.TK[nd.nl - 1] <= 8 && p.TM[nd.nl - 2] <= 8;
    if (const char* e = getenv("TBNN_LAY_LAST")) if (atoi(e) == 0) p.last_ok = 0;       // A/B runs
}
// k_lay_dw's tile blocks: LAY_DBU output tiles x LAY_DBK input tiles per wave.  The kernel is bound by operand traffic (an operand
// block feeds LAY_DBK resp. LAY_DBU MFMAs per k-step): 4 x 4 blocks load 0.5 blocks per MFMA.  8 x 4 blocks (0.375; -DLAY_DBU=8)
// need 256 + 256 registers, one wave per SIMD, and run SLOWER: 8 -> 300 -> 300 -> 1 at 5e4 rows 180 us against 122 us.
#ifndef LAY_DBU
#define LAY_DBU 4
#endif
#define LAY_DBK 4
static inline void lay_plan_rows(const NetDev& nd, long n, LayPlan& p) {
    p.ntiles = (n + 15) / 16;
    long o = 0;
    for (int l = 0; l <= nd.nl; ++l) { p.aOff[l] = o; o += p.ntiles * 256 * (l < nd.nl ? p.TK[l] : p.TM[nd.nl - 1]); }
    for (int l = 0; l < nd.nl; ++l) { p.dOff[l] = o; o += p.ntiles * 256 * p.TM[l]; }
    p.store_floats = o;
    p.NLK = (int)std::max<long>(1, std::min<long>(256, (p.ntiles * 16 * nd.d_out + 255) / 256));         // (row, output) elements per thread: ~1
    p.dw_items = 0;
    for (int l = 0; l < nd.nl; ++l) p.dw_items += lay_cdiv(p.TM[l], LAY_DBU) * lay_cdiv(p.TK[l], LAY_DBK);       // tile blocks (k_lay_dw)
    // k_lay_dw's grid: NY workgroups share out the tile blocks (one block per wave when that takes at most 16 of them), NS row
    // ranges -- one gradient slab each -- so that NS x NY x 4 waves are about 2048 (2 per SIMD; its row loop is double-buffered),
    // and no more: every (slab, block) pair pays an epilogue of ~640 vector instructions + 64 scattered stores, and k_update reads
    // NS x P floats (8 -> 300 -> 300 -> 1 at 5e4 rows with round 3's 256 slabs: 95 MB, 12 row tiles per block and slab).
    // Measured and not kept (round 4): (block, row range) pairs of EQUAL MFMA count from a host-built list (small blocks get long
    // ranges), in item order 128-187 us, grouped by rows and dealt to the XCDs 146-180 us, against 121-125 us for this plain grid:
    // the kernel is bound by operand traffic (every stored block is read by ~5 tile blocks: 600 MB per gradient at 8 -> 300 ->
    // 300 -> 1), and the plain grid's co-resident waves walk the SAME rows at the same time.
    p.NY = std::max(1, std::min(16, lay_cdiv(p.dw_items, 4)));
    long waves = 2048;
    if (const char* e = getenv("TBNN_LAY_DW_WAVES")) { const long w = atol(e); if (w >= 4) waves = w; }
    const long ns_want = (waves + 4 * p.NY - 1) / (4 * p.NY);
    p.NS = (int)std::max<long>(1, std::min<long>(std::min<long>(256, ns_want), p.ntiles / 8));
    // the tail's grid: at most PSTAT_CAP workgroups (one statistic entry each) of four waves, every wave the same number of row
    // tiles; its grid-stride loop takes whatever a clamped grid leaves
    const long cap = 4L * PSTAT_CAP;
    const long rounds = std::max<long>(1, (p.ntiles + cap - 1) / cap);
    p.GT = (int)std::min<long>(PSTAT_CAP, std::max<long>(1, (p.ntiles + 4 * rounds - 1) / (4 * rounds)));
    p.tail = p.l0 < nd.nl && p.ntiles < cap;       // it saves launches; over many rows the separate GEMMs are as fast (measured: r03_notes)
    p.last = p.last_ok && !p.tail && p.ntiles < cap;
    p.NP = (p.tail || p.last) ? std::max(p.NLK, p.GT) : p.NLK;
}
// theta index j -> image positions: map[j] (W_l, biases in the ones-slot column), map[P + j] (W_l^T; -1: none)
static inline void lay_image_map(const NetDev& nd, const LayPlan& p, int* map) {
    for (int l = 0; l < nd.nl; ++l) {
        const int in = nd.in[l], out = nd.out[l], wp = 16 * p.TK[l];
        for (int u = 0; u < out; ++u) {
            for (int k = 0; k < in; ++k) {
                const int j = nd.offW[l] + u * in + k;
                map[j] = p.wOff[l] + u * wp + k;
                map[nd.P + j] = l >= 1 ? p.tOff[l] + k * (16 * p.TM[l]) + u : -1;
            }
            map[nd.offB[l] + u] = p.wOff[l] + u * wp + in;
            map[nd.P + nd.offB[l] + u] = -1;
        }
    }
}

// X[n][d_in] -> a_0 blocks [row tile][TK0][16][16]: units, a 1 in slot d_in, zeros behind; rows past n: zeros, slot d_in = 1
__global__ __launch_bounds__(256) void k_lay_pack_x(const float* __restrict__ X, long n, int d_in, int TK0, long ntiles, float* __restrict__ a0) {
    const long total = ntiles * TK0 * 256;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long blk = e >> 8; const int r = (int)(e & 255) >> 4, c = (int)(e & 15);
        const long rt = blk / TK0; const int kt = (int)(blk - rt * TK0);
        const long row = rt * 16 + r; const int slot = 16 * kt + c;
        float v = 0.f;
        if (slot < d_in) v = row < n ? X[row * d_in + slot] : 0.f;
        else if (slot == d_in) v = 1.f;
        a0[e] = v;
    }
}

// OUT[rt][t] = epilogue( sum over k-groups  IMG[16 t + i][16 kg + ..] x IN[rt][kg] ),  one wave = RB row tiles x up to LAY_TB (1, 2 or 4) output tiles.
// MODE 0 (forward): v = act(acc) for the n_units real units, 1 in the ones slot, 0 behind.
// MODE 1 (backward): v = acc * act'(AUX[rt][t]) for the n_units real units (AUX = the activations those units produced), 0 behind.
// SK (split K over the workgroup): when the work items do not fill the chip (few rows, long fan-in: 784 -> 20 at n = 12 k is 750
// items of 50 dependent k-groups each) one item goes to a WORKGROUP, its 4 waves take a quarter of the k-groups each with 8
// k-groups of loads in flight, and the partial tiles meet in LDS.  Otherwise one item per wave.
template <int MODE, int RB, bool SK, int LAY_TB>
__global__ __launch_bounds__(256) void k_lay_gemm(
    const float* __restrict__ img, int wpitch, const float* __restrict__ in, int KG, float* __restrict__ outb, int MT,
    const float* __restrict__ aux, int auxT, long ntiles, int act, int n_units, int ones_slot, int rev)
{
    __shared__ f32x4 part[SK ? 4 * RB * LAY_TB * 64 : 1];
    const int lane = threadIdx.x & 63, i16 = lane & 15, g = lane >> 4, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int TG = (MT + LAY_TB - 1) / LAY_TB;
    const long RP = (ntiles + RB - 1) / RB;
    const long items = RP * TG;
    const long first = SK ? (long)blockIdx.x : (long)blockIdx.x * 4 + wave, stride = SK ? (long)gridDim.x : (long)gridDim.x * 4;
    const int KQ = SK ? (KG + 3) / 4 : KG;
    const int k_lo = SK ? wave * KQ : 0, k_hi = SK ? (k_lo + KQ < KG ? k_lo + KQ : KG) : KG;
    for (long itf = first; itf < items; itf += stride) {
        // rev: walk the row tiles from the far end -- consecutive GEMMs alternate, so a launch starts on the rows whose blocks the launch
        // before it touched last (an XCD's L2 still holds that launch's last 4 MB)
        const long it = rev ? items - 1 - itf : itf;
        const long rp = it / TG; const int t0 = (int)(it - rp * TG) * LAY_TB;
        const long rt0 = rp * RB;
        f32x4 acc[RB][LAY_TB];
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int t = 0; t < LAY_TB; ++t) acc[r][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        // operands by buffer loads: tiles / row tiles past the end are CLAMPED to the last valid one (their accumulators are never
        // stored), so every load of the k loop is unconditional; the lane's offsets are constant VGPRs per item, the k-group is an
        // SGPR offset -- no vector instruction computes an address inside the k loop
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(img), 0, MT * 16 * wpitch * 4, LAY_RSRC_FLAGS);
        const long rows_here = ntiles - rt0 < RB ? ntiles - rt0 : RB;
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in) + (size_t)rt0 * KG * 256, 0, (int)rows_here * KG * 1024, LAY_RSRC_FLAGS);
        int wv[LAY_TB], bo[RB];
#pragma unroll
        for (int t = 0; t < LAY_TB; ++t) { const int tt = t0 + t < MT ? t0 + t : MT - 1; wv[t] = ((16 * tt + i16) * wpitch + 4 * g) * 4; }
        const int bv = (i16 * 16 + 4 * g) * 4;
#pragma unroll
        for (int r = 0; r < RB; ++r) bo[r] = (r < rows_here ? r : (int)rows_here - 1) * KG * 1024;
        // two operand sets: k-group kg + 1 is requested before the MFMAs of k-group kg
        auto ldk = [&](f32x4 (&A)[LAY_TB], f32x4 (&B)[RB], int kg) __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < RB; ++r) B[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, bv, bo[r] + kg * 1024, 0));
#pragma unroll
            for (int t = 0; t < LAY_TB; ++t) A[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, wv[t], kg * 64, 0));
        };
        auto mmk = [&](const f32x4 (&A)[LAY_TB], const f32x4 (&B)[RB]) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < LAY_TB; ++t)
#pragma unroll
                    for (int r = 0; r < RB; ++r) acc[r][t] = mfma16(A[t][j], B[r][j], acc[r][t]);
        };
        auto kloop = [&](int k0, int k1) __attribute__((always_inline)) {       // k-groups [k0, k1)
            if (k1 <= k0) return;
            f32x4 A0[LAY_TB], B0[RB], A1[LAY_TB], B1[RB];
            ldk(A0, B0, k0);
            int kg = k0;
            for (; kg + 2 <= k1; kg += 2) {
                ldk(A1, B1, kg + 1);
                __builtin_amdgcn_sched_barrier(0);
                mmk(A0, B0);
                ldk(A0, B0, kg + 2 < k1 ? kg + 2 : k1 - 1);
                __builtin_amdgcn_sched_barrier(0);
                mmk(A1, B1);
            }
            if (kg < k1) mmk(A0, B0);
        };
        if (SK) {
            // (underfilled, latency-bound.  An `unroll 8` here -- eight k-groups of loads in flight -- is refused by the optimizer: the
            // operand loads carry scheduling fences; the loop runs as written)
            for (int kg = k_lo; kg < k_hi; ++kg) { f32x4 A[LAY_TB], B[RB]; ldk(A, B, kg); mmk(A, B); }
            // partial tiles -> LDS; wave w finishes the (r, t) pairs with (r * LAY_TB + t) % 4 == w, summing the waves in fixed order
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int t = 0; t < LAY_TB; ++t) part[((wave * RB + r) * LAY_TB + t) * 64 + lane] = acc[r][t];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int t = 0; t < LAY_TB; ++t) {
                    if ((r * LAY_TB + t) % 4 != wave) continue;
                    f32x4 s4 = part[((0 * RB + r) * LAY_TB + t) * 64 + lane];
#pragma unroll
                    for (int w = 1; w < 4; ++w) { const f32x4 q = part[((w * RB + r) * LAY_TB + t) * 64 + lane]; s4[0] += q[0]; s4[1] += q[1]; s4[2] += q[2]; s4[3] += q[3]; }
                    acc[r][t] = s4;
                }
        } else {
            kloop(0, KG);
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            if (rt0 + r >= ntiles) continue;
#pragma unroll
            for (int t = 0; t < LAY_TB; ++t) {
                if (t0 + t >= MT) continue;
                if (SK && (r * LAY_TB + t) % 4 != wave) continue;
                f32x4 v;
                const int u0 = 16 * (t0 + t) + 4 * g;
                const bool whole = 16 * (t0 + t) + 16 <= n_units;          // wave-uniform: no padding, no ones slot in this tile
                // (the activation switch is taken once per tile, not per element: act is wave-uniform)
                if (MODE == 0) {
                    f32x4 z = acc[r][t];
                    switch (act) {
                        case TBNN_ACT_RELU:
#pragma unroll
                            for (int j = 0; j < 4; ++j) z[j] = fmaxf(z[j], 0.f);
                            break;
                        case TBNN_ACT_NONE: break;
                        default:
#pragma unroll
                            for (int j = 0; j < 4; ++j) z[j] = act_fwd(z[j], act);
                    }
                    if (whole) v = z;
                    else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = u0 + j < n_units ? z[j] : (u0 + j == ones_slot ? 1.f : 0.f);
                    }
                } else {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(aux + ((size_t)(rt0 + r) * auxT + t0 + t) * 256 + i16 * 16 + 4 * g);
                    f32x4 z = acc[r][t];
                    switch (act) {
                        case TBNN_ACT_RELU:
#pragma unroll
                            for (int j = 0; j < 4; ++j) z[j] = a[j] > 0.f ? z[j] : 0.f;
                            break;
                        case TBNN_ACT_NONE: break;
                        default:
#pragma unroll
                            for (int j = 0; j < 4; ++j) z[j] = z[j] * act_bwd(a[j], act);
                    }
                    if (whole) v = z;
                    else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = u0 + j < n_units ? z[j] : 0.f;
                    }
                }
                lay_block_store(outb + ((size_t)(rt0 + r) * MT + t0 + t) * 256 + i16 * 16 + 4 * g, v);
            }
        }
        if (SK) __syncthreads();            // `part` is free again
    }
}
// launch one GEMM: two row tiles per item when there is plenty of work, one otherwise; split K over the workgroup when even
// that leaves most of the 1024 SIMDs without a wave and the fan-in is long enough to pay for the LDS round
// two row tiles per wave from this many (row tile, tile group) items on (TBNN_LAY_RB2: A/B runs).  Round 5: 4096 -> 1024.  784 -> 100 at
// n = 12 k is 1,500 items: with one row tile per wave 52 us, with two 31 us -- a k-group's operand loads feed 32 MFMAs instead of 16 (the
// whole 784 -> 100 -> 100 -> 10 gradient 125 -> 105 us); four row tiles per wave were slower at every size tried (8 -> 300 -> 300 -> 1: 357 -> 390 us),
// and so was sharing a tile group's weights across the waves of a workgroup through LDS (one barrier per k-group: 105 -> 131 us)
static inline long lay_rb2_items() { static const long v = [] { const char* e = getenv("TBNN_LAY_RB2"); return e ? atol(e) : 1024L; }(); return v; }
template <int MODE, int LAY_TB>
static inline void lay_gemm_launch_tb(hipStream_t st, const float* img, int wpitch, const float* in, int KG, float* outb, int MT,
                                      const float* aux, int auxT, long ntiles, int act, int n_units, int ones_slot, int rev) {
    const int TG = (MT + LAY_TB - 1) / LAY_TB;
    const long items1 = ntiles * TG;
    // (TBNN_LAY_SK2=0: off.  784 -> 100 at n = 12 k: 31 -> 27 us; the 784 -> 100 -> 100 -> 10 gradient 100.6 -> 96.8 us)
    static const int sk2 = [] { const char* e = getenv("TBNN_LAY_SK2"); return e ? atoi(e) : 1; }();
    if (sk2 && items1 >= lay_rb2_items() && items1 < 4096 && KG >= 16) {
        // two row tiles per item AND the k-groups split over the workgroup's four waves: a long fan-in in front of few row tiles
        const long items = ((ntiles + 1) / 2) * TG;
        hipLaunchKernelGGL((k_lay_gemm<MODE, 2, true, LAY_TB>), dim3((int)std::max<long>(1, items)), dim3(256), 0, st, img, wpitch, in, KG, outb, MT, aux, auxT,
                           ntiles, act, n_units, ones_slot, rev);
    } else if (items1 >= lay_rb2_items()) {
        const long items = ((ntiles + 1) / 2) * TG;
        hipLaunchKernelGGL((k_lay_gemm<MODE, 2, false, LAY_TB>), dim3((int)std::min<long>((items + 3) / 4, 8192)), dim3(256), 0, st, img, wpitch, in, KG, outb, MT, aux, auxT,
                           ntiles, act, n_units, ones_slot, rev);
    } else if (items1 < 1024 && KG >= 8) {
        hipLaunchKernelGGL((k_lay_gemm<MODE, 1, true, LAY_TB>), dim3((int)std::max<long>(1, items1)), dim3(256), 0, st, img, wpitch, in, KG, outb, MT, aux, auxT,
                           ntiles, act, n_units, ones_slot, rev);
    } else {
        hipLaunchKernelGGL((k_lay_gemm<MODE, 1, false, LAY_TB>), dim3((int)std::max<long>(1, (items1 + 3) / 4)), dim3(256), 0, st, img, wpitch, in, KG, outb, MT, aux, auxT,
                           ntiles, act, n_units, ones_slot, rev);
    }
}

// output tiles per item: 4, or all of them when the layer has only 1 or 2 (a 20-unit layer would otherwise issue its MFMAs twice)
template <int MODE>
static inline void lay_gemm_launch(hipStream_t st, const float* img, int wpitch, const float* in, int KG, float* outb, int MT,
                                   const float* aux, int auxT, long ntiles, int act, int n_units, int ones_slot, int rev) {
    if (MT == 1) lay_gemm_launch_tb<MODE, 1>(st, img, wpitch, in, KG, outb, MT, aux, auxT, ntiles, act, n_units, ones_slot, rev);
    else if (MT == 2) lay_gemm_launch_tb<MODE, 2>(st, img, wpitch, in, KG, outb, MT, aux, auxT, ntiles, act, n_units, ones_slot, rev);
    else lay_gemm_launch_tb<MODE, 4>(st, img, wpitch, in, KG, outb, MT, aux, auxT, ntiles, act, n_units, ones_slot, rev);
}

// BernoulliLikelihood of one (row, output) element (likelihood.py:226-236: p clipped to [1e-8, 1 - 1e-7]; tfd.Bernoulli.log_prob =
// xlogy(y, p) + xlog1py(1 - y, -p)): adds the log-probability to stat, returns d/df.  Hardware log2 / reciprocal (about 1 ulp), as in the
// fused families' lik_delta (kernels_fast.hpp): the library logf / log1pf / IEEE divisions are some 300 instructions per element -- with ten
// outputs per row they were half of the last layer's launch (784 -> 100 -> 100 -> 10: k_lay_last 12.7 us)
__device__ __forceinline__ float lay_bernoulli(float fi, float y, double& stat) {
    const float p = fminf(fmaxf(fi, 1e-8f), 1.f - 1e-7f), q = 1.f - p;
    const bool inside = (fi > 1e-8f) && (fi < 1.f - 1e-7f);
    const float t1 = (y == 0.f) ? 0.f : y * __logf(p);
    const float t2 = (1.f - y == 0.f) ? 0.f : (1.f - y) * __logf(q);
    stat += (double)(t1 + t2);
    return inside ? (y * __builtin_amdgcn_rcpf(p) - (1.f - y) * __builtin_amdgcn_rcpf(q)) : 0.f;
}

// likelihood (restated as in kernels_generic.hpp): statistic (Gaussian: sum of squared residuals; Bernoulli: log-prob) and
// dz of the last layer = dL/df * act'(f).  f, dz: blocks [row tile][TMl][16][16]; one thread per data row.  Only the real
// (row, output) entries of dz are written: the padding was zeroed when the store was allocated and nothing else writes it.
__global__ __launch_bounds__(256) void k_lay_lik(NetDev nd, const float* __restrict__ eta, const float* __restrict__ f, const float* __restrict__ Y,
                                                  long n, int TMl, float* __restrict__ dz, double* __restrict__ pstat) {
    __shared__ double red[4];
    const float sigma = lik_sigma(nd, eta);
    const float inv_var = 1.f / (sigma * sigma);
    const int lact = nd.act[nd.nl - 1];
    double stat = 0.0;
    // one thread per (row, output) element: with many outputs a thread per ROW ran the library logarithms of all of them one after the
    // other (784 -> 100 -> 100 -> 10 at n = 12 k: 9.6 us for 120 k elements)
    const long nel = n * nd.d_out;
    for (long el = (long)blockIdx.x * 256 + threadIdx.x; el < nel; el += (long)gridDim.x * 256) {
        const long row = el / nd.d_out;
        {
            const int o = (int)(el - row * nd.d_out);
            const size_t e = (size_t)(row >> 4) * TMl * 256 + (row & 15) * 16 + (size_t)(o >> 4) * 256 + (o & 15);
            const float fi = f[e], y = Y[el];
            float da;
            if (nd.lik == TBNN_LIK_BERNOULLI) {
                da = lay_bernoulli(fi, y, stat);
            } else {
                const float res = y - fi;                                   // likelihood.py:88-94
                stat += (double)res * (double)res;
                da = res * inv_var;
            }
            dz[e] = da * act_bwd(fi, lact);
        }
    }
    const double tot = block_sum(stat, red);
    if (threadIdx.x == 0) pstat[blockIdx.x] = tot;
}

// dW_l[u][k] = sum over rows dz_l[row][u] [a_l, 1][row][k], every layer, over this workgroup's row tiles; output tiles in
// blocks of up to LAY_DBU x LAY_DBK (an operand block feeds LAY_DBK resp. LAY_DBU MFMAs), dealt out over the waves (and over grid.y); every
// entry of the slab is written by exactly one wave.
// The row loop of one item.  Every operand is a buffer load: the resource is rebuilt per row tile from a wave-uniform pointer
// (SALU), the lane's offset is one of four constant VGPRs (lane-linear read of a block + 256 B per k-step), the block inside the
// item an SGPR offset -- no vector instruction computes an address.  (Round 3 indexed global pointers: 2.1 VALU and 3.2 SALU
// instructions per MFMA, the MFMA pipe 52 % busy: on this chip the f32 MFMA and the VALU share their issue.)
template <int NU, int NK>
__device__ __forceinline__ void lay_dw_rows(const float* dz0, const float* a0, long TMs, long TKs, long cnt, int lane, f32x4 (&acc)[LAY_DBU][LAY_DBK]) {
    int vo[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) vo[s] = lane * 4 + 256 * s;
    // two operand sets: the next row tile's loads are issued BEFORE this one's MFMAs (the waves of a SIMD then need not cover a
    // whole memory round trip per row tile: 2048 waves ran at 155 us, 4096 at 123 us before this)
    auto ld = [&](float (&A)[NU][4], float (&B)[NK][4], long rt) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dz0 + rt * TMs), 0, NU * 1024, LAY_RSRC_FLAGS);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a0 + rt * TKs), 0, NK * 1024, LAY_RSRC_FLAGS);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int a = 0; a < NU; ++a) A[a][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, vo[s], 1024 * a, 0));
#pragma unroll
            for (int b = 0; b < NK; ++b) B[b][s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, vo[s], 1024 * b, 0));
        }
    };
    auto mma = [&](const float (&A)[NU][4], const float (&B)[NK][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int a = 0; a < NU; ++a)
#pragma unroll
                for (int b = 0; b < NK; ++b) acc[a][b] = mfma16(A[a][s], B[b][s], acc[a][b]);
    };
    float A0[NU][4], B0[NK][4], A1[NU][4], B1[NK][4];
    ld(A0, B0, 0);
    long rt = 0;
    for (; rt + 2 <= cnt; rt += 2) {
        ld(A1, B1, rt + 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(A0, B0);
        ld(A0, B0, rt + 2 < cnt ? rt + 2 : cnt - 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(A1, B1);
    }
    if (rt < cnt) mma(A0, B0);
}
// the block's size is wave-uniform: one instantiation per (nu, nk), so that no MFMA is tested and no operand block is loaded in vain
__device__ __forceinline__ void lay_dw_rows_any(int nu, int nk, const float* dz0, const float* a0, long TMs, long TKs, long cnt, int lane,
                                                f32x4 (&acc)[LAY_DBU][LAY_DBK]) {
#define LAY_DW_CASE(U, K) case (U) * 8 + (K): lay_dw_rows<U, K>(dz0, a0, TMs, TKs, cnt, lane, acc); break;
#define LAY_DW_ROW(U) LAY_DW_CASE(U, 1) LAY_DW_CASE(U, 2) LAY_DW_CASE(U, 3) LAY_DW_CASE(U, 4)
    switch (nu * 8 + nk) {
        LAY_DW_ROW(1) LAY_DW_ROW(2) LAY_DW_ROW(3) LAY_DW_ROW(4)
#if LAY_DBU == 8
        LAY_DW_ROW(5) LAY_DW_ROW(6) LAY_DW_ROW(7) LAY_DW_ROW(8)
#endif
        default: break;
    }
#undef LAY_DW_ROW
#undef LAY_DW_CASE
}
__global__ __launch_bounds__(256) void k_lay_dw(NetDev nd, LayPlan p, const float* __restrict__ store, float* __restrict__ slabs, int pitch) {
    const int lane = threadIdx.x & 63, i16 = lane & 15, g = lane >> 4, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long per = (p.ntiles + gridDim.x - 1) / gridDim.x;
    const long lo = (long)blockIdx.x * per, hi = lo + per < p.ntiles ? lo + per : p.ntiles;
    float* slab = slabs + (size_t)blockIdx.x * pitch;
    for (int item = wave + 4 * blockIdx.y; item < p.dw_items; item += 4 * gridDim.y) {
        int l = 0, rem = item;
        for (; l < p.nl; ++l) { const int c = ((p.TM[l] + LAY_DBU - 1) / LAY_DBU) * ((p.TK[l] + LAY_DBK - 1) / LAY_DBK); if (rem < c) break; rem -= c; }
        const int KB = (p.TK[l] + LAY_DBK - 1) / LAY_DBK;
        const int tu0 = LAY_DBU * (rem / KB), tk0 = LAY_DBK * (rem % KB);
        const int TMl = p.TM[l], TKl = p.TK[l];
        const int nu = TMl - tu0 < LAY_DBU ? TMl - tu0 : LAY_DBU, nk = TKl - tk0 < LAY_DBK ? TKl - tk0 : LAY_DBK;     // wave-uniform
        f32x4 acc[LAY_DBU][LAY_DBK];
#pragma unroll
        for (int a = 0; a < LAY_DBU; ++a)
#pragma unroll
            for (int b = 0; b < LAY_DBK; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (hi > lo) {
            const float* dz0 = store + p.dOff[l] + ((size_t)lo * TMl + tu0) * 256;
            const float* a0 = store + p.aOff[l] + ((size_t)lo * TKl + tk0) * 256;
            lay_dw_rows_any(nu, nk, dz0, a0, (long)TMl * 256, (long)TKl * 256, hi - lo, lane, acc);
        }
        // D[m][n]: lane (i, g) reg j = dW[unit 16 tu + 4 g + j][slot 16 tk + i]
        const int in = nd.in[l], out = nd.out[l];
#pragma unroll
        for (int a = 0; a < LAY_DBU; ++a)
#pragma unroll
            for (int b = 0; b < LAY_DBK; ++b) {
                if (a >= nu || b >= nk) continue;
                const int k = 16 * (tk0 + b) + i16;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int u = 16 * (tu0 + a) + 4 * g + j;
                    if (u < out) {
                        if (k < in) slab[nd.offW[l] + u * in + k] = acc[a][b][j];
                        else if (k == in) slab[nd.offB[l] + u] = acc[a][b][j];
                    }
                }
            }
    }
}

// The fused tail: one wave takes a row tile through layers l0 .. nl-1, the likelihood and the delta chain back to dz_{l0-1}
// (dz_0 when l0 = 0) with the activations in registers -- the D layout of one layer is the B-operand layout of the next, as in
// the fused families -- for layers of at most TT tiles.  It stores every a_l / dz_l it produces (k_lay_dw and the remaining
// backward GEMMs read them) and re-reads a_l for act' (just written: L2).  Register arrays are indexed at compile time only:
// the k-group / tile loops always run TT x TT, operand tiles past a layer's end are CLAMPED loads multiplied by zero
// activations, result tiles past the end are zeroed.  Replaces 2 (nl - l0) launches: 784 -> 20 -> 20 -> 1 runs 4 launches per
// gradient instead of 8.
template <int TT>
__global__ __launch_bounds__(256) void k_lay_tail(NetDev nd, LayPlan p, const float* __restrict__ img, const float* __restrict__ eta,
                                                  const float* __restrict__ Y, long n, float* __restrict__ store, double* __restrict__ pstat) {
    __shared__ double red[4];
    const int lane = threadIdx.x & 63, i16 = lane & 15, g = lane >> 4, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float sigma = lik_sigma(nd, eta);
    const float inv_var = 1.f / (sigma * sigma);
    const int nl = p.nl, l0 = p.l0, lact = nd.act[nl - 1], lane_off = i16 * 16 + 4 * g;
    double stat = 0.0;
    for (long rt = (long)blockIdx.x * 4 + wave; rt < p.ntiles; rt += (long)gridDim.x * 4) {
        f32x4 a[TT];
        {
            const int TKl = p.TK[l0];
            const float* ab = store + p.aOff[l0] + (size_t)rt * TKl * 256 + lane_off;
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(ab + (t < TKl ? t : TKl - 1) * 256);
                a[t] = t < TKl ? v : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        // ---- forward
        for (int l = l0; l < nl; ++l) {
            const int KG = p.TK[l], MT = p.TO[l], wp = 16 * KG, out = nd.out[l], act = nd.act[l];
            const int ones = l + 1 < nl ? out : -1;
            const float* w = img + p.wOff[l] + (size_t)i16 * wp + 4 * g;
            f32x4 A[TT][TT], acc[TT];
#pragma unroll
            for (int kg = 0; kg < TT; ++kg)
#pragma unroll
                for (int t = 0; t < TT; ++t)
                    A[kg][t] = *reinterpret_cast<const f32x4*>(w + (size_t)(16 * (t < MT ? t : MT - 1)) * wp + 16 * (kg < KG ? kg : KG - 1));
#pragma unroll
            for (int t = 0; t < TT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kg = 0; kg < TT; ++kg)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int t = 0; t < TT; ++t) acc[t] = mfma16(A[kg][t][j], a[kg][j], acc[t]);      // a[kg] == 0 for kg >= KG
            float* ob = store + p.aOff[l + 1] + (size_t)rt * MT * 256 + lane_off;
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int u = 16 * t + 4 * g + j;
                    v[j] = (t < MT && u < out) ? act_fwd(acc[t][j], act) : (u == ones ? 1.f : 0.f);
                }
                a[t] = v;
                if (t < MT) lay_block_store(ob + t * 256, v);
            }
        }
        // ---- likelihood: a = f (lane (row i16, g) holds outputs 16 t + 4 g + j)
        f32x4 dz[TT];
        {
            const long row = rt * 16 + i16;
#pragma unroll
            for (int t = 0; t < TT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int o = 16 * t + 4 * g + j;
                    float d = 0.f;
                    if (row < n && o < nd.d_out) {
                        const float fi = a[t][j], y = Y[row * nd.d_out + o];
                        float da;
                        if (nd.lik == TBNN_LIK_BERNOULLI) {
                            da = lay_bernoulli(fi, y, stat);
                        } else {
                            const float res = y - fi;                                  // likelihood.py:88-94
                            stat += (double)res * (double)res;
                            da = res * inv_var;
                        }
                        d = da * act_bwd(fi, lact);
                    }
                    dz[t][j] = d;
                }
        }
        // ---- delta chain: dz_l stored, dz_{l-1} = (W_l^T dz_l) * act'_{l-1}(a_l)
        const int lend = l0 > 1 ? l0 : 1;
        for (int l = nl - 1; l >= lend; --l) {
            const int KG = p.TM[l], MT = p.TM[l - 1], wp = 16 * KG, out = nd.out[l - 1], act = nd.act[l - 1], TKl = p.TK[l];
            float* db = store + p.dOff[l] + (size_t)rt * KG * 256 + lane_off;
#pragma unroll
            for (int t = 0; t < TT; ++t) if (t < KG) lay_block_store(db + t * 256, dz[t]);
            const float* w = img + p.tOff[l] + (size_t)i16 * wp + 4 * g;
            const float* ab = store + p.aOff[l] + (size_t)rt * TKl * 256 + lane_off;
            f32x4 A[TT][TT], acc[TT], aux[TT];
#pragma unroll
            for (int kg = 0; kg < TT; ++kg)
#pragma unroll
                for (int t = 0; t < TT; ++t)
                    A[kg][t] = *reinterpret_cast<const f32x4*>(w + (size_t)(16 * (t < MT ? t : MT - 1)) * wp + 16 * (kg < KG ? kg : KG - 1));
#pragma unroll
            for (int t = 0; t < TT; ++t) aux[t] = *reinterpret_cast<const f32x4*>(ab + (t < MT ? t : MT - 1) * 256);
#pragma unroll
            for (int t = 0; t < TT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kg = 0; kg < TT; ++kg)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int t = 0; t < TT; ++t) acc[t] = mfma16(A[kg][t][j], dz[kg][j], acc[t]);     // dz[kg] == 0 for kg >= KG
#pragma unroll
            for (int t = 0; t < TT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int u = 16 * t + 4 * g + j;
                    dz[t][j] = (t < MT && u < out) ? acc[t][j] * act_bwd(aux[t][j], act) : 0.f;
                }
        }
        {
            const int TMl = p.TM[lend - 1];
            float* db = store + p.dOff[lend - 1] + (size_t)rt * TMl * 256 + lane_off;
#pragma unroll
            for (int t = 0; t < TT; ++t) if (t < TMl) lay_block_store(db + t * 256, dz[t]);
        }
    }
    const double wtot = wave_sum_lane0(stat);
    if (lane == 0) red[wave] = wtot;
    __syncthreads();
    if (threadIdx.x == 0) pstat[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// The last layer alone (LayPlan::last): one wave takes a row tile through a_L -> f (one output tile, fan-in of up to 8 tiles), the
// likelihood, dz_L and dz_{L-1} = (W_L^T dz_L) * act'_{L-1}(a_L) -- a_L stays in registers for act'.  Stores f, dz_L, dz_{L-1}.
__global__ __launch_bounds__(256) void k_lay_last(NetDev nd, LayPlan p, const float* __restrict__ img, const float* __restrict__ eta,
                                                  const float* __restrict__ Y, long n, float* __restrict__ store, double* __restrict__ pstat) {
    constexpr int TK8 = 8;
    __shared__ double red[4];
    const int lane = threadIdx.x & 63, i16 = lane & 15, g = lane >> 4, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float sigma = lik_sigma(nd, eta);
    const float inv_var = 1.f / (sigma * sigma);
    const int L = p.nl - 1, lact = nd.act[L], lane_off = i16 * 16 + 4 * g;
    const int KG = p.TK[L], wp = 16 * KG, MTp = p.TM[L - 1], outp = nd.out[L - 1], actp = nd.act[L - 1];
    // the weights, once per wave: W_L rows (A operand of the forward tile) and W_L^T (A operand of the delta step; its k dimension is the
    // one output tile: pitch 16)
    f32x4 Af[TK8], Ab[TK8];
    {
        const float* w = img + p.wOff[L] + (size_t)i16 * wp + 4 * g;
        const float* wt = img + p.tOff[L] + (size_t)i16 * 16 + 4 * g;
#pragma unroll
        for (int t = 0; t < TK8; ++t) {
            Af[t] = *reinterpret_cast<const f32x4*>(w + 16 * (t < KG ? t : KG - 1));
            Ab[t] = *reinterpret_cast<const f32x4*>(wt + (size_t)(16 * (t < MTp ? t : MTp - 1)) * 16);
        }
    }
    double stat = 0.0;
    for (long rt = (long)blockIdx.x * 4 + wave; rt < p.ntiles; rt += (long)gridDim.x * 4) {
        f32x4 a[TK8];
        {
            const float* ab = store + p.aOff[L] + (size_t)rt * KG * 256 + lane_off;
#pragma unroll
            for (int t = 0; t < TK8; ++t) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(ab + (t < KG ? t : KG - 1) * 256);
                a[t] = t < KG ? v : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};          // two chains: even / odd k-groups
#pragma unroll
        for (int kg = 0; kg < TK8; kg += 2)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc0 = mfma16(Af[kg][j], a[kg][j], acc0);                       // a[kg] == 0 for kg >= KG
                acc1 = mfma16(Af[kg + 1][j], a[kg + 1][j], acc1);
            }
        f32x4 f, dz;
        const long row = rt * 16 + i16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int o = 4 * g + j;
            f[j] = o < nd.d_out ? act_fwd(acc0[j] + acc1[j], lact) : 0.f;
            float d = 0.f;
            if (row < n && o < nd.d_out) {
                const float fi = f[j], y = Y[row * nd.d_out + o];
                float da;
                if (nd.lik == TBNN_LIK_BERNOULLI) {
                    da = lay_bernoulli(fi, y, stat);
                } else {
                    const float res = y - fi;                                  // likelihood.py:88-94
                    stat += (double)res * (double)res;
                    da = res * inv_var;
                }
                d = da * act_bwd(fi, lact);
            }
            dz[j] = d;
        }
        lay_block_store(store + p.aOff[L + 1] + (size_t)rt * 256 + lane_off, f);
        lay_block_store(store + p.dOff[L] + (size_t)rt * 256 + lane_off, dz);
        float* db = store + p.dOff[L - 1] + (size_t)rt * MTp * 256 + lane_off;
#pragma unroll
        for (int t = 0; t < TK8; ++t) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = mfma16(Ab[t][j], dz[j], acc);
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int u = 16 * t + 4 * g + j;
                v[j] = (t < MTp && u < outp) ? acc[j] * act_bwd(a[t][j], actp) : 0.f;
            }
            if (t < MTp) lay_block_store(db + t * 256, v);
        }
    }
    const double wtot = wave_sum_lane0(stat);
    if (lane == 0) red[wave] = wtot;
    __syncthreads();
    if (threadIdx.x == 0) pstat[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// network output f blocks -> fout[d_out][n] (network.predict, network.py:141-171)
__global__ __launch_bounds__(256) void k_lay_unpack_f(const float* __restrict__ f, long n, long ntiles, int TMl, int d_out, float* __restrict__ fout) {
    const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
    for (long rt = blockIdx.x; rt < ntiles; rt += gridDim.x) {
        const long row = rt * 16 + r;
        for (int tt = 0; tt < TMl; ++tt) {
            const int o = 16 * tt + c;
            if (row < n && o < d_out) fout[(size_t)o * n + row] = f[((size_t)rt * TMl + tt) * 256 + threadIdx.x];
        }
    }
}

// the forward chain: a_0 (packed) -> ... -> f
// consecutive GEMMs walk the row tiles in alternating directions (TBNN_LAY_ALT=0: all forwards, as before round 5)
static inline int lay_alt() { static const int v = [] { const char* e = getenv("TBNN_LAY_ALT"); return e ? atoi(e) : 1; }(); return v; }
static inline void lay_forward_chain(const NetDev& nd, const LayPlan& p, hipStream_t st, const float* img, float* store) {
    for (int l = 0; l < nd.nl; ++l)
        lay_gemm_launch<0>(st, img + p.wOff[l], 16 * p.TK[l], store + p.aOff[l], p.TK[l], store + p.aOff[l + 1], p.TO[l], nullptr, 0, p.ntiles,
                           nd.act[l], nd.out[l], l + 1 == nd.nl ? -1 : nd.out[l], lay_alt() ? (l & 1) : 0);
}
// one gradient: forward chain, likelihood, delta chain, dW slabs (p.NS slabs of `pitch` floats; pstat[p.NP])
static inline int lay_launch(const NetDev& nd, const LayPlan& p, hipStream_t st, const float* img, const float* eta, const float* Y, long n,
                             float* store, float* slabs, int pitch, double* pstat) {
    const int L = nd.nl - 1;
    int lb;                                   // the backward GEMMs still to run: layers lb .. 1
    if (p.tail) {
        for (int l = 0; l < p.l0; ++l)
            lay_gemm_launch<0>(st, img + p.wOff[l], 16 * p.TK[l], store + p.aOff[l], p.TK[l], store + p.aOff[l + 1], p.TO[l], nullptr, 0, p.ntiles,
                               nd.act[l], nd.out[l], nd.out[l], lay_alt() ? (l & 1) : 0);
        if (p.TT == 2) hipLaunchKernelGGL(k_lay_tail<2>, dim3(p.GT), dim3(256), 0, st, nd, p, img, eta, Y, n, store, pstat);
        else hipLaunchKernelGGL(k_lay_tail<4>, dim3(p.GT), dim3(256), 0, st, nd, p, img, eta, Y, n, store, pstat);
        lb = (p.l0 > 1 ? p.l0 : 1) - 1;
    } else if (p.last) {
        for (int l = 0; l < L; ++l)
            lay_gemm_launch<0>(st, img + p.wOff[l], 16 * p.TK[l], store + p.aOff[l], p.TK[l], store + p.aOff[l + 1], p.TO[l], nullptr, 0, p.ntiles,
                               nd.act[l], nd.out[l], nd.out[l], lay_alt() ? (l & 1) : 0);
        hipLaunchKernelGGL(k_lay_last, dim3(p.GT), dim3(256), 0, st, nd, p, img, eta, Y, n, store, pstat);
        lb = L - 1;
    } else {
        lay_forward_chain(nd, p, st, img, store);
        hipLaunchKernelGGL(k_lay_lik, dim3(p.NLK), dim3(256), 0, st, nd, eta, (const float*)(store + p.aOff[nd.nl]), Y, n, p.TM[L], store + p.dOff[L], pstat);
        lb = L;
    }
    // (the forward chain's last GEMM walked in direction (nl - 1) & 1; the likelihood kernel is a plain forward sweep)
    for (int l = lb; l >= 1; --l)
        lay_gemm_launch<1>(st, img + p.tOff[l], 16 * p.TM[l], store + p.dOff[l], p.TM[l], store + p.dOff[l - 1], p.TM[l - 1], store + p.aOff[l], p.TK[l],
                           p.ntiles, nd.act[l - 1], nd.out[l - 1], -1, lay_alt() ? ((lb - l + 1) & 1) : 0);
    hipLaunchKernelGGL(k_lay_dw, dim3(p.NS, p.NY), dim3(256), 0, st, nd, p, (const float*)store, slabs, pitch);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
