// k_fwd_bwd_mid: the fused forward + likelihood + backward pass for MID-WIDTH networks (BASELINE configs[4]:
// 20 -> 100 -> 100 -> 2) -- hidden layers too wide for the narrow family's LDS plan (W and W^T images of every layer
// next to per-wave transposed images) but whose dW tiles still fit ONE wave's accumulator registers:
// sum over the MFMA layers of ceil(out/16) x ceil((in+1)/16) <= 63 tiles = 252 AccVGPRs.
//
// Round 1-2 ran these shapes on the wide path (kernels_wide.hpp): k_chain_wide stored a_l and delta_l of the middle
// layer to HBM (437 MB per gradient at configs[4]) for a second kernel, k_dw_wide, to contract over the rows.  Here
// the whole gradient stays on the chip, the narrow family's design at this width:
//   * one wave = one 16-row tile through forward, likelihood, delta chain AND every dW contraction; all dW
//     accumulators pinned to AccVGPRs for the whole row loop (mfma16_acc), chain MFMAs in VGPR form;
//   * weights resident in LDS.  A middle layer keeps ONE row-major image W_l[out slot][in slot] (pitch == 4 mod 8):
//     the forward pass reads its A operands as 16-B rows (lane (i, g): W[16t+i][16kg+4g .. +3]), the delta chain reads
//     W^T from the SAME image with strided 4-B loads (lane (i, g): W[16kg+4g+s][16u+i]; 4g x pitch == 16g mod 32:
//     conflict-free) -- no transposed copy, which is what makes two 100 x 100 layers' worth of operands fit;
//   * the dW operands (contraction over the 16 data rows, which sit on the lanes of the C/D layout) go through
//     per-wave LDS blocks [16 rows][16 slots]: written as they stand in the D layout (one 16-B store per lane and
//     tile), read back lane-linearly (float offset 64 s + lane = row 4s + g, slot i): both MFMA operands are plain
//     conflict-free 4-B reads; the bias gradient rides as a constant-1 slot of a_l;
//   * the <= 2-output last layer runs on the VALU (per-lane partial sums, reduced once per launch).
// No barrier inside the row loop, no HBM traffic but the rows themselves and one gradient slab per workgroup
// (k_update reduces the slabs, as for the narrow family: same launch signature, same FusedOps family).
//
// Reference math: layer.py:278 (W@a+b), activationFunctions.py:36/49/62, likelihood.py:88-94,226-236,
// BNN_functions.py:23-32; reverse mode SURVEY A12; the path: network.py:394-408.
#pragma once
#include <type_traits>
#include "kernels_fast.hpp"

#ifndef TBNN_SFOR_DEFINED
#define TBNN_SFOR_DEFINED
// compile-time loop: f(std::integral_constant<int, I>{}) for I in [I0, N)
template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}
#define SFOR_LAMBDA(name) [&](auto name##_) __attribute__((always_inline))
#define SFOR_VAL(name) decltype(name##_)::value
// y = sum over the 16 lanes of a lane group (same lane >> 4)
__device__ __forceinline__ float row16_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}
#endif

#define MID_WAVES 4
#define MID_THREADS 256
// pitch padding of the middle layers' row-major images (floats; == 4 mod 8 keeps the strided W^T reads conflict-free)
#ifndef MID_WPAD
#define MID_WPAD 4
#endif
// diagnostic build only (-DMID_STAMPS): shader-clock stamps of workgroup 0 / wave 0 during its SECOND tile
#ifdef MID_STAMPS
__device__ unsigned long long g_mid_stamps[64];
#define MSTAMP(k) do { __builtin_amdgcn_sched_barrier(0); if (mstamp_on) g_mid_stamps[k] = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define MSTAMP(k) do { } while (0)
#endif

template <class S>
struct MidCfg {
    static constexpr int NL = S::NL;
    static_assert(NL >= 3, "the mid-width path needs at least one middle layer");
    static constexpr int in(int l) { return S::D[l]; }
    static constexpr int out(int l) { return S::D[l + 1]; }
    static constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
    static constexpr int r4(int a) { return (a + 3) & ~3; }
    static constexpr int NM = NL - 2;                 // middle layers 1 .. NM
    static constexpr int LL = NL - 1;                 // last layer (VALU)
    static constexpr int d_in = in(0), d_out = out(LL);
    static_assert(d_out <= 2, "last layer runs on the VALU (<= 2 outputs)");
    static_assert(d_in <= 32, "layer-0 fan-in above 32 is not laid out");
    // a_l = input of layer l (l = 1..LL) in the padded slot order of kernels_fast.hpp (slot_of / unit_of / ones_slot)
    static constexpr int TR(int l) { return cdiv(in(l), 16); }          // register tiles (real units)
    static constexpr int TA(int l) { return cdiv(in(l) + 1, 16); }      // block tiles (with the ones slot)
    static constexpr int ksteps(int K, int kg) { int rem = K - 16 * kg; return rem >= 16 ? 4 : (rem <= 0 ? 0 : (rem + 3) / 4); }
    static constexpr int KG(int K) { return cdiv(K, 16); }
    static constexpr int maxT() { int m = 0; for (int l = 1; l <= LL; ++l) m = TR(l) > m ? TR(l) : m; return m; }
    static constexpr int MAXT = maxT();
    // layer 0
    static constexpr int KG0 = KG(d_in), NT0 = cdiv(d_in + 1, 16), MT0 = TR(1);
    // ---- dW accumulator tiles: layer 0: MT0 x NT0, middle layer l: TR(l+1) x TA(l)
    static constexpr int dwt(int l) { return l == 0 ? MT0 * NT0 : TR(l + 1) * TA(l); }
    static constexpr int dwoff(int l) { int o = 0; for (int m = 0; m < l; ++m) o += dwt(m); return o; }
    static constexpr int DW_TILES = dwoff(NM + 1);
    static_assert(DW_TILES <= 63, "dW accumulators exceed the AccVGPR file (252 registers)");
    // ---- weight image (LDS resident; k_update scatters theta into the HBM copy through image_map)
    //   W_0: MFMA A-operand granules [M tile][k group][lane (i, g)][s]: lane holds W_0[16t+i][16kg+4g+s]
    //   biases 0..NM: [16 TR(l+1)] in slot order;  W_LL: [d_out][16 TR(LL)], b_LL
    //   middle layer l: row-major [16 TR(l+1) out slots][LDM(l)]
    static constexpr int W0_OFF = 0;
    static constexpr int W0_FLOATS = MT0 * KG0 * 256;
    static constexpr int boff(int l) { int o = W0_OFF + W0_FLOATS; for (int m = 0; m < l; ++m) o += 16 * TR(m + 1); return o; }
    static constexpr int WLP = 16 * TR(LL);
    static constexpr int WL_OFF = boff(NM + 1);
    static constexpr int BL_OFF = WL_OFF + d_out * WLP;
    static constexpr int PERM_FLOATS = r4(BL_OFF + d_out);
    static constexpr int LDM(int l) { return 16 * TR(l) + MID_WPAD; }
    static constexpr int wmoff(int l) { int o = PERM_FLOATS; for (int m = 1; m < l; ++m) o += 16 * TR(m + 1) * LDM(m); return o; }
    static constexpr int IMG_FLOATS = r4(wmoff(NM + 1));
    // ---- per-wave blocks (256 floats = [16 rows][16 slots] each)
    static constexpr int XB_OFF = 0;                                            // x (+ ones slot): NT0 blocks
    static constexpr int aboff(int l) { int o = XB_OFF + NT0 * 256; for (int m = 1; m < l; ++m) o += TA(m) * 256; return o; }   // a_l, l = 1..NM
    static constexpr int DB_OFF = aboff(NM + 1);                                // delta_l (one layer at a time): MAXT blocks
    static constexpr int WAVE_FLOATS = DB_OFF + MAXT * 256;
    static constexpr int MIN_LDS = IMG_FLOATS + MID_WAVES * WAVE_FLOATS;
    // epilogue staging: tiles per pass with all 4 waves' copies resident
    static constexpr int EP_TILES = MIN_LDS / (MID_WAVES * 256) < DW_TILES ? MIN_LDS / (MID_WAVES * 256) : DW_TILES;
    static constexpr int LDS_FLOATS = MIN_LDS;
    // ---- parameters
    static constexpr int offW(int l) { int p = 0; for (int m = 0; m < l; ++m) p += in(m) * out(m) + out(m); return p; }
    static constexpr int P() { return offW(NL); }
};

template <class S>
struct MidLast {               // per-lane partial sums of the VALU last layer's dW / db
    using C = MidCfg<S>;
    f32x4 acc[C::d_out][C::TR(C::LL)];
    float accb[C::d_out];
};

// FWD: forward pass only (network.predict, network.py:141-171; predictor.py:132-155 with blockIdx.y = network):
// no likelihood, no delta chain, no dW; fout[d_out][n] receives the network output.
template <class S, bool FWD = false>
__global__ __launch_bounds__(MID_THREADS, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_fwd_bwd_mid(
    NetDev nd, const float* __restrict__ qimgs, long img_stride, const float* __restrict__ eta,
    const float* __restrict__ X, const float* __restrict__ Y, long n,
    float* __restrict__ slabs, int pitch, double* __restrict__ pstat, float* __restrict__ fouts, long out_stride)
{
    using C = MidCfg<S>;
    static_assert(C::LDS_FLOATS * 4 + 64 <= 160 * 1024, "LDS budget");
    static_assert(C::IMG_FLOATS % 4 == 0 && C::WAVE_FLOATS % 4 == 0, "16-B addressable sections");
    __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS];
    __shared__ double red[MID_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    constexpr int d_in = C::d_in, d_out = C::d_out, NM = C::NM, LL = C::LL;
    const float* qimg = qimgs + (size_t)blockIdx.y * img_stride;
    float* fout = FWD ? fouts + (size_t)blockIdx.y * out_stride : nullptr;
    const long ntiles = (n + 15) / 16;
    const long W = (long)gridDim.x * MID_WAVES;

    // rows of this wave's first tile: requested before the image loads (their latency hides under the prologue)
    float xn[C::KG0 * 4], yn[d_out];
    auto fetch = [&](long tile) {
        const long row = tile * 16 + i16;
        const bool ok = tile < ntiles && row < n;
#pragma unroll
        for (int k = 0; k < C::KG0 * 4; ++k) {
            const int u = unit_of(d_in, 16 * (k / 4) + 4 * g + (k % 4), false);
            xn[k] = (ok && u >= 0) ? X[row * d_in + u] : 0.f;
        }
#pragma unroll
        for (int o = 0; o < d_out; ++o) yn[o] = (!FWD && ok) ? Y[row * d_out + o] : 0.f;
    };
    long tile = (long)blockIdx.x * MID_WAVES + wave;
    fetch(tile);

    // ---- prologue: weight image -> LDS (all loads in flight, then the zero fill of the per-wave blocks, then the stores)
    float* wl = lds + C::IMG_FLOATS + wave * C::WAVE_FLOATS;
    {
        constexpr int N4 = C::IMG_FLOATS / 4, IT = (N4 + MID_THREADS - 1) / MID_THREADS;
        const float4* src = reinterpret_cast<const float4*>(qimg);
        float4* dst = reinterpret_cast<float4*>(lds);
        float4 v[IT];
#pragma unroll
        for (int k = 0; k < IT; ++k) { const int e = tid + k * MID_THREADS; v[k] = e < N4 ? src[e] : make_float4(0.f, 0.f, 0.f, 0.f); }
        __builtin_amdgcn_sched_barrier(0);
        float4* z = reinterpret_cast<float4*>(wl);
        for (int e = lane; e < C::WAVE_FLOATS / 4; e += 64) z[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < IT; ++k) { const int e = tid + k * MID_THREADS; if (e < N4) dst[e] = v[k]; }
    }
    __syncthreads();

    const float sigma = FWD ? 1.f : lik_sigma(nd, eta);
    const float inv_var = 1.f / (sigma * sigma);
    double stat = 0.0;
    f32x4 dW[FWD ? 1 : C::DW_TILES];
#pragma unroll
    for (int t = 0; t < (FWD ? 1 : C::DW_TILES); ++t) dW[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    MidLast<S> LR;
#pragma unroll
    for (int o = 0; o < d_out; ++o) {
        LR.accb[o] = 0.f;
#pragma unroll
        for (int t = 0; t < C::TR(LL); ++t) LR.acc[o][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    for (; tile < ntiles; tile += W) {
#ifdef MID_STAMPS
        const bool mstamp_on = blockIdx.x == 0 && tid == 0 && tile == W;
#endif
        MSTAMP(0);
        const bool rvalid = tile * 16 + i16 < n;
        float x[C::KG0 * 4], y[d_out];
#pragma unroll
        for (int k = 0; k < C::KG0 * 4; ++k) x[k] = xn[k];
#pragma unroll
        for (int o = 0; o < d_out; ++o) y[o] = yn[o];
        fetch(tile + W);
        if constexpr (!FWD) {
            // x blocks for dW_0 (slot order, the ones slot behind the last input unit)
#pragma unroll
            for (int kg = 0; kg < C::NT0; ++kg) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (kg < C::KG0) v = f32x4{x[4 * kg], x[4 * kg + 1], x[4 * kg + 2], x[4 * kg + 3]};
                constexpr int os = ones_slot(d_in);
                if (kg == os / 16 && g == (os % 16) / 4) v[os % 4] = 1.f;
                *reinterpret_cast<f32x4*>(wl + C::XB_OFF + kg * 256 + i16 * 16 + 4 * g) = v;
            }
        }

        // ---- layer 0
        f32x4 a[C::MAXT];                  // the current layer's input a_l, D layout: tile t reg j of lane (r, g) = slot 16t+4g+j of row r
        {
            constexpr int MT = C::MT0;
            f32x4 acc[MT];
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t] = *reinterpret_cast<const f32x4*>(lds + C::boff(0) + 16 * t + 4 * g);
#pragma unroll
            for (int kg = 0; kg < C::KG0; ++kg) {
                f32x4 A[MT];
#pragma unroll
                for (int t = 0; t < MT; ++t) A[t] = *reinterpret_cast<const f32x4*>(lds + C::W0_OFF + (t * C::KG0 + kg) * 256 + lane * 4);
#pragma unroll
                for (int s = 0; s < C::ksteps(d_in, kg); ++s)
#pragma unroll
                    for (int t = 0; t < MT; ++t) acc[t] = mfma16(A[t][s], x[4 * kg + s], acc[t]);
            }
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) a[t][r] = actc_fwd<S::act(0)>(acc[t][r]);
        }
        MSTAMP(1);

        // ---- middle layers, forward: a_l -> a_{l+1}; a_l (+ ones slot) goes to its blocks for dW_l
        sfor<1, NM + 1>(SFOR_LAMBDA(l) {
            constexpr int l = SFOR_VAL(l);
            if constexpr (!FWD) {
#pragma unroll
                for (int t = 0; t < C::TA(l); ++t) {
                    f32x4 v = t < C::TR(l) ? a[t] : f32x4{0.f, 0.f, 0.f, 0.f};
                    constexpr int os = ones_slot(C::in(l));
                    if (t == os / 16 && g == (os % 16) / 4) v[os % 4] = 1.f;
                    *reinterpret_cast<f32x4*>(wl + C::aboff(l) + t * 256 + i16 * 16 + 4 * g) = v;
                }
            }
            constexpr int MT = C::TR(l + 1), K = C::in(l);
            f32x4 acc[MT];
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t] = *reinterpret_cast<const f32x4*>(lds + C::boff(l) + 16 * t + 4 * g);
            const float* wrow = lds + C::wmoff(l) + i16 * C::LDM(l) + 4 * g;
#pragma unroll
            for (int kg = 0; kg < C::KG(K); ++kg) {
                f32x4 A[MT];
#pragma unroll
                for (int t = 0; t < MT; ++t) A[t] = load_ks(wrow + 16 * t * C::LDM(l) + 16 * kg, C::ksteps(K, kg));
#pragma unroll
                for (int s = 0; s < C::ksteps(K, kg); ++s)
#pragma unroll
                    for (int t = 0; t < MT; ++t) acc[t] = mfma16(A[t][s], a[kg][s], acc[t]);
            }
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) a[t][r] = actc_fwd<S::HACT>(acc[t][r]);
        });
        MSTAMP(2);

        // ---- last layer on the VALU: f_o = b_o + sum_u W[o][u] a_LL[u]
        constexpr int TP = C::TR(LL);
        float dzl[d_out];
        {
#pragma unroll
            for (int o = 0; o < d_out; ++o) {
                float p = 0.f;
#pragma unroll
                for (int t = 0; t < TP; ++t) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(lds + C::WL_OFF + o * C::WLP + 16 * t + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) p = fmaf(w[r], a[t][r], p);
                }
                p += __shfl_xor(p, 16, 64);
                p += __shfl_xor(p, 32, 64);
                const float fi = actc_fwd<S::LACT>(p + lds[C::BL_OFF + o]);
                if constexpr (FWD) {
                    if (rvalid && g == 0) fout[(size_t)o * n + tile * 16 + i16] = fi;      // [d_out][n]
                    dzl[o] = 0.f;
                } else {
                    dzl[o] = rvalid ? lik_delta<S>(fi, y[o], inv_var, g == 0, stat) : 0.f;
                    LR.accb[o] += dzl[o];
                }
            }
        }
        if constexpr (!FWD) {
        f32x4 dz[C::MAXT];
        {
#pragma unroll
            for (int t = 0; t < TP; ++t) {
                f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int o = 0; o < d_out; ++o) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(lds + C::WL_OFF + o * C::WLP + 16 * t + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        LR.acc[o][t][r] = fmaf(dzl[o], a[t][r], LR.acc[o][t][r]);
                        d[r] = fmaf(w[r], dzl[o], d[r]);
                    }
                }
                dz[t] = actc_bwd_mul4<S::HACT, true>(d, a[t]);
            }
        }
        MSTAMP(3);

        // ---- backward through the middle layers.  For layer l (NM .. 1), delta_l in `dz` (D layout):
        //   W(delta_l blocks) R(operands of dW_l) | delta chain: delta_{l-1} = (W_l^T delta_l) * act'(a_l) | dW_l MFMAs
        // (the operand round trip through LDS lands under the delta-chain MFMAs; the dW MFMAs run from registers)
        sfor<0, NM>(SFOR_LAMBDA(li) {
            constexpr int l = NM - SFOR_VAL(li);
            constexpr int TZ = C::TR(l + 1), TAl = C::TA(l), MU = C::TR(l), K = C::out(l);
            float* db = wl + C::DB_OFF;
            const float* ab = wl + C::aboff(l);
#pragma unroll
            for (int t = 0; t < TZ; ++t) *reinterpret_cast<f32x4*>(db + t * 256 + i16 * 16 + 4 * g) = dz[t];
            float Aop[TZ][4], Bop[TAl][4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int t = 0; t < TZ; ++t) Aop[t][s] = db[t * 256 + 64 * s + lane];
#pragma unroll
                for (int u = 0; u < TAl; ++u) Bop[u][s] = ab[u * 256 + 64 * s + lane];
            }
            // delta chain: A operand = W_l^T from the row-major image (lane (i, g): W[16kg+4g+s][16u+i])
            f32x4 acc[MU];
#pragma unroll
            for (int u = 0; u < MU; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float* wcol = lds + C::wmoff(l) + 4 * g * C::LDM(l) + i16;
#pragma unroll
            for (int kg = 0; kg < C::KG(K); ++kg) {
#pragma unroll
                for (int s = 0; s < C::ksteps(K, kg); ++s) {
                    float A[MU];
#pragma unroll
                    for (int u = 0; u < MU; ++u) A[u] = wcol[(16 * kg + s) * C::LDM(l) + 16 * u];
#pragma unroll
                    for (int u = 0; u < MU; ++u) acc[u] = mfma16(A[u], dz[kg][s], acc[u]);
                }
            }
            // act'(a_l) from the blocks the forward pass wrote (the ones slot multiplies an exact zero: padded W columns)
            f32x4 dzp[MU];
#pragma unroll
            for (int u = 0; u < MU; ++u) {
                const f32x4 al = *reinterpret_cast<const f32x4*>(ab + u * 256 + i16 * 16 + 4 * g);
                dzp[u] = actc_bwd_mul4<S::act(l - 1), false>(acc[u], al);
            }
            MSTAMP(4 + 2 * SFOR_VAL(li));
            // dW_l += delta_l^T [a_l, 1]
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int t = 0; t < TZ; ++t)
#pragma unroll
                    for (int u = 0; u < TAl; ++u) mfma16_acc<(TZ * TAl > 1)>(dW[C::dwoff(l) + t * TAl + u], Aop[t][s], Bop[u][s]);
            MSTAMP(5 + 2 * SFOR_VAL(li));
#pragma unroll
            for (int u = 0; u < C::MAXT; ++u) if (u < MU) dz[u] = dzp[u];
        });

        // ---- dW_0 += delta_0^T [x, 1]
        {
            float* db = wl + C::DB_OFF;
            const float* xb = wl + C::XB_OFF;
#pragma unroll
            for (int t = 0; t < C::MT0; ++t) *reinterpret_cast<f32x4*>(db + t * 256 + i16 * 16 + 4 * g) = dz[t];
            float Aop[C::MT0][4], Bop[C::NT0][4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int t = 0; t < C::MT0; ++t) Aop[t][s] = db[t * 256 + 64 * s + lane];
#pragma unroll
                for (int u = 0; u < C::NT0; ++u) Bop[u][s] = xb[u * 256 + 64 * s + lane];
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int t = 0; t < C::MT0; ++t)
#pragma unroll
                    for (int u = 0; u < C::NT0; ++u) mfma16_acc<(C::MT0 * C::NT0 > 1)>(dW[C::dwoff(0) + t * C::NT0 + u], Aop[t][s], Bop[u][s]);
        }
        MSTAMP(4 + 2 * NM);
        }   // !FWD
    }
    if constexpr (FWD) return;
    mfma_drain();

    // ---- epilogue: every wave stages its dW tiles [wave][tile][lane] (16 B per lane), wave t % 4 sums the 4 copies of
    // tile t in fixed order and writes the dense slab (theta order); EP_TILES per pass.  Deterministic.
    const double wtot = wave_sum(stat);
    if (lane == 0) red[wave] = wtot;
    float* slab = slabs + (size_t)blockIdx.x * pitch;
#pragma unroll
    for (int t0 = 0; t0 < C::DW_TILES; t0 += C::EP_TILES) {
        __syncthreads();                   // images / blocks (or the previous pass) are dead
        f32x4* mine = reinterpret_cast<f32x4*>(lds) + wave * (C::EP_TILES * 64);
#pragma unroll
        for (int t = t0; t < t0 + C::EP_TILES && t < C::DW_TILES; ++t) mine[(t - t0) * 64 + lane] = dW[t];
        __syncthreads();
        sfor<0, NM + 1>(SFOR_LAMBDA(l) {
            constexpr int l = SFOR_VAL(l);
            constexpr int inl = C::in(l), outl = C::out(l), MT = l == 0 ? C::MT0 : C::TR(l + 1), NT = l == 0 ? C::NT0 : C::TA(l);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int t = C::dwoff(l) + mt * NT + nt - t0;
                    if (t >= 0 && t < C::EP_TILES && (t & (MID_WAVES - 1)) == wave) {
                        const f32x4* src = reinterpret_cast<const f32x4*>(lds) + t * 64 + lane;
                        const f32x4 c0 = src[0], c1 = src[C::EP_TILES * 64], c2 = src[2 * C::EP_TILES * 64], c3 = src[3 * C::EP_TILES * 64];
                        // D layout: lane (n = i16, g) reg r = dW[out slot 16mt+4g+r][in slot 16nt+n]
                        const int col = unit_of(inl, 16 * nt + i16, true);          // inl: the ones pseudo-unit (bias column)
                        if (col >= 0) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int row = unit_of(outl, 16 * mt + 4 * g + r, false);
                                if (row >= 0)
                                    slab[C::offW(l) + (col < inl ? row * inl + col : inl * outl + row)] = (c0[r] + c1[r]) + (c2[r] + c3[r]);
                            }
                        }
                    }
                }
        });
    }
    {
        // last layer: reduce the per-row partials over the 16 lanes of a lane group, then over the 4 waves
        constexpr int TP = C::TR(LL), inL = C::in(LL);
        float* lb = lds;                               // [wave][o][slot], then [wave][o] biases
        static_assert(MID_WAVES * d_out * (16 * TP + 1) <= C::LDS_FLOATS, "last-layer staging does not fit");
        __syncthreads();
#pragma unroll
        for (int o = 0; o < d_out; ++o) {
#pragma unroll
            for (int t = 0; t < TP; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = row16_sum(LR.acc[o][t][r]);
                    if (i16 == 0) lb[(wave * d_out + o) * (16 * TP + 1) + 16 * t + 4 * g + r] = v;
                }
            const float vb = row16_sum(LR.accb[o]);
            if (lane == 0) lb[(wave * d_out + o) * (16 * TP + 1) + 16 * TP] = vb;
        }
        __syncthreads();
        for (int e = tid; e < d_out * (inL + 1); e += MID_THREADS) {
            const int o = e / (inL + 1), u = e - o * (inL + 1);
            const int s = u < inL ? slot_of(inL, u) : 16 * TP;
            float v[MID_WAVES];
#pragma unroll
            for (int w = 0; w < MID_WAVES; ++w) v[w] = lb[(w * d_out + o) * (16 * TP + 1) + s];
            slab[C::offW(LL) + (u < inL ? o * inL + u : inL * d_out + o)] = (v[0] + v[1]) + (v[2] + v[3]);
        }
    }
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < MID_WAVES; ++w) t += red[w];
        pstat[blockIdx.x] = t;
    }
}

// host: flat parameter index -> offset in the weight image (map[j]); no transposed copy (map[P + j] = -1)
template <class S>
static void mid_image_map(int* map) {
    using C = MidCfg<S>;
    const int P = C::P();
    for (int l = 0; l < C::NL; ++l) {
        const int in = C::in(l), out = C::out(l), ow = C::offW(l);
        for (int i = 0; i < out; ++i) {
            const int ri = slot_of(out, i);
            for (int k = 0; k < in; ++k) {
                const int ck = slot_of(in, k);
                int m0;
                if (l == 0) m0 = C::W0_OFF + (((ri / 16) * C::KG0 + ck / 16) * 64 + ((ck % 16) / 4) * 16 + ri % 16) * 4 + ck % 4;
                else if (l == C::LL) m0 = C::WL_OFF + i * C::WLP + ck;
                else m0 = C::wmoff(l) + ri * C::LDM(l) + ck;
                map[ow + i * in + k] = m0;
                map[P + ow + i * in + k] = -1;
            }
            map[ow + in * out + i] = l == C::LL ? C::BL_OFF + i : C::boff(l) + ri;
            map[P + ow + in * out + i] = -1;
        }
    }
}

// one workgroup (4 waves, 1 wave per SIMD) per CU; fewer when there are not enough tiles
static inline int mid_grid(long n) {
    const long ntiles = (n + 15) / 16, wgs = (ntiles + MID_WAVES - 1) / MID_WAVES;
    return (int)(wgs < 256 ? wgs : 256);
}
template <class S>
static inline int mid_launch_t(int grid, hipStream_t st, const NetDev& nd, const float* qimg, const float* eta, const float* X,
                               const float* Y, long n, float* slabs, int pitch, double* pstat) {
    hipLaunchKernelGGL((k_fwd_bwd_mid<S, false>), dim3(grid), dim3(MID_THREADS), 0, st, nd, qimg, 0L, eta, X, Y, n, slabs, pitch, pstat,
                       (float*)nullptr, 0L);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
// forward only: `nets` networks (grid.y; images img_stride floats apart), fouts[net][d_out][n]
template <class S>
static inline int mid_forward_t(int gx, int nets, hipStream_t st, const float* qimgs, long img_stride, const float* X, long n,
                                float* fouts, long out_stride) {
    NetDev nd{};
    hipLaunchKernelGGL((k_fwd_bwd_mid<S, true>), dim3(gx, nets), dim3(MID_THREADS), 0, st, nd, qimgs, img_stride, (const float*)nullptr, X,
                       (const float*)nullptr, n, (float*)nullptr, 0, (double*)nullptr, fouts, out_stride);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
