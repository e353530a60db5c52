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

// sum over the 4 lane groups (same lane & 15), every lane gets it: the gfx950 row-swap instructions, VALU only (two
// ds_bpermute round trips would put ~2 x 100 cycles of LDS latency on the last layer's dependency chain)
__device__ __forceinline__ float lane_group_sum(float p) {
    const unsigned a = __float_as_uint(p);
    const auto r = __builtin_amdgcn_permlane32_swap(a, a, false, false);   // lanes l and l ^ 32
    const float s = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    const unsigned b = __float_as_uint(s);
    const auto q = __builtin_amdgcn_permlane16_swap(b, b, false, false);   // rows r and r ^ 1
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

// Order of issue inside a k-group is fixed by hand with scheduling fences (nothing crosses __builtin_amdgcn_sched_barrier(0)):
// the LDS reads of the NEXT group's operands go out one per MFMA under the first MFMAs of this group, a whole group ahead
// of their use.  Left to itself the compiler requests an operand two MFMAs before its first use and stalls on the LDS
// round trip in every k-group (measured: 18-20 cycles of tile time per LDS instruction); sched_group_barrier pipelines made
// it serialise the accumulators (one tile's whole K loop at a time, a dependent MFMA chain).
#define MID_FENCE() __builtin_amdgcn_sched_barrier(0)
// Placement of NE LDS operations under the NMF MFMAs of a dW phase (k-step-major MFMA order): the first N1 of them are the
// k-steps 2, 3 operands, which must be on their way before the MFMAs of k-step 2 read their registers -- they ride under the
// MFMAs of k-step 0 (slots [0, NMF / 4)); the others follow in slots [NMF / 4, NMF).  ops_before(j) = operations in slots < j.
__host__ __device__ constexpr int mid_op_slot(int e, int N1, int NE, int NMF) {
    const int Q = NMF / 4;
    return e < N1 ? (e * Q) / N1 : Q + ((e - N1) * (NMF - Q)) / (NE - N1 > 0 ? NE - N1 : 1);
}
__host__ __device__ constexpr int mid_ops_before(int j, int N1, int NE, int NMF) {
    int c = 0;
    for (int e = 0; e < NE; ++e) c += mid_op_slot(e, N1, NE, NMF) < j ? 1 : 0;
    return c;
}

#define MID_WAVES 4
#define MID_THREADS 256
// pitch padding of the middle layers' row-major images (floats; == 4 mod 8 keeps the strided W^T reads conflict-free)
#ifndef MID_WPAD
#define MID_WPAD 4
#endif
// layer-0 fan-in: the rows of a tile and the next tile's prefetched copy live in registers (2 x 4 x ceil(d_in / 16) VGPRs);
// above this the tall family (kernels_tall.hpp) splits the fan-in over the waves of a workgroup
#ifndef MID_MAX_FANIN
#define MID_MAX_FANIN 128
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
    static constexpr int LL = NL - 1;                 // last layer
    static constexpr int d_in = in(0), d_out = out(LL);
    // <= 2 outputs: the last layer runs on the VALU (16 FMAs per output instead of a padded MFMA tile).  3 .. 16 outputs (round 6;
    // network.add takes any stack, tensorBNN/network.py:173-191, and BernoulliLikelihood sums over [d_out, n], likelihood.py:226-236): the
    // last layer is one more MFMA layer -- ONE output tile, its activation the network's last one -- and the likelihood reads the tile
    static constexpr bool VL = d_out <= 2;
    static constexpr int NM = VL ? NL - 2 : NL - 1;   // MFMA layers behind layer 0: 1 .. NM
    static_assert(d_out <= 16, "the last layer is one output tile at most");
    static_assert(d_in <= MID_MAX_FANIN, "layer-0 fan-in: x and its prefetch copy live in registers");
    // a_l = input of layer l (l = 1..LL) in the padded slot order of kernels_fast.hpp (slot_of / unit_of / ones_slot)
    static constexpr int TR(int l) { return cdiv(in(l), 16); }          // register tiles (real units)
    static constexpr int TA(int l) { return cdiv(in(l) + 1, 16); }      // block tiles (with the ones slot)
    static constexpr int ksteps(int K, int kg) { int rem = K - 16 * kg; return rem >= 16 ? 4 : (rem <= 0 ? 0 : (rem + 3) / 4); }
    static constexpr int KG(int K) { return cdiv(K, 16); }
    // Fringe (round 6; VERDICT rounds 3-5, configs[4]: 100 = 6 x 16 + 4 units pay for 112): a layer side of 16 T + F units with 1 <= F <= 4 runs its last
    // tile on the 16-block v_mfma_f32_4x4x1 -- 8 cycles per k-step instead of 32 -- where the operands allow it without new LDS traffic:
    //   fr_out(l): layer l's OUTPUT units in the forward pass (A operand: the image rows of the F units, the same 16-byte reads as a full tile's);
    //   fr_in(l):  layer l's INPUT units in the delta chain (A operand: the image columns of the F units, the same strided 4-byte reads).
    // The D layout of the previous layer is the 4x4x1 form's B operand as it stands (block = (k phase g, row quad), as the narrow family uses it);
    // the four k phases are summed with the row-swap instructions (gsum) and unit 16 T + e lands in register 0 of lane group e: its slot (slot_of).
    // Worth it when the k loop is long (>= 8 k-steps: the sums cost ~20 VALU instructions).  dW's fringe strips are not built.
#ifndef MID_FRINGE
#define MID_FRINGE 1
#endif
    static constexpr bool fr_units(int U) { return MID_FRINGE && U > 16 && U % 16 >= 1 && U % 16 <= 4; }
    static constexpr int total_ksteps(int K) { return (K + 3) / 4; }
    static constexpr bool fr_out(int l) { return l >= 1 && fr_units(out(l)) && total_ksteps(in(l)) >= 8; }
    static constexpr bool fr_in(int l) { return l >= 1 && fr_units(in(l)) && total_ksteps(out(l)) >= 8; }
    static constexpr int maxT() { int m = 0; for (int l = 1; l <= (VL ? LL : NL); ++l) m = TR(l) > m ? TR(l) : m; return m; }
    static constexpr int MAXT = maxT();
    static constexpr int maxTA() { int m = cdiv(in(0) + 1, 16); for (int l = 1; l <= NM; ++l) m = TA(l) > m ? TA(l) : m; return m; }
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
    static constexpr int WLP = VL ? 16 * TR(LL) : 0;                    // (the VALU last layer's own image; an MFMA last layer is a middle layer's)
    static constexpr int WL_OFF = boff(NM + 1);
    static constexpr int BL_OFF = WL_OFF + d_out * WLP;
    static constexpr int PERM_FLOATS = r4(BL_OFF + (VL ? d_out : 0));
    static constexpr int LDM(int l) { return 16 * TR(l) + MID_WPAD; }
    static constexpr int wmoff(int l) { int o = PERM_FLOATS; for (int m = 1; m < l; ++m) o += 16 * TR(m + 1) * LDM(m); return o; }
    static constexpr int IMG_FLOATS = r4(wmoff(NM + 1));
    // ---- per-wave blocks (256 floats = [16 rows][16 slots] each)
    static constexpr int XB_OFF = 0;                                            // x (+ ones slot): NT0 blocks
    static constexpr int aboff(int l) { int o = XB_OFF + NT0 * 256; for (int m = 1; m < l; ++m) o += TA(m) * 256; return o; }   // a_l, l = 1..NM
    // delta_l blocks: two regions, layer parity selects one (delta_{l-1} is written while the operands of dW_l are still
    // being read from delta_l's region)
    static constexpr int DB_OFF = aboff(NM + 1);
    static constexpr int dboff(int l) { return DB_OFF + (l & 1) * MAXT * 256; }
    static constexpr int WAVE_FLOATS = DB_OFF + 2 * MAXT * 256;
    static constexpr int MIN_LDS = IMG_FLOATS + MID_WAVES * WAVE_FLOATS;
    // epilogue staging: tiles per pass with all 4 waves' copies resident
    static constexpr int EP_TILES = MIN_LDS / (MID_WAVES * 256) < DW_TILES ? MIN_LDS / (MID_WAVES * 256) : DW_TILES;
    static constexpr int LDS_FLOATS = MIN_LDS;
    // ---- parameters
    static constexpr int offW(int l) { int p = 0; for (int m = 0; m < l; ++m) p += in(m) * out(m) + out(m); return p; }
    static constexpr int P() { return offW(NL); }
};

template <class S>
struct MidLast {               // per-lane partial sums of the VALU last layer's dW / db (nothing when the last layer is an MFMA layer)
    using C = MidCfg<S>;
    static constexpr int NO = C::VL ? C::d_out : 1, NT = C::VL ? C::TR(C::LL) : 1;
    f32x4 acc[NO][NT];
    float accb[NO];
};

// FWD: forward pass only (network.predict, network.py:141-171; predictor.py:132-155 with blockIdx.y = network):
// no likelihood, no delta chain, no dW; fout[d_out][n] receives the network output.
template <class S, bool FWD = false>
__global__ __launch_bounds__(MID_THREADS, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_fwd_bwd_mid(
    NetDev nd, const float* __restrict__ qimgs, long img_stride, const float* __restrict__ eta,
    const float* __restrict__ X, const float* __restrict__ Y, long n,
    float* __restrict__ slabs, int pitch, double* __restrict__ pstat, float* __restrict__ fouts, long out_stride, ChainStride cs)
{
    using C = MidCfg<S>;
    // gridDim.y: networks of an ensemble (FWD) or chains of a multi-chain handle (tbnn_create_multi; cs.img == img_stride then)
    if constexpr (!FWD) {
        if (chain_done(cs.ctl, cs.t, blockIdx.y)) return;       // a chain past its own L (per-chain step control)
        eta += (size_t)blockIdx.y * cs.eta; slabs += (size_t)blockIdx.y * cs.slab; pstat += (size_t)blockIdx.y * PSTAT_CAP;
    }
    // the opaque packed instructions (pkfma*: no hazard handling by the compiler) may read hidden activations only when those
    // come out of a plain VALU instruction: relu (v_max), tanh / elu (a select).  A raw MFMA result (no activation) or a
    // transcendental result (sigmoid: v_rcp, exp: v_exp) would need wait states nobody inserts
    constexpr int ALAST = S::act(C::LL - 1);          // the activation in front of the last layer
    constexpr bool PKA = ALAST == TBNN_ACT_RELU || ALAST == TBNN_ACT_TANH || ALAST == TBNN_ACT_ELU;
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
    // targets of a row: <= 2 outputs: y[o] in every lane group; else the D layout of the output tile (lane (row, g): slots 4g .. 4g + 3)
    constexpr int YN = C::VL ? d_out : 4;
    constexpr int LRO = MidLast<S>::NO, LRT = MidLast<S>::NT;
    float xn[C::KG0 * 4], yn[YN];
    auto fetch = [&](long tile) {
        const long row = tile * 16 + i16;
        const bool ok = tile < ntiles && row < n;
#pragma unroll
        for (int k = 0; k < C::KG0 * 4; ++k) {
            const int u = unit_of(d_in, 16 * (k / 4) + 4 * g + (k % 4), false);
            xn[k] = (ok && u >= 0) ? X[row * d_in + u] : 0.f;
        }
#pragma unroll
        for (int o = 0; o < YN; ++o) {
            const int u = C::VL ? o : unit_of(d_out, 4 * g + o, false);
            yn[o] = (!FWD && ok && u >= 0) ? Y[row * d_out + u] : 0.f;
        }
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
    for (int o = 0; o < LRO; ++o) {
        LR.accb[o] = 0.f;
#pragma unroll
        for (int t = 0; t < LRT; ++t) LR.acc[o][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // operand registers that are requested ahead of their layer (and, for layer 0, ahead of their tile)
    f32x4 Apre[C::MAXT], Bpre[C::MAXT];                    // first A group and bias tiles of the next MFMA layer
    f32x4 wL[LRO][LRT];                                    // VALU last layer's weights, slot order: lane (r, g) holds slots 16t+4g..+3
    // slot `slot` of `nslot`: the loads are dealt out over the MFMA slots of the k-group they are issued under
    auto preload_mid = [&](auto l_, int slot, int nslot) __attribute__((always_inline)) {
        constexpr int l = decltype(l_)::value;
        const float* wrow = lds + C::wmoff(l) + i16 * C::LDM(l) + 4 * g;
        const float* frow = lds + C::wmoff(l) + (4 * (lane & 3)) * C::LDM(l) + 4 * g;       // fringe tile: lane (k phase g, row quad, i): unit 16 T + i = slot 16 T + 4 i
#pragma unroll
        for (int t = 0; t < C::TR(l + 1); ++t)
            if (t % nslot == slot) {
                Bpre[t] = *reinterpret_cast<const f32x4*>(lds + C::boff(l) + 16 * t + 4 * g);
                Apre[t] = load_ks(((C::fr_out(l) && t == C::TR(l + 1) - 1) ? frow : wrow) + 16 * t * C::LDM(l), C::ksteps(C::in(l), 0));
            }
    };
    auto preload_l0 = [&](int slot, int nslot) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < C::MT0; ++t)
            if (t % nslot == slot) {
                Bpre[t] = *reinterpret_cast<const f32x4*>(lds + C::boff(0) + 16 * t + 4 * g);
                Apre[t] = *reinterpret_cast<const f32x4*>(lds + C::W0_OFF + (t * C::KG0) * 256 + lane * 4);
            }
    };
    auto preload_last = [&](int slot, int nslot) __attribute__((always_inline)) {
        if constexpr (C::VL) {
#pragma unroll
            for (int o = 0; o < d_out; ++o)
#pragma unroll
                for (int t = 0; t < C::TR(LL); ++t)
                    if ((o * C::TR(LL) + t) % nslot == slot) wL[o][t] = *reinterpret_cast<const f32x4*>(lds + C::WL_OFF + o * C::WLP + 16 * t + 4 * g);
        }
    };
    if constexpr (!FWD) preload_l0(0, 1);

    for (; tile < ntiles; tile += W) {
#ifdef MID_STAMPS
        const bool mstamp_on = blockIdx.x == 0 && tid == 0 && tile == W;
#endif
        MSTAMP(0);
        const bool rvalid = tile * 16 + i16 < n;
        float x[C::KG0 * 4], y[YN];
#pragma unroll
        for (int k = 0; k < C::KG0 * 4; ++k) x[k] = xn[k];
#pragma unroll
        for (int o = 0; o < YN; ++o) y[o] = yn[o];
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

        // ---- forward.  LDS operand traffic is placed by hand (sched_group_barrier): the A operands of k-group kg+1 are
        // requested under the first MFMAs of k-group kg, one read per MFMA -- left to itself the compiler issues a read two
        // MFMAs ahead of its use and stalls on it -- and the first group + bias tiles of the NEXT layer (the last layer's
        // weights after the last middle layer) under the last k-group of this one.
        f32x4 a[C::MAXT];                  // the current layer's input a_l, D layout: tile t reg j of lane (r, g) = slot 16t+4g+j of row r
        // ---- layer 0
        {
            constexpr int MT = C::MT0;
            f32x4 acc[MT], An[MT];
            if constexpr (FWD) preload_l0(0, 1);
#pragma unroll
            for (int t = 0; t < MT; ++t) { acc[t] = Bpre[t]; An[t] = Apre[t]; }
            sfor<0, C::KG0>(SFOR_LAMBDA(kg) {
                constexpr int kg = SFOR_VAL(kg);
                f32x4 A[MT];
#pragma unroll
                for (int t = 0; t < MT; ++t) A[t] = An[t];
                MID_FENCE();
#pragma unroll
                for (int s = 0; s < C::ksteps(d_in, kg); ++s) {
#pragma unroll
                    for (int t = 0; t < MT; ++t) {
                        acc[t] = mfma16(A[t][s], x[4 * kg + s], acc[t]);
                        if (s == 0) {
                            if constexpr (kg + 1 < C::KG0) An[t] = *reinterpret_cast<const f32x4*>(lds + C::W0_OFF + (t * C::KG0 + kg + 1) * 256 + lane * 4);
                            else preload_mid(std::integral_constant<int, 1>{}, t, MT);
                            MID_FENCE();
                        }
                    }
                    MID_FENCE();
                }
            });
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
            constexpr int MT = C::TR(l + 1), K = C::in(l), KGn = C::KG(K);
            constexpr bool FR = C::fr_out(l);
            constexpr int MTF = FR ? MT - 1 : MT;                 // full tiles; tile MTF: the fringe units on the 4x4x1 form
            f32x4 acc[MT], An[MT];
#pragma unroll
            for (int t = 0; t < MT; ++t) { acc[t] = (FR && t == MTF) ? f32x4{0.f, 0.f, 0.f, 0.f} : Bpre[t]; An[t] = Apre[t]; }
            const float fbias = FR ? Bpre[FR ? MTF : 0][0] : 0.f;  // lane group e: the bias of unit 16 MTF + e (slot 16 MTF + 4 e)
            const float* wrow = lds + C::wmoff(l) + i16 * C::LDM(l) + 4 * g;
            const float* frow = lds + C::wmoff(l) + (4 * (lane & 3)) * C::LDM(l) + 4 * g;
            sfor<0, KGn>(SFOR_LAMBDA(kg) {
                constexpr int kg = SFOR_VAL(kg);
                f32x4 A[MT];
#pragma unroll
                for (int t = 0; t < MT; ++t) A[t] = An[t];
                MID_FENCE();
#pragma unroll
                for (int s = 0; s < C::ksteps(K, kg); ++s) {
#pragma unroll
                    for (int t = 0; t < MT; ++t) {
                        if (FR && t == MTF) acc[t] = mfma4(A[t][s], a[kg][s], acc[t]);
                        else acc[t] = mfma16(A[t][s], a[kg][s], acc[t]);
                        if (s == 0) {
                            if constexpr (kg + 1 < KGn) An[t] = load_ks(((FR && t == MTF) ? frow : wrow) + 16 * t * C::LDM(l) + 16 * (kg + 1), C::ksteps(K, kg + 1));
                            else if constexpr (l < NM) preload_mid(std::integral_constant<int, (l < NM ? l + 1 : l)>{}, t, MT);
                            else preload_last(t, MT);
                            MID_FENCE();
                        }
                    }
                    MID_FENCE();
                }
            });
            if constexpr (FR) {
                // register i of lane (k phase g, row): unit 16 MTF + i over the k slots of phase g; the four phases summed, unit 16 MTF + g kept
                float z[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) z[i] = gsum(acc[MTF][i]);
                const float sel = g == 0 ? z[0] : (g == 1 ? z[1] : (g == 2 ? z[2] : z[3]));
                acc[MTF] = f32x4{sel + fbias, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) a[t][r] = (FR && t == MTF && r > 0) ? 0.f : actc_fwd<S::act(l)>(acc[t][r]);
        });
        MSTAMP(2);

        // ---- last layer on the VALU: f_o = b_o + sum_u W[o][u] a_LL[u] (packed FMAs kept packed: pkfma*, kernels_fast.hpp)
        constexpr int TP = LRT;
        float dzl[LRO];
        if constexpr (!C::VL && FWD) {
            // MFMA last layer: a[0] is the output tile (lane (row i16, g): slots 4g .. 4g + 3)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int u = unit_of(d_out, 4 * g + r, false);
                if (rvalid && u >= 0) fout[(size_t)u * n + tile * 16 + i16] = a[0][r];      // [d_out][n]
            }
        }
        if constexpr (C::VL) {
#pragma unroll
            for (int o = 0; o < d_out; ++o) {
                f32x2 p2 = {0.f, 0.f};
#pragma unroll
                for (int t = 0; t < TP; ++t) {
                    const f32x4 w = wL[o][t];
                    if constexpr (PKA) {
                        p2 = pkfma(f32x2{w[0], w[1]}, f32x2{a[t][0], a[t][1]}, p2);
                        p2 = pkfma(f32x2{w[2], w[3]}, f32x2{a[t][2], a[t][3]}, p2);
                    } else {                           // compiler-visible arithmetic
                        p2 = f32x2{w[0], w[1]} * f32x2{a[t][0], a[t][1]} + p2;
                        p2 = f32x2{w[2], w[3]} * f32x2{a[t][2], a[t][3]} + p2;
                    }
                }
                MSTAMP(16 + 3 * o);
                const float fsum = lane_group_sum(p2[0] + p2[1]);
                MSTAMP(17 + 3 * o);
                const float fi = actc_fwd<S::LACT>(fsum + lds[C::BL_OFF + o]);
                if constexpr (FWD) {
                    if (rvalid && g == 0) fout[(size_t)o * n + tile * 16 + i16] = fi;      // [d_out][n]
                    dzl[o] = 0.f;
                } else {
                    dzl[o] = rvalid ? lik_delta<S>(fi, y[o], inv_var, g == 0, stat) : 0.f;
                    LR.accb[o] += dzl[o];
                }
                MSTAMP(18 + 3 * o);
            }
        }
        if constexpr (!FWD) {
        f32x4 dz[C::MAXT];
        if constexpr (!C::VL) {
            // likelihood on the output tile: delta_LL (w.r.t. the pre-activation) in the D layout, every (row, output) element once
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int u = unit_of(d_out, 4 * g + r, false);
                dz[0][r] = (rvalid && u >= 0) ? lik_delta<S>(a[0][r], y[r], inv_var, true, stat) : 0.f;
            }
        } else {
            // dW_LL partial sums and delta_{LL-1} = (W_LL^T dz_LL) * act'(a_LL)
            const f32x2 dd = {dzl[0], dzl[d_out - 1]};
#pragma unroll
            for (int t = 0; t < TP; ++t) {
                f32x2 d01 = {0.f, 0.f}, d23 = {0.f, 0.f};
                sfor<0, d_out>(SFOR_LAMBDA(o) {
                    constexpr int o = SFOR_VAL(o);
                    const f32x4 w = wL[o][t];
                    if constexpr (PKA) {
                        const f32x2 s01 = pkfma_bc<o>(f32x2{a[t][0], a[t][1]}, dd, f32x2{LR.acc[o][t][0], LR.acc[o][t][1]});
                        const f32x2 s23 = pkfma_bc<o>(f32x2{a[t][2], a[t][3]}, dd, f32x2{LR.acc[o][t][2], LR.acc[o][t][3]});
                        LR.acc[o][t] = f32x4{s01[0], s01[1], s23[0], s23[1]};
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) LR.acc[o][t][r] = fmaf(dzl[o], a[t][r], LR.acc[o][t][r]);
                    }
                    if constexpr (o == 0) { d01 = pkmul_bc<0>(f32x2{w[0], w[1]}, dd); d23 = pkmul_bc<0>(f32x2{w[2], w[3]}, dd); }
                    else { d01 = pkfma_bc<o>(f32x2{w[0], w[1]}, dd, d01); d23 = pkfma_bc<o>(f32x2{w[2], w[3]}, dd, d23); }
                });
                if constexpr (ALAST != TBNN_ACT_NONE) dz[t] = actc_bwd_mul4<ALAST, true>(f32x4{d01[0], d01[1], d23[0], d23[1]}, a[t]);
                else dz[t] = f32x4{d01[0], d01[1], d23[0], d23[1]};
            }
        }
        MSTAMP(3);

        // ---- backward through the middle layers.  Every LDS access is placed by hand under MFMAs (one or two per MFMA slot,
        // a scheduling fence after each slot):
        //   B(l)  delta chain through W_l: delta_{l-1} = (W_l^T delta_l) * act'(a_l).  A operands straight from the row-major
        //         image, the next k-group's under this one's first MFMAs.  B(NM) also parks delta_NM in its blocks (first
        //         k-group); the last k-groups request the k-steps 0, 1 operands of dW_l and a_l (for act').
        //   D(l)  dW_l += delta_l^T [a_l, 1] from registers (AccVGPR accumulators, asm MFMAs: program order is issue order).
        //         Under them: the k-steps 2, 3 operands of dW_l, the parking of delta_{l-1} (the other block region), then
        //         what the next phase starts from: the first k-group of B(l-1), or the k-steps 0, 1 operands of dW_0.
        //   D(0)  dW_0 += delta_0^T [x, 1]; under it its k-steps 2, 3 operands, then the first A group + bias tiles of the
        //         NEXT tile's layer 0.
        constexpr int MAXB = C::maxTA();
        float Aop[C::MAXT][4], Bop[MAXB][4];        // operands of the dW phase that runs next
        float Wn[4][C::MAXT];                       // the chain's next k-group: W^T values [k-step][tile]
        // lane-linear pair read of a block: rows 4s+g and 4(s+1)+g of slot i16 (one ds_read2st64_b32)
        auto rd2 = [&](const float* blk, int s, float& v0, float& v1) __attribute__((always_inline)) {
            v0 = blk[64 * s + lane]; v1 = blk[64 * (s + 1) + lane];
        };
        auto chain_load = [&](auto l_, int kg, int u) __attribute__((always_inline)) {
            constexpr int l = decltype(l_)::value;
            // (fringe tile of the chain's output = layer l's input units 16 T .. 16 T + F - 1: lane (k phase g, row quad, i) reads column slot 16 T + 4 i)
            const float* wcol = lds + C::wmoff(l) + 4 * g * C::LDM(l) + ((C::fr_in(l) && u == C::TR(l) - 1) ? 4 * (lane & 3) : i16);
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (s < C::ksteps(C::out(l), kg)) Wn[s][u] = wcol[(16 * kg + s) * C::LDM(l) + 16 * u];
        };
        // first k-group of B(NM): requested here, its round trip hides under nothing but is issued once per tile
        {
#pragma unroll
            for (int u = 0; u < C::TR(NM); ++u) chain_load(std::integral_constant<int, NM>{}, 0, u);
        }
        MID_FENCE();
        sfor<0, NM>(SFOR_LAMBDA(li) {
            constexpr int l = NM - SFOR_VAL(li);
            constexpr int TZ = C::TR(l + 1), TAl = C::TA(l), MU = C::TR(l), K = C::out(l), KGn = C::KG(K);
            float* dbl = wl + C::dboff(l);
            const float* ab = wl + C::aboff(l);
            // ---- B(l)
            f32x4 acc[MU], al[MU];
#pragma unroll
            for (int u = 0; u < MU; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            // extra LDS operations threaded through the chain: NW parkings (B(NM) only), then NR reads in the last RG k-groups
            // (with two k-groups the reads wait for the second one: every parking must precede every read in program order)
            constexpr int NW = l == NM ? TZ : 0, NR = TZ + TAl + MU, RG = KGn >= 3 ? 2 : 1;
            auto extra = [&](auto e_) __attribute__((always_inline)) {
                constexpr int e = decltype(e_)::value;
                if constexpr (e < NW) *reinterpret_cast<f32x4*>(dbl + e * 256 + i16 * 16 + 4 * g) = dz[e];
                else if constexpr (e < NW + TZ) rd2(dbl + (e - NW) * 256, 0, Aop[e - NW][0], Aop[e - NW][1]);
                else if constexpr (e < NW + TZ + TAl) rd2(ab + (e - NW - TZ) * 256, 0, Bop[e - NW - TZ][0], Bop[e - NW - TZ][1]);
                else al[e - NW - TZ - TAl] = *reinterpret_cast<const f32x4*>(ab + (e - NW - TZ - TAl) * 256 + i16 * 16 + 4 * g);
            };
            sfor<0, KGn>(SFOR_LAMBDA(kg) {
                constexpr int kg = SFOR_VAL(kg), NS = C::ksteps(K, kg);
                float A[4][MU];
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int u = 0; u < MU; ++u) A[s][u] = Wn[s][u];
                MID_FENCE();
                // LDS operations that ride in this k-group, in program order: parkings (first k-group of B(NM)), the next
                // k-group's W^T values (two k-steps = one ds_read2_b32 per operation), the dW / act' operand reads (last RG
                // k-groups).  Operation k goes behind MFMA k * SPAN / NOPS: one per MFMA, never in the last k-step (its
                // results are needed MU MFMAs later at the earliest).
                constexpr int NWg = kg == 0 ? NW : 0;
                constexpr int NCg = kg + 1 < KGn ? 2 * MU : 0;
                constexpr int rg = kg - (KGn - RG);                                         // index among the read groups (< 0: none)
                constexpr int r0 = rg >= 0 ? (rg * NR) / RG : 0, r1 = rg >= 0 ? ((rg + 1) * NR) / RG : 0;
                constexpr int NOPS = NWg + NCg + (r1 - r0);
                constexpr int NMFg = NS * MU, SPAN = NS > 1 ? (NS - 1) * MU : MU;
                auto op = [&](auto k_) __attribute__((always_inline)) {
                    constexpr int k = decltype(k_)::value;
                    if constexpr (k < NWg) extra(std::integral_constant<int, k>{});
                    else if constexpr (k < NWg + NCg) {
                        constexpr int c = k - NWg, u = c / 2, h = c % 2;
                        const float* wcol = lds + C::wmoff(l) + 4 * g * C::LDM(l) + ((C::fr_in(l) && u == MU - 1) ? 4 * (lane & 3) : i16);
#pragma unroll
                        for (int s2 = 2 * h; s2 < 2 * h + 2; ++s2)
                            if (s2 < C::ksteps(K, kg + 1)) Wn[s2][u] = wcol[(16 * (kg + 1) + s2) * C::LDM(l) + 16 * u];
                    } else extra(std::integral_constant<int, NW + r0 + (k - NWg - NCg)>{});
                };
                sfor<0, NMFg>(SFOR_LAMBDA(j) {
                    constexpr int j = SFOR_VAL(j), sj = j / MU, uj = j % MU;
                    if constexpr (C::fr_in(l) && uj == MU - 1) acc[uj] = mfma4(A[sj][uj], dz[kg][sj], acc[uj]);
                    else acc[uj] = mfma16(A[sj][uj], dz[kg][sj], acc[uj]);
                    // operations k with k * SPAN / NOPS == j
                    constexpr int k0 = NOPS > 0 ? (j * NOPS + SPAN - 1) / SPAN : 0, k1 = NOPS > 0 ? ((j + 1) * NOPS + SPAN - 1) / SPAN : 0;
                    constexpr int ka = k0 < NOPS ? k0 : NOPS, kb = j + 1 >= NMFg ? NOPS : (k1 < NOPS ? k1 : NOPS);
                    sfor<ka, kb>(SFOR_LAMBDA(k) { op(std::integral_constant<int, SFOR_VAL(k)>{}); });
                    if constexpr (kb > ka || uj == MU - 1) MID_FENCE();
                });
            });
            MSTAMP(25 + 3 * SFOR_VAL(li));
            f32x4 dzp[C::MAXT];
            constexpr bool PKR = S::act(l - 1) == TBNN_ACT_RELU && TBNN_F3_RELU_PK && MU <= 8;
            if constexpr (PKR) mfma_settle(acc);
            if constexpr (C::fr_in(l)) {
                float z[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) z[i] = gsum(acc[MU - 1][i]);
                acc[MU - 1] = f32x4{g == 0 ? z[0] : (g == 1 ? z[1] : (g == 2 ? z[2] : z[3])), 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < MU; ++u) dzp[u] = actc_bwd_mul4<S::act(l - 1), PKR>(acc[u], al[u]);
            MSTAMP(4 + 2 * SFOR_VAL(li));
            MID_FENCE();
            // ---- D(l)
            {
                constexpr int NMF = 4 * TZ * TAl;
                float* dbn = wl + C::dboff(l - 1);
                // operations under the MFMAs, in this order: k-steps 2, 3 operands (TZ + TAl), parkings of delta_{l-1} (MU),
                // then the next phase's first requests
                constexpr int N1 = TZ + TAl, N2 = N1 + MU;
                constexpr int NX = l > 1 ? C::TR(l > 1 ? l - 1 : 1) : C::MT0 + C::NT0;
                constexpr int NE = N2 + NX;
                float An2[C::MAXT][2], Bn2[MAXB][2];
                auto under = [&](auto e_) __attribute__((always_inline)) {
                    constexpr int e = decltype(e_)::value;
                    if constexpr (e < TZ) rd2(dbl + e * 256, 2, Aop[e][2], Aop[e][3]);
                    else if constexpr (e < N1) rd2(ab + (e - TZ) * 256, 2, Bop[e - TZ][2], Bop[e - TZ][3]);
                    else if constexpr (e < N2) *reinterpret_cast<f32x4*>(dbn + (e - N1) * 256 + i16 * 16 + 4 * g) = dzp[e - N1];
                    else if constexpr (l > 1) chain_load(std::integral_constant<int, (l > 1 ? l - 1 : 1)>{}, 0, e - N2);
                    else if constexpr (e - N2 < C::MT0) rd2(dbn + (e - N2) * 256, 0, An2[e - N2][0], An2[e - N2][1]);
                    else rd2(wl + C::XB_OFF + (e - N2 - C::MT0) * 256, 0, Bn2[e - N2 - C::MT0][0], Bn2[e - N2 - C::MT0][1]);
                };
                sfor<0, NMF>(SFOR_LAMBDA(j) {
                    constexpr int j = SFOR_VAL(j), sj = j / (TZ * TAl), tj = (j / TAl) % TZ, uj = j % TAl;
                    mfma16_acc<(TZ * TAl > 2)>(dW[C::dwoff(l) + tj * TAl + uj], Aop[tj][sj], Bop[uj][sj]);
                    sfor<mid_ops_before(j, N1, NE, NMF), mid_ops_before(j + 1, N1, NE, NMF)>(SFOR_LAMBDA(e) { under(std::integral_constant<int, SFOR_VAL(e)>{}); });
                    MID_FENCE();
                });
                if constexpr (l == 1) {
#pragma unroll
                    for (int t = 0; t < C::MT0; ++t) { Aop[t][0] = An2[t][0]; Aop[t][1] = An2[t][1]; }
#pragma unroll
                    for (int u = 0; u < C::NT0; ++u) { Bop[u][0] = Bn2[u][0]; Bop[u][1] = Bn2[u][1]; }
                }
            }
            MSTAMP(5 + 2 * SFOR_VAL(li));
#pragma unroll
            for (int u = 0; u < C::MAXT; ++u) if (u < MU) dz[u] = dzp[u];
        });

        // ---- D(0): dW_0 += delta_0^T [x, 1]
        {
            constexpr int NMF = 4 * C::MT0 * C::NT0, N1 = C::MT0 + C::NT0, NE = N1 + C::MT0;
            const float* db0 = wl + C::dboff(0);
            auto under = [&](auto e_) __attribute__((always_inline)) {
                constexpr int e = decltype(e_)::value;
                if constexpr (e < C::MT0) rd2(db0 + e * 256, 2, Aop[e][2], Aop[e][3]);
                else if constexpr (e < N1) rd2(wl + C::XB_OFF + (e - C::MT0) * 256, 2, Bop[e - C::MT0][2], Bop[e - C::MT0][3]);
                else preload_l0(e - N1, C::MT0);                      // the next tile's layer 0: bias tiles + first A group
            };
            sfor<0, NMF>(SFOR_LAMBDA(j) {
                constexpr int j = SFOR_VAL(j), sj = j / (C::MT0 * C::NT0), tj = (j / C::NT0) % C::MT0, uj = j % C::NT0;
                mfma16_acc<(C::MT0 * C::NT0 > 2)>(dW[C::dwoff(0) + tj * C::NT0 + uj], Aop[tj][sj], Bop[uj][sj]);
                sfor<mid_ops_before(j, N1, NE, NMF), mid_ops_before(j + 1, N1, NE, NMF)>(SFOR_LAMBDA(e) { under(std::integral_constant<int, SFOR_VAL(e)>{}); });
                MID_FENCE();
            });
        }
        MSTAMP(4 + 2 * NM);
        }   // !FWD
    }
    if constexpr (!FWD) {
    mfma_drain_acc(dW);

    // ---- epilogue: every wave stages its dW tiles [wave][tile][lane] (16 B per lane), wave t % 4 sums the 4 copies of
    // tile t in fixed order and writes the dense slab (theta order); EP_TILES per pass.  Deterministic.
    const double wtot = wave_sum_lane0(stat);
    if (lane == 0) red[wave] = wtot;
    float* slab = slabs + (size_t)blockIdx.x * pitch;
    // (compile-time pass and tile indices throughout: a run-time index into dW would turn the accumulators into a
    // scratch array, read back without the wait states an MFMA result needs)
    sfor<0, (C::DW_TILES + C::EP_TILES - 1) / C::EP_TILES>(SFOR_LAMBDA(pass) {
        constexpr int t0 = SFOR_VAL(pass) * C::EP_TILES, t1 = t0 + C::EP_TILES < C::DW_TILES ? t0 + C::EP_TILES : C::DW_TILES;
        __syncthreads();                   // images / blocks (or the previous pass) are dead
        f32x4* mine = reinterpret_cast<f32x4*>(lds) + wave * (C::EP_TILES * 64);
        sfor<t0, t1>(SFOR_LAMBDA(t) { mine[(SFOR_VAL(t) - t0) * 64 + lane] = dW[SFOR_VAL(t)]; });
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
                                    slab_store<(C::P() >= 2048)>(slab + C::offW(l) + (col < inl ? row * inl + col : inl * outl + row), (c0[r] + c1[r]) + (c2[r] + c3[r]));
                            }
                        }
                    }
                }
        });
    });
    if constexpr (C::VL) {
        // VALU last layer: reduce the per-row partials over the 16 lanes of a lane group, then over the 4 waves
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
            slab_store<(C::P() >= 2048)>(slab + C::offW(LL) + (u < inL ? o * inL + u : inL * d_out + o), (v[0] + v[1]) + (v[2] + v[3]));
        }
    }
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < MID_WAVES; ++w) t += red[w];
        pstat[blockIdx.x] = t;
    }
    }   // !FWD
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
                else if (C::VL && l == C::LL) m0 = C::WL_OFF + i * C::WLP + ck;
                else m0 = C::wmoff(l) + ri * C::LDM(l) + ck;
                map[ow + i * in + k] = m0;
                map[P + ow + i * in + k] = -1;
            }
            map[ow + in * out + i] = (C::VL && l == C::LL) ? C::BL_OFF + i : C::boff(l) + ri;
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
                               const float* Y, long n, float* slabs, int pitch, double* pstat, int nchains = 1, ChainStride cs = ChainStride{0, 0, 0, nullptr, 0}) {
    hipLaunchKernelGGL((k_fwd_bwd_mid<S, false>), dim3(grid, nchains), dim3(MID_THREADS), 0, st, nd, qimg, cs.img, eta, X, Y, n, slabs, pitch, pstat,
                       (float*)nullptr, 0L, cs);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
// forward only: `nets` networks (grid.y; images img_stride floats apart), fouts[net][d_out][n]
template <class S>
static inline int mid_forward_t(int gx, int nets, hipStream_t st, const float* qimgs, long img_stride, const float* X, long n,
                                float* fouts, long out_stride) {
    NetDev nd{};
    hipLaunchKernelGGL((k_fwd_bwd_mid<S, true>), dim3(gx, nets), dim3(MID_THREADS), 0, st, nd, qimgs, img_stride, (const float*)nullptr, X,
                       (const float*)nullptr, n, (float*)nullptr, 0, (double*)nullptr, fouts, out_stride, ChainStride{0, 0, 0, nullptr, 0});
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
