// k_fwd_bwd_fast3: k_fwd_bwd_fast with the *fringe* units of every layer off the 16x16x4 MFMA tiles.
//
// A 50-unit layer fills 3 full 16-unit MFMA tiles plus a 4th tile that carries 2 units: a quarter of the
// forward / delta-chain MFMAs and 4 of the 16 dW tiles of a layer exist for 2 units.  Here a layer with
// out % 16 in {1, 2} keeps MTF = out/16 full tiles on the 16x16x4 path and computes its NF fringe units apart:
//   forward   z_u = b_u + sum_k W[u][k] a[k]: per-lane partial sums over the lane's own k-slots on the 16-block
//             v_mfma_f32_4x4x1 (fringe_partials), summed over the 4 lane groups by ONE 16x16x4 MFMA with A = 1
//             (gsum_mfma); the result (all lanes) is dropped into register 0 of the fringe tile (lane group f) so
//             that it feeds the next layer's last k-step exactly like an MFMA result would;
//   delta     d_u = act'(a_u) sum_i W[i][u] delta[i]: same shape, W^T image;
//   dW rows   dW[u][k] += d_u a_{l-1}[k]: in the B-operand layout of the a_{l-1} image registers that feed the MFMA
//             part of dW_l anyway (fdw: one v_pk_fma per N tile and data row), 2 registers per N tile, reduced over
//             the lane groups and waves once per launch in the epilogue (FringeOut).
// The last layer (<= 2 outputs) is the all-fringe case (MTF = 0) and keeps per-lane sums in the D layout.
// On gfx950 the f32 MFMA and the f32 VALU share one issue slot (tools/ubench/coexec.hip: every VALU instruction
// adds ~9 cycles to the MFMA stream), so the tile body is shaped by instruction count: AccVGPR-pinned dW
// accumulators next to VGPR-form chain MFMAs (kernels_fast.hpp, mfma16_acc), LDS traffic threaded by hand through
// the dW MFMAs (Pipe3), packed instructions kept packed (pkfma*), a 1.5-instruction relu derivative (actc_bwd_mul4).
// C2: 270 + 67 (4x4x1) + 11 MFMAs, 227 VALU and 148 LDS instructions per 16-row tile.  DESIGN.md section 4.
#pragma once
#include <type_traits>
#include "kernels_fast.hpp"

// compile-time loop (indices usable as template arguments / array subscripts that must stay registers)
template <int I, int N, class F>
__device__ __forceinline__ void sfor3(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor3<I + 1, N>(f); }
}
#ifndef TBNN_F3_HAND
#define TBNN_F3_HAND 1
#endif

// MFMAs per LDS instruction when issue(l) is threaded through dW_{l+1} (0: separate phases, the round-1 schedule).
// Measured on configs[1]: 54.3 us (0) -> 53.9 us (1..3).
#ifndef TBNN_SGB
#define TBNN_SGB 2
#endif
// 1: fringe dW rows in the B-operand layout (fewer registers: the f32 VALU shares the MFMA's issue slot, every
// spill copy costs ~9 cycles of tile time); 0: per-lane sums in the D layout (the first fast3 version)
#ifndef TBNN_F3_FB
#define TBNN_F3_FB 1
#endif
// 1: thread the LDS instructions of issue(l) through the (asm) dW MFMAs of layer l+1 by hand
#ifndef TBNN_F3_THREAD
#define TBNN_F3_THREAD 1
#endif
template <class S>
struct F3Cfg : FastCfg<S> {
    using B = FastCfg<S>;
    static constexpr int NF(int l) { return (B::out(l) % 16 == 1 || B::out(l) % 16 == 2) ? B::out(l) % 16 : 0; }
    static constexpr int MTF(int l) { return NF(l) ? B::MT(l) - 1 : B::MT(l); }      // tiles on the MFMA path
    static constexpr int fslot(int l, int f) { return 16 * MTF(l) + 4 * f; }          // slot of fringe unit f (lane group f, reg 0)
    static constexpr int funit(int l, int f) { return 16 * MTF(l) + f; }
    // number of layers whose dW has an MFMA part (a trailing all-fringe layer has none)
    static constexpr int NLM3 = MTF(B::NL - 1) == 0 ? B::NL - 1 : B::NL;
    static_assert(NLM3 == B::NLM, "fast3 expects the same MFMA/VALU split of the last layer as fast");
    // N-side fringe of dW_l (round 5): 50 inputs + the bias column are 3 full N tiles + a 4th that holds 3 useful columns of 16 (units
    // 48, 49 and the ones slot: image slots 48 + 4 j).  With at most 4 such columns the last N tile leaves the 16x16x4 path (MTF x 4
    // MFMAs of 32 cycles per tile and layer) for the 16-block 4x4x1 form: block b = units 4b .. 4b+3, columns = slots 48 + 4j, one
    // instruction (8 cycles) per data row -- 16 per tile and layer, ONE accumulator tile instead of MTF.  NTF: N tiles on the 16x16x4 path.
#ifndef TBNN_F3_NFRINGE
#define TBNN_F3_NFRINGE 1
#endif
    // (only where the epilogue stages all dW tiles in ONE pass -- at most 39 tiles before the change: the pair of N-fringe accumulators is summed there)
    static constexpr int dw_tiles_plain() { int o = 0; for (int m = 0; m < NLM3; ++m) o += MTF(m) * B::NT(m); return o; }
    static constexpr bool NCF(int l) {
        return TBNN_F3_NFRINGE && TBNN_ACC_AGPR && TBNN_F3_THREAD && dw_tiles_plain() <= 39 && l < NLM3 && B::NT(l) >= 2 && MTF(l) >= 1 && MTF(l) <= 4 && B::in(l) + 1 - 16 * (B::NT(l) - 1) <= 4;
    }
    static constexpr int NTF(int l) { return NCF(l) ? B::NT(l) - 1 : B::NT(l); }
    // (two N-fringe accumulators, even / odd data rows: a 4x4x1 MFMA that accumulates into the result of the one before it needs
    // wait states the inline-asm form does not get -- back to back it read a stale accumulator; the pair is summed in the epilogue)
    static constexpr int dwt3(int l) { return MTF(l) * NTF(l) + (NCF(l) ? 2 : 0); }        // accumulator tiles of layer l
    static constexpr int dwoff3(int l) { int o = 0; for (int m = 0; m < l && m < NLM3; ++m) o += dwt3(m); return o; }
    static constexpr int dwfr3(int l) { return dwoff3(l) + MTF(l) * NTF(l); }               // layer l's N-fringe tile
    static constexpr int DW3_TILES = dwoff3(B::NL);
    // per-lane k-slots of layer l's input: natural x for layer 0, else 4 registers of every tile of a_{l-1}
    static constexpr int KIN(int l) { return l == 0 ? B::KS0 : 4 * B::MT(l - 1); }
    // fringe rows of dW: layers with an MFMA dW part accumulate them in the *B-operand* layout (the a_{l-1} image
    // registers that feed dW_l anyway): 2 registers per N tile instead of NF*(KIN+1) per-lane sums in the D layout.
    static constexpr bool FB(int l) { return TBNN_F3_FB && NF(l) > 0 && MTF(l) > 0; }
    static constexpr int fpn(int l) { return NF(l) == 0 ? 0 : (FB(l) ? 2 * B::NT(l) : NF(l) * (KIN(l) + 1)); }
    static constexpr int fpoff(int l) { int o = 0; for (int m = 0; m < l; ++m) o += fpn(m); return o; }
    static constexpr int FP_REGS = fpoff(B::NL);
    static constexpr int maxNF() { int m = 0; for (int l = 0; l < B::NL; ++l) m = NF(l) > m ? NF(l) : m; return m; }
    static constexpr int EP3_WANT = DW3_TILES * FAST_WAVES * 256 <= 39936 ? DW3_TILES : (DW3_TILES < 16 ? DW3_TILES : 16);
    // transposed images, column-major: [input slot][row], pitch PR floats.  A lane's 4 operand rows (4g..4g+3)
    // are then one 16-B read (a single wave gets poor ds_read_b32 throughput); the D-layout writes become
    // bank-conflict-free b32 pairs.
    static constexpr int PR = 16 + TBNN_PPAD;
    static constexpr int aoff3(int l) { int o = 0; for (int m = 0; m < l && m < NLM3; ++m) o += 16 * B::NT(m) * PR; return o; }
    static constexpr int doff3 = aoff3(NLM3);
    static constexpr int fdoff3 = doff3 + 16 * B::maxMT() * PR;   // fringe deltas [lane group copy][row][NF]: 4 x 32 floats
    static constexpr int WAVE3_FLOATS = fdoff3 + 128;
    static constexpr int MIN3 = B::STATIC_FLOATS + FAST_WAVES * WAVE3_FLOATS;
    static constexpr int LDS3_A = MIN3 > EP3_WANT * FAST_WAVES * 256 ? MIN3 : EP3_WANT * FAST_WAVES * 256;
    static constexpr int LDS3_FLOATS = LDS3_A > FAST_WAVES * (FP_REGS > 0 ? FP_REGS : 1) * 64 ? LDS3_A : FAST_WAVES * FP_REGS * 64;
    static constexpr int EP3_TILES = LDS3_FLOATS / (FAST_WAVES * 256) < DW3_TILES ? LDS3_FLOATS / (FAST_WAVES * 256) : (DW3_TILES > 0 ? DW3_TILES : 1);
};

template <class S>
struct Tile3 {
    using C = F3Cfg<S>;
    f32x4 a[C::ACT_TILES];                 // outputs of every layer, D layout (fringe tile: reg 0 of lane group f)
    float af[C::NL][C::maxNF() > 0 ? C::maxNF() : 1];   // fringe activations (all lanes: value of row lane&15)
    float x0[C::KS0];
};

// TBNN_F3_M4 (default): the fringe dot products on the 16-block v_mfma_f32_4x4x1_f32 instead of VALU FMAs.
// Block b = lane/4 = (lane group g, row quad i16/4) multiplies A[m = i16&3] = W[fringe unit m][k-slot of group g] by
// B[n = i16&3] = a[k-slot][row i16] -- the D-layout activation register as it stands -- so after one instruction per
// k-slot register, accumulator register m of every lane holds exactly the per-lane partial sum that dot_slots computes
// for fringe unit m (2 passes = 8 cycles per instruction against ~12 cycles per dependent packed FMA).  The sum over
// the 4 lane groups is ONE 16x16x4 MFMA with A = 1 (B[k = g][n = i16] is the register of partials as it stands):
// every register of every lane of the result holds the finished pre-activation of row i16.
#ifndef TBNN_F3_M4
#define TBNN_F3_M4 1
#endif
#ifndef TBNN_F3_M4ACC
#define TBNN_F3_M4ACC 2      // (4 until round 5; measured at configs[1]: 2 -> +0.5 %, 1 -> -0.2 %)
#endif
// prow: this lane's A-operand row (image row 16*MTF + 4*(i16&3), + 4g); tiles: the K dimension in D-layout registers
template <class S, int K>
__device__ __forceinline__ f32x4 fringe_partials(const float* __restrict__ prow, const f32x4* tiles) {
    using C = F3Cfg<S>;
    // TBNN_F3_M4ACC accumulators, k-groups dealt round-robin, summed at the end (dependent 4x4x1 MFMAs issue back to back
    // at full rate, tools/ubench/chain.hip; more accumulators only shorten the chain the gsum MFMA waits for)
    constexpr int KG = C::cdiv(K, 16), NA = KG < TBNN_F3_M4ACC ? KG : TBNN_F3_M4ACC;
    f32x4 acc[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < KG; ++kt) {
        const f32x4 w = load_ks(prow + 16 * kt, C::ksteps(K, kt));
#pragma unroll
        for (int s = 0; s < C::ksteps(K, kt); ++s) acc[kt % NA] = mfma4(w[s], tiles[kt][s], acc[kt % NA]);
    }
#pragma unroll
    for (int st = 1; st < NA; st *= 2)
#pragma unroll
        for (int a = 0; a + st < NA; a += 2 * st) acc[a] += acc[a + st];
    return acc[0];
}
// sum_k w[k-slot] * v[k-slot] over this lane's k-slots of a K dimension living in D-layout tiles
// PK: `tiles` hold VALU results (activation outputs / masked deltas), so the opaque packed FMA may read them
template <class S, int K, bool PK>
__device__ __forceinline__ float dot_slots(const float* __restrict__ row, const f32x4* tiles, int g) {
    using C = F3Cfg<S>;
    // pairs of k-slots on v_pk_fma_f32 (two partial sums), the odd one out on a plain FMA
    f32x2 p2 = {0.f, 0.f};
    float p = 0.f;
#pragma unroll
    for (int kt = 0; kt < C::cdiv(K, 16); ++kt) {
        const f32x4 w = load_ks(row + 16 * kt + 4 * g, C::ksteps(K, kt));
        const int ns = C::ksteps(K, kt);
#pragma unroll
        for (int s = 0; s + 1 < ns; s += 2) {
            if constexpr (PK) p2 = pkfma(f32x2{w[s], w[s + 1]}, f32x2{tiles[kt][s], tiles[kt][s + 1]}, p2);
            else p2 = f32x2{w[s], w[s + 1]} * f32x2{tiles[kt][s], tiles[kt][s + 1]} + p2;
        }
        if (ns & 1) p = fmaf(w[ns - 1], tiles[kt][ns - 1], p);
    }
    return (p2[0] + p2[1]) + p;
}

// IMG: write the transposed activation images the dW MFMAs read (off in the forward-only kernel)
template <class S, int l, bool IMG = true>
struct Fwd3 {
    using C = F3Cfg<S>;
    static __device__ __forceinline__ void preload(f32x4 (&An)[C::MTF(l) > 0 ? C::MTF(l) : 1], f32x4 (&Bn)[C::MTF(l) > 0 ? C::MTF(l) : 1],
                                                    const float* __restrict__ lds, int i16, int g) {
        constexpr int MT = C::MTF(l);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) Bn[mt] = *reinterpret_cast<const f32x4*>(lds + C::boff(l) + 16 * mt + 4 * g);
        if constexpr (l == 0) {
            static_assert(C::KS0 <= 4, "layer-0 fan-in above 16 needs the grouped path");
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int t = 0; t < C::KS0; ++t) An[mt][t] = lds[C::woff(0) + (16 * mt + i16) * C::LDW(0) + 4 * t + g];
        } else {
            const float* wrow = lds + C::woff(l) + i16 * C::LDW(l) + 4 * g;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) An[mt] = load_ks(wrow + 16 * mt * C::LDW(l), C::ksteps(C::in(l), 0));
        }
    }

    // ---- hand-scheduled form (TBNN_F3_HAND, default).  Left to itself the compiler requests an LDS operand two MFMAs before
    // the MFMA that reads it and stalls on the round trip in every k-group, and it gathers the fringe 4x4x1 MFMAs at the end
    // of the layer, in front of the lane-sum MFMAs that depend on them (ISA of round 2: ~740 of a hidden layer's ~2,150
    // cycles are not MFMA time).  Here the issue order of a k-group is written out and pinned with scheduling fences
    // (nothing crosses __builtin_amdgcn_sched_barrier(0)): k-step-major MFMAs, the fringe 4x4x1 behind each k-step, and
    // under the first MFMAs of the group one LDS read each -- the operands of the NEXT k-group, and in the last full group
    // the first operands of the next LAYER (A tiles, bias tiles, fringe weights).
    static constexpr int MTd = C::MTF(l) > 0 ? C::MTF(l) : 1;
    static constexpr bool M4F = C::NF(l) > 0 && TBNN_F3_M4;                        // fringe dot products on the 4x4x1 MFMA
    static constexpr int KGl = l == 0 ? 1 : C::KG(l);
    static constexpr int NFG = (C::MTF(l) == 0 && l > 0) ? KGl : 1;              // fringe-weight k-groups requested ahead (all of them for an all-fringe layer)
    struct Pre { f32x4 A[MTd]; f32x4 B[MTd]; f32x4 F[NFG]; };
    static constexpr int NPRE = 2 * C::MTF(l) + (M4F ? NFG : 0);                   // LDS reads of one preload (layer 0: b32 reads, counted per tile)
    static __device__ __forceinline__ const float* frow(const float* __restrict__ lds, int i16, int g) {
        return lds + C::woff(l) + (16 * C::MTF(l) + 4 * (i16 & 3)) * C::LDW(l) + (l == 0 ? g : 4 * g);
    }
    // read number i of the preload list: A tiles, bias tiles, fringe weights
    template <int i>
    static __device__ __forceinline__ void pre_step(Pre& P, const float* __restrict__ lds, int i16, int g) {
        constexpr int MT = C::MTF(l);
        if constexpr (i < MT) {
            if constexpr (l == 0) {
#pragma unroll
                for (int t = 0; t < C::KS0; ++t) P.A[i][t] = lds[C::woff(0) + (16 * i + i16) * C::LDW(0) + 4 * t + g];
            } else {
                P.A[i] = load_ks(lds + C::woff(l) + i16 * C::LDW(l) + 4 * g + 16 * i * C::LDW(l), C::ksteps(C::in(l), 0));
            }
        } else if constexpr (i < 2 * MT) {
            P.B[i - MT] = *reinterpret_cast<const f32x4*>(lds + C::boff(l) + 16 * (i - MT) + 4 * g);
        } else {
            constexpr int k = i - 2 * MT;
            if constexpr (l == 0) {
                P.F[0] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < C::KS0; ++t) P.F[0][t] = frow(lds, i16, g)[4 * t];
            } else {
                P.F[k] = load_ks(frow(lds, i16, g) + 16 * k, C::ksteps(C::in(l), k));
            }
        }
    }
    static __device__ __forceinline__ void pre_all(Pre& P, const float* __restrict__ lds, int i16, int g) {
        sfor3<0, NPRE>([&](auto i_) __attribute__((always_inline)) { pre_step<decltype(i_)::value>(P, lds, i16, g); });
    }

    static __device__ __forceinline__ void run_h(Tile3<S>& T, const float* __restrict__ lds, float* wl, int i16, int g, const Pre& P0) {
        constexpr int MT = C::MTF(l), NF = C::NF(l), K = l == 0 ? C::in(0) : C::in(l), KG = KGl;
        constexpr bool more = l + 1 < C::NL;
        using NX = Fwd3<S, (more ? l + 1 : l), IMG>;
        typename NX::Pre Pn;
        constexpr int NPN = more ? NX::NPRE : 0;
        constexpr int NA = KG < TBNN_F3_M4ACC ? KG : TBNN_F3_M4ACC;
        f32x4 acc[MTd], pa[NA];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = P0.B[mt];
#pragma unroll
        for (int a = 0; a < NA; ++a) pa[a] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 An[MTd], Fn = P0.F[0];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) An[mt] = P0.A[mt];
        // the next layer's preload rides in the last FULL k-group (or the only one)
        constexpr int PG = (KG >= 2 && (l == 0 ? 4 : C::ksteps(K, KG - 1)) < 4) ? KG - 2 : KG - 1;
        sfor3<0, KG>([&](auto kt_) __attribute__((always_inline)) {
            constexpr int kt = decltype(kt_)::value;
            constexpr int ns = l == 0 ? C::KS0 : C::ksteps(K, kt);
            f32x4 A4[MTd], F4;
            if constexpr (MT == 0 && l > 0) F4 = P0.F[kt]; else F4 = Fn;       // (constexpr: the other arm's subscript is out of range)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) A4[mt] = An[mt];
            // reads that ride in this group: the next group's operands (MT tiles + fringe weights), then the preload
            constexpr int NLD1 = (kt + 1 < KG && MT > 0) ? MT + (M4F ? 1 : 0) : 0;
            constexpr int NLD = NLD1 + (kt == PG ? NPN : 0);
            auto issue = [&](auto q_) __attribute__((always_inline)) {
                constexpr int q = decltype(q_)::value;
                if constexpr (q < NLD1) {
                    if constexpr (q < MT) An[q] = load_ks(lds + C::woff(l) + i16 * C::LDW(l) + 4 * g + 16 * q * C::LDW(l) + 16 * (kt + 1), C::ksteps(K, kt + 1));
                    else Fn = load_ks(frow(lds, i16, g) + 16 * (kt + 1), C::ksteps(K, kt + 1));
                } else if constexpr (q < NLD) {
                    NX::template pre_step<q - NLD1>(Pn, lds, i16, g);
                }
            };
            __builtin_amdgcn_sched_barrier(0);
            sfor3<0, ns>([&](auto s_) __attribute__((always_inline)) {
                constexpr int s = decltype(s_)::value;
                sfor3<0, MT + (M4F ? 1 : 0)>([&](auto m_) __attribute__((always_inline)) {
                    constexpr int m = decltype(m_)::value, q = s * (MT + (M4F ? 1 : 0)) + m;
                    float bop;
                    if constexpr (l == 0) bop = T.x0[s]; else bop = T.a[C::aroff(l > 0 ? l - 1 : 0) + kt][s];
                    if constexpr (m < MT) acc[m] = mfma16(A4[m][s], bop, acc[m]);
                    else pa[kt % NA] = mfma4(F4[s], bop, pa[kt % NA]);
                    if constexpr (q < NLD) { issue(std::integral_constant<int, q>{}); __builtin_amdgcn_sched_barrier(0); }
                });
                __builtin_amdgcn_sched_barrier(0);
            });
            // reads that found no MFMA to ride under (short groups)
            sfor3<ns * (MT + (M4F ? 1 : 0)), NLD>([&](auto q_) __attribute__((always_inline)) { issue(q_); });
            __builtin_amdgcn_sched_barrier(0);
        });
        // ---- layer epilogue: fringe lane sums (MFMA), activations, images -- the compiler's order
        f32x4 pacc = pa[0];
        if constexpr (M4F) {
#pragma unroll
            for (int st = 1; st < NA; st *= 2)
#pragma unroll
                for (int a = 0; a + st < NA; a += 2 * st) pa[a] += pa[a + st];
            pacc = pa[0];
        }
        if constexpr (MT > 0) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = actc_fwd<S::act(l)>(acc[mt][r]);
                T.a[C::aroff(l) + mt] = v;
            }
        }
        if constexpr (NF > 0) {
            static_assert(TBNN_F3_M4, "the hand-scheduled form computes the fringe units on the 4x4x1 MFMA");
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                T.af[l][f] = gsum_mfma(pacc[f], [&](float s) { return actc_fwd<S::act(l)>(s + lds[C::boff(l) + C::fslot(l, f)]); });
                if (g == f) v[0] = T.af[l][f];
            }
            T.a[C::aroff(l) + MT] = v;
        }
        if constexpr (more) {
            if constexpr (IMG && C::MTF(l + 1 < C::NL ? l + 1 : l) > 0) {
                constexpr int u1 = C::in(l + 1 < C::NL ? l + 1 : l);
                float* aimg = wl + C::aoff3(l + 1);
#pragma unroll
                for (int mt = 0; mt < C::MT(l); ++mt) {
                    f32x4 v = T.a[C::aroff(l) + mt];
                    if constexpr (u1 % 16 != 0) {
                        constexpr int osl = ones_slot(u1);
                        if (mt == osl / 16 && g == (osl % 16) / 4) v[osl % 4] = 1.f;
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) aimg[(16 * mt + 4 * g + r) * C::PR + i16] = v[r];
                }
            }
            TSTAMP(1 + l);
            __builtin_amdgcn_sched_barrier(0);
            NX::run_h(T, lds, wl, i16, g, Pn);
        } else { TSTAMP(1 + l); }
    }

    static __device__ __forceinline__ void run(Tile3<S>& T, const float* __restrict__ lds, float* wl, int i16, int g,
                                                const f32x4 (&A0)[C::MTF(l) > 0 ? C::MTF(l) : 1],
                                                const f32x4 (&B0)[C::MTF(l) > 0 ? C::MTF(l) : 1]) {
        constexpr int MT = C::MTF(l), NF = C::NF(l);
        constexpr bool more = l + 1 < C::NL;
        constexpr int MTN = more ? (C::MTF(l + 1) > 0 ? C::MTF(l + 1) : 1) : 1;
        f32x4 Anext[MTN], Bnext[MTN];
        // ---- fringe units on the VALU (issued first: their two lane shuffles land under the MFMAs below)
        float pf[NF > 0 ? NF : 1];
        f32x4 pacc = {0.f, 0.f, 0.f, 0.f};
        if constexpr (NF > 0 && TBNN_F3_M4) {
            const float* prow = lds + C::woff(l) + (16 * MT + 4 * (i16 & 3)) * C::LDW(l);
            if constexpr (l == 0) {
#pragma unroll
                for (int t = 0; t < C::KS0; ++t) pacc = mfma4(prow[4 * t + g], T.x0[t], pacc);
            } else {
                pacc = fringe_partials<S, C::in(l)>(prow + 4 * g, &T.a[C::aroff(l - 1)]);
            }
        }
#pragma unroll
        for (int f = 0; f < (TBNN_F3_M4 ? 0 : NF); ++f) {
            const float* row = lds + C::woff(l) + C::fslot(l, f) * C::LDW(l);
            float p = 0.f;
            if constexpr (l == 0) {
#pragma unroll
                for (int t = 0; t < C::KS0; ++t) p = fmaf(row[4 * t + g], T.x0[t], p);
            } else {
                p = dot_slots<S, C::in(l), S::act(l - 1) != TBNN_ACT_NONE>(row, &T.a[C::aroff(l - 1)], g);
            }
            pf[f] = p;
        }
        // ---- full tiles on the MFMA path
        if constexpr (MT > 0) {
            f32x4 acc[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = B0[mt];
            if constexpr (l == 0) {
                if constexpr (more && C::MTF(l + 1) > 0) Fwd3<S, l + 1>::preload(Anext, Bnext, lds, i16, g);
#pragma unroll
                for (int t = 0; t < C::KS0; ++t)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma16(A0[mt][t], T.x0[t], acc[mt]);
            } else {
                constexpr int KG = C::KG(l);
                const float* wrow = lds + C::woff(l) + i16 * C::LDW(l) + 4 * g;
                f32x4 An[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) An[mt] = A0[mt];
#pragma unroll
                for (int kt = 0; kt < KG; ++kt) {
                    f32x4 A4[MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) A4[mt] = An[mt];
                    if (kt + 1 < KG) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            An[mt] = load_ks(wrow + 16 * mt * C::LDW(l) + 16 * (kt + 1), C::ksteps(C::in(l), kt + 1));
                    } else {
                        if constexpr (more && C::MTF(l + 1) > 0) Fwd3<S, l + 1>::preload(Anext, Bnext, lds, i16, g);
                    }
#pragma unroll
                    for (int s = 0; s < C::ksteps(C::in(l), kt); ++s)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma16(A4[mt][s], T.a[C::aroff(l - 1) + kt][s], acc[mt]);
                }
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = actc_fwd<S::act(l)>(acc[mt][r]);
                T.a[C::aroff(l) + mt] = v;
            }
        }
        // ---- finish the fringe units: combine the lane groups, bias, activation; drop into the fringe tile
        if constexpr (NF > 0) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                if constexpr (TBNN_F3_M4) T.af[l][f] = gsum_mfma(pacc[f], [&](float s) { return actc_fwd<S::act(l)>(s + lds[C::boff(l) + C::fslot(l, f)]); });
                else T.af[l][f] = actc_fwd<S::act(l)>(gsum(pf[f]) + lds[C::boff(l) + C::fslot(l, f)]);
                if (g == f) v[0] = T.af[l][f];
            }
            T.a[C::aroff(l) + MT] = v;
        }
        if constexpr (more) {
            if constexpr (IMG && C::MTF(l + 1) > 0) {
                // transposed image of a_{l+1}'s input (= this layer's output) for the MFMA part of dW_{l+1}
                constexpr int u1 = C::in(l + 1);
                float* aimg = wl + C::aoff3(l + 1);
#pragma unroll
                for (int mt = 0; mt < C::MT(l); ++mt) {
                    f32x4 v = T.a[C::aroff(l) + mt];
                    if constexpr (u1 % 16 != 0) {
                        constexpr int osl = ones_slot(u1);
                        if (mt == osl / 16 && g == (osl % 16) / 4) v[osl % 4] = 1.f;
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) aimg[(16 * mt + 4 * g + r) * C::PR + i16] = v[r];
                }
            }
            TSTAMP(1 + l);
            Fwd3<S, l + 1, IMG>::run(T, lds, wl, i16, g, Anext, Bnext);
        } else { TSTAMP(1 + l); }
    }
};

// per-lane partial sums of the fringe rows of dW (and db): FP[fpoff(l) + f*(KIN+1) + k], k == KIN: bias
template <class S, int l>
struct FringeDW {
    using C = F3Cfg<S>;
    static __device__ __forceinline__ void run(float (&FP)[C::FP_REGS > 0 ? C::FP_REGS : 1], const Tile3<S>& T,
                                                const float (&dzf)[C::maxNF() > 0 ? C::maxNF() : 1], int g) {
        constexpr int NF = (C::FB(l) || (TBNN_SKEL & 2)) ? 0 : C::NF(l), KIN = C::KIN(l);
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            float* fp = FP + C::fpoff(l) + f * (KIN + 1);
            if constexpr (l == 0) {
#pragma unroll
                for (int t = 0; t < C::KS0; ++t) fp[t] = fmaf(dzf[f], T.x0[t], fp[t]);
            } else if constexpr (S::act(l - 1) != TBNN_ACT_NONE) {
                const f32x2 d2 = {dzf[f], dzf[f]};
#pragma unroll
                for (int kt = 0; kt < C::MT(l - 1); ++kt)
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        const f32x2 v = pkfma_bc<0>(f32x2{T.a[C::aroff(l - 1) + kt][r], T.a[C::aroff(l - 1) + kt][r + 1]}, d2,
                                                    f32x2{fp[4 * kt + r], fp[4 * kt + r + 1]});
                        fp[4 * kt + r] = v[0]; fp[4 * kt + r + 1] = v[1];
                    }
            } else {
#pragma unroll
                for (int kt = 0; kt < C::MT(l - 1); ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) fp[4 * kt + r] = fmaf(dzf[f], T.a[C::aroff(l - 1) + kt][r], fp[4 * kt + r]);
            }
            fp[KIN] += dzf[f];              // every lane group adds it; the epilogue takes lane group 0
        }
    }
};

// backward of layer l: dz (full tiles, D layout) + dzf (fringe deltas, all lanes) are known.
template <class S, int l>
struct Bwd3 {
    using C = F3Cfg<S>;
    static constexpr int MT = C::MTF(l), NT = C::NT(l), NF = C::NF(l);
    static constexpr int MTd = MT > 0 ? MT : 1, NFd = C::maxNF() > 0 ? C::maxNF() : 1;
    static constexpr int NTM = C::NTF(l);                         // N tiles of dW_l on the 16x16x4 path (NT - 1 with the N-side fringe)
    static constexpr bool NCF = C::NCF(l);

    // W(D_l) R(op_l): only when the layer has an MFMA dW part
    static __device__ __forceinline__ void issue(const f32x4 (&dz)[C::MT(l)], const float (&dzf)[NFd], float* wl, int i16, int g,
                                                  float (&Aop)[MTd][4], float (&Bop)[NT][4], float (&Fop)[8]) {
        if constexpr (MT > 0) {
            float* dimg = wl + C::doff3;
            const float* aimg = wl + C::aoff3(l);
            if constexpr (C::FB(l)) {
                // fringe deltas of rows 4g..4g+3 for every lane: each lane group keeps its own copy [row][NF]
                float* fd = wl + C::fdoff3 + 32 * g;
                if constexpr (NF == 2) *reinterpret_cast<f32x2*>(fd + 2 * i16) = f32x2{dzf[0], dzf[1]};
                else fd[i16] = dzf[0];
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) dimg[(16 * mt + 4 * g + r) * C::PR + i16] = dz[mt][r];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(aimg + (16 * nt + i16) * C::PR + 4 * g);   // rows 4g..4g+3
#pragma unroll
                for (int s = 0; s < 4; ++s) Bop[nt][s] = b[s];
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(dimg + (16 * mt + i16) * C::PR + 4 * g);
#pragma unroll
                for (int s = 0; s < 4; ++s) Aop[mt][s] = a[s];
            }
            if constexpr (C::FB(l)) {
                const float* fd = wl + C::fdoff3 + 32 * g;
                if constexpr (NF == 2) {
                    const f32x4 u = *reinterpret_cast<const f32x4*>(fd + 8 * g), v = *reinterpret_cast<const f32x4*>(fd + 8 * g + 4);
#pragma unroll
                    for (int s = 0; s < 4; ++s) { Fop[s] = u[s]; Fop[4 + s] = v[s]; }
                } else {
                    const f32x4 u = *reinterpret_cast<const f32x4*>(fd + 4 * g);
#pragma unroll
                    for (int s = 0; s < 4; ++s) Fop[s] = u[s];
                }
            }
        }
    }
    // issue() one LDS instruction (pair) at a time, so that Pipe3 can thread it through the dW MFMAs of the layer above
    // (an asm-volatile MFMA pins the memory operations around it, the order written here is the order issued)
    static constexpr int NWR = MT > 0 ? 2 * MT + (C::FB(l) ? 1 : 0) : 0;
    static constexpr int NRD = MT > 0 ? NT + MT + (C::FB(l) ? (NF == 2 ? 2 : 1) : 0) : 0;
    static constexpr int NLDS = NWR + NRD;
    static __device__ __forceinline__ void issue_step(int i, const f32x4 (&dz)[C::MT(l)], const float (&dzf)[NFd], float* wl, int i16, int g,
                                                       float (&Aop)[MTd][4], float (&Bop)[NT][4], float (&Fop)[8]) {
        if constexpr (MT > 0) {
            float* dimg = wl + C::doff3;
            const float* aimg = wl + C::aoff3(l);
            float* fd = wl + C::fdoff3 + 32 * g;
            if (i < 2 * MT) {
                const int mt = i >> 1, r = 2 * (i & 1);
                dimg[(16 * mt + 4 * g + r) * C::PR + i16] = dz[mt][r];
                dimg[(16 * mt + 4 * g + r + 1) * C::PR + i16] = dz[mt][r + 1];
            } else if (C::FB(l) && i == 2 * MT) {
                if constexpr (NF == 2) *reinterpret_cast<f32x2*>(fd + 2 * i16) = f32x2{dzf[0], dzf[1]};
                else fd[i16] = dzf[0];
            } else {
                const int j = i - NWR;
                if (j < NT) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(aimg + (16 * j + i16) * C::PR + 4 * g);
#pragma unroll
                    for (int s = 0; s < 4; ++s) Bop[j][s] = b[s];
                } else if (j < NT + MT) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(dimg + (16 * (j - NT) + i16) * C::PR + 4 * g);
#pragma unroll
                    for (int s = 0; s < 4; ++s) Aop[j - NT][s] = a[s];
                } else {
                    const int h = j - NT - MT;
                    const f32x4 u = *reinterpret_cast<const f32x4*>(fd + (NF == 2 ? 8 : 4) * g + 4 * h);
#pragma unroll
                    for (int s = 0; s < 4; ++s) Fop[4 * h + s] = u[s];
                }
            }
        }
    }
    static constexpr int NMF = MT > 0 ? 4 * MT * NTM : 0;
    static __device__ __forceinline__ void dw_step(int j, f32x4 (&dW)[C::DW3_TILES > 0 ? C::DW3_TILES : 1], const float (&Aop)[MTd][4],
                                                    const float (&Bop)[NT][4]) {
        if constexpr (MT > 0) {
            const int s = j / (MT * NTM), mt = (j / NTM) % MT, nt = j % NTM;
            if ((TBNN_SKEL & 8) && NT == 4 && nt == 3) return;
            mfma16_acc<(MT * NTM > 2)>(dW[C::dwoff3(l) + mt * NTM + nt], Aop[mt][s], Bop[nt][s]);
        }
    }
    // ---- N-side fringe of dW_l (F3Cfg::NCF): its operands go from LDS straight into AccVGPRs (an MFMA's A / B operands may be
    // AccVGPRs on gfx950; the ~250 ArchVGPRs of this kernel have no room for 32 more):
    //   A[q][s]: lane L = delta_l[unit L][row 4q + s]      -- the delta image row L (units 0 .. 16 MTF - 1; lanes above: unused blocks)
    //   B[q][s]: lane L = a_{l-1}[slot 16 (NT-1) + 4 (L & 3)][row 4q + s]   -- units 48, 49, the ones slot, a zero slot
    // nf_load runs once layer l's delta image is complete (issue / issue_step), nf_mfma a few dozen vector instructions later.
    // Inline-asm LDS reads are invisible to the compiler's wait-count bookkeeping: nf_mfma waits for lgkmcnt(0) itself (extra
    // outstanding reads only make the compiler's own counted waits stronger: LDS returns in order).
    struct NFOps { f32x4 A[4], B[4]; };
    static __device__ __forceinline__ void nf_load(NFOps& o, const float* wl, int lane) {
        if constexpr (NCF) {
            const float* da = wl + C::doff3 + lane * C::PR;
            const float* ab = wl + C::aoff3(l) + (16 * (NT - 1) + 4 * (lane & 3)) * C::PR;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(o.A[q]) : "v"((unsigned)(size_t)da), "n"(16 * q) : "memory");
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(o.B[q]) : "v"((unsigned)(size_t)ab), "n"(16 * q) : "memory");
            }
        }
    }
    static __device__ __forceinline__ void nf_mfma(f32x4 (&dW)[C::DW3_TILES > 0 ? C::DW3_TILES : 1], NFOps& o) {
        if constexpr (NCF) {
            // two rows per statement, one per accumulator, and the wait states a dependent 4x4x1 needs written out (two: what the
            // compiler keeps between dependent 4x4x1 builtins): the next statement accumulates into acc0 behind this one's second MFMA
            // and the s_nop, whatever the compiler puts between the statements
            f32x4& acc0 = dW[C::dwfr3(l)];
            f32x4& acc1 = dW[C::dwfr3(l) + 1];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int s = 0; s < 4; s += 2) {
                    if (q == 0 && s == 0)
                        asm volatile("s_waitcnt lgkmcnt(0)\n\tv_mfma_f32_4x4x1_16b_f32 %0, %2, %3, %0\n\tv_mfma_f32_4x4x1_16b_f32 %1, %4, %5, %1\n\ts_nop 0"
                                     : "+a"(acc0), "+a"(acc1) : "a"(o.A[q][s]), "a"(o.B[q][s]), "a"(o.A[q][s + 1]), "a"(o.B[q][s + 1]));
                    else
                        asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %2, %3, %0\n\tv_mfma_f32_4x4x1_16b_f32 %1, %4, %5, %1\n\ts_nop 0"
                                     : "+a"(acc0), "+a"(acc1) : "a"(o.A[q][s]), "a"(o.B[q][s]), "a"(o.A[q][s + 1]), "a"(o.B[q][s + 1]));
                }
        }
    }
    // fringe rows of dW_l in the B-operand layout: lane (i16, g) sums delta_f[row 4g+s] * a_{l-1}[slot 16nt+i16][row 4g+s]
    // over its 4 rows; the epilogue adds the 4 lane groups.  NF == 2: one v_pk_fma per (nt, s) for both units;
    // NF == 1: pairs of rows.
    static __device__ __forceinline__ void fdw(float (&FP)[C::FP_REGS > 0 ? C::FP_REGS : 1], const float (&Fop)[8],
                                                const float (&Bop)[NT][4]) {
        if constexpr (C::FB(l) && !(TBNN_SKEL & 2)) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x2 acc = {FP[C::fpoff(l) + 2 * nt], FP[C::fpoff(l) + 2 * nt + 1]};
                const f32x2 b01 = {Bop[nt][0], Bop[nt][1]}, b23 = {Bop[nt][2], Bop[nt][3]};
                if constexpr (NF == 2) {
                    acc = pkfma_bc<0>(f32x2{Fop[0], Fop[1]}, b01, acc);
                    acc = pkfma_bc<1>(f32x2{Fop[2], Fop[3]}, b01, acc);
                    acc = pkfma_bc<0>(f32x2{Fop[4], Fop[5]}, b23, acc);
                    acc = pkfma_bc<1>(f32x2{Fop[6], Fop[7]}, b23, acc);
                } else {
                    acc = pkfma(f32x2{Fop[0], Fop[1]}, b01, acc);
                    acc = pkfma(f32x2{Fop[2], Fop[3]}, b23, acc);
                }
                FP[C::fpoff(l) + 2 * nt] = acc[0];
                FP[C::fpoff(l) + 2 * nt + 1] = acc[1];
            }
        }
    }
    static __device__ __forceinline__ void dw(f32x4 (&dW)[C::DW3_TILES > 0 ? C::DW3_TILES : 1], const float (&Aop)[MTd][4],
                                               const float (&Bop)[NT][4]) {
        if constexpr (MT > 0) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NTM; ++nt)
                        if (!((TBNN_SKEL & 8) && NT == 4 && nt == 3))
                            mfma16_acc<(MT * NTM > 2)>(dW[C::dwoff3(l) + mt * NTM + nt], Aop[mt][s], Bop[nt][s]);
        }
    }
    // delta_{l-1} (full tiles + fringe) from delta_l
    static __device__ __forceinline__ void da(const Tile3<S>& T, const float* __restrict__ lds, int i16, int g,
                                               const f32x4 (&dz)[C::MT(l)], const float (&dzf)[NFd],
                                               f32x4 (&dzp)[C::MT(l > 0 ? l - 1 : 0)], float (&dzpf)[NFd]) {
        if constexpr (l > 0) {
            constexpr int MTP = C::MTF(l - 1), NFP = C::NF(l - 1), K = C::out(l);
#if TBNN_F3_HAND
            if constexpr (MT > 0) {
                // hand-scheduled chain (see Fwd3::run_h): k-step-major MFMAs, the fringe 4x4x1 behind each k-step, the next
                // k-group's operands (MTP tiles of W^T + the fringe rows) one read per MFMA under the first MFMAs of the group
                static_assert(TBNN_F3_M4, "the hand-scheduled form computes the fringe units on the 4x4x1 MFMA");
                constexpr int KG = C::cdiv(K, 16), MTPd = MTP > 0 ? MTP : 1, NSL = MTP + (NFP > 0 ? 1 : 0);
                constexpr int NA = KG < TBNN_F3_M4ACC ? KG : TBNN_F3_M4ACC;
                f32x4 acc[MTPd], pa[NA], An[MTPd], Fn = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int m = 0; m < MTP; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int a = 0; a < NA; ++a) pa[a] = f32x4{0.f, 0.f, 0.f, 0.f};
                const float* trow = lds + C::toff(l) + i16 * C::LDT(l) + 4 * g;
                const float* frow = lds + C::toff(l) + (16 * MTP + 4 * (i16 & 3)) * C::LDT(l) + 4 * g;
#pragma unroll
                for (int m = 0; m < MTP; ++m) An[m] = load_ks(trow + 16 * m * C::LDT(l), C::ksteps(K, 0));
                if constexpr (NFP > 0) Fn = load_ks(frow, C::ksteps(K, 0));
                sfor3<0, KG>([&](auto kt_) __attribute__((always_inline)) {
                    constexpr int kt = decltype(kt_)::value, ns = C::ksteps(K, kt);
                    f32x4 A4[MTPd], F4 = Fn;
#pragma unroll
                    for (int m = 0; m < MTP; ++m) A4[m] = An[m];
                    constexpr int NLD = kt + 1 < KG ? NSL : 0;
                    auto issue = [&](auto q_) __attribute__((always_inline)) {
                        constexpr int q = decltype(q_)::value;
                        if constexpr (q < MTP) An[q] = load_ks(trow + 16 * q * C::LDT(l) + 16 * (kt + 1), C::ksteps(K, kt + 1));
                        else Fn = load_ks(frow + 16 * (kt + 1), C::ksteps(K, kt + 1));
                    };
                    __builtin_amdgcn_sched_barrier(0);
                    sfor3<0, ns>([&](auto s_) __attribute__((always_inline)) {
                        constexpr int s = decltype(s_)::value;
                        sfor3<0, NSL>([&](auto m_) __attribute__((always_inline)) {
                            constexpr int m = decltype(m_)::value, q = s * NSL + m;
                            if constexpr (m < MTP) acc[m] = mfma16(A4[m][s], dz[kt][s], acc[m]);
                            else pa[kt % NA] = mfma4(F4[s], dz[kt][s], pa[kt % NA]);
                            if constexpr (q < NLD) { issue(std::integral_constant<int, q>{}); __builtin_amdgcn_sched_barrier(0); }
                        });
                        __builtin_amdgcn_sched_barrier(0);
                    });
                    sfor3<ns * NSL, NLD>([&](auto q_) __attribute__((always_inline)) { issue(q_); });
                    __builtin_amdgcn_sched_barrier(0);
                });
                f32x4 pacc = pa[0];
                if constexpr (NFP > 0) {
#pragma unroll
                    for (int st = 1; st < NA; st *= 2)
#pragma unroll
                        for (int a = 0; a + st < NA; a += 2 * st) pa[a] += pa[a + st];
                    pacc = pa[0];
                }
                if constexpr (S::act(l - 1) == TBNN_ACT_RELU && TBNN_F3_RELU_PK && MTP >= 1 && MTP <= 4) mfma_settle(acc);
#pragma unroll
                for (int m = 0; m < MTP; ++m)
                    dzp[m] = actc_bwd_mul4<S::act(l - 1), (MTP >= 1 && MTP <= 4)>(acc[m], T.a[C::aroff(l - 1) + m]);
#pragma unroll
                for (int f = 0; f < NFP; ++f)
                    dzpf[f] = gsum_mfma(pacc[f], [&](float s) { return actc_bwd_mul<S::act(l - 1)>(s, T.af[l - 1][f]); });
#else
            // fringe units of layer l-1 first (their shuffles land under the MFMAs)
            float pf[NFP > 0 ? NFP : 1];
            f32x4 pacc = {0.f, 0.f, 0.f, 0.f};
            if constexpr (MT > 0 && NFP > 0 && TBNN_F3_M4)
                pacc = fringe_partials<S, K>(lds + C::toff(l) + (16 * MTP + 4 * (i16 & 3)) * C::LDT(l) + 4 * g, dz);
            if constexpr (MT > 0 && !TBNN_F3_M4) {
#pragma unroll
                for (int f = 0; f < NFP; ++f)
                    pf[f] = dot_slots<S, K, S::act(l) != TBNN_ACT_NONE>(lds + C::toff(l) + C::fslot(l - 1, f) * C::LDT(l), dz, g);
            }
            if constexpr (MT > 0) {
                constexpr int KG = C::cdiv(K, 16);
                f32x4 acc[MTP > 0 ? MTP : 1];
#pragma unroll
                for (int m = 0; m < MTP; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
                const float* trow = lds + C::toff(l) + i16 * C::LDT(l) + 4 * g;
                f32x4 An[MTP > 0 ? MTP : 1];
#pragma unroll
                for (int m = 0; m < MTP; ++m) An[m] = load_ks(trow + 16 * m * C::LDT(l), C::ksteps(K, 0));
#pragma unroll
                for (int kt = 0; kt < KG; ++kt) {
                    f32x4 A4[MTP > 0 ? MTP : 1];
#pragma unroll
                    for (int m = 0; m < MTP; ++m) A4[m] = An[m];
                    if (kt + 1 < KG) {
#pragma unroll
                        for (int m = 0; m < MTP; ++m) An[m] = load_ks(trow + 16 * m * C::LDT(l) + 16 * (kt + 1), C::ksteps(K, kt + 1));
                    }
#pragma unroll
                    for (int s = 0; s < C::ksteps(K, kt); ++s)
#pragma unroll
                        for (int m = 0; m < MTP; ++m) acc[m] = mfma16(A4[m][s], dz[kt][s], acc[m]);
                }
                if constexpr (S::act(l - 1) == TBNN_ACT_RELU && TBNN_F3_RELU_PK && MTP >= 1 && MTP <= 4) mfma_settle(acc);
#pragma unroll
                for (int m = 0; m < MTP; ++m)
                    dzp[m] = actc_bwd_mul4<S::act(l - 1), (MTP >= 1 && MTP <= 4)>(acc[m], T.a[C::aroff(l - 1) + m]);
#pragma unroll
                for (int f = 0; f < NFP; ++f)
                    {
                    if constexpr (TBNN_F3_M4) dzpf[f] = gsum_mfma(pacc[f], [&](float s) { return actc_bwd_mul<S::act(l - 1)>(s, T.af[l - 1][f]); });
                    else dzpf[f] = actc_bwd_mul<S::act(l - 1)>(gsum(pf[f]), T.af[l - 1][f]);
                }
#endif
            } else {
                // all-fringe layer (the VALU last layer): K = NF fringe deltas, weights W_l[o][slot] read per lane
#pragma unroll
                for (int m = 0; m < MTP; ++m) {
                    f32x2 d01, d23;
#pragma unroll
                    for (int o = 0; o < NF; ++o) {
                        const f32x4 w = *reinterpret_cast<const f32x4*>(lds + C::woff(l) + C::fslot(l, o) * C::LDW(l) + 16 * m + 4 * g);
                        const f32x2 dd = {dzf[o], dzf[o]};
                        if (o == 0) { d01 = pkmul_bc<0>(f32x2{w[0], w[1]}, dd); d23 = pkmul_bc<0>(f32x2{w[2], w[3]}, dd); }
                        else { d01 = pkfma_bc<0>(f32x2{w[0], w[1]}, dd, d01); d23 = pkfma_bc<0>(f32x2{w[2], w[3]}, dd, d23); }
                    }
                    dzp[m] = actc_bwd_mul4<S::act(l - 1), true>(f32x4{d01[0], d01[1], d23[0], d23[1]}, T.a[C::aroff(l - 1) + m]);
                }
#pragma unroll
                for (int f = 0; f < NFP; ++f) {
                    float d = 0.f;
#pragma unroll
                    for (int o = 0; o < NF; ++o) d = fmaf(lds[C::woff(l) + C::fslot(l, o) * C::LDW(l) + C::fslot(l - 1, f)], dzf[o], d);
                    dzpf[f] = actc_bwd_mul<S::act(l - 1)>(d, T.af[l - 1][f]);
                }
            }
            // fringe tile of delta_{l-1} (K operand of the next delta step)
            if constexpr (NFP > 0) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int f = 0; f < NFP; ++f)
                    if (g == f) v[0] = dzpf[f];
                dzp[MTP] = v;
            }
        }
    }
};

// software pipeline as BwdPipe: W(D_l) R(op_l) | dW_{l+1} (MFMA) | dA_l | fringe dW_l | recurse
template <class S, int l>
struct Pipe3 {
    using C = F3Cfg<S>;
    static constexpr int NFd = C::maxNF() > 0 ? C::maxNF() : 1;
    static __device__ __forceinline__ void run(f32x4 (&dW)[C::DW3_TILES > 0 ? C::DW3_TILES : 1], float (&FP)[C::FP_REGS > 0 ? C::FP_REGS : 1],
                                                const Tile3<S>& T, const float* __restrict__ lds, float* wl, int i16, int g,
                                                const f32x4 (&dz)[C::MT(l)], const float (&dzf)[NFd],
                                                const float (&Aup)[Bwd3<S, l + 1>::MTd][4], const float (&Bup)[C::NT(l + 1)][4],
                                                const float (&Fup)[8]) {
        float Aop[Bwd3<S, l>::MTd][4], Bop[C::NT(l)][4], Fop[8];
        typename Bwd3<S, l>::NFOps NFop;
        static_assert(!(C::NCF(l) || C::NCF(l + 1)) || (TBNN_ACC_AGPR && TBNN_F3_THREAD), "the N-side fringe rides in the hand-threaded pipeline");
#if !(TBNN_ACC_AGPR && TBNN_F3_THREAD)
        Bwd3<S, l>::issue(dz, dzf, wl, i16, g, Aop, Bop, Fop);
#endif
#if TBNN_SGB > 0 && !TBNN_ACC_AGPR
        // the ~20 LDS instructions of issue(l) cost ~450 issue cycles on their own: thread them through the
        // 48 MFMAs of dW_{l+1} (operands already in registers) -- 2 MFMAs, 1 DS, 2 MFMAs, 1 DS, ...
        Bwd3<S, l + 1>::dw(dW, Aup, Bup);
        {
            constexpr int NDS = Bwd3<S, l>::MT > 0 ? 4 * Bwd3<S, l>::MT + C::NT(l) + Bwd3<S, l>::MT : 0;
#pragma unroll
            for (int i = 0; i < NDS; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, TBNN_SGB, 0);
                __builtin_amdgcn_sched_group_barrier(0x080, 1, 0);
            }
        }
        SCHED_FENCE();
        TSTAMP(10 + 4 * l);
#elif TBNN_ACC_AGPR && TBNN_F3_THREAD
        {
            constexpr int NM = Bwd3<S, l + 1>::NMF, NL = Bwd3<S, l>::NLDS;
            static_assert(NM > 0, "the layer above has an MFMA dW part");
            constexpr int NWRl = Bwd3<S, l>::NWR;
            constexpr int JNF = NL > 0 ? ((NWRl * NM + NL - 1) / NL < NM ? (NWRl * NM + NL - 1) / NL : NM - 1) : NM - 1;   // first j whose steps include the last write
#pragma unroll
            for (int j = 0; j < NM; ++j) {
                Bwd3<S, l + 1>::dw_step(j, dW, Aup, Bup);
#pragma unroll
                for (int k = (j * NL) / NM; k < ((j + 1) * NL) / NM; ++k) Bwd3<S, l>::issue_step(k, dz, dzf, wl, i16, g, Aop, Bop, Fop);
                // once this layer's delta image is complete (the image writes are the first NWR issue steps) its N-fringe operands
                // leave for the AccVGPRs: they arrive under the rest of the dW MFMAs and are used behind the packed FMAs of the fringe
                // rows below (held any longer -- over the delta step -- the register allocator starts moving dW tiles between the two
                // register files)
                if (j == JNF) Bwd3<S, l>::nf_load(NFop, wl, 16 * g + i16);
            }
        }
        SCHED_FENCE();
        TSTAMP(10 + 4 * l);
#else
        SCHED_FENCE();
        TSTAMP(10 + 4 * l);
        Bwd3<S, l + 1>::dw(dW, Aup, Bup);
        SCHED_FENCE();
#endif
        TSTAMP(11 + 4 * l);
        Bwd3<S, l + 1>::fdw(FP, Fup, Bup);
        FringeDW<S, l>::run(FP, T, dzf, g);
        Bwd3<S, l>::nf_mfma(dW, NFop);
        TSTAMP(12 + 4 * l);
        if constexpr (l > 0) {
            f32x4 dzp[C::MT(l - 1)];
            float dzpf[NFd];
            Bwd3<S, l>::da(T, lds, i16, g, dz, dzf, dzp, dzpf);
            SCHED_FENCE();
            TSTAMP(13 + 4 * l);
            Pipe3<S, l - 1>::run(dW, FP, T, lds, wl, i16, g, dzp, dzpf, Aop, Bop, Fop);
        } else {
            Bwd3<S, 0>::dw(dW, Aop, Bop);
            Bwd3<S, 0>::fdw(FP, Fop, Bop);
            TSTAMP(13);
        }
    }
};

// slab write-out of the MFMA dW tiles (rows = full tiles only)
template <class S, int l>
struct SlabOut3 {
    using C = F3Cfg<S>;
    // LD: slab is the workgroup's dense copy in LDS (plain stores; the kernel writes it out 16 bytes per lane)
    template <bool LD = false>
    static __device__ __forceinline__ void run(const float* buf, float* __restrict__ slab, int wave, int lane, int t0, int cnt) {
        constexpr int in = C::in(l), out = C::out(l), MT = C::MTF(l), NT = C::NTF(l);
        if constexpr (C::NCF(l)) {
            // the N-fringe tile: lane (b, j) register i = dW[unit 4b + i][slot 16 (NT_all - 1) + 4j] (units of the full tiles: slot == unit)
            const int t = C::dwfr3(l) - t0;
            static_assert(C::EP3_TILES == C::DW3_TILES || !C::NCF(l), "the N-fringe pair is staged in one pass");
            if (t >= 0 && t + 1 < cnt && (t & (FAST_WAVES - 1)) == wave) {
                const f32x4* src = reinterpret_cast<const f32x4*>(buf) + t * 64 + lane;
                // (even-row and odd-row accumulators: tiles t and t + 1)
                const f32x4 c0 = src[0] + src[64], c1 = src[C::EP3_TILES * 64] + src[C::EP3_TILES * 64 + 64],
                            c2 = src[2 * C::EP3_TILES * 64] + src[2 * C::EP3_TILES * 64 + 64], c3 = src[3 * C::EP3_TILES * 64] + src[3 * C::EP3_TILES * 64 + 64];
                const int b = lane >> 2, col = unit_of(in, 16 * (C::NT(l) - 1) + 4 * (lane & 3), true);
                if (b < 4 * MT && col >= 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = unit_of(out, 4 * b + r, false);           // (a partly filled last M tile keeps its units in slot order)
                        if (row >= 0)
                            slab_store<(!LD && C::P() >= 2048)>(slab + C::offW(l) + (col < in ? row * in + col : in * out + row), (c0[r] + c1[r]) + (c2[r] + c3[r]));
                    }
                }
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int t = C::dwoff3(l) + mt * NT + nt - t0;
                if (t >= 0 && t < cnt && (t & (FAST_WAVES - 1)) == wave) {
                    const f32x4* src = reinterpret_cast<const f32x4*>(buf) + t * 64 + lane;
                    const f32x4 c0 = src[0], c1 = src[C::EP3_TILES * 64], c2 = src[2 * C::EP3_TILES * 64], c3 = src[3 * C::EP3_TILES * 64];
                    const int cs = 16 * nt + (lane & 15), row0 = 16 * mt + 4 * (lane >> 4);
                    const int col = l == 0 ? (cs <= in ? cs : -1) : unit_of(in, cs, true);
                    if (col >= 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = unit_of(out, row0 + r, false);
                            if (row >= 0)
                                slab_store<(!LD && C::P() >= 2048)>(slab + C::offW(l) + (col < in ? row * in + col : in * out + row), (c0[r] + c1[r]) + (c2[r] + c3[r]));
                        }
                    }
                }
            }
        if constexpr (l + 1 < C::NLM3) SlabOut3<S, l + 1>::template run<LD>(buf, slab, wave, lane, t0, cnt);
    }
};

// fringe partials -> slab: staged [wave][reg][i16] (lane group g selects which k-slots the register means)
template <class S, int l>
struct FringeOut {
    using C = F3Cfg<S>;
    // staging: lb[(wave * FP_REGS + reg) * 64 + lane]
    template <bool LD = false>
    static __device__ __forceinline__ void run(const float* lb, float* __restrict__ slab, int tid) {
        constexpr int NF = C::NF(l), KIN = C::KIN(l), in = C::in(l), out = C::out(l);
        if constexpr (C::FB(l)) {
            // B-operand layout: item = (f, nt, i16): sum over the 4 waves x 4 lane groups (x 2 row pairs when NF == 1)
            constexpr int NT = C::NT(l), NI = NF * NT * 16;
            for (int it = tid; it < NI; it += FAST_THREADS) {
                const int i16 = it & 15, nt = (it >> 4) % NT, f = it / (16 * NT);
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < FAST_WAVES; ++w) {
                    float vw = 0.f;
#pragma unroll
                    for (int gg = 0; gg < 4; ++gg) {
                        if constexpr (NF == 2) vw += lb[(size_t)(w * C::FP_REGS + C::fpoff(l) + 2 * nt + f) * 64 + 16 * gg + i16];
                        else vw += lb[(size_t)(w * C::FP_REGS + C::fpoff(l) + 2 * nt) * 64 + 16 * gg + i16] +
                                   lb[(size_t)(w * C::FP_REGS + C::fpoff(l) + 2 * nt + 1) * 64 + 16 * gg + i16];
                    }
                    v += vw;
                }
                const int cs = 16 * nt + i16, u = C::funit(l, f);
                const int col = l == 0 ? (cs <= in ? cs : -1) : unit_of(in, cs, true);
                if (col >= 0) slab_store<(!LD && C::P() >= 2048)>(slab + C::offW(l) + (col < in ? u * in + col : in * out + u), v);
            }
        } else
        // D layout: thread (item = (f, kslot, g), wave): item count per layer = NF * (KIN+1) * 4 lane groups
        if constexpr (NF > 0) {
            constexpr int NI = NF * (KIN + 1) * 4;
            for (int base = 0; base < NI * FAST_WAVES; base += FAST_THREADS) {
                const int t = base + tid;
                const int it = t >> 2, w = t & 3;
                float v = 0.f;
                int dest = -1;
                if (it < NI) {
                    const int gg = it & 3, fk = it >> 2;
                    const int f = fk / (KIN + 1), k = fk - f * (KIN + 1);
                    const int reg = C::fpoff(l) + f * (KIN + 1) + k;
                    const f32x4* src = reinterpret_cast<const f32x4*>(lb + ((size_t)(w * C::FP_REGS + reg) * 4 + gg) * 16);
                    const f32x4 p0 = src[0], p1 = src[1], p2 = src[2], p3 = src[3];
                    v = ((p0[0] + p0[1]) + (p0[2] + p0[3])) + ((p1[0] + p1[1]) + (p1[2] + p1[3])) +
                        (((p2[0] + p2[1]) + (p2[2] + p2[3])) + ((p3[0] + p3[1]) + (p3[2] + p3[3])));
                    const int u = C::funit(l, f);
                    if (k == KIN) { if (gg == 0) dest = C::offW(l) + in * out + u; }
                    else {
                        // k-slot -> input unit: layer 0: x unit 4k+g; else slot 16*(k/4) + 4g + k%4
                        const int col = l == 0 ? (4 * k + gg < in ? 4 * k + gg : -1) : unit_of(in, 16 * (k >> 2) + 4 * gg + (k & 3), false);
                        if (col >= 0) dest = C::offW(l) + u * in + col;
                    }
                }
                v += __shfl_xor(v, 1, 64);
                v += __shfl_xor(v, 2, 64);
                if (dest >= 0 && w == 0) slab_store<(!LD && C::P() >= 2048)>(slab + dest, v);
            }
        }
        if constexpr (l + 1 < C::NL) FringeOut<S, l + 1>::template run<LD>(lb, slab, tid);
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// Cooperative tail.  The tile loop hands every wave whole 16-row tiles; n = 100,000 rows are 6250 tiles = 6 rounds of
// 1024 waves + 106 tiles, and a 7th round with 106 of 1024 waves busy costs a full tile time (12 % of the row loop).
// The left-over tiles (at most 2 per workgroup) are instead run by the FOUR waves of a workgroup together, the unit
// dimension split over the waves: wave w owns M tile w of every hidden layer (the last wave the fringe units),
//   forward   own tile of a_l = act(W_l[tile w] a_{l-1}) over the full K -> exchanged through LDS (1 KB per tile, one
//             barrier), after which every wave holds all of a_l in the D layout exactly as the tile loop would;
//   last layer, likelihood and delta_{L-1} (VALU, a few dozen instructions) redundantly on every wave;
//   delta     own tile of delta_{l-1} = act' (W_l^T[tile w] delta_l), exchanged when another delta step follows;
//   dW        wave w: the N tiles of its own M tile (A operand: its delta tile through its own transposed image),
//             into a small separate accumulator set dWc that the epilogue adds to the wave's staged tiles; the fringe
//             owner: the fringe rows (fdw); wave 0: the last layer's per-lane sums.
// A wave issues ~90 MFMAs instead of ~350 and meets 4 barriers: ~0.4 of a tile time, for every workgroup at once.
// Shapes with more than FAST_WAVES M tiles in a hidden layer (or no room for the 8.4-KB exchange buffers) keep the plain loop.
// ---------------------------------------------------------------------------------------------------------------------
#ifndef TBNN_F3_COOP
#define TBNN_F3_COOP 1
#endif

template <class S>
struct Coop3 {
    using C = F3Cfg<S>;
    static constexpr int L = C::NL - 1;                                   // the all-fringe last layer
    static constexpr int NFd = C::maxNF() > 0 ? C::maxNF() : 1;
    static constexpr int HL = L > 0 ? L : 1;                              // hidden layers
    static constexpr bool shape_ok() {
        if (C::NL < 2 || !TBNN_F3_M4 || !TBNN_F3_FB) return false;
        for (int l = 0; l < L; ++l) if (C::MT(l) > FAST_WAVES || C::MTF(l) < 1) return false;
        return true;
    }
    static constexpr int XAF = FAST_WAVES * 256;                          // fringe values behind the 4 tiles of an exchange buffer
    static constexpr int XB = XAF + 32;                                   // floats per exchange buffer; two alternate
    static constexpr bool ENABLED = TBNN_F3_COOP && shape_ok() && ((size_t)C::LDS3_FLOATS + 2 * XB) * 4 + 64 <= 160 * 1024;
    static constexpr int cwoff(int l) { int o = 0; for (int m = 0; m < l && m < C::NLM3; ++m) o += C::MTF(m) > 0 ? C::NT(m) : 0; return o; }
    static constexpr int DWC = cwoff(C::NL) > 0 ? cwoff(C::NL) : 1;
    static constexpr int FPd = C::FP_REGS > 0 ? C::FP_REGS : 1;

    // the transposed image of layer l's output (input of layer l+1) for the dW MFMAs, as Fwd3::run writes it
    template <int l>
    static __device__ __forceinline__ void image(const Tile3<S>& T, float* wl, int i16, int g) {
        if constexpr (l + 1 < C::NL && C::MTF(l + 1 < C::NL ? l + 1 : l) > 0) {
            constexpr int u1 = C::in(l + 1);
            float* aimg = wl + C::aoff3(l + 1);
#pragma unroll
            for (int mt = 0; mt < C::MT(l); ++mt) {
                f32x4 v = T.a[C::aroff(l) + mt];
                if constexpr (u1 % 16 != 0) {
                    constexpr int osl = ones_slot(u1);
                    if (mt == osl / 16 && g == (osl % 16) / 4) v[osl % 4] = 1.f;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) aimg[(16 * mt + 4 * g + r) * C::PR + i16] = v[r];
            }
        }
    }

    // hidden layer l: own tile, exchange, all tiles
    template <int l>
    static __device__ __forceinline__ void fwd(Tile3<S>& T, f32x4 (&own)[HL], const float* __restrict__ lds, float* wl, float* xch, int& xsel,
                                                int wave, int lane, int i16, int g) {
        constexpr int MT = C::MTF(l), NF = C::NF(l);
        float* xb = xch + xsel * XB;
        xsel ^= 1;
        f32x4 mine = {0.f, 0.f, 0.f, 0.f};
        if (wave < MT) {
            f32x4 acc = *reinterpret_cast<const f32x4*>(lds + C::boff(l) + 16 * wave + 4 * g);
            if constexpr (l == 0) {
#pragma unroll
                for (int t = 0; t < C::KS0; ++t) acc = mfma16(lds[C::woff(0) + (16 * wave + i16) * C::LDW(0) + 4 * t + g], T.x0[t], acc);
            } else {
                // one M tile per wave: a single chain of 13 MFMAs would run at the 40-cycle dependent latency; two
                // accumulators (even / odd k-groups) keep it at the 32-cycle issue rate
                const float* wrow = lds + C::woff(l) + (16 * wave + i16) * C::LDW(l) + 4 * g;
                f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kt = 0; kt < C::KG(l); ++kt) {
                    const f32x4 A = load_ks(wrow + 16 * kt, C::ksteps(C::in(l), kt));
#pragma unroll
                    for (int s = 0; s < C::ksteps(C::in(l), kt); ++s) {
                        if (kt & 1) acc1 = mfma16(A[s], T.a[C::aroff(l - 1) + kt][s], acc1);
                        else acc = mfma16(A[s], T.a[C::aroff(l - 1) + kt][s], acc);
                    }
                }
                acc += acc1;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[r] = actc_fwd<S::act(l)>(acc[r]);
            *reinterpret_cast<f32x4*>(xb + wave * 256 + lane * 4) = mine;
        } else if (NF > 0 && wave == MT) {
            const float* prow = lds + C::woff(l) + (16 * MT + 4 * (i16 & 3)) * C::LDW(l);
            f32x4 pacc = {0.f, 0.f, 0.f, 0.f};
            if constexpr (l == 0) {
#pragma unroll
                for (int t = 0; t < C::KS0; ++t) pacc = mfma4(prow[4 * t + g], T.x0[t], pacc);
            } else {
                pacc = fringe_partials<S, C::in(l)>(prow + 4 * g, &T.a[C::aroff(l > 0 ? l - 1 : 0)]);
            }
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const float a = gsum_mfma(pacc[f], [&](float s) { return actc_fwd<S::act(l)>(s + lds[C::boff(l) + C::fslot(l, f)]); });
                if (g == f) mine[0] = a;
                if (g == 0) xb[XAF + 16 * f + i16] = a;
            }
            *reinterpret_cast<f32x4*>(xb + MT * 256 + lane * 4) = mine;
        }
        own[l] = mine;
        TSTAMP(41 + 3 * l);
        __syncthreads();
        TSTAMP(42 + 3 * l);
#pragma unroll
        for (int mt = 0; mt < C::MT(l); ++mt) T.a[C::aroff(l) + mt] = *reinterpret_cast<const f32x4*>(xb + mt * 256 + lane * 4);
#pragma unroll
        for (int f = 0; f < NF; ++f) T.af[l][f] = xb[XAF + 16 * f + i16];
        image<l>(T, wl, i16, g);
        TSTAMP(43 + 3 * l);
        if constexpr (l + 1 < L) fwd<l + 1>(T, own, lds, wl, xch, xsel, wave, lane, i16, g);
    }

    // layer l backward: dW_l for the own M tile (own_dz: this wave's tile of delta_l; dzf: the fringe deltas, valid on the
    // fringe owner), then the own tile of delta_{l-1} from ALL of delta_l (dz)
    template <int l>
    static __device__ __forceinline__ void bwd(f32x4 (&dWc)[DWC], float (&FP)[FPd], const Tile3<S>& T, const f32x4 (&own)[HL],
                                                const float* __restrict__ lds, float* wl, float* xch, int& xsel, int wave, int lane, int i16, int g,
                                                const f32x4 (&dz)[C::MT(l)], const f32x4 own_dz, const float (&dzf)[NFd]) {
        constexpr int MT = C::MTF(l), NT = C::NT(l), NF = C::NF(l);
        if constexpr (MT > 0) {
            float* dimg = wl + C::doff3;
            const float* aimg = wl + C::aoff3(l);
            const bool fown = C::FB(l) && wave == MT;
            if (wave < MT || fown) {
                float Bop[NT][4];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(aimg + (16 * nt + i16) * C::PR + 4 * g);
#pragma unroll
                    for (int s = 0; s < 4; ++s) Bop[nt][s] = b[s];
                }
                if (wave < MT) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) dimg[(16 * wave + 4 * g + r) * C::PR + i16] = own_dz[r];
                    const f32x4 a = *reinterpret_cast<const f32x4*>(dimg + (16 * wave + i16) * C::PR + 4 * g);
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) dWc[cwoff(l) + nt] = mfma16(a[s], Bop[nt][s], dWc[cwoff(l) + nt]);
                    // These MFMAs END a wave-uniform block, and at the join behind it the compiler may move their results (registers differ
                    // between the paths) -- without the wait states an MFMA result needs: its hazard recognizer does not look across the
                    // join (13 -> 36 -> 16 -> 33 -> 32 -> 2, second cooperative tile: `v_mfma v[52:55]` ... two branches ... `v_mov_b64 v[90:91],
                    // v[54:55]`: the bias column of dW_2 wrong in registers 2, 3; found by narrow_fuzz.py, seen by hazard_lint.py once it
                    // followed the control flow).  The results are settled before the block ends.
                    coop_settle<NT>(&dWc[cwoff(l)]);
                } else {
                    float* fd = wl + C::fdoff3 + 32 * g;
                    float Fop[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    if constexpr (NF == 2) {
                        *reinterpret_cast<f32x2*>(fd + 2 * i16) = f32x2{dzf[0], dzf[NF > 1 ? 1 : 0]};
                        const f32x4 u = *reinterpret_cast<const f32x4*>(fd + 8 * g), v = *reinterpret_cast<const f32x4*>(fd + 8 * g + 4);
#pragma unroll
                        for (int s = 0; s < 4; ++s) { Fop[s] = u[s]; Fop[4 + s] = v[s]; }
                    } else {
                        fd[i16] = dzf[0];
                        const f32x4 u = *reinterpret_cast<const f32x4*>(fd + 4 * g);
#pragma unroll
                        for (int s = 0; s < 4; ++s) Fop[s] = u[s];
                    }
                    Bwd3<S, l>::fdw(FP, Fop, Bop);
                }
            }
        }
        if constexpr (NF > 0 && !C::FB(l)) {
            if (wave == MT) FringeDW<S, l>::run(FP, T, dzf, g);
        }
        TSTAMP(50 + 2 * (L - 1 - l));
        if constexpr (l > 0) {
            constexpr int MTP = C::MTF(l - 1), NFP = C::NF(l - 1), K = C::out(l);
            constexpr bool XCHG = l - 1 > 0;                              // delta_0 feeds no further delta step
            float* xb = xch + xsel * XB;
            if constexpr (XCHG) xsel ^= 1;
            f32x4 odz = {0.f, 0.f, 0.f, 0.f};
            float dzpf[NFd];
#pragma unroll
            for (int f = 0; f < NFd; ++f) dzpf[f] = 0.f;
            if (wave < MTP) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                const float* trow = lds + C::toff(l) + (16 * wave + i16) * C::LDT(l) + 4 * g;
#pragma unroll
                for (int kt = 0; kt < C::cdiv(K, 16); ++kt) {
                    const f32x4 A = load_ks(trow + 16 * kt, C::ksteps(K, kt));
#pragma unroll
                    for (int s = 0; s < C::ksteps(K, kt); ++s) {
                        if (kt & 1) acc1 = mfma16(A[s], dz[kt][s], acc1);
                        else acc = mfma16(A[s], dz[kt][s], acc);
                    }
                }
                acc += acc1;
                odz = actc_bwd_mul4<S::act(l - 1), false>(acc, own[l - 1]);
                if constexpr (XCHG) *reinterpret_cast<f32x4*>(xb + wave * 256 + lane * 4) = odz;
            } else if (NFP > 0 && wave == MTP) {
                const f32x4 pacc = fringe_partials<S, K>(lds + C::toff(l) + (16 * MTP + 4 * (i16 & 3)) * C::LDT(l) + 4 * g, dz);
#pragma unroll
                for (int f = 0; f < NFP; ++f) {
                    dzpf[f] = gsum_mfma(pacc[f], [&](float s) { return actc_bwd_mul<S::act(l - 1)>(s, T.af[l - 1][f]); });
                    if (g == f) odz[0] = dzpf[f];
                    if constexpr (XCHG) { if (g == 0) xb[XAF + 16 * f + i16] = dzpf[f]; }
                }
                if constexpr (XCHG) *reinterpret_cast<f32x4*>(xb + MTP * 256 + lane * 4) = odz;
            }
            f32x4 dzp[C::MT(l - 1)];
#pragma unroll
            for (int mt = 0; mt < C::MT(l - 1); ++mt) dzp[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (XCHG) {
                __syncthreads();
#pragma unroll
                for (int mt = 0; mt < C::MT(l - 1); ++mt) dzp[mt] = *reinterpret_cast<const f32x4*>(xb + mt * 256 + lane * 4);
#pragma unroll
                for (int f = 0; f < NFP; ++f) dzpf[f] = xb[XAF + 16 * f + i16];
            }
            TSTAMP(51 + 2 * (L - 1 - l));
            bwd<l - 1>(dWc, FP, T, own, lds, wl, xch, xsel, wave, lane, i16, g, dzp, odz, dzpf);
        }
    }

    // one 16-row tile on the four waves of the workgroup (every wave holds the tile's x / y of row i16)
    static __device__ __forceinline__ void tile(f32x4 (&dWc)[DWC], float (&FP)[FPd], double& stat, const float* __restrict__ lds, float* wl,
                                                 float* xch, int wave, int lane, int i16, int g, float inv_var,
                                                 const float (&x)[C::KS0], const float (&y)[C::out(C::NL - 1)], bool rvalid) {
        constexpr int d_in = C::in(0), d_out = C::out(C::NL - 1);
        Tile3<S> T;
        f32x4 own[HL];
        int xsel = 0;
        TSTAMPO(40);
#pragma unroll
        for (int t = 0; t < C::KS0; ++t) {
            T.x0[t] = x[t];
            const int u = 4 * t + g;
            if (u < d_in) wl[C::aoff3(0) + u * C::PR + i16] = x[t];
        }
        fwd<0>(T, own, lds, wl, xch, xsel, wave, lane, i16, g);
        {   // the all-fringe last layer: every wave (a dozen 4x4x1 MFMAs)
            const f32x4 dA[1] = {f32x4{0.f, 0.f, 0.f, 0.f}}, dB[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
            Fwd3<S, L>::run(T, lds, wl, i16, g, dA, dB);
        }
        TSTAMP(25);
        float dzf[NFd];
#pragma unroll
        for (int o = 0; o < NFd; ++o) dzf[o] = 0.f;
#pragma unroll
        for (int o = 0; o < d_out; ++o) dzf[o] = rvalid ? lik_delta<S>(T.af[L][o], y[o], inv_var, g == 0 && wave == 0, stat) : 0.f;
        if (wave == 0) FringeDW<S, L>::run(FP, T, dzf, g);
        f32x4 dzL[C::MT(L)];
#pragma unroll
        for (int mt = 0; mt < C::MT(L); ++mt) dzL[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 dzp[C::MT(L - 1)];
        float dzpf[NFd];
#pragma unroll
        for (int f = 0; f < NFd; ++f) dzpf[f] = 0.f;
        TSTAMP(26);
        Bwd3<S, L>::da(T, lds, i16, g, dzL, dzf, dzp, dzpf);             // delta_{L-1}: VALU, every tile on every wave
        TSTAMP(27);
        f32x4 odz = dzp[0];
#pragma unroll
        for (int mt = 1; mt < C::MTF(L - 1); ++mt) if (wave == mt) odz = dzp[mt];
        bwd<L - 1>(dWc, FP, T, own, lds, wl, xch, xsel, wave, lane, i16, g, dzp, odz, dzpf);
        TSTAMPO(56);
    }

    // epilogue: add this wave's cooperative accumulators to its staged tiles [t0, t0 + cnt) (mine[t - t0][lane])
    template <int l>
    static __device__ __forceinline__ void merge(f32x4* mine, const f32x4 (&dWc)[DWC], int wave, int lane, int t0, int cnt) {
        if constexpr (C::MTF(l) > 0) {
            if (wave < C::MTF(l)) {
#pragma unroll
                for (int nt = 0; nt < C::NTF(l); ++nt) {
                    const int t = C::dwoff3(l) + wave * C::NTF(l) + nt - t0;
                    if (t >= 0 && t < cnt) mine[t * 64 + lane] += dWc[cwoff(l) + nt];
                }
                if constexpr (C::NCF(l)) {
                    // the cooperative rounds keep a whole 16x16x4 tile for the last N tile (D layout: lane (i16, g) register r = unit
                    // 16 wave + 4g + r, slot 16 (NT - 1) + i16); its columns 4j go to lane 4 (4 wave + g) + j of the N-fringe tile
                    const int t = C::dwfr3(l) - t0, i16 = lane & 15, g = lane >> 4;
                    if (t >= 0 && t < cnt && (i16 & 3) == 0) mine[t * 64 + 16 * wave + 4 * g + (i16 >> 2)] += dWc[cwoff(l) + C::NTF(l)];
                }
            }
        }
        if constexpr (l + 1 < C::NLM3) merge<l + 1>(mine, dWc, wave, lane, t0, cnt);
    }
};

// FringeOut's sums as (value, slab position) pairs in registers, nothing stored: every load of every layer is issued before the
// first store (Epi3).  Threads without an item load item 0 (clamped) and get position -1.
template <class S, int l>
struct FringeVals {
    using C = F3Cfg<S>;
    static constexpr int NF = C::NF(l), KIN = C::KIN(l), in = C::in(l), out = C::out(l);
    static constexpr int NI = C::FB(l) ? NF * C::NT(l) * 16 : NF * (KIN + 1) * 4;                     // items
    static constexpr int TI = C::FB(l) ? NI : NI * FAST_WAVES;                                        // threads per pass over them
    static constexpr int R = NF > 0 ? (TI + FAST_THREADS - 1) / FAST_THREADS : 0;
    static constexpr int total() { if constexpr (l + 1 < C::NL) return R + FringeVals<S, l + 1>::total(); else return R > 0 ? R : 1; }
    template <int AT, int NR>
    static __device__ __forceinline__ void run(const float* lb, int tid, float (&fv)[NR], int (&fd)[NR]) {
        if constexpr (NF > 0 && C::FB(l)) {
            // B-operand layout: item = (f, nt, i16): sum over the 4 waves x 4 lane groups (x 2 row pairs when NF == 1)
            constexpr int NT = C::NT(l);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int it0 = tid + r * FAST_THREADS, it = it0 < NI ? it0 : 0;
                const int i16 = it & 15, nt = (it >> 4) % NT, f = it / (16 * NT);
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < FAST_WAVES; ++w) {
                    float vw = 0.f;
#pragma unroll
                    for (int gg = 0; gg < 4; ++gg) {
                        if constexpr (NF == 2) vw += lb[(size_t)(w * C::FP_REGS + C::fpoff(l) + 2 * nt + f) * 64 + 16 * gg + i16];
                        else vw += lb[(size_t)(w * C::FP_REGS + C::fpoff(l) + 2 * nt) * 64 + 16 * gg + i16] +
                                   lb[(size_t)(w * C::FP_REGS + C::fpoff(l) + 2 * nt + 1) * 64 + 16 * gg + i16];
                    }
                    v += vw;
                }
                const int cs = 16 * nt + i16, u = C::funit(l, f);
                const int col = l == 0 ? (cs <= in ? cs : -1) : unit_of(in, cs, true);
                fv[AT + r] = v;
                fd[AT + r] = (it0 < NI && col >= 0) ? C::offW(l) + (col < in ? u * in + col : in * out + u) : -1;
            }
        } else if constexpr (NF > 0) {
            // D layout: thread (item = (f, kslot, g), wave)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int t = tid + r * FAST_THREADS;
                const int it0 = t >> 2, w = t & 3, it = it0 < NI ? it0 : 0;
                const int gg = it & 3, fk = it >> 2;
                const int f = fk / (KIN + 1), k = fk - f * (KIN + 1);
                const int reg = C::fpoff(l) + f * (KIN + 1) + k;
                const f32x4* src = reinterpret_cast<const f32x4*>(lb + ((size_t)(w * C::FP_REGS + reg) * 4 + gg) * 16);
                const f32x4 p0 = src[0], p1 = src[1], p2 = src[2], p3 = src[3];
                float v = ((p0[0] + p0[1]) + (p0[2] + p0[3])) + ((p1[0] + p1[1]) + (p1[2] + p1[3])) +
                          (((p2[0] + p2[1]) + (p2[2] + p2[3])) + ((p3[0] + p3[1]) + (p3[2] + p3[3])));
                int dest = -1;
                const int u = C::funit(l, f);
                if (k == KIN) { if (gg == 0) dest = C::offW(l) + in * out + u; }
                else {
                    const int col = l == 0 ? (4 * k + gg < in ? 4 * k + gg : -1) : unit_of(in, 16 * (k >> 2) + 4 * gg + (k & 3), false);
                    if (col >= 0) dest = C::offW(l) + u * in + col;
                }
                v += __shfl_xor(v, 1, 64);
                v += __shfl_xor(v, 2, 64);
                fv[AT + r] = v;
                fd[AT + r] = (it0 < NI && w == 0) ? dest : -1;
            }
        }
        if constexpr (l + 1 < C::NL) FringeVals<S, l + 1>::template run<AT + R>(lb, tid, fv, fd);
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// Epilogue of the one-pass shapes, written for ONE wave per SIMD (every instruction is paid at issue, nothing hides a latency):
//  * specialised on the wave (a switch over four instantiations): which tiles a wave sums, where its copies go and which of its
//    cooperative accumulators belong to which tile are then compile-time facts -- straight-line code, no per-tile predicate;
//  * a wave keeps its own copy of the tiles it sums (t = wave mod 4) in registers: three staged copies per tile instead of four
//    (the N-fringe accumulator pairs keep four: their cooperative share lands through a cross-lane LDS update);
//  * the summed tiles and fringe rows go to a dense copy of the slab in LDS (it fits behind the three-copy staging area) and
//    leave as 16-byte write-through stores, one contiguous KB per wave instruction, instead of one 4-byte fabric write per entry;
//  * the log-likelihood statistic is reduced after the stores are issued.
// The sum over the four waves keeps its fixed order (c0 + c1) + (c2 + c3).
// ---------------------------------------------------------------------------------------------------------------------
#ifndef TBNN_F3_EPI
#define TBNN_F3_EPI 1
#endif
#ifndef TBNN_F3_COOP_PEEL
#define TBNN_F3_COOP_PEEL 1          // two copies of the cooperative tile instead of a loop (0: the loop; configs[1]: +0.3 us of register shuffles around it)
#endif
template <class S>
struct Epi3 {
    using C = F3Cfg<S>;
    using CO = Coop3<S>;
    static constexpr int T = C::DW3_TILES > 0 ? C::DW3_TILES : 1;
    static constexpr int KMAX = (T + FAST_WAVES - 1) / FAST_WAVES;
    static constexpr int P4 = (C::P() + 3) / 4 * 4;
    static constexpr int FPd = C::FP_REGS > 0 ? C::FP_REGS : 1;
    // tiles of an N-fringe accumulator pair: index among them, or -1
    static constexpr int nf_index(int t) {
        int k = 0;
        for (int l = 0; l < C::NLM3; ++l)
            if (C::NCF(l)) {
                if (t == C::dwfr3(l)) return k;
                if (t == C::dwfr3(l) + 1) return k + 1;
                k += 2;
            }
        return -1;
    }
    static constexpr int nf_count() { int k = 0; for (int l = 0; l < C::NLM3; ++l) if (C::NCF(l)) k += 2; return k; }
    static constexpr int NFT = nf_count();
    // staging slot (KB) of source wave c's copy of tile t, or -1: the summing wave (t mod 4) keeps it in registers
    static constexpr int slot(int t, int c) {
        const int own = t & (FAST_WAVES - 1);
        if (c == own) return nf_index(t) >= 0 ? 3 * T + nf_index(t) : -1;
        return ((c - own + 3) & 3) * T + t;
    }
    static constexpr int ST_FLOATS = (3 * T + NFT) * 256;                         // staged tile copies
    static constexpr int LB_FLOATS = FAST_WAVES * C::FP_REGS * 64;                // fringe partials
    static constexpr int DENSE_OFF = ST_FLOATS + LB_FLOATS;
    static constexpr bool ENABLED = TBNN_F3_EPI && C::DW3_TILES > 0 && C::EP3_TILES == C::DW3_TILES && C::P() >= 2048 &&
                                    (size_t)DENSE_OFF + P4 <= (size_t)C::LDS3_FLOATS;

    // wave W's cooperative accumulator for its copy of tile t (full tiles of its own M tile), or -1
    template <int W>
    static constexpr int coop_index(int t) {
        if (!CO::ENABLED) return -1;
        for (int l = 0; l < C::NLM3; ++l)
            if (C::MTF(l) > 0 && W < C::MTF(l))
                for (int nt = 0; nt < C::NTF(l); ++nt)
                    if (t == C::dwoff3(l) + W * C::NTF(l) + nt) return CO::cwoff(l) + nt;
        return -1;
    }

    // stage: every copy another wave sums -> LDS; own[k] = this wave's copy of tile 4 k + W
    template <int W, int t = 0>
    static __device__ __forceinline__ void stage(f32x4* st, f32x4 (&own)[KMAX], const f32x4 (&dW)[T], const f32x4 (&dWc)[CO::DWC], int lane) {
        if constexpr (t < C::DW3_TILES) {
            f32x4 v = dW[t];
            constexpr int ci = coop_index<W>(t);
            if constexpr (ci >= 0) v += dWc[ci];                                  // (zero when the workgroup ran no cooperative tile)
            constexpr int sl = slot(t, W);
            if constexpr (sl >= 0) st[sl * 64 + lane] = v;
            else own[t / FAST_WAVES] = v;
            stage<W, t + 1>(st, own, dW, dWc, lane);
        }
    }
    // the cooperative rounds keep a whole 16x16x4 tile for the last N tile (D layout: lane (i16, g) register r = unit 16 W + 4 g + r,
    // slot 16 (NT - 1) + i16); its columns 4 j go to lane 4 (4 W + g) + j of the wave's copy of the N-fringe tile (Coop3::merge)
    template <int W, int l = 0>
    static __device__ __forceinline__ void stage_nf_coop(f32x4* st, const f32x4 (&dWc)[CO::DWC], int lane) {
        if constexpr (CO::ENABLED && l < C::NLM3) {
            if constexpr (C::MTF(l) > 0 && W < C::MTF(l) && C::NCF(l)) {
                const int i16 = lane & 15, g = lane >> 4;
                if ((i16 & 3) == 0) st[slot(C::dwfr3(l), W) * 64 + 16 * W + 4 * g + (i16 >> 2)] += dWc[CO::cwoff(l) + C::NTF(l)];
            }
            stage_nf_coop<W, l + 1>(st, dWc, lane);
        }
    }
    // source wave c's copy of tile t as wave W = t mod 4 sees it: staged, or its own registers
    template <int W, int t, int c>
    static __device__ __forceinline__ f32x4 cpy(const f32x4* st, const f32x4 (&own)[KMAX], int lane) {
        constexpr int sl = slot(t, c);
        if constexpr (sl >= 0) return st[sl * 64 + lane];
        else return own[t / FAST_WAVES];
    }
    // The sums first, every store after them: the dense copy and the staging area are one LDS object, the compiler keeps a load
    // behind any earlier store -- a tile summed and stored at a time pays a full LDS round trip per tile.
    static constexpr int NLd = C::NLM3 > 0 ? C::NLM3 : 1;
    template <int W, int k = 0>
    static __device__ __forceinline__ void tiles_sum(const f32x4* st, const f32x4 (&own)[KMAX], int lane, f32x4 (&sum)[KMAX]) {
        if constexpr (k < KMAX) {
            constexpr int t = FAST_WAVES * k + W;
            if constexpr (t < C::DW3_TILES && nf_index(t < C::DW3_TILES ? t : 0) < 0) {
                const f32x4 c0 = cpy<W, t, 0>(st, own, lane), c1 = cpy<W, t, 1>(st, own, lane), c2 = cpy<W, t, 2>(st, own, lane), c3 = cpy<W, t, 3>(st, own, lane);
                sum[k] = (c0 + c1) + (c2 + c3);
            }
            tiles_sum<W, k + 1>(st, own, lane, sum);
        }
    }
    // the N-fringe accumulator pairs (even-row and odd-row accumulators: tiles t and t + 1)
    template <int W, int l = 0>
    static __device__ __forceinline__ void nf_sum(const f32x4* st, const f32x4 (&own)[KMAX], int lane, f32x4 (&nfs)[NLd]) {
        if constexpr (l < C::NLM3) {
            if constexpr (C::NCF(l) && (C::dwfr3(l) & (FAST_WAVES - 1)) == W) {
                constexpr int t = C::dwfr3(l);
                const f32x4 c0 = cpy<W, t, 0>(st, own, lane) + cpy<W, t + 1, 0>(st, own, lane), c1 = cpy<W, t, 1>(st, own, lane) + cpy<W, t + 1, 1>(st, own, lane),
                            c2 = cpy<W, t, 2>(st, own, lane) + cpy<W, t + 1, 2>(st, own, lane), c3 = cpy<W, t, 3>(st, own, lane) + cpy<W, t + 1, 3>(st, own, lane);
                nfs[l] = (c0 + c1) + (c2 + c3);
            }
            nf_sum<W, l + 1>(st, own, lane, nfs);
        }
    }
    // layer l's tiles of wave W -> the dense slab copy in LDS (destinations as SlabOut3)
    template <int W, int l = 0>
    static __device__ __forceinline__ void tiles_out(float* dense, const f32x4 (&sum)[KMAX], const f32x4 (&nfs)[NLd], int lane) {
        if constexpr (l < C::NLM3) {
            constexpr int in = C::in(l), out = C::out(l), MT = C::MTF(l);
            if constexpr (C::NCF(l) && (C::dwfr3(l) & (FAST_WAVES - 1)) == W) {
                // the N-fringe tile: lane (b, j) register i = dW[unit 4b + i][slot 16 (NT_all - 1) + 4j]
                const int b = lane >> 2, col = unit_of(in, 16 * (C::NT(l) - 1) + 4 * (lane & 3), true);
                if (b < 4 * MT && col >= 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = unit_of(out, 4 * b + r, false);           // (a partly filled last M tile keeps its units in slot order)
                        if (row >= 0) dense[C::offW(l) + (col < in ? row * in + col : in * out + row)] = nfs[l][r];
                    }
                }
            }
            tiles_out_full<W, l, 0>(dense, sum, lane);
            tiles_out<W, l + 1>(dense, sum, nfs, lane);
        }
    }
    template <int W, int l, int k>
    static __device__ __forceinline__ void tiles_out_full(float* dense, const f32x4 (&sum)[KMAX], int lane) {
        constexpr int in = C::in(l), out = C::out(l), MT = C::MTF(l), NT = C::NTF(l);
        if constexpr (k < MT * NT) {
            constexpr int t = C::dwoff3(l) + k, mt = k / (NT > 0 ? NT : 1), nt = k % (NT > 0 ? NT : 1);
            if constexpr ((t & (FAST_WAVES - 1)) == W) {
                const int cs = 16 * nt + (lane & 15), row0 = 16 * mt + 4 * (lane >> 4);
                const int col = l == 0 ? (cs <= in ? cs : -1) : unit_of(in, cs, true);
                if (col >= 0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = unit_of(out, row0 + r, false);
                        if (row >= 0) dense[C::offW(l) + (col < in ? row * in + col : in * out + row)] = sum[t / FAST_WAVES][r];
                    }
                }
            }
            tiles_out_full<W, l, k + 1>(dense, sum, lane);
        }
    }

    template <int W>
    static __device__ __forceinline__ void run(float* __restrict__ lds, float* __restrict__ slab, const f32x4 (&dW)[T], const f32x4 (&dWc)[CO::DWC],
                                               const float (&FP)[FPd], int tid, int lane) {
        const int i16 = lane & 15, g = lane >> 4;
        f32x4* st = reinterpret_cast<f32x4*>(lds);
        float* lb = lds + ST_FLOATS;
        float* dense = lds + DENSE_OFF;
        f32x4 own[KMAX];                                                          // (entries of staged tiles stay unset and unread)
        __syncthreads();                                                          // images and exchange buffers are dead
        stage<W>(st, own, dW, dWc, lane);
        stage_nf_coop<W>(st, dWc, lane);
        {   // (one base register + immediate offsets: the area lies above the 64 KB an LDS offset field reaches from 0)
            int o = (W * C::FP_REGS * 4 + g) * 16 + i16;
            asm volatile("" : "+v"(o));                                          // (an opaque index: the pointer itself would leave the LDS address space)
#pragma unroll
            for (int r = 0; r < C::FP_REGS; ++r) lb[o + r * 64] = FP[r];
        }
        if (tid < P4 - C::P()) dense[C::P() + tid] = 0.f;
        __syncthreads();
        TSTAMPO(63);
        f32x4 sum[KMAX], nfs[NLd];
        tiles_sum<W>(st, own, lane, sum);
        nf_sum<W>(st, own, lane, nfs);
        float fv[FringeVals<S, 0>::total()];
        int fd[FringeVals<S, 0>::total()];
        FringeVals<S, 0>::template run<0>(lb, tid, fv, fd);
        TSTAMPO(35);
        tiles_out<W>(dense, sum, nfs, lane);
#pragma unroll
        for (int i = 0; i < FringeVals<S, 0>::total(); ++i) if (fd[i] >= 0) dense[fd[i]] = fv[i];
        TSTAMPO(36);
        __syncthreads();
        TSTAMPO(37);
        {   // every read of the dense copy, then every store (a loop of read -> store pays an LDS round trip per 16 bytes)
            constexpr int N4 = P4 / 4, NE = (N4 + FAST_THREADS - 1) / FAST_THREADS;
            f32x4 d[NE];
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                const int e = tid + k * FAST_THREADS;
                d[k] = *reinterpret_cast<const f32x4*>(dense + 4 * (e < N4 ? e : 0));
            }
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                const int e = tid + k * FAST_THREADS;
                if (e < N4) store16<true>(slab + 4 * e, d[k]);
            }
        }
    }
};

// 16 bytes per lane from global memory straight into LDS (global_load_lds_dwordx4: lane i lands at lbase + 16 i; lbase is
// wave-uniform).  The builtin exists in the device pass only; the host pass needs just the kernel's stub.
__device__ __forceinline__ void lds_dma16(const float* gsrc, float* lbase) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds(gsrc, lbase, 16, 0, 0);
#else
    (void)gsrc; (void)lbase;
#endif
}

template <class S>
__global__ __launch_bounds__(FAST_THREADS, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_fwd_bwd_fast3(
    NetDev nd, const float* __restrict__ qimg, const float* __restrict__ eta,
    const float* __restrict__ X, const float* __restrict__ Y, long n,
    float* __restrict__ slabs, int pitch, double* __restrict__ pstat, unsigned long long* __restrict__ stamps, ChainStride cs)
{
    using C = F3Cfg<S>;
    // gridDim.y = chains of a multi-chain handle (tbnn_create_multi): this workgroup's chain
    if (chain_done(cs.ctl, cs.t, blockIdx.y)) return;           // a chain past its own L (per-chain step control)
    qimg += (size_t)blockIdx.y * cs.img; eta += (size_t)blockIdx.y * cs.eta; slabs += (size_t)blockIdx.y * cs.slab; pstat += (size_t)blockIdx.y * PSTAT_CAP;
    static_assert(C::VL && C::NF(C::NL - 1) == C::out(C::NL - 1) && C::MTF(C::NL - 1) == 0, "fast3: last layer must be all-fringe");
    using CO = Coop3<S>;
    static_assert((C::LDS3_FLOATS + (CO::ENABLED ? 2 * CO::XB : 4)) * 4 + 64 <= 160 * 1024, "LDS budget");
    static_assert((size_t)FAST_WAVES * C::FP_REGS * 64 <= (size_t)C::LDS3_FLOATS, "fringe staging does not fit");
#define TB_STAMP(i) do { if (stamps && threadIdx.x == 0 && blockIdx.x == 0) { stamps[i] = wall_clock64(); stamps[8 + i] = clock64(); } } while (0)
    TB_STAMP(0); TSTAMPO(57);
    __shared__ __attribute__((aligned(16))) float lds[C::LDS3_FLOATS];
    __shared__ __attribute__((aligned(16))) float xch[CO::ENABLED ? 2 * CO::XB : 4];     // cooperative tail: exchange buffers
    __shared__ double red[FAST_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: wave-uniform branches stay s_cbranch
    const int i16 = lane & 15, g = lane >> 4;
    float* wl = lds + C::STATIC_FLOATS + wave * C::WAVE3_FLOATS;
    constexpr int d_in = C::in(0), d_out = C::out(C::NL - 1), L = C::NL - 1;
    const long ntiles = (n + 15) / 16;
    const long W = (long)gridDim.x * FAST_WAVES;
    const long wg = (long)blockIdx.x * FAST_WAVES + wave;
    // the rows of this wave's first tile (and of the cooperative tiles) are requested before anything else: their HBM
    // latency (the first tile took 7.5 us instead of 6.1) hides under the prologue's image loads
    float xn[C::KS0], yn[d_out];
    auto fetch = [&](long tile) {
        const long row = tile * 16 + i16;
        const bool ok = tile < ntiles && row < n;
#pragma unroll
        for (int t = 0; t < C::KS0; ++t) {
            const int u = 4 * t + g;
            xn[t] = (ok && u < d_in) ? X[row * d_in + u] : 0.f;
        }
#pragma unroll
        for (int o = 0; o < d_out; ++o) yn[o] = ok ? Y[row * d_out + o] : 0.f;
    };
    // full rounds of W tiles run on the tile loop; a remainder of at most 2 tiles per workgroup runs as cooperative rounds
    // (Coop3), tile = main_end + round * gridDim.x + blockIdx.x
    long main_end = ntiles;
    int ncoop = 0;
    if constexpr (CO::ENABLED) {
        const long G = gridDim.x, rem = ntiles % W;
        if (rem > 0 && rem <= 2 * G) { main_end = ntiles - rem; ncoop = rem <= G ? 1 : 2; }
    }
    if constexpr ((TBNN_SKEL & 1) != 0) { main_end = 0; ncoop = 0; }       // diagnostic: the launch's fixed cost
    // the cooperative tiles' rows are fetched now: their latency hides under the whole tile loop
    float xc[2][C::KS0], yc[2][d_out];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const long ct = main_end + (long)j * gridDim.x + blockIdx.x;
        const long row = ct * 16 + i16;
        const bool ok = CO::ENABLED && j < ncoop && ct < ntiles && row < n;
#pragma unroll
        for (int t = 0; t < C::KS0; ++t) {
            const int u = 4 * t + g;
            xc[j][t] = (ok && u < d_in) ? X[row * d_in + u] : 0.f;
        }
#pragma unroll
        for (int o = 0; o < d_out; ++o) yc[j][o] = ok ? Y[row * d_out + o] : 0.f;
    }
    long tile = wg;
    fetch(tile);
    // The W^T images (delta chain: needed from the first BACKWARD pass on) do not go through registers: each wave sends its
    // share straight to LDS with global_load_lds (1 KB per instruction, no VGPRs held), and the wait + barrier that makes
    // them visible sits behind the first tile's forward pass -- the prologue proper stages the W / bias images only.
#ifndef TBNN_F3_WTDMA
#define TBNN_F3_WTDMA 1
#endif
    constexpr bool WTDMA = TBNN_F3_WTDMA && C::STATIC_FLOATS > C::WB_FLOATS;
    if constexpr (WTDMA) {
        constexpr int WT = C::STATIC_FLOATS - C::WB_FLOATS, NCHUNK = (WT + 255) / 256;
        static_assert(C::WB_FLOATS % 4 == 0 && WT % 4 == 0, "16-B pieces");
        for (int c = wave; c < NCHUNK; c += FAST_WAVES) {
            const int off = C::WB_FLOATS + c * 256;
            if (c * 256 + lane * 4 < WT) lds_dma16(qimg + off + lane * 4, lds + off);
        }
    }
    {
        // all image loads in flight first, the zero fill of the per-wave images under their latency, then the LDS stores
        constexpr int N4 = (WTDMA ? C::WB_FLOATS : C::STATIC_FLOATS) / 4, IT = (N4 + FAST_THREADS - 1) / FAST_THREADS;
        const float4* src = reinterpret_cast<const float4*>(qimg);
        float4* dst = reinterpret_cast<float4*>(lds);
        float4 v[IT];
#pragma unroll
        for (int k = 0; k < IT; ++k) { const int e = tid + k * FAST_THREADS; v[k] = e < N4 ? src[e] : make_float4(0.f, 0.f, 0.f, 0.f); }
        __builtin_amdgcn_sched_barrier(0);
        float4* z = reinterpret_cast<float4*>(wl);
        for (int e = lane; e < C::WAVE3_FLOATS / 4; e += 64) z[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < IT; ++k) { const int e = tid + k * FAST_THREADS; if (e < N4) dst[e] = v[k]; }
    }
    __syncthreads();
    TB_STAMP(1); TSTAMPO(58);

    constexpr int NFd = C::maxNF() > 0 ? C::maxNF() : 1;
    f32x4 dW[C::DW3_TILES > 0 ? C::DW3_TILES : 1];
#pragma unroll
    for (int t = 0; t < C::DW3_TILES; ++t) dW[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float FP[C::FP_REGS > 0 ? C::FP_REGS : 1];
#pragma unroll
    for (int t = 0; t < C::FP_REGS; ++t) FP[t] = 0.f;
    const float sigma = lik_sigma(nd, eta);
    const float inv_var = 1.f / (sigma * sigma);
    double stat = 0.0;

    if (g == 0) wl[C::aoff3(0) + d_in * C::PR + i16] = 1.f;
#pragma unroll
    for (int l = 1; l < C::NLM3; ++l)
        if (C::in(l) % 16 == 0 && g == 0) wl[C::aoff3(l) + C::in(l) * C::PR + i16] = 1.f;
#if TBNN_F3_HAND
    typename Fwd3<S, 0>::Pre P0;                 // layer 0's operands (A, bias, fringe weights): the same for every tile
    Fwd3<S, 0>::pre_all(P0, lds, i16, g);
#else
    f32x4 A0[C::MTF(0) > 0 ? C::MTF(0) : 1], B0[C::MTF(0) > 0 ? C::MTF(0) : 1];
    Fwd3<S, 0>::preload(A0, B0, lds, i16, g);
#endif

    bool first = true;
    // the W^T images in flight: every wave of the workgroup must meet ONE barrier behind its own wait.  When all four waves
    // have a tile in the loop that barrier sits behind the first forward pass; otherwise here.
    bool wt_pending = WTDMA;
    if (WTDMA && !((long)blockIdx.x * FAST_WAVES + FAST_WAVES - 1 < main_end)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        wt_pending = false;
    }
    for (; tile < main_end; tile += W) {
        Tile3<S> T;
        float y[d_out];
        const bool rvalid = tile * 16 + i16 < n;
#pragma unroll
        for (int t = 0; t < C::KS0; ++t) {
            T.x0[t] = xn[t];
            const int u = 4 * t + g;
            if (u < d_in) wl[C::aoff3(0) + u * C::PR + i16] = xn[t];
        }
#pragma unroll
        for (int o = 0; o < d_out; ++o) y[o] = yn[o];
        fetch(tile + W);
        TSTAMP(0);
#if TBNN_F3_HAND
        Fwd3<S, 0>::run_h(T, lds, wl, i16, g, P0);
#else
        Fwd3<S, 0>::run(T, lds, wl, i16, g, A0, B0);
#endif
        if (wt_pending) {                       // first tile only, the same on all four waves
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            wt_pending = false;
        }
        // likelihood on the all-fringe last layer
        float dzf[NFd];
#pragma unroll
        for (int o = 0; o < d_out; ++o) dzf[o] = rvalid ? lik_delta<S>(T.af[L][o], y[o], inv_var, g == 0, stat) : 0.f;
        TSTAMP(30);
        // (the last layer's fringe sums -- a dozen packed FMAs -- wait until the N-fringe operands of layer L-1 are on their way: NCF)
        if constexpr (!C::NCF(L > 0 ? L - 1 : 0)) FringeDW<S, L>::run(FP, T, dzf, g);
        // delta of layer L-1, then the pipeline
        f32x4 dzL[C::MT(L)];
        f32x4 dzp[C::MT(L - 1)];
        float dzpf[NFd];
        Bwd3<S, L>::da(T, lds, i16, g, dzL, dzf, dzp, dzpf);
        TSTAMP(31);
        {
            constexpr int LM = L - 1;
            float Aop[Bwd3<S, LM>::MTd][4], Bop[C::NT(LM)][4], Fop[8];
            typename Bwd3<S, LM>::NFOps NFop;
            Bwd3<S, LM>::issue(dzp, dzpf, wl, i16, g, Aop, Bop, Fop);
            Bwd3<S, LM>::nf_load(NFop, wl, lane);
            if constexpr (C::NCF(LM)) FringeDW<S, L>::run(FP, T, dzf, g);
            TSTAMP(32);
            FringeDW<S, LM>::run(FP, T, dzpf, g);
#ifndef TBNN_F3_NF_LATE
#define TBNN_F3_NF_LATE 0
#endif
            if constexpr (!(TBNN_F3_NF_LATE && LM > 0)) Bwd3<S, LM>::nf_mfma(dW, NFop);
            TSTAMP(33);
            if constexpr (LM > 0) {
                f32x4 dzq[C::MT(LM - 1)];
                float dzqf[NFd];
                Bwd3<S, LM>::da(T, lds, i16, g, dzp, dzpf, dzq, dzqf);
                if constexpr (TBNN_F3_NF_LATE != 0) Bwd3<S, LM>::nf_mfma(dW, NFop);
                SCHED_FENCE();
                TSTAMP(34);
                Pipe3<S, LM - 1>::run(dW, FP, T, lds, wl, i16, g, dzq, dzqf, Aop, Bop, Fop);
            } else {
                Bwd3<S, 0>::dw(dW, Aop, Bop);
                Bwd3<S, 0>::fdw(FP, Fop, Bop);
            }
        }
        if (first) { TB_STAMP(2); TSTAMPO(59); first = false; }
    }
    TB_STAMP(3); TSTAMPO(60);
    mfma_drain_acc(dW);
    f32x4 dWc[CO::DWC];
#pragma unroll
    for (int t = 0; t < CO::DWC; ++t) dWc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (CO::ENABLED) {
        if constexpr (TBNN_F3_COOP_PEEL && C::P() >= 1024) {
            // straight-line copies of the body, no loop: no loop-carried register shuffle around a tile (conditions are workgroup-uniform:
            // the barriers inside are met by all 4 waves).  Small networks keep the loop: their launch is fetch- and launch-bound, the second
            // copy costs configs[0] 1 %
            const long ct0 = main_end + blockIdx.x, ct1 = ct0 + gridDim.x;
            if (ncoop > 0 && ct0 < ntiles) CO::tile(dWc, FP, stat, lds, wl, xch, wave, lane, i16, g, inv_var, xc[0], yc[0], ct0 * 16 + i16 < n);
            if (ncoop > 1 && ct1 < ntiles) {
                CO::tile(dWc, FP, stat, lds, wl, xch, wave, lane, i16, g, inv_var, xc[1], yc[1], ct1 * 16 + i16 < n);
            }
        } else {
#pragma unroll 1
            for (int j = 0; j < ncoop; ++j) {                       // one copy of the body: the second round's rows by select
                const long ct = main_end + (long)j * gridDim.x + blockIdx.x;
                if (ct >= ntiles) break;                            // workgroup-uniform: the barriers inside are met by all 4 waves
                float xj[C::KS0], yj[d_out];
#pragma unroll
                for (int t = 0; t < C::KS0; ++t) xj[t] = j == 0 ? xc[0][t] : xc[1][t];
#pragma unroll
                for (int o = 0; o < d_out; ++o) yj[o] = j == 0 ? yc[0][o] : yc[1][o];
                CO::tile(dWc, FP, stat, lds, wl, xch, wave, lane, i16, g, inv_var, xj, yj, ct * 16 + i16 < n);
            }
        }
    }
    TB_STAMP(5); TSTAMPO(62);

    // ---- epilogue
    if constexpr (Epi3<S>::ENABLED) {
        float* slab = slabs + (size_t)blockIdx.x * pitch;
        switch (wave) {
            case 0: Epi3<S>::template run<0>(lds, slab, dW, dWc, FP, tid, lane); break;
            case 1: Epi3<S>::template run<1>(lds, slab, dW, dWc, FP, tid, lane); break;
            case 2: Epi3<S>::template run<2>(lds, slab, dW, dWc, FP, tid, lane); break;
            default: Epi3<S>::template run<3>(lds, slab, dW, dWc, FP, tid, lane); break;
        }
        const double wtot = wave_sum_lane0(stat);                // (lane 0: the bits of wave_sum)
        if (lane == 0) red[wave] = wtot;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int w = 0; w < FAST_WAVES; ++w) t += red[w];
            pstat[blockIdx.x] = t;
        }
    } else {
        const double wtot = wave_sum_lane0(stat);
        if (lane == 0) red[wave] = wtot;
        float* slab = slabs + (size_t)blockIdx.x * pitch;
        // one staging pass for the dW tiles AND the fringe partials when both fit (configs[1]: 108 + 35 KB): two barriers
        // instead of four
        constexpr bool ONE_PASS = C::DW3_TILES > 0 && C::EP3_TILES == C::DW3_TILES &&
                                  (size_t)FAST_WAVES * (C::DW3_TILES * 256 + C::FP_REGS * 64) <= (size_t)C::LDS3_FLOATS;
        if constexpr (ONE_PASS) {
            float* lb = lds + FAST_WAVES * C::DW3_TILES * 256;
            TSTAMP(20);
            __syncthreads();
            TSTAMP(21);
            f32x4* mine = reinterpret_cast<f32x4*>(lds) + wave * (C::DW3_TILES * 64);
#pragma unroll
            for (int t = 0; t < C::DW3_TILES; ++t) mine[t * 64 + lane] = dW[t];
            TSTAMP(22);
            if constexpr (CO::ENABLED) { if (ncoop > 0) CO::template merge<0>(mine, dWc, wave, lane, 0, C::DW3_TILES); }
            TSTAMP(23);
#pragma unroll
            for (int r = 0; r < C::FP_REGS; ++r) lb[((size_t)(wave * C::FP_REGS + r) * 4 + g) * 16 + i16] = FP[r];
            TSTAMP(24);
            __syncthreads();
            TB_STAMP(6); TSTAMPO(63);
            SlabOut3<S, 0>::run(lds, slab, wave, lane, 0, C::DW3_TILES);
            TB_STAMP(7);
            FringeOut<S, 0>::run(lb, slab, tid);
        } else {
            if constexpr (C::DW3_TILES > 0) {
#pragma unroll
                for (int t0 = 0; t0 < C::DW3_TILES; t0 += C::EP3_TILES) {
                    __syncthreads();
                    f32x4* mine = reinterpret_cast<f32x4*>(lds) + wave * (C::EP3_TILES * 64);
#pragma unroll
                    for (int t = t0; t < t0 + C::EP3_TILES && t < C::DW3_TILES; ++t) mine[(t - t0) * 64 + lane] = dW[t];
                    const int cnt = (C::DW3_TILES - t0) < C::EP3_TILES ? (C::DW3_TILES - t0) : C::EP3_TILES;
                    if constexpr (CO::ENABLED) { if (ncoop > 0) CO::template merge<0>(mine, dWc, wave, lane, t0, cnt); }
                    __syncthreads();
                    SlabOut3<S, 0>::run(lds, slab, wave, lane, t0, cnt);
                }
            }
            {   // fringe partials: [wave][reg][g][i16]
                __syncthreads();
                float* lb = lds;
#pragma unroll
                for (int r = 0; r < C::FP_REGS; ++r) lb[((size_t)(wave * C::FP_REGS + r) * 4 + g) * 16 + i16] = FP[r];
                __syncthreads();
                FringeOut<S, 0>::run(lb, slab, tid);
            }
        }
        if (tid == 0) {
            double t = 0.0;
            for (int w = 0; w < FAST_WAVES; ++w) t += red[w];
            pstat[blockIdx.x] = t;
        }
    }
    TB_STAMP(4); TSTAMPO(61);
#undef TB_STAMP
}


// Forward pass only (predictions): the same register-chained MFMA layers, no images, no likelihood, no backward.
// blockIdx.y selects the network of an ensemble: weight image qimgs + y * img_stride, output fout + y * out_stride,
// fout[d_out][n] per network (predictor.py:132-155 evaluates every saved network on the same rows).
template <class S>
__global__ __launch_bounds__(FAST_THREADS, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_forward_fast3(
    const float* __restrict__ qimgs, long img_stride, const float* __restrict__ X, long n, float* __restrict__ fouts, long out_stride)
{
    using C = F3Cfg<S>;
    static_assert(C::VL && C::NF(C::NL - 1) == C::out(C::NL - 1) && C::MTF(C::NL - 1) == 0, "fast3: last layer must be all-fringe");
    __shared__ __attribute__((aligned(16))) float lds[C::WB_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, g = lane >> 4;
    const float* qimg = qimgs + (size_t)blockIdx.y * img_stride;
    float* fout = fouts + (size_t)blockIdx.y * out_stride;
    {
        constexpr int N4 = C::WB_FLOATS / 4;
        static_assert(C::WB_FLOATS % 4 == 0, "image sections are 16-B multiples");
        const float4* src = reinterpret_cast<const float4*>(qimg);
        float4* dst = reinterpret_cast<float4*>(lds);
        for (int e = tid; e < N4; e += FAST_THREADS) dst[e] = src[e];
    }
    __syncthreads();
    constexpr int d_in = C::in(0), d_out = C::out(C::NL - 1), L = C::NL - 1;
    const long ntiles = (n + 15) / 16;
    const long W = (long)gridDim.x * FAST_WAVES;
#if TBNN_F3_HAND
    typename Fwd3<S, 0, false>::Pre P0;
    Fwd3<S, 0, false>::pre_all(P0, lds, i16, g);
#else
    f32x4 A0[C::MTF(0) > 0 ? C::MTF(0) : 1], B0[C::MTF(0) > 0 ? C::MTF(0) : 1];
    Fwd3<S, 0, false>::preload(A0, B0, lds, i16, g);
#endif
    for (long tile = (long)blockIdx.x * FAST_WAVES + wave; tile < ntiles; tile += W) {
        Tile3<S> T;
        const long row = tile * 16 + i16;
        const bool ok = row < n;
#pragma unroll
        for (int t = 0; t < C::KS0; ++t) {
            const int u = 4 * t + g;
            T.x0[t] = (ok && u < d_in) ? X[row * d_in + u] : 0.f;
        }
#if TBNN_F3_HAND
        Fwd3<S, 0, false>::run_h(T, lds, nullptr, i16, g, P0);
#else
        Fwd3<S, 0, false>::run(T, lds, nullptr, i16, g, A0, B0);
#endif
        if (ok && g == 0) {
#pragma unroll
            for (int o = 0; o < d_out; ++o) fout[(size_t)o * n + row] = T.af[L][o];
        }
    }
}

#ifndef TBNN_NO_FAST_REGISTRY
static inline bool fast3_available(int id) { return id == 0 || id == 1 || id == 2; }
static inline int fast3_launch(int id, int grid, hipStream_t st, const NetDev& nd, const float* qimg, const float* eta,
                               const float* X, const float* Y, long n, float* slabs, int pitch, double* pstat,
                               unsigned long long* stamps = nullptr, int nchains = 1, ChainStride cs = ChainStride{0, 0, 0, nullptr, 0}) {
    switch (id) {
        case 0: hipLaunchKernelGGL(k_fwd_bwd_fast3<ShapeC2>, dim3(grid, nchains), dim3(FAST_THREADS), 0, st, nd, qimg, eta, X, Y, n, slabs, pitch, pstat, stamps, cs); break;
        case 1: hipLaunchKernelGGL(k_fwd_bwd_fast3<ShapeC1>, dim3(grid, nchains), dim3(FAST_THREADS), 0, st, nd, qimg, eta, X, Y, n, slabs, pitch, pstat, stamps, cs); break;
        case 2: hipLaunchKernelGGL(k_fwd_bwd_fast3<ShapeTR>, dim3(grid, nchains), dim3(FAST_THREADS), 0, st, nd, qimg, eta, X, Y, n, slabs, pitch, pstat, stamps, cs); break;
        default: return -1;
    }
    return 0;
}
// forward-only launch: `nets` networks (grid.y), gx workgroups each
static inline int fast3_forward(int id, int gx, int nets, hipStream_t st, const float* qimgs, long img_stride, const float* X, long n,
                                float* fouts, long out_stride) {
    switch (id) {
        case 0: hipLaunchKernelGGL(k_forward_fast3<ShapeC2>, dim3(gx, nets), dim3(FAST_THREADS), 0, st, qimgs, img_stride, X, n, fouts, out_stride); break;
        case 1: hipLaunchKernelGGL(k_forward_fast3<ShapeC1>, dim3(gx, nets), dim3(FAST_THREADS), 0, st, qimgs, img_stride, X, n, fouts, out_stride); break;
        case 2: hipLaunchKernelGGL(k_forward_fast3<ShapeTR>, dim3(gx, nets), dim3(FAST_THREADS), 0, st, qimgs, img_stride, X, n, fouts, out_stride); break;
        default: return -1;
    }
    return 0;
}
#endif  // TBNN_NO_FAST_REGISTRY
