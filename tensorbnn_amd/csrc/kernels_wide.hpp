// Wide-layer path (hidden widths above the register budget of k_fwd_bwd_fast3: BASELINE configs[3]
// 10->200->200->200->1 and configs[4] 20->100->100->2), gfx950 f32 MFMA (v_mfma_f32_16x16x4_f32).
//
// A 200x200 layer has 169 dW tiles = 676 accumulator registers per lane: the gradient of a middle
// layer cannot ride along in the registers of the wave that runs the forward/backward chain.  The
// pass is therefore split in two kernels:
//
//   k_chain_wide   one wave = one 16-row tile through forward, likelihood and the delta chain with
//                  every activation in registers (the C/D layout of layer l-1 is the B-operand layout
//                  of layer l, kernels_fast.hpp).  The middle layers' weights do not fit in LDS
//                  (200x200 fp32 = 160 KB): every wave walks ONE weight stream (W_1 .. W_NM, then
//                  W_NM^T .. W_1^T, pre-swizzled into MFMA A-operand order by k_update; it lives in
//                  L2) and fetches the next k-group's operands itself with buffer loads, one chunk
//                  ahead of use (WIDE_DIRECT, round 3; WIDE_DIRECT=0 and shapes whose weights fit in
//                  LDS: the round-1 design, one stream per workgroup through a 4-slot LDS ring).
//                  dW of the FIRST layer (fan-in <= 32) and of the LAST layer (<= 2 outputs, VALU) are
//                  accumulated in registers as in the narrow kernel.  For every middle layer l the wave
//                  stores a_l (+ ones slot) and delta_l to HBM in 1-KB [16 rows][16 slots] blocks,
//                  already transposed for their consumer (WIDE_TBLOCK, round 3).
//   k_dw_wide      dW_l = delta_l^T a_l (contraction over ALL rows) for the middle layers: each
//                  workgroup owns a row range and the whole 13x13-tile output (43 tiles per wave),
//                  reads every operand with one 16-byte buffer load (no LDS; WIDE_TBLOCK=0: the blocks
//                  go through a 4-slot LDS ring to be turned) and writes its partial dW to a private
//                  slab; k_reduce_wide sums the slabs in fixed order (deterministic).
//
// Algorithmic HBM traffic of the split: 2 arrays (a_l, delta_l) x n x 4 B x padded width per middle
// layer, written once and read once (C4: 3.3 GB each way per gradient ~ 1 ms of HBM time against
// 3.1 ms of MFMA time at peak).
//
// Reference math: layer.py:278 (W@a+b), activationFunctions.py:36/49/62, likelihood.py:88-94,226-236,
// BNN_functions.py:23-32; reverse mode SURVEY A12.
#pragma once
#include <type_traits>
#include <algorithm>
#include "kernels_fast.hpp"
#include "wide_api.hpp"

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

#define WIDE_WAVES 4
// diagnostic build only (-DWIDE_STAMPS): shader-clock stamps of workgroup 0 / wave 0 during its SECOND block
#ifdef WIDE_STAMPS
__device__ unsigned long long g_wide_stamps[256];
#define WSTAMP(k) do { __builtin_amdgcn_sched_barrier(0); if (wstamp_on) g_wide_stamps[k] = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define WSTAMP(k) do { } while (0)
#endif
#define WIDE_THREADS 256
#define WIDE_RING 4
#ifndef WIDE_DW_PD
// dW_0 accumulators pinned to AccVGPRs (kernels_fast.hpp, mfma16_acc) next to VGPR-form chain MFMAs (build.py): -1.5 % at configs[4]
#ifndef WIDE_DW0_AGPR
#define WIDE_DW0_AGPR 1
#endif
#define WIDE_DW_PD 1   // k_dw_wide: row tiles in flight from HBM beyond the two parked in LDS (measured: 1 = 2 = 4)
#endif
#ifndef WIDE_PD
#define WIDE_PD 2      // weight-stream register sets: chunk c+2+k (k < WIDE_PD) is in flight while chunk c is consumed
#endif
#ifdef WIDE_DBG_NOBARRIER
#define WIDE_CHUNK_BARRIER() do {} while (0)           // timing experiment (results wrong)
#else
#define WIDE_CHUNK_BARRIER() do { if constexpr (!C::RESIDENT) __syncthreads(); } while (0)
#endif

// test hook: shapes listed here take the streamed-weights path even though their image would fit in LDS
template <class S> struct WideForceStream { static constexpr bool value = false; };

template <class S>
struct WideCfg {
    static constexpr int NL = S::NL;
    static_assert(NL >= 3, "the wide path needs at least one middle layer");
    static constexpr int in(int l) { return S::D[l]; }
    static constexpr int out(int l) { return S::D[l + 1]; }
    static constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
    static constexpr int r4(int a) { return (a + 3) & ~3; }
    static constexpr int LL = NL - 1;                 // last layer
    static constexpr int d_in = in(0), d_out = out(LL);
    // <= 2 outputs: the last layer runs on the VALU.  3 .. 16 outputs (round 6; network.add takes any stack, tensorBNN/network.py:173-191, the
    // likelihoods sum over [d_out, n], likelihood.py:88-94, 226-236): the last layer is one more MIDDLE layer -- its weights two more segments of
    // the stream, ONE output tile, the likelihood read off the tile, a_LL / delta_LL stored for k_dw_wide like every middle layer's
    static constexpr bool VL = d_out <= 2;
    static constexpr int NM = VL ? NL - 2 : NL - 1;   // middle layers 1 .. NM
    static_assert(d_out <= 16, "the last layer is one output tile at most");
    // a_l = input of layer l (l = 1..LL): in(l) real units + the ones pseudo-unit, in the padded slot order of
    // kernels_fast.hpp (slot_of / unit_of / ones_slot: identity on full 16-unit groups, the last partial group
    // spread over the lane groups first so that it needs only ceil(rem/4) MFMA k-steps)
    static constexpr int TR(int l) { return cdiv(in(l), 16); }          // register tiles (real units)
    static constexpr int TA(int l) { return cdiv(in(l) + 1, 16); }      // stored tiles (with the ones slot)
    static constexpr int ksteps(int K, int kg) { int rem = K - 16 * kg; return rem >= 16 ? 4 : (rem <= 0 ? 0 : (rem + 3) / 4); }
    static constexpr int KG(int K) { return cdiv(K, 16); }
    static constexpr int aroff(int l) { int o = 0; for (int m = 1; m < l; ++m) o += TR(m); return o; }   // a_l in the register file
    static constexpr int ACT_TILES = aroff(LL + 1);
    static constexpr int maxT() { int m = 0; for (int l = 1; l <= (VL ? LL : NL); ++l) m = TR(l) > m ? TR(l) : m; return m; }
    static constexpr int MAXT = maxT();
    // layer 0
    static constexpr int KG0 = KG(d_in), NT0 = cdiv(d_in + 1, 16), MT0 = TR(1);
    static constexpr int DW0_TILES = MT0 * NT0;
    // ---- permanent part of the weight image (LDS resident): W_0 operands, biases 0..NM, W_LL, b_LL
    static constexpr int W0_OFF = 0;
    static constexpr int W0_FLOATS = MT0 * KG0 * 256;
    static constexpr int boff(int l) { int o = W0_OFF + W0_FLOATS; for (int m = 0; m < l; ++m) o += 16 * TR(m + 1); return o; }   // bias of layer l <= NM
    static constexpr int WLP = VL ? 16 * TR(LL) : 0;                          // pitch of W_LL rows (the VALU last layer's own image)
    static constexpr int WL_OFF = boff(NM + 1);
    static constexpr int BL_OFF = WL_OFF + d_out * WLP;
    static constexpr int PERM_FLOATS = r4(BL_OFF + (VL ? d_out : 0));
    // ---- the weight stream: chunks of one k-group each.  Segment order F(1)..F(NM), B(NM)..B(1).
    //   F(l) chunk kg: granule t < TR(l+1): lane (i, g) holds W_l[16t+i][16kg+4g+s], s = 0..3
    //   B(l) chunk kg: granule u < TR(l)  : lane (i, g) holds W_l[16kg+4g+s][16u+i]
    static constexpr int GF(int l) { return r4(TR(l + 1)); }
    static constexpr int GB(int l) { return r4(TR(l)); }
    static constexpr int cF(int l) { int c = 0; for (int m = 1; m < l; ++m) c += KG(in(m)); return c; }           // first chunk of F(l)
    static constexpr int cB(int l) { int c = cF(NM + 1); for (int m = NM; m > l; --m) c += KG(out(m)); return c; } // first chunk of B(l)
    static constexpr int NCH = cB(0);
    static constexpr int NCHF = cF(NM + 1);            // chunks of the forward segments alone (forward-only kernel)
    static constexpr int chunk_gran(int c) {
        for (int l = 1; l <= NM; ++l) if (c >= cF(l) && c < cF(l) + KG(in(l))) return GF(l);
        for (int l = NM; l >= 1; --l) if (c >= cB(l) && c < cB(l) + KG(out(l))) return GB(l);
        return 0;
    }
    static constexpr int chunk_tiles(int c) {                  // tiles a wave reads from chunk c (unpadded)
        for (int l = 1; l <= NM; ++l) if (c >= cF(l) && c < cF(l) + KG(in(l))) return TR(l + 1);
        for (int l = NM; l >= 1; --l) if (c >= cB(l) && c < cB(l) + KG(out(l))) return TR(l);
        return 0;
    }
    static constexpr int chunk_ksteps(int c) {                 // k-steps of chunk c (the last k-group of a segment may be partial)
        for (int l = 1; l <= NM; ++l) if (c >= cF(l) && c < cF(l) + KG(in(l))) return ksteps(in(l), c - cF(l));
        for (int l = NM; l >= 1; --l) if (c >= cB(l) && c < cB(l) + KG(out(l))) return ksteps(out(l), c - cB(l));
        return 0;
    }
    static constexpr int maxGran() { int m = 0; for (int c = 0; c < NCH; ++c) m = chunk_gran(c) > m ? chunk_gran(c) : m; return m; }
    // 1-KB granule j of chunk c sits at PERM + (j * GS + c) KB: the granules of ONE chunk -- which every workgroup of
    // the grid reads at about the same time -- are GS KB apart (GS odd), so they spread over the L2 channels
    // instead of queueing on the one or two channels a contiguous 16-KB chunk maps to (measured: 2.6x on k_chain_wide)
    static constexpr int PX = 16 * NT0 + 4;                     // x image pitch
    static constexpr int XIMG_FLOATS = 16 * PX;
    static constexpr int DZB = 4, PZ = 16 * DZB + 4;            // delta_0 transposed DZB tiles at a time
    // dW_0's accumulators on the asm form (pinned to AccVGPRs, no software wait states): never where a group of delta_0 tiles -- DZB of them, or
    // the MT0 % DZB left over -- takes exactly TWO accumulators in turn (one other MFMA between two visits: the distance at which the asm
    // form read a stale accumulator; one accumulator back to back and three or more in turn are the tested cases)
    static constexpr int dw0_turn(int tiles) { return NT0 * tiles; }
    static constexpr bool DW0_FAR = dw0_turn(DZB < MT0 ? DZB : MT0) > 1 && dw0_turn(DZB < MT0 ? DZB : MT0) != 2 &&
                                    (MT0 <= DZB || MT0 % DZB == 0 || dw0_turn(MT0 % DZB) != 2);
    static constexpr int SCR_FLOATS = 16 * PZ;
    // RESIDENT: the whole image (dense, unpadded chunks) fits in LDS next to the per-wave scratch -> no stream, no
    // ring, no barriers in the row loop (configs[4]: 100-wide layers); otherwise the 4-slot ring (configs[3])
    static constexpr int dense_off(int c) { int o = PERM_FLOATS; for (int k = 0; k < c; ++k) o += chunk_tiles(k) * 256; return o; }
    static constexpr bool RESIDENT = !WideForceStream<S>::value &&
                                     (dense_off(NCH) + WIDE_WAVES * (XIMG_FLOATS + SCR_FLOATS)) * 4 + 64 <= 160 * 1024;
    static constexpr int GS = NCH | 1;
    static constexpr int gran_off(int c, int j) { return RESIDENT ? dense_off(c) + j * 256 : PERM_FLOATS + (j * GS + c) * 256; }
    static constexpr int gran_step(int c) { return gran_off(c, 1) - gran_off(c, 0); }
    static constexpr int IMG_FLOATS = RESIDENT ? dense_off(NCH) : PERM_FLOATS + maxGran() * GS * 256;
    static constexpr int SLOT_FLOATS = maxGran() * 256;
    static constexpr int NGW = maxGran() / WIDE_WAVES;          // granules per wave per chunk
    static_assert(NCH >= 3, "ring priming assumes >= 3 chunks");
    // ---- LDS layout (floats)
    static constexpr int RING_OFF = PERM_FLOATS;
    static constexpr int XIMG_OFF = RESIDENT ? IMG_FLOATS : RING_OFF + WIDE_RING * SLOT_FLOATS;
    static constexpr int SCR_OFF = XIMG_OFF + WIDE_WAVES * XIMG_FLOATS;
    static constexpr int LDS_FLOATS = SCR_OFF + WIDE_WAVES * SCR_FLOATS;
    // ---- parameters
    static constexpr int offW(int l) { int p = 0; for (int m = 0; m < l; ++m) p += in(m) * out(m) + out(m); return p; }
    static constexpr int P() { return offW(NL); }
    // compact slab of k_chain_wide: [layer 0 params][last layer params]
    static constexpr int SA_L0 = 0, SA_LL = in(0) * out(0) + out(0);
    static constexpr int SA_FLOATS = r4(SA_LL + (VL ? in(LL) * d_out + d_out : 0));
    // ---- HBM activation / delta arrays (per middle layer l): blocks [row tile][tile][16 rows][16 slots]
    static constexpr int TZ(int l) { return TR(l + 1); }        // tiles of delta_l (outputs of layer l)
    static constexpr long act_off(int l, long ntiles) { long o = 0; for (int m = 1; m < l; ++m) o += ntiles * (TA(m) + TZ(m)) * 256; return o; }   // a_l
    static constexpr long dz_off(int l, long ntiles) { return act_off(l, ntiles) + ntiles * TA(l) * 256; }
    static constexpr long store_floats(long ntiles) { return act_off(NM + 1, ntiles); }
    // ---- k_dw_wide: tiles of dW_l per wave
    static constexpr int QM(int l) { return TZ(l) / 4; }                        // whole M tiles per wave (m = wave + 4j)
    static constexpr int RM(int l) { return TZ(l) % 4; }                        // left-over M tiles, shared out as (m, u) pairs
    static constexpr int QP(int l) { return cdiv(RM(l) * TA(l), 4); }           // pairs per wave
    static constexpr int DWT(int l) { return QM(l) * TA(l) + QP(l); }           // accumulator tiles per wave
    static constexpr int maxDWT() { int m = 0; for (int l = 1; l <= NM; ++l) m = DWT(l) > m ? DWT(l) : m; return m; }
    static constexpr int SB(int l) { return TA(l) + TZ(l); }                    // 1-KB blocks per row tile
    static constexpr int maxSB() { int m = 0; for (int l = 1; l <= NM; ++l) m = SB(l) > m ? SB(l) : m; return m; }
    static constexpr int DW_SLOT_FLOATS = maxSB() * 256;
    static constexpr int DW_NGW = cdiv(maxSB(), WIDE_WAVES);
    static constexpr int slabB_off(int l) { int o = 0; for (int m = 1; m < l; ++m) o += r4(in(m) * out(m) + out(m)); return o; }   // within one WG's slab
    static constexpr int SB_FLOATS = slabB_off(NM + 1);
#ifndef WIDE_LL_COST8
#define WIDE_LL_COST8 4          // eighths of a tile per stored block (dw_cost)
#endif
#ifndef WIDE_DW_G
#define WIDE_DW_G 2              // row tiles whose operands a wave of the one-tile layer requests at once (two such groups in flight)
#endif
    // share of k_dw_wide's workgroups: by MFMA count -- but the one-tile last layer of a 3 .. 16-output network (QM = 0: a few pair MFMAs per row
    // tile behind TA + 1 block loads) is bound by its loads: priced by its blocks (10 -> 200 -> 200 -> 10 at 1e5 rows, k_dw_wide: 267 us with the MFMA
    // count alone, the last layer's 8 % of the workgroups being the long pole)
    static constexpr long dw_cost(int l) { return (!VL && l == LL) ? (DWT(l) > WIDE_LL_COST8 * SB(l) / 8 ? DWT(l) : WIDE_LL_COST8 * SB(l) / 8) : (long)DWT(l); }
    // k_dw_wide workgroups per CU: a wave that owns few accumulator tiles (configs[4]: 13 tiles, 52 MFMAs between two
    // barriers) leaves the MFMA pipe idle at every barrier / LDS round trip; a second resident workgroup fills it when
    // two rings fit in LDS and two waves fit in a SIMD's registers
#ifndef WIDE_DW_OCC_MAX
#define WIDE_DW_OCC_MAX 2
#endif
#if !defined(WIDE_TBLOCK) || WIDE_TBLOCK
    // (transposed blocks, operands straight into registers: no LDS; two waves per SIMD when accumulators + operand sets fit)
    static constexpr int maxSmall() { int m = 0; for (int l = 1; l <= NM; ++l) { int v = TA(l) + 2 * QM(l) + 4 * QP(l); m = v > m ? v : m; } return m; }
    // (a layer with two M tiles per wave takes its a-blocks in pairs -- dw_wide_layer --: two operand blocks and their refills live at once)
    static constexpr bool anyQM2() { for (int l = 1; l <= NM; ++l) if (QM(l) == 2) return true; return false; }
    static constexpr int DW_OCC = (WIDE_DW_OCC_MAX >= 2 && 4 * maxDWT() + 4 * maxSmall() + 40 + (anyQM2() ? 36 : 0) <= 232) ? 2 : 1;
#else
    static constexpr int DW_OCC = (WIDE_DW_OCC_MAX >= 2 && 2 * WIDE_RING * DW_SLOT_FLOATS * 4 <= 150 * 1024 &&
                                   4 * maxDWT() + 4 * DW_NGW * WIDE_DW_PD + 48 <= 232) ? 2 : 1;
#endif
};

// ---------------------------------------------------------------------------------------------
// k_chain_wide
// ---------------------------------------------------------------------------------------------
template <class S>
struct WideRegs {
    using C = WideCfg<S>;
    f32x4 a[C::MAXT];               // the current layer's input a_l, D layout: tile t reg j of lane (r, g) = unit 16t+4g+j of row r
    float x[C::KG0 * 4];            // x[4kg+s] = X[row][16kg+4g+s]
    // relu hidden layers: act'(a_l) for the delta chain is one bit per element (bit 4t+j of word (4t+j)/32),
    // kept for l = 1..NM -- no activation stays in registers and nothing is re-read
    unsigned relu[C::NM > 0 ? C::NM : 1][(4 * C::MAXT + 31) / 32];
};

// a -> relu mask bits of layer slot l (index l-1)
template <class S, int NT>
__device__ __forceinline__ void relu_mask_make(unsigned (&m)[(4 * WideCfg<S>::MAXT + 31) / 32], const f32x4* a) {
#pragma unroll
    for (int w = 0; w < (4 * WideCfg<S>::MAXT + 31) / 32; ++w) m[w] = 0u;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) m[(4 * t + r) / 32] |= (a[t][r] > 0.f) ? (1u << ((4 * t + r) % 32)) : 0u;
}
// acc * relu'(a): sign-extended bit field -> all-ones / zero -> AND (2 VALU ops)
__device__ __forceinline__ float relu_mask_apply(unsigned m, int bit, float acc) {
    const int keep = __builtin_amdgcn_sbfe(m, bit, 1);
    return __int_as_float(__float_as_int(acc) & keep);
}

// top of chunk c: park the chunk fetched one step ago (c+2) in its ring slot, fetch chunk c+3.
template <class S, int c, bool FWD = false>
__device__ __forceinline__ void wide_stage(int base, f32x4 (&stgs)[WIDE_PD][WideCfg<S>::NGW], float* __restrict__ ring,
                                           const float* __restrict__ img, int wave, int lane) {
    using C = WideCfg<S>;
#ifdef WIDE_DBG_NOSTREAM
    return;                                            // timing experiment: ring never refilled (results wrong)
#endif
    if constexpr (C::RESIDENT) return;
    constexpr int NCHE = FWD ? C::NCHF : C::NCH;       // the forward-only kernel cycles through the F segments only
    constexpr int cw = (c + 2) % NCHE, cl = (c + 2 + WIDE_PD) % NCHE;
    f32x4 (&stg)[C::NGW] = stgs[0];                    // oldest set: chunk c+2
    float* dst = ring + ((base + c + 2) & (WIDE_RING - 1)) * C::SLOT_FLOATS + wave * 256 + lane * 4;
#pragma unroll
    for (int j = 0; j < C::NGW; ++j)
        if (j < C::chunk_gran(cw) / WIDE_WAVES) {
#ifdef WIDE_DBG_NOPARK
            asm volatile("" :: "v"(stg[j]));           // timing experiment: loads stay alive, no LDS write (results wrong)
#else
            *reinterpret_cast<f32x4*>(dst + j * WIDE_WAVES * 256) = stg[j];
#endif
        }
    // rotate the sets (register renaming inside the unrolled block; plain moves only at the loop back-edge)
#pragma unroll
    for (int k = 0; k + 1 < WIDE_PD; ++k)
#pragma unroll
        for (int j = 0; j < C::NGW; ++j) stgs[k][j] = stgs[k + 1][j];
    const float* src = img + C::gran_off(cl, 0) + wave * C::gran_step(cl) + lane * 4;
#pragma unroll
    for (int j = 0; j < C::NGW; ++j)
        if (j < C::chunk_gran(cl) / WIDE_WAVES) {
#ifdef WIDE_DBG_NOLOAD
            asm volatile("" : "+v"(stgs[WIDE_PD - 1][j]));   // timing experiment: no global load (results wrong)
#else
            stgs[WIDE_PD - 1][j] = *reinterpret_cast<const f32x4*>(src + j * WIDE_WAVES * C::gran_step(cl));
#endif
        }
}

// ---- hand-scheduled chunk (WIDE_HANDSCHED, default): the compiler issues the 13 A-operand reads of the next chunk in one
// burst in front of the chunk barrier and drains them there (s_waitcnt lgkmcnt(0) precedes every s_barrier): ~300 of a
// chunk's ~2,150 cycles at configs[3].  Here the MFMAs of a chunk run in tile GROUPS of two (k-step-major inside a group: an
// accumulator is revisited every second MFMA, 64 cycles > the 40-cycle dependent latency), so a group's operand registers are
// dead after 8 MFMAs and take the SAME tiles of chunk c+1 right then -- that slot was parked two barriers ago and is visible
// to every wave.  Only the last group's operands have no MFMAs behind them in this chunk: they are requested at the top of
// the next one, where they are needed last.  Scheduling fences (nothing crosses sched_barrier(0)) pin the order.
#ifndef WIDE_HANDSCHED
#define WIDE_HANDSCHED 1
#endif
#define WIDE_FENCE() __builtin_amdgcn_sched_barrier(0)
template <class S>
struct WideSched {
    using C = WideCfg<S>;
    static constexpr int ngroups(int nt) { return nt / 2 > 1 ? nt / 2 : 1; }
    static constexpr int gstart(int nt, int gi) { return 2 * gi; }
    static constexpr int gend(int nt, int gi) { return gi == ngroups(nt) - 1 ? nt : 2 * gi + 2; }
    static constexpr int deferred_from(int nt) { return gstart(nt, ngroups(nt) - 1); }   // first tile of the last group
};
// address of tile t of chunk cc in ring position rc (compile-time cc; RESIDENT: the image itself)
// WIDE_DIRECT (default; streamed shapes only): the A operands do not go through LDS at all.  The image k_update maintains is
// already in MFMA A-operand order -- granule t of a chunk is [lane][4], exactly the 16 bytes lane needs for tile t -- so every
// wave loads its operands straight from L2 (the 4 waves of a workgroup walk the same chunks: the later ones hit the CU's L1)
// into the registers the hand-scheduled chunk frees, ONE CHUNK (~1,700 cycles) ahead of their use.  No ring, no parking writes,
// no chunk barrier, no staging registers; the waves of a workgroup no longer wait for each other.  WIDE_DIRECT=0: the 4-slot
// LDS ring of rounds 1-2.
#ifndef WIDE_DIRECT
#define WIDE_DIRECT 1
#endif
template <class S> struct WideDirect { static constexpr bool value = WIDE_DIRECT && WIDE_HANDSCHED && !WideCfg<S>::RESIDENT; };
template <class S, bool FWD>
__device__ __forceinline__ const float* wide_tile_ptr(int base, int c, int cc_gran_off, const float* __restrict__ ring, int lane, int t) {
    using C = WideCfg<S>;
    return (C::RESIDENT ? ring + cc_gran_off : ring + ((base + c) & (WIDE_RING - 1)) * C::SLOT_FLOATS) + lane * 4 + t * 256;
}
// direct mode: tile t of stream chunk cc in the global image (granules of one chunk are GS KB apart: L2-channel spreading).
// A buffer load: resource descriptor in SGPRs, the granule's byte offset an SGPR (+ `opq`, an opaque zero renewed every row
// tile so that the loads are not hoisted out of the row loop), the lane's 16 bytes the only VGPR -- no address arithmetic on
// the VALU (global_load with 64-bit pointers cost two VALU adds per load: +280 instructions per 16-row tile).
struct WideImg { __amdgpu_buffer_rsrc_t rs; int opq; };
// ---- stored blocks, transposed layout (WIDE_TBLOCK, default).  A 1-KB block holds [16 rows][16 slots] of a_l or delta_l.
// k_dw_wide contracts over the ROWS: its MFMA operand for lane (i, g) and k-step s is (row 4s+g, slot i).  Round 1-2 stored a
// block row-major (one 16-B store per lane of the D layout) and k_dw_wide re-read it lane-linearly through an LDS ring.  Here
// element (row, slot) sits at slot*16 + (row % 4)*4 + row / 4: a lane's four k-steps are 16 contiguous bytes, so k_dw_wide
// loads its operands straight from L2 into registers with ONE 16-B buffer load per block (no LDS ring, no parking, no barrier),
// and the chain kernel pays four 4-B stores per block instead of one 16-B store.
#ifndef WIDE_TBLOCK
#define WIDE_TBLOCK 1
#endif
#define WIDE_RSRC_FLAGS 0x00020000
// byte offset of this lane's component j in a block: writer lane (r = i16, g) holds (row r, slot 4g+j)
__device__ __forceinline__ int tblk_wr_off(int i16, int g) { return ((4 * g) * 16 + (i16 & 3) * 4 + (i16 >> 2)) * 4; }   // + j * 64
// reader lane (i = slot, g = row % 4): 16 bytes = rows g, 4+g, 8+g, 12+g of slot i
__device__ __forceinline__ int tblk_rd_off(int lane) { return ((lane & 15) * 16 + (lane >> 4) * 4) * 4; }
// store the D-layout tile v as block `blk` of the array behind `rs` (one row tile's blocks)
__device__ __forceinline__ void tblk_store(__amdgpu_buffer_rsrc_t rs, int voff, int blk, const f32x4& v) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float f = v[j];                       // (__builtin_bit_cast applied to v[j] itself reads element 0 of the vector)
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(f), rs, voff, blk * 1024 + j * 64, 0);
    }
}
__device__ __forceinline__ f32x4 tblk_load_d(__amdgpu_buffer_rsrc_t rs, int voff, int blk) {      // back in the D layout (4 x 4-B loads)
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, blk * 1024 + j * 64, 0));
    return v;
}
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
template <class S>
__device__ __forceinline__ f32x4 wide_tile_direct(const WideImg& im, int cc, int lane, int t) {
    using C = WideCfg<S>;
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(im.rs, lane * 16, im.opq + (int)((C::PERM_FLOATS + (t * C::GS + cc) * 256) * 4), 0);
    return __builtin_bit_cast(f32x4, v);
}
// chunk c of the stream: Anx[t] holds (or has in flight) the A operand of tile t for t < deferred_from(tiles of chunk c-1)
template <class S, int c, bool FWD>
__device__ __forceinline__ void wide_chunk(int base, f32x4 (&Anx)[WideCfg<S>::MAXT], f32x4* acc, const f32x4& bk, f32x4 (&stgs)[WIDE_PD][WideCfg<S>::NGW],
                                           float* __restrict__ ring, const float* __restrict__ img, const WideImg& im, int wave, int lane) {
    using C = WideCfg<S>;
    using W = WideSched<S>;
    constexpr int NCHE = FWD ? C::NCHF : C::NCH;
    constexpr int cc = c % NCHE, cp = (c + NCHE - 1) % NCHE, cn = (c + 1) % NCHE;
    constexpr int NT = C::chunk_tiles(cc), NTP = C::chunk_tiles(cp), NTN = C::chunk_tiles(cn);
    constexpr int K = C::chunk_ksteps(cc);
    // the operands the previous chunk could not request behind its own MFMAs
    constexpr bool DIRECT = WideDirect<S>::value;
#pragma unroll
    for (int t = W::deferred_from(NTP); t < NT; ++t)
        Anx[t] = DIRECT ? wide_tile_direct<S>(im, cc, lane, t) : *reinterpret_cast<const f32x4*>(wide_tile_ptr<S, FWD>(base, c, C::gran_off(cc, 0), ring, lane, t));
    WIDE_FENCE();
    constexpr int NG = W::ngroups(NT);
    sfor<0, NG>(SFOR_LAMBDA(gi) {
        constexpr int gi = SFOR_VAL(gi), t0 = W::gstart(NT, gi), t1 = W::gend(NT, gi);
#pragma unroll
        for (int s = 0; s < K; ++s)
#pragma unroll
            for (int t = t0; t < t1; ++t) acc[t] = mfma16(Anx[t][s], bk[s], acc[t]);
        WIDE_FENCE();
        if constexpr (gi == 0 && !DIRECT) {       // park chunk c+2, fetch chunk c+2+PD: under the second group's MFMAs
            wide_stage<S, c, FWD>(base, stgs, ring, img, wave, lane);
            WIDE_FENCE();
        }
        if constexpr (gi + 1 < NG) {
#pragma unroll
            for (int t = t0; t < t1; ++t)
                if (t < NTN)
                    Anx[t] = DIRECT ? wide_tile_direct<S>(im, cn, lane, t)
                                    : *reinterpret_cast<const f32x4*>(wide_tile_ptr<S, FWD>(base, c + 1, C::gran_off(cn, 0), ring, lane, t));
            WIDE_FENCE();
        }
    });
    if constexpr (!DIRECT) WIDE_CHUNK_BARRIER();
}

// A operands of chunk c (already visible in its ring slot: parked two chunks earlier, one barrier ago)
template <class S, int c, bool FWD = false>
__device__ __forceinline__ void wide_load_A(int base, f32x4 (&A)[WideCfg<S>::MAXT], const float* __restrict__ ring, int lane) {
    using C = WideCfg<S>;
    // RESIDENT: `ring` is the LDS base and the chunk sits at its image offset
    constexpr int cc = c % (FWD ? C::NCHF : C::NCH);
    const float* sl = (C::RESIDENT ? ring + C::gran_off(cc, 0) : ring + ((base + c) & (WIDE_RING - 1)) * C::SLOT_FLOATS) + lane * 4;
#pragma unroll
    for (int t = 0; t < C::MAXT; ++t)
        if (t < C::chunk_tiles(cc)) A[t] = *reinterpret_cast<const f32x4*>(sl + t * 256);
}

// FWD: forward pass only (network.predict, network.py:141-171): no likelihood, no delta chain, no stores;
// fout[d_out][n] receives the network output.  Y, eta, store, slabA, pstat are unused (null).
template <class S, bool FWD = false>
__global__ __launch_bounds__(WIDE_THREADS, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_chain_wide(
    NetDev nd, const float* __restrict__ qimg, const float* __restrict__ eta,
    const float* __restrict__ X, const float* __restrict__ Y, long n,
    float* __restrict__ store, float* __restrict__ slabA, double* __restrict__ pstat, float* __restrict__ fout)
{
    using C = WideCfg<S>;
    static_assert(!FWD || C::RESIDENT || C::NCHF >= 3, "forward-only ring priming needs >= 3 forward chunks");
    static_assert(C::LDS_FLOATS * 4 + 64 <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS];
    __shared__ double red[WIDE_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: see k_dw_wide
    const int i16 = lane & 15, g = lane >> 4;
    constexpr int d_in = C::d_in, d_out = C::d_out, NM = C::NM, LL = C::LL;
    const long ntiles = (n + 15) / 16;
    const long nblk = (ntiles + WIDE_WAVES - 1) / WIDE_WAVES;

    // ---- prologue: permanent image -> LDS; prime the ring (chunks 0, 1 in slots 0, 1; chunk 2 in registers)
    for (int e = tid; e < (C::RESIDENT ? C::IMG_FLOATS : C::PERM_FLOATS) / 4; e += WIDE_THREADS)
        reinterpret_cast<float4*>(lds)[e] = reinterpret_cast<const float4*>(qimg)[e];
    float* ring = C::RESIDENT ? lds : lds + C::RING_OFF;
    f32x4 stg[WIDE_PD][C::NGW];
    constexpr bool DIRECT = WideDirect<S>::value;
    sfor<0, 2>(SFOR_LAMBDA(c) {
        constexpr int c = SFOR_VAL(c);
#pragma unroll
        for (int j = 0; j < C::NGW; ++j)
            if (!C::RESIDENT && !DIRECT && j < C::chunk_gran(c) / WIDE_WAVES)
                *reinterpret_cast<f32x4*>(ring + c * C::SLOT_FLOATS + (wave + j * WIDE_WAVES) * 256 + lane * 4) =
                    *reinterpret_cast<const f32x4*>(qimg + C::gran_off(c, 0) + (wave + j * WIDE_WAVES) * C::gran_step(c) + lane * 4);
    });
    sfor<0, WIDE_PD>(SFOR_LAMBDA(k) {
        constexpr int k = SFOR_VAL(k), c = (2 + k) % (FWD ? C::NCHF : C::NCH);
#pragma unroll
        for (int j = 0; j < C::NGW; ++j)
            stg[k][j] = (!C::RESIDENT && !DIRECT && j < C::chunk_gran(c) / WIDE_WAVES)
                            ? *reinterpret_cast<const f32x4*>(qimg + C::gran_off(c, 0) + (wave + j * WIDE_WAVES) * C::gran_step(c) + lane * 4)
                            : f32x4{0.f, 0.f, 0.f, 0.f};
    });
    float* ximg = lds + C::XIMG_OFF + wave * C::XIMG_FLOATS;
    float* scr = lds + C::SCR_OFF + wave * C::SCR_FLOATS;
    for (int e = lane; e < C::XIMG_FLOATS; e += 64) ximg[e] = 0.f;
    __syncthreads();
    if (g == 0) ximg[i16 * C::PX + d_in] = 1.f;               // ones column of the x image (db_0)
#if WIDE_HANDSCHED
    // operands of chunk 0 that the steady state requests behind the MFMAs of the chunk before it (wide_chunk)
    f32x4 Anx[C::MAXT];
    {
        constexpr int NCHE = FWD ? C::NCHF : C::NCH;
#pragma unroll
        for (int t = 0; t < C::MAXT; ++t) {
            Anx[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (t < WideSched<S>::deferred_from(C::chunk_tiles(NCHE - 1)) && t < C::chunk_tiles(0))
                Anx[t] = DIRECT ? wide_tile_direct<S>(WideImg{__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(qimg), 0, C::IMG_FLOATS * 4, 0x00020000), 0}, 0, lane, t) : *reinterpret_cast<const f32x4*>(wide_tile_ptr<S, FWD>(0, 0, C::gran_off(0, 0), ring, lane, t));
        }
    }
#endif

    const float sigma = FWD ? 1.f : lik_sigma(nd, eta);
    const float inv_var = 1.f / (sigma * sigma);
    double stat = 0.0;
    f32x4 dW0[C::DW0_TILES];
#pragma unroll
    for (int t = 0; t < C::DW0_TILES; ++t) dW0[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // (the VALU last layer's per-lane partial sums; nothing when the last layer is an MFMA layer)
    constexpr int LRO = C::VL ? d_out : 1, LRT = C::VL ? C::TR(LL) : 1;
    f32x4 accL[LRO][LRT];
    float accbL[LRO];
#pragma unroll
    for (int o = 0; o < LRO; ++o) {
        accbL[o] = 0.f;
#pragma unroll
        for (int t = 0; t < LRT; ++t) accL[o][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // targets of a row: <= 2 outputs: y[o] in every lane group; else the D layout of the output tile (lane (row, g): slots 4g .. 4g + 3)
    constexpr int YN = C::VL ? d_out : 4;
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
    fetch((long)blockIdx.x * WIDE_WAVES + wave);
    int base = 0;                                              // ring slot of chunk 0 of the current block

    for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        // the weight stream is the same for every block: hide the pointer from loop-invariant code motion,
        // or every chunk load is hoisted out of the row loop (and spilled)
#ifdef WIDE_STAMPS
        const bool wstamp_on = blockIdx.x == 0 && tid == 0 && blk == (long)blockIdx.x + gridDim.x;
#endif
        WSTAMP(0);
        int opaque0 = 0;
        asm volatile("" : "+s"(opaque0));
        const float* img = qimg + opaque0;
        const WideImg im = {__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(qimg), 0, C::IMG_FLOATS * 4, 0x00020000), opaque0};
        const long tile = blk * WIDE_WAVES + wave;
        const bool tvalid = tile < ntiles;
#ifdef WIDE_DBG_STORE0
        const long stile = tile & 1023;                // timing experiment: a_l / delta_l blocks stay in L2 (results wrong)
#else
        const long stile = tile;
#endif
        const bool rvalid = tile * 16 + i16 < n;
        WideRegs<S> T;
        float y[YN];
#pragma unroll
        for (int k = 0; k < C::KG0 * 4; ++k) T.x[k] = xn[k];
#pragma unroll
        for (int o = 0; o < YN; ++o) y[o] = yn[o];
        fetch((blk + gridDim.x) * WIDE_WAVES + wave);
        // x image for dW_0 (slots < d_in only: the ones column stays)
#pragma unroll
        for (int k = 0; k < C::KG0 * 4; ++k) {
            const int u = unit_of(d_in, 16 * (k / 4) + 4 * g + (k % 4), false);
            if (u >= 0) ximg[i16 * C::PX + u] = T.x[k];
        }

        // ---- layer 0 (weights resident in LDS)
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
                    for (int t = 0; t < MT; ++t) acc[t] = mfma16(A[t][s], T.x[4 * kg + s], acc[t]);
            }
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) T.a[t][r] = actc_fwd<S::act(0)>(acc[t][r]);
            if constexpr (S::act(0) == TBNN_ACT_RELU) relu_mask_make<S, MT>(T.relu[0], T.a);
        }

        WSTAMP(1);
        // ---- middle layers, forward: a_l -> a_{l+1}; a_l (+ ones slot) goes to HBM for k_dw_wide
        sfor<1, NM + 1>(SFOR_LAMBDA(l) {
            constexpr int l = SFOR_VAL(l);
            {   // store a_l
                f32x4 v[C::MAXT + 1];
#pragma unroll
                for (int t = 0; t < C::TA(l); ++t) {
                    v[t] = t < C::TR(l) ? T.a[t] : f32x4{0.f, 0.f, 0.f, 0.f};
                    constexpr int os = ones_slot(C::in(l));
                    if (t == os / 16 && g == (os % 16) / 4) v[t][os % 4] = 1.f;
                }
                if (!FWD && tvalid) {
#if WIDE_TBLOCK
                    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(store + C::act_off(l, ntiles) + ((size_t)stile * C::TA(l)) * 256, 0,
                                                                                         C::TA(l) * 1024, WIDE_RSRC_FLAGS);
#pragma unroll
                    for (int t = 0; t < C::TA(l); ++t) tblk_store(rsw, tblk_wr_off(i16, g), t, v[t]);
#else
                    float* p = store + C::act_off(l, ntiles) + ((size_t)stile * C::TA(l)) * 256 + i16 * 16 + g * 4;
#pragma unroll
                    for (int t = 0; t < C::TA(l); ++t) *reinterpret_cast<f32x4*>(p + t * 256) = v[t];
#endif
                }
            }
            constexpr int MT = C::TR(l + 1);
            f32x4 acc[MT];
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t] = *reinterpret_cast<const f32x4*>(lds + C::boff(l) + 16 * t + 4 * g);
            sfor<0, C::KG(C::in(l))>(SFOR_LAMBDA(kg) {
                constexpr int kg = SFOR_VAL(kg), c = C::cF(l) + kg;
#if WIDE_HANDSCHED
                wide_chunk<S, c, FWD>(base, Anx, acc, T.a[kg], stg, ring, img, im, wave, lane);
                WSTAMP(9 + 2 * c);
#else
                wide_stage<S, c, FWD>(base, stg, ring, img, wave, lane);
                f32x4 Acur[C::MAXT];
                wide_load_A<S, c, FWD>(base, Acur, ring, lane);
#pragma unroll
                for (int s = 0; s < C::ksteps(C::in(l), kg); ++s)
#pragma unroll
                    for (int t = 0; t < MT; ++t) acc[t] = mfma16(Acur[t][s], T.a[kg][s], acc[t]);
                WSTAMP(8 + 2 * c);
                WIDE_CHUNK_BARRIER();
                WSTAMP(9 + 2 * c);
#endif
            });
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) T.a[t][r] = actc_fwd<S::act(l)>(acc[t][r]);
            if constexpr (S::act(l) == TBNN_ACT_RELU && l + 1 <= NM) relu_mask_make<S, MT>(T.relu[l], T.a);
        });

        WSTAMP(2);
        // ---- last layer on the VALU: f_o = b_o + sum_u W[o][u] a_LL[u]
        f32x4 dz[C::MAXT];
        if constexpr (!C::VL) {
            // MFMA last layer: T.a[0] is the output tile (lane (row i16, g): slots 4g .. 4g + 3); the likelihood reads it, every (row, output)
            // element once, and delta_LL (w.r.t. the pre-activation) stands in the D layout the delta chain starts from
            if constexpr (FWD) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int u = unit_of(d_out, 4 * g + r, false);
                    if (rvalid && u >= 0) fout[(size_t)u * n + tile * 16 + i16] = T.a[0][r];      // [d_out][n]
                }
                base = (base + C::NCHF) & (WIDE_RING - 1);
                continue;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int u = unit_of(d_out, 4 * g + r, false);
                    dz[0][r] = (rvalid && u >= 0) ? lik_delta<S>(T.a[0][r], y[r], inv_var, true, stat) : 0.f;
                }
            }
        } else {
            constexpr int TP = C::TR(LL);
            float dzl[d_out];
#pragma unroll
            for (int o = 0; o < d_out; ++o) {
                float p = 0.f;
#pragma unroll
                for (int t = 0; t < TP; ++t) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(lds + C::WL_OFF + o * C::WLP + 16 * t + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) p = fmaf(w[r], T.a[t][r], p);
                }
                p += __shfl_xor(p, 16, 64);
                p += __shfl_xor(p, 32, 64);
                const float fi = actc_fwd<S::LACT>(p + lds[C::BL_OFF + o]);
                if constexpr (FWD) {
                    if (rvalid && g == 0) fout[(size_t)o * n + tile * 16 + i16] = fi;      // [d_out][n]
                    dzl[o] = 0.f;
                } else {
                    dzl[o] = rvalid ? lik_delta<S>(fi, y[o], inv_var, g == 0, stat) : 0.f;
                    accbL[o] += dzl[o];
                }
            }
            if constexpr (FWD) { base = (base + C::NCHF) & (WIDE_RING - 1); continue; }
#pragma unroll
            for (int t = 0; t < TP; ++t) {
                f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int o = 0; o < d_out; ++o) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(lds + C::WL_OFF + o * C::WLP + 16 * t + 4 * g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        accL[o][t][r] = fmaf(dzl[o], T.a[t][r], accL[o][t][r]);
                        d[r] = fmaf(w[r], dzl[o], d[r]);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) dz[t][r] = actc_bwd_mul<S::act(C::NL - 2)>(d[r], T.a[t][r]);
            }
        }

        WSTAMP(3);
        // ---- delta chain through the middle layers: delta_l (held in dz) -> delta_{l-1}
        sfor<0, NM>(SFOR_LAMBDA(li) {
            constexpr int l = NM - SFOR_VAL(li);
            if (tvalid) {       // delta_l -> HBM
#if WIDE_TBLOCK
                const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(store + C::dz_off(l, ntiles) + ((size_t)stile * C::TZ(l)) * 256, 0,
                                                                                     C::TZ(l) * 1024, WIDE_RSRC_FLAGS);
#pragma unroll
                for (int t = 0; t < C::MAXT; ++t)
                    if (t < C::TZ(l)) tblk_store(rsw, tblk_wr_off(i16, g), t, dz[t]);
#else
                float* p = store + C::dz_off(l, ntiles) + ((size_t)stile * C::TZ(l)) * 256 + i16 * 16 + g * 4;
#pragma unroll
                for (int t = 0; t < C::MAXT; ++t)
                    if (t < C::TZ(l)) *reinterpret_cast<f32x4*>(p + t * 256) = dz[t];
#endif
            }
            constexpr int MU = C::TR(l);
            // a_l (for act') comes back from the block this lane stored in the forward pass: nothing but the
            // current layer's operand stays in registers across the chain
            constexpr bool RELU = S::act(l - 1) == TBNN_ACT_RELU;
            // issue the re-read three chunks before the end of the segment: short live range, latency still covered
            constexpr int KGB = C::KG(C::out(l)), KG_RELOAD = KGB > 3 ? KGB - 3 : 0;
            f32x4 arel[RELU ? 1 : MU];
            f32x4 acc[MU];
#pragma unroll
            for (int u = 0; u < MU; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            sfor<0, C::KG(C::out(l))>(SFOR_LAMBDA(kg) {
                constexpr int kg = SFOR_VAL(kg), c = C::cB(l) + kg;
                if constexpr (!RELU && kg == KG_RELOAD) {
#if WIDE_TBLOCK
                    const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(store + C::act_off(l, ntiles) + ((size_t)(tvalid ? stile : 0) * C::TA(l)) * 256,
                                                                                         0, C::TA(l) * 1024, WIDE_RSRC_FLAGS);
#pragma unroll
                    for (int u = 0; u < MU; ++u) arel[u] = tvalid ? tblk_load_d(rsr, tblk_wr_off(i16, g), u) : f32x4{0.f, 0.f, 0.f, 0.f};
#else
                    const float* p = store + C::act_off(l, ntiles) + ((size_t)(tvalid ? stile : 0) * C::TA(l)) * 256 + i16 * 16 + g * 4;
#pragma unroll
                    for (int u = 0; u < MU; ++u) arel[u] = tvalid ? *reinterpret_cast<const f32x4*>(p + u * 256) : f32x4{0.f, 0.f, 0.f, 0.f};
#endif
                }
#if WIDE_HANDSCHED
                wide_chunk<S, c, false>(base, Anx, acc, dz[kg], stg, ring, img, im, wave, lane);
                WSTAMP(9 + 2 * c);
#else
                wide_stage<S, c>(base, stg, ring, img, wave, lane);
                f32x4 Acur[C::MAXT];
                wide_load_A<S, c>(base, Acur, ring, lane);
#pragma unroll
                for (int s = 0; s < C::ksteps(C::out(l), kg); ++s)
#pragma unroll
                    for (int u = 0; u < MU; ++u) acc[u] = mfma16(Acur[u][s], dz[kg][s], acc[u]);
                WSTAMP(8 + 2 * c);
                WIDE_CHUNK_BARRIER();
                WSTAMP(9 + 2 * c);
#endif
            });
#pragma unroll
            for (int u = 0; u < MU; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (RELU) dz[u][r] = relu_mask_apply(T.relu[l - 1][(4 * u + r) / 32], (4 * u + r) % 32, acc[u][r]);
                    else dz[u][r] = actc_bwd_mul<S::act(l - 1)>(acc[u][r], arel[u][r]);
                }
        });

        WSTAMP(4);
        // ---- dW_0 += delta_0^T [x, 1]: contraction over the 16 rows (on the lanes): transpose through LDS
        {
            float Bop[C::NT0][4];
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int nt = 0; nt < C::NT0; ++nt) Bop[nt][s] = ximg[(4 * g + s) * C::PX + 16 * nt + i16];
#pragma unroll
            for (int b0 = 0; b0 < C::MT0; b0 += C::DZB) {
#pragma unroll
                for (int t = b0; t < b0 + C::DZB && t < C::MT0; ++t)
                    *reinterpret_cast<f32x4*>(scr + i16 * C::PZ + 16 * (t - b0) + 4 * g) = dz[t];
                float Aop[C::DZB][4];
#pragma unroll
                for (int t = b0; t < b0 + C::DZB && t < C::MT0; ++t)
#pragma unroll
                    for (int s = 0; s < 4; ++s) Aop[t - b0][s] = scr[(4 * g + s) * C::PZ + 16 * (t - b0) + i16];
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int t = b0; t < b0 + C::DZB && t < C::MT0; ++t)
#pragma unroll
                        for (int nt = 0; nt < C::NT0; ++nt)
#if WIDE_DW0_AGPR
                            mfma16_acc<C::DW0_FAR>(dW0[t * C::NT0 + nt], Aop[t - b0][s], Bop[nt][s]);
#else
                            dW0[t * C::NT0 + nt] = mfma16(Aop[t - b0][s], Bop[nt][s], dW0[t * C::NT0 + nt]);
#endif
            }
        }
        WSTAMP(5);
        base = (base + C::NCH) & (WIDE_RING - 1);
    }

    if constexpr (FWD) return;
#if WIDE_DW0_AGPR
    mfma_drain_acc(dW0);
#endif
    // ---- epilogue: compact slab [layer 0][last layer] of this workgroup
    const double wtot = wave_sum_lane0(stat);
    if (lane == 0) red[wave] = wtot;
    __syncthreads();                                   // ring is dead
    float* slab = slabA + (size_t)blockIdx.x * C::SA_FLOATS;
    {
        // dW_0: stage the 4 waves' copies [wave][tile][lane] (EP tiles per pass, whole LDS is free now), wave t%4 sums tile t
        constexpr int EP = C::LDS_FLOATS / (WIDE_WAVES * 256) < C::DW0_TILES ? C::LDS_FLOATS / (WIDE_WAVES * 256) : C::DW0_TILES;
        static_assert(EP >= 1, "no room to stage dW_0");
        f32x4* stgb = reinterpret_cast<f32x4*>(lds);
#pragma unroll
        for (int t0 = 0; t0 < C::DW0_TILES; t0 += EP) {
#pragma unroll
            for (int t = t0; t < t0 + EP && t < C::DW0_TILES; ++t) stgb[(wave * EP + (t - t0)) * 64 + lane] = dW0[t];
            __syncthreads();
#pragma unroll
            for (int t = t0; t < t0 + EP && t < C::DW0_TILES; ++t)
                if ((t & (WIDE_WAVES - 1)) == wave) {
                    const f32x4 c0 = stgb[(t - t0) * 64 + lane], c1 = stgb[(EP + t - t0) * 64 + lane],
                                c2 = stgb[(2 * EP + t - t0) * 64 + lane], c3 = stgb[(3 * EP + t - t0) * 64 + lane];
                    const int mt = t / C::NT0, nt = t % C::NT0;
                    const int col = 16 * nt + i16, row0 = 16 * mt + 4 * g;
                    if (col <= d_in) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = unit_of(C::out(0), row0 + r, false);
                            if (row >= 0)
                                slab[C::SA_L0 + (col < d_in ? row * d_in + col : d_in * C::out(0) + row)] = (c0[r] + c1[r]) + (c2[r] + c3[r]);
                        }
                    }
                }
            __syncthreads();
        }
    }
    if constexpr (C::VL) {
        // last layer: reduce the per-row partials over the 16 lanes of a lane group, then over the 4 waves
        constexpr int TP = C::TR(LL), inL = C::in(LL);
        float* lb = lds;                               // [wave][o][slot], then [wave][o] biases
        static_assert(WIDE_WAVES * d_out * (16 * TP + 1) <= C::LDS_FLOATS, "last-layer staging does not fit");
#pragma unroll
        for (int o = 0; o < d_out; ++o) {
#pragma unroll
            for (int t = 0; t < TP; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = row16_sum(accL[o][t][r]);
                    if (i16 == 0) lb[(wave * d_out + o) * (16 * TP + 1) + 16 * t + 4 * g + r] = v;
                }
            const float vb = row16_sum(accbL[o]);
            if (lane == 0) lb[(wave * d_out + o) * (16 * TP + 1) + 16 * TP] = vb;
        }
        __syncthreads();
        for (int e = tid; e < d_out * (inL + 1); e += WIDE_THREADS) {
            const int o = e / (inL + 1), u = e - o * (inL + 1);
            const int s = u < inL ? slot_of(inL, u) : 16 * TP;
            float v[WIDE_WAVES];
#pragma unroll
            for (int w = 0; w < WIDE_WAVES; ++w) v[w] = lb[(w * d_out + o) * (16 * TP + 1) + s];
            slab[C::SA_LL + (u < inL ? o * inL + u : inL * d_out + o)] = (v[0] + v[1]) + (v[2] + v[3]);
        }
    }
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < WIDE_WAVES; ++w) t += red[w];
        pstat[blockIdx.x] = t;
    }
}

// ---------------------------------------------------------------------------------------------
// k_dw_wide: dW_l / db_l of the middle layers from the stored blocks
// ---------------------------------------------------------------------------------------------
// Workgroup b of layer l's group (wg_lo[l] <= b < wg_lo[l+1]) owns row tiles [rt0, rt1) and the whole
// dW_l: wave w accumulates M tiles {w, w+4, ..} x all N tiles plus QP (m, u) pairs of the left-over
// M tiles.  A operand = delta block (k = data row 4s+g, m = out slot), B operand = a block: both are
// plain lane-linear reads of the [16 rows][16 slots] block (float offset 64 s + lane).
struct WideDwArgs {
    int wg_lo[TBNN_MAX_LAYERS + 1];      // first workgroup of middle layer l (index l-1); [NM] = total
};

#if WIDE_TBLOCK
// k_dw_wide on transposed blocks: every operand one 16-B buffer load straight into registers.  Loop order inside a row tile:
// N-tile-major -- for a-block u the QM x 4 MFMAs of this wave's M tiles (an accumulator is revisited every QM-th MFMA), after
// which B[u]'s registers are dead and take block u of the NEXT row tile (a whole row tile, ~170 MFMAs, ahead of its use); the
// few A operands (delta blocks of the wave's M tiles, left-over pairs) are double-buffered, the row loop unrolled by two so
// that the buffers swap by renaming.  Accumulators pinned to AccVGPRs (asm MFMAs: program order is issue order).
template <class S, int l>
__device__ __forceinline__ void dw_wide_layer(const float* __restrict__ store, long ntiles, long rt0, long rt1,
                                              float* __restrict__ slab, float* lds, int wave, int lane) {
    using C = WideCfg<S>;
    constexpr int TAl = C::TA(l), TZl = C::TZ(l), QM = C::QM(l), RM = C::RM(l), QP = C::QP(l);
    constexpr int NT = QM * TAl + QP, QMd = QM > 0 ? QM : 1, QPd = QP > 0 ? QP : 1;
    const float* abase = store + C::act_off(l, ntiles);
    const float* zbase = store + C::dz_off(l, ntiles);
    f32x4 acc[NT > 0 ? NT : 1];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = acc_zero();
    // left-over pairs of this wave: p = wave + 4q -> (m = 4 QM + p / TA, u = p % TA); invalid ones recompute pair 0
    int pm[QPd], pu[QPd];
    bool pv[QPd];
#pragma unroll
    for (int q = 0; q < QP; ++q) {
        const int p = wave + 4 * q;
        pv[q] = p < RM * TAl;
        const int pp = pv[q] ? p : 0;
        pm[q] = 4 * QM + pp / TAl; pu[q] = pp % TAl;
    }
    const int voff = tblk_rd_off(lane);
    auto rsa = [&](long rt) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(abase) + (size_t)rt * TAl * 256, 0, TAl * 1024, WIDE_RSRC_FLAGS);
    };
    auto rsz = [&](long rt) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(zbase) + (size_t)rt * TZl * 256, 0, TZl * 1024, WIDE_RSRC_FLAGS);
    };
    auto ld = [&](__amdgpu_buffer_rsrc_t rs, int blk) __attribute__((always_inline)) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, blk * 1024, 0));
    };
    if (rt0 >= rt1) {                                               // an empty row range still owes its slab: zeros
        constexpr int np = C::in(l) * C::out(l) + C::out(l);
        for (int e = wave * 64 + lane; e < np; e += WIDE_THREADS) slab[e] = 0.f;
        return;
    }
    if constexpr (QM == 0) {
        // A layer with NO whole M tile per wave -- the one-tile last layer of a network with 3 .. 16 outputs (round 6): QP <= 4 (m, u) pairs per
        // wave and row tile, i.e. 16 MFMAs behind two block loads each.  Walked like the layers below (one row tile ahead) the loop runs at the
        // load latency (10 -> 200 -> 200 -> 10 at 1e5 rows: k_dw_wide 267 us against 80 us for the same network with one output); here a wave
        // asks for the operands of WIDE_DW_G row tiles at once, two such groups in flight, and reads only the blocks of its own pairs.
        constexpr int G = WIDE_DW_G;
        f32x4 Ag0[G][QPd], Bg0[G][QPd], Ag1[G][QPd], Bg1[G][QPd];
        auto loadg = [&](long rt, f32x4 (&Ag)[G][QPd], f32x4 (&Bg)[G][QPd]) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const long r = rt + i < rt1 ? rt + i : rt1 - 1;        // (clamped: a group past the end is loaded and not used)
                const __amdgpu_buffer_rsrc_t rz = rsz(r), ra = rsa(r);
#pragma unroll
                for (int q = 0; q < QP; ++q) { Ag[i][q] = ld(rz, pm[q]); Bg[i][q] = ld(ra, pu[q]); }
            }
        };
        auto mmg = [&](long rt, const f32x4 (&Ag)[G][QPd], const f32x4 (&Bg)[G][QPd]) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < G; ++i)
                if (rt + i < rt1) {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int q = 0; q < QP; ++q) mfma16_acc<false>(acc[q], Ag[i][q][s], Bg[i][q][s]);
                }
        };
        loadg(rt0, Ag0, Bg0);
        for (long rt = rt0; rt < rt1; rt += 2 * G) {
            loadg(rt + G, Ag1, Bg1);
            WIDE_FENCE();
            mmg(rt, Ag0, Bg0);
            loadg(rt + 2 * G, Ag0, Bg0);
            WIDE_FENCE();
            mmg(rt + G, Ag1, Bg1);
        }
    } else {
    // operand registers: B[u] = a-block u (refreshed in place), A = delta blocks of this wave's M tiles {w, w+4, ..}, Z = the
    // RM left-over delta blocks (their (m, u) pairs are dealt over the waves: pair p = u + k TA belongs to wave p % 4, its
    // accumulator is number p / 4 of that wave's left-over set -- all compile-time once the wave is known, so the pair MFMAs
    // sit in wave-uniform branches inside the u-groups and read B[u] while it is still there).  The asm MFMAs must not get
    // operands that the VALU has just written (an operand parked in an AccVGPR comes back through v_accvgpr_read, and an MFMA
    // issued right behind it read the OLD register): the operand set is kept small enough to stay in ArchVGPRs.
    constexpr int RMd = RM > 0 ? RM : 1;
    f32x4 B[TAl], A0[QMd], A1[QMd], Z0[RMd], Z1[RMd];
    auto load_small = [&](long rt, f32x4 (&A)[QMd], f32x4 (&Z)[RMd]) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rz = rsz(rt);
#pragma unroll
        for (int j = 0; j < QM; ++j) A[j] = ld(rz, wave + 4 * j);
#pragma unroll
        for (int k = 0; k < RM; ++k) Z[k] = ld(rz, 4 * QM + k);
    };
    {
        const __amdgpu_buffer_rsrc_t ra = rsa(rt0);
#pragma unroll
        for (int u = 0; u < TAl; ++u) B[u] = ld(ra, u);
        load_small(rt0, A0, Z0);
    }
    // one row tile: operands of `rt` in (A, Z) and B; requests those of rt + 1 (clamped) into (An, Zn) and B
    auto body = [&](long rt, f32x4 (&A)[QMd], f32x4 (&Z)[RMd], f32x4 (&An)[QMd], f32x4 (&Zn)[RMd]) __attribute__((always_inline)) {
        const long rn = rt + 1 < rt1 ? rt + 1 : rt;
        load_small(rn, An, Zn);
        const __amdgpu_buffer_rsrc_t ran = rsa(rn);
        WIDE_FENCE();
        if constexpr (QM == 2) {
            // TWO M tiles per wave: the a-blocks are taken in PAIRS -- four accumulators in turn instead of two (an asm-form accumulator is
            // revisited back to back or after at least two other MFMAs in every family); an odd last block runs on the builtin form.
            // (Round 5: 32 -> 116 -> 187 -> 114 -> 1 and 15 -> 170 -> 114 -> 1 had wrong, unrepeatable dW tiles here.  The cause that was
            // FOUND in the disassembly is the operand side: with two waves per SIMD the register allocator parks a-blocks in AccVGPRs, and
            // the v_accvgpr_read that brings one back stood straight in front of the asm MFMA that reads it -- a VALU write and an MFMA read
            // with no wait state between them.  build.py / jit.py now check every unit for that pair and rebuild it with the wait states
            // inside the asm statement: tensorbnn_amd/hazard_lint.py.)
            sfor<0, (TAl + 1) / 2>(SFOR_LAMBDA(h) {
                constexpr int u = 2 * SFOR_VAL(h);
                constexpr bool two = u + 1 < TAl;
                constexpr int u1 = two ? u + 1 : u;
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int j = 0; j < QM; ++j) {
                        mfma16_acc<two>(acc[j * TAl + u], A[j][s], B[u][s]);
                        if constexpr (two) mfma16_acc<true>(acc[j * TAl + u1], A[j][s], B[u1][s]);
                    }
                sfor<0, RM>(SFOR_LAMBDA(k) {                          // left-over pairs on a-blocks u, u + 1
                    constexpr int pp = u + SFOR_VAL(k) * TAl;
                    if (wave == pp % 4) {
#pragma unroll
                        for (int s = 0; s < 4; ++s) mfma16_acc<false>(acc[QM * TAl + pp / 4], Z[SFOR_VAL(k)][s], B[u][s]);
                    }
                    if constexpr (two) {
                        constexpr int pq = u1 + SFOR_VAL(k) * TAl;
                        if (wave == pq % 4) {
#pragma unroll
                            for (int s = 0; s < 4; ++s) mfma16_acc<false>(acc[QM * TAl + pq / 4], Z[SFOR_VAL(k)][s], B[u1][s]);
                        }
                    }
                });
                WIDE_FENCE();
                B[u] = ld(ran, u);
                if constexpr (two) B[u1] = ld(ran, u1);
                WIDE_FENCE();
            });
        } else {
        sfor<0, TAl>(SFOR_LAMBDA(u) {
            constexpr int u = SFOR_VAL(u);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < QM; ++j) mfma16_acc<(QM > 2)>(acc[j * TAl + u], A[j][s], B[u][s]);
            sfor<0, RM>(SFOR_LAMBDA(k) {                              // left-over pairs on a-block u
                constexpr int pp = u + SFOR_VAL(k) * TAl;
                if (wave == pp % 4) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) mfma16_acc<false>(acc[QM * TAl + pp / 4], Z[SFOR_VAL(k)][s], B[u][s]);
                }
            });
            WIDE_FENCE();
            B[u] = ld(ran, u);
            WIDE_FENCE();
        });
        }
    };
    for (long rt = rt0; rt < rt1; rt += 2) {
        body(rt, A0, Z0, A1, Z1);
        if (rt + 1 < rt1) body(rt + 1, A1, Z1, A0, Z0);
        else break;
    }
    }
    mfma_drain_acc(acc);
    // write-out in theta order: D layout lane (n = lane & 15, g) reg j = dW[out 16m+4g+j][in 16u+n]
    const int nn = lane & 15, gg = lane >> 4;
    constexpr int inl = C::in(l), outl = C::out(l);
    auto put = [&](int m, int u, const f32x4& v) {
        const int col = unit_of(inl, 16 * u + nn, true);          // inl: the ones pseudo-unit (bias column)
        if (col >= 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = unit_of(outl, 16 * m + 4 * gg + j, false);
                if (row >= 0) slab[col < inl ? row * inl + col : inl * outl + row] = v[j];
            }
        }
    };
#pragma unroll
    for (int j = 0; j < QM; ++j)
#pragma unroll
        for (int u = 0; u < TAl; ++u) put(wave + 4 * j, u, acc[j * TAl + u]);
#pragma unroll
    for (int q = 0; q < QP; ++q)
        if (pv[q]) put(pm[q], pu[q], acc[QM * TAl + q]);
}
#else
template <class S, int l>
__device__ __forceinline__ void dw_wide_layer(const float* __restrict__ store, long ntiles, long rt0, long rt1,
                                              float* __restrict__ slab, float* lds, int wave, int lane) {
    using C = WideCfg<S>;
    constexpr int TAl = C::TA(l), TZl = C::TZ(l), QM = C::QM(l), RM = C::RM(l), QP = C::QP(l), SB = C::SB(l);
    constexpr int NT = QM * TAl + QP;
    constexpr int NG = C::cdiv(SB, WIDE_WAVES);
    const float* abase = store + C::act_off(l, ntiles);
    const float* zbase = store + C::dz_off(l, ntiles);
    f32x4 acc[NT > 0 ? NT : 1];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = acc_zero();
    // left-over pairs of this wave: p = wave + 4q -> (m = 4 QM + p / TA, u = p % TA); invalid ones recompute pair 0
    int pm[QP > 0 ? QP : 1], pu[QP > 0 ? QP : 1];
    bool pv[QP > 0 ? QP : 1];
#pragma unroll
    for (int q = 0; q < QP; ++q) {
        const int p = wave + 4 * q;
        pv[q] = p < RM * TAl;
        const int pp = pv[q] ? p : 0;
        pm[q] = 4 * QM + pp / TAl; pu[q] = pp % TAl;
    }
    // slot layout: [a blocks 0..TA-1][delta blocks 0..TZ-1], 256 floats each
    f32x4 stg[WIDE_DW_PD][NG];
    auto gload = [&](long rt, f32x4 (&dst)[NG]) {
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int b = wave + 4 * j;
            if (b < SB && rt < rt1) {
                const float* src = b < TAl ? abase + ((size_t)rt * TAl + b) * 256 : zbase + ((size_t)rt * TZl + (b - TAl)) * 256;
                dst[j] = *reinterpret_cast<const f32x4*>(src + lane * 4);
            } else dst[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto park = [&](int slot, const f32x4 (&src)[NG]) {
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int b = wave + 4 * j;
            if (b < SB) *reinterpret_cast<f32x4*>(lds + slot * C::DW_SLOT_FLOATS + b * 256 + lane * 4) = src[j];
        }
    };
    // prime: row tiles rt0, rt0+1 parked; rt0+2 .. rt0+1+PD in registers
    gload(rt0, stg[0]); park(0, stg[0]);
    gload(rt0 + 1, stg[0]); park(1, stg[0]);
#pragma unroll
    for (int k = 0; k < WIDE_DW_PD; ++k) gload(rt0 + 2 + k, stg[k]);
    __syncthreads();
    int it = 0;
    for (long rt = rt0; rt < rt1; ++rt, ++it) {
        park((it + 2) & (WIDE_RING - 1), stg[0]);
#pragma unroll
        for (int k = 0; k + 1 < WIDE_DW_PD; ++k)
#pragma unroll
            for (int j = 0; j < NG; ++j) stg[k][j] = stg[k + 1][j];
        gload(rt + 2 + WIDE_DW_PD, stg[WIDE_DW_PD - 1]);
        const float* sl = lds + (it & (WIDE_RING - 1)) * C::DW_SLOT_FLOATS + lane;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float B[TAl], A[QM > 0 ? QM : 1];
#pragma unroll
            for (int u = 0; u < TAl; ++u) B[u] = sl[u * 256 + 64 * s];
#pragma unroll
            for (int j = 0; j < QM; ++j) A[j] = sl[(TAl + wave + 4 * j) * 256 + 64 * s];
#pragma unroll
            for (int j = 0; j < QM; ++j)
#pragma unroll
                for (int u = 0; u < TAl; ++u) acc[j * TAl + u] = mfma16(A[j], B[u], acc[j * TAl + u]);
#pragma unroll
            for (int q = 0; q < QP; ++q) {
                const float a = sl[(TAl + pm[q]) * 256 + 64 * s], b = sl[pu[q] * 256 + 64 * s];
                acc[QM * TAl + q] = mfma16(a, b, acc[QM * TAl + q]);
            }
        }
        __syncthreads();
    }
    // write-out in theta order: D layout lane (n = lane & 15, g) reg j = dW[out 16m+4g+j][in 16u+n]
    const int nn = lane & 15, gg = lane >> 4;
    constexpr int inl = C::in(l), outl = C::out(l);
    auto put = [&](int m, int u, const f32x4& v) {
        const int col = unit_of(inl, 16 * u + nn, true);          // inl: the ones pseudo-unit (bias column)
        if (col >= 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = unit_of(outl, 16 * m + 4 * gg + j, false);
                if (row >= 0) slab[col < inl ? row * inl + col : inl * outl + row] = v[j];
            }
        }
    };
#pragma unroll
    for (int j = 0; j < QM; ++j)
#pragma unroll
        for (int u = 0; u < TAl; ++u) put(wave + 4 * j, u, acc[j * TAl + u]);
#pragma unroll
    for (int q = 0; q < QP; ++q)
        if (pv[q]) put(pm[q], pu[q], acc[QM * TAl + q]);
}

#endif  // WIDE_TBLOCK

template <class S>
__global__ __launch_bounds__(WIDE_THREADS, WideCfg<S>::DW_OCC) void k_dw_wide(
    WideDwArgs args, const float* __restrict__ store, long n, float* __restrict__ slabB)
{
    using C = WideCfg<S>;
#if WIDE_TBLOCK
    float* lds = nullptr;                                // operands come straight from L2 (dw_wide_layer)
#else
    static_assert(WIDE_RING * C::DW_SLOT_FLOATS * 4 <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) float lds[WIDE_RING * C::DW_SLOT_FLOATS];
#endif
    // wave index as a SCALAR (the compiler does not know threadIdx.x >> 6 is wave-uniform: every `b = wave + 4j < SB` below
    // would become an exec-mask branch with zero-filled else arms, and the block addresses vector arithmetic)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long ntiles = (n + 15) / 16;
    const int b = blockIdx.x;
    int l = 1;
#pragma unroll
    for (int m = 2; m <= C::NM; ++m) if (b >= args.wg_lo[m - 1]) l = m;
    const int nwg = args.wg_lo[l] - args.wg_lo[l - 1], bl = b - args.wg_lo[l - 1];
    const long per = (ntiles + nwg - 1) / nwg;
    const long rt0 = (long)bl * per, rt1 = rt0 + per < ntiles ? rt0 + per : ntiles;
    float* slab = slabB + (size_t)b * C::SB_FLOATS;     // every workgroup gets SB_FLOATS (only its layer's part is used)
    const long r0 = rt0 < ntiles ? rt0 : ntiles, r1 = rt1 > r0 ? rt1 : r0;
    sfor<1, C::NM + 1>(SFOR_LAMBDA(m) {
        constexpr int m = SFOR_VAL(m);
        if (l == m) dw_wide_layer<S, m>(store, ntiles, r0, r1, slab + C::slabB_off(m), lds, wave, lane);
    });
}

// sum the partial slabs in fixed order into one dense gradient row (pitch P) for k_update:
// 64 parameters x 4 slab groups per block, fixed-order combine (deterministic)
template <class S>
__global__ __launch_bounds__(256) void k_reduce_wide(WideDwArgs args, const float* __restrict__ slabA, int nA,
                                                      const float* __restrict__ slabB, float* __restrict__ out)
{
    using C = WideCfg<S>;
    __shared__ float part[4][64];
    const int x = threadIdx.x, y = threadIdx.y;
    const int j = blockIdx.x * 64 + x;
    float tot = 0.f;
    if (j < C::P()) {
        int l = 0;
#pragma unroll
        for (int m = 1; m < C::NL; ++m) if (j >= C::offW(m)) l = m;
        const int k = j - C::offW(l);
        const float* src; int cnt; size_t pitch;
        if (l == 0) { src = slabA + C::SA_L0 + k; cnt = nA; pitch = C::SA_FLOATS; }
        else if (C::VL && l == C::LL) { src = slabA + C::SA_LL + k; cnt = nA; pitch = C::SA_FLOATS; }
        else { src = slabB + (size_t)args.wg_lo[l - 1] * C::SB_FLOATS + C::slabB_off(l) + k; cnt = args.wg_lo[l] - args.wg_lo[l - 1]; pitch = C::SB_FLOATS; }
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int w = y;
        for (; w + 12 < cnt; w += 16) {
            s0 += src[(size_t)w * pitch]; s1 += src[(size_t)(w + 4) * pitch];
            s2 += src[(size_t)(w + 8) * pitch]; s3 += src[(size_t)(w + 12) * pitch];
        }
        for (; w < cnt; w += 4) s0 += src[(size_t)w * pitch];
        tot = (s0 + s1) + (s2 + s3);
    }
    part[y][x] = tot;
    __syncthreads();
    if (y == 0 && j < C::P()) out[j] = (part[0][x] + part[1][x]) + (part[2][x] + part[3][x]);
}

// host: flat parameter index -> offsets in the wide weight image
template <class S>
static void wide_image_map(int* map) {
    using C = WideCfg<S>;
    const int P = C::P();
    for (int l = 0; l < C::NL; ++l) {
        const int in = C::in(l), out = C::out(l), ow = C::offW(l);
        for (int i = 0; i < out; ++i) {
            const int ri = slot_of(out, i);
            for (int k = 0; k < in; ++k) {
                const int ck = slot_of(in, k);
                int m0, m1 = -1;
                if (l == 0) {
                    m0 = C::W0_OFF + (((ri / 16) * C::KG0 + ck / 16) * 64 + ((ck % 16) / 4) * 16 + ri % 16) * 4 + ck % 4;
                } else if (C::VL && l == C::LL) {
                    m0 = C::WL_OFF + i * C::WLP + ck;
                } else {
                    m0 = C::gran_off(C::cF(l) + ck / 16, ri / 16) + (((ck % 16) / 4) * 16 + ri % 16) * 4 + ck % 4;
                    m1 = C::gran_off(C::cB(l) + ri / 16, ck / 16) + (((ri % 16) / 4) * 16 + ck % 16) * 4 + ri % 4;
                }
                map[ow + i * in + k] = m0;
                map[P + ow + i * in + k] = m1;
            }
            map[ow + in * out + i] = (C::VL && l == C::LL) ? C::BL_OFF + i : C::boff(l) + ri;
            map[P + ow + in * out + i] = -1;
        }
    }
}

// host: workgroup plan and the three launches of one fused pass
template <class S>
static inline void wide_plan_t(long n, WidePlan& p) {
    using C = WideCfg<S>;
    const long ntiles = (n + 15) / 16, nblk = (ntiles + WIDE_WAVES - 1) / WIDE_WAVES;
    p.gridA = (int)std::min<long>(nblk, 256);
    // 256 (x DW_OCC) dW workgroups shared out over the middle layers in proportion to their MFMA count
    long tot = 0;
    for (int l = 1; l <= C::NM; ++l) tot += C::dw_cost(l);
    int budget = (int)std::min<long>(256 * C::DW_OCC, std::max<long>(ntiles, C::NM)), used = 0;
    p.wg_lo[0] = 0;
    for (int l = 1; l <= C::NM; ++l) {
        int w = l == C::NM ? budget - used : (int)std::max<long>(1, (budget * C::dw_cost(l)) / tot);
        w = std::max(1, w);
        used += w;
        p.wg_lo[l] = used;
    }
    p.gridB = used;
    p.store_floats = (size_t)C::store_floats(ntiles);
    p.slabA_floats = (size_t)p.gridA * C::SA_FLOATS;
    p.slabB_floats = (size_t)p.gridB * C::SB_FLOATS;
    p.img_floats = C::IMG_FLOATS;
}

template <class S>
static inline int wide_launch_t(const WidePlan& p, hipStream_t st, const NetDev& nd, const float* qimg, const float* eta, const float* X,
                    const float* Y, long n, float* store, float* slabA, float* slabB, double* pstat, float* out) {
    using C = WideCfg<S>;
    WideDwArgs a;
    for (int i = 0; i <= TBNN_MAX_LAYERS; ++i) a.wg_lo[i] = p.wg_lo[i];
    hipLaunchKernelGGL((k_chain_wide<S, false>), dim3(p.gridA), dim3(WIDE_THREADS), 0, st, nd, qimg, eta, X, Y, n, store, slabA, pstat,
                       (float*)nullptr);
    hipLaunchKernelGGL(k_dw_wide<S>, dim3(p.gridB), dim3(WIDE_THREADS), 0, st, a, (const float*)store, n, slabB);
    hipLaunchKernelGGL(k_reduce_wide<S>, dim3((C::P() + 63) / 64), dim3(64, 4), 0, st, a, (const float*)slabA, p.gridA,
                       (const float*)slabB, out);
    return 0;
}

// forward only: fout[d_out][n] = network(X) for the weights in qimg; false when this shape's ring cannot be primed
template <class S>
static inline bool wide_forward_ok() { return WideCfg<S>::RESIDENT || WideCfg<S>::NCHF >= 3; }
template <class S>
static inline int wide_forward_t(hipStream_t st, const NetDev& nd, const float* qimg, const float* X, long n, float* fout) {
    if constexpr (WideCfg<S>::RESIDENT || WideCfg<S>::NCHF >= 3) {
        const long ntiles = (n + 15) / 16, nblk = (ntiles + WIDE_WAVES - 1) / WIDE_WAVES;
        const int grid = (int)std::min<long>(nblk, 256);
        hipLaunchKernelGGL((k_chain_wide<S, true>), dim3(grid), dim3(WIDE_THREADS), 0, st, nd, qimg, (const float*)nullptr, X,
                           (const float*)nullptr, n, (float*)nullptr, (float*)nullptr, (double*)nullptr, fout);
        return 0;
    } else {
        return -1;
    }
}
