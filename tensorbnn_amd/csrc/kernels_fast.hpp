// Shape-specialised fused forward + likelihood + backward kernel for gfx950:
// f32 MFMA (v_mfma_f32_16x16x4_f32) for every contraction, weights resident in
// LDS, activations and ALL dW accumulators resident in registers.
//
// One wave owns 16-row tiles of the training matrix (two at a time, interleaved
// for ILP and shared weight-operand loads; a single one for the remainder) and
// needs no cross-wave synchronisation inside the row loop:
//   forward   Z_l[unit, row] = W_l . A_{l-1}      (A operand = W_l from LDS,
//             B operand = the previous layer's accumulator registers AS THEY
//             STAND: the MFMA's k index is mapped to unit 16kt+4g+s so that
//             the C/D layout of layer l-1 *is* the B layout of layer l)
//   backward  D_{l-1} = (W_l^T . D_l) * act'      (same trick: D_l registers
//             feed the next MFMA directly, A operand = W_l read transposed)
//   dW_l      += D_l . A_{l-1}^T                  (contraction over the data
//             rows, which sit on the lanes of the C/D layout => one 16x16
//             transpose per operand through a per-wave LDS image; the bias
//             gradient rides along as a constant-1 column of A_{l-1})
// The per-wave dW accumulators are combined across the 4 waves of the
// workgroup in LDS (fixed order, deterministic) and written to the
// workgroup's slab; k_update reduces the slabs.
//
// Reference math: layer.py:278 (W@a+b), activationFunctions.py:36/49/62,
// likelihood.py:88-94,226-236, BNN_functions.py:23-32; reverse mode SURVEY A12.
#pragma once
#include "common.hpp"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define FAST_WAVES 4
#define FAST_THREADS (FAST_WAVES * 64)

// HACT: activation after every hidden layer, LACT: after the last layer,
// BERN: Bernoulli likelihood (else the Gaussian family) -- all compile-time so
// the tile body is one straight-line block the scheduler can interleave.
// Round 6: network.add takes any sequence of layers and activations (tensorBNN/network.py:173-191) -- a stack whose hidden layers do NOT all
// carry the same activation has HACT = TBNN_ACT_PACKED | sum_l act_l << 3 l (hidden layer l = 0 .. NL - 2; at most 9 of them: fused_ops.hpp,
// jit.shape_of); the kernels ask act(l) per layer and never HACT itself.
template <int HACT_, int LACT_, bool BERN_, int... Ds>
struct Shape {
    static constexpr int NL = sizeof...(Ds) - 1;
    static constexpr int D[sizeof...(Ds)] = {Ds...};
    static constexpr int HCODE = HACT_, LACT = LACT_;
    static constexpr bool BERN = BERN_;
    static_assert((HACT_ & TBNN_ACT_PACKED) == 0 || NL - 1 <= 9, "a packed activation code holds 9 hidden layers");
    static constexpr int act(int l) { return l == NL - 1 ? LACT_ : ((HACT_ & TBNN_ACT_PACKED) ? (HACT_ >> (3 * l)) & 7 : HACT_); }
};

#ifndef TBNN_FAST_ACT
#define TBNN_FAST_ACT 1
#endif
// DIAGNOSTIC builds only (TBNN_BUILD_TAG=skel.. TBNN_EXTRA_FLAGS=-DTBNN_SKEL=n, tools/experiments/skeleton.sh; wrong results, right
// instruction streams -- never the product): what the one-wave-per-SIMD design of k_fwd_bwd_fast3 can reach.
//   bit 0 (1): no row tiles at all -- prologue + epilogue only: the launch's fixed cost
//   bit 1 (2): the tile body without the VALU work that is not an MFMA operand's address or value: activations and their
//              derivatives pass their argument through, the fringe rows of dW (packed FMAs) are not accumulated
//   bit 2 (4): the forward / delta chains' weight operands are not read from LDS (as if they lived in registers)
//   bit 3 (8): the fourth N tile of dW_1 / dW_2 (3 useful columns of 16: units 48, 49 and the bias) is not computed
#ifndef TBNN_SKEL
#define TBNN_SKEL 0
#endif
template <int ACT>
__device__ __forceinline__ float actc_fwd(float z) {
    if constexpr ((TBNN_SKEL & 2) != 0) return z;
    // relu as ONE integer max on the bit pattern (negative floats are negative ints; no NaN canonicalisation op)
    else if constexpr (ACT == TBNN_ACT_RELU) return __int_as_float(max(__float_as_int(z), 0));
#if TBNN_FAST_ACT
    // hardware exp2 / reciprocal (about 2 ulp) instead of the library tanhf / expf + IEEE division: in the fused kernels
    // every VALU instruction costs MFMA time
    else if constexpr (ACT == TBNN_ACT_TANH) {
        // 1 - 2/(1 + e^2z) cancels for small |z| (absolute error ~1e-7 whatever z is): below 0.3 the odd series
        // z (1 - z^2/3 + 2 z^4/15 - 17 z^6/315 + 62 z^8/2835), whose next term is < 6e-8 relative there
        const float t = z * z;
        const float small = z * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 62.f / 2835.f, -17.f / 315.f), 2.f / 15.f), -1.f / 3.f), 1.f);
        const float big = 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * z));
        return fabsf(z) < 0.3f ? small : big;
    }
    else if constexpr (ACT == TBNN_ACT_SIGMOID) return __builtin_amdgcn_rcpf(1.f + __expf(-z));
#else
    else if constexpr (ACT == TBNN_ACT_TANH) return tanhf(z);
    else if constexpr (ACT == TBNN_ACT_SIGMOID) return 1.f / (1.f + expf(-z));
#endif
    else if constexpr (ACT == TBNN_ACT_EXP) return expf(z);
    else if constexpr (ACT == TBNN_ACT_ELU) return z > 0.f ? z : expm1f(z);
    else return z;
}
template <int ACT>
__device__ __forceinline__ float actc_bwd(float a) {
    if constexpr (ACT == TBNN_ACT_RELU) return a > 0.f ? 1.f : 0.f;
    else if constexpr (ACT == TBNN_ACT_TANH) return 1.f - a * a;
    else if constexpr (ACT == TBNN_ACT_SIGMOID) return a * (1.f - a);
    else if constexpr (ACT == TBNN_ACT_EXP) return a;
    else if constexpr (ACT == TBNN_ACT_ELU) return a > 0.f ? 1.f : a + 1.f;
    else return 1.f;
}

// Unit <-> padded-slot map of a layer's unit dimension.  Full 16-unit groups are identity; the last,
// partial group spreads its units over the lane groups first (unit 16T+e -> slot 16T + 4*(e%4) + e/4),
// so that group needs only ceil(rem/4) MFMA k-steps (50 units: 13 steps instead of 14).  The dW bias
// column ("ones" pseudo-unit U) takes the next free slot.
__host__ __device__ constexpr int slot_of(int U, int u) {
    const int T = U / 16;
    if (u < 16 * T) return u;
    const int e = u - 16 * T;
    return 16 * T + 4 * (e % 4) + e / 4;
}
// -1: padding; U: the ones pseudo-unit (only when with_ones)
__host__ __device__ constexpr int unit_of(int U, int slot, bool with_ones) {
    const int T = U / 16, rem = U % 16;
    if (slot < 16 * T) return slot;
    const int sl = slot - 16 * T;
    if (sl >= 16) return -1;
    const int e = 4 * (sl % 4) + sl / 4;
    if (e < rem) return 16 * T + e;
    return (with_ones && e == rem) ? U : -1;
}

// slot of the ones pseudo-unit (bias column of dW) behind U real units
__host__ __device__ constexpr int ones_slot(int U) { return 16 * (U / 16) + 4 * ((U % 16) % 4) + (U % 16) / 4; }

// acc * act'(a) with act' expressed through the activation output a (relu: a select, no multiply)
template <int ACT>
__device__ __forceinline__ float actc_bwd_mul(float acc, float a) {
    if constexpr ((TBNN_SKEL & 2) != 0) return acc;
    else if constexpr (ACT == TBNN_ACT_RELU) return __float_as_int(a) > 0 ? acc : 0.f;
    else return acc * actc_bwd<ACT>(a);
}

// diagnostic: padding of the weight-image rows (TBNN_WPAD) and of fast3's transposed images (TBNN_PPAD), in floats; 4 / 4 in
// the product (8 makes the b128 operand reads bank-conflict-free on gfx950 but does not fit configs[1]'s LDS)
#ifndef TBNN_WPAD
#define TBNN_WPAD 4
#endif
#ifndef TBNN_PPAD
#define TBNN_PPAD 4
#endif
template <class S>
struct FastCfg {
    static constexpr int NL = S::NL;
    static constexpr int in(int l) { return S::D[l]; }
    static constexpr int out(int l) { return S::D[l + 1]; }
    static constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
    static constexpr int MT(int l) { return cdiv(out(l), 16); }          // M tiles of layer l's output
    static constexpr int NT(int l) { return cdiv(in(l) + 1, 16); }       // N tiles of dW_l (+1: ones column -> db)
    static constexpr int KG(int l) { return cdiv(in(l), 16); }           // 16-unit k groups of layer l's input
    static constexpr int LDW(int l) { return 16 * KG(l) + TBNN_WPAD; }   // pitch of the W_l image (== 4 mod 8)
    static constexpr int maxMT() { int m = 0; for (int l = 0; l < NL; ++l) m = MT(l) > m ? MT(l) : m; return m; }
    static constexpr int PA(int l) { return 16 * NT(l) + 4; }            // pitch of the image of a_l = input of layer l (== 4 mod 8)
    static constexpr int PD = 16 * maxMT() + 4;                          // pitch of the delta image
    // number of valid k-steps s in group kt of a K dimension of size K (unit = 16kt+4g+s)
    static constexpr int ksteps(int K, int kt) { int rem = K - 16 * kt; return rem >= 16 ? 4 : (rem <= 0 ? 0 : (rem + 3) / 4); }
    // LDS layout (floats): [W images][bias images][W^T images][per wave: images of a_0..a_{NL-1}, delta image]
    static constexpr int woff(int l) { int o = 0; for (int m = 0; m < l; ++m) o += 16 * MT(m) * LDW(m); return o; }
    static constexpr int W_FLOATS = woff(NL);
    static constexpr int boff(int l) { int o = W_FLOATS; for (int m = 0; m < l; ++m) o += 16 * MT(m); return o; }
    static constexpr int WB_FLOATS = boff(NL);                           // == the global padded image (k_update writes it)
    // transposed weight images W_l^T [in-unit][out-unit] for the delta chain (l >= 1), built in the prologue
    static constexpr int LDT(int l) { return 16 * MT(l) + TBNN_WPAD; }
    // last layer with <= 2 outputs runs on the VALU (16 FMAs per output instead of padded MFMA tiles)
    static constexpr bool VL = NL >= 2 && out(NL - 1) <= 2;
    static constexpr int NLM = VL ? NL - 1 : NL;                          // layers on the MFMA path
    static constexpr int toff(int l) { int o = WB_FLOATS; for (int m = 1; m < l && m < NLM; ++m) o += 16 * KG(m) * LDT(m); return o; }
    static constexpr int STATIC_FLOATS = toff(NL);
    static constexpr int aoff(int l) { int o = 0; for (int m = 0; m < l && m < NLM; ++m) o += 16 * PA(m); return o; }   // per wave
    static constexpr int doff = aoff(NL);
    static constexpr int WAVE_FLOATS = doff + 16 * PD;
    static constexpr int P() { int p = 0; for (int l = 0; l < NL; ++l) p += in(l) * out(l) + out(l); return p; }
    static constexpr int offW(int l) { int p = 0; for (int m = 0; m < l; ++m) p += in(m) * out(m) + out(m); return p; }
    static constexpr int dwoff(int l) { int o = 0; for (int m = 0; m < l && m < NLM; ++m) o += MT(m) * NT(m); return o; }
    static constexpr int DW_TILES = dwoff(NL);
    // epilogue staging: tiles per pass with all 4 waves' copies resident
    static constexpr int MIN_LDS = STATIC_FLOATS + FAST_WAVES * WAVE_FLOATS;
    // one staging pass when all dW tiles of the 4 waves fit in 156 KB of LDS, else as many tiles per pass as fit
    static constexpr int EP_TILES_WANT = DW_TILES * FAST_WAVES * 256 <= 39936 ? DW_TILES : (DW_TILES < 16 ? DW_TILES : 16);
    static constexpr int LDS_FLOATS = MIN_LDS > EP_TILES_WANT * FAST_WAVES * 256 ? MIN_LDS : EP_TILES_WANT * FAST_WAVES * 256;
    static constexpr int EP_TILES = LDS_FLOATS / (FAST_WAVES * 256) < DW_TILES ? LDS_FLOATS / (FAST_WAVES * 256) : DW_TILES;
    static constexpr int aroff(int l) { int o = 0; for (int m = 0; m < l; ++m) o += MT(m); return o; }   // act register tiles
    static constexpr int ACT_TILES = aroff(NL);
    static constexpr int KS0 = cdiv(in(0), 4);
    static constexpr int MTL = MT(NL - 1);
};

// diagnostic build only (-DTBNN_TILE_STAMPS): shader-clock stamps at the phase boundaries of a tile step
// (-DTBNN_TILE_STAMPS=2: only the kernel's outer phases, TSTAMPO -- the fine stamps serialise what they bracket)
#ifdef TBNN_TILE_STAMPS
#define TSTAMPO(k) do { __builtin_amdgcn_sched_barrier(0); if (blockIdx.x == 0 && threadIdx.x == 0) g_tile_stamps[k] = clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define TSTAMPO(k) do { } while (0)
#endif
#if defined(TBNN_TILE_STAMPS) && (TBNN_TILE_STAMPS + 0) != 2
#define TSTAMP(k) TSTAMPO(k)
#else
#define TSTAMP(k) do { } while (0)
#endif

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// dW accumulation with the accumulator pinned to AccVGPRs.  With -mllvm -amdgpu-mfma-vgpr-form (build.py, jit.py) the
// compiler's own MFMAs -- the forward / delta chain, whose results the VALU consumes -- write ArchVGPRs directly (no
// v_accvgpr_read per result register; on gfx950 every VALU instruction costs ~9 cycles of tile time because the f32
// MFMA and the VALU share the issue slot, tools/ubench/coexec.hip), and the dW tiles, which only MFMAs touch inside
// the row loop, must then not compete for the 256 ArchVGPRs.  No software wait states are inserted around inline asm:
// FAR says the same accumulator is revisited only after at least TWO other MFMAs; otherwise the builtin is used.  (Round 5: with
// ONE other 16x16x4 MFMA in between -- two accumulator tiles taken in turn -- the asm form read a stale accumulator: 8.5e-2 relative
// error in dW of a 7 -> 17 -> 33 -> 2 network on the narrow family's hand-threaded dW block; a 4x4x1 form accumulating into its predecessor's
// result likewise.  The wide family's k_dw_wide takes its a-blocks in PAIRS when a wave owns two M tiles -- four accumulators in turn, kernels_wide.hpp: QM == 2, DW0_FAR.)  Operands come from LDS loads; the kernels drain the pipe
// (mfma_drain) before the epilogue reads the accumulators.  TBNN_ACC_AGPR=0: builtin everywhere.
#ifndef TBNN_ACC_AGPR
#define TBNN_ACC_AGPR 1
#endif
// TBNN_ASM_MFMA_NOP=1: every asm MFMA carries two wait states of its own.  The hardware wants them between a VALU write of a VGPR and an MFMA
// that reads it as SrcA / SrcB / SrcC; the compiler keeps them for its own MFMAs and inserts nothing around inline asm.  An operand the register
// allocator parked in an AccVGPR comes back through v_accvgpr_read (a VALU write), possibly right in front of the asm MFMA (round 5: wrong,
// unrepeatable dW tiles of k_dw_wide).
// Round 6: ON by default -- the rule of the container's HIP guide (section 5.7: a pair with one side inside an asm string needs its s_nop INSIDE
// the string).  Measured against the build without them (bench.py, two alternating runs each): configs[1] -0.1 .. -0.3 %, configs[3] -0.35 %,
// configs[4] 0.0 %: the two wait states hide under the previous MFMA's passes.  The check in the compile (hazard_lint through checked_compile)
// stays as the belt for every other pair.
#ifndef TBNN_ASM_MFMA_NOP
#define TBNN_ASM_MFMA_NOP 1
#endif
#if TBNN_ASM_MFMA_NOP
#define TBNN_MFMA_PRE "s_nop 1\n\t"
#else
#define TBNN_MFMA_PRE ""
#endif
template <bool FAR>
__device__ __forceinline__ void mfma16_acc(f32x4& c, float a, float b) {
#if TBNN_ACC_AGPR
    if constexpr (FAR) asm volatile(TBNN_MFMA_PRE "v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else c = mfma16(a, b, c);
#else
    c = mfma16(a, b, c);
#endif
}
template <bool FAR>
__device__ __forceinline__ void mfma4_acc(f32x4& c, float a, float b) {        // the 16-block 4x4x1 form, accumulator pinned like mfma16_acc
#if TBNN_ACC_AGPR
    if constexpr (FAR) asm volatile(TBNN_MFMA_PRE "v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
#else
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
#endif
}
// An accumulator tile born in AccVGPRs.  `acc = f32x4{0, 0, 0, 0}` is a v_mov into ArchVGPRs as far as the register allocator is concerned: the
// loop-carried value of a "+a" accumulator then has TWO register classes (VGPR from the initialisation, AGPR from the asm MFMAs), the copies
// between them cannot be coalesced, and the allocator keeps the tile in ArchVGPRs across the loop's back edge -- k_dw_wide at configs[3] moved
// 36 of its 43 tiles AGPR -> VGPR at the top of every row-tile pair and VGPR -> AGPR in front of each block (300 VALU moves per 430 MFMAs, on
// a machine where the f32 MFMA and the VALU share the issue slot: round 6, found in the disassembly after PMC showed one non-MFMA VALU
// instruction per MFMA).  Written through asm "=a" outputs the initial value is of the accumulators' own class.
__device__ __forceinline__ f32x4 acc_zero() {
#if TBNN_ACC_AGPR
    float z0, z1, z2, z3;
    asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(z0));
    asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(z1));
    asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(z2));
    asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(z3));
    return f32x4{z0, z1, z2, z3};
#else
    return f32x4{0.f, 0.f, 0.f, 0.f};
#endif
}
__device__ __forceinline__ void mfma_drain() {
#if TBNN_ACC_AGPR
    asm volatile("s_nop 15\n\ts_nop 15");
#endif
}
// ... and every later read of the accumulators ordered BEHIND the drain: volatile asm statements keep their order, and an
// empty one that "modifies" acc[t] makes each later use of acc[t] depend on it.  (mfma_drain() alone orders nothing that has
// no side effect: in k_dw_wide the compiler placed a v_accvgpr_read of the last accumulator straight behind the MFMA that
// wrote it, ahead of the drain -- garbage in the last tiles.)
template <int N>
__device__ __forceinline__ void mfma_drain_acc(f32x4 (&acc)[N]) {
    mfma_drain();
#if TBNN_ACC_AGPR
#pragma unroll
    for (int t = 0; t < N; ++t) asm volatile("" : "+a"(acc[t]));
#endif
}

// ---- instruction-level helpers shared by the narrow (kernels_fast3.hpp) and wide (kernels_wide.hpp) kernels ----
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Packed f32 FMA as an opaque instruction.  The compiler's pre-emit peephole splits a v_pk_fma_f32 that follows an
// MFMA into two v_fma_f32 on the assumption that they run in the MFMA's shadow; on gfx950 an f32 MFMA and the f32
// VALU share the issue slot (tools/ubench/coexec.hip: +8.6 cycles per VALU instruction either way), so the split
// doubles the cost.  Operands of these helpers are never raw MFMA results (no software wait states are inserted
// for inline asm): callers pass activation outputs, LDS loads or VALU results only.
__device__ __forceinline__ f32x2 pkfma(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// a * (b[H], b[H]) + c
template <int H>
__device__ __forceinline__ f32x2 pkfma_bc(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 d;
    if constexpr (H == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// Relu derivative of a pair of activation outputs as {0, 1} floats in ONE packed instruction: clamp(a * FLT_MAX) to
// [0, 1].  a >= 0 always; every normal a > 0 gives 1, a == 0 gives 0 (a positive denormal below 2.9e-39 would give a
// fraction: a pre-activation in that interval does not occur in fp32 arithmetic of this size).  The mask is applied
// with v_mul_legacy_f32 (0 * x = 0 for every x, inf and NaN included), so a masked-out element is an exact zero
// like the select it replaces: 1.5 VALU instructions per element instead of v_cmp + v_cndmask.
#ifndef TBNN_F3_RELU_PK
#define TBNN_F3_RELU_PK 1
#endif
__device__ __forceinline__ f32x2 relu_step2(f32x2 a, f32x2 big) {
    f32x2 d;
    asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(big));
    return d;
}
#ifndef TBNN_F3_RELU_PKMUL
#define TBNN_F3_RELU_PKMUL 1
#endif
__device__ __forceinline__ f32x2 pkmul2(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ float mul_legacy(float s, float x) {
    float d;
    asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(d) : "v"(s), "v"(x));
    return d;
}
// The opaque instructions above read MFMA results, and no software wait states are inserted for inline asm: settle()
// stands between the MFMAs that produced `acc` and the first opaque reader (11 wait states cover the 8-pass
// v_mfma_f32_16x16x4_f32 -> VALU read requirement of 10).
template <int N>
__device__ __forceinline__ void mfma_settle(f32x4 (&acc)[N]) {
    static_assert(N >= 1 && N <= 8, "one asm statement ties up to 8 tiles");
    if constexpr (N == 1) asm volatile("s_nop 10" : "+v"(acc[0]));
    if constexpr (N == 2) asm volatile("s_nop 10" : "+v"(acc[0]), "+v"(acc[1]));
    if constexpr (N == 3) asm volatile("s_nop 10" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]));
    if constexpr (N == 4) asm volatile("s_nop 10" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
    if constexpr (N == 5) asm volatile("s_nop 10" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]));
    if constexpr (N == 6) asm volatile("s_nop 10" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]));
    if constexpr (N == 7) asm volatile("s_nop 10" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]));
    if constexpr (N == 8) asm volatile("s_nop 10" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]));
}
// N consecutive accumulator tiles settled (N up to 8 per statement)
template <int N>
__device__ __forceinline__ void coop_settle(f32x4* acc) {
    if constexpr (N <= 8) mfma_settle(*reinterpret_cast<f32x4 (*)[N]>(acc));
    else { mfma_settle(*reinterpret_cast<f32x4 (*)[8]>(acc)); coop_settle<N - 8>(acc + 8); }
}
// acc * act'(a) for the 4 registers of a tile; SETTLED: the caller has called mfma_settle on acc (or acc is a VALU result)
template <int ACT, bool SETTLED>
__device__ __forceinline__ f32x4 actc_bwd_mul4(f32x4 acc, f32x4 a) {
    f32x4 r;
    if constexpr ((TBNN_SKEL & 2) != 0) return acc;
    else if constexpr (ACT == TBNN_ACT_RELU && TBNN_F3_RELU_PK && SETTLED) {
        const f32x2 big = {3.402823466e38f, 3.402823466e38f};
        const f32x2 s01 = relu_step2(f32x2{a[0], a[1]}, big), s23 = relu_step2(f32x2{a[2], a[3]}, big);
#if TBNN_F3_RELU_PKMUL
        // the mask applied by a PACKED multiply too: one instruction per register pair instead of one v_mul_legacy per register.
        // (0 * inf = NaN here where the legacy multiply gives 0: a non-finite delta means a diverged trajectory, whose energy is
        // non-finite either way -> rejected; the oracle's `delta * (a > 0)` makes the same NaN)
        const f32x2 r01 = pkmul2(s01, f32x2{acc[0], acc[1]}), r23 = pkmul2(s23, f32x2{acc[2], acc[3]});
        r[0] = r01[0]; r[1] = r01[1]; r[2] = r23[0]; r[3] = r23[1];
#else
        r[0] = mul_legacy(s01[0], acc[0]); r[1] = mul_legacy(s01[1], acc[1]);
        r[2] = mul_legacy(s23[0], acc[2]); r[3] = mul_legacy(s23[1], acc[3]);
#endif
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = actc_bwd_mul<ACT>(acc[i], a[i]);
    }
    return r;
}

// a * (b[H], b[H])
template <int H>
__device__ __forceinline__ f32x2 pkmul_bc(f32x2 a, f32x2 b) {
    f32x2 d;
    if constexpr (H == 0) asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
    else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}
// sum over the 4 lane groups, broadcast to every lane (row i16); C is the inline constant 0 (a bias travelling in C
// would cost four v_mov to splat it, one v_add afterwards is cheaper)
// Only register 0 of the result is used.  The other three are dead the moment the MFMA is issued as far as the register allocator knows -- and
// the MFMA writes them eight passes LATER: an inline-asm VALU instruction (relu_step2, pkmul2 ..: invisible to the compiler's hazard recognizer)
// allocated to one of them would have its result overwritten when the MFMA lands (round 6, hazard_lint rule R2c: `v_pk_mul_f32 v[40:41] .. clamp`
// one instruction behind `v_mfma_f32_16x16x4_f32 v[38:41]` in the configs[1] kernel; every reader happened to come before the MFMA landed).
// `then` is the consumer of the sum, compiler-generated VALU code (which waits for the result by itself); the empty asm behind it keeps the
// three unused registers reserved until then.
__device__ __forceinline__ float gsum(float p);
// Round 6: the lane-group sum of the fringe units runs on the row-swap instructions (gsum: 4 VALU instructions) by default.  The MFMA form
// (ones x p, rounds 4-5) costs the matrix pipe 8 passes per sum and makes its consumer wait 10 wait states for register 0 of the result
// (the configs[1] tile loop carried ~85 wait states of s_nop behind its 11 lane-sum MFMAs); measured on one box, alternating runs of
// bench.py --workload c2: 20,749 / 20,723 leapfrog steps/s with the MFMA form, 20,967 / 20,959 with the swaps (+1.1 %; the fused pass by
// hipEvent 45.45 -> 44.98 us).  The two forms add the four lane groups in different orders (fp32 rounding: not bit-equal).
#ifndef TBNN_GSUM_PERMLANE
#define TBNN_GSUM_PERMLANE 1
#endif
template <class F>
__device__ __forceinline__ float gsum_mfma(float p, F then) {
    if constexpr (TBNN_GSUM_PERMLANE) return then(gsum(p));      // A/B: the lane-group sum on the row-swap instructions (4 VALU, no matrix pipe time, no 10-wait-state read)
    const f32x4 r = mfma16(1.f, p, f32x4{0.f, 0.f, 0.f, 0.f});
    float z = then(r[0]);
    asm("" : "+v"(z) : "v"(r[1]), "v"(r[2]), "v"(r[3]));
    return z;
}


// sum over the 4 lane groups (same lane&15) with the gfx950 row-swap instructions: VALU only, no
// LDS round trip (ds_bpermute would put ~2 x 100 cycles of latency on the layer chain)
__device__ __forceinline__ float gsum(float p) {
    const unsigned a = __float_as_uint(p);
    const auto r = __builtin_amdgcn_permlane32_swap(a, a, false, false);   // lanes l and l^32
    const float s = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    const unsigned b = __float_as_uint(s);
    const auto q = __builtin_amdgcn_permlane16_swap(b, b, false, false);   // rows r and r^1
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}


// the NS (1..4) k-step operands of one k-group: 4, 8 or 16 bytes from LDS
// (ns is a constant after unrolling: the branches fold)
__device__ __forceinline__ f32x4 load_ks(const float* p, int ns) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#if (TBNN_SKEL & 4)
    asm volatile("" : "=v"(v));            // diagnostic: the chain's weight operands cost no LDS read (whatever the registers hold)
    return v;
#endif
    if (ns >= 3) v = *reinterpret_cast<const f32x4*>(p);
    else if (ns == 2) { const float2 t = *reinterpret_cast<const float2*>(p); v[0] = t.x; v[1] = t.y; }
    else v[0] = p[0];
    return v;
}

// per-tile register state (outputs of every layer + the layer-0 operand)
template <class S>
struct TileRegs {
    using C = FastCfg<S>;
    f32x4 a[C::ACT_TILES];
    float x0[C::KS0];
};

#ifdef TBNN_NOFENCE
#define SCHED_FENCE() do {} while (0)
#else
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

// ---- forward: layer l.  Its first k-group of A operands and its bias tiles arrive preloaded
// (An, Bn); before its last MFMAs it preloads those of layer l+1.  The transposed image of its
// output (input of layer l+1, ones column at unit in(l+1)) is written right after the activation,
// i.e. under the next layer's MFMAs and far from the backward pass that reads it.
template <class S, int l>
struct FwdLayer {
    using C = FastCfg<S>;
    static __device__ __forceinline__ void preload(f32x4 (&An)[C::MT(l)], f32x4 (&Bn)[C::MT(l)], const float* __restrict__ lds,
                                                    int i16, int g) {
        constexpr int MT = C::MT(l);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) Bn[mt] = *reinterpret_cast<const f32x4*>(lds + C::boff(l) + 16 * mt + 4 * g);
        if constexpr (l == 0) {
            // natural k mapping (unit 4t+g): one b32 per (mt, t); KS0 <= 4 steps packed into An[mt]
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

    static __device__ __forceinline__ void run(TileRegs<S>& T, const float* __restrict__ lds, float* wl, int i16, int g,
                                                const f32x4 (&A0)[C::MT(l)], const f32x4 (&B0)[C::MT(l)]) {
        constexpr int MT = C::MT(l);
        constexpr int MTN = C::MT(l + 1 < C::NLM ? l + 1 : l);
        f32x4 acc[MT];
        f32x4 Anext[MTN], Bnext[MTN];          // next layer's preloads
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = B0[mt];
        if constexpr (l == 0) {
            if constexpr (l + 1 < C::NLM) FwdLayer<S, l + 1>::preload(Anext, Bnext, lds, i16, g);
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
                    if constexpr (l + 1 < C::NLM) FwdLayer<S, l + 1>::preload(Anext, Bnext, lds, i16, g);
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
        if constexpr (l + 1 < C::NLM) {
            constexpr int u1 = C::in(l + 1);
            float* aimg = wl + C::aoff(l + 1);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x4 v = T.a[C::aroff(l) + mt];
                if constexpr (u1 % 16 != 0) {
                    constexpr int osl = ones_slot(u1);
                    if (mt == osl / 16 && g == (osl % 16) / 4) v[osl % 4] = 1.f;
                }
                *reinterpret_cast<f32x4*>(aimg + i16 * C::PA(l + 1) + 16 * mt + 4 * g) = v;
            }
            TSTAMP(1 + l);
            FwdLayer<S, l + 1>::run(T, lds, wl, i16, g, Anext, Bnext);
        } else { TSTAMP(1 + l); }
    }
};

// ---- backward, as an explicit software pipeline (scheduling fences keep the order):
//   W(D_l) R(op_l) | dW_{l+1} MFMAs | dA_l MFMAs -> delta_{l-1} | recurse(l-1)
// so every LDS write->read round trip of layer l sits under the 64 dW MFMAs of layer l+1.
template <class S, int l>
struct BwdOps {
    using C = FastCfg<S>;
    // delta_l image [row][unit] -> A operands of dW_l; B operands from the image of a_{l-1} (k = data row 4g+s)
    static __device__ __forceinline__ void issue(const f32x4 (&dz)[C::MT(l)], float* wl, int i16, int g,
                                                  float (&Aop)[C::MT(l)][4], float (&Bop)[C::NT(l)][4]) {
        constexpr int MT = C::MT(l), NT = C::NT(l);
        float* dimg = wl + C::doff;
        const float* aimg = wl + C::aoff(l);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) *reinterpret_cast<f32x4*>(dimg + i16 * C::PD + 16 * mt + 4 * g) = dz[mt];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) Bop[nt][s] = aimg[(4 * g + s) * C::PA(l) + 16 * nt + i16];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) Aop[mt][s] = dimg[(4 * g + s) * C::PD + 16 * mt + i16];
        }
    }
    static __device__ __forceinline__ void dw(f32x4 (&dW)[C::DW_TILES], const float (&Aop)[C::MT(l)][4],
                                               const float (&Bop)[C::NT(l)][4]) {
        constexpr int MT = C::MT(l), NT = C::NT(l);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    mfma16_acc<(MT * NT > 2)>(dW[C::dwoff(l) + mt * NT + nt], Aop[mt][s], Bop[nt][s]);
    }
    // delta_{l-1} = (W_l^T dz) * act'(a_{l-1}); A operands from the transposed image, one k-group ahead
    static __device__ __forceinline__ void da(const TileRegs<S>& T, const float* __restrict__ lds, int i16, int g,
                                               const f32x4 (&dz)[C::MT(l)], f32x4 (&dzp)[C::MT(l > 0 ? l - 1 : 0)]) {
        if constexpr (l > 0) {
            constexpr int MTP = C::MT(l - 1);
            constexpr int KG = C::cdiv(C::out(l), 16);
            f32x4 acc[MTP];
#pragma unroll
            for (int m = 0; m < MTP; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float* trow = lds + C::toff(l) + i16 * C::LDT(l) + 4 * g;
            f32x4 An[MTP];
#pragma unroll
            for (int m = 0; m < MTP; ++m) An[m] = load_ks(trow + 16 * m * C::LDT(l), C::ksteps(C::out(l), 0));
#pragma unroll
            for (int kt = 0; kt < KG; ++kt) {
                f32x4 A4[MTP];
#pragma unroll
                for (int m = 0; m < MTP; ++m) A4[m] = An[m];
                if (kt + 1 < KG) {
#pragma unroll
                    for (int m = 0; m < MTP; ++m)
                        An[m] = load_ks(trow + 16 * m * C::LDT(l) + 16 * (kt + 1), C::ksteps(C::out(l), kt + 1));
                }
#pragma unroll
                for (int s = 0; s < C::ksteps(C::out(l), kt); ++s)
#pragma unroll
                    for (int m = 0; m < MTP; ++m) acc[m] = mfma16(A4[m][s], dz[kt][s], acc[m]);
            }
#pragma unroll
            for (int m = 0; m < MTP; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) dzp[m][r] = actc_bwd_mul<S::act(l - 1)>(acc[m][r], T.a[C::aroff(l - 1) + m][r]);
        }
    }
};

template <class S, int l>
struct BwdPipe {
    using C = FastCfg<S>;
    // (Aup, Bup): operands of dW_{l+1}, already in flight; consumed here under layer l's LDS traffic
    static __device__ __forceinline__ void run(f32x4 (&dW)[C::DW_TILES], const TileRegs<S>& T, const float* __restrict__ lds,
                                                float* wl, int i16, int g, const f32x4 (&dz)[C::MT(l)],
                                                const float (&Aup)[C::MT(l + 1)][4], const float (&Bup)[C::NT(l + 1)][4]) {
        float Aop[C::MT(l)][4], Bop[C::NT(l)][4];
        BwdOps<S, l>::issue(dz, wl, i16, g, Aop, Bop);
        SCHED_FENCE();
        TSTAMP(10 + 3 * l);
        BwdOps<S, l + 1>::dw(dW, Aup, Bup);
        SCHED_FENCE();
        TSTAMP(11 + 3 * l);
        if constexpr (l > 0) {
            f32x4 dzp[C::MT(l - 1)];
            BwdOps<S, l>::da(T, lds, i16, g, dz, dzp);
            SCHED_FENCE();
            TSTAMP(12 + 3 * l);
            BwdPipe<S, l - 1>::run(dW, T, lds, wl, i16, g, dzp, Aop, Bop);
        } else {
            BwdOps<S, 0>::dw(dW, Aop, Bop);
            TSTAMP(12);
        }
    }
};

// state of the VALU last layer (C::VL): weights/bias (loaded once) and per-lane dW/db partial sums
template <class S>
struct LastRegs {
    using C = FastCfg<S>;
    static constexpr int O = C::VL ? C::out(C::NL - 1) : 1, MTP = C::MT(C::NL >= 2 ? C::NL - 2 : 0);
    f32x4 w[O][MTP];       // W_L[o][16mt+4g+r]
    float b[O];
    f32x4 acc[O][MTP];     // sum over this lane's rows of dz_o * a_{L-1}[unit][row]
    float accb[O];
};

// likelihood for one output value: statistic (counted when `count`), returns dL/df * act'
#ifndef TBNN_FAST_BERN
#define TBNN_FAST_BERN 1
#endif
template <class S>
__device__ __forceinline__ float lik_delta(float fi, float yy, float inv_var, bool count, double& stat) {
    float da;
    if constexpr (S::BERN) {
        const float p = fminf(fmaxf(fi, 1e-8f), 1.f - 1e-7f);
        const bool inside = (fi > 1e-8f) && (fi < 1.f - 1e-7f);
#if TBNN_FAST_BERN
        // hardware log2 / reciprocal (1 ulp) instead of the library logf / log1pf / IEEE division sequences: the fused
        // kernels are VALU-bound next to the f32 MFMA (every instruction ~9 cycles), the likelihood is ~100 of them
        const float q = 1.f - p;
        const float t1 = (yy == 0.f) ? 0.f : yy * __logf(p);
        const float t2 = (1.f - yy == 0.f) ? 0.f : (1.f - yy) * __logf(q);
        if (count) stat += (double)(t1 + t2);
        da = inside ? (yy * __builtin_amdgcn_rcpf(p) - (1.f - yy) * __builtin_amdgcn_rcpf(q)) : 0.f;
#else
        const float t1 = (yy == 0.f) ? 0.f : yy * logf(p);
        const float t2 = (1.f - yy == 0.f) ? 0.f : (1.f - yy) * log1pf(-p);
        if (count) stat += (double)(t1 + t2);
        da = inside ? (yy / p - (1.f - yy) / (1.f - p)) : 0.f;
#endif
    } else {
        const float res = yy - fi;
        if (count) stat += (double)res * (double)res;
        da = res * inv_var;
    }
    return da * actc_bwd<S::LACT>(fi);
}

// one step of the row loop: one 16-row tile through forward, likelihood, backward.
// y: C::VL ? y[o] of row i16 (all lane groups) : D layout [MTL][4]
template <class S>
struct TileStep {
    using C = FastCfg<S>;
    static constexpr int YN = C::VL ? C::out(C::NL - 1) : 4 * C::MTL;
    static __device__ __forceinline__ void run(f32x4 (&dW)[C::DW_TILES], LastRegs<S>& LR, double& stat, const float* __restrict__ lds,
                                                float* wl, int i16, int g, float inv_var, const float (&x)[C::KS0],
                                                const float (&y)[YN], bool rvalid,
                                                const f32x4 (&A0)[C::MT(0)], const f32x4 (&B0)[C::MT(0)]) {
        constexpr int d_out = C::out(C::NL - 1), d_in = C::in(0), L = C::NL - 1, LM = C::NLM - 1;
        TileRegs<S> T;
#pragma unroll
        for (int t = 0; t < C::KS0; ++t) {
            T.x0[t] = x[t];
            const int u = 4 * t + g;
            if (u < d_in) wl[C::aoff(0) + i16 * C::PA(0) + u] = x[t];       // transposed image of x (ones column preset)
        }
#ifdef TBNN_IGLP
        __builtin_amdgcn_iglp_opt(TBNN_IGLP);
#endif
        TSTAMP(0);
        FwdLayer<S, 0>::run(T, lds, wl, i16, g, A0, B0);
        f32x4 dz[C::MT(LM)];
        if constexpr (C::VL) {
            // ---- last layer on the VALU: f_o = b_o + sum_u W[o][u] a[u]; lane (row i16, group g) holds units 16mt+4g+r
            constexpr int MTP = C::MT(L - 1);
            float dzl[d_out];
#pragma unroll
            for (int o = 0; o < d_out; ++o) {
                float p = 0.f;
#pragma unroll
                for (int mt = 0; mt < MTP; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) p = fmaf(LR.w[o][mt][r], T.a[C::aroff(L - 1) + mt][r], p);
                p += __shfl_xor(p, 16, 64);
                p += __shfl_xor(p, 32, 64);
                const float fi = actc_fwd<S::LACT>(p + LR.b[o]);
                dzl[o] = rvalid ? lik_delta<S>(fi, y[o], inv_var, g == 0, stat) : 0.f;
            }
            TSTAMP(9);
            // dW_L / db_L partial sums, delta_{L-1}
#pragma unroll
            for (int mt = 0; mt < MTP; ++mt) {
                f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int o = 0; o < d_out; ++o)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        LR.acc[o][mt][r] = fmaf(dzl[o], T.a[C::aroff(L - 1) + mt][r], LR.acc[o][mt][r]);
                        d[r] = fmaf(LR.w[o][mt][r], dzl[o], d[r]);
                    }
#pragma unroll
                for (int r = 0; r < 4; ++r) dz[mt][r] = actc_bwd_mul<S::act(L - 1)>(d[r], T.a[C::aroff(L - 1) + mt][r]);
            }
#pragma unroll
            for (int o = 0; o < d_out; ++o) LR.accb[o] += dzl[o];
        } else {
            // likelihood: f = a_L in D layout (unit 16mt+4g+r, row = lane&15)
#pragma unroll
            for (int mt = 0; mt < C::MTL; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int u = unit_of(d_out, 16 * mt + 4 * g + r, false);
                    dz[mt][r] = (rvalid && u >= 0) ? lik_delta<S>(T.a[C::aroff(L) + mt][r], y[4 * mt + r], inv_var, true, stat) : 0.f;
                }
            TSTAMP(9);
        }
        // top of the backward pipeline (layer LM)
        float Aop[C::MT(LM)][4], Bop[C::NT(LM)][4];
        BwdOps<S, LM>::issue(dz, wl, i16, g, Aop, Bop);
        TSTAMP(10 + 3 * LM);
        if constexpr (LM > 0) {
            f32x4 dzp[C::MT(LM - 1)];
            BwdOps<S, LM>::da(T, lds, i16, g, dz, dzp);
            SCHED_FENCE();
            TSTAMP(12 + 3 * LM);
            BwdPipe<S, LM - 1>::run(dW, T, lds, wl, i16, g, dzp, Aop, Bop);
        } else {
            BwdOps<S, 0>::dw(dW, Aop, Bop);
        }
    }
};

// dense slab write-out for the staged tiles [t0, t0+cnt) ([wave][tile][lane] x 16 B): tile t is handled
// by wave t % 4, every lane sums the 4 waves' copies of its 4 registers (fixed order) and stores them;
// no division anywhere
template <class S, int l>
struct SlabOut {
    using C = FastCfg<S>;
    static __device__ __forceinline__ void run(const float* buf, float* __restrict__ slab, int wave, int lane, int t0, int cnt) {
        constexpr int in = C::in(l), out = C::out(l), MT = C::MT(l), NT = C::NT(l);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int t = C::dwoff(l) + mt * NT + nt - t0;
                if (t >= 0 && t < cnt && (t & (FAST_WAVES - 1)) == wave) {
                    const f32x4* src = reinterpret_cast<const f32x4*>(buf) + t * 64 + lane;
                    const f32x4 c0 = src[0], c1 = src[C::EP_TILES * 64], c2 = src[2 * C::EP_TILES * 64], c3 = src[3 * C::EP_TILES * 64];
                    const int cs = 16 * nt + (lane & 15), row0 = 16 * mt + 4 * (lane >> 4);
                    // layer 0's input columns are natural (x), later layers' columns are slots
                    const int col = l == 0 ? (cs <= in ? cs : -1) : unit_of(in, cs, true);
                    if (col >= 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = unit_of(out, row0 + r, false);
                            if (row >= 0)
                                slab_store<(C::P() >= 2048)>(slab + C::offW(l) + (col < in ? row * in + col : in * out + row),
                                        (c0[r] + c1[r]) + (c2[r] + c3[r]));                // col == in: bias (ones column)
                        }
                    }
                }
            }
        if constexpr (l + 1 < C::NLM) SlabOut<S, l + 1>::run(buf, slab, wave, lane, t0, cnt);
    }
};

// host: flat parameter index -> offsets in the padded weight image: map[j] (W / bias images) and
// map[P+j] (transposed image, -1 if none)
template <class S, int l>
struct ImageMap {
    using C = FastCfg<S>;
    static void run(int* map) {
        constexpr int in = C::in(l), out = C::out(l);
        for (int i = 0; i < out; ++i) {
            for (int k = 0; k < in; ++k) {
                const int ri = slot_of(out, i), ck = l == 0 ? k : slot_of(in, k);      // layer 0 reads x in natural order
                map[C::offW(l) + i * in + k] = C::woff(l) + ri * C::LDW(l) + ck;
                map[C::P() + C::offW(l) + i * in + k] = (l >= 1 && l < C::NLM) ? C::toff(l) + ck * C::LDT(l) + ri : -1;   // W_l^T
            }
            map[C::offW(l) + in * out + i] = C::boff(l) + slot_of(out, i);
            map[C::P() + C::offW(l) + in * out + i] = -1;
        }
        if constexpr (l + 1 < C::NL) ImageMap<S, l + 1>::run(map);
    }
};

template <class S>
__global__ __launch_bounds__(FAST_THREADS, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_fwd_bwd_fast(
    NetDev nd, const float* __restrict__ qimg, const float* __restrict__ eta,
    const float* __restrict__ X, const float* __restrict__ Y, long n,
    float* __restrict__ slabs, int pitch, double* __restrict__ pstat, unsigned long long* __restrict__ stamps, ChainStride cs)
{
    using C = FastCfg<S>;
    // gridDim.y = chains of a multi-chain handle (tbnn_create_multi): this workgroup's chain
    if (chain_done(cs.ctl, cs.t, blockIdx.y)) return;           // a chain past its own L (per-chain step control)
    qimg += (size_t)blockIdx.y * cs.img; eta += (size_t)blockIdx.y * cs.eta; slabs += (size_t)blockIdx.y * cs.slab; pstat += (size_t)blockIdx.y * PSTAT_CAP;
#define TB_STAMP(i) do { if (stamps && threadIdx.x == 0 && blockIdx.x == 0) { stamps[i] = wall_clock64(); stamps[8 + i] = clock64(); } } while (0)
    TB_STAMP(0);
    static_assert(C::WB_FLOATS % 4 == 0 && C::STATIC_FLOATS % 4 == 0 && C::WAVE_FLOATS % 4 == 0, "images must be float4-addressable");
    static_assert(C::LDS_FLOATS * 4 + 64 <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS];
    __shared__ double red[FAST_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, g = lane >> 4;

    // ---- prologue: the padded weight image (built by k_update / k_make_image) -> LDS;
    // every 16-B load is issued before the first store (one latency, not eleven)
    {
        constexpr int N4 = C::STATIC_FLOATS / 4, IT = (N4 + FAST_THREADS - 1) / FAST_THREADS;
        const float4* src = reinterpret_cast<const float4*>(qimg);
        float4* dst = reinterpret_cast<float4*>(lds);
        float4 v[IT];
#pragma unroll
        for (int k = 0; k < IT; ++k) { const int e = tid + k * FAST_THREADS; v[k] = e < N4 ? src[e] : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
        for (int k = 0; k < IT; ++k) { const int e = tid + k * FAST_THREADS; if (e < N4) dst[e] = v[k]; }
    }
    float* wl = lds + C::STATIC_FLOATS + wave * C::WAVE_FLOATS;  // this wave's transposed activation/delta images
    {
        float4* z = reinterpret_cast<float4*>(wl);
        for (int e = lane; e < C::WAVE_FLOATS / 4; e += 64) z[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    TB_STAMP(1);

    f32x4 dW[C::DW_TILES];
#pragma unroll
    for (int t = 0; t < C::DW_TILES; ++t) dW[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float sigma = lik_sigma(nd, eta);
    const float inv_var = 1.f / (sigma * sigma);
    double stat = 0.0;
    constexpr int d_in = C::in(0), d_out = C::out(C::NL - 1);
    const long ntiles = (n + 15) / 16;
    const long W = (long)gridDim.x * FAST_WAVES;                  // waves in the grid
    const long wg = (long)blockIdx.x * FAST_WAVES + wave;

    // ones columns of the activation images (never overwritten: x image; own tile column when in(l) % 16 == 0)
    if (g == 0) wl[C::aoff(0) + i16 * C::PA(0) + d_in] = 1.f;
#pragma unroll
    for (int l = 1; l < C::NLM; ++l)
        if (C::in(l) % 16 == 0 && g == 0) wl[C::aoff(l) + i16 * C::PA(l) + C::in(l)] = 1.f;
    // layer-0 operands and bias tiles are the same for every tile: loaded once
    f32x4 A0[C::MT(0)], B0[C::MT(0)];
    FwdLayer<S, 0>::preload(A0, B0, lds, i16, g);

    // tile t of this wave = wg + t*W; x / y of the next tile are fetched under the current one
    constexpr int YN = TileStep<S>::YN;
    float xn[C::KS0], yn[YN];
    LastRegs<S> LR;
    if constexpr (C::VL) {
#pragma unroll
        for (int o = 0; o < d_out; ++o) {
            LR.b[o] = lds[C::boff(C::NL - 1) + slot_of(d_out, o)];
            LR.accb[o] = 0.f;
#pragma unroll
            for (int mt = 0; mt < LastRegs<S>::MTP; ++mt) {
                LR.w[o][mt] = *reinterpret_cast<const f32x4*>(lds + C::woff(C::NL - 1) + slot_of(d_out, o) * C::LDW(C::NL - 1) + 16 * mt + 4 * g);
                LR.acc[o][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    }
    auto fetch = [&](long tile) {
        const long row = tile * 16 + i16;
        const bool ok = tile < ntiles && row < n;
#pragma unroll
        for (int t = 0; t < C::KS0; ++t) {
            const int u = 4 * t + g;
            xn[t] = (ok && u < d_in) ? X[row * d_in + u] : 0.f;
        }
        if constexpr (C::VL) {
#pragma unroll
            for (int o = 0; o < d_out; ++o) yn[o] = ok ? Y[row * d_out + o] : 0.f;
        } else {
#pragma unroll
            for (int mt = 0; mt < C::MTL; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int u = unit_of(d_out, 16 * mt + 4 * g + r, false);
                    yn[4 * mt + r] = (ok && u >= 0) ? Y[row * d_out + u] : 0.f;
                }
        }
    };
    long tile = wg;
    fetch(tile);
    bool first = true;
    for (; tile < ntiles; tile += W) {
        float x[C::KS0], y[YN];
        const bool rv = tile * 16 + i16 < n;
#pragma unroll
        for (int t = 0; t < C::KS0; ++t) x[t] = xn[t];
#pragma unroll
        for (int k = 0; k < YN; ++k) y[k] = yn[k];
        fetch(tile + W);
        TileStep<S>::run(dW, LR, stat, lds, wl, i16, g, inv_var, x, y, rv, A0, B0);
        if (first) { TB_STAMP(2); first = false; }
    }
    TB_STAMP(3);
    mfma_drain_acc(dW);
    if (stamps && blockIdx.x == 0 && lane == 0) stamps[12 + wave] = wall_clock64();

    // ---- epilogue: every wave stages its dW tiles [wave][tile][lane] (16 B per lane), wave t%4 sums the
    // 4 copies of tile t in fixed order and writes the dense slab; EP_TILES per pass
    const double wtot = wave_sum_lane0(stat);
    if (lane == 0) red[wave] = wtot;
    float* slab = slabs + (size_t)blockIdx.x * pitch;
#pragma unroll
    for (int t0 = 0; t0 < C::DW_TILES; t0 += C::EP_TILES) {
        __syncthreads();                   // images (or the previous pass) are dead
        f32x4* mine = reinterpret_cast<f32x4*>(lds) + wave * (C::EP_TILES * 64);
#pragma unroll
        for (int t = t0; t < t0 + C::EP_TILES && t < C::DW_TILES; ++t) mine[(t - t0) * 64 + lane] = dW[t];
        __syncthreads();
        const int cnt = (C::DW_TILES - t0) < C::EP_TILES ? (C::DW_TILES - t0) : C::EP_TILES;
        TB_STAMP(5);
        SlabOut<S, 0>::run(lds, slab, wave, lane, t0, cnt);
        TB_STAMP(6);
    }
    if constexpr (C::VL) {
        // dW_L / db_L: every lane stages its per-row-group partials [wave][o][unit][i16]; thread (entry, wave)
        // sums the 16 lanes with 16-B reads, the 4 waves are combined with two lane shuffles (fixed order)
        constexpr int Lr = C::NL - 1, inL = C::in(Lr), MTP = LastRegs<S>::MTP, UP = 16 * MTP;
        static_assert(FAST_WAVES * 2 * (UP + 1) * 16 <= C::LDS_FLOATS, "last-layer staging does not fit");
        __syncthreads();
        float* lb = lds;
#pragma unroll
        for (int o = 0; o < d_out; ++o) {
#pragma unroll
            for (int mt = 0; mt < MTP; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    lb[((wave * d_out + o) * (UP + 1) + 16 * mt + 4 * g + r) * 16 + i16] = LR.acc[o][mt][r];
            if (g == 0) lb[((wave * d_out + o) * (UP + 1) + UP) * 16 + i16] = LR.accb[o];
        }
        __syncthreads();
        constexpr int NE = d_out * (inL + 1);
        for (int base = 0; base < NE * FAST_WAVES; base += FAST_THREADS) {
            const int t = base + tid;
            const int e = t >> 2, w = t & 3;
            float v = 0.f;
            int o = 0, u = 0;
            if (e < NE) {
                o = e / (inL + 1); u = e - o * (inL + 1);
                const f32x4* src = reinterpret_cast<const f32x4*>(lb + ((w * d_out + o) * (UP + 1) + (u < inL ? slot_of(inL, u) : UP)) * 16);
                const f32x4 p0 = src[0], p1 = src[1], p2 = src[2], p3 = src[3];
                v = ((p0[0] + p0[1]) + (p0[2] + p0[3])) + ((p1[0] + p1[1]) + (p1[2] + p1[3])) +
                    (((p2[0] + p2[1]) + (p2[2] + p2[3])) + ((p3[0] + p3[1]) + (p3[2] + p3[3])));
            }
            v += __shfl_xor(v, 1, 64);
            v += __shfl_xor(v, 2, 64);
            if (e < NE && w == 0) slab_store<(C::P() >= 2048)>(slab + C::offW(Lr) + (u < inL ? o * inL + u : inL * d_out + o), v);
        }
    }
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < FAST_WAVES; ++w) t += red[w];
        pstat[blockIdx.x] = t;
    }
    TB_STAMP(4);
#undef TB_STAMP
}

#ifndef TBNN_NO_FAST_REGISTRY
// ---------------------------------------------------------------------------
// registry of ahead-of-time instantiations (shapes of BASELINE.json's configs)
// ---------------------------------------------------------------------------
using ShapeC2 = Shape<TBNN_ACT_RELU, TBNN_ACT_NONE, false, 5, 50, 50, 50, 1>;      // configs[1], configs[2]
using ShapeC1 = Shape<TBNN_ACT_RELU, TBNN_ACT_NONE, false, 1, 10, 10, 1>;          // configs[0]
using ShapeTR = Shape<TBNN_ACT_TANH, TBNN_ACT_NONE, false, 1, 10, 10, 10, 1>;      // Examples/trainRegression.py
using ShapeT3 = Shape<TBNN_ACT_SIGMOID, TBNN_ACT_NONE, false, 4, 7, 3>;            // test shape: last layer on the MFMA path

template <class S>
static bool shape_matches(const NetDev& nd) {
    if (nd.nl != S::NL) return false;
    if ((nd.lik == TBNN_LIK_BERNOULLI) != S::BERN) return false;
    for (int l = 0; l < S::NL; ++l)
        if (nd.in[l] != S::D[l] || nd.out[l] != S::D[l + 1] || nd.act[l] != S::act(l)) return false;
    return true;
}

static inline int fast_lookup(const NetDev& nd) {
    if (shape_matches<ShapeC2>(nd)) return 0;
    if (shape_matches<ShapeC1>(nd)) return 1;
    if (shape_matches<ShapeTR>(nd)) return 2;
    if (shape_matches<ShapeT3>(nd)) return 3;
    return -1;
}
static inline const char* fast_name(int id) {
    switch (id) {
        case 0: return "fast<relu;5,50,50,50,1>";
        case 1: return "fast<relu;1,10,10,1>";
        case 2: return "fast<tanh;1,10,10,10,1>";
        case 3: return "fast<sigmoid;4,7,3>";
        default: return "fast<none>";
    }
}
// one workgroup (4 waves, 1 wave per SIMD) per CU; fewer when there are not enough tiles
static inline int fast_grid(int, long n) {
    const long ntiles = (n + 15) / 16;
    const long wgs = (ntiles + FAST_WAVES - 1) / FAST_WAVES;
    return (int)(wgs < 256 ? wgs : 256);
}
// qimg: the padded weight image of the position to evaluate
static inline int fast_launch(int id, int grid, hipStream_t st, const NetDev& nd, const float* qimg, const float* eta,
                              const float* X, const float* Y, long n, float* slabs, int pitch, double* pstat,
                              unsigned long long* stamps = nullptr, int nchains = 1, ChainStride cs = ChainStride{0, 0, 0, nullptr, 0}) {
    switch (id) {
        case 0: hipLaunchKernelGGL(k_fwd_bwd_fast<ShapeC2>, dim3(grid, nchains), dim3(FAST_THREADS), 0, st, nd, qimg, eta, X, Y, n, slabs, pitch, pstat, stamps, cs); break;
        case 1: hipLaunchKernelGGL(k_fwd_bwd_fast<ShapeC1>, dim3(grid, nchains), dim3(FAST_THREADS), 0, st, nd, qimg, eta, X, Y, n, slabs, pitch, pstat, stamps, cs); break;
        case 2: hipLaunchKernelGGL(k_fwd_bwd_fast<ShapeTR>, dim3(grid, nchains), dim3(FAST_THREADS), 0, st, nd, qimg, eta, X, Y, n, slabs, pitch, pstat, stamps, cs); break;
        case 3: hipLaunchKernelGGL(k_fwd_bwd_fast<ShapeT3>, dim3(grid, nchains), dim3(FAST_THREADS), 0, st, nd, qimg, eta, X, Y, n, slabs, pitch, pstat, stamps, cs); break;
        default: return -1;
    }
    return 0;
}
// floats in the padded weight image and the parameter -> image-offset map
static inline int fast_image_floats(int id) {
    switch (id) {
        case 0: return FastCfg<ShapeC2>::STATIC_FLOATS;
        case 1: return FastCfg<ShapeC1>::STATIC_FLOATS;
        case 2: return FastCfg<ShapeTR>::STATIC_FLOATS;
        case 3: return FastCfg<ShapeT3>::STATIC_FLOATS;
        default: return 0;
    }
}
static inline void fast_image_map(int id, int* map) {
    switch (id) {
        case 0: ImageMap<ShapeC2, 0>::run(map); break;
        case 1: ImageMap<ShapeC1, 0>::run(map); break;
        case 2: ImageMap<ShapeTR, 0>::run(map); break;
        case 3: ImageMap<ShapeT3, 0>::run(map); break;
        default: break;
    }
}
#endif  // TBNN_NO_FAST_REGISTRY
