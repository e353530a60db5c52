// k_fwd_bwd_tall: the fused forward + likelihood + backward pass for networks with a LONG first-layer fan-in and narrow
// hidden layers -- the reference's own classification tutorial, 784 -> 20 -> 20 -> 1 on MNIST pixels
// (docs/ClassificationExample.md:103-173), is the type: dW_0 alone is 2 x 50 tiles, more than one wave's accumulator file,
// and the rows (3 KB each) are what the pass streams.  Until round 4 such shapes ran on the layered family (five launches,
// the rows read twice per gradient).
//
// One WORKGROUP walks GROUPS of G row tiles (G = 1 .. 4 by the row count, tall_group_tiles); the fan-in is split over its four
// waves (wave w owns the column tiles [w CH, (w+1) CH) of the rows, of W_0 and of dW_0).  A group runs in three phases:
//   A. layer 0 of the group's tiles: W_0's chunk lives in the wave's REGISTERS for the whole launch, as MFMA A operands (MT0 x CH x 4
//      VGPRs), fetched once from the padded image k_update maintains; every wave multiplies its chunk of a tile's rows (16-B buffer
//      loads: lane (row, g) holds columns 16kt+4g .. +3, which is the MFMA B operand as it stands) into partial pre-activations
//      and leaves them in the exchange buffer [tile][wave]; units 16 T .. 16 T + F - 1 (F <= 4) of a 16 T + F unit layer take the
//      16-block 4x4x1 MFMA; barrier;
//   B. the narrow rest of the network -- middle layers on MFMAs with the weights in LDS, the <= 2-output last layer on the VALU, the
//      likelihood, the delta chain, the narrow layers' dW -- for ONE tile per wave (wave w: the group's tile w): the four partials
//      summed in fixed order; delta_0 goes to the group's shared blocks; barrier.  (Until round 4's last version every wave ran this
//      stretch for the one tile of its workgroup redundantly: it is latency-bound -- dependent MFMAs, LDS round trips,
//      transcendentals -- and half of a tile's cycles; shared out over a group it costs a quarter per tile.)
//   C. dW_0 += delta_0^T [x, 1] for the group's tiles, this wave's column tiles, accumulated in the wave's AccVGPRs (MT0 x CH x 4):
//      A operand from the shared delta_0 blocks, B operand = the rows once more as the transposed view (4-byte buffer loads, L2
//      hits, requested two k-steps ahead).
// The epilogue's fixed-order sum over the four waves -- the one every fused family has -- puts the narrow layers' pieces together;
// every wave writes its own column chunk of dW_0 (16-byte write-through stores after an LDS transpose).
// Launch signature, slab layout and FusedOps family are the narrow family's (one gradient slab per workgroup, k_update
// reduces).  The rows are read from HBM once per gradient (and once more from L2 for dW_0).
//
// Reference math: layer.py:278 (W@a+b), activationFunctions.py:36/49/62, likelihood.py:88-94,226-236,
// BNN_functions.py:23-32; reverse mode SURVEY A12; the path: network.py:394-408.
#pragma once
#include "kernels_mid.hpp"
#include <atomic>

// diagnostic build (-DTBNN_TILE_STAMPS, tools/experiments/tall_stamps.py): shader-clock stamps of workgroup 0 / wave 0 along the launch
#ifdef TBNN_TILE_STAMPS
#define TALL_STAMP(k) do { if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) g_tile_stamps[k] = clock64(); } while (0)
#define TALL_STAMP_G(k) do { if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0 && grp == 0) g_tile_stamps[k] = clock64(); } while (0)
#else
#define TALL_STAMP(k) do {} while (0)
#define TALL_STAMP_G(k) do {} while (0)
#endif

// Row tiles per group (1 .. 4 = TallCfg::GMAX: the instantiations of the kernel) at a given tile count.  A workgroup walks a group in three phases: layer 0 of its tiles (a per tile),
// the narrow stretch of ONE tile per wave (b per group, whatever the tile count), dW_0 of its tiles (part of a): with a ~ 2.3 us and
// b ~ 5.8 us incl. the two barriers (784 -> 20 -> 20 -> 1) a launch over `ntiles` tiles costs rounds(g) x (g a + b), rounds(g) = ceil(ceil(ntiles / g) / 256
// workgroups).  Big groups share the narrow stretch out best (3.65 us per tile at g = 4 against 5.0 at g = 1), small ones fill the
// chip at few rows.  The same function sizes the grid and picks the instantiation.
static inline std::atomic<int>& tall_forced_g() { static std::atomic<int> g{0}; return g; }
static inline void tall_read_forced_g() { const char* e = getenv("TBNN_TALL_G"); tall_forced_g().store(e ? atoi(e) : 0, std::memory_order_relaxed); }
static inline int tall_group_tiles(long ntiles, int gmax) {
    // A/B runs: TBNN_TALL_G.  The environment is read where the grid is sized (tall_grid, at tbnn_set_data) and remembered; a launch
    // -- once per leapfrog step on the host's hot path -- takes the remembered value, so it calls no getenv and agrees with the grid of
    // the most recent set_data (a diagnostic knob: whatever the pairing, the group loop strides over the grid and stays correct)
    const int forced = tall_forced_g().load(std::memory_order_relaxed);
    if (forced >= 1 && forced <= gmax) return forced;
    int best = 1; long best_cost = 0;
    for (int g = 1; g <= gmax; g = g < 4 ? g + 1 : 2 * g) {
        const long groups = (ntiles + g - 1) / g, rounds = (groups + 255) / 256;
        const long cost = rounds * (23 * g + 58 + (g > 4 ? 24 * (g / 4 - 1) : 0));      // (g = 8: every wave runs the narrow stretch twice)
        if (g == 1 || cost < best_cost) { best = g; best_cost = cost; }
    }
    return best;
}

// Waves per workgroup = ways the fan-in is split: four, one per SIMD.  (Eight -- two per SIMD, 256 registers each, all
// accumulators in ArchVGPRs because the compiler halves a wave's budget as soon as one AccVGPR is asked for -- was built and
// measured in round 4: 784 -> 20 -> 20 -> 1 at 12,000 rows 29.2 us against 28.4 us.  The two waves of a SIMD run the same program
// in lockstep between the per-tile barriers, so one's serial stretch does not run under the other's MFMA blocks, and the
// narrow layers -- which every wave then ran for the workgroup's one tile -- doubled.  Not kept.)
template <class S, int NW_ = 4>
struct TallCfg {
    static constexpr int NW = NW_;
    static_assert(NW == 4, "four waves per workgroup");
    static constexpr int THREADS = 64 * NW;
    static constexpr int NL = S::NL;
    static_assert(NL >= 2, "the tall path needs a hidden layer");
    static constexpr int in(int l) { return S::D[l]; }
    static constexpr int out(int l) { return S::D[l + 1]; }
    static constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
    static constexpr int r4(int a) { return (a + 3) & ~3; }
    static constexpr int LL = NL - 1;                 // last layer
    static constexpr int d_in = in(0), d_out = out(LL);
    // <= 2 outputs: the last layer on the VALU; 3 .. 16 (round 6, as kernels_mid.hpp): one more MFMA layer with ONE output tile, the
    // likelihood reads the tile -- a ten-class 784 -> 20 -> 20 -> 10 is the reference's MNIST tutorial with all its digits
    static constexpr bool VL = d_out <= 2;
    static constexpr int NM = VL ? NL - 2 : NL - 1;   // MFMA layers behind layer 0: 1 .. NM (weights in LDS)
    static_assert(d_out <= 16, "the last layer is one output tile at most");
    static constexpr int TR(int l) { return cdiv(in(l), 16); }          // register tiles of a_l (l >= 1: slot order of kernels_fast.hpp)
    static constexpr int TA(int l) { return cdiv(in(l) + 1, 16); }      // block tiles of a_l (with the ones slot)
    static constexpr int ksteps(int K, int kg) { int rem = K - 16 * kg; return rem >= 16 ? 4 : (rem <= 0 ? 0 : (rem + 3) / 4); }
    static constexpr int KG(int K) { return cdiv(K, 16); }
    static constexpr int maxT() { int m = 0; for (int l = 1; l <= (VL ? LL : NL); ++l) m = TR(l) > m ? TR(l) : m; return m; }
    static constexpr int MAXT = maxT();
    // layer 0: the rows' columns in their own order (slot = column, the ones slot at column d_in), split over the waves
    static constexpr int MT0 = TR(1);
    // fringe: a first layer of 16 T + F units with 1 <= F <= 4 (20 = 16 + 4: the MNIST example) computes units 16 T .. 16 T + F - 1
    // on the 16-block v_mfma_f32_4x4x1 (8 cycles instead of 32 per k-step and column tile), in the forward pass and in dW_0
#ifndef TALL_FRINGE
#define TALL_FRINGE 1
#endif
    static constexpr int F0 = out(0) % 16;
    static constexpr bool FR0 = TALL_FRINGE && F0 >= 1 && F0 <= 4;
    static constexpr int MTF = FR0 ? MT0 - 1 : MT0;                     // full tiles of layer 0's units
    static constexpr int NT0 = cdiv(d_in + 1, 16);                      // column tiles of [x, 1]
    static constexpr int CH = cdiv(NT0, NW);                            // column tiles per wave
    static constexpr int NTP = NW * CH;                         // padded tile count of the W_0 image
    static constexpr bool ALIGNED = d_in % 4 == 0;                      // rows are 16-B aligned: one load per lane and tile
    // dW accumulator tiles of the middle layers (layer l: TR(l+1) x TA(l)); dW_0: MT0 x CH per wave
    static constexpr int dwt(int l) { return TR(l + 1) * TA(l); }
    static constexpr int dwoff(int l) { int o = 0; for (int m = 1; m < l; ++m) o += dwt(m); return o; }
    static constexpr int DWM_TILES = dwoff(NM + 1);
    // ---- weight image in HBM (k_update scatters theta into it through image_map)
    //   W_0: MFMA A-operand granules [M tile][column tile (NTP)][lane (i, g)][s]: lane holds W_0[slot 16t+i][column 16kt+4g+s]
    //   then the part that is copied to LDS: biases 0..NM in slot order, W_LL [d_out][16 TR(LL)], b_LL, the middle layers
    //   row-major [16 TR(l+1) out slots][LDM(l)] (pitch == 4 mod 8: the strided W^T reads of the delta chain are conflict-free)
    static constexpr int W0_FLOATS = MT0 * NTP * 256;
    static constexpr int boff(int l) { int o = 0; for (int m = 0; m < l; ++m) o += 16 * TR(m + 1); return o; }   // in the LDS part
    static constexpr int WLP = VL ? 16 * TR(LL) : 0;
    static constexpr int WL_OFF = boff(NM + 1);
    static constexpr int BL_OFF = WL_OFF + d_out * WLP;
    static constexpr int PERM_FLOATS = r4(BL_OFF + (VL ? d_out : 0));
    static constexpr int LDM(int l) { return 16 * TR(l) + MID_WPAD; }
    static constexpr int wmoff(int l) { int o = PERM_FLOATS; for (int m = 1; m < l; ++m) o += 16 * TR(m + 1) * LDM(m); return o; }
    static constexpr int SMALL_FLOATS = r4(wmoff(NM + 1));              // the LDS-resident part
    static constexpr int IMG_FLOATS = W0_FLOATS + SMALL_FLOATS;
    // ---- LDS: [small image][exchange: G tiles x 4 waves x MT0 tiles x 64 lanes x 4 | epilogue staging of dW_0: 4 waves x CH blocks]
    //           [delta_0 of the group's G tiles: MT0 blocks each][per wave: a_l, delta_l blocks of the middle layers]
    // row tiles per group: at most GMAX.  -DTALL_GMAX=8 (two tiles per wave in the narrow stretch, one after the other, where the LDS holds
    // eight tiles' exchange buffer) was measured: 60,000 rows 76.0 us against 73.8 with groups of four, 24,000 rows 42.8 against 41.4 --
    // the second pass through the narrow stretch costs more than the barriers it saves
    static constexpr int lds_main(int gmax) {
        const int ex = gmax * NW * MT0 * 256, stg = NW * CH * 256;
        int wave = 0;
        for (int m = 1; m <= NM; ++m) wave += (TA(m) + TR(m + 1)) * 256;
        return SMALL_FLOATS + (ex > stg ? ex : stg) + gmax * MT0 * 256 + NW * wave;
    }
#ifndef TALL_C_REVERSE
#define TALL_C_REVERSE 1
#endif
#ifndef TALL_GMAX
#define TALL_GMAX 4
#endif
    static constexpr int GMAX = (TALL_GMAX >= 8 && lds_main(8) * 4 + 128 <= 160 * 1024) ? 8 : 4;
    static constexpr int G = GMAX;
    static constexpr int EX_OFF = SMALL_FLOATS;
    static constexpr int EX_FLOATS = G * NW * MT0 * 256;
    static constexpr int STG_OFF = EX_OFF;
    static constexpr int STG_FLOATS = NW * CH * 256;
    static constexpr int DB0_OFF = EX_OFF + (EX_FLOATS > STG_FLOATS ? EX_FLOATS : STG_FLOATS);
    static constexpr int DB0_FLOATS = G * MT0 * 256;
    static constexpr int aboff(int l) { int o = 0; for (int m = 1; m < l; ++m) o += TA(m) * 256; return o; }                    // a_l, l = 1..NM
    static constexpr int dboff(int l) { int o = aboff(NM + 1); for (int m = 1; m < l; ++m) o += TR(m + 1) * 256; return o; }    // delta_l, l = 1..NM
    static constexpr int WAVE_FLOATS = dboff(NM + 1);
    static constexpr int WAVE_OFF = DB0_OFF + DB0_FLOATS;
    static constexpr int LDS_MAIN = WAVE_OFF + NW * WAVE_FLOATS;
    static_assert(LDS_MAIN == lds_main(GMAX), "LDS layout");
    // epilogue staging of the middle layers' dW tiles ([wave][tile][lane] x 16 B) and of the last layer's sums
    static constexpr int EP_FLOATS = NW * (DWM_TILES > 0 ? DWM_TILES : 1) * 256;
    static constexpr int LL_FLOATS = VL ? NW * d_out * (16 * TR(LL) + 1) : 0;
    static constexpr int LDS_FLOATS = LDS_MAIN > EP_FLOATS ? (LDS_MAIN > LL_FLOATS ? LDS_MAIN : LL_FLOATS) : (EP_FLOATS > LL_FLOATS ? EP_FLOATS : LL_FLOATS);
    // ---- parameters
    static constexpr int offW(int l) { int p = 0; for (int m = 0; m < l; ++m) p += in(m) * out(m) + out(m); return p; }
    static constexpr int P() { return offW(NL); }
    static constexpr bool LDS_OK = LDS_FLOATS * 4 + 64 <= 160 * 1024;
    // a wave's registers, roughly: its chunks of W_0 and dW_0, the rows' chunk, the narrow layers' accumulators + working set
    static constexpr int REG_EST = 2 * 4 * MT0 * CH + 4 * CH + 4 * DWM_TILES + (VL ? 4 * d_out * TR(LL) : 0) + 12 * MAXT + 48;
};
template <class S> struct TallPick { static constexpr int NW = 4; };

template <class S, int NW>
struct TallLast {              // per-lane partial sums of the VALU last layer's dW / db
    using C = TallCfg<S, NW>;
    static constexpr int NO = C::VL ? C::d_out : 1, NT = C::VL ? C::TR(C::LL) : 1;      // (nothing when the last layer is an MFMA layer)
    f32x4 acc[NO][NT];
    float accb[NO];
};

// FWD: forward pass only (network.predict, network.py:141-171; predictor.py:132-155 with blockIdx.y = network)
// G: row tiles per group (1 .. 4 -- tall_group_tiles -- a template parameter: with the group size a run-time value the tile loops
// of phases A and C end in branches, every tile becomes a scheduling region of its own, and the launch measured 9 % slower)
template <class S, int NW, bool FWD = false, int G = 4>
__global__ __launch_bounds__(64 * NW, NW / 4) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4))) void k_fwd_bwd_tall(
    NetDev nd, const float* __restrict__ qimgs, long img_stride, const float* __restrict__ eta,
    const float* __restrict__ X, const float* __restrict__ Y, long n,
    float* __restrict__ slabs, int pitch, double* __restrict__ pstat, float* __restrict__ fouts, long out_stride, ChainStride cs)
{
    using C = TallCfg<S, NW>;
    // gridDim.y: networks of an ensemble (FWD) or chains of a multi-chain handle (tbnn_create_multi; cs.img == img_stride then)
    if constexpr (!FWD) {
        if (chain_done(cs.ctl, cs.t, blockIdx.y)) return;       // a chain past its own L (per-chain step control)
        eta += (size_t)blockIdx.y * cs.eta; slabs += (size_t)blockIdx.y * cs.slab; pstat += (size_t)blockIdx.y * PSTAT_CAP;
    }
    static_assert(C::LDS_OK, "LDS budget");
    constexpr int TALL_WAVES = NW, TALL_THREADS = 64 * NW;
    constexpr bool ACC_A = true;       // accumulators pinned to AccVGPRs by asm MFMAs (as in kernels_mid.hpp)
    __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS];
    __shared__ double red[NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, g = lane >> 4;
    constexpr int d_in = C::d_in, d_out = C::d_out, NM = C::NM, LL = C::LL, MT0 = C::MT0, CH = C::CH;
    const float* qimg = qimgs + (size_t)blockIdx.y * img_stride;
    float* fout = FWD ? fouts + (size_t)blockIdx.y * out_stride : nullptr;
    const long ntiles = (n + 15) / 16;
    const int kt0 = wave * CH;                          // first column tile of this wave

    // the rows of a tile, this wave's columns: lane (row i16, g) holds columns 16 (kt0 + c) + 4g .. +3.  16-B aligned rows: one
    // buffer load per lane and column tile through a resource that covers exactly the tile's valid rows (rows past n and tiles
    // past the end read as zeros: no branch, no 64-bit address per load; the column tile is the instruction's immediate offset);
    // fix() zeroes the columns past d_in (which lie in range: the next row) and sets the ones slot when the values are used
    const int vbase = i16 * (d_in * 4) + 64 * kt0 + 16 * g;
    auto fetch = [&](long tile, f32x4 (&xv)[CH]) __attribute__((always_inline)) {
        if constexpr (C::ALIGNED) {
            const bool in = tile < ntiles;
            const long left = n - tile * 16;
            const int rows = in ? (int)(left < 16 ? left : 16) : 0;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X) + (in ? tile : 0) * 16 * d_in, 0, rows * d_in * 4, 0x00020000);
#pragma unroll
            for (int c = 0; c < CH; ++c) xv[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vbase, 64 * c, 0));
        } else {
            const long row = tile * 16 + i16;
            const bool ok = tile < ntiles && row < n;
            const float* xr = X + row * d_in;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int col0 = 16 * (kt0 + c) + 4 * g;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 4; ++s) if (ok && col0 + s < d_in) v[s] = xr[col0 + s];
                xv[c] = v;
            }
        }
    };
    auto fix = [&](f32x4 v, int c) __attribute__((always_inline)) {
        if (16 * (kt0 + c) + 16 > d_in) {              // wave-uniform: only the column tile(s) at and behind the end of the rows
            const int col0 = 16 * (kt0 + c) + 4 * g;
#pragma unroll
            for (int s = 0; s < 4; ++s) v[s] = col0 + s < d_in ? v[s] : (col0 + s == d_in ? 1.f : 0.f);
        }
        return v;
    };
    // the rows once more for dW_0, as the transposed operand: lane (column i16, row phase g) of k-step s holds
    // x[row 4 s + g][column 16 (kt0 + c) + i16] -- 4-byte buffer loads (hits: the group's rows were read for layer 0 a moment ago);
    // the resource covers the tile's valid rows, the tail wave puts the ones slot / zeros behind the end of the rows
    const int vT = (g * d_in + 16 * kt0 + i16) * 4;
    auto rsrc_rows = [&](long tile) __attribute__((always_inline)) {
        const bool in = tile < ntiles;
        const long left = n - tile * 16;
        const int rows = in ? (int)(left < 16 ? left : 16) : 0;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X) + (in ? tile : 0) * 16 * d_in, 0, rows * d_in * 4, 0x00020000);
    };
    static_assert(G >= 1 && G <= C::G, "group size");
    constexpr int Gr = G;
    const long ngroups = (ntiles + Gr - 1) / Gr;
    f32x4 xn[CH];
    long grp = blockIdx.x;
    TALL_STAMP(16);
    fetch(grp * Gr, xn);

    // ---- prologue: this wave's chunk of W_0 -> registers; the small image -> LDS; the per-wave blocks zeroed
    // (fringe tile: the 4x4x1 MFMA's A operand of lane l is W_0[unit 16 T + (l & 3)][k-phase l / 16] -- fringe unit e sits in slot
    // 16 T + 4 e (slot_of), i.e. in image lane 16 (l / 16) + 4 (l & 3))
    f32x4 Wr[MT0][CH];
    const int flane = (lane & 48) | ((lane & 3) << 2);
    {
        const f32x4* w4 = reinterpret_cast<const f32x4*>(qimg);
#pragma unroll
        for (int t = 0; t < MT0; ++t)
#pragma unroll
#ifdef TALL_DBG_NOPRO          // diagnostic build: what the W_0 chunk's loads cost
            for (int c = 0; c < CH; ++c) Wr[t][c] = f32x4{0.001f * lane, 0.002f, 0.003f * c, 0.004f * t};
#else
            for (int c = 0; c < CH; ++c) Wr[t][c] = w4[(size_t)(t * C::NTP + kt0 + c) * 64 + ((C::FR0 && t == C::MTF) ? flane : lane)];
#endif
    }
    float* wl = lds + C::WAVE_OFF + wave * C::WAVE_FLOATS;
    {
        constexpr int N4 = C::SMALL_FLOATS / 4;
        const float4* src = reinterpret_cast<const float4*>(qimg + C::W0_FLOATS);
        float4* dst = reinterpret_cast<float4*>(lds);
        for (int e = tid; e < N4; e += TALL_THREADS) dst[e] = src[e];
        float4* z = reinterpret_cast<float4*>(wl);
        for (int e = lane; e < C::WAVE_FLOATS / 4; e += 64) z[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();

    TALL_STAMP(17);
    const float sigma = FWD ? 1.f : lik_sigma(nd, eta);
    const float inv_var = 1.f / (sigma * sigma);
    double stat = 0.0;
    f32x4 dW0[FWD ? 1 : MT0 * CH];
    f32x4 dWm[(FWD || C::DWM_TILES == 0) ? 1 : C::DWM_TILES];
#pragma unroll
    for (int t = 0; t < (FWD ? 1 : MT0 * CH); ++t) dW0[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < ((FWD || C::DWM_TILES == 0) ? 1 : C::DWM_TILES); ++t) dWm[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    TallLast<S, NW> LR;
    constexpr int LRO = TallLast<S, NW>::NO, LRT = TallLast<S, NW>::NT, YN = C::VL ? d_out : 4;
#pragma unroll
    for (int o = 0; o < LRO; ++o) {
        LR.accb[o] = 0.f;
#pragma unroll
        for (int t = 0; t < LRT; ++t) LR.acc[o][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // VALU last layer's weights, slot order: lane (r, g) holds slots 16t+4g .. +3
    f32x4 wL[LRO][LRT];
    if constexpr (C::VL) {
#pragma unroll
        for (int o = 0; o < d_out; ++o)
#pragma unroll
            for (int t = 0; t < C::TR(LL); ++t) wL[o][t] = *reinterpret_cast<const f32x4*>(lds + C::WL_OFF + o * C::WLP + 16 * t + 4 * g);
    }

    const bool tail_wave = 16 * (kt0 + CH) > d_in;             // (wave-uniform) this wave holds the end of the rows
    for (; grp < ngroups; grp += gridDim.x) {
        // ---- A: layer 0 of the group's G row tiles, this wave's share of the fan-in -> partial pre-activations in the exchange buffer.
        // Two accumulator sets (even / odd column tiles) keep the MFMA chain four deep (a lone pair of accumulators is revisited
        // after 32 cycles, 8 short of the dependent latency)
        // The group's rows arrive two tiles ahead of their use: tile 0 was requested a group ago (xn), tile 1 is requested here (xm), tile
        // t + 2 behind tile t's MFMAs into the set tile t vacated; behind xn's last tile of the group that is the NEXT group's tile 0.  (One
        // tile ahead -- 2.2 k cycles of MFMAs -- is less than the rows' round trip: phase A took 4 k cycles per tile.)
        f32x4 xm[G > 1 ? CH : 1];
        if constexpr (G > 1) fetch(grp * Gr + 1, xm);
#pragma unroll
        for (int tg = 0; tg < G; ++tg) {
            f32x4 x[CH];
            // (only the tail wave has anything to fix: a real branch -- left to the compiler it becomes four selects per column tile
            // in every wave, and on this chip a vector instruction costs what an MFMA pass costs)
            if (tail_wave) {
                asm volatile("; tail wave");
#pragma unroll
                for (int c = 0; c < CH; ++c) x[c] = fix((tg & 1) ? xm[G > 1 ? c : 0] : xn[c], c);
            } else {
#pragma unroll
                for (int c = 0; c < CH; ++c) x[c] = (tg & 1) ? xm[G > 1 ? c : 0] : xn[c];
            }
            f32x4 acc0[2][MT0];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int t = 0; t < MT0; ++t) acc0[h][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < CH; ++c)
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int t = 0; t < MT0; ++t) {
                        if (C::FR0 && t == C::MTF) acc0[c & 1][t] = mfma4(Wr[t][c][s], x[c][s], acc0[c & 1][t]);
                        else acc0[c & 1][t] = mfma16(Wr[t][c][s], x[c][s], acc0[c & 1][t]);
                    }
            // x's registers are free from here on
            if (tg + 2 < G) {
                if constexpr (G > 2) { if (tg & 1) fetch(grp * Gr + tg + 2, xm); else fetch(grp * Gr + tg + 2, xn); }
            } else if ((tg & 1) == 0) fetch((grp + gridDim.x) * Gr, xn);      // xn's last tile of this group: the next group's tile 0
            f32x4* ex = reinterpret_cast<f32x4*>(lds + C::EX_OFF) + (tg * TALL_WAVES + wave) * (MT0 * 64);
#pragma unroll
            for (int t = 0; t < MT0; ++t) {
                f32x4 z = acc0[0][t] + acc0[1][t];
                if (C::FR0 && t == C::MTF) {
                    // block b = lane / 4 = (k-phase g, row quad): register m of lane (row, g) = unit 16 T + m over the k-slots of phase
                    // g; summed over the four phases, unit 16 T + g dropped into register 0 of lane group g (its slot, 16 T + 4 g)
#pragma unroll
                    for (int m = 0; m < 4; ++m) z[m] = gsum(z[m]);
                    const float sel = g == 0 ? z[0] : (g == 1 ? z[1] : (g == 2 ? z[2] : z[3]));
                    z = f32x4{sel, 0.f, 0.f, 0.f};
                }
                ex[t * 64 + lane] = z;
            }
        }
        // dW_0's first operands (tile 0, k-step 0): requested here, used behind the narrow stretch
        float Bq[3][FWD ? 1 : CH];
        __amdgpu_buffer_rsrc_t rsT = rsrc_rows(grp * Gr + (TALL_C_REVERSE ? Gr - 1 : 0));
        auto ldT = [&](int s, float (&B)[FWD ? 1 : CH]) __attribute__((always_inline)) {
            if constexpr (!FWD) {
#pragma unroll
#ifdef TALL_DBG_NOLDT      // timing experiment only (wrong gradients): what the transposed re-read of the rows costs in phase C
                for (int c = 0; c < CH; ++c) B[c] = 0.001f * (lane + s + c);
#else
                for (int c = 0; c < CH; ++c) B[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsT, vT, (4 * s * d_in + 16 * c) * 4, 0));
#endif
            }
        };
#ifndef TALL_DBG_NOC       // timing experiment only (no dW_0 at all: wrong gradients): what the pass costs WITHOUT phase C and without its slab stores --
                           // the first kernel of a two-pass form whose second pass would contract delta_0^T X over column stripes (round 6, NOTES)
        ldT(0, Bq[0]);
        if (Gr * 4 > 1) ldT(1, Bq[1]);
#endif
        TALL_STAMP_G(21);
        __syncthreads();
        TALL_STAMP_G(22);

        // ---- B: the narrow rest of the network for ONE tile per wave (tile = group's tile `wave`): the four partials summed in fixed
        // order, middle layers on MFMAs from the LDS image, last layer on the VALU, likelihood, delta chain, the narrow layers' dW
#pragma unroll
        for (int tb = 0; tb < (Gr + NW - 1) / NW; ++tb) {
        const int slot = tb * NW + wave;                      // this wave's tile of the group
        if (slot < Gr) {
        const long tile = grp * Gr + slot;
        const bool rvalid = tile * 16 + i16 < n;
        float y[YN];          // <= 2 outputs: y[o] in every lane group; else the D layout of the output tile (lane (row, g): slots 4g .. 4g + 3)
#pragma unroll
        for (int o = 0; o < YN; ++o) {
            const int u = C::VL ? o : unit_of(d_out, 4 * g + o, false);
            y[o] = (!FWD && rvalid && u >= 0) ? Y[(tile * 16 + i16) * d_out + u] : 0.f;
        }
        f32x4 a[C::MAXT];                  // the current layer's input a_l, D layout: tile t reg j of lane (r, g) = slot 16t+4g+j of row r
        {
            const f32x4* ex = reinterpret_cast<const f32x4*>(lds + C::EX_OFF) + slot * (TALL_WAVES * MT0 * 64);
#pragma unroll
            for (int t = 0; t < MT0; ++t) {
                f32x4 z;
                {
                    const f32x4 e0 = ex[(0 * MT0 + t) * 64 + lane], e1 = ex[(1 * MT0 + t) * 64 + lane];
                    const f32x4 e2 = ex[(2 * MT0 + t) * 64 + lane], e3 = ex[(3 * MT0 + t) * 64 + lane];
                    z = (e0 + e1) + (e2 + e3);
                }
                const f32x4 b = *reinterpret_cast<const f32x4*>(lds + C::boff(0) + 16 * t + 4 * g);
                z = b + z;
#pragma unroll
                for (int r = 0; r < 4; ++r) a[t][r] = actc_fwd<S::act(0)>(z[r]);
            }
        }

        // ---- middle layers, forward: a_l -> a_{l+1}; a_l (+ ones slot) to this wave's blocks for dW_l / act'
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
            f32x4 acc[MT];
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t] = *reinterpret_cast<const f32x4*>(lds + C::boff(l) + 16 * t + 4 * g);
            const float* wrow = lds + C::wmoff(l) + i16 * C::LDM(l) + 4 * g;
            sfor<0, KGn>(SFOR_LAMBDA(kg) {
                constexpr int kg = SFOR_VAL(kg);
                f32x4 A[MT];
#pragma unroll
                for (int t = 0; t < MT; ++t) A[t] = load_ks(wrow + 16 * t * C::LDM(l) + 16 * kg, C::ksteps(K, kg));
#pragma unroll
                for (int s = 0; s < C::ksteps(K, kg); ++s)
#pragma unroll
                    for (int t = 0; t < MT; ++t) acc[t] = mfma16(A[t][s], a[kg][s], acc[t]);
            });
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) a[t][r] = actc_fwd<S::act(l)>(acc[t][r]);
        });

        // ---- last layer on the VALU: f_o = b_o + sum_u W[o][u] a_LL[u]
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
            float p = 0.f;
#pragma unroll
            for (int t = 0; t < TP; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) p = fmaf(wL[o][t][r], a[t][r], p);
            const float fi = actc_fwd<S::LACT>(lane_group_sum(p) + lds[C::BL_OFF + o]);
            if constexpr (FWD) {
                if (rvalid && g == 0) fout[(size_t)o * n + tile * 16 + i16] = fi;      // [d_out][n]
                dzl[o] = 0.f;
            } else {
                dzl[o] = rvalid ? lik_delta<S>(fi, y[o], inv_var, g == 0, stat) : 0.f;
            }
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
            // the last layer's dW / db sums of this tile's rows; delta_{LL-1} = (W_LL^T dz_LL) * act'(a_LL)
#pragma unroll
            for (int o = 0; o < d_out; ++o) {
                LR.accb[o] += dzl[o];
#pragma unroll
                for (int t = 0; t < TP; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) LR.acc[o][t][r] = fmaf(dzl[o], a[t][r], LR.acc[o][t][r]);
            }
#pragma unroll
            for (int t = 0; t < TP; ++t) {
                f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int o = 0; o < d_out; ++o)
#pragma unroll
                    for (int r = 0; r < 4; ++r) d[r] = fmaf(wL[o][t][r], dzl[o], d[r]);
                dz[t] = actc_bwd_mul4<S::act(LL - 1), false>(d, a[t]);
            }
        }
        // ---- backward through the middle layers l = NM .. 1
        sfor<0, NM>(SFOR_LAMBDA(li) {
            constexpr int l = NM - SFOR_VAL(li);
            constexpr int TZ = C::TR(l + 1), TAl = C::TA(l), MU = C::TR(l), K = C::out(l), KGn = C::KG(K);
            float* dbl = wl + C::dboff(l);
            const float* ab = wl + C::aboff(l);
#pragma unroll
            for (int t = 0; t < TZ; ++t) *reinterpret_cast<f32x4*>(dbl + t * 256 + i16 * 16 + 4 * g) = dz[t];
            // dW_l += delta_l^T [a_l, 1] over this tile's 16 rows (k-step s = rows 4 s .. 4 s + 3)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float Aop[TZ], Bop[TAl];
#pragma unroll
                for (int t = 0; t < TZ; ++t) Aop[t] = dbl[t * 256 + 64 * s + lane];
#pragma unroll
                for (int u = 0; u < TAl; ++u) Bop[u] = ab[u * 256 + 64 * s + lane];
#pragma unroll
                for (int t = 0; t < TZ; ++t)
#pragma unroll
                    for (int u = 0; u < TAl; ++u) mfma16_acc<(ACC_A && TZ * TAl > 2)>(dWm[C::dwoff(l) + t * TAl + u], Aop[t], Bop[u]);
            }
            // delta_{l-1} = (W_l^T delta_l) * act'(a_l): W^T from the row-major image, strided 4-B reads
            f32x4 acc[MU];
#pragma unroll
            for (int u = 0; u < MU; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float* wcol = lds + C::wmoff(l) + 4 * g * C::LDM(l) + i16;
            sfor<0, KGn>(SFOR_LAMBDA(kg) {
                constexpr int kg = SFOR_VAL(kg);
#pragma unroll
                for (int s = 0; s < C::ksteps(K, kg); ++s)
#pragma unroll
                    for (int u = 0; u < MU; ++u) acc[u] = mfma16(wcol[(16 * kg + s) * C::LDM(l) + 16 * u], dz[kg][s], acc[u]);
            });
#pragma unroll
            for (int u = 0; u < MU; ++u) {
                const f32x4 al = *reinterpret_cast<const f32x4*>(ab + u * 256 + i16 * 16 + 4 * g);
                dz[u] = actc_bwd_mul4<S::act(l - 1), false>(acc[u], al);
            }
        });
        // delta_0 of this wave's tile -> the group's shared blocks
        {
            float* db0 = lds + C::DB0_OFF + slot * (MT0 * 256);
#pragma unroll
            for (int t = 0; t < MT0; ++t) *reinterpret_cast<f32x4*>(db0 + t * 256 + i16 * 16 + 4 * g) = dz[t];
        }
        }   // !FWD
        }   // slot < Gr
        }   // tb
        TALL_STAMP_G(23);
        __syncthreads();
        TALL_STAMP_G(24);

        // ---- C: dW_0 += delta_0^T [x, 1] over the group's tiles, this wave's column tiles; one k-step (rows 4 s .. 4 s + 3 of
        // tile tg) at a time, its rows requested two k-steps ahead
#ifdef TALL_DBG_NOC
        if constexpr (!FWD) {                                  // delta_0 of this wave's tile leaves for HBM instead (what the second pass would read)
            if (wave < Gr) {
                const float* db0 = lds + C::DB0_OFF + wave * (MT0 * 256);
                float* dst = slabs + ((size_t)(grp * Gr + wave) * MT0) * 256;          // (inside the slab area: its contents mean nothing in this build)
#pragma unroll
                for (int t = 0; t < MT0; ++t) *reinterpret_cast<f32x4*>(dst + t * 256 + lane * 4) = *reinterpret_cast<const f32x4*>(db0 + t * 256 + lane * 4);
            }
        }
#else
        if constexpr (!FWD) {
#pragma unroll
            for (int q = 0; q < 4 * G; ++q) {
                // (TALL_C_REVERSE: the group's LAST tile first -- its rows were read most recently in phase A and are the likeliest
                // to be in L2 still: 32 workgroups' groups of three are 4.8 MB against an XCD's 4 MB)
                const int tg = TALL_C_REVERSE ? Gr - 1 - (q >> 2) : (q >> 2), s = q & 3;
                if (q + 2 < 4 * Gr) {                                // two k-steps (52 MFMAs) ahead, three operand sets
                    if (((q + 2) & 3) == 0) rsT = rsrc_rows(grp * Gr + (TALL_C_REVERSE ? Gr - 1 - ((q + 2) >> 2) : ((q + 2) >> 2)));
                    ldT((q + 2) & 3, Bq[(q + 2) % 3]);
                }
                float (&Bop)[CH] = Bq[q % 3];
                if (tail_wave) {
                    asm volatile("; tail wave");
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        const int col = 16 * (kt0 + c) + i16;
                        Bop[c] = col < d_in ? Bop[c] : (col == d_in ? 1.f : 0.f);
                    }
                }
                const float* db0 = lds + C::DB0_OFF + tg * (MT0 * 256);
                float Aop[MT0];
#pragma unroll
                for (int t = 0; t < MT0; ++t) {
                    // fringe tile: block b = (row phase g, column quad), A[m] = delta_0[row 4 s + g][unit 16 T + m] (slot 4 m of the
                    // block); B is the full tiles' operand as it stands; register m = dW_0[16 T + m][column] over the rows of phase g
                    if (C::FR0 && t == C::MTF) Aop[t] = db0[t * 256 + (4 * s + g) * 16 + 4 * (lane & 3)];
                    else Aop[t] = db0[t * 256 + 64 * s + lane];
                }
                // column-tile-major: a fringe tile's 2-pass 4x4x1 then follows an 8-pass 16x16x4, whose passes hide the two wait states the
                // asm form carries in front (TBNN_ASM_MFMA_NOP); thirteen 4x4x1 in a row paid them in full (8 cycles each behind an 8-cycle
                // MFMA: round 6)
#pragma unroll
                for (int c = 0; c < CH; ++c)
#pragma unroll
                    for (int t = 0; t < MT0; ++t) {
                        if (C::FR0 && t == C::MTF) mfma4_acc<(ACC_A && MT0 * CH > 2)>(dW0[t * CH + c], Aop[t], Bop[c]);
                        else mfma16_acc<(ACC_A && MT0 * CH > 2)>(dW0[t * CH + c], Aop[t], Bop[c]);
                    }
            }
        }
#endif
    }
    TALL_STAMP(18);
    if constexpr (!FWD) {
    if constexpr (ACC_A) {
        mfma_drain_acc(dW0);
        if constexpr (C::DWM_TILES > 0) mfma_drain_acc(dWm);
    }
    float* slab = slabs + (size_t)blockIdx.x * pitch;
#ifndef TALL_WT
#define TALL_WT 1
#endif
    constexpr bool WT = TALL_WT && C::P() >= 2048;
    // ---- dW_0: every wave writes its own column tiles.  D layout: lane (n = i16, g) reg r = dW[out slot 16t+4g+r][column 16kt+n]:
    // a lane's four registers are four ROWS of the slab.  With 16-B aligned rows the tiles of one M tile are turned through the
    // wave's own share of the staging area (the exchange buffer, dead by now; written [m][n], read back lane-linearly: lane l holds row
    // l / 4, columns 4 (l % 4) .. +3)
    // and leave as 16-byte write-through stores -- a 4-byte sc1 store is one fabric write per lane
#ifndef TALL_DBG_NOC
    {
        constexpr int out0 = C::out(0);
        sfor<0, MT0>(SFOR_LAMBDA(t) {
            constexpr int t = SFOR_VAL(t);
            if constexpr (C::ALIGNED) {
                float* stg = lds + C::STG_OFF + wave * (CH * 256);
                sfor<0, CH>(SFOR_LAMBDA(c) {
                    constexpr int c = SFOR_VAL(c);
                    const f32x4 v = dW0[t * CH + c];
#pragma unroll
                    for (int r = 0; r < 4; ++r) stg[c * 256 + (4 * g + r) * 16 + i16] = v[r];
                });
                const int m = lane >> 2, n4 = (lane & 3) * 4;
                constexpr bool FT = C::FR0 && t == C::MTF;             // fringe tile: staged as [row phase g][unit m = r], summed over the phases here
                const int row = FT ? (m < C::F0 ? 16 * t + m : -1) : unit_of(out0, 16 * t + m, false);
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(stg + c * 256 + 4 * (FT ? lane & 15 : lane));
                    if constexpr (FT) {
#pragma unroll
                        for (int ph = 1; ph < 4; ++ph) v += *reinterpret_cast<const f32x4*>(stg + c * 256 + 64 * ph + 4 * (lane & 15));
                    }
                    const int col = 16 * (kt0 + c) + n4;
                    if (row >= 0) {
#ifdef TALL_DBG_NOSTORE        // diagnostic build: what the dW_0 slab stores cost
                        if (col < d_in && v[0] == 123.456f) store16<WT>(slab + row * d_in + col, v);
#else
                        if (col < d_in) store16<WT>(slab + row * d_in + col, v);
#endif
                        else if (col == d_in) slab_store<WT>(slab + d_in * out0 + row, v[0]);      // the ones column: db_0
                    }
                }
            } else {
                sfor<0, CH>(SFOR_LAMBDA(c) {
                    constexpr int c = SFOR_VAL(c);
                    const int col = 16 * (kt0 + c) + i16;
                    f32x4 v = dW0[t * CH + c];
                    constexpr bool FT = C::FR0 && t == C::MTF;
                    if constexpr (FT) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = gsum(v[r]);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = FT ? ((g == 0 && r < C::F0) ? 16 * t + r : -1) : unit_of(out0, 16 * t + 4 * g + r, false);
                        if (row >= 0 && col <= d_in)
                            slab_store<WT>(slab + (col < d_in ? row * d_in + col : d_in * out0 + row), v[r]);
                    }
                });
            }
        });
    }
#endif
    TALL_STAMP(19);
    const double wtot = wave_sum_lane0(stat);
    if (lane == 0) red[wave] = wtot;
    // ---- middle layers: the four waves' k-step shares of every tile, summed in fixed order
    if constexpr (C::DWM_TILES > 0) {
        __syncthreads();                   // images / blocks are dead
        f32x4* mine = reinterpret_cast<f32x4*>(lds) + wave * (C::DWM_TILES * 64);
        sfor<0, C::DWM_TILES>(SFOR_LAMBDA(t) { mine[SFOR_VAL(t) * 64 + lane] = dWm[SFOR_VAL(t)]; });
        __syncthreads();
        sfor<1, NM + 1>(SFOR_LAMBDA(l) {
            constexpr int l = SFOR_VAL(l);
            constexpr int inl = C::in(l), outl = C::out(l), MT = C::TR(l + 1), NT = C::TA(l);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int t = C::dwoff(l) + mt * NT + nt;
                    if ((t & (TALL_WAVES - 1)) == wave) {
                        const f32x4* src = reinterpret_cast<const f32x4*>(lds) + t * 64 + lane;
                        const f32x4 c0 = src[0], c1 = src[C::DWM_TILES * 64], c2 = src[2 * C::DWM_TILES * 64], c3 = src[3 * C::DWM_TILES * 64];
                        const int col = unit_of(inl, 16 * nt + i16, true);          // inl: the ones pseudo-unit (bias column)
                        if (col >= 0) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int row = unit_of(outl, 16 * mt + 4 * g + r, false);
                                if (row >= 0)
                                    slab_store<WT>(slab + C::offW(l) + (col < inl ? row * inl + col : inl * outl + row), (c0[r] + c1[r]) + (c2[r] + c3[r]));
                            }
                        }
                    }
                }
        });
    }
    if constexpr (C::VL) {
        // VALU last layer: reduce the per-row partials over the 16 lanes of a lane group, then over the 4 waves
        constexpr int TP = C::TR(LL), inL = C::in(LL);
        float* lb = lds;                               // [wave][o][slot], then [wave][o] biases
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
        for (int e = tid; e < d_out * (inL + 1); e += TALL_THREADS) {
            const int o = e / (inL + 1), u = e - o * (inL + 1);
            const int s = u < inL ? slot_of(inL, u) : 16 * TP;
            float v[TALL_WAVES];
#pragma unroll
            for (int w = 0; w < TALL_WAVES; ++w) v[w] = lb[(w * d_out + o) * (16 * TP + 1) + s];
            slab_store<WT>(slab + C::offW(LL) + (u < inL ? o * inL + u : inL * d_out + o), (v[0] + v[1]) + (v[2] + v[3]));
        }
    }
    TALL_STAMP(20);
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < TALL_WAVES; ++w) t += red[w];
        pstat[blockIdx.x] = t;
    }
    }   // !FWD
}

// host: flat parameter index -> offset in the weight image (map[j]); no transposed copy (map[P + j] = -1)
template <class S>
static void tall_image_map(int* map) {
    using C = TallCfg<S, TallPick<S>::NW>;
    const int P = C::P();
    for (int l = 0; l < C::NL; ++l) {
        const int in = C::in(l), out = C::out(l), ow = C::offW(l);
        for (int i = 0; i < out; ++i) {
            const int ri = slot_of(out, i);
            for (int k = 0; k < in; ++k) {
                int m0;
                if (l == 0) m0 = (((ri / 16) * C::NTP + k / 16) * 64 + ((k % 16) / 4) * 16 + ri % 16) * 4 + k % 4;
                else if (C::VL && l == C::LL) m0 = C::W0_FLOATS + C::WL_OFF + i * C::WLP + slot_of(in, k);
                else m0 = C::W0_FLOATS + C::wmoff(l) + ri * C::LDM(l) + slot_of(in, k);
                map[ow + i * in + k] = m0;
                map[P + ow + i * in + k] = -1;
            }
            map[ow + in * out + i] = C::W0_FLOATS + ((C::VL && l == C::LL) ? C::BL_OFF + i : C::boff(l) + ri);
            map[P + ow + in * out + i] = -1;
        }
    }
}

// one workgroup per group of four 16-row tiles and pass; at most one workgroup per CU, the groups dealt out evenly
template <class S>
static inline int tall_grid(long n) {
    const long ntiles = (n + 15) / 16;
    tall_read_forced_g();
    const int g = tall_group_tiles(ntiles, TallCfg<S, TallPick<S>::NW>::GMAX);
    const long ngroups = (ntiles + g - 1) / g, rounds = (ngroups + 255) / 256;
    return (int)((ngroups + rounds - 1) / rounds);
}
template <class S>
static inline int tall_launch_t(int grid, hipStream_t st, const NetDev& nd, const float* qimg, const float* eta, const float* X,
                                const float* Y, long n, float* slabs, int pitch, double* pstat, int nchains = 1, ChainStride cs = ChainStride{0, 0, 0, nullptr, 0}) {
    constexpr int NW = TallPick<S>::NW;
#define TALL_LAUNCH(GG) hipLaunchKernelGGL((k_fwd_bwd_tall<S, NW, false, GG>), dim3(grid, nchains), dim3(64 * NW), 0, st, nd, qimg, cs.img, eta, X, Y, n, \
                                           slabs, pitch, pstat, (float*)nullptr, 0L, cs)
    constexpr int GMAX = TallCfg<S, NW>::GMAX;
    switch (tall_group_tiles((n + 15) / 16, GMAX)) {
        case 1: TALL_LAUNCH(1); break;
        case 2: TALL_LAUNCH(2); break;
        case 3: TALL_LAUNCH(3); break;
        case 4: TALL_LAUNCH(4); break;
        default: if constexpr (GMAX >= 8) TALL_LAUNCH(8); break;
    }
#undef TALL_LAUNCH
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
// forward only: `nets` networks (grid.y; images img_stride floats apart), fouts[net][d_out][n]
template <class S>
static inline int tall_forward_t(int gx, int nets, hipStream_t st, const float* qimgs, long img_stride, const float* X, long n,
                                 float* fouts, long out_stride) {
    NetDev nd{};
    constexpr int NW = TallPick<S>::NW;
    // (forward only: groups of four -- the caller's grid is one workgroup per four row tiles, capped for ensembles: narrow_forward)
    hipLaunchKernelGGL((k_fwd_bwd_tall<S, NW, true, 4>), dim3(gx, nets), dim3(64 * NW), 0, st, nd, qimgs, img_stride, (const float*)nullptr, X,
                       (const float*)nullptr, n, (float*)nullptr, 0, (double*)nullptr, fouts, out_stride, ChainStride{0, 0, 0, nullptr, 0});
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
