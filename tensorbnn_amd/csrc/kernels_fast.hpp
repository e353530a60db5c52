// Shape-specialised fused forward + likelihood + backward kernel for gfx950:
// f32 MFMA (v_mfma_f32_16x16x4_f32) for every contraction, weights resident in
// LDS, activations and ALL dW accumulators resident in registers.
//
// One wave owns a 16-row tile of the training matrix at a time and needs no
// cross-wave synchronisation inside the row loop:
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

template <int... Ds>
struct Shape {
    static constexpr int NL = sizeof...(Ds) - 1;
    static constexpr int D[sizeof...(Ds)] = {Ds...};
};

template <class S>
struct FastCfg {
    static constexpr int NL = S::NL;
    static constexpr int in(int l) { return S::D[l]; }
    static constexpr int out(int l) { return S::D[l + 1]; }
    static constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
    static constexpr int MT(int l) { return cdiv(out(l), 16); }          // M tiles of layer l's output
    static constexpr int NT(int l) { return cdiv(in(l) + 1, 16); }       // N tiles of dW_l (+1: ones column -> db)
    static constexpr int KG(int l) { return cdiv(in(l), 16); }           // 16-unit k groups of layer l's input
    static constexpr int LDW(int l) { return 16 * KG(l) + 4; }           // pitch of the W_l image (== 4 mod 8)
    static constexpr int PA(int l) { return 16 * NT(l) + 4; }            // pitch of the A_{l} (input of layer l) image
    static constexpr int maxMT() { int m = 0; for (int l = 0; l < NL; ++l) m = MT(l) > m ? MT(l) : m; return m; }
    static constexpr int PD = 16 * maxMT() + 4;                          // pitch of the delta image
    // number of valid k-steps s in group kt of a K dimension of size K (unit = 16kt+4g+s)
    static constexpr int ksteps(int K, int kt) { int rem = K - 16 * kt; return rem >= 4 ? 4 : (rem < 0 ? 0 : rem); }
    // LDS layout (floats)
    static constexpr int woff(int l) { int o = 0; for (int m = 0; m < l; ++m) o += 16 * MT(m) * LDW(m); return o; }
    static constexpr int W_FLOATS = woff(NL);
    static constexpr int boff(int l) { int o = W_FLOATS; for (int m = 0; m < l; ++m) o += 16 * MT(m); return o; }
    static constexpr int WB_FLOATS = boff(NL);
    static constexpr int aoff(int l) { int o = 0; for (int m = 0; m < l; ++m) o += 16 * PA(m); return o; }   // per wave
    static constexpr int doff = aoff(NL);
    static constexpr int WAVE_FLOATS = doff + 16 * PD;
    static constexpr int P() { int p = 0; for (int l = 0; l < NL; ++l) p += in(l) * out(l) + out(l); return p; }
    static constexpr int offW(int l) { int p = 0; for (int m = 0; m < l; ++m) p += in(m) * out(m) + out(m); return p; }
    static constexpr int REGION = (FAST_WAVES * WAVE_FLOATS > P()) ? FAST_WAVES * WAVE_FLOATS : P();
    static constexpr int LDS_FLOATS = WB_FLOATS + REGION;
    static constexpr int dwoff(int l) { int o = 0; for (int m = 0; m < l; ++m) o += MT(m) * NT(m); return o; }
    static constexpr int DW_TILES = dwoff(NL);
    static constexpr int aroff(int l) { int o = 0; for (int m = 0; m < l; ++m) o += MT(m); return o; }   // act register tiles
    static constexpr int ACT_TILES = aroff(NL);
};

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Everything one wave keeps in registers across the row loop.
template <class S>
struct FastRegs {
    using C = FastCfg<S>;
    f32x4 dW[C::DW_TILES];      // dW_l tiles [mt][nt], D layout
    f32x4 a[C::ACT_TILES];      // outputs of every layer for the current tile
};

template <class S, int l>
struct FwdLayer {
    using C = FastCfg<S>;
    // bprev: B operands of this layer = previous layer's output tiles (l >= 1)
    static __device__ __forceinline__ void run(FastRegs<S>& R, const float* __restrict__ lds, float* wl, int i16, int g,
                                                const NetDev& nd, const float (&x0)[C::cdiv(C::in(0), 4)]) {
        constexpr int MT = C::MT(l);
        f32x4 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            acc[mt] = *reinterpret_cast<const f32x4*>(lds + C::boff(l) + 16 * mt + 4 * g);   // bias in D layout
        if constexpr (l == 0) {
            // natural k mapping: step t covers units 4t+g, B operand straight from X
            constexpr int KS = C::cdiv(C::in(0), 4);
#pragma unroll
            for (int t = 0; t < KS; ++t) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const float A = lds[C::woff(0) + (16 * mt + i16) * C::LDW(0) + 4 * t + g];
                    acc[mt] = mfma16(A, x0[t], acc[mt]);
                }
            }
        } else {
            constexpr int KG = C::KG(l);
#pragma unroll
            for (int kt = 0; kt < KG; ++kt) {
                f32x4 A4[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    A4[mt] = *reinterpret_cast<const f32x4*>(lds + C::woff(l) + (16 * mt + i16) * C::LDW(l) + 16 * kt + 4 * g);
#pragma unroll
                for (int s = 0; s < C::ksteps(C::in(l), kt); ++s) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[mt] = mfma16(A4[mt][s], R.a[C::aroff(l - 1) + kt][s], acc[mt]);
                }
            }
        }
        const int act = nd.act[l];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = act_fwd(acc[mt][r], act);
            R.a[C::aroff(l) + mt] = v;
        }
        if constexpr (l + 1 < C::NL) {
            // transposed image of a_{l+1} (= input of layer l+1) for dW_{l+1}; ones column at unit in(l+1)
            constexpr int u1 = C::in(l + 1);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x4 v = R.a[C::aroff(l) + mt];
                if constexpr (u1 % 16 != 0) {
                    if (mt == u1 / 16 && g == (u1 % 16) / 4) v[u1 % 4] = 1.f;
                }
                *reinterpret_cast<f32x4*>(wl + C::aoff(l + 1) + i16 * C::PA(l + 1) + 16 * mt + 4 * g) = v;
            }
        }
    }
};

template <class S, int l>
struct BwdLayer {
    using C = FastCfg<S>;
    // dz: delta tiles of layer l (D layout).  Accumulates dW_l, returns delta of layer l-1 in dzp.
    static __device__ __forceinline__ void run(FastRegs<S>& R, const float* __restrict__ lds, float* wl, int i16, int g,
                                                const NetDev& nd, const f32x4 (&dz)[C::MT(l)],
                                                f32x4 (&dzp)[C::MT(l > 0 ? l - 1 : 0)]) {
        constexpr int MT = C::MT(l), NT = C::NT(l);
        // delta image [row][unit]
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            *reinterpret_cast<f32x4*>(wl + C::doff + i16 * C::PD + 16 * mt + 4 * g) = dz[mt];
        // dW_l += dz . a_{l-1}^T : k = data row 4g+s
        float Aop[MT][4], Bop[NT][4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) Aop[mt][s] = wl[C::doff + (4 * g + s) * C::PD + 16 * mt + i16];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) Bop[nt][s] = wl[C::aoff(l) + (4 * g + s) * C::PA(l) + 16 * nt + i16];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    R.dW[C::dwoff(l) + mt * NT + nt] = mfma16(Aop[mt][s], Bop[nt][s], R.dW[C::dwoff(l) + mt * NT + nt]);
        if constexpr (l > 0) {
            // delta_{l-1} = (W_l^T dz) * act'(a_{l-1}) ; M = in(l) units, K = out(l) units
            constexpr int MTP = C::MT(l - 1);
            constexpr int KG = C::cdiv(C::out(l), 16);
            f32x4 acc[MTP];
#pragma unroll
            for (int m = 0; m < MTP; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < KG; ++kt) {
#pragma unroll
                for (int s = 0; s < C::ksteps(C::out(l), kt); ++s) {
#pragma unroll
                    for (int m = 0; m < MTP; ++m) {
                        const float A = lds[C::woff(l) + (16 * kt + 4 * g + s) * C::LDW(l) + 16 * m + i16];
                        acc[m] = mfma16(A, dz[kt][s], acc[m]);
                    }
                }
            }
            const int act = nd.act[l - 1];
#pragma unroll
            for (int m = 0; m < MTP; ++m) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dzp[m][r] = acc[m][r] * act_bwd(R.a[C::aroff(l - 1) + m][r], act);
            }
        }
    }
};

template <class S, int l>
struct BwdChain {
    using C = FastCfg<S>;
    static __device__ __forceinline__ void run(FastRegs<S>& R, const float* __restrict__ lds, float* wl, int i16, int g,
                                                const NetDev& nd, const f32x4 (&dz)[C::MT(l)]) {
        f32x4 dzp[C::MT(l > 0 ? l - 1 : 0)];
        BwdLayer<S, l>::run(R, lds, wl, i16, g, nd, dz, dzp);
        if constexpr (l > 0) BwdChain<S, l - 1>::run(R, lds, wl, i16, g, nd, dzp);
    }
};

template <class S, int l>
struct FwdChain {
    using C = FastCfg<S>;
    static __device__ __forceinline__ void run(FastRegs<S>& R, const float* __restrict__ lds, float* wl, int i16, int g,
                                                const NetDev& nd, const float (&x0)[C::cdiv(C::in(0), 4)]) {
        FwdLayer<S, l>::run(R, lds, wl, i16, g, nd, x0);
        if constexpr (l + 1 < C::NL) FwdChain<S, l + 1>::run(R, lds, wl, i16, g, nd, x0);
    }
};

// stage theta into the padded LDS images
template <class S, int l>
struct StageW {
    using C = FastCfg<S>;
    static __device__ __forceinline__ void run(float* lds, const float* __restrict__ q, int tid) {
        constexpr int in = C::in(l), out = C::out(l);
        for (int e = tid; e < in * out; e += FAST_THREADS) {
            const int i = e / in, k = e - i * in;
            lds[C::woff(l) + i * C::LDW(l) + k] = q[C::offW(l) + e];
        }
        for (int e = tid; e < out; e += FAST_THREADS) lds[C::boff(l) + e] = q[C::offW(l) + in * out + e];
        if constexpr (l + 1 < C::NL) StageW<S, l + 1>::run(lds, q, tid);
    }
};

// write one wave's dW tiles into the LDS gradient buffer (first = store, else add)
template <class S, int l>
struct FlushDW {
    using C = FastCfg<S>;
    static __device__ __forceinline__ void run(const FastRegs<S>& R, float* gbuf, int i16, int g, bool first) {
        constexpr int in = C::in(l), out = C::out(l), MT = C::MT(l), NT = C::NT(l);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = 16 * nt + i16;                 // in-unit (== in -> bias)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * mt + 4 * g + r;       // out-unit
                    if (row < out && col <= in) {
                        const int idx = C::offW(l) + (col < in ? row * in + col : in * out + row);
                        const float v = R.dW[C::dwoff(l) + mt * NT + nt][r];
                        gbuf[idx] = first ? v : gbuf[idx] + v;
                    }
                }
            }
        if constexpr (l + 1 < C::NL) FlushDW<S, l + 1>::run(R, gbuf, i16, g, first);
    }
};

template <class S>
__global__ __launch_bounds__(FAST_THREADS, 1) void k_fwd_bwd_fast(
    NetDev nd, const float* __restrict__ q, const float* __restrict__ eta,
    const float* __restrict__ X, const float* __restrict__ Y, long n,
    float* __restrict__ slabs, double* __restrict__ pstat)
{
    using C = FastCfg<S>;
    __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS];
    __shared__ double red[FAST_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, g = lane >> 4;

    for (int e = tid; e < C::LDS_FLOATS; e += FAST_THREADS) lds[e] = 0.f;
    __syncthreads();
    StageW<S, 0>::run(lds, q, tid);
    float* wl = lds + C::WB_FLOATS + wave * C::WAVE_FLOATS;      // this wave's images
    // ones column of the layer-0 input image (never overwritten)
    if (g == 0) wl[C::aoff(0) + i16 * C::PA(0) + C::in(0)] = 1.f;
    // ones column of later images when in(l) is a multiple of 16 (own tile column, never overwritten)
#pragma unroll
    for (int l = 1; l < C::NL; ++l)
        if (C::in(l) % 16 == 0 && g == 0) wl[C::aoff(l) + i16 * C::PA(l) + C::in(l)] = 1.f;
    __syncthreads();

    FastRegs<S> R;
#pragma unroll
    for (int t = 0; t < C::DW_TILES; ++t) R.dW[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float sigma = lik_sigma(nd, eta);
    const float inv_var = 1.f / (sigma * sigma);
    double stat = 0.0;
    constexpr int d_in = C::in(0), d_out = C::out(C::NL - 1);
    constexpr int KS0 = C::cdiv(d_in, 4);
    const long ntiles = (n + 15) / 16;

    for (long tile = (long)blockIdx.x * FAST_WAVES + wave; tile < ntiles; tile += (long)gridDim.x * FAST_WAVES) {
        const long row = tile * 16 + i16;
        const bool rvalid = row < n;
        // layer-0 B operand (natural k mapping: unit 4t+g) + transposed image of x
        float x0[KS0];
#pragma unroll
        for (int t = 0; t < KS0; ++t) {
            const int u = 4 * t + g;
            x0[t] = (rvalid && u < d_in) ? X[row * d_in + u] : 0.f;
            if (u < d_in) wl[C::aoff(0) + i16 * C::PA(0) + u] = x0[t];
        }
        FwdChain<S, 0>::run(R, lds, wl, i16, g, nd, x0);

        // likelihood: f = a_L in D layout (unit 16mt+4g+r, row = lane&15)
        constexpr int MTL = C::MT(C::NL - 1);
        f32x4 dz[MTL];
        const int actL = nd.act[C::NL - 1];
#pragma unroll
        for (int mt = 0; mt < MTL; ++mt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int u = 16 * mt + 4 * g + r;
                float d = 0.f;
                if (rvalid && u < d_out) {
                    const float fi = R.a[C::aroff(C::NL - 1) + mt][r];
                    const float y = Y[row * d_out + u];
                    float da;
                    if (nd.lik == TBNN_LIK_BERNOULLI) {
                        const float p = fminf(fmaxf(fi, 1e-8f), 1.f - 1e-7f);
                        const bool inside = (fi > 1e-8f) && (fi < 1.f - 1e-7f);
                        const float t1 = (y == 0.f) ? 0.f : y * logf(p);
                        const float t2 = (1.f - y == 0.f) ? 0.f : (1.f - y) * log1pf(-p);
                        stat += (double)(t1 + t2);
                        da = inside ? (y / p - (1.f - y) / (1.f - p)) : 0.f;
                    } else {
                        const float res = y - fi;
                        stat += (double)res * (double)res;
                        da = res * inv_var;
                    }
                    d = da * act_bwd(fi, actL);
                }
                dz[mt][r] = d;
            }
        }
        BwdChain<S, C::NL - 1>::run(R, lds, wl, i16, g, nd, dz);
    }

    // ---- combine the 4 waves' dW in LDS (fixed order), write the slab
    const double wtot = wave_sum(stat);
    __syncthreads();                       // all images dead from here on
    if (lane == 0) red[wave] = wtot;
    float* gbuf = lds + C::WB_FLOATS;
    for (int w = 0; w < FAST_WAVES; ++w) {
        if (wave == w) FlushDW<S, 0>::run(R, gbuf, i16, g, w == 0);
        __syncthreads();
    }
    float* slab = slabs + (size_t)blockIdx.x * C::P();
    for (int e = tid; e < C::P(); e += FAST_THREADS) slab[e] = gbuf[e];
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < FAST_WAVES; ++w) t += red[w];
        pstat[blockIdx.x] = t;
    }
}

// ---------------------------------------------------------------------------
// registry of ahead-of-time instantiations (shapes of BASELINE.json's configs)
// ---------------------------------------------------------------------------
using ShapeC2 = Shape<5, 50, 50, 50, 1>;      // configs[1], configs[2]
using ShapeC1 = Shape<1, 10, 10, 1>;          // configs[0]
using ShapeTR = Shape<1, 10, 10, 10, 1>;      // Examples/trainRegression.py

template <class S>
static bool shape_matches(const NetDev& nd) {
    if (nd.nl != S::NL) return false;
    for (int l = 0; l < S::NL; ++l)
        if (nd.in[l] != S::D[l] || nd.out[l] != S::D[l + 1]) return false;
    return true;
}

static inline int fast_lookup(const NetDev& nd) {
    if (shape_matches<ShapeC2>(nd)) return 0;
    if (shape_matches<ShapeC1>(nd)) return 1;
    if (shape_matches<ShapeTR>(nd)) return 2;
    return -1;
}
static inline const char* fast_name(int id) {
    switch (id) {
        case 0: return "fast<5,50,50,50,1>";
        case 1: return "fast<1,10,10,1>";
        case 2: return "fast<1,10,10,10,1>";
        default: return "fast<none>";
    }
}
// one workgroup (4 waves, 1 wave per SIMD) per CU; fewer when there are not enough tiles
static inline int fast_grid(int, long n) {
    const long ntiles = (n + 15) / 16;
    const long wgs = (ntiles + FAST_WAVES - 1) / FAST_WAVES;
    return (int)(wgs < 256 ? wgs : 256);
}
static inline int fast_launch(int id, int grid, hipStream_t st, const NetDev& nd, const float* q, const float* eta,
                              const float* X, const float* Y, long n, float* slabs, double* pstat) {
    switch (id) {
        case 0: hipLaunchKernelGGL(k_fwd_bwd_fast<ShapeC2>, dim3(grid), dim3(FAST_THREADS), 0, st, nd, q, eta, X, Y, n, slabs, pstat); break;
        case 1: hipLaunchKernelGGL(k_fwd_bwd_fast<ShapeC1>, dim3(grid), dim3(FAST_THREADS), 0, st, nd, q, eta, X, Y, n, slabs, pstat); break;
        case 2: hipLaunchKernelGGL(k_fwd_bwd_fast<ShapeTR>, dim3(grid), dim3(FAST_THREADS), 0, st, nd, q, eta, X, Y, n, slabs, pstat); break;
        default: return -1;
    }
    return 0;
}
