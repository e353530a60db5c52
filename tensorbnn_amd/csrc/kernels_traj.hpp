// Whole leapfrog trajectories of SMALL problems in one launch (round 5).
//
// configs[0] (1 -> 10 -> 10 -> 1, 1,000 rows) is launch-bound on the two-kernel step: a fused pass of 16 workgroups (~3 us) + k_update
// (~2 us) + two launch gaps = 7.3 - 9 us per leapfrog step, of which the arithmetic is a few hundred ns.  When ONE workgroup can take
// all the rows (the narrow family's tile step, weights resident in LDS) nothing has to leave the workgroup between two steps: the
// four waves' dW tiles are summed through LDS into a dense gradient, every thread finishes its parameters (prior gradient, kick,
// drift: k_update's arithmetic, update_ops.hpp) and writes the new position straight into the weight images in LDS, and the next
// step starts behind one barrier.  One launch per TRANSITION instead of 2 L; a multi-chain handle runs one workgroup per chain
// (gridDim.y), each at its own (eps, L).
//
// What stays outside: the opening half kick + drift (k_update UPD_FIRST: it builds the image this kernel starts from), k_begin,
// k_energy (decision, record, commit).  Reference: the TFP leapfrog integrator the reference drives through
// tfp.mcmc.HamiltonianMonteCarlo, network.py:315-329 / :394-408 (one_step -> L leapfrog steps).
//
// The gradient's summation order differs from the two-kernel path (one workgroup's four waves instead of a tree over 16 slabs):
// the same values to fp32 rounding, not bit for bit.  The path is a function of the handle's shape and row count (and TBNN_TRAJ=0: never;
// a traced or profiled transition takes the per-step kernels), so chain groups still equal their solo chains bit for bit.
#pragma once
#include "kernels_fast3.hpp"
#include "update_ops.hpp"

// Measured at configs[0]'s network, one chain, us per leapfrog step (tools/experiments/traj_time.py): rows 64 / 256 / 512 / 1000 / 2000:
// 16 waves 2.3 / 3.2 / 4.6 / 7.4 / 12.9, 4 waves 1.6 / 3.3 / 5.6 / 10.0 / 18.8, two-kernel step 14.4 / 9.4 / 9.0 / 8.3 / 10.0;
// 64 chains at 1000 rows: 8.2 against 11.5.  Above ~1,200 rows the row tiles want more than one workgroup: the two-kernel step.
// With 4 waves (networks whose accumulators do not fit four waves per SIMD; 6 -> 24 -> 24 -> 1: 3.2 / 6.8 / 11.6 / 21.2 us at 64 / 256 / 512 / 1000
// rows against 11.3 on the two-kernel step) the crossover is at ~500 rows.
#ifndef TBNN_TRAJ_MAX_ROWS
#define TBNN_TRAJ_MAX_ROWS 1200          // 16 waves per workgroup
#endif
#ifndef TBNN_TRAJ_MAX_ROWS_4
#define TBNN_TRAJ_MAX_ROWS_4 384         // 4 waves per workgroup
#endif
#define TRAJ_MAX_KP 4                    // parameters per thread held in registers: P <= 1024

// NW waves per workgroup (4 or 16: one or four per SIMD).  A tile of a 10-wide network is a chain of dependent LDS round trips and MFMA
// latencies (0.56 us per tile and wave, measured), not arithmetic: four waves per SIMD overlap four tiles' chains.
template <class S, int NW>
struct TrajCfg {
    using C = F3Cfg<S>;
    static constexpr int T = C::DW3_TILES > 0 ? C::DW3_TILES : 1;
    static constexpr int FPR = C::FP_REGS > 0 ? C::FP_REGS : 1;
    static constexpr int P4 = (C::P() + 3) / 4 * 4;
    static constexpr int THREADS = 64 * NW;
    static constexpr int KP = (C::P() + THREADS - 1) / THREADS;
    static constexpr int IMG_FLOATS = C::STATIC_FLOATS + NW * C::WAVE3_FLOATS;        // weight images + NW waves' activation / delta images
    static constexpr int ST_OFF = (IMG_FLOATS + 3) / 4 * 4;                           // images stay alive across the steps: staging behind them
    static constexpr int LB_OFF = ST_OFF + FAST_WAVES * T * 256;                      // (SlabOut3 / FringeOut sum FAST_WAVES = 4 staged copies)
    static constexpr int DENSE_OFF = LB_OFF + FAST_WAVES * FPR * 64;
    static constexpr int COPY_FLOATS = T * 256 + FPR * 64;                            // NW > 4: every wave's copy first, summed four by four into the 4
    static constexpr int RAW_OFF = DENSE_OFF + P4;
    static constexpr int LDS_FLOATS = RAW_OFF + (NW > FAST_WAVES ? NW * COPY_FLOATS : 0);
    // four waves per SIMD have 128 registers each: only networks whose accumulators and fringe sums are a handful (configs[0]: 2 tiles + 5)
    static constexpr bool REGS_OK = NW == 4 || (4 * T + FPR <= 40 && C::maxMT() <= 2);
    static constexpr bool OK = (NW == 4 || NW == 16) && REGS_OK && C::VL && C::DW3_TILES > 0 && C::EP3_TILES == C::DW3_TILES && KP <= TRAJ_MAX_KP &&
                               (size_t)LDS_FLOATS * 4 + 256 <= 160 * 1024;
};

template <class S, int NW>
__global__ __launch_bounds__(64 * NW, 1) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4))) void k_traj_fast3(
    NetDev nd, const float* __restrict__ qimg, long img_stride, const float* __restrict__ eta, const float* __restrict__ X, const float* __restrict__ Y, long n,
    float* __restrict__ q, float* __restrict__ p, float* __restrict__ g, float* __restrict__ gd, const int* __restrict__ imgmap,
    double* __restrict__ pstat, int nstat, float eps, int L, const StepCtl* __restrict__ ctl)
{
    using C = F3Cfg<S>;
    using TC = TrajCfg<S, NW>;
    constexpr int THREADS = TC::THREADS;
    static_assert(TC::OK, "shape not eligible for the trajectory kernel");
    constexpr int d_in = C::in(0), d_out = C::out(C::NL - 1), LL = C::NL - 1, P = C::P(), KP = TC::KP;
    constexpr int NFd = C::maxNF() > 0 ? C::maxNF() : 1;
    __shared__ __attribute__((aligned(16))) float lds[TC::LDS_FLOATS];
    __shared__ double red[NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, gq = lane >> 4;
    {   // this workgroup's chain
        const size_t c = blockIdx.y, cp = c * (size_t)P;
        qimg += c * (size_t)img_stride; eta += c * nd.H; q += cp; p += cp; g += cp; gd += cp; pstat += c * (size_t)PSTAT_CAP;
        if (ctl) { const StepCtl me = ctl[c]; eps = me.eps; L = me.L; }
    }
    float* wl = lds + C::STATIC_FLOATS + wave * C::WAVE3_FLOATS;
    const long ntiles = (n + 15) / 16;
    // ---- the weight images (as k_update left them after the opening drift), zeroed per-wave images, the ones slots
    {
        constexpr int N4 = C::STATIC_FLOATS / 4;
        static_assert(C::STATIC_FLOATS % 4 == 0, "16-B pieces");
        const float4* src = reinterpret_cast<const float4*>(qimg);
        float4* dst = reinterpret_cast<float4*>(lds);
        for (int e = tid; e < N4; e += THREADS) dst[e] = src[e];
        float4* z = reinterpret_cast<float4*>(wl);
        for (int e = lane; e < C::WAVE3_FLOATS / 4; e += 64) z[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // this thread's parameters: position, momentum, prior, image slots -- in registers for the whole trajectory
    float qj[KP], pj[KP], loc[KP], scale[KP];
    int prior[KP], m0[KP], m1[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        const int j = tid + k * THREADS;
        qj[k] = pj[k] = loc[k] = 0.f; scale[k] = 1.f; prior[k] = 0; m0[k] = m1[k] = -1;
        if (j < P) {
            prior_params(nd, eta, j, prior[k], loc[k], scale[k]);
            qj[k] = q[j]; pj[k] = p[j];
            m0[k] = imgmap[j]; m1[k] = imgmap[P + j];
        }
    }
    const float sigma = lik_sigma(nd, eta);
    const float inv_var = 1.f / (sigma * sigma);
    const float sg = nd.lik == TBNN_LIK_GAUSSIAN ? sigma : 1.f;
    __syncthreads();
    if (gq == 0) wl[C::aoff3(0) + d_in * C::PR + i16] = 1.f;
#pragma unroll
    for (int l = 1; l < C::NLM3; ++l)
        if (C::in(l) % 16 == 0 && gq == 0) wl[C::aoff3(l) + C::in(l) * C::PR + i16] = 1.f;

    f32x4* const stage = reinterpret_cast<f32x4*>(lds + TC::ST_OFF);
    float* const lb = lds + TC::LB_OFF;
    float* const dense = lds + TC::DENSE_OFF;
    double stat = 0.0;
#pragma unroll 1
    for (int t = 1; t <= L; ++t) {
        f32x4 dW[TC::T];
#pragma unroll
        for (int k = 0; k < C::DW3_TILES; ++k) dW[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        float FP[C::FP_REGS > 0 ? C::FP_REGS : 1];
#pragma unroll
        for (int k = 0; k < C::FP_REGS; ++k) FP[k] = 0.f;
        stat = 0.0;
        // layer 0's operands (A, bias, fringe weights): the same for every tile of this step
#if TBNN_F3_HAND
        typename Fwd3<S, 0>::Pre P0;
        Fwd3<S, 0>::pre_all(P0, lds, i16, gq);
#else
        f32x4 A0[C::MTF(0) > 0 ? C::MTF(0) : 1], B0[C::MTF(0) > 0 ? C::MTF(0) : 1];
        Fwd3<S, 0>::preload(A0, B0, lds, i16, gq);
#endif
#pragma unroll 1
        for (long tile = wave; tile < ntiles; tile += NW) {
            Tile3<S> T;
            float y[d_out];
            const long row = tile * 16 + i16;
            const bool rvalid = row < n;
#pragma unroll
            for (int k = 0; k < C::KS0; ++k) {
                const int u = 4 * k + gq;
                const float xv = (rvalid && u < d_in) ? X[row * d_in + u] : 0.f;
                T.x0[k] = xv;
                if (u < d_in) wl[C::aoff3(0) + u * C::PR + i16] = xv;
            }
#pragma unroll
            for (int o = 0; o < d_out; ++o) y[o] = rvalid ? Y[row * d_out + o] : 0.f;
#if TBNN_F3_HAND
            Fwd3<S, 0>::run_h(T, lds, wl, i16, gq, P0);
#else
            Fwd3<S, 0>::run(T, lds, wl, i16, gq, A0, B0);
#endif
            float dzf[NFd];
#pragma unroll
            for (int o = 0; o < NFd; ++o) dzf[o] = 0.f;
#pragma unroll
            for (int o = 0; o < d_out; ++o) dzf[o] = rvalid ? lik_delta<S>(T.af[LL][o], y[o], inv_var, gq == 0, stat) : 0.f;
            if constexpr (!C::NCF(LL > 0 ? LL - 1 : 0)) FringeDW<S, LL>::run(FP, T, dzf, gq);
            f32x4 dzL[C::MT(LL)];
            f32x4 dzp[C::MT(LL - 1)];
            float dzpf[NFd];
            Bwd3<S, LL>::da(T, lds, i16, gq, dzL, dzf, dzp, dzpf);
            {
                constexpr int LM = LL - 1;
                float Aop[Bwd3<S, LM>::MTd][4], Bop[C::NT(LM)][4], Fop[8];
                typename Bwd3<S, LM>::NFOps NFop;
                Bwd3<S, LM>::issue(dzp, dzpf, wl, i16, gq, Aop, Bop, Fop);
                Bwd3<S, LM>::nf_load(NFop, wl, lane);
                if constexpr (C::NCF(LM)) FringeDW<S, LL>::run(FP, T, dzf, gq);
                FringeDW<S, LM>::run(FP, T, dzpf, gq);
                Bwd3<S, LM>::nf_mfma(dW, NFop);
                if constexpr (LM > 0) {
                    f32x4 dzq[C::MT(LM - 1)];
                    float dzqf[NFd];
                    Bwd3<S, LM>::da(T, lds, i16, gq, dzp, dzpf, dzq, dzqf);
                    SCHED_FENCE();
                    Pipe3<S, LM - 1>::run(dW, FP, T, lds, wl, i16, gq, dzq, dzqf, Aop, Bop, Fop);
                } else {
                    Bwd3<S, 0>::dw(dW, Aop, Bop);
                    Bwd3<S, 0>::fdw(FP, Fop, Bop);
                }
            }
        }
        mfma_drain_acc(dW);
        // ---- the waves' tiles and fringe partials -> the dense gradient (data term) in LDS, fixed order (SlabOut3 / FringeOut over four
        // staged copies; sixteen waves: copy v = (c[4v] + c[4v+1]) + (c[4v+2] + c[4v+3]) first)
        if constexpr (NW == FAST_WAVES) {
            f32x4* mine = stage + wave * (C::EP3_TILES * 64);
#pragma unroll
            for (int k = 0; k < C::DW3_TILES; ++k) mine[k * 64 + lane] = dW[k];
#pragma unroll
            for (int r = 0; r < C::FP_REGS; ++r) lb[((size_t)(wave * C::FP_REGS + r) * 4 + gq) * 16 + i16] = FP[r];
            __syncthreads();
        } else {
            float* raw = lds + TC::RAW_OFF + wave * TC::COPY_FLOATS;              // [tile][lane] x 16 B, then [reg][lane]
#pragma unroll
            for (int k = 0; k < C::DW3_TILES; ++k) *reinterpret_cast<f32x4*>(raw + (k * 64 + lane) * 4) = dW[k];
#pragma unroll
            for (int r = 0; r < C::FP_REGS; ++r) raw[TC::T * 256 + r * 64 + lane] = FP[r];
            __syncthreads();
            const float* rawb = lds + TC::RAW_OFF;
            for (int e = tid; e < FAST_WAVES * TC::COPY_FLOATS; e += THREADS) {
                const int v = e / TC::COPY_FLOATS, x = e - v * TC::COPY_FLOATS;
                const float* c = rawb + (size_t)(4 * v) * TC::COPY_FLOATS + x;
                const float sum = (c[0] + c[TC::COPY_FLOATS]) + (c[2 * TC::COPY_FLOATS] + c[3 * TC::COPY_FLOATS]);
                if (x < TC::T * 256) reinterpret_cast<float*>(stage)[(size_t)v * (C::EP3_TILES * 256) + x] = sum;
                else {                                                             // lb[((v * FP_REGS + r) * 4 + g) * 16 + i16]: lane = 16 g + i16
                    const int y = x - TC::T * 256;
                    lb[(size_t)v * C::FP_REGS * 64 + y] = sum;
                }
            }
            __syncthreads();
        }
        if (wave < FAST_WAVES) SlabOut3<S, 0>::template run<true>(reinterpret_cast<const float*>(stage), dense, wave, lane, 0, C::DW3_TILES);
        if (tid < FAST_THREADS) FringeOut<S, 0>::template run<true>(lb, dense, tid);
        __syncthreads();
        // ---- every thread finishes its parameters: k_update's UPD_MID / UPD_LAST (update_ops.hpp: upd_finish), the new position into
        // the images in LDS
        const bool last = t == L;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const int j = tid + k * THREADS;
            if (j < P) {
                float gj = dense[j];
                if (last && gd) gd[j] = gj * (sg * sg);
                gj += prior_grad(prior[k], loc[k], scale[k], qj[k]);
                float pn = pj[k] + eps * gj;                              // full kick
                if (!last) {
                    const float qn = qj[k] + eps * pn;                    // drift
                    pj[k] = pn; qj[k] = qn;
                    lds[m0[k]] = qn;
                    if (m1[k] >= 0) lds[m1[k]] = qn;
                } else {
                    g[j] = gj;
                    pn = pn - 0.5f * eps * gj;                            // undo half kick
                    pj[k] = pn;
                    p[j] = pn; q[j] = qj[k];
                }
            }
        }
        __syncthreads();
    }
    // the statistic of the LAST evaluation (the proposal's log-likelihood term): k_energy sums nstat entries
    const double wtot = wave_sum_lane0(stat);
    if (lane == 0) red[wave] = wtot;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int w = 0; w < NW; ++w) s += red[w];
        pstat[0] = s;
    }
    for (int e = 1 + tid; e < nstat; e += THREADS) pstat[e] = 0.0;
}

#ifndef TBNN_TRAJ_WAVES
#define TBNN_TRAJ_WAVES 16
#endif
#ifndef TBNN_NO_FAST_REGISTRY
// ahead-of-time instantiations: the registry ids of kernels_fast3.hpp whose shapes are eligible; 16 waves where they fit
template <class S> struct TrajPick { static constexpr int NW = (TBNN_TRAJ_WAVES == 16 && TrajCfg<S, 16>::OK) ? 16 : 4; };
template <class S> static inline long fast3_traj_rows_t() {
    return !TrajCfg<S, TrajPick<S>::NW>::OK ? 0 : (TrajPick<S>::NW == 16 ? TBNN_TRAJ_MAX_ROWS : TBNN_TRAJ_MAX_ROWS_4);
}
// row count up to which registry shape `id` runs its transitions on the trajectory kernel (0: never)
static inline long fast3_traj_max_rows(int id) { return id == 1 ? fast3_traj_rows_t<ShapeC1>() : id == 2 ? fast3_traj_rows_t<ShapeTR>() : 0; }
template <class S>
static inline void fast3_traj_launch_t(int nchains, hipStream_t st, const NetDev& nd, const float* qimg, long img_stride, const float* eta,
                                       const float* X, const float* Y, long n, float* q, float* p, float* g, float* gd, const int* imgmap,
                                       double* pstat, int nstat, float eps, int L, const StepCtl* ctl) {
    constexpr int NW = TrajPick<S>::NW;
    hipLaunchKernelGGL((k_traj_fast3<S, NW>), dim3(1, nchains), dim3(64 * NW), 0, st, nd, qimg, img_stride, eta, X, Y, n, q, p, g, gd, imgmap, pstat, nstat, eps, L, ctl);
}
static inline int fast3_traj_launch(int id, int nchains, hipStream_t st, const NetDev& nd, const float* qimg, long img_stride, const float* eta,
                                    const float* X, const float* Y, long n, float* q, float* p, float* g, float* gd, const int* imgmap,
                                    double* pstat, int nstat, float eps, int L, const StepCtl* ctl) {
    switch (id) {
        case 1: fast3_traj_launch_t<ShapeC1>(nchains, st, nd, qimg, img_stride, eta, X, Y, n, q, p, g, gd, imgmap, pstat, nstat, eps, L, ctl); break;
        case 2: fast3_traj_launch_t<ShapeTR>(nchains, st, nd, qimg, img_stride, eta, X, Y, n, q, p, g, gd, imgmap, pstat, nstat, eps, L, ctl); break;
        default: return -1;
    }
    return 0;
}
#endif
