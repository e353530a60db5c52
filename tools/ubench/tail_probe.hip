// microbenchmark: where the in-kernel leapfrog update (update_ops.hpp: tail_update) spends its time.  256 workgroups (one
// per CU: 140 KB of LDS each) write a configs[1]-sized slab, then run the tail; every workgroup stamps the 100-MHz wall
// clock at: 0 slab written, 1 ticket taken (release), 2 all arrived (reducers), 3 acquire done, 4 partial sums staged,
// 5 finished, 6 last barrier.  Build: hipcc --offload-arch=gfx950 -O3 -DTAIL_STAMPS -I tensorbnn_amd/csrc tools/ubench/tail_probe.hip
#define TAIL_STAMPS 1
#include "update_ops.hpp"

// the same sums from slabs OTHER workgroups of the running kernel wrote (tail_update): device-coherent loads (sc1: served
// at the point the XCDs share, never from this XCD's L2) of slabs stored the same way -- no cache write-back / invalidate
typedef unsigned int tb_u32x4 __attribute__((ext_vector_type(4)));
#define TB_SC1 16      // cache-policy bit of the buffer builtins: agent scope
__device__ __forceinline__ float4 upd_column_partial_coh(__amdgpu_buffer_rsrc_t rs, int nslab, int pitch, int c4, int ty) {
    float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 s0 = z, s1 = z, s2 = z, s3 = z;
    if (c4 * 4 < pitch) {
        const int p16 = pitch * 4;                          // bytes per slab
        int w = ty;
        auto ld = [&](int ww) {
            const tb_u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, ww * p16 + c4 * 16, 0, TB_SC1);
            return make_float4(__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[2]), __uint_as_float(r[3]));
        };
        for (; w + 7 * UPD_GROUPS < nslab; w += 8 * UPD_GROUPS) {
            float4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = ld(w + k * UPD_GROUPS);
#pragma unroll
            for (int k = 0; k < 8; k += 4) {
                s0.x += v[k].x; s0.y += v[k].y; s0.z += v[k].z; s0.w += v[k].w;
                s1.x += v[k + 1].x; s1.y += v[k + 1].y; s1.z += v[k + 1].z; s1.w += v[k + 1].w;
                s2.x += v[k + 2].x; s2.y += v[k + 2].y; s2.z += v[k + 2].z; s2.w += v[k + 2].w;
                s3.x += v[k + 3].x; s3.y += v[k + 3].y; s3.z += v[k + 3].z; s3.w += v[k + 3].w;
            }
        }
        for (; w < nslab; w += UPD_GROUPS) {
            const float4 a = ld(w);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
        }
    }
    float4 s;
    s.x = (s0.x + s1.x) + (s2.x + s3.x); s.y = (s0.y + s1.y) + (s2.y + s3.y);
    s.z = (s0.z + s1.z) + (s2.z + s3.z); s.w = (s0.w + s1.w) + (s2.w + s3.w);
    return s;
}

// ---- the UPD_MID update as the TAIL of the fused kernel (narrow family; no k_update launch between two leapfrog steps) ----
// Every workgroup, its slab written, takes a ticket; the first G - R to finish leave, the last R stay as reducers: they wait
// until all G tickets are taken (bounded spin: only workgroups that are already running are waited for, and R < the number
// of CUs so that co-tenant kernels always find free CUs), then run k_update's blocks r, r + R, ... -- the same columns,
// the same fixed summation order, the same finishing arithmetic, so the result is bit-identical to the separate launch.
// ctr[0] arrivals, ctr[1] departures (the last reducer to leave zeroes both for the next launch), ctr[2] sticky error flag
// (a reducer gave up after TAIL_SPIN_TICKS of the 100-MHz wall clock: the transition is then invalid and the host says so).
struct TailUpd {
    unsigned* ctr = nullptr;   // null: no tail (the caller launches k_update)
    int R = 0;                 // reducers
    float eps = 0.f;
    float* q = nullptr; float* p = nullptr; float* g = nullptr; float* gd = nullptr;
    const int* imgmap = nullptr; float* qimg = nullptr;
#ifdef TAIL_STAMPS
    unsigned long long* stamps = nullptr;   // [gridDim.x][8] wall-clock stamps (tools/ubench/tail_probe.hip)
#endif
};
#ifdef TAIL_STAMPS
#define TAIL_STAMP(i) do { if (tu.stamps && threadIdx.x == 0) tu.stamps[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define TAIL_STAMP(i) do { } while (0)
#endif
#define TAIL_ROUNDS 3
#define TAIL_SPIN_TICKS 200000000ull     // 2 s
// smem: >= TAIL_ROUNDS * UPD_GROUPS * UPD_COLS float4 (12 KB), free for reuse; blockDim.x == 256
// slabs: the kernel's own slab pointer (gridDim.x slabs of `pitch` floats)
__device__ __forceinline__ void tail_update(const NetDev& nd, const TailUpd& tu, const float* __restrict__ eta, const float* slabs, int pitch, float* smem) {
    __shared__ unsigned s_ticket;
    const unsigned G = gridDim.x;
    const unsigned R = (unsigned)tu.R < G ? (unsigned)tu.R : G;
    TAIL_STAMP(0);
    __syncthreads();                                            // this workgroup's slab stores are issued and counted
#ifndef TAIL_SC1
#define TAIL_SC1 1
#endif
    if (threadIdx.x == 0)
        s_ticket = __hip_atomic_fetch_add(tu.ctr, 1u, TAIL_SC1 ? __ATOMIC_RELAXED : __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    TAIL_STAMP(1);
    const unsigned t = s_ticket;
#ifdef TAIL_STAMPS
    if (tu.stamps && threadIdx.x == 0) tu.stamps[blockIdx.x * 8 + 7] = t;
#endif
    if (t < G - R) return;
    const int r = (int)(t - (G - R));
    const int tx = threadIdx.x & (UPD_COLS - 1), ty = threadIdx.x / UPD_COLS;
    const int NB = (pitch / 4 + UPD_COLS - 1) / UPD_COLS;
    // the first round's finishing operands do not depend on the other workgroups: fetched while waiting
    UpdPre u[TAIL_ROUNDS];
#pragma unroll
    for (int k = 0; k < TAIL_ROUNDS; ++k) {
        const int jf = ((r + k * (int)R) * UPD_COLS + tx) * 4 + ty;
        if (r + k * (int)R < NB && ty < 4 && jf < nd.P) upd_prefetch(u[k], nd, UPD_MID, eta, jf, nullptr, nullptr, tu.q, tu.p, tu.imgmap);
    }
    if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(tu.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < G) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > TAIL_SPIN_TICKS) { __hip_atomic_store(tu.ctr + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
    }
    TAIL_STAMP(2);
    __syncthreads();
#if !TAIL_SC1
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // every wave: no stale line of another XCD's slabs
#endif
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(slabs), 0, (int)(G * (unsigned)pitch * 4u), 0x00020000);
    float4* part = reinterpret_cast<float4*>(smem);             // [round][group][col]
    TAIL_STAMP(3);
    for (int b0 = r; b0 < NB; b0 += TAIL_ROUNDS * (int)R) {
        if (b0 != r) {
#pragma unroll
            for (int k = 0; k < TAIL_ROUNDS; ++k) {
                const int b = b0 + k * (int)R;
                const int jf = (b * UPD_COLS + tx) * 4 + ty;
                u[k] = UpdPre();
                if (b < NB && ty < 4 && jf < nd.P) upd_prefetch(u[k], nd, UPD_MID, eta, jf, nullptr, nullptr, tu.q, tu.p, tu.imgmap);
            }
            __syncthreads();                                    // the previous pass is through with `part`
        }
#pragma unroll
        for (int k = 0; k < TAIL_ROUNDS; ++k) {
            const int b = b0 + k * (int)R;
            if (b < NB) part[(k * UPD_GROUPS + ty) * UPD_COLS + tx] = TAIL_SC1 ? upd_column_partial_coh(rs, (int)G, pitch, b * UPD_COLS + tx, ty)
                                                                         : upd_column_partial(slabs, (int)G, pitch, b * UPD_COLS + tx, ty);
        }
        __syncthreads();
        TAIL_STAMP(4);
#pragma unroll
        for (int h = UPD_GROUPS / 2; h > 0; h >>= 1) {
            if (ty < h) {
#pragma unroll
                for (int k = 0; k < TAIL_ROUNDS; ++k) {
                    if (b0 + k * (int)R < NB) {
                        float4* pp = part + (k * UPD_GROUPS + ty) * UPD_COLS + tx;
                        const float4 a = pp[0], b = pp[h * UPD_COLS];
                        pp[0] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
                    }
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int k = 0; k < TAIL_ROUNDS; ++k) {
            const int b = b0 + k * (int)R;
            const int jf = (b * UPD_COLS + tx) * 4 + ty;
            if (b < NB && ty < 4 && jf < nd.P) {
                const float4 gs = part[(k * UPD_GROUPS) * UPD_COLS + tx];
                const float gj = ty == 0 ? gs.x : ty == 1 ? gs.y : ty == 2 ? gs.z : gs.w;
                upd_finish(u[k], nd, UPD_MID, tu.eps, eta, jf, gj, tu.q, tu.p, tu.g, tu.imgmap, tu.qimg, tu.gd);
            }
        }
    }
    TAIL_STAMP(5);
    __syncthreads();
    TAIL_STAMP(6);
    if (threadIdx.x == 0) {
        const unsigned d = __hip_atomic_fetch_add(tu.ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (d == R - 1) {                                       // every reducer is past its wait: re-arm for the next launch
            __hip_atomic_store(tu.ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(tu.ctr + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}


#include <cstdio>
#include <cmath>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(256, 1) void k(NetDev nd, const float* __restrict__ eta, float* __restrict__ slabs, int pitch, TailUpd tu, int spin, int launch) {
    extern __shared__ float lds[];
    // uneven finish times: workgroup b idles b % 8 * spin clocks
    for (int i = 0; i < (int)(blockIdx.x % 8) * spin; ++i) __builtin_amdgcn_s_sleep(8);
    float* slab = slabs + (size_t)blockIdx.x * pitch;
#if TAIL_SC1
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slab, 0, pitch * 4, 0x00020000);
    for (int j = threadIdx.x; j < pitch; j += 256) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(1e-3f * (float)((j + blockIdx.x + launch) % 97)), rs, j * 4, 0, TB_SC1);
#else
    for (int j = threadIdx.x; j < pitch; j += 256) slab[j] = 1e-3f * (float)((j + blockIdx.x + launch) % 97);
#endif
    tail_update(nd, tu, eta, slabs, pitch, lds);
}

int main() {
    const int dims[5] = {5, 50, 50, 50, 1};
    NetDev nd{};
    nd.nl = 4;
    int off = 0;
    for (int l = 0; l < 4; ++l) {
        nd.in[l] = dims[l]; nd.out[l] = dims[l + 1]; nd.act[l] = l < 3 ? TBNN_ACT_RELU : TBNN_ACT_NONE; nd.prior[l] = TBNN_PRIOR_CAUCHY;
        nd.offW[l] = off; off += dims[l] * dims[l + 1]; nd.offB[l] = off; off += dims[l + 1];
    }
    nd.P = off; nd.H = 17; nd.lik = TBNN_LIK_GAUSSIAN; nd.d_in = 5; nd.d_out = 1;
    const int P = nd.P, pitch = (P + 3) / 4 * 4, G = 256;
    float *eta, *slabs, *q, *p, *g, *gd, *qimg; int* imgmap; unsigned* ctr; unsigned long long* st;
    (void)hipMalloc(&eta, 64 * 4); (void)hipMalloc(&slabs, (size_t)G * pitch * 4);
    (void)hipMalloc(&q, P * 4); (void)hipMalloc(&p, P * 4); (void)hipMalloc(&g, P * 4); (void)hipMalloc(&gd, P * 4);
    (void)hipMalloc(&qimg, 2 * P * 4); (void)hipMalloc(&imgmap, 2 * P * 4); (void)hipMalloc(&ctr, 16); (void)hipMalloc(&st, G * 8 * 8);
    std::vector<float> he(64, 0.5f); (void)hipMemcpy(eta, he.data(), 64 * 4, hipMemcpyHostToDevice);
    std::vector<int> hm(2 * P); for (int j = 0; j < 2 * P; ++j) hm[j] = j; (void)hipMemcpy(imgmap, hm.data(), 2 * P * 4, hipMemcpyHostToDevice);
    (void)hipMemset(q, 0, P * 4); (void)hipMemset(p, 0, P * 4); (void)hipMemset(ctr, 0, 16);
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    for (int R : {64, 96, 171, 32}) for (int spin : {0, 4}) {
        TailUpd tu; tu.ctr = ctr; tu.R = R; tu.eps = 1e-4f; tu.q = q; tu.p = p; tu.g = g; tu.gd = gd; tu.imgmap = imgmap; tu.qimg = qimg; tu.stamps = st;
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        float ms = 0;
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipMemset(st, 0, G * 8 * 8); (void)hipMemset(p, 0, P * 4); (void)hipMemset(q, 0, P * 4);
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(k, dim3(G), dim3(256), 140 * 1024, 0, nd, eta, slabs, pitch, tu, spin, rep);
            (void)hipEventRecord(b);
            (void)hipDeviceSynchronize();
            (void)hipEventElapsedTime(&ms, a, b);
        }
        std::vector<unsigned long long> h(G * 8); (void)hipMemcpy(h.data(), st, G * 8 * 8, hipMemcpyDeviceToHost);
        {   // the gradient the reducers saw, read back from the momentum (p = 0 + eps (sum of this launch's slabs + prior gradient at q = 0;
            // Cauchy prior, loc 0.5, scale 0.25: -3.2); UPD_MID does not store the gradient itself)
            std::vector<float> hp(P); (void)hipMemcpy(hp.data(), p, P * 4, hipMemcpyDeviceToHost);
            (void)hipMemset(p, 0, P * 4); (void)hipMemset(q, 0, P * 4);
            int bad = 0;
            for (int j = 0; j < P; ++j) {
                double e = 0; for (int b2 = 0; b2 < G; ++b2) e += 1e-3 * (double)((j + b2 + 3) % 97);
                if (fabs(hp[j] / 1e-4 - (e - 3.2)) > 1e-3 * fabs(e) + 1e-3) ++bad;
            }
            printf("  stale/wrong columns: %d of %d\n", bad, P);
        }
        unsigned hc[4]; (void)hipMemcpy(hc, ctr, 16, hipMemcpyDeviceToHost);
        unsigned long long t0min = ~0ull, t0max = 0, t1max = 0, t2max = 0, t3max = 0, t4max = 0, t5max = 0, t6max = 0;
        double rel = 0; int nred = 0;
        for (int b2 = 0; b2 < G; ++b2) {
            const unsigned long long* s = &h[b2 * 8];
            t0min = std::min(t0min, s[0]); t0max = std::max(t0max, s[0]); t1max = std::max(t1max, s[1]);
            rel += (double)(s[1] - s[0]);
            if (s[2]) { ++nred; t2max = std::max(t2max, s[2]); t3max = std::max(t3max, s[3]); t4max = std::max(t4max, s[4]); t5max = std::max(t5max, s[5]); t6max = std::max(t6max, s[6]); }
        }
        auto us = [&](unsigned long long t) { return (double)(t - t0max) * 0.01; };
        printf("R %3d spin %d: launch %.2f us | reducers %d | slab-written spread %.2f us | ticket (release) mean %.2f us | after the LAST slab: "
               "last ticket %.2f, all-arrived seen %.2f, acquired %.2f, staged %.2f, finished %.2f, out %.2f us | ctr %u %u err %u\n",
               R, spin, ms * 1000.f, nred, (double)(t0max - t0min) * 0.01, rel / G * 0.01, us(t1max), us(t2max), us(t3max), us(t4max), us(t5max), us(t6max),
               hc[0], hc[1], hc[2]);
    }
    return 0;
}
