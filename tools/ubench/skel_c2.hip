// microbenchmark: the MFMA stream of k_fwd_bwd_fast3 at configs[1] with every operand in registers -- per 16-row tile
// 270 v_mfma_f32_16x16x4_f32 + 67 v_mfma_f32_4x4x1_16b_f32 + 11 lane-sum 16x16x4, in the kernel's dependency pattern (forward / delta
// chain: 3 rotating accumulators + 1 fringe accumulator, k-step major; dW: 27 AccVGPR tiles visited once per k-step) -- on the kernel's
// geometry: 256 workgroups x 4 waves, one wave per SIMD, 6 (7, 1) tiles per wave, random operands.
// What it answers: how long the tile loop of configs[1] takes when nothing but its MFMAs is issued, at the clock the chip holds.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void m16v(f32x4& c, float a, float b) { asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }
__device__ __forceinline__ void m16a(f32x4& c, float a, float b) { asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b)); }
__device__ __forceinline__ void m4v(f32x4& c, float a, float b) { asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }

// one chain layer: KS k-steps x (3 full tiles + the fringe 4x4x1), then NS lane-sum MFMAs that depend on the fringe accumulator
template <int KS, int NS>
__device__ __forceinline__ void chain_layer(f32x4 (&acc)[3], f32x4& fr, f32x4& gs, const float (&w)[8], const float (&x)[8]) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        m16v(acc[0], w[s & 7], x[(s + 1) & 7]);
        m16v(acc[1], w[(s + 2) & 7], x[(s + 1) & 7]);
        m16v(acc[2], w[(s + 4) & 7], x[(s + 1) & 7]);
        m4v(fr, w[(s + 5) & 7], x[(s + 1) & 7]);
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) m16v(gs, 1.f, fr[k & 3]);
}

__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_skel(const float* __restrict__ in, float* out, int tiles2,
                                                                                          unsigned long long* stamps) {
    const int lane = threadIdx.x & 63;
    float w[8], x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { w[i] = in[(blockIdx.x * 256 + threadIdx.x) * 16 + i]; x[i] = in[(blockIdx.x * 256 + threadIdx.x) * 16 + 8 + i]; }
    f32x4 dW[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) dW[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 acc[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, fr = {0.f, 0.f, 0.f, 0.f}, gs = {0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = clock64(), r0 = wall_clock64();
    for (int t = 0; t < tiles2; ++t) {
        // forward: layer 0 (2 k-steps), layers 1, 2 (13 k-steps), all-fringe last layer (13 4x4x1 + 1 lane sum)
        chain_layer<2, 2>(acc, fr, gs, w, x);
        chain_layer<13, 2>(acc, fr, gs, w, x);
        chain_layer<13, 2>(acc, fr, gs, w, x);
#pragma unroll
        for (int s = 0; s < 13; ++s) m4v(fr, w[s & 7], x[(s + 3) & 7]);
        m16v(gs, 1.f, fr[0]);
        // delta chain: two 13-k-step layers (W^T), lane sums of their fringe units
        chain_layer<13, 2>(acc, fr, gs, w, x);
        chain_layer<13, 2>(acc, fr, gs, w, x);
        // dW: layers 2 and 1 (12 tiles x 4 k-steps each), layer 0 (3 tiles x 4 k-steps)
#pragma unroll
        for (int l = 0; l < 2; ++l)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int tt = 0; tt < 12; ++tt) m16a(dW[3 + 12 * l + tt], w[(s + tt) & 7], x[(tt + l) & 7]);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int tt = 0; tt < 3; ++tt) m16a(dW[tt], w[(s + tt) & 7], x[tt & 7]);
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    const unsigned long long t1 = clock64(), r1 = wall_clock64();
    float s = gs[0] + fr[1] + acc[0][0] + acc[1][1] + acc[2][2];
#pragma unroll
    for (int t = 0; t < 27; ++t) { asm volatile("" : "+a"(dW[t])); s += dW[t][lane & 3]; }
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
    const int G = 256;
    std::vector<float> h((size_t)G * 256 * 16);
    unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) * (1.f / 16777216.f) - 0.5f) * 0.25f; }
    float *in, *out; unsigned long long* st;
    (void)hipMalloc(&in, h.size() * 4); (void)hipMalloc(&out, (size_t)G * 256 * 4); (void)hipMalloc(&st, (size_t)G * 16);
    (void)hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int tiles2 : {6, 7, 1}) {          // 6 tiles = the rounds of configs[1]'s tile loop; 7; 1
        for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k_skel, dim3(G), dim3(256), 0, 0, in, out, tiles2, st);      // settle the clock under load
        (void)hipDeviceSynchronize();
        const int N = 2000;
        (void)hipEventRecord(a, 0);
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_skel, dim3(G), dim3(256), 0, 0, in, out, tiles2, st);
        (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
        float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
        std::vector<unsigned long long> hs((size_t)G * 2);
        (void)hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
        double cyc = 0, wall = 0; for (int g = 0; g < G; ++g) { cyc += hs[2 * g]; wall += hs[2 * g + 1]; }
        cyc /= G; wall /= G;                                        // wall clock: 100 MHz ticks
        const double tiles = tiles2, mf = tiles * (281 * 32 + 67 * 8);
        printf("MFMA-only stream, %4.1f tiles per wave: %7.0f shader cycles in the loop (%.0f per tile; MFMA issue time %.0f = %.3f), in-kernel clock %.2f GHz, "
               "loop %.2f us, back-to-back launch %.2f us\n", tiles, cyc, cyc / tiles, mf / tiles, mf / cyc, cyc / (wall * 10.0),
               wall * 0.01, ms * 1000.0 / N);
    }
    return 0;
}
