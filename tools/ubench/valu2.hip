// microbenchmark v2: thread-per-row 50x50 layer on the VALU, OB outputs in flight, weights pre-interleaved
// [chunk][k][OB] so one scalar load feeds OB independent accumulators; MFMA partner waves optional.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define D 50
template <int OB>
__device__ __forceinline__ void valu_layer(const float* __restrict__ W, float (&a)[D], float* ldsrow, int lane) {
    constexpr int NCH = (D + OB - 1) / OB;
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
        const float* wc = W + c * (D * OB);
        float z[OB];
#pragma unroll
        for (int o = 0; o < OB; ++o) z[o] = 0.f;
#pragma unroll
        for (int k = 0; k < D; ++k)
#pragma unroll
            for (int o = 0; o < OB; ++o) z[o] = fmaf(wc[k * OB + o], a[k], z[o]);
#pragma unroll
        for (int o = 0; o < OB; ++o)
            if (c * OB + o < D) ldsrow[(c * OB + o) * 65 + lane] = fmaxf(z[o], 0.f);
    }
#pragma unroll
    for (int k = 0; k < D; ++k) a[k] = ldsrow[k * 65 + lane];
}
template <int OB>
__global__ __launch_bounds__(512, 2) void k(const float* __restrict__ W, float* out, int mode, int reps, unsigned long long* cyc) {
    __shared__ float lds[4][D * 65];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long t0 = clock64();
    if (wave < 4) {
        if (mode & 1) {
            float a[D];
#pragma unroll
            for (int k = 0; k < D; ++k) a[k] = 0.01f * (k + lane);
            for (int r = 0; r < reps; ++r) valu_layer<OB>(W, a, lds[wave], lane);
            float s = 0; for (int k = 0; k < D; ++k) s += a[k];
            out[blockIdx.x * 512 + threadIdx.x] = s;
        }
    } else if (mode & 2) {
        f32x4 acc[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = f32x4{0, 0, 0, 0};
        float av = lane * 0.001f, bv = lane * 0.002f;
        for (int r = 0; r < reps; ++r) {
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int t = 0; t < 16; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[t], 0, 0, 0);
        }
        float s = 0; for (int t = 0; t < 16; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    }
    unsigned long long t1 = clock64();
    if (blockIdx.x == 0 && lane == 0) cyc[wave] = t1 - t0;
}
template <int OB> void run(const float* W, float* out, unsigned long long* cyc) {
    const int reps = 200;
    for (int mode = 1; mode <= 3; mode += 2) {
        for (int it = 0; it < 2; ++it) {
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k<OB>, dim3(256), dim3(512), 0, 0, W, out, mode, reps, cyc);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            unsigned long long hc[8]; (void)hipMemcpy(hc, cyc, 64, hipMemcpyDeviceToHost);
            if (it == 1) printf("OB=%d mode %d: %.1f us; per layer: valu wave %.0f cycles (2500 FMA; ideal 5000), mfma wave %.0f cycles (ideal 8192)\n",
                                OB, mode, ms * 1000, (double)hc[0] / reps, (double)hc[4] / reps);
        }
    }
}
int main() {
    float *W, *out; unsigned long long* cyc;
    (void)hipMalloc(&W, 64 * 64 * 4); (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 64);
    std::vector<float> hw(64 * 64, 0.01f); (void)hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    run<2>(W, out, cyc); run<4>(W, out, cyc); run<8>(W, out, cyc); run<10>(W, out, cyc); run<16>(W, out, cyc);
    return 0;
}
