// microbenchmark: VALU thread-per-row dense layer (weights via scalar loads) vs MFMA waves on the same SIMDs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define D 50
// one 50x50 layer per thread, REPS times; W row-major [i][k] read through the scalar cache
__device__ __forceinline__ void valu_layer(const float* __restrict__ W, float (&a)[D], float* ldsrow, int lane) {
    for (int i = 0; i < D; ++i) {
        const float* wr = W + i * 52;      // padded row pitch 52 (16-B aligned rows)
        float z0 = 0.f, z1 = 0.f;
#pragma unroll
        for (int k = 0; k < D; k += 2) { z0 = fmaf(wr[k], a[k], z0); z1 = fmaf(wr[k + 1], a[k + 1], z1); }
        ldsrow[i * 64 + lane] = fmaxf(z0 + z1, 0.f);
    }
#pragma unroll
    for (int k = 0; k < D; ++k) a[k] = ldsrow[k * 64 + lane];
}
__global__ __launch_bounds__(512, 2) void k(const float* __restrict__ W, float* out, int mode, int reps, unsigned long long* cyc) {
    __shared__ float lds[8][D * 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool valu_role = wave < 4;
    unsigned long long t0 = clock64();
    if (valu_role) {
        if (mode & 1) {
            float a[D];
#pragma unroll
            for (int k = 0; k < D; ++k) a[k] = 0.01f * (k + lane);
            for (int r = 0; r < reps; ++r) valu_layer(W, a, lds[wave], lane);
            float s = 0; for (int k = 0; k < D; ++k) s += a[k];
            out[blockIdx.x * 512 + threadIdx.x] = s;
        }
    } else {
        if (mode & 2) {
            f32x4 acc[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[t] = f32x4{0, 0, 0, 0};
            float av = lane * 0.001f, bv = lane * 0.002f;
            // per "layer" of the valu role the dW partner does 256 MFMAs (64 rows, one 50x51 layer)
            for (int r = 0; r < reps; ++r) {
#pragma unroll
                for (int s = 0; s < 16; ++s)
#pragma unroll
                    for (int t = 0; t < 16; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[t], 0, 0, 0);
            }
            float s = 0; for (int t = 0; t < 16; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
            out[blockIdx.x * 512 + threadIdx.x] = s;
        }
    }
    unsigned long long t1 = clock64();
    if (blockIdx.x == 0 && lane == 0) cyc[wave] = t1 - t0;
}
int main() {
    float *W, *out; unsigned long long* cyc;
    hipMalloc(&W, 52 * 50 * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64);
    std::vector<float> hw(52 * 50, 0.01f); hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    const int reps = 200;
    for (int mode = 1; mode <= 3; ++mode) {
        for (int it = 0; it < 2; ++it) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, W, out, mode, reps, cyc);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long hc[8]; hipMemcpy(hc, cyc, 64, hipMemcpyDeviceToHost);
            if (it == 1) printf("mode %d (1=valu,2=mfma,3=both): %.1f us; per rep: valu wave %.0f cycles (ideal %d), mfma wave %.0f cycles (ideal %d)\n",
                                mode, ms * 1000, (double)hc[0] / reps, 2500 * 2, (double)hc[4] / reps, 256 * 32);
        }
    }
    return 0;
}
