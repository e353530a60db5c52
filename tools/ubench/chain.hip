// microbenchmark: cycles per v_mfma_f32_16x16x4_f32 with R rotating accumulators (dependent every R-th instruction),
// accumulators in AccVGPRs ("a") or ArchVGPRs ("v"); and the same for v_mfma_f32_4x4x1_16b_f32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int R, int KIND>
__global__ __launch_bounds__(256, 1) void k(float* out, int reps, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[12];
    for (int i = 0; i < 12; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = lane * 0.001f, b = lane * 0.002f;
    unsigned long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int s = 0; s < 24; ++s) {
            if constexpr (KIND == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[s % R]) : "v"(a), "v"(b));
            if constexpr (KIND == 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[s % R]) : "v"(a), "v"(b));
            if constexpr (KIND == 2) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc[s % R]) : "v"(a), "v"(b));
        }
    }
    unsigned long long t1 = clock64();
    float s = 0; for (int i = 0; i < 12; ++i) s += acc[i][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int R, int KIND> void run(float* out, unsigned long long* cyc) {
    const int reps = 2000; unsigned long long hc = 0;
    for (int it = 0; it < 2; ++it) {
        hipLaunchKernelGGL((k<R, KIND>), dim3(256), dim3(256), 0, 0, out, reps, cyc);
        (void)hipDeviceSynchronize(); (void)hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    }
    const char* kn[] = {"16x16x4 acc=AGPR", "16x16x4 acc=VGPR", "4x4x1   acc=VGPR"};
    printf("%s  R=%2d : %6.1f cycles per MFMA\n", kn[KIND], R, (double)hc / reps / 24);
}
int main() {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 64);
    run<1, 0>(out, cyc); run<2, 0>(out, cyc); run<3, 0>(out, cyc); run<4, 0>(out, cyc); run<12, 0>(out, cyc);
    run<1, 1>(out, cyc); run<2, 1>(out, cyc); run<3, 1>(out, cyc); run<4, 1>(out, cyc); run<12, 1>(out, cyc);
    run<1, 2>(out, cyc); run<2, 2>(out, cyc); run<3, 2>(out, cyc); run<4, 2>(out, cyc); run<12, 2>(out, cyc);
    return 0;
}
