// microbenchmark: which instruction classes overlap with an f32 MFMA stream inside ONE wave (1 wave per SIMD)?
// body: 16 x { v_mfma_f32_16x16x4_f32 ; K x <instr T> }, independent operands.  cycles/rep: 512 = fully hidden.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MF(i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b))
template <int T, int K, bool mf>
__global__ __launch_bounds__(256, 1) void k(float* out, int reps, unsigned long long* cyc) {
    __shared__ float lds[4096];
    const int lane = threadIdx.x & 63;
    f32x4 acc[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
    float a = lane * 0.001f, b = lane * 0.002f;
    float x[8]; for (int i = 0; i < 8; ++i) x[i] = lane + i;
    float y[8]; for (int i = 0; i < 8; ++i) y[i] = 0.5f * lane + i;
    f32x4 q[4]; for (int i = 0; i < 4; ++i) q[i] = f32x4{1.f * i, 2, 3, 4};
    lds[threadIdx.x] = lane; __syncthreads();
    const unsigned addr = (threadIdx.x & 255) * 16;
    unsigned long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if constexpr (mf) MF(s & 3);
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const int i = (s * K + j) & 7;
                if constexpr (T == 1) asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(y[i]));
                if constexpr (T == 2) asm volatile("v_max_i32 %0, 0, %1" : "=v"(x[i]) : "v"(y[i]));
                if constexpr (T == 3) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(x[i]) : "v"(y[i]), "v"(a));
                if constexpr (T == 4) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x[i]) : "a"(q[i & 3][0]));
                if constexpr (T == 5) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(q[i & 3][1]) : "v"(y[i]));
                if constexpr (T == 6) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(y[i]), "v"(a));
                if constexpr (T == 7) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double*)&q[i & 3]) : "v"(*(double*)&y[(i & 3) * 2]), "v"(*(double*)&y[0]));
                if constexpr (T == 8) asm volatile("v_cmp_lt_i32 vcc, 0, %0" :: "v"(y[i]) : "vcc");
                if constexpr (T == 9) asm volatile("ds_read_b128 %0, %1" : "=v"(q[i & 3]) : "v"(addr));
                if constexpr (T == 10) asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(y[i]));
                if constexpr (T == 11) asm volatile("v_add_u32 %0, %1, %2" : "=v"(x[i]) : "v"(y[i]), "v"(a));
                if constexpr (T == 12) asm volatile("s_nop 0");
                if constexpr (T == 13) asm volatile("v_cmp_lt_i32 %0, 0, %1" : "=s"(*(unsigned long long*)&q[i&3]) : "v"(y[i]));
            }
        }
        if constexpr (T == 9 || T == 10) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    unsigned long long t1 = clock64();
    float s = 0; for (int i = 0; i < 4; ++i) s += acc[i][0] + q[i][0] + q[i][1]; for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int T, int K> void run(const char* name, float* out, unsigned long long* cyc) {
    const int reps = 2000;
    double r[2];
    for (int mf = 0; mf <= 1; ++mf) {
        unsigned long long hc = 0;
        for (int it = 0; it < 2; ++it) {
            if (mf) hipLaunchKernelGGL((k<T, K, true>), dim3(256), dim3(256), 0, 0, out, reps, cyc);
            else hipLaunchKernelGGL((k<T, K, false>), dim3(256), dim3(256), 0, 0, out, reps, cyc);
            hipDeviceSynchronize();
            hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
        }
        r[mf] = (double)hc / reps;
    }
    printf("%-18s K=%d : alone %7.1f  with 16 MFMA %7.1f cycles per 16-group\n", name, K, r[0], r[1]);
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 64);
    run<0, 1>("none", out, cyc);
    run<1, 2>("v_mov_b32", out, cyc);
    run<2, 2>("v_max_i32", out, cyc);
    run<3, 2>("v_cndmask", out, cyc);
    run<4, 2>("accvgpr_read", out, cyc);
    run<5, 2>("accvgpr_write", out, cyc);
    run<6, 2>("v_fma_f32", out, cyc);
    run<7, 2>("v_pk_fma_f32", out, cyc);
    run<8, 2>("v_cmp(vcc)", out, cyc);
    run<9, 1>("ds_read_b128", out, cyc);
    run<10, 2>("ds_write_b32", out, cyc);
    run<11, 2>("v_add_u32", out, cyc);
    run<12, 4>("s_nop", out, cyc);
    run<1, 6>("v_mov_b32", out, cyc);
    run<6, 6>("v_fma_f32", out, cyc);
    return 0;
}
