// check of the transposed-block store / load helpers of kernels_wide.hpp (buffer instructions): hipcc --offload-arch=gfx950 tblock.hip && ./a.out
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
#define WIDE_RSRC_FLAGS 0x00020000
__device__ __forceinline__ int tblk_wr_off(int i16, int g) { return ((4 * g) * 16 + (i16 & 3) * 4 + (i16 >> 2)) * 4; }
__device__ __forceinline__ int tblk_rd_off(int lane) { return ((lane & 15) * 16 + (lane >> 4) * 4) * 4; }
__global__ void k(float* blk, float* out_rd, float* out_d) {
    const int lane = threadIdx.x, i16 = lane & 15, g = lane >> 4;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(blk, 0, 2 * 1024, WIDE_RSRC_FLAGS);
    for (int b = 0; b < 2; ++b) {
        f32x4 v;
        for (int j = 0; j < 4; ++j) v[j] = 1000.f * b + 100.f * i16 + (4 * g + j);          // value encodes (block, row, slot)
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float f = v[j]; __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(f), rs, tblk_wr_off(i16, g), b * 1024 + j * 64, 0); }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __threadfence();
    for (int b = 0; b < 2; ++b) {
        f32x4 r = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, tblk_rd_off(lane), b * 1024, 0));
        for (int s = 0; s < 4; ++s) out_rd[(b * 64 + lane) * 4 + s] = r[s];
        for (int j = 0; j < 4; ++j) out_d[(b * 64 + lane) * 4 + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, tblk_wr_off(i16, g), b * 1024 + j * 64, 0));
    }
}
int main() {
    float *blk, *o1, *o2;
    hipMalloc(&blk, 4096); hipMalloc(&o1, 2048); hipMalloc(&o2, 2048); hipMemset(blk, 0, 4096);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, blk, o1, o2);
    std::vector<float> h1(512), h2(512);
    hipMemcpy(h1.data(), o1, 2048, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), o2, 2048, hipMemcpyDeviceToHost);
    int bad1 = 0, bad2 = 0;
    for (int b = 0; b < 2; ++b) for (int lane = 0; lane < 64; ++lane) for (int s = 0; s < 4; ++s) {
        const int i = lane & 15, g = lane >> 4;
        const float want_rd = 1000.f * b + 100.f * (4 * s + g) + i;          // reader: (row 4s+g, slot i)
        const float want_d = 1000.f * b + 100.f * i + (4 * g + s);           // D layout: (row i, slot 4g+j)
        if (h1[(b * 64 + lane) * 4 + s] != want_rd) { if (bad1 < 5) printf("rd mismatch b %d lane %d s %d: %g want %g\n", b, lane, s, h1[(b * 64 + lane) * 4 + s], want_rd); ++bad1; }
        if (h2[(b * 64 + lane) * 4 + s] != want_d) { if (bad2 < 5) printf("d mismatch b %d lane %d j %d: %g want %g\n", b, lane, s, h2[(b * 64 + lane) * 4 + s], want_d); ++bad2; }
    }
    printf("reader-layout mismatches %d, D-layout mismatches %d\n", bad1, bad2);
    return 0;
}
