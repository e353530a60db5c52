// check (on the device): wave_sum_lane0 (common.hpp: permlane swaps + row_shl DPP) returns in lane 0 the bits of wave_sum (six __shfl_down steps)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I tensorbnn_amd/csrc -I include -o tools/ubench/wsum tools/ubench/wsum.hip && tools/ubench/wsum
#include "common.hpp"
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void k(const double* in, double* a, double* b) {
    const double v = in[blockIdx.x * 64 + threadIdx.x];
    const double s0 = wave_sum(v), s1 = wave_sum_lane0(v);
    if (threadIdx.x == 0) { a[blockIdx.x] = s0; b[blockIdx.x] = s1; }
}
int main() {
    const int B = 4096;
    std::vector<double> h((size_t)B * 64);
    unsigned long long s = 88172645463325252ull;
    for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = ((double)(s >> 11) / 9007199254740992.0 - 0.5) * ((s & 7) ? 1.0 : 1e6); }
    double *in, *a, *b;
    (void)hipMalloc(&in, h.size() * 8); (void)hipMalloc(&a, B * 8); (void)hipMalloc(&b, B * 8);
    (void)hipMemcpy(in, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(B), dim3(64), 0, 0, in, a, b);
    std::vector<double> ha(B), hb(B);
    (void)hipMemcpy(ha.data(), a, B * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(hb.data(), b, B * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < B; ++i) if (std::memcmp(&ha[i], &hb[i], 8) != 0) { if (bad < 5) printf("block %d: %.17g vs %.17g\n", i, ha[i], hb[i]); ++bad; }
    printf("wave_sum_lane0 vs wave_sum: %d of %d blocks differ\n", bad, B);
    return bad != 0;
}
