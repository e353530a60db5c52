#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#define TBNN_NO_FAST_REGISTRY
#include "../../tensorbnn_amd/csrc/kernels_fast.hpp"
__global__ void k(const float* z, float* o, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) o[i] = actc_fwd<TBNN_ACT_TANH>(z[i]); }
int main() {
    const int n = 1 << 20; float *hz = new float[n], *ho = new float[n], *dz, *dout;
    for (int i = 0; i < n; ++i) { double u = (double)i / n; hz[i] = (float)((i & 1 ? -1 : 1) * pow(10.0, -6.0 + 7.3 * u)); }
    hipMalloc(&dz, n * 4); hipMalloc(&dout, n * 4); hipMemcpy(dz, hz, n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dz, dout, n); hipMemcpy(ho, dout, n * 4, hipMemcpyDeviceToHost);
    double worst = 0, wz = 0;
    for (int i = 0; i < n; ++i) { double r = tanh((double)hz[i]); double e = fabs(ho[i] - r) / fabs(r); if (e > worst) { worst = e; wz = hz[i]; } }
    printf("max relative error of the fused tanh over |z| in [1e-6, 20]: %.2e at z = %g\n", worst, wz);
    return 0;
}
