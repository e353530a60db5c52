// microbenchmark: cost of a grid-wide barrier (atomic counter at device scope + agent-scope fences) over 256 workgroups,
// one per CU (140 KB of LDS each), launched cooperatively so that a grid that cannot be co-resident is refused
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256, 1) void k(unsigned long long* counter, int iters, unsigned long long* cyc, float* sink) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = threadIdx.x;
    unsigned long long target = 0;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        target += gridDim.x;
        __threadfence();                                   // release: this workgroup's stores
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            long spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 20000000) break;             // never hang the GPU
            }
        }
        __syncthreads();
        __threadfence();                                   // acquire
    }
    const unsigned long long t1 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
    sink[blockIdx.x] = lds[threadIdx.x & 63];
}
int main() {
    unsigned long long *counter, *cyc; float* sink;
    (void)hipMalloc(&counter, 8); (void)hipMalloc(&cyc, 8); (void)hipMalloc(&sink, 4096);
    (void)hipMemset(counter, 0, 8);
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    for (int grid : {64, 128, 256}) {
        for (int rep = 0; rep < 2; ++rep) {
            int iters = 200;
            (void)hipMemset(counter, 0, 8);
            void* args[] = {&counter, &iters, &cyc, &sink};
            hipError_t e = hipLaunchCooperativeKernel((const void*)k, dim3(grid), dim3(256), args, 140 * 1024, 0);
            if (e != hipSuccess) { printf("grid %d: cooperative launch refused: %s\n", grid, hipGetErrorString(e)); break; }
            (void)hipDeviceSynchronize();
            unsigned long long hc = 0; (void)hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
            if (rep) printf("grid %3d: %.2f us per grid barrier (100 MHz wall clock)\n", grid, (double)hc / iters / 100.0);
        }
    }
    return 0;
}
