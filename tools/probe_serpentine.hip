// Design probe: does a consumer that sweeps a tensor in the OPPOSITE direction of its producer's sweep find more of it in the
// 256 MB Infinity Cache?  copy kernels, 16 B per lane, grid-stride sweep, 1024 workgroups resident.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_serpentine.hip -o /tmp/probe_serp && /tmp/probe_serp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void copy_k(const uint4* __restrict__ s, uint4* __restrict__ d, long long n, int rev) {
    const long long G = (long long)gridDim.x * 256;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += G) {
        const long long j = rev ? n - 1 - i : i;
        d[j] = s[j];
    }
}
int main() {
    hipStream_t st; hipStreamCreate(&st);
    for (int mb : {67, 134, 201, 268}) {
        const long long n = (long long)mb * 1000 * 1000 / 16;
        uint4 *a, *b, *c, *big;
        hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&c, n * 16); hipMalloc(&big, 600LL * 1000 * 1000);
        hipMemset(a, 1, n * 16); hipMemset(b, 0, n * 16); hipMemset(c, 0, n * 16);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int mode = 0; mode < 3; ++mode) {               // 0: cold consumer, 1: same direction, 2: opposite direction
            float best = 1e9f;
            for (int rep = 0; rep < 7; ++rep) {
                hipMemsetAsync(big, rep, 600LL * 1000 * 1000, st);
                if (mode) hipLaunchKernelGGL(copy_k, dim3(1024), dim3(256), 0, st, a, b, n, 0);
                hipEventRecord(e0, st);
                hipLaunchKernelGGL(copy_k, dim3(1024), dim3(256), 0, st, b, c, n, mode == 2 ? 1 : 0);
                hipEventRecord(e1, st);
                hipStreamSynchronize(st);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("%4d MB  %-18s consumer %.1f us  (%.2f TB/s r+w)\n", mb, mode == 0 ? "cold" : mode == 1 ? "same direction" : "opposite direction", best * 1e3,
                   2.0 * mb / (best * 1e3));
        }
        hipFree(a); hipFree(b); hipFree(c); hipFree(big);
    }
    return 0;
}
