// Probe: what does s_memtime tick in?  Runs a dependent chain of N v_fma_f32 (4 cycles each on a
// wave64/SIMD32? measured) and reports s_memtime delta, s_memrealtime (100 MHz) delta.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out, unsigned long long* t, int n) {
    float x = out[threadIdx.x];
    unsigned long long r0 = wall_clock64();
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    unsigned long long cc0 = clock64();
    for (int i = 0; i < n; ++i) x = __builtin_fmaf(x, 1.0001f, 0.5f);
    unsigned long long cc1 = clock64();
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = wall_clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) { t[0] = c1 - c0; t[1] = r1 - r0; t[2] = cc1 - cc0; }
}
int main() {
    float* o; unsigned long long* t; hipMalloc(&o, 256); hipMalloc(&t, 64);
    hipMemset(o, 0, 256);
    for (int n : {100000, 1000000}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, t, n);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("host events: %.1f us; ", ms * 1e3);
        unsigned long long h[3]; hipMemcpy(h, t, 24, hipMemcpyDeviceToHost);
        printf("n=%d fma chain: s_memtime=%llu  realtime(100MHz)=%llu (=%.1f us)  clock64=%llu -> memtime ticks/us=%.1f, ticks per fma=%.2f\n",
               n, h[0], h[1], h[1] / 100.0, h[2], h[0] / (h[1] / 100.0), (double)h[0] / n);
    }
    return 0;
}
