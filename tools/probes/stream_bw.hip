// Probe: pure load pipeline of the streaming kernel, no math.
// Persistent workgroups (wgs_per_cu x 256 CUs), each walks `per` consecutive tiles of one head (head-major spans
// like stage1_stream_kernel) and keeps `ahead` tiles in flight.  A tile = 128 token rows x {K,V} x 256 B,
// token stride 16 KB (Llama-2-7B pool).  mode 0: LDS-DMA (global_load_lds_dwordx4), mode 1: register loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16(const void* g, unsigned lds) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds) : "memory");
}
template <int AHEAD>
__global__ __launch_bounds__(256) void k(const char* base, unsigned* out, int ntile, int nhead, int mode) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
    const long long U = (long long)ntile * nhead;
    const int u0 = (int)(U * blockIdx.x / gridDim.x), u1 = (int)(U * (blockIdx.x + 1) / gridDim.x);
    unsigned acc = 0;
    auto issue = [&](int u, int buf) {
        const int head = u / ntile, tile = u % ntile;
#pragma unroll
        for (int i = 0; i < 16; ++i) {  // wave w: rows 32w..32w+31; i<8 K, i>=8 V; 4 rows per instruction
            const int row = 32 * w + (i & 7) * 4 + (l >> 4);
            const char* src = base + ((size_t)tile * 128 + row) * 16384 + (i >= 8 ? 8192 : 0) + (size_t)head * 256 + (l & 15) * 16;
            if (mode == 0) dma16(src, buf * 65536 + (i >= 8 ? 32768 : 0) + (32 * w + (i & 7) * 4) * 256);
            else acc += ((const unsigned*)src)[0];
        }
    };
    for (int a = 0; a < AHEAD && u0 + a < u1; ++a) issue(u0 + a, a % (AHEAD + 1));
    for (int u = u0; u < u1; ++u) {
        if (u + AHEAD < u1) issue(u + AHEAD, (u - u0 + AHEAD) % (AHEAD + 1));
        // wait for tile u: everything but the AHEAD younger tiles
        const int younger = (u1 - 1 - u) < AHEAD ? (u1 - 1 - u) : AHEAD;
        if (younger >= 2) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        acc += ((unsigned*)smem)[((u - u0) % (AHEAD + 1)) * 16384 + tid];
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    const int ntile = 82, nhead = 32, layers = 16;
    const size_t layer = (size_t)ntile * 128 * 16384;
    char* d; unsigned* o;
    hipMalloc(&d, layer * layers); hipMalloc(&o, 4);
    hipMemset(d, 1, layer * layers);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    struct Cfg { int ahead, wgs, mode, lds; } cfgs[] = {
        {1, 256, 0, 131072}, {2, 256, 0, 160 * 1024 - 4096 * 0 - 0}, {1, 512, 0, 0}, {1, 256, 1, 131072}, {1, 512, 1, 70000}, {1, 1024, 1, 40000}, {1, 2048, 1, 20000}};
    for (auto c : cfgs) {
        if (c.ahead == 2) c.lds = 3 * 65536 > 160 * 1024 ? 160 * 1024 : 3 * 65536;
        if (c.mode == 0 && c.lds == 0) c.lds = 70000;  // 2/CU needs a tile stage < 80 KB: only the first 64 KB used... (ahead 1 => 2 bufs = 128 KB), skip
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            for (int l2 = 0; l2 < layers; ++l2) {
                if (c.ahead == 1) hipLaunchKernelGGL(k<1>, dim3(c.wgs), dim3(256), c.lds, 0, d + l2 * layer, o, ntile, nhead, c.mode);
                else hipLaunchKernelGGL(k<2>, dim3(c.wgs), dim3(256), c.lds, 0, d + l2 * layer, o, ntile, nhead, c.mode);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        double bytes = (double)ntile * nhead * 65536 * layers;
        printf("ahead %d wgs %4d mode %s lds %6d: %.1f us/layer  %.2f TB/s  (%s)\n", c.ahead, c.wgs, c.mode ? "reg" : "dma", c.lds,
               best / layers * 1e3, bytes / (best * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
