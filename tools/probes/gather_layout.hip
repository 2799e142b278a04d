// Probe: non-persistent workgroups, each loads one (tile, head) = 64 KB, either into registers (16 x dwordx4
// per thread) or by LDS-DMA (16 x global_load_lds_dwordx4 per thread), then exits.  Occupancy by LDS size.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16(const void* g, unsigned lds) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds) : "memory");
}
__global__ __launch_bounds__(256) void k(const char* base, unsigned* out, int nhead, int dma, int hm, int ntile) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x, tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
    const int tile = b / nhead, head = b % nhead;
    unsigned acc = 0;
    if (dma) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = 32 * w + (i & 7) * 4 + (l >> 4);
            dma16(hm ? base + (i >= 8 ? (size_t)ntile * 128 * 8192 : 0) + ((size_t)head * ntile * 128 + (size_t)tile * 128 + row) * 256 + (l & 15) * 16 : base + ((size_t)tile * 128 + row) * 16384 + (i >= 8 ? 8192 : 0) + (size_t)head * 256 + (l & 15) * 16,
                  (i >= 8 ? 32768 : 0) + (32 * w + (i & 7) * 4) * 256);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc = ((unsigned*)smem)[tid];
    } else {
        uintx4 r[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = 32 * w + (i & 7) * 4 + (l >> 4);
            r[i] = *(const uintx4*)(hm ? base + (i >= 8 ? (size_t)ntile * 128 * 8192 : 0) + ((size_t)head * ntile * 128 + (size_t)tile * 128 + row) * 256 + (l & 15) * 16 : base + ((size_t)tile * 128 + row) * 16384 + (i >= 8 ? 8192 : 0) + (size_t)head * 256 + (l & 15) * 16);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += r[i].x ^ r[i].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    const int ntile = 82, nhead = 32, layers = 16;
    const size_t layer = (size_t)ntile * 128 * 16384;
    char* d; unsigned* o;
    hipMalloc(&d, layer * layers); hipMalloc(&o, 4); hipMemset(d, 1, layer * layers);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int hm = 0; hm < 2; ++hm) for (int dma = 0; dma < 2; ++dma)
        for (int lds : {65536, 81920, 163840}) {
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                for (int l2 = 0; l2 < layers; ++l2) hipLaunchKernelGGL(k, dim3(ntile * nhead), dim3(256), lds, 0, d + l2 * layer, o, nhead, dma, hm, ntile);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep && ms < best) best = ms;
            }
            printf("%s %s lds %3d KB (%d WG/CU): %.1f us/layer %.2f TB/s\n", hm ? "head-major " : "token-major", dma ? "dma" : "reg", lds / 1024, 163840 / lds, best / layers * 1e3,
                   (double)ntile * nhead * 65536 * layers / (best * 1e-3) / 1e12);
        }
    return 0;
}
