// Probe: how fast can ONE CU pull L2-resident data -- by LDS-DMA (global_load_lds_dwordx4, what stage 1 uses) and by ordinary
// 16-byte loads into registers -- when every workgroup of the launch streams the SAME few megabytes over and over (so that
// after the first pass everything hits in L2)?  Round 4 question: a GQA / many-query tile is staged once per 32-row pass, i.e. the
// same 64 KB crosses L2 -> LDS four to six times; is the per-CU load path what bounds those launches (2 workgroups x 64 KB per
// ~2.2 us = ~58 GB/s per CU measured in stage 1)?
//   grid = CUs x wgs_per_cu, 256 threads; every workgroup reads `iters` x 64 KB, 16 DMA (or load) instructions per wave per 64 KB,
//   `depth` 64 KB stages in flight before the oldest is waited for.  Prints GB/s per CU and aggregate.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));
#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

template <bool NT>
__device__ __forceinline__ void dma16(const void* g, unsigned lds) {
    if (NT) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" ::"v"(g), "s"(lds) : "memory");
    else asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds) : "memory");
}

// mode 0: LDS-DMA, 1: LDS-DMA nt, 2: register loads
template <int MODE, int DEPTH>
__global__ __launch_bounds__(256) void pull(const char* base, size_t region, int iters, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
    // every workgroup walks the region from its own starting tile, 64 KB at a time: row r of a tile = 256 B
    size_t off = ((size_t)blockIdx.x * 65536u * 7u) % region;
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        const char* t = base + off + (size_t)w * 16384 + (size_t)l * 16;
        if (MODE == 2) {
            uintx4 r[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) r[i] = *(const uintx4*)(t + i * 1024);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc += r[i].x ^ r[i].w;
        } else {
            const unsigned stage = (unsigned)(it % DEPTH) * 65536u + (unsigned)w * 16384u;
#pragma unroll
            for (int i = 0; i < 16; ++i) dma16<MODE == 1>(t + i * 1024, stage + i * 1024);
            // wait until at most (DEPTH - 1) stages of this wave are still in flight
            if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        }
        off += 65536;
        if (off >= region) off -= region;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE != 2) acc = ((unsigned*)smem)[tid];
    if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE, int DEPTH>
static void run(const char* what, const char* d, size_t region, int cus, int wgs_per_cu, unsigned* o, hipEvent_t e0, hipEvent_t e1) {
    const int iters = 64;
    const size_t lds = (size_t)DEPTH * 65536;
    CK(hipFuncSetAttribute((const void*)pull<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(MODE == 2 ? 1024 : lds)));
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((pull<MODE, DEPTH>), dim3(cus * wgs_per_cu), dim3(256), MODE == 2 ? 1024 : lds, 0, d, region, iters, o);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    const double bytes = (double)cus * wgs_per_cu * iters * 65536.0;
    printf("  %-34s %d WG/CU, region %4zu MB: %7.1f us  %6.1f GB/s per CU  %6.2f TB/s  (%.2f us per 64 KB per workgroup)\n", what, wgs_per_cu,
           region >> 20, best * 1e3, bytes / (best * 1e-3) / 1e9 / cus, bytes / (best * 1e-3) / 1e12, best * 1e3 / iters);
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t cap = (size_t)1 << 30;
    char* d;
    unsigned* o;
    CK(hipMalloc(&d, cap));
    CK(hipMalloc(&o, 4));
    CK(hipMemset(d, 1, cap));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("%d CUs; every workgroup streams 64 x 64 KB out of a shared region (small region = L2 / Infinity Cache hits)\n", cus);
    for (size_t region : {(size_t)4 << 20, (size_t)16 << 20, (size_t)128 << 20, (size_t)1 << 30}) {
        run<0, 1>("LDS-DMA, 1 stage (wait each tile)", d, region, cus, 2, o, e0, e1);
        run<0, 2>("LDS-DMA, 2 stages in flight", d, region, cus, 2, o, e0, e1);
        run<1, 2>("LDS-DMA nt, 2 stages in flight", d, region, cus, 2, o, e0, e1);
        run<0, 2>("LDS-DMA, 2 stages in flight", d, region, cus, 1, o, e0, e1);
        run<2, 1>("register loads, 16 x 16 B in flight", d, region, cus, 2, o, e0, e1);
        run<2, 1>("register loads, 16 x 16 B in flight", d, region, cus, 4, o, e0, e1);
    }
    return 0;
}
