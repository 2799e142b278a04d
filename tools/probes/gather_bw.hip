// Probe: HBM bandwidth of the paged-KV gather pattern vs a contiguous stream.
//   mode 0: contiguous   - workgroup b reads 64 KB at b*64 KB
//   mode 1: head gather  - workgroup b = (tile, head): 128 tokens x {K row, V row} of 256 B,
//                          token stride 16 KB, head offset 256 B, V at +8 KB   (Llama-2-7B pool)
//   mode 2: like 1 but heads-fastest order flipped (tile fastest)
//   mode 3: 4 heads per workgroup x 32 tokens (1 KB contiguous pieces)
// Each thread issues 16 x 16-byte loads before consuming them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(const char* base, unsigned* out, int mode, int ntile, int nhead) {
    extern __shared__ char dyn[];
    if (base == nullptr) dyn[threadIdx.x] = 1;  // keep the dynamic LDS allocation (occupancy knob)
    const int b = blockIdx.x, tid = threadIdx.x;
    uintx4 r[16];
    size_t off[16];
    if (mode == 0) {
        for (int i = 0; i < 16; ++i) off[i] = (size_t)b * 65536 + (size_t)i * 4096 + tid * 16;
    } else if (mode == 1 || mode == 2) {
        int tile = (mode == 1) ? b / nhead : b % ntile;
        int head = (mode == 1) ? b % nhead : b / ntile;
        for (int i = 0; i < 16; ++i) {  // i<8: K rows, i>=8: V rows; 16 rows per instruction-set of 256 threads
            int row = (i & 7) * 16 + (tid >> 4);
            off[i] = ((size_t)tile * 128 + row) * 16384 + (i >= 8 ? 8192 : 0) + (size_t)head * 256 + (tid & 15) * 16;
        }
    } else {
        int tile = b / (nhead / 4), hg = b % (nhead / 4);   // 32-token tiles, 4 heads
        for (int i = 0; i < 16; ++i) {
            int row = (i & 7) * 4 + (tid >> 6);
            off[i] = ((size_t)tile * 32 + row) * 16384 + (i >= 8 ? 8192 : 0) + (size_t)hg * 1024 + (tid & 63) * 16;
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = *(const uintx4*)(base + off[i]);
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += r[i].x ^ r[i].y ^ r[i].z ^ r[i].w;
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    const int ntile = 80, nhead = 32;                 // 80 tiles x 128 tokens x 16 KB = 168 MB per layer
    const size_t layer = (size_t)ntile * 128 * 16384;
    const int layers = 16;
    char* d; unsigned* o;
    hipMalloc(&d, layer * layers); hipMalloc(&o, 4);
    hipMemset(d, 1, layer * layers);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int lds_opts[5] = {0, 20 * 1024, 40 * 1024, 80 * 1024, 160 * 1024};  // -> 8, 8, 4, 2, 1 workgroups per CU
    for (int li = 0; li < 5; ++li)
    for (int mode = 0; mode < 2; ++mode) {
        const int lds = lds_opts[li];
        int grid = (mode == 3) ? ntile * 4 * (nhead / 4) : ntile * nhead;
        float best = 1e9, tot = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            for (int l = 0; l < layers; ++l) hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, d + l * layer, o, mode, mode == 3 ? ntile * 4 : ntile, nhead);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) { tot += ms; if (ms < best) best = ms; }
        }
        double bytes = (double)grid * 65536 * layers;
        printf("lds %3d KB mode %d: grid %d  %.1f us/layer  %.2f TB/s (best)\n", lds / 1024, mode, grid, best / layers * 1e3, bytes / (best * 1e-3) / 1e12);
    }
    return 0;
}
