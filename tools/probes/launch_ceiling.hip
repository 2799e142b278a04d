// Probe: what ONE cold launch of a given size can draw from HBM on this part, with nothing but the loads in it -- the ceiling the
// stage-1 kernel's small launches are judged against (stream_read.hip prices 172 MB / 1 GB / 4 GB only).
//
// For every BASELINE shape (unique KV tokens x bytes per token slot) two bare readers, each over 32 rotating "layer pools" (the
// working set is far beyond the 256 MB Infinity Cache, like bench.py's rotating layers), back to back on one stream:
//   contig  : grid-stride float4 reads of the same number of bytes, 256 / 512 / 1024 workgroups
//   gather  : what stage 1's data path does and nothing else -- a workgroup per (chunk of C tiles, KV head), 4 waves, wave w
//             fetching rows [32w, 32w+32) of each 128-slot tile (K row and V row of its head: 256 B each at the pool's slot stride)
//             by global_load_lds_dwordx4 into its own LDS slices, two workgroups per CU (77 KB of LDS), tile i+1 requested
//             when tile i has landed; slot lists as a decode loop leaves them: the prompt contiguous, the branches' tokens
//             interleaved step by step (slot = prefix + step * width + branch), read in DFS order
// Output: best and median us per launch; the smallest is that shape's `ceiling_us`.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

__global__ __launch_bounds__(256) void rd_contig(const uintx4* p, size_t n, unsigned* out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    unsigned acc = 0;
    for (; i + 3 * stride < n; i += 4 * stride) {
        uintx4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc += a.x ^ b.y ^ c.z ^ d.w;
    }
    for (; i < n; i += stride) acc += p[i].x;
    if (acc == 0x12345678u) out[0] = acc;
}

__device__ __forceinline__ void dma16nt(const void* g, unsigned lds) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" ::"v"(g), "s"(lds) : "memory");
}

// slots: [tiles][128] int32 (pads repeat a valid slot); one workgroup per (chunk, head): chunk c of a run takes tiles c, c + S, ...
__global__ __launch_bounds__(256, 2) void rd_gather(const char* pool, const int* slots, int tiles, int S, int Hkv, int slot_bytes,
                                                    unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
    const int chunk = blockIdx.x / Hkv, head = blockIdx.x % Hkv;
    const char* kb = pool + (size_t)head * 256 + (l & 15) * 16;
    const char* vb = kb + (size_t)Hkv * 256;
    for (int t = chunk; t < tiles; t += S) {
        const int* sl = slots + (size_t)t * 128 + 32 * w;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const size_t row = (size_t)sl[4 * i + (l >> 4)] * slot_bytes;
            dma16nt(kb + row, (unsigned)(w * 8192 + i * 1024));
            dma16nt(vb + row, (unsigned)(32768 + w * 8192 + i * 1024));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (reinterpret_cast<unsigned*>(smem)[tid] == 0x12345678u) out[0] = 1;
}

// TWO tiles in flight per wave (the gather above is a depth-1 pipeline -- tile i + 1 is requested when tile i has
// landed -- so is its number a ceiling?).  Double-buffered slices: 128 KB of LDS per workgroup, hence ONE workgroup per CU instead
// of two; tile i + 2 is requested as soon as tile i has landed (counted s_waitcnt: 16 requests may stay in flight).
__global__ __launch_bounds__(256, 1) void rd_gather_d2(const char* pool, const int* slots, int tiles, int S, int Hkv, int slot_bytes,
                                                       unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
    const int chunk = blockIdx.x / Hkv, head = blockIdx.x % Hkv;
    const char* kb = pool + (size_t)head * 256 + (l & 15) * 16;
    const char* vb = kb + (size_t)Hkv * 256;
    auto issue = [&](int t, int buf) {
        const int* sl = slots + (size_t)t * 128 + 32 * w;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const size_t row = (size_t)sl[4 * i + (l >> 4)] * slot_bytes;
            dma16nt(kb + row, (unsigned)(buf * 65536 + w * 8192 + i * 1024));
            dma16nt(vb + row, (unsigned)(buf * 65536 + 32768 + w * 8192 + i * 1024));
        }
    };
    int t = chunk, buf = 0;
    if (t < tiles) issue(t, 0);
    for (; t < tiles; t += S, buf ^= 1) {
        if (t + S < tiles) {
            issue(t + S, buf ^ 1);
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // tile t has landed, tile t + S is in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    if (reinterpret_cast<unsigned*>(smem)[tid] == 0x12345678u) out[0] = 1;
}

// The same bytes per step and per wave, but a workgroup serves a GROUP of HG adjacent KV heads: a step stages 128 (token, head)
// rows, head fastest -- the HG heads of one token are HG x 256 contiguous bytes of the pool ([slot][K|V][Hkv][128]), so every
// token contributes one 512-byte / 1-KB burst instead of a 256-byte piece per workgroup (would a small launch
// read faster that way?).  A 128-token tile takes HG steps; chunk c of a run takes tiles c, c + S, ...
template <int HG>
__global__ __launch_bounds__(256, 2) void rd_gather_hg(const char* pool, const int* slots, int tiles, int S, int Hkv, int slot_bytes,
                                                       unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
    const int ng = Hkv / HG;
    const int chunk = blockIdx.x / ng, hg = blockIdx.x % ng;
    const char* kb = pool + (size_t)hg * HG * 256 + (l & 15) * 16;
    const char* vb = kb + (size_t)Hkv * 256;
    for (int t = chunk; t < tiles; t += S) {
        for (int step = 0; step < HG; ++step) {
            const int* sl = slots + (size_t)t * 128 + step * (128 / HG);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = 32 * w + 4 * i + (l >> 4);  // (token, head) pair of this step, head fastest
                const size_t row = (size_t)sl[r / HG] * slot_bytes + (size_t)(r % HG) * 256;
                dma16nt(kb + row, (unsigned)(w * 8192 + i * 1024));
                dma16nt(vb + row, (unsigned)(32768 + w * 8192 + i * 1024));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    if (reinterpret_cast<unsigned*>(smem)[tid] == 0x12345678u) out[0] = 1;
}

struct Shape {
    const char* name;
    int prefix, width, blen, Hkv;  // tokens: prefix + width * blen; bytes per token slot = 4 * Hkv * 128
    double algo_mb;                // SURVEY 8(d) B_algo of the shape (adds Q read + O written)
};

int main() {
    const Shape shapes[] = {
        {"medusa64_node   (Llama-2-7B, 1016 + 64 x 1)", 1016, 64, 1, 32, 18.74},
        {"tot50_4k        (Llama-3-8B, 7680 tokens)", 4096, 42, 85, 8, 32.15},   // 4096 + 7 x 128 + 42 x 64 ~ 4096 + 42 x 85
        {"forest_8kx8 one (Llama-3-8B, 8192 + 8 x 64)", 8192, 8, 64, 8, 35.78},
        {"gqa_4kx32       (Llama-3-8B, 4096 + 32 x 200)", 4096, 32, 200, 8, 43.52},
        {"northstar len 1 (Llama-2-7B, 4096 + 32 x 1)", 4096, 32, 1, 32, 68.16},
        {"fewshot_1kx32   (Llama-2-7B, 1024 + 32 x 200)", 1024, 32, 200, 32, 122.16},
        {"northstar       (Llama-2-7B, 4096 + 32 x 200)", 4096, 32, 200, 32, 172.49},
    };
    const int layers = 32;
    unsigned* o;
    CK(hipMalloc(&o, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void*)rd_gather, hipFuncAttributeMaxDynamicSharedMemorySize, 77 * 1024));
    CK(hipFuncSetAttribute((const void*)rd_gather_d2, hipFuncAttributeMaxDynamicSharedMemorySize, 130 * 1024));
    CK(hipFuncSetAttribute((const void*)rd_gather_hg<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 77 * 1024));
    CK(hipFuncSetAttribute((const void*)rd_gather_hg<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 77 * 1024));
    // an empty launch, back to back: the fixed cost of a kernel boundary on this stream
    {
        std::vector<float> ts;
        for (int rep = 0; rep < 9; ++rep) {
            CK(hipEventRecord(e0));
            for (int l2 = 0; l2 < layers; ++l2) hipLaunchKernelGGL(rd_contig, dim3(512), dim3(256), 0, 0, (const uintx4*)o, 0, o);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            ts.push_back(ms * 1e3f / layers);
        }
        std::sort(ts.begin(), ts.end());
        printf("empty launch (512 workgroups, back to back): best %.2f us, median %.2f us\n\n", ts[0], ts[ts.size() / 2]);
    }
    for (const Shape& s : shapes) {
        const int ntok = s.prefix + s.width * s.blen;
        const int slot_bytes = 4 * s.Hkv * 128;
        const int pool_slots = ntok + 64;
        const size_t layer_bytes = (size_t)pool_slots * slot_bytes;
        const size_t kv_bytes = (size_t)ntok * slot_bytes;
        char* d;
        CK(hipMalloc(&d, layer_bytes * layers));
        CK(hipMemset(d, 1, layer_bytes * layers));
        // DFS order of the slots: prompt, then branch by branch (its tokens sit `width` apart)
        std::vector<int> order;
        for (int i = 0; i < s.prefix; ++i) order.push_back(i);
        for (int b = 0; b < s.width; ++b)
            for (int t = 0; t < s.blen; ++t) order.push_back(s.prefix + t * s.width + b);
        const int tiles = (ntok + 127) / 128;
        std::vector<int> tab((size_t)tiles * 128);
        for (size_t i = 0; i < tab.size(); ++i) tab[i] = order[i < order.size() ? i : order.size() - 1];
        int* dslots;
        CK(hipMalloc(&dslots, tab.size() * 4));
        CK(hipMemcpy(dslots, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
        printf("%s: %d tokens x %d B = %.2f MB of K/V per layer (B_algo %.2f MB), %d tiles x %d heads\n", s.name, ntok, slot_bytes,
               kv_bytes / 1e6, s.algo_mb, tiles, s.Hkv);
        double ceiling = 1e9;
        auto run = [&](const char* what, auto&& launch) {
            std::vector<float> ts;
            for (int rep = 0; rep < 12; ++rep) {
                CK(hipEventRecord(e0));
                for (int l2 = 0; l2 < layers; ++l2) launch(d + (size_t)l2 * layer_bytes);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep >= 2) ts.push_back(ms * 1e3f / layers);
            }
            std::sort(ts.begin(), ts.end());
            const double med = ts[ts.size() / 2];
            printf("    %-44s best %6.2f us  median %6.2f us  %5.2f TB/s  (%.3f of 8 TB/s)\n", what, ts[0], med, kv_bytes / med / 1e6,
                   kv_bytes / med / 1e6 / 8.0);
            ceiling = std::min(ceiling, med);
        };
        for (int wgs : {256, 512, 1024}) {
            char what[96];
            snprintf(what, sizeof what, "contig, %d workgroups", wgs);
            run(what, [&](const char* base) { hipLaunchKernelGGL(rd_contig, dim3(wgs), dim3(256), 0, 0, (const uintx4*)base, kv_bytes / 16, o); });
        }
        for (int C : {1, 2, 4, 8}) {
            if (C > 1 && tiles / C < 1) continue;
            const int S = (tiles + C - 1) / C;
            char what[96];
            snprintf(what, sizeof what, "gather, %d-tile chunks: %d workgroups", C, S * s.Hkv);
            run(what, [&](const char* base) {
                hipLaunchKernelGGL(rd_gather, dim3(S * s.Hkv), dim3(256), 77 * 1024, 0, base, dslots, tiles, S, s.Hkv, slot_bytes, o);
            });
        }
        // two tiles in flight per wave, one workgroup per CU (a question about pipeline depth, not part of the shape's ceiling unless it wins)
        for (int C : {2, 4, 8}) {
            if (tiles / C < 1) continue;
            const int S = (tiles + C - 1) / C;
            char what[96];
            snprintf(what, sizeof what, "gather depth 2, %d-tile chunks: %d workgroups", C, S * s.Hkv);
            run(what, [&](const char* base) {
                hipLaunchKernelGGL(rd_gather_d2, dim3(S * s.Hkv), dim3(256), 130 * 1024, 0, base, dslots, tiles, S, s.Hkv, slot_bytes, o);
            });
        }
        // head groups: the same steps per workgroup as the 1-head row of C tiles (C = HG x tiles per chunk)
        for (int HG : {2, 4}) {
            if (s.Hkv % HG) continue;
            for (int Ct : {1, 2}) {  // tiles per chunk -> HG x Ct steps per workgroup
                if (tiles / Ct < 1) continue;
                const int S = (tiles + Ct - 1) / Ct;
                const int wgs = S * (s.Hkv / HG);
                char what[112];
                snprintf(what, sizeof what, "gather, %d heads x %d-tile chunks (%d steps): %d wgs", HG, Ct, HG * Ct, wgs);
                const double before = ceiling;
                run(what, [&](const char* base) {
                    if (HG == 2) hipLaunchKernelGGL(rd_gather_hg<2>, dim3(wgs), dim3(256), 77 * 1024, 0, base, dslots, tiles, S, s.Hkv, slot_bytes, o);
                    else hipLaunchKernelGGL(rd_gather_hg<4>, dim3(wgs), dim3(256), 77 * 1024, 0, base, dslots, tiles, S, s.Hkv, slot_bytes, o);
                });
                ceiling = before;  // (a question about the layout, not part of the shape's ceiling)
            }
        }
        printf("    ceiling_us %.2f  (%.3f of 8 TB/s on B_algo)\n\n", ceiling, s.algo_mb / ceiling / 8.0);
        CK(hipFree(d));
        CK(hipFree(dslots));
    }
    return 0;
}
