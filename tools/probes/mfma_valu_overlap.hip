// Do MFMA work of one wave and VALU work of ANOTHER wave on the same SIMD overlap on gfx950?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_valu_overlap.hip -o /tmp/ovl && /tmp/ovl
// One workgroup per CU; `mode` picks what the 8 waves do: waves 0-3 sit on SIMDs 0-3, waves 4-7 on the same SIMDs again.
//   mode 0: waves 0-3 MFMA only (4-7 exit)         mode 1: waves 4-7 VALU only (0-3 exit)
//   mode 2: waves 0-3 MFMA, waves 4-7 VALU          mode 3: all 8 waves alternate MFMA phase / VALU phase in lockstep
//   mode 4: like 3, waves 4-7 start with the VALU phase (anti-phase)
//   mode 5: waves 0-3 only, MFMAs and VALU work INTERLEAVED in one instruction stream (1 MFMA : 4 VALU)
//   mode 6: all 8 waves, each interleaved like 5
//   mode 7 / 8: waves 0-3 / all 8 waves: 64 MFMAs whose A operand is a FRESH 1 KB fragment from LDS each (ds_read_b128, conflict-free,
//               fragments of MFMA i + 4 read before MFMA i): the prefill kernel's LDS diet without its VALU work
//   mode 9: all 8 waves: mode 8's MFMAs + LDS reads with the VALU phase after them (the prefill tile's shape)
//   mode 10: mode 9 + one workgroup barrier per iteration        mode 11: mode 10 + 8 LDS-DMA instructions (8 KB) per wave and iteration
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void mfma_phase(floatx16 (&acc)[4], half8 a, half8 b) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);  // 32 MFMAs
}
__device__ __forceinline__ void valu_phase(float (&x)[16], float s) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[j], s, -1.0f)) + x[(j + 1) & 15] * 0.5f;  // 64 exp + 128 fma/mul/add
}
__device__ __forceinline__ void mixed_phase(floatx16 (&acc)[4], half8 a, half8 b, float (&x)[16], float s) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {  // 64 MFMAs, 64 exp + 192 others: per MFMA one exp + three plain VALU
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
            const int q = (4 * i + j) & 15;
            x[q] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[q], s, -1.0f)) + x[(q + 1) & 15] * 0.5f;
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);  // 4 VALU
        }
    }
}
__device__ __forceinline__ void lds_mfma_phase(floatx16 (&acc)[4], const char* lds, half8 b) {
    half8 af[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) af[0][j] = *reinterpret_cast<const half8*>(lds + j * 1024);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (i + 1 < 16) {
#pragma unroll
            for (int j = 0; j < 4; ++j) af[(i + 1) & 1][j] = *reinterpret_cast<const half8*>(lds + ((i + 1) * 4 + j) * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i & 1][j], b, acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}
__global__ __launch_bounds__(512, 1) void k(float* out, int mode, int iters, const char* src) {
    extern __shared__ char smem[];
    const int w = threadIdx.x >> 6;
    floatx16 acc[4] = {};
    half8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)(threadIdx.x * 0.001f + i), b[i] = (_Float16)(i * 0.01f);
    float x[16];
    for (int j = 0; j < 16; ++j) x[j] = threadIdx.x * 0.01f + j;
    const bool lo = w < 4;
    for (int i = threadIdx.x; i < 128 * 1024 / 4; i += 512) reinterpret_cast<float*>(smem)[i] = i * 1e-6f;
    __syncthreads();
    const char* lds = smem + (w & 7) * 8192 + (threadIdx.x & 63) * 16;  // lane-linear 16-byte reads: conflict-free; 64 KB window per wave pair
    if ((mode == 0 && !lo) || (mode == 1 && lo) || (mode == 5 && !lo) || (mode == 7 && !lo)) return;
    for (int it = 0; it < iters; ++it) {
        if (mode == 7 || mode == 8) { lds_mfma_phase(acc, lds, b); }
        else if (mode == 9) { lds_mfma_phase(acc, lds, b); valu_phase(x, 0.999f); }
        else if (mode == 10 || mode == 11) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (mode == 11) {
                const char* g = src + ((size_t)(blockIdx.x * 37 + it) % 960) * 65536 + w * 8192 + (threadIdx.x & 63) * 16;
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    const unsigned dst = __builtin_amdgcn_readfirstlane(64 * 1024 + w * 8192 + d * 1024);
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g + d * 1024), "s"(dst) : "memory");
                }
            }
            lds_mfma_phase(acc, lds, b);
            valu_phase(x, 0.999f);
        }
        else if (mode >= 5) { mixed_phase(acc, a, b, x, 0.999f); }
        else if (mode == 0 || (mode == 2 && lo)) { mfma_phase(acc, a, b); mfma_phase(acc, a, b); }
        else if (mode == 1 || (mode == 2 && !lo)) { valu_phase(x, 0.999f); }
        else if (mode == 3 || (mode == 4 && lo)) { mfma_phase(acc, a, b); mfma_phase(acc, a, b); valu_phase(x, 0.999f); }
        else { valu_phase(x, 0.999f); mfma_phase(acc, a, b); mfma_phase(acc, a, b); }
    }
    float r = 0;
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) r += acc[j][i];
    for (int j = 0; j < 16; ++j) r += x[j];
    out[blockIdx.x * 512 + threadIdx.x] = r;
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024);
    char* src; hipMalloc(&src, 64 << 20); hipMemset(src, 0, 64 << 20);
    for (int mode = 0; mode < 12; ++mode) {
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 136 * 1024, 0, out, mode, 100, src);
        hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(256), dim3(512), 136 * 1024, 0, out, mode, iters, src); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d: %.1f us total, %.3f us per iteration (64 MFMAs and/or one VALU phase per wave)\n", mode, ms * 1e3, ms * 1e3 / iters);
    }
    return 0;
}
