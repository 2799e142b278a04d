// Probe: issue cost, in shader cycles, of the instructions a one-wave-per-SIMD attention loop is made of -- alone, and as fillers
// between v_mfma_f32_32x32x16_f16 (one wave per SIMD: 256 threads, 128 KB of LDS per workgroup, so one workgroup per CU).
//   hipcc --offload-arch=gfx950 -O3 -o issue_cost issue_cost.hip && ./issue_cost
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

// MODE: which instruction stream is timed (64 repetitions of the unit, ITERS times)
template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(unsigned long long* out, const char* g, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int l = threadIdx.x & 63;
    unsigned addr = (threadIdx.x * 16) & 0x3fff;
    const char* gp = g + (size_t)threadIdx.x * 16 + (size_t)blockIdx.x * 65536;
    for (int i = threadIdx.x; i < 32768; i += 256) ((float*)smem)[i] = 1.0f;
    __syncthreads();
    float x0 = l * 0.001f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) asm volatile(REP16("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        if (MODE == 1) asm volatile(REP16("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        if (MODE == 2) asm volatile(REP16("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        if (MODE == 3) asm volatile(REP16("v_mfma_f32_32x32x16_f16 a[0:15], v[20:23], v[24:27], a[0:15]\n v_mfma_f32_32x32x16_f16 a[16:31], v[20:23], v[24:27], a[16:31]\n"
                                          "v_mfma_f32_32x32x16_f16 a[32:47], v[20:23], v[24:27], a[32:47]\n v_mfma_f32_32x32x16_f16 a[48:63], v[20:23], v[24:27], a[48:63]\n") ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
        // MFMA + 4 fma fillers each
        if (MODE == 4) asm volatile(REP64("v_mfma_f32_32x32x16_f16 a[0:15], v[20:23], v[24:27], a[0:15]\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)::"v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
        // MFMA + 1 exp + 3 fma
        if (MODE == 5) asm volatile(REP64("v_mfma_f32_32x32x16_f16 a[0:15], v[20:23], v[24:27], a[0:15]\n v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)::"v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
        // MFMA + 2 exp + 4 fma
        if (MODE == 6) asm volatile(REP64("v_mfma_f32_32x32x16_f16 a[0:15], v[20:23], v[24:27], a[0:15]\n v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_exp_f32 %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %3, %3, %3, %3\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)::"v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
        // ds_read_b128 alone / tr_b64 alone
        if (MODE == 7) asm volatile(REP64("ds_read_b128 v[28:31], %0\n") "s_waitcnt lgkmcnt(0)\n" ::"v"(addr) : "v28", "v29", "v30", "v31");
        if (MODE == 8) asm volatile(REP64("ds_read_b64_tr_b16 v[28:29], %0\n") "s_waitcnt lgkmcnt(0)\n" ::"v"(addr) : "v28", "v29");
        // MFMA + one b128 read
        if (MODE == 9) asm volatile(REP64("v_mfma_f32_32x32x16_f16 a[0:15], v[20:23], v[24:27], a[0:15]\n ds_read_b128 v[28:31], %0\n") "s_waitcnt lgkmcnt(0)\n" ::"v"(addr) : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31");
        // MFMA + two tr reads
        if (MODE == 10) asm volatile(REP64("v_mfma_f32_32x32x16_f16 a[0:15], v[20:23], v[24:27], a[0:15]\n ds_read_b64_tr_b16 v[28:29], %0\n ds_read_b64_tr_b16 v[30:31], %0 offset:2048\n") "s_waitcnt lgkmcnt(0)\n" ::"v"(addr) : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31");
        // MFMA + b128 + 4 fma
        if (MODE == 11) asm volatile(REP64("v_mfma_f32_32x32x16_f16 a[0:15], v[20:23], v[24:27], a[0:15]\n ds_read_b128 v[28:31], %4\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n") "s_waitcnt lgkmcnt(0)\n" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(addr) : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31");
        // LDS-DMA, 16 per iteration among MFMAs (4 MFMAs per DMA): vaddr form and saddr form
        if (MODE == 12) asm volatile(REP16("s_mov_b32 m0, %1\n s_nop 0\n global_load_lds_dwordx4 %0, off\n" REP4("v_mfma_f32_32x32x16_f16 a[0:15], v[20:23], v[24:27], a[0:15]\n")) "s_waitcnt vmcnt(0)\n" ::"v"(gp), "s"(65536u) : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "memory");
        if (MODE == 13) asm volatile(REP16("s_mov_b32 m0, %2\n s_nop 0\n global_load_lds_dwordx4 %0, %1\n" REP4("v_mfma_f32_32x32x16_f16 a[0:15], v[20:23], v[24:27], a[0:15]\n")) "s_waitcnt vmcnt(0)\n" ::"v"(addr), "s"(g), "s"(65536u) : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "memory");
        // 4 MFMAs alone (same accumulator) for reference of 12/13
        if (MODE == 14) asm volatile(REP64("v_mfma_f32_32x32x16_f16 a[0:15], v[20:23], v[24:27], a[0:15]\n") ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
        // MFMA + 6 fma fillers / + 8
        if (MODE == 15) asm volatile(REP64("v_mfma_f32_32x32x16_f16 a[0:15], v[20:23], v[24:27], a[0:15]\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)::"v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
        if (MODE == 16) asm volatile(REP64("v_mfma_f32_32x32x16_f16 a[0:15], v[20:23], v[24:27], a[0:15]\n v_cvt_pk_f16_f32 %0, %0, %1\n v_max3_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)::"v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27");
        // the exp phase of the woven softmax as written: per pair two fma, two exp2 (of the previous pair), one conversion (of the pair
        // before that); registers rotate over three pairs
        if (MODE == 20 || MODE == 21) {
#define PAIR(a0, a1, b0, b1, c0, c1, d)                                                                                              \
    "v_fma_f32 " a0 ", v40, v41, v42\n v_fma_f32 " a1 ", v43, v41, v42\n v_exp_f32 " b0 ", " b0 "\n v_exp_f32 " b1 ", " b1 "\n v_cvt_pk_f16_f32 " d ", " c0 ", " c1 "\n"
#define MF "v_mfma_f32_32x32x16_f16 a[0:15], v[20:23], v[24:27], a[0:15]\n"
            if (MODE == 20)
                asm volatile(REP16(PAIR("v30", "v31", "v32", "v33", "v34", "v35", "v36") PAIR("v34", "v35", "v30", "v31", "v32", "v33", "v37")
                                   PAIR("v32", "v33", "v34", "v35", "v30", "v31", "v38")) ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v40", "v41", "v42", "v43");
            else
                asm volatile(REP16(MF PAIR("v30", "v31", "v32", "v33", "v34", "v35", "v36") MF PAIR("v34", "v35", "v30", "v31", "v32", "v33", "v37")
                                   MF PAIR("v32", "v33", "v34", "v35", "v30", "v31", "v38")) ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v40", "v41", "v42", "v43");
        }
        if (MODE == 22) asm volatile(REP64("v_max3_f32 %0, %0, %2, %3\n v_max3_f32 %1, %1, %2, %3\n") : "+v"(x0), "+v"(x1) : "v"(x2), "v"(x3));
        if (MODE == 23) asm volatile(REP64("v_cvt_pk_f16_f32 %0, %2, %3\n v_cvt_pk_f16_f32 %1, %2, %3\n") : "+v"(x0), "+v"(x1) : "v"(x2), "v"(x3));
        if (MODE == 24) asm volatile(REP64("v_exp_f32 %0, %2\n v_exp_f32 %1, %3\n") : "+v"(x0), "+v"(x1) : "v"(x2), "v"(x3));
        if (MODE == 25) asm volatile(REP64("v_fma_f32 %0, %2, %3, %2\n v_fma_f32 %1, %3, %2, %3\n") : "+v"(x0), "+v"(x1) : "v"(x2), "v"(x3));
        // one PV piece of the one-wave-per-SIMD attention loop as it is issued there: counted wait, MFMA on a fragment read three pieces
        // earlier (ring of four), two transpose reads for the piece three ahead, a softmax pair (2 fma, 2 exp2, 1 cvt, staggered)
#define PCE(acc, fr, nx, a0, a1, b0, b1, c0, c1, d)                                                                                 \
    "s_waitcnt lgkmcnt(4)\n v_mfma_f32_32x32x16_f16 " acc ", " fr ", v[60:63], " acc "\n ds_read_b64_tr_b16 " nx "\n"                 \
    "v_fma_f32 " a0 ", v40, v41, v42\n v_fma_f32 " a1 ", v43, v41, v42\n v_exp_f32 " b0 ", " b0 "\n v_exp_f32 " b1 ", " b1 "\n v_cvt_pk_f16_f32 " d ", " c0 ", " c1 "\n"
        if (MODE == 40 || MODE == 41 || MODE == 42) {
            asm volatile("ds_read_b64_tr_b16 v[44:45], %0\n ds_read_b64_tr_b16 v[46:47], %0 offset:2048\n ds_read_b64_tr_b16 v[48:49], %0 offset:4096\n ds_read_b64_tr_b16 v[50:51], %0 offset:6144\n"
                         "ds_read_b64_tr_b16 v[52:53], %0 offset:8192\n ds_read_b64_tr_b16 v[54:55], %0 offset:10240\n" ::"v"(addr) : "v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55");
            if (MODE == 40)
                asm volatile(REP16(PCE("a[0:15]", "v[44:47]", "v[56:57], %0 offset:12288\n ds_read_b64_tr_b16 v[58:59], %0 offset:14336", "v30", "v31", "v32", "v33", "v34", "v35", "v36")
                                   PCE("a[16:31]", "v[48:51]", "v[44:45], %0\n ds_read_b64_tr_b16 v[46:47], %0 offset:2048", "v34", "v35", "v30", "v31", "v32", "v33", "v37")
                                   PCE("a[32:47]", "v[52:55]", "v[48:49], %0 offset:4096\n ds_read_b64_tr_b16 v[50:51], %0 offset:6144", "v32", "v33", "v34", "v35", "v30", "v31", "v38")
                                   PCE("a[48:63]", "v[56:59]", "v[52:53], %0 offset:8192\n ds_read_b64_tr_b16 v[54:55], %0 offset:10240", "v30", "v31", "v32", "v33", "v34", "v35", "v36"))
                             "s_waitcnt lgkmcnt(0)\n" ::"v"(addr) : "v30","v31","v32","v33","v34","v35","v36","v37","v38","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63");
        }
#define F4 "v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
#define CLB "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31"
        // four rotating accumulation-register accumulators, 4 fma behind each MFMA
        if (MODE == 30) asm volatile(REP16("v_mfma_f32_32x32x16_f16 a[0:15], v[20:23], v[24:27], a[0:15]\n" F4 "v_mfma_f32_32x32x16_f16 a[16:31], v[20:23], v[24:27], a[16:31]\n" F4
                                           "v_mfma_f32_32x32x16_f16 a[32:47], v[20:23], v[24:27], a[32:47]\n" F4 "v_mfma_f32_32x32x16_f16 a[48:63], v[20:23], v[24:27], a[48:63]\n" F4) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)::CLB);
        // ordinary-register accumulators (two rotating)
        if (MODE == 31) asm volatile(REP16("v_mfma_f32_32x32x16_f16 v[64:79], v[20:23], v[24:27], v[64:79]\n" F4 "v_mfma_f32_32x32x16_f16 v[80:95], v[20:23], v[24:27], v[80:95]\n" F4
                                           "v_mfma_f32_32x32x16_f16 v[64:79], v[20:23], v[24:27], v[64:79]\n" F4 "v_mfma_f32_32x32x16_f16 v[80:95], v[20:23], v[24:27], v[80:95]\n" F4) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)::CLB, "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95");
        // B operand in accumulation registers, ordinary accumulators
        if (MODE == 32) asm volatile(REP16("v_mfma_f32_32x32x16_f16 v[64:79], v[20:23], a[24:27], v[64:79]\n" F4 "v_mfma_f32_32x32x16_f16 v[80:95], v[20:23], a[28:31], v[80:95]\n" F4
                                           "v_mfma_f32_32x32x16_f16 v[64:79], v[20:23], a[24:27], v[64:79]\n" F4 "v_mfma_f32_32x32x16_f16 v[80:95], v[20:23], a[28:31], v[80:95]\n" F4) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)::CLB, "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95");
        // A operand read from the LDS two MFMAs ahead, counted wait in front of the MFMA
        if (MODE == 33) asm volatile("ds_read_b128 v[20:23], %4\n ds_read_b128 v[24:27], %4 offset:4096\n"
                                     REP16("s_waitcnt lgkmcnt(1)\n v_mfma_f32_32x32x16_f16 a[0:15], v[20:23], v[28:31], a[0:15]\n ds_read_b128 v[20:23], %4 offset:8192\n" F4
                                           "s_waitcnt lgkmcnt(1)\n v_mfma_f32_32x32x16_f16 a[16:31], v[24:27], v[28:31], a[16:31]\n ds_read_b128 v[24:27], %4 offset:12288\n" F4
                                           "s_waitcnt lgkmcnt(1)\n v_mfma_f32_32x32x16_f16 a[32:47], v[20:23], v[28:31], a[32:47]\n ds_read_b128 v[20:23], %4\n" F4
                                           "s_waitcnt lgkmcnt(1)\n v_mfma_f32_32x32x16_f16 a[48:63], v[24:27], v[28:31], a[48:63]\n ds_read_b128 v[24:27], %4 offset:4096\n" F4) "s_waitcnt lgkmcnt(0)\n" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(addr) : CLB);
        // B operand written by VALU (conversions) shortly before
        if (MODE == 34) asm volatile(REP16("v_mfma_f32_32x32x16_f16 a[0:15], v[20:23], v[24:27], a[0:15]\n v_cvt_pk_f16_f32 v28, %0, %1\n v_cvt_pk_f16_f32 v29, %2, %3\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n"
                                           "v_mfma_f32_32x32x16_f16 a[16:31], v[20:23], v[28:31], a[16:31]\n v_cvt_pk_f16_f32 v24, %0, %1\n v_cvt_pk_f16_f32 v25, %2, %3\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                                           "v_mfma_f32_32x32x16_f16 a[32:47], v[20:23], v[24:27], a[32:47]\n v_cvt_pk_f16_f32 v30, %0, %1\n v_cvt_pk_f16_f32 v31, %2, %3\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n"
                                           "v_mfma_f32_32x32x16_f16 a[48:63], v[20:23], v[28:31], a[48:63]\n v_cvt_pk_f16_f32 v26, %0, %1\n v_cvt_pk_f16_f32 v27, %2, %3\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)::CLB);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (x0 + x1 + x2 + x3 == 12345.f) out[1] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int MODE>
static void run(const char* what, int units, unsigned long long* d_out, const char* g) {
    const int iters = 50;
    CK(hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    unsigned long long h = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((probe<MODE>), dim3(256), dim3(256), 131072, 0, d_out, g, iters);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost));
    }
    printf("  %-62s %7.1f cycles per unit\n", what, (double)h / iters / units);
}

int main() {
    unsigned long long* d_out;
    char* g;
    CK(hipMalloc(&d_out, 64));
    CK(hipMalloc(&g, (size_t)256 * 65536 + (1 << 20)));
    CK(hipMemset(g, 0, (size_t)256 * 65536 + (1 << 20)));
    printf("one wave per SIMD, all 256 CUs busy; unit = one repetition of the named group\n");
    run<0>("v_fma_f32 (independent)", 64, d_out, g);
    run<1>("v_exp_f32 (independent)", 64, d_out, g);
    run<2>("1 v_exp_f32 + 3 v_fma_f32", 16, d_out, g);
    run<3>("v_mfma_f32_32x32x16_f16, four accumulators", 64, d_out, g);
    run<14>("v_mfma_f32_32x32x16_f16, one accumulator", 64, d_out, g);
    run<4>("MFMA + 4 v_fma_f32", 64, d_out, g);
    run<15>("MFMA + 6 v_fma_f32", 64, d_out, g);
    run<5>("MFMA + 1 v_exp_f32 + 3 v_fma_f32", 64, d_out, g);
    run<6>("MFMA + 2 v_exp_f32 + 4 v_fma_f32", 64, d_out, g);
    run<16>("MFMA + v_cvt_pk_f16_f32 + v_max3_f32 + 2 v_fma_f32", 64, d_out, g);
    run<7>("ds_read_b128", 64, d_out, g);
    run<8>("ds_read_b64_tr_b16", 64, d_out, g);
    run<9>("MFMA + ds_read_b128", 64, d_out, g);
    run<10>("MFMA + 2 ds_read_b64_tr_b16", 64, d_out, g);
    run<11>("MFMA + ds_read_b128 + 4 v_fma_f32", 64, d_out, g);
    run<12>("LDS-DMA (64-bit lane addresses) + 4 MFMAs", 16, d_out, g);
    run<13>("LDS-DMA (scalar base + 32-bit lane offsets) + 4 MFMAs", 16, d_out, g);
    run<30>("MFMA + 4 v_fma_f32, four rotating accumulators", 64, d_out, g);
    run<31>("MFMA + 4 v_fma_f32, accumulators in ordinary registers", 64, d_out, g);
    run<32>("MFMA + 4 v_fma_f32, ordinary accumulators, B operand in acc registers", 64, d_out, g);
    run<33>("MFMA + 4 v_fma_f32, A operand from the LDS two MFMAs ahead", 64, d_out, g);
    run<34>("MFMA + 2 cvt_pk + 2 v_fma_f32, B operand written by the cvt", 64, d_out, g);
    run<40>("PV piece: wait + MFMA + 2 tr reads (3 ahead) + softmax pair", 64, d_out, g);
    run<25>("2 v_fma_f32 (no dependencies at all)", 64, d_out, g);
    run<24>("2 v_exp_f32 (no dependencies at all)", 64, d_out, g);
    run<23>("2 v_cvt_pk_f16_f32 (no dependencies)", 64, d_out, g);
    run<22>("2 v_max3_f32 (two chains)", 64, d_out, g);
    run<20>("softmax pair: 2 fma + 2 exp2 + 1 cvt_pk (staggered)", 48, d_out, g);
    run<21>("MFMA + softmax pair (5 instructions)", 48, d_out, g);
    return 0;
}
