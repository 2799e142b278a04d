/* The drop-in boundary exercised from plain C (gcc, no Python, no torch): include/deft_amd.h compiles as C, libdeft_amd.so links
 * against a C program, and deft_flatten_decode_f16 -- DeFTAttention.deft_flatten_forward's operator, tree_attention.py:599-661 --
 * on a hand-built TreeMetadata of a two-leaf tree matches a double-precision restatement written out below.
 *   gcc -std=c11 -O1 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include tests/c_abi/decode_from_c.c \
 *       -L deft_amd/lib -ldeft_amd -L /opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,deft_amd/lib -Wl,-rpath,/opt/rocm/lib -o decode_from_c
 * tests/test_gpu_parity.py::test_c_program_through_the_c_abi builds and runs it on the GPU box. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "deft_amd.h"

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
            return 2;                                                                      \
        }                                                                                  \
    } while (0)

static uint16_t f2h(float f) { /* round to nearest even; inputs here are small normals */
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u, mant = x & 0x7fffffu;
    int exp = (int)((x >> 23) & 0xff) - 127 + 15;
    if (exp <= 0) return (uint16_t)sign;
    if (exp >= 31) return (uint16_t)(sign | 0x7c00u);
    uint32_t h = ((uint32_t)exp << 10) | (mant >> 13);
    uint32_t rem = mant & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;
    return (uint16_t)(sign | h);
}
static float h2f(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1f, mant = h & 0x3ffu, x;
    if (exp == 0) {
        float v = (float)mant * (1.0f / 16777216.0f); /* 2^-24 */
        return (h & 0x8000u) ? -v : v;
    }
    x = sign | ((exp - 15 + 127) << 23) | (mant << 13);
    float f;
    memcpy(&f, &x, 4);
    return f;
}
static uint32_t rng_state = 12345u;
static float rnd(void) { /* a dyadic value in [-2, 2): exact in fp16 */
    rng_state = rng_state * 1664525u + 1013904223u;
    return (float)((int)((rng_state >> 16) & 0xff) - 128) / 64.0f;
}

enum { HQ = 4, HKV = 2, D = 128, NQ = 2, SLOTS = 512, NB = 4, P = 6, PREFIX = 150, LEN0 = 40, LEN1 = 7 };

int main(void) {
    if (deft_abi_version() != 2 || !deft_supported(HQ, HKV, D)) {
        fprintf(stderr, "abi / geometry\n");
        return 1;
    }
    /* the pool: [slot][K|V][Hkv][D] fp16 (memory_pool.py:61-66); q [nq][Hq][D] */
    static uint16_t pool[SLOTS][2][HKV][D], q[NQ][HQ][D], out[NQ][HQ][D];
    for (int s = 0; s < SLOTS; ++s)
        for (int kv = 0; kv < 2; ++kv)
            for (int h = 0; h < HKV; ++h)
                for (int d = 0; d < D; ++d) pool[s][kv][h][d] = f2h(rnd());
    for (int i = 0; i < NQ; ++i)
        for (int h = 0; h < HQ; ++h)
            for (int d = 0; d < D; ++d) q[i][h][d] = f2h(rnd());
    /* the tree: a 150-token prompt in slots 0..149; leaf 0 (query 0) holds 40 tokens in slots 200.., leaf 1 (query 1) 7 in 300..
     * TreeMetadata (tree_cache.py:618-881), written by hand: block 0 = prompt tokens 0..127, block 1 = the other 22 (both leaves see
     * them), block 2 = leaf 0's tokens, block 3 = leaf 1's */
    static int64_t block_q[P] = {0, 1, 0, 1, 0, 1}, block_q_cnts[NB] = {2, 2, 1, 1}, block_q_offset[NB] = {0, 2, 4, 5};
    static int64_t block_bitmasks[NB * 128], block_kv[NB * 128], block_lens[NB] = {128, PREFIX - 128, LEN0, LEN1};
    for (int b = 0; b < NB; ++b)
        for (int i = 0; i < 128; ++i) {
            const int live = i < block_lens[b];
            block_bitmasks[b * 128 + i] = live ? (b < 2 ? 3 : 1) : 0;
            block_kv[b * 128 + i] = !live ? -1 : b == 0 ? i : b == 1 ? 128 + i : b == 2 ? 200 + i : 300 + i;
        }
    void *d_pool, *d_q, *d_out, *d_md, *d_ws;
    const size_t md_bytes = sizeof block_q + sizeof block_q_cnts + sizeof block_q_offset + sizeof block_bitmasks + sizeof block_kv + sizeof block_lens;
    const size_t ws_bytes = deft_flatten_workspace_bytes(NB, P, NQ, HQ, HKV, D);
    CK(hipMalloc(&d_pool, sizeof pool));
    CK(hipMalloc(&d_q, sizeof q));
    CK(hipMalloc(&d_out, sizeof out));
    CK(hipMalloc(&d_md, md_bytes));
    CK(hipMalloc(&d_ws, ws_bytes ? ws_bytes : 1));
    CK(hipMemcpy(d_pool, pool, sizeof pool, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_q, q, sizeof q, hipMemcpyHostToDevice));
    char* m = (char*)d_md;
    int64_t *d_bq = (int64_t*)m, *d_cnt = d_bq + P, *d_off = d_cnt + NB, *d_bm = d_off + NB, *d_kv = d_bm + NB * 128, *d_len = d_kv + NB * 128;
    CK(hipMemcpy(d_bq, block_q, sizeof block_q, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_cnt, block_q_cnts, sizeof block_q_cnts, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_off, block_q_offset, sizeof block_q_offset, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_bm, block_bitmasks, sizeof block_bitmasks, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_kv, block_kv, sizeof block_kv, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_len, block_lens, sizeof block_lens, hipMemcpyHostToDevice));
    const uint16_t* kb = (const uint16_t*)d_pool;
    const int rc = deft_flatten_decode_f16(d_q, HQ * D, D, kb, kb + HKV * D, 2 * HKV * D, D, d_out, HQ * D, D, d_bq, d_cnt, d_off, d_bm, d_kv,
                                           d_len, NB, P, NQ, HQ, HKV, D, 1.0f / sqrtf((float)D), NULL, d_ws, ws_bytes, NULL);
    if (rc != DEFT_OK) {
        fprintf(stderr, "deft_flatten_decode_f16: %d (%s)\n", rc, deft_last_error());
        return 1;
    }
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out, d_out, sizeof out, hipMemcpyDeviceToHost));
    /* the restatement: query i attends to the prompt and to its own leaf, softmax in double */
    double worst = 0.0;
    for (int i = 0; i < NQ; ++i)
        for (int h = 0; h < HQ; ++h) {
            const int kvh = h / (HQ / HKV), own = i == 0 ? LEN0 : LEN1, base = i == 0 ? 200 : 300, n = PREFIX + own;
            static double s[PREFIX + LEN0];
            double mx = -1e300, den = 0.0;
            for (int j = 0; j < n; ++j) {
                const int slot = j < PREFIX ? j : base + (j - PREFIX);
                double dot = 0.0;
                for (int d = 0; d < D; ++d) dot += (double)h2f(q[i][h][d]) * (double)h2f(pool[slot][0][kvh][d]);
                s[j] = dot / sqrt((double)D);
                if (s[j] > mx) mx = s[j];
            }
            for (int j = 0; j < n; ++j) {
                s[j] = exp(s[j] - mx);
                den += s[j];
            }
            for (int d = 0; d < D; ++d) {
                double acc = 0.0;
                for (int j = 0; j < n; ++j) acc += s[j] * (double)h2f(pool[j < PREFIX ? j : base + (j - PREFIX)][1][kvh][d]);
                const double err = fabs(acc / den - (double)h2f(out[i][h][d]));
                if (err > worst) worst = err;
            }
        }
    printf("deft_flatten_decode_f16 from C: %d queries x %d heads, worst |err| vs double = %.3e\n", NQ, HQ, worst);
    hipFree(d_pool), hipFree(d_q), hipFree(d_out), hipFree(d_md), hipFree(d_ws);
    return worst < 1e-3 ? 0 : 1; /* the tolerance of the Python parity tests */
}
