// DeFT paged tree-attention decode kernels for MI355X (gfx950 / CDNA4, wave64).
//
// What the reference computes (DeFT/deft/layers/attention/tree_attention.py):
//   stage 1  per (head, KV tile): S = Q K^T / sqrt(D), mask by the per-slot query
//            bitmask, softmax over the tile, partial = P V / sum, lse = max + log(sum)
//            (Flatten: kernel2 :859-976, one 128-slot block per program;
//             Node: :170-293, a serial walk over the node in 16-token tiles)
//   stage 2  merge all partial rows of a query by their log-sum-exp (:296-546)
//
// How it is laid out here:
//   * one workgroup (4 waves) owns one (KV tile of <=128 pool slots, KV head) and loops
//     over the Hq/Hkv query heads that share that KV head, so every K/V byte is read
//     from HBM once per tile (the reference re-reads it per query head, :894).
//   * K and V rows are gathered through the slot list into LDS with 16-byte
//     per-lane loads (16 lanes cover one 256-byte row).  K is stored with an XOR
//     swizzle of its 16-byte chunks so the MFMA A-fragment reads are conflict-free.
//   * S^T = K Q^T on v_mfma_f32_32x32x16_f16: keys are the M dimension, so K
//     fragments are 16-byte row pieces straight from LDS and Q fragments are
//     16-byte row pieces straight from global memory; each lane ends up holding
//     16 key scores of ONE query, so the row max / row sum are in-lane reductions
//     plus one cross-half shuffle and a 4-wave exchange through LDS.
//   * P (fp16) goes through LDS once; O^T = V^T P^T on the same MFMA, each wave
//     producing 32 of the D output columns for all 32 queries.
//   * Node mode reuses the same kernel: a prep kernel cuts every node entry into
//     128-slot tiles (all queries of the entry see all slots), which removes the
//     reference's serial walk over long nodes.
//   * the merge is a deterministic gather (no atomics): true max, fp32 accumulate,
//     one fp16 rounding.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "common.h"

namespace deft {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef uint32_t uintx4 __attribute__((ext_vector_type(4)));

constexpr int TILE = DEFT_BLOCK_LEN;  // 128 KV slots per tile
constexpr int MQ = DEFT_MAX_Q_LEN;    // 32 query rows per tile
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

struct Stage1Params {
    const _Float16* q;
    int64_t q_st, q_sh;
    const _Float16* k;
    const _Float16* v;
    int64_t kv_ss, kv_sh;
    // Flatten sources (MODE 0)
    const int64_t* block_q;
    const int64_t* block_q_cnts;
    const int64_t* block_q_offset;
    const int64_t* block_bitmasks;
    const int64_t* block_kv;
    const int64_t* block_lens;
    // Node sources (MODE 1 / node plan)
    const int64_t* node_kv;
    const int64_t* node_q;
    const int64_t* node_kv_offset;
    const int64_t* node_kv_len;
    const int64_t* node_q_offset;
    const int64_t* node_q_len;
    const int32_t* desc;  // [tiles][8] = kv_begin, len, q_begin, cnt, prow, -, -, -
    // outputs
    float* partial_o;
    float* partial_lse;
    int32_t* row_q;
    int64_t rows;  // partial rows per head (stride of partial_o / partial_lse)
    int Hkv, G;
    float scale_log2e;
    int ablate;  // experiments build only: profiling bits of the head_dim-128 kernel (stage1_np.h)
};

template <int D>
struct Stage1Smem {
    static constexpr int K_OFF = 0;
    static constexpr int V_OFF = K_OFF + TILE * D * 2;
    static constexpr int P_OFF = V_OFF + TILE * D * 2;
    static constexpr int MASK_OFF = P_OFF + MQ * TILE * 2;
    static constexpr int WMAX_OFF = MASK_OFF + TILE * 4;
    static constexpr int WSUM_OFF = WMAX_OFF + 4 * MQ * 4;
    static constexpr int SLOT_OFF = WSUM_OFF + 4 * MQ * 4;
    static constexpr int BYTES = SLOT_OFF + TILE * 4;
};

// ---------------------------------------------------------------------------
// stage 1
// ---------------------------------------------------------------------------
template <int D, int MODE>
__global__ __launch_bounds__(256, 2) void stage1_kernel(Stage1Params p) {
    constexpr int CH = D / 8;    // 16-byte chunks per K/V row
    constexpr int KS = D / 16;   // MFMA k-steps of S^T = K Q^T
    constexpr int MB = (D + 31) / 32;  // 32-column output blocks of O^T (head_dim 16: one block, half of it unused)
    using SM = Stage1Smem<D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* sK = reinterpret_cast<_Float16*>(smem + SM::K_OFF);
    _Float16* sV = reinterpret_cast<_Float16*>(smem + SM::V_OFF);
    _Float16* sP = reinterpret_cast<_Float16*>(smem + SM::P_OFF);
    uint32_t* sMask = reinterpret_cast<uint32_t*>(smem + SM::MASK_OFF);
    float* sWmax = reinterpret_cast<float*>(smem + SM::WMAX_OFF);
    float* sWsum = reinterpret_cast<float*>(smem + SM::WSUM_OFF);
    int* sSlot = reinterpret_cast<int*>(smem + SM::SLOT_OFF);

    const int tid = threadIdx.x;
    const int w = tid >> 6;
    const int l = tid & 63;
    const int c = l & 31;  // query row owned by this lane in the S^T / O^T fragments
    const int h = l >> 5;
    const int tile = blockIdx.x / p.Hkv;
    const int kvh = blockIdx.x - tile * p.Hkv;

    // ---- tile descriptor (wave-uniform) -------------------------------------
    int len, cnt, prow;
    const int64_t* slots;
    const int64_t* masks;
    const int64_t* qrows;
    if (MODE == 0) {
        len = (int)p.block_lens[tile];
        cnt = (int)p.block_q_cnts[tile];
        prow = (int)p.block_q_offset[tile];
        slots = p.block_kv + (int64_t)tile * TILE;
        masks = p.block_bitmasks + (int64_t)tile * TILE;
        qrows = p.block_q + prow;
    } else {
        const int32_t* d = p.desc + (int64_t)tile * 8;
        len = d[1];
        if (len <= 0) return;
        cnt = d[3];
        prow = d[4];
        slots = p.node_kv + d[0];
        masks = nullptr;
        qrows = p.node_q + d[2];
    }

    // ---- slot list and bitmasks -> LDS ---------------------------------------
    // Padded positions (>= len) re-read the tile's first slot with an empty mask:
    // their probability is exactly 0 and the row they alias is valid, finite data.
    if (tid < TILE) {
        const bool live = tid < len;
        sSlot[tid] = (int)slots[live ? tid : 0];
        uint32_t m = 0u;
        if (live) m = (MODE == 0) ? (uint32_t)masks[tid] : 0xffffffffu;
        sMask[tid] = m;
    }
    __syncthreads();

    // ---- gather K and V rows of this KV head into LDS ---------------------------
    // wave w stages keys [32w, 32w+32): 16-byte pieces, CH lanes per row.
    {
        constexpr int ITER = 32 * CH / 64;
        uintx4 kreg[ITER], vreg[ITER];
#pragma unroll
        for (int i = 0; i < ITER; ++i) {
            const int idx = i * 64 + l;
            const int key = 32 * w + idx / CH;
            const int chunk = idx % CH;
            const int64_t off = (int64_t)sSlot[key] * p.kv_ss + (int64_t)kvh * p.kv_sh + chunk * 8;
            kreg[i] = *reinterpret_cast<const uintx4*>(p.k + off);
            vreg[i] = *reinterpret_cast<const uintx4*>(p.v + off);
        }
#pragma unroll
        for (int i = 0; i < ITER; ++i) {
            const int idx = i * 64 + l;
            const int key = 32 * w + idx / CH;
            const int chunk = idx % CH;
            const int pos = chunk ^ (key & (CH - 1));
            *reinterpret_cast<uintx4*>(sK + key * D + pos * 8) = kreg[i];
            *reinterpret_cast<uintx4*>(sV + key * D + chunk * 8) = vreg[i];
        }
    }

    const bool qvalid = c < cnt;
    const int64_t qrow = qvalid ? qrows[c] : 0;
    for (int g = 0; g < p.G; ++g) {
        const int hq = kvh * p.G + g;

        // ---- Q fragments (B operand of S^T): 8 halfs of query row c per k-step ----
        half8 qf[KS];
        {
            const _Float16* qp = p.q + qrow * p.q_st + (int64_t)hq * p.q_sh + 8 * h;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                qf[ks] = qvalid ? *reinterpret_cast<const half8*>(qp + 16 * ks) : z;
            }
        }
        __syncthreads();  // K/V/masks staged (g == 0); sP / sWmax / sWsum free again (g > 0)

        // ---- S^T[key][query] for this wave's 32 keys ------------------------------
        floatx16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        {
            const int key = 32 * w + c;
            const _Float16* krow = sK + key * D;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int pos = (2 * ks + h) ^ (key & (CH - 1));
                const half8 a = *reinterpret_cast<const half8*>(krow + pos * 8);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qf[ks], acc, 0, 0, 0);
            }
        }
        // acc[r] = S^T[key = 32w + (r&3) + 8*(r>>2) + 4h][query = c]

        // ---- scale, mask, row max ---------------------------------------------------
        float s[16];
        float mx = -INFINITY;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const uintx4 m4 = *reinterpret_cast<const uintx4*>(sMask + 32 * w + 8 * g4 + 4 * h);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * g4 + i;
                const bool vis = qvalid && ((m4[i] >> c) & 1u);
                s[r] = vis ? acc[r] * p.scale_log2e : -INFINITY;
                mx = fmaxf(mx, s[r]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (h == 0) sWmax[w * MQ + c] = mx;
        __syncthreads();
        const float m = fmaxf(fmaxf(sWmax[c], sWmax[MQ + c]), fmaxf(sWmax[2 * MQ + c], sWmax[3 * MQ + c]));
        const float msafe = (m == -INFINITY) ? 0.f : m;

        // ---- P = exp2(s - m) as fp16, row sums of the ROUNDED values -------------------
        float sum = 0.f;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            half4 p4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const _Float16 ph = (_Float16)exp2f(s[4 * g4 + i] - msafe);
                p4[i] = ph;
                sum += (float)ph;
            }
            // keys 32w + 8*g4 + 4h + (0..3) of query row c; 16-byte chunks XOR-swizzled by row
            const int pos = (4 * w + g4) ^ (c & 15);
            *reinterpret_cast<half4*>(sP + c * TILE + pos * 8 + 4 * h) = p4;
        }
        sum += __shfl_xor(sum, 32);
        if (h == 0) sWsum[w * MQ + c] = sum;
        __syncthreads();

        // ---- O^T[d][query] = sum_key V[key][d] * P[query][key], 32 d-columns per wave ----
        if (w < MB) {
            floatx16 o;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = 0.f;
            const _Float16* vcol = sV + 32 * w + c;
            const _Float16* prow_p = sP + c * TILE;
#pragma unroll
            for (int ks = 0; ks < TILE / 16; ++ks) {
                half8 a;
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = (32 * w + c < D) ? vcol[(16 * ks + 8 * h + j) * D] : (_Float16)0.f;
                const half8 b = *reinterpret_cast<const half8*>(prow_p + (((2 * ks + h) ^ (c & 15)) * 8));
                o = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, o, 0, 0, 0);
            }
            // o[r] = O^T[d = 32w + (r&3) + 8*(r>>2) + 4h][query = c]
            const float lsum = sWsum[c] + sWsum[MQ + c] + sWsum[2 * MQ + c] + sWsum[3 * MQ + c];
            const float inv = lsum > 0.f ? 1.f / lsum : 0.f;
            if (qvalid) {
                const int64_t prow_idx = (int64_t)hq * p.rows + prow + c;
                float* po = p.partial_o + prow_idx * D + 32 * w + 4 * h;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    if (32 * w + 8 * g4 + 4 * h >= D) continue;  // (head_dim 16: columns 16..31 of the MFMA block do not exist)
                    floatx4 v4 = {o[4 * g4] * inv, o[4 * g4 + 1] * inv, o[4 * g4 + 2] * inv, o[4 * g4 + 3] * inv};
                    *reinterpret_cast<floatx4*>(po + 8 * g4) = v4;
                }
                if (w == 0 && h == 0) {
                    p.partial_lse[prow_idx] = (lsum > 0.f) ? (m + log2f(lsum)) * LN2 : -INFINITY;
                    if (hq == 0) p.row_q[prow + c] = (int32_t)qrow;
                }
            }
        }
    }
}

}  // namespace deft
#include "plan_records.h"
#include "plan_kernels.h"
#include "tree_plan.h"
#include "window.h"
#include "merge.h"
#include "stage1_np.h"
#include "prefill.h"
namespace deft {

// ---------------------------------------------------------------------------
// Node mode: cut entries into 128-slot tiles (one workgroup, once per call)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void node_prep_kernel(const int64_t* node_kv_offset, const int64_t* node_kv_len,
                                                         const int64_t* node_q_offset, const int64_t* node_q_len, int NE,
                                                         int64_t max_tiles, int64_t max_rows, int32_t* desc,
                                                         int32_t* row_q) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* sTiles = reinterpret_cast<int*>(smem);  // [NE+1] exclusive scan of tiles per entry
    int* sRows = sTiles + (NE + 1);              // [NE+1] exclusive scan of partial rows per entry
    const int tid = threadIdx.x;
    for (int e = tid; e < NE; e += blockDim.x) {
        const int nt = (int)((node_kv_len[e] + TILE - 1) / TILE);
        sTiles[e + 1] = nt;
        sRows[e + 1] = nt * (int)node_q_len[e];
    }
    for (int64_t i = tid; i < max_rows; i += blockDim.x) row_q[i] = -1;
    __syncthreads();
    if (tid == 0) {
        sTiles[0] = 0;
        sRows[0] = 0;
        for (int e = 0; e < NE; ++e) {
            sTiles[e + 1] += sTiles[e];
            sRows[e + 1] += sRows[e];
        }
    }
    __syncthreads();
    const int total = sTiles[NE];
    for (int e = tid; e < NE; e += blockDim.x) {
        const int t0 = sTiles[e];
        const int nt = sTiles[e + 1] - t0;
        const int kv0 = (int)node_kv_offset[e];
        const int kvl = (int)node_kv_len[e];
        const int q0 = (int)node_q_offset[e];
        const int ql = (int)node_q_len[e];
        for (int t = 0; t < nt; ++t) {
            int32_t* d = desc + (int64_t)(t0 + t) * 8;
            d[0] = kv0 + t * TILE;
            d[1] = min(TILE, kvl - t * TILE);
            d[2] = q0;
            d[3] = ql;
            d[4] = sRows[e] + t * ql;
            d[5] = d[6] = d[7] = 0;
        }
    }
    for (int64_t t = total + tid; t < max_tiles; t += blockDim.x) desc[t * 8 + 1] = 0;
}

// ---------------------------------------------------------------------------
// stage 2 as its own launch (merge.h): one workgroup per (head, four queries), a wave per query; the grid is laid out so
// that a head is merged on the XCD whose L2 holds its partial rows.
// ---------------------------------------------------------------------------
constexpr int MERGE_LIST_CAP = 4096;  // row ids per wave in LDS; longer lists are merged window by window
template <int D>
__global__ __launch_bounds__(256) void merge_kernel(const float* partial_o, const float* partial_lse, const int32_t* row_q,
                                                     int64_t rows, _Float16* out, int64_t o_st, int64_t o_sh, int Hq, int cap,
                                                     int lists, const int32_t* qoff, const int32_t* qlist, const int32_t* qinl,
                                                     int nq_total, int hgroup) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // (uniform: buffer descriptors stay in SGPRs)
    // EXPERIMENT (XCD alignment): a workgroup = one HEAD x four queries, head fastest in the grid: workgroup b runs on XCD b % 8 = the
    // XCD whose stage-1 workgroups (KV head = b % 8 for MHA) wrote this head's partial rows
    // Head fastest in the grid, four QUERIES of one head per workgroup: workgroup b runs on XCD b % 8, and the heads are dealt
    // so that this is the XCD whose stage-1 workgroups wrote the head's partial rows (stage-1 item i = (record, KV head or head
    // pair i % HP) runs on XCD i % 8 for grids that are multiples of 8) -- the rows are still in that XCD's L2 (measured round 3:
    // Medusa-64 14.7 -> 13.4 us per layer; the same grid with heads rotated by 3 gains nothing).  hgroup = query heads per item.
    const int ng = Hq / hgroup, bx = blockIdx.x;
    // (round 5 measured NH = 2 / 4 heads of one XCD per wave -- a quarter of the workgroups to dispatch, the same round trips per
    //  wave: +1.5 / +4.6 us per layer on every workload, profiles/r5_merge_heads_per_wave_negative.txt: the merge is bound by what ONE
    //  wave executes, not by dispatch)
    const int hq = (bx % ng) * hgroup + bx / ng;
    const int q = blockIdx.y * 4 + w;
    if (q >= nq_total) return;
    _Float16* out_q = out + (int64_t)q * o_st;
    int* list = reinterpret_cast<int*>(smem) + w * cap;
#ifdef DEFT_EXPERIMENTS
    if (lists >= 2) {  // TIMING ONLY (wrong results): row ids computed, not loaded -- what a merge without its first dependent round trip costs
        const int n = lists - 1;
        if (lane < n) list[lane] = (int)(((int64_t)q * n + lane) % rows);
        __builtin_amdgcn_wave_barrier();
        merge_heads_wave<D, 0, 1>(partial_o, partial_lse, row_q, rows, q, hq, 1, Hq, list, cap, n, out_q, o_sh, lane);
        return;
    }
#endif
    if (lists) {  // the plan lists every query's rows: ONE round trip for {count, rows}, one for the rows themselves
        const int mine = lane < 16 ? qinl[q * 16 + lane] : 0;
        const int n = __builtin_amdgcn_readfirstlane(mine);
        if (n <= 15 && n <= cap) {
            if (lane >= 1 && lane <= n) list[lane - 1] = mine;
            __builtin_amdgcn_wave_barrier();
            merge_heads_wave<D, 0, 1>(partial_o, partial_lse, row_q, rows, q, hq, 1, Hq, list, cap, n, out_q, o_sh, lane);
        } else if (n <= cap) {  // a longer list: the query's stretch of qlist, staged in LDS with one coalesced read
            const int o = qoff[q];
            for (int j = lane; j < n; j += 64) list[j] = qlist[o + j];
            __builtin_amdgcn_wave_barrier();
            merge_heads_wave<D, 0, 1>(partial_o, partial_lse, row_q, rows, q, hq, 1, Hq, list, cap, n, out_q, o_sh, lane);
        } else {  // longer than the LDS area: read in place
            const int o = qoff[q];
            merge_heads_wave<D, 0, 1>(partial_o, partial_lse, row_q, rows, q, hq, 1, Hq, const_cast<int*>(qlist) + o, n, n, out_q, o_sh, lane);
        }
        return;
    }
    int have = scan_rows_wave(row_q, 0, rows < cap ? rows : cap, q, list, cap, lane);
    if (rows > cap) have = -1;
    __builtin_amdgcn_wave_barrier();
    merge_heads_wave<D, 0, 1>(partial_o, partial_lse, row_q, rows, q, hq, 1, Hq, list, cap, have, out_q, o_sh, lane);
}

// Long row lists (a long shared prefix under few queries: 64k tokens x 8 branches is 66 rows per query, and only
// nq x Hq / 4 workgroups): ONE (query, head) pair per workgroup, its row list dealt round-robin to the four waves, each
// folding a quarter (merge_accumulate: 8 rows per round trip, so 66 rows cost 3 trips instead of 9), then one combine of the
// four (m, L, acc) states through LDS.  Deterministic (the deal is a function of the list); its fp32 order differs from
// merge_kernel's, so a launch uses one or the other as a function of its SHAPE only (launch_merge).
template <int D>
__global__ __launch_bounds__(256) void merge_coop_kernel(const float* partial_o, const float* partial_lse, int64_t rows, _Float16* out,
                                                          int64_t o_st, int64_t o_sh, int cap, const int32_t* qoff,
                                                          const int32_t* qlist, const int32_t* qinl, int Hq, int hgroup) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LPR = D / 4;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int ng = Hq / hgroup, bx = blockIdx.x;  // head fastest, on the XCD that wrote its rows (see merge_kernel)
    const int q = blockIdx.y, hq = (bx % ng) * hgroup + bx / ng;
    int* sub = reinterpret_cast<int*>(smem) + w * cap;                         // this wave's quarter of the list
    float* xacc = reinterpret_cast<float*>(smem + sizeof(int) * 4 * (size_t)cap);  // [4][D]
    float* xml = xacc + 4 * D;                                                 // [4][2]: m, L
    const int mine = lane < 16 ? qinl[q * 16 + lane] : 0;
    const int n = __builtin_amdgcn_readfirstlane(mine);
    if (n <= 8) {  // one round trip's worth: wave 0 alone, in merge_kernel's order (same bits), no combine
        if (w == 0) {
            if (lane >= 1 && lane <= n) sub[lane - 1] = mine;
            __builtin_amdgcn_wave_barrier();
            merge_heads_wave<D, 0, 1>(partial_o, partial_lse, nullptr, rows, q, hq, 1, hq + 1, sub, cap, n, out + (int64_t)q * o_st, o_sh, lane);
        }
        return;
    }
    int nw = 0;  // rows of this wave: list entries w, w + 4, ...
    if (n <= 15) {
        const int src = 1 + w + 4 * lane;  // lane i takes entry w + 4 i
        const int v = __shfl(mine, src < 16 ? src : 0);
        nw = (n - w + 3) / 4;
        if (lane < nw) sub[lane] = v;
    } else {
        const int o = qoff[q];
        nw = (n - w + 3) / 4;
        for (int j = lane; j < nw && j < cap; j += 64) sub[j] = qlist[o + w + 4 * j];
        if (nw > cap) nw = cap;  // (launch_merge sizes cap >= rows / 4 + 1: never)
    }
    __builtin_amdgcn_wave_barrier();
    __amdgpu_buffer_rsrc_t po[1] = {make_rsrc(partial_o + (int64_t)hq * rows * D)};
    __amdgpu_buffer_rsrc_t ls[1] = {make_rsrc(partial_lse + (int64_t)hq * rows)};
    MergeState st[1];
    if (nw > 0) merge_accumulate<D, 0, 1>(st, po, ls, sub, nw, lane);
    floatx4 a = st[0].acc;
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] += __shfl_xor(a[e], off);
    if (lane < LPR) *reinterpret_cast<floatx4*>(xacc + w * D + 4 * lane) = a;
    if (lane == 0) {
        xml[2 * w] = st[0].m;
        xml[2 * w + 1] = st[0].L;
    }
    __syncthreads();
    if (w == 0 && lane < LPR) {
        float m[4], M = -INFINITY;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            m[i] = xml[2 * i];
            M = fmaxf(M, m[i]);
        }
        float L = 0.f;
        floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float f = (m[i] == -INFINITY) ? 0.f : __expf(m[i] - M);
            L += f * xml[2 * i + 1];
            acc += *reinterpret_cast<const floatx4*>(xacc + i * D + 4 * lane) * f;
        }
        const float inv = L > 0.f ? 1.f / L : 0.f;
        _Float16* dst = out + (int64_t)q * o_st + (int64_t)hq * o_sh;
        half2v lo = {(_Float16)(acc[0] * inv), (_Float16)(acc[1] * inv)};
        half2v hi = {(_Float16)(acc[2] * inv), (_Float16)(acc[3] * inv)};
        *reinterpret_cast<half2v*>(dst + 4 * lane) = lo;
        *reinterpret_cast<half2v*>(dst + 4 * lane + 2) = hi;
    }
}

// ---------------------------------------------------------------------------
// paged KV append: one 16-byte piece per thread
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kv_append_kernel(_Float16* k_base, _Float16* v_base, int64_t kv_ss, int64_t kv_sh,
                                                         const int32_t* cache_loc, const _Float16* k_new,
                                                         const _Float16* v_new, int64_t new_st, int n, int Hkv, int D) {
    const int ch_per_head = D / 8;
    const int ch_per_tok = Hkv * ch_per_head;
    const int64_t total = (int64_t)n * ch_per_tok;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int tok = (int)(i / ch_per_tok);
        const int rem = (int)(i - (int64_t)tok * ch_per_tok);
        const int hd = rem / ch_per_head;
        const int ch = rem - hd * ch_per_head;
        const int64_t src = (int64_t)tok * new_st + (int64_t)hd * D + ch * 8;
        const int64_t dst = (int64_t)cache_loc[tok] * kv_ss + (int64_t)hd * kv_sh + ch * 8;
        *reinterpret_cast<uintx4*>(k_base + dst) = *reinterpret_cast<const uintx4*>(k_new + src);
        *reinterpret_cast<uintx4*>(v_base + dst) = *reinterpret_cast<const uintx4*>(v_new + src);
    }
}

// ---------------------------------------------------------------------------
// host-side launch helpers
// ---------------------------------------------------------------------------
// Experiment knobs exist in the experiments build only (`make exp`, -DDEFT_EXPERIMENTS -> libdeft_amd_exp.so): there a
// knob reads its environment variable at every call, so that one process can A/B settings on the same pools
// (tools/ab.py).  The shipped library reads no environment: every knob is its default.
#ifdef DEFT_EXPERIMENTS
static int knob(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
static unsigned long long* g_dbg = nullptr;  // per-workgroup time stamps, see deft_debug_set_buffer
// Test hook (deft_debug_plan_form, experiments build only): force the plan kernels' fallback forms, which are otherwise
// reached only by trees whose run tables exceed the LDS.
static int g_plan_serial = 0, g_plan_runcap = 0;
#else
#ifdef DEFT_FIXED_KNOBS
// Rule variants of the SHIPPED code (`make rules KNOBS="DEFT_NP_CHUNK=5;DEFT_NP_UNION=2" NAME=x` -> libdeft_amd_rules_x.so): the
// knobs named in the compile-time string take its values, everything else is the shipped build -- no environment, no time stamps,
// no ablation branches.  tools/ab_rules.sh compares such builds on one box: the experiments build is 1.5-2 us per layer slower
// than the shipped one and does not always rank rules the same way (head_dim-64 union groups of 3: -0.9 us there, +0.4 here).
static int knob(const char* name, int dflt) {
    const size_t n = strlen(name);
    for (const char* p = DEFT_FIXED_KNOBS; p && *p; p = strchr(p, ';') ? strchr(p, ';') + 1 : nullptr)
        if (!strncmp(p, name, n) && p[n] == '=') return atoi(p + n + 1);
    return dflt;
}
#else
static inline int knob(const char*, int dflt) { return dflt; }
#endif
static constexpr unsigned long long* g_dbg = nullptr;
static constexpr int g_plan_serial = 0, g_plan_runcap = 0;
#endif

static int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return DEFT_EHIP;
    }
    return DEFT_OK;
}

// Per-device facts: the CU count, and which kernels already had their dynamic-LDS cap raised (a function attribute
// is per device in HIP).
struct DeviceState {
    int cus = 0;
    unsigned attrs = 0;
};
static DeviceState& dev_state() {
    static DeviceState st[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    return st[dev];
}
static int num_cus() {
    DeviceState& d = dev_state();
    if (d.cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            d.cus = prop.multiProcessorCount;
        if (d.cus <= 0) d.cus = 256;
    }
    return d.cus;
}
enum : unsigned { ATTR_NP = 4,  // (bits 0-1, 8-11: the head_dim 64 / 32 / 16 kernels)
                  ATTR_UNITS = 8, ATTR_NODE_UNITS = 16, ATTR_PREFILL = 32, ATTR_TREE = 64, ATTR_NP_ROPE = 128, ATTR_NP_T = 8192, ATTR_NP_ROPE_T = 16384, ATTR_NP_HD2 = 32768, ATTR_NP_HD2_T = 65536, ATTR_PREFILL_64 = 1u << 17, ATTR_NP_DYN = 1u << 18, ATTR_NP_ROPE_DYN = 1u << 19, ATTR_NP_HD2_DYN = 1u << 20 };
static int raise_lds(const void* fn, int bytes, unsigned bit, const char* what) {  // idempotent; races are harmless
    DeviceState& d = dev_state();
    if (d.attrs & bit) return DEFT_OK;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        set_error("hipFuncSetAttribute(%s): %s", what, hipGetErrorString(e));
        return DEFT_EHIP;
    }
    d.attrs |= bit;
    return DEFT_OK;
}

// head_dim 16 / 32 / 64: one workgroup per (tile, KV head)
template <int D, int MODE>
static int launch_stage1_small(const Stage1Params& p, int64_t tiles, hipStream_t stream) {
    using SM = Stage1Smem<D>;
    constexpr unsigned bit = (D == 64 ? 1u : (D == 32 ? 256u : 1024u)) << MODE;
    int rc = raise_lds(reinterpret_cast<const void*>(&stage1_kernel<D, MODE>), SM::BYTES, bit, "stage1");
    if (rc) return rc;
    const int64_t grid = tiles * p.Hkv;
    if (grid <= 0) return DEFT_OK;
    if (grid > 0x7fffffffLL) {
        set_error("stage1 grid too large: %lld", (long long)grid);
        return DEFT_EINVAL;
    }
    hipLaunchKernelGGL((stage1_kernel<D, MODE>), dim3((unsigned)grid), dim3(256), SM::BYTES, stream, p);
    return check_launch("stage1 launch");
}
template <int MODE>
static int launch_stage1_d64(int D, const Stage1Params& p, int64_t tiles, hipStream_t stream) {
    if (D == 64) return launch_stage1_small<64, MODE>(p, tiles, stream);
    if (D == 32) return launch_stage1_small<32, MODE>(p, tiles, stream);
    if (D == 16) return launch_stage1_small<16, MODE>(p, tiles, stream);
    set_error("unsupported head_dim %d", D);
    return DEFT_EUNSUPPORTED;
}

// Optional fused paged append (deft_*_decode_append_f16): this step's new K/V rows and their pool slots.
struct AppendArgs {
    const _Float16* k_new = nullptr;
    const _Float16* v_new = nullptr;
    const int32_t* cache_loc = nullptr;
    int64_t new_st = 0;
    int n_new = 0;
    // fused rotary embedding (optional, with the fused append only): cos | sin row of every new token, [n_new][D] fp32
    const float* cos_sin = nullptr;
};

static UnitList unit_list(const PlanView& pv) {
    UnitList ul;
    ul.src = pv.units;
    ul.aux = pv.units + pv.cap;
    ul.pass = pv.units + 2 * pv.cap;
    ul.flags = pv.units + 3 * pv.cap;
    ul.prow = pv.units + 4 * pv.cap;
    ul.perm = pv.units + 5 * pv.cap;
    ul.ch_n = pv.units + 6 * pv.cap;
    ul.ch_fb = pv.units + 7 * pv.cap;
    ul.gn = pv.units + 8 * pv.cap;
    ul.gq = pv.units + 9 * pv.cap;
    ul.grow = pv.units + (9 + DEFT_UNION_CAP) * pv.cap;
    return ul;
}

// Per-query row lists of a plan (plan_kernels.h): after the records kernel has written row_q.
static int launch_qrows(const PlanView& pv, hipStream_t stream) {
    const int64_t rows = pv.rows;
    if (rows <= 0 || rows > QROWS_MAX) return DEFT_OK;  // hdr[HDR_QLISTS] stays 0: the merge scans row_q itself
    if (rows <= QROWS_FUSED_MAX) {  // one launch for the usual sizes
        hipLaunchKernelGGL(qrows_fused_kernel, dim3(1), dim3(1024), 0, stream, pv.row_q, (int)rows, pv.qoff, pv.qlist, pv.qinl, pv.hdr);
        return check_launch("qrows launch");
    }
    hipLaunchKernelGGL(qrows_hist_kernel, dim3(1), dim3(1024), sizeof(int) * (size_t)rows, stream, pv.row_q, (int)rows, pv.qoff, pv.hdr);
    int rc = check_launch("qrows hist launch");
    if (rc) return rc;
    hipLaunchKernelGGL(qrows_fill_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, pv.row_q, (int)rows, pv.qoff, pv.qlist,
                       pv.qinl);
    return check_launch("qrows fill launch");
}

static int np_chunk_knob() { return knob("DEFT_NP_CHUNK", 0); }  // tiles per chunk (0 = the plan kernel's rule)
// a chunk whose tiles are folded by more 32-row passes than this asks for its K / V rows with the temporal cache policy
// (launch_stage1_np; the record kernels write the flag)
static int np_nt_passes_knob() { return knob("DEFT_NP_NT_PASSES", 5); }
static int np_union_knob() { return knob("DEFT_NP_UNION", 0) | (knob("DEFT_NP_TAPER", 0) << 16); }  // leaf tiles per union group (1 = off, 0 = rule)

// Work items of stage 1 per chunk leader, for the plan's chunk-length rules (they weigh the number of workgroups against the
// resident slots): one per KV head -- per head PAIR where stage 1 will run head_dim 64 two heads to a row (hd2_geometry).  The plan
// calls carry no head_dim; the q head stride does (64 elements = contiguous heads of 64).  Only the rules depend on it.
// (head pairs are passed NEGATED: np_record_order)
static int plan_items_per_leader(const Stage1Params& p) { return (p.q_sh == 64 && p.Hkv % 2 == 0) ? -(p.Hkv / 2) : p.Hkv; }

// Flatten plan: unit list (one workgroup) then one record per unit.
static int launch_plan(const Stage1Params& p, int NB, const PlanView& pv, const AppendArgs& ap, hipStream_t stream,
                       const int32_t* dims = nullptr, int win_tiles = 0, int32_t* win_tab = nullptr) {
    if (NB <= 0) return DEFT_OK;
    // The unit kernel is one workgroup and may use the CU's whole LDS: 9216 blocks = 1.18 M KV tokens per call,
    // more than a 7B model's KV cache fits in 288 GB.
    constexpr size_t UNIT_LDS = 156 * 1024;
    if (sizeof(int) * 4 * (size_t)NB > 144 * 1024) {
        set_error("plan: %d blocks exceed the unit kernel's LDS", NB);
        return DEFT_EUNSUPPORTED;
    }
    int rc = raise_lds(reinterpret_cast<const void*>(&flatten_units_kernel), (int)UNIT_LDS, ATTR_UNITS, "flatten_units");
    if (rc) return rc;
    // block tables (+ the small blocks' query lists for the union groups, when they fit) and a run table.  With an
    // entry for every possible run (7 words each) the kernel writes units and record order with all its waves;
    // otherwise one lane emits them and the table holds as many runs as fit (beyond that it scans).
    const int qtab = sizeof(int) * (4 + DEFT_UNION_CAP) * (size_t)NB <= 100 * 1024;
    const size_t blk = (qtab ? 4 + DEFT_UNION_CAP : 4) * (size_t)NB;
    int64_t run_cap = pv.cap;
    int par = !g_plan_serial;
    if (par && sizeof(int) * (blk + 7 * (size_t)run_cap + 8) > UNIT_LDS) {  // as many runs as fit (the kernel falls back if more turn up)
        run_cap = ((int64_t)(UNIT_LDS / sizeof(int)) - (int64_t)blk - 8) / 7;
        if (run_cap < 256) par = 0, run_cap = pv.cap;
    }
    if (par && g_plan_runcap > 0) run_cap = std::max(1, std::min((int)run_cap, g_plan_runcap));  // tests: force the fallback
    if (!par)
        while (sizeof(int) * (blk + 3 * (size_t)run_cap + 8) > UNIT_LDS && run_cap > 0) run_cap /= 2;
    const size_t lds = sizeof(int) * (blk + (par ? 7 : 3) * (size_t)run_cap + 8);
    const UnitList ul = unit_list(pv);
    hipLaunchKernelGGL(flatten_units_kernel, dim3(1), dim3(1024), lds, stream, p.block_q, p.block_q_cnts, p.block_q_offset, NB,
                       p.G, (int)pv.cap, ul, pv.hdr, plan_items_per_leader(p), 2 * num_cus(), np_chunk_knob(), np_union_knob(), (int)run_cap, qtab,
                       par, dims, pv.row_q, (int)pv.rows, win_tiles, p.block_lens, p.G == 1 ? knob("DEFT_NP_SOLO_FULL", 1) : 0);
    rc = check_launch("flatten units launch");
    if (rc) return rc;
    hipLaunchKernelGGL(flatten_records_kernel, dim3((unsigned)(pv.cap + 1)), dim3(128), 0, stream, p.block_q, p.block_q_cnts,
                       p.block_bitmasks, p.block_kv, p.block_lens, p.G, (int)p.rows, p.q_st, p.q_sh, p.kv_ss, ul, pv.hdr,
                       pv.records, pv.row_q, ap.cache_loc, ap.n_new, ap.new_st * 2, np_nt_passes_knob(), NB, dims, win_tiles, win_tab);
    rc = check_launch("flatten records launch");
    if (rc) return rc;
    return launch_qrows(pv, stream);
}

// Stage 1 (head_dim 128): one workgroup per record slot and KV head; slots that are not chunk leaders exit at once
// (they are at the end of the grid).
// head_dim 64 on the tile-parallel kernel: two adjacent KV heads per 256-byte pool row (stage1_np.h, HD2) -- an even number of
// KV heads laid out contiguously ([slot][K|V][Hkv][64], the reference's pool, memory_pool.py:61-66)
static bool hd2_geometry(int D, int Hkv, int64_t kv_sh) { return D == 64 && Hkv % 2 == 0 && kv_sh == 64; }

static int launch_stage1_np(const Stage1Params& p, int64_t unit_cap, const PlanView& pv, const AppendArgs& ap,
                            hipStream_t stream, int nq, bool reread = false, bool hd2 = false) {
    using SM = NpSmem<128>;
    const bool rope = ap.cos_sin != nullptr;
    const int HP = hd2 ? p.Hkv / 2 : p.Hkv;  // work items per chunk leader: KV heads, or head pairs
    // K / V rows by NON-TEMPORAL LDS-DMA wherever a row is read by the few 32-row passes of its tile and never again: the
    // tree modes.  Same box, stage 1 (tools/ab.py, experiments build DEFT_NP_NT=0/1): north-star tree 36.0 -> 32.3 us, 1k x 32
    // 26.5 -> 24.1, 400-token branches 54.0 -> 48.4; whole layer: 8-tree forest 64.7 -> 60.0, Llama-3 north-star tree 23.5 ->
    // 22.8, Medusa-64 18.7 -> 18.0, ToT-50 (six passes per root tile) 24.0 -> 24.3.  NOT the sequential comparator, where every
    // leaf re-reads the shared prefix through the caches: 216 -> 298 us per layer (`reread`).
    // ... and not for the chunks whose tiles are folded by more than five 32-row passes (ToT-50 on Llama-3-8B: 50 queries x 4 = seven
    // passes over every root tile; round 4, tools/ab.py DEFT_NP_NT=0,1 on the whole launch: stage 1 17.99 vs 18.50 us without /
    // with, while four passes -- the north-star tree on Llama-3-8B -- still gain: 18.29 -> 16.93): the later passes find the rows in
    // L2 only if the first ones left them there.  That is a property of a RUN of tiles, not of the launch (an 8-tree forest has 64
    // queries and one pass per tile), so the plan's record kernels flag it per chunk leader (desc[6]) and the kernel picks the
    // policy per work item.
    // (MHA keeps its launch-level rule: more than 1024 virtual rows read temporally.  Medusa with 256 queries -- eight passes over
    //  the root, 32 KV heads -- measured 0.4 us per layer SLOWER with its root chunks temporal: the per-chunk flag is GQA's.)
    const bool nt = knob("DEFT_NP_NT", 1) != 0 && !reread && (p.G > 1 || (int64_t)nq <= 1024);
    // DYN instantiations (stage1_np.h): per-chunk cache policy and the mirrored item order -- GQA launches
    // ... that can use either: launches small enough for the mirrored order (below), or with enough virtual rows (> 160) for a
    // node to be folded by more than five passes.  The rest (a 64k-token prefix under 8 branches: +0.45 us of 54 with the branches
    // in) run the plain instantiation.
    const bool dyn = nt && p.G > 1 && (unit_cap * HP <= 16LL * num_cus() || (int64_t)nq * p.G > 160);
    int rc;
    if (hd2) rc = dyn  ? raise_lds(reinterpret_cast<const void*>(&stage1_np_kernel<128, false, true, false, true, true>), SM::BYTES, ATTR_NP_HD2_DYN, "stage1_np_hd2_dyn")
                 : nt ? raise_lds(reinterpret_cast<const void*>(&stage1_np_kernel<128, false, true, false, true>), SM::BYTES, ATTR_NP_HD2, "stage1_np_hd2")
                      : raise_lds(reinterpret_cast<const void*>(&stage1_np_kernel<128, false, false, false, true>), SM::BYTES, ATTR_NP_HD2_T, "stage1_np_hd2_t");
    else if (rope) rc = dyn  ? raise_lds(reinterpret_cast<const void*>(&stage1_np_kernel<128, true, true, true, false, true>), SM::BYTES, ATTR_NP_ROPE_DYN, "stage1_np_rope_dyn")
                       : nt ? raise_lds(reinterpret_cast<const void*>(&stage1_np_kernel<128, true, true>), SM::BYTES, ATTR_NP_ROPE, "stage1_np_rope")
                            : raise_lds(reinterpret_cast<const void*>(&stage1_np_kernel<128, true, false>), SM::BYTES, ATTR_NP_ROPE_T, "stage1_np_rope_t");
    else rc = dyn  ? raise_lds(reinterpret_cast<const void*>(&stage1_np_kernel<128, false, true, false, false, true>), SM::BYTES, ATTR_NP_DYN, "stage1_np_dyn")
             : nt ? raise_lds(reinterpret_cast<const void*>(&stage1_np_kernel<128, false, true>), SM::BYTES, ATTR_NP, "stage1_np")
                  : raise_lds(reinterpret_cast<const void*>(&stage1_np_kernel<128, false, false>), SM::BYTES, ATTR_NP_T, "stage1_np_t");
    if (rc) return rc;
    if (unit_cap <= 0) return DEFT_OK;
    int64_t grid = unit_cap * HP;
    if (grid > 0x7fffffffLL) {
        set_error("stage1 grid too large: %lld", (long long)grid);
        return DEFT_EINVAL;
    }
    // The grid is sized by record CAPACITY (the leader count lives on the device); slots beyond the leaders exit at
    // once but still cost a dispatch each -- tens of thousands for the sequential comparator's one-query entries --
    // so the grid is capped at a few times the resident slots and a workgroup whose index has more than one item takes
    // them in a loop (item, item + grid, ...).  Measured (same box, DEFT_NP_GRIDCAP=0,1,2,3,8 in the experiments
    // build): 3 x slots is never worse and up to 22 % better for MHA (sequential north-star tree 261 -> 204 us,
    // 400-token branches 56.8 -> 54.2); GQA, where every workgroup's tiles come from L2 after the first pass, prefers
    // resident workgroups only (ToT-50 28.1 -> 24.0 us, 8-tree forest 65.5 -> 60.9).
    bool mirror;
    {
        // GQA launches whose whole record capacity is within 8 x the resident slots (a single tree: the Llama-3
        // north-star tree, ToT-50) have about one item per workgroup anyway and gain 2 us from 2 x slots -- or (late round 4) from
        // the resident workgroups alone taking their further items in MIRRORED order (b, 2W-1-b, 2W+b, ...): the items are sorted
        // long chunks first, so the extra ones go to the workgroups whose first item was a short leaf tile, and start without a
        // dispatch (tools/knob_layer.sh, us per layer: GQA 4k x 32 20.7 hardware-dispatched at 2 x slots / 22.7 resident in plain
        // order / 20.4 mirrored; ToT-50 21.8 / 23.3 / 21.6; one 8k x 8 tree equal; NOT the large launches: north-star 36.2 / 41.4 /
        // 39.8, the 8-tree forest 54.8 / 55.0 / 56.6).
        const bool small_gqa = unit_cap * HP <= 16LL * num_cus();
        // (the mirrored order lives in the DYN instantiations only; a small GQA launch that cannot run one -- the sequential
        //  comparator's re-reads, which must not be slowed against the tree modes -- keeps hardware dispatch at 2 x slots: ADVICE r4)
        mirror = knob("DEFT_NP_MIRROR", dyn && small_gqa ? 1 : 0) != 0 && dyn;
        const int capx = knob("DEFT_NP_GRIDCAP", p.G > 1 ? ((small_gqa && !dyn) ? 2 : 1) : 3);
        const int64_t cap_wgs = (int64_t)capx * 2LL * num_cus();
        if (cap_wgs > 0 && grid > cap_wgs) grid = cap_wgs;
        const int gx = knob("DEFT_NP_GRID", 0);  // (experiments build, tests: a tiny grid = many rounds of the item loop)
        if (gx > 0 && grid > gx) grid = gx;
    }
    NpParams npp{};
    npp.s = p;
    npp.hdr = pv.hdr;
    // The speculative ramp (stage1_np.h: tile 0's offsets requested before the descriptor is known) for the workgroups resident at
    // launch.  (Half of them -- fast256 -- gains Medusa-64 0.3 us per layer: its 256 items leave the second half of the resident
    // workgroups without one, and those wait for their speculative requests before they may exit.  But it costs launches with more
    // items than that: the north-star tree through DeFT-Node +0.8 us, a 4k x 8 tree +0.2 (tools/shape_ab.sh) -- and the host does not
    // know the item count.  Not adopted.)
    npp.fast_n = knob("DEFT_NP_FAST", 2 * num_cus());
    npp.mirror = mirror ? 1 : 0;
#ifdef DEFT_EXPERIMENTS
    npp.skew_full = mirror ? 0 : knob("DEFT_NP_XCDSKEW", 0);
    npp.head_rot = knob("DEFT_NP_HEADROT", 0);
#endif
    npp.s.ablate = knob("DEFT_STAGE1_ABLATE", 0);
    npp.plan = pv.records;
    npp.k_new = ap.k_new;
    npp.v_new = ap.v_new;
    npp.cache_loc = ap.cache_loc;
    npp.new_st = ap.new_st;
    npp.n_new = ap.k_new ? ap.n_new : 0;
    npp.dbg = g_dbg;
    npp.cos_sin = ap.cos_sin;
    const dim3 g((unsigned)grid), b(256);
    if (hd2 && dyn) hipLaunchKernelGGL((stage1_np_kernel<128, false, true, false, true, true>), g, b, SM::BYTES, stream, npp);
    else if (rope && dyn) hipLaunchKernelGGL((stage1_np_kernel<128, true, true, true, false, true>), g, b, SM::BYTES, stream, npp);
    else if (dyn) hipLaunchKernelGGL((stage1_np_kernel<128, false, true, false, false, true>), g, b, SM::BYTES, stream, npp);
    else if (hd2 && nt) hipLaunchKernelGGL((stage1_np_kernel<128, false, true, false, true>), g, b, SM::BYTES, stream, npp);
    else if (hd2) hipLaunchKernelGGL((stage1_np_kernel<128, false, false, false, true>), g, b, SM::BYTES, stream, npp);
    else if (rope && nt) hipLaunchKernelGGL((stage1_np_kernel<128, true, true>), g, b, SM::BYTES, stream, npp);
    else if (rope) hipLaunchKernelGGL((stage1_np_kernel<128, true, false>), g, b, SM::BYTES, stream, npp);
    else if (nt) hipLaunchKernelGGL((stage1_np_kernel<128, false, true>), g, b, SM::BYTES, stream, npp);
    else hipLaunchKernelGGL((stage1_np_kernel<128, false, false>), g, b, SM::BYTES, stream, npp);
    return check_launch("stage1 np launch");
}

static int launch_merge(int D, const Workspace& ws, const PlanView* pv, const int32_t* row_q, int64_t rows, void* out,
                        int64_t o_st, int64_t o_sh, int nq, int Hq, int hgroup, hipStream_t stream) {
    if (nq <= 0) return DEFT_OK;
    const int cap = (int)std::min<int64_t>(std::max<int64_t>(rows, 64), MERGE_LIST_CAP);
    const size_t lds = sizeof(int) * 4 * (size_t)cap;
    if (hgroup <= 0 || Hq % hgroup) hgroup = 1;
    dim3 grid((unsigned)Hq, (unsigned)((nq + 3) / 4));
    // the plan's per-query row lists exist iff its row count fits the histogram kernel (launch_qrows): known on the host
    int lists = pv && pv->rows > 0 && pv->rows <= QROWS_MAX ? 1 : 0;
#ifdef DEFT_EXPERIMENTS
    if (const int fake = knob("DEFT_MERGE_FAKE", 0)) lists = 1 + fake;  // (timing experiment: `fake` computed row ids per query)
#endif
    const int32_t* qoff = pv ? pv->qoff : nullptr;
    const int32_t* qlist = pv ? pv->qlist : nullptr;
    const int32_t* qinl = pv ? pv->qinl : nullptr;
    // Few queries (where a long shared prefix means long row lists and merge_kernel's grid leaves most CUs empty): the
    // cooperative form.  The choice depends on nq and Hq alone -- NOT on row capacities, which differ between the eager
    // path and a captured session of the same step -- and the kernel itself falls back to merge_kernel's single-wave order
    // for lists of up to 8 rows, so the two paths stay bit-identical.
    if (D == 128 && lists && nq <= 16 && (int64_t)nq * Hq <= 1024) {
        const int ccap = (int)std::min<int64_t>(std::max<int64_t>(rows / 4 + 2, 16), MERGE_LIST_CAP);  // (>= 8: the single-wave path stages up to 8 rows)
        const size_t clds = sizeof(int) * 4 * (size_t)ccap + sizeof(float) * (4 * 128 + 8);
        hipLaunchKernelGGL((merge_coop_kernel<128>), dim3((unsigned)Hq, (unsigned)nq), dim3(256), clds, stream, ws.partial_o,
                           ws.partial_lse, rows, static_cast<_Float16*>(out), o_st, o_sh, ccap, qoff, qlist, qinl, Hq, hgroup);
        return check_launch("merge (cooperative) launch");
    }
#define DEFT_MERGE_LAUNCH(DD)                                                                                              \
    hipLaunchKernelGGL((merge_kernel<DD>), grid, dim3(256), lds, stream, ws.partial_o, ws.partial_lse, row_q, rows,         \
                       static_cast<_Float16*>(out), o_st, o_sh, Hq, cap, lists, qoff, qlist, qinl, nq, hgroup)
    if (D == 128) DEFT_MERGE_LAUNCH(128);
    else if (D == 64) DEFT_MERGE_LAUNCH(64);
    else if (D == 32) DEFT_MERGE_LAUNCH(32);
    else DEFT_MERGE_LAUNCH(16);
#undef DEFT_MERGE_LAUNCH
    return check_launch("merge launch");
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static int check_common(const void* q, int64_t q_st, int64_t q_sh, const void* k, const void* v, int64_t kv_ss,
                        int64_t kv_sh, const void* out, int64_t o_st, int64_t o_sh, int nq, int Hq, int Hkv, int D) {
    if (!q || !k || !v || !out) {
        set_error("null tensor pointer");
        return DEFT_EINVAL;
    }
    if (nq < 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv != 0) {
        set_error("bad geometry nq=%d Hq=%d Hkv=%d", nq, Hq, Hkv);
        return DEFT_EINVAL;
    }
    if (!deft_supported(Hq, Hkv, D)) {
        set_error("unsupported geometry Hq=%d Hkv=%d D=%d (head_dim must be 16, 32, 64 or 128)", Hq, Hkv, D);
        return DEFT_EUNSUPPORTED;
    }
    if (!aligned16(q) || !aligned16(k) || !aligned16(v) || (q_st % 8) || (q_sh % 8) || (kv_ss % 8) || (kv_sh % 8)) {
        set_error("q/k/v rows must be 16-byte aligned (pointers and strides multiples of 8 elements)");
        return DEFT_EINVAL;
    }
    if ((reinterpret_cast<uintptr_t>(out) & 3u) || (o_st % 2) || (o_sh % 2)) {
        set_error("out rows must be 4-byte aligned");
        return DEFT_EINVAL;
    }
    return DEFT_OK;
}

}  // namespace deft

using namespace deft;

extern "C" {

int deft_abi_version(void) { return 2; }  // (2: deft_tree_layout swallows the pending journal; window plans; deft_stage_fetch)

// Everything a plan's layout depends on besides the caller's arguments.  The shipped library has no such thing (0);
// the experiments build folds its plan knobs into the value, so that callers which cache plans key them by it.
int deft_plan_variant(void) {
    // (the union knob's bits 8 and up carry a query cap that changes the plan as well: folded in whole)
    const unsigned u = (unsigned)np_union_knob();
    unsigned key = (((unsigned)np_chunk_knob() & 0xff) << 4) | ((u & 0xff) << 12) | ((g_plan_serial ? 1u : 0u) << 24) |
                   (((unsigned)g_plan_runcap & 0x3f) << 25);
    key ^= (unsigned)(np_nt_passes_knob() - 5) & 0xf;
    key ^= (((u >> 8) * 0x9E3779B1u) >> 8) << 4;
    return (int)(key & 0x7fffffffu);
}

// Internal hooks of the EXPERIMENTS build only (libdeft_amd_exp.so; the shipped library exports exactly what
// include/deft_amd.h declares).  deft_debug_plan_form: tests force the plan kernels' fallback forms (serial: one lane emits
// the plan; runcap > 0: a run table of that many entries).  deft_debug_set_buffer: device buffer of 8192 x 8 u64 receiving
// per-workgroup time stamps of stage 1.
#ifdef DEFT_EXPERIMENTS
__attribute__((visibility("default"))) void deft_debug_plan_form(int serial, int runcap) {
    g_plan_serial = serial;
    g_plan_runcap = runcap;
}
__attribute__((visibility("default"))) void deft_debug_set_buffer(void* dev_ptr) { g_dbg = static_cast<unsigned long long*>(dev_ptr); }
#endif

// Measurement aid (include/deft_amd.h): a bare coalesced read of a byte range, the ceiling of a launch of that size.
namespace deft {
__global__ __launch_bounds__(256) void probe_stream_read_kernel(const uintx4* p, size_t n, uint32_t* sink) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    uint32_t acc = 0;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const uintx4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc += a.x ^ b.y ^ c.z ^ d.w;
    }
    for (; i < n; i += stride) acc += p[i].x;
    if (acc == 0x9e3779b9u && sink) sink[0] = acc;  // (keeps the loads alive; practically never taken)
}
}  // namespace deft
int deft_probe_stream_read(const void* base, size_t bytes, int workgroups, void* stream) {
    if (!base || !aligned16(base) || workgroups <= 0) {
        set_error("deft_probe_stream_read: null / misaligned base or no workgroups");
        return DEFT_EINVAL;
    }
    if (bytes < 16) return DEFT_OK;
    hipLaunchKernelGGL(probe_stream_read_kernel, dim3((unsigned)workgroups), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uintx4*>(base), bytes / 16, static_cast<uint32_t*>(nullptr));
    return check_launch("probe launch");
}

int deft_supported(int Hq, int Hkv, int D) {
    return (Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && (D == 16 || D == 32 || D == 64 || D == 128)) ? 1 : 0;
}

size_t deft_flatten_workspace_bytes(int NB, int P, int nq, int Hq, int Hkv, int D) {
    (void)nq;
    (void)Hkv;
    if (Hq <= 0 || Hkv <= 0 || Hq % Hkv) return 0;
    return carve(nullptr, Hq, D, P, 0, plan_view(nullptr, flatten_unit_cap(NB, Hq / Hkv), P).bytes).bytes;
}

// Partial rows per tile of a Node launch: an entry has at most DEFT_MAX_Q_LEN queries -- and never more than the P (query, entry)
// pairs of the whole call, which is what keeps the scratch of a 100k-token prefix under three branches at a tenth of the
// 32-rows-per-tile figure.  Every sizing function and the launch derive it from P alone, so they agree.
static inline int64_t node_rows_per_tile(int P) { return P <= 0 ? 1 : (P < DEFT_MAX_Q_LEN ? P : DEFT_MAX_Q_LEN); }

size_t deft_node_workspace_bytes(int NE, int P, int64_t total_kv, int nq, int Hq, int Hkv, int D) {
    (void)nq;
    if (Hq <= 0 || Hkv <= 0 || Hq % Hkv) return 0;
    const int64_t tiles = node_max_tiles(NE, total_kv);
    const int64_t rows = tiles * node_rows_per_tile(P);
    return carve(nullptr, Hq, D, rows, tiles, plan_view(nullptr, tiles * (Hq / Hkv), rows).bytes).bytes;
}

static int check_append(const AppendArgs& ap, int Hkv, int D) {
    if (!ap.k_new && !ap.v_new && !ap.cache_loc) return DEFT_OK;
    if (!ap.k_new || !ap.v_new || !ap.cache_loc || ap.n_new < 0 || (ap.new_st % 8) || ap.new_st < (int64_t)Hkv * D ||
        !aligned16(ap.k_new) || !aligned16(ap.v_new)) {
        set_error("bad fused-append arguments (n_new=%d new_stride=%lld)", ap.n_new, (long long)ap.new_st);
        return DEFT_EINVAL;
    }
    return DEFT_OK;
}

// Fused rotary embedding (the *_rope_append_* entry points): NeoX pairing over the whole head, head_dim 128 (the
// tile-parallel kernel), token-major q so that a virtual row's query index is its element offset / q_stride_tok.
static int fill_rope(AppendArgs& ap, const float* cos_sin_rows, int rotary_dim, int is_neox_style, int64_t q_st,
                     int64_t q_sh, int Hq, int D) {
    if (!cos_sin_rows || !aligned16(cos_sin_rows)) {
        set_error("bad fused-rope arguments (cos_sin_rows: [n_new][head_dim] fp32, 16-byte aligned)");
        return DEFT_EINVAL;
    }
    if (D != 128 || rotary_dim != D || !is_neox_style) {
        set_error("fused rope: head_dim 128, rotary_dim == head_dim, NeoX pairing (got D=%d rotary_dim=%d neox=%d); "
                  "use deft_rope_qk_f16 + the append entry point", D, rotary_dim, is_neox_style);
        return DEFT_EUNSUPPORTED;
    }
    if (q_st <= 0 || q_st > 0x7fffffffLL || (int64_t)(Hq - 1) * q_sh + D > q_st || q_sh < 0) {
        set_error("fused rope needs token-major q (q_stride_tok=%lld q_stride_head=%lld)", (long long)q_st, (long long)q_sh);
        return DEFT_EUNSUPPORTED;
    }
    ap.cos_sin = cos_sin_rows;
    return DEFT_OK;
}

// Shared body of the Flatten entry points: stage 1 into the workspace; reports which
// partial-row -> query map the merge must read.
static int flatten_stage1_impl(const void* q, int64_t q_stride_tok, int64_t q_stride_head, const void* k_base,
                               const void* v_base, int64_t kv_stride_slot, int64_t kv_stride_head, const int64_t* block_q,
                               const int64_t* block_q_cnts, const int64_t* block_q_offset, const int64_t* block_bitmasks,
                               const int64_t* block_kv, const int64_t* block_lens, int NB, int P, int nq, int Hq, int Hkv,
                               int D, float scale, const void* plan, void* workspace, size_t workspace_bytes, void* stream,
                               const AppendArgs& ap, Workspace* ws_out, const int32_t** row_q_out,
                               PlanView* pv_out) {
    pv_out->hdr = nullptr;  // (no plan: head_dim 64)
    // `workspace` doubles as the (unused) output pointer for the shared argument check
    int rc = check_common(q, q_stride_tok, q_stride_head, k_base, v_base, kv_stride_slot, kv_stride_head, workspace, 2, 2,
                          nq, Hq, Hkv, D);
    if (rc) return rc;
    if (NB < 0 || P < 0 || (NB > 0 && (!block_q || !block_q_cnts || !block_q_offset || !block_bitmasks || !block_kv ||
                                       !block_lens))) {
        set_error("bad Flatten metadata (NB=%d P=%d)", NB, P);
        return DEFT_EINVAL;
    }
    const int64_t cap = flatten_unit_cap(NB, Hq / Hkv);
    const Workspace ws = carve(workspace, Hq, D, P, 0, plan_view(nullptr, cap, P).bytes);
    if (workspace_bytes < ws.bytes) {
        set_error("workspace too small: %zu < %zu", workspace_bytes, ws.bytes);
        return DEFT_EWORKSPACE;
    }
    Stage1Params p{};
    p.q = static_cast<const _Float16*>(q);
    p.q_st = q_stride_tok;
    p.q_sh = q_stride_head;
    p.k = static_cast<const _Float16*>(k_base);
    p.v = static_cast<const _Float16*>(v_base);
    p.kv_ss = kv_stride_slot;
    p.kv_sh = kv_stride_head;
    p.block_q = block_q;
    p.block_q_cnts = block_q_cnts;
    p.block_q_offset = block_q_offset;
    p.block_bitmasks = block_bitmasks;
    p.block_kv = block_kv;
    p.block_lens = block_lens;
    p.partial_o = ws.partial_o;
    p.partial_lse = ws.partial_lse;
    p.row_q = ws.row_q;
    p.rows = P;
    p.Hkv = Hkv;
    p.G = Hq / Hkv;
    p.scale_log2e = scale * LOG2E;
    *ws_out = ws;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool hd2 = hd2_geometry(D, Hkv, kv_stride_head) && !ap.cos_sin;
    if (D == 128 || hd2) {
        PlanView pv;
        if (plan) {
            pv = plan_view(const_cast<void*>(plan), cap, P);
        } else {
            pv = plan_view(ws.plan, cap, P);
            rc = launch_plan(p, NB, pv, ap, st);
            if (rc) return rc;
        }
        *row_q_out = pv.row_q;
        *pv_out = pv;
        return launch_stage1_np(p, cap, pv, ap, st, nq, false, hd2);
    }
    if (ap.k_new) {  // head_dim 64 (tile-per-workgroup form): separate append launch first
        rc = deft_kv_append_f16(const_cast<void*>(k_base), const_cast<void*>(v_base), kv_stride_slot, kv_stride_head,
                                ap.cache_loc, ap.k_new, ap.v_new, ap.new_st, ap.n_new, Hkv, D, stream);
        if (rc) return rc;
    }
    *row_q_out = ws.row_q;
    return launch_stage1_d64<0>(D, p, NB, st);
}

size_t deft_flatten_plan_bytes(int NB, int P, int Hq, int Hkv) {
    if (Hq <= 0 || Hkv <= 0 || Hq % Hkv) return 0;
    return plan_view(nullptr, flatten_unit_cap(NB, Hq / Hkv), P).bytes;
}

int deft_flatten_build_plan(const int64_t* block_q, const int64_t* block_q_cnts, const int64_t* block_q_offset,
                            const int64_t* block_bitmasks, const int64_t* block_kv, const int64_t* block_lens, int NB, int P,
                            int Hq, int Hkv, int64_t q_stride_tok, int64_t q_stride_head, int64_t kv_stride_slot,
                            const int32_t* cache_loc, int n_new, int64_t new_stride_tok, void* plan, size_t plan_bytes,
                            void* stream) {
    if (NB < 0 || P < 0 || !plan || n_new < 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv ||
        (NB > 0 && (!block_q || !block_q_cnts || !block_q_offset || !block_bitmasks || !block_kv || !block_lens))) {
        set_error("bad plan arguments (NB=%d P=%d Hq=%d Hkv=%d)", NB, P, Hq, Hkv);
        return DEFT_EINVAL;
    }
    const PlanView pv = plan_view(plan, flatten_unit_cap(NB, Hq / Hkv), P);
    if (plan_bytes < pv.bytes) {
        set_error("plan buffer too small: %zu < %zu", plan_bytes, pv.bytes);
        return DEFT_EWORKSPACE;
    }
    Stage1Params p{};
    p.block_q = block_q;
    p.block_q_cnts = block_q_cnts;
    p.block_q_offset = block_q_offset;
    p.block_bitmasks = block_bitmasks;
    p.block_kv = block_kv;
    p.block_lens = block_lens;
    p.rows = P;
    p.G = Hq / Hkv;
    p.Hkv = Hkv;
    p.q_st = q_stride_tok;
    p.q_sh = q_stride_head;
    p.kv_ss = kv_stride_slot;
    AppendArgs ap;
    ap.cache_loc = cache_loc;
    ap.n_new = cache_loc ? n_new : 0;
    ap.new_st = new_stride_tok;
    return launch_plan(p, NB, pv, ap, static_cast<hipStream_t>(stream));
}

// The same for metadata that was built ON THE DEVICE (deft_tree_dev_build_md): NB and P are the CAPACITIES of the arrays
// (they size the plan, the grids and the partial-row stride) and the block count of this step is read from `dims`
// (dims[5], the scratch header of deft_tree_dev_build_md) by the kernel -- so the launch has the same arguments for every
// decode step of a structural epoch of the tree and can sit in a captured hipGraph.
int deft_flatten_build_plan_dims(const int64_t* block_q, const int64_t* block_q_cnts, const int64_t* block_q_offset,
                                 const int64_t* block_bitmasks, const int64_t* block_kv, const int64_t* block_lens, int NB, int P,
                                 const int32_t* dims, int Hq, int Hkv, int64_t q_stride_tok, int64_t q_stride_head,
                                 int64_t kv_stride_slot, const int32_t* cache_loc, int n_new, int64_t new_stride_tok, void* plan,
                                 size_t plan_bytes, void* stream) {
    if (NB < 0 || P < 0 || !plan || !dims || n_new < 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv ||
        (NB > 0 && (!block_q || !block_q_cnts || !block_q_offset || !block_bitmasks || !block_kv || !block_lens))) {
        set_error("bad plan arguments (NB=%d P=%d Hq=%d Hkv=%d)", NB, P, Hq, Hkv);
        return DEFT_EINVAL;
    }
    const PlanView pv = plan_view(plan, flatten_unit_cap(NB, Hq / Hkv), P);
    if (plan_bytes < pv.bytes) {
        set_error("plan buffer too small: %zu < %zu", plan_bytes, pv.bytes);
        return DEFT_EWORKSPACE;
    }
    Stage1Params p{};
    p.block_q = block_q;
    p.block_q_cnts = block_q_cnts;
    p.block_q_offset = block_q_offset;
    p.block_bitmasks = block_bitmasks;
    p.block_kv = block_kv;
    p.block_lens = block_lens;
    p.rows = P;
    p.G = Hq / Hkv;
    p.Hkv = Hkv;
    p.q_st = q_stride_tok;
    p.q_sh = q_stride_head;
    p.kv_ss = kv_stride_slot;
    AppendArgs ap;
    ap.cache_loc = cache_loc;
    ap.n_new = cache_loc ? n_new : 0;
    ap.new_st = new_stride_tok;
    return launch_plan(p, NB, pv, ap, static_cast<hipStream_t>(stream), dims);
}

int deft_flatten_stage1_f16(const void* q, int64_t q_stride_tok, int64_t q_stride_head, const void* k_base,
                            const void* v_base, int64_t kv_stride_slot, int64_t kv_stride_head, const int64_t* block_q,
                            const int64_t* block_q_cnts, const int64_t* block_q_offset, const int64_t* block_bitmasks,
                            const int64_t* block_kv, const int64_t* block_lens, int NB, int P, int nq, int Hq, int Hkv,
                            int D, float scale, const void* plan, void* workspace, size_t workspace_bytes, void* stream) {
    if (!workspace) {
        set_error("null workspace");
        return DEFT_EINVAL;
    }
    Workspace ws;
    const int32_t* row_q = nullptr;
    PlanView pv;
    return flatten_stage1_impl(q, q_stride_tok, q_stride_head, k_base, v_base, kv_stride_slot, kv_stride_head, block_q,
                               block_q_cnts, block_q_offset, block_bitmasks, block_kv, block_lens, NB, P, nq, Hq, Hkv, D,
                               scale, plan, workspace, workspace_bytes, stream, AppendArgs(), &ws, &row_q, &pv);
}

static int flatten_decode_impl(const void* q, int64_t q_stride_tok, int64_t q_stride_head, const void* k_base,
                               const void* v_base, int64_t kv_stride_slot, int64_t kv_stride_head, void* out,
                               int64_t o_stride_tok, int64_t o_stride_head, const int64_t* block_q,
                               const int64_t* block_q_cnts, const int64_t* block_q_offset, const int64_t* block_bitmasks,
                               const int64_t* block_kv, const int64_t* block_lens, int NB, int P, int nq, int Hq, int Hkv,
                               int D, float scale, const void* plan, void* workspace, size_t workspace_bytes, void* stream,
                               const AppendArgs& ap) {
    int rc = check_common(q, q_stride_tok, q_stride_head, k_base, v_base, kv_stride_slot, kv_stride_head, out,
                          o_stride_tok, o_stride_head, nq, Hq, Hkv, D);
    if (rc) return rc;
    rc = check_append(ap, Hkv, D);
    if (rc) return rc;
    if (!workspace) {
        set_error("null workspace");
        return DEFT_EINVAL;
    }
    Workspace ws;
    const int32_t* row_q = nullptr;
    PlanView pv;
    rc = flatten_stage1_impl(q, q_stride_tok, q_stride_head, k_base, v_base, kv_stride_slot, kv_stride_head, block_q,
                             block_q_cnts, block_q_offset, block_bitmasks, block_kv, block_lens, NB, P, nq, Hq, Hkv, D, scale,
                             plan, workspace, workspace_bytes, stream, ap, &ws, &row_q, &pv);
    if (rc) return rc;
    return launch_merge(D, ws, pv.hdr ? &pv : nullptr, row_q, P, out, o_stride_tok, o_stride_head, nq, Hq,
                        (Hq / Hkv) * (hd2_geometry(D, Hkv, kv_stride_head) && !ap.cos_sin ? 2 : 1), static_cast<hipStream_t>(stream));
}

int deft_flatten_decode_f16(const void* q, int64_t q_stride_tok, int64_t q_stride_head, const void* k_base,
                            const void* v_base, int64_t kv_stride_slot, int64_t kv_stride_head, void* out,
                            int64_t o_stride_tok, int64_t o_stride_head, const int64_t* block_q,
                            const int64_t* block_q_cnts, const int64_t* block_q_offset, const int64_t* block_bitmasks,
                            const int64_t* block_kv, const int64_t* block_lens, int NB, int P, int nq, int Hq, int Hkv,
                            int D, float scale, const void* plan, void* workspace, size_t workspace_bytes, void* stream) {
    return flatten_decode_impl(q, q_stride_tok, q_stride_head, k_base, v_base, kv_stride_slot, kv_stride_head, out,
                               o_stride_tok, o_stride_head, block_q, block_q_cnts, block_q_offset, block_bitmasks, block_kv,
                               block_lens, NB, P, nq, Hq, Hkv, D, scale, plan, workspace, workspace_bytes, stream,
                               AppendArgs());
}

int deft_flatten_decode_append_f16(const void* q, int64_t q_stride_tok, int64_t q_stride_head, void* k_base, void* v_base,
                                   int64_t kv_stride_slot, int64_t kv_stride_head, void* out, int64_t o_stride_tok,
                                   int64_t o_stride_head, const int64_t* block_q, const int64_t* block_q_cnts,
                                   const int64_t* block_q_offset, const int64_t* block_bitmasks, const int64_t* block_kv,
                                   const int64_t* block_lens, int NB, int P, int nq, int Hq, int Hkv, int D, float scale,
                                   const int32_t* cache_loc, const void* k_new, const void* v_new, int64_t new_stride_tok,
                                   int n_new, const void* plan, void* workspace, size_t workspace_bytes, void* stream) {
    AppendArgs ap;
    ap.k_new = static_cast<const _Float16*>(k_new);
    ap.v_new = static_cast<const _Float16*>(v_new);
    ap.cache_loc = cache_loc;
    ap.new_st = new_stride_tok;
    ap.n_new = n_new;
    if (!k_new || !v_new || !cache_loc) {
        set_error("fused append needs k_new, v_new and cache_loc");
        return DEFT_EINVAL;
    }
    return flatten_decode_impl(q, q_stride_tok, q_stride_head, k_base, v_base, kv_stride_slot, kv_stride_head, out,
                               o_stride_tok, o_stride_head, block_q, block_q_cnts, block_q_offset, block_bitmasks, block_kv,
                               block_lens, NB, P, nq, Hq, Hkv, D, scale, plan, workspace, workspace_bytes, stream, ap);
}

int deft_flatten_decode_rope_append_f16(const void* q, int64_t q_stride_tok, int64_t q_stride_head, void* k_base, void* v_base,
                                        int64_t kv_stride_slot, int64_t kv_stride_head, void* out, int64_t o_stride_tok,
                                        int64_t o_stride_head, const int64_t* block_q, const int64_t* block_q_cnts,
                                        const int64_t* block_q_offset, const int64_t* block_bitmasks, const int64_t* block_kv,
                                        const int64_t* block_lens, int NB, int P, int nq, int Hq, int Hkv, int D, float scale,
                                        const int32_t* cache_loc, const void* k_new, const void* v_new, int64_t new_stride_tok,
                                        int n_new, const float* cos_sin_rows, int rotary_dim, int is_neox_style,
                                        const void* plan, void* workspace, size_t workspace_bytes, void* stream) {
    AppendArgs ap;
    ap.k_new = static_cast<const _Float16*>(k_new);
    ap.v_new = static_cast<const _Float16*>(v_new);
    ap.cache_loc = cache_loc;
    ap.new_st = new_stride_tok;
    ap.n_new = n_new;
    if (!k_new || !v_new || !cache_loc) {
        set_error("fused append needs k_new, v_new and cache_loc");
        return DEFT_EINVAL;
    }
    const int rc = fill_rope(ap, cos_sin_rows, rotary_dim, is_neox_style, q_stride_tok, q_stride_head, Hq, D);
    if (rc) return rc;
    return flatten_decode_impl(q, q_stride_tok, q_stride_head, k_base, v_base, kv_stride_slot, kv_stride_head, out,
                               o_stride_tok, o_stride_head, block_q, block_q_cnts, block_q_offset, block_bitmasks, block_kv,
                               block_lens, NB, P, nq, Hq, Hkv, D, scale, plan, workspace, workspace_bytes, stream, ap);
}

static int launch_node_plan(const Stage1Params& p, int NE, int64_t rows_cap, const PlanView& pv, const AppendArgs& ap,
                            hipStream_t stream, int keep_err = 0, const int32_t* dims = nullptr, int win_tiles = 0,
                            int32_t* win_tab = nullptr) {
    const UnitList ul = unit_list(pv);
    // a run table in LDS: 10 words per run and every possible run (the kernel then writes units and
    // record order with all its waves), or as many runs as fit (it falls back to one lane if more turn up)
    constexpr size_t UNIT_LDS = 156 * 1024;
    int rc = raise_lds(reinterpret_cast<const void*>(&node_units_kernel), (int)UNIT_LDS, ATTR_NODE_UNITS, "node_units");
    if (rc) return rc;
    int64_t run_cap = pv.cap > 0 ? pv.cap : 1;
    int par = !g_plan_serial;
    if (par && sizeof(int) * (10 * (size_t)run_cap + 8) > UNIT_LDS) {
        run_cap = ((int64_t)(UNIT_LDS / sizeof(int)) - 8) / 10;
        if (run_cap < 256) par = 0, run_cap = pv.cap > 0 ? pv.cap : 1;
    }
    if (par && g_plan_runcap > 0) run_cap = std::max(1, std::min((int)run_cap, g_plan_runcap));  // tests: force the fallback
    if (!par)
        while (sizeof(int) * (3 * (size_t)run_cap + 8) > UNIT_LDS && run_cap > 1) run_cap /= 2;
    hipLaunchKernelGGL(node_units_kernel, dim3(1), dim3(1024), sizeof(int) * ((par ? 10 : 3) * (size_t)run_cap + 8), stream,
                       p.node_kv_len, p.node_q_len, p.node_q, p.node_q_offset, NE, p.G, (int)pv.cap, rows_cap, ul, pv.hdr, pv.row_q,
                       plan_items_per_leader(p),
                       2 * num_cus(), np_chunk_knob(), (int)run_cap, par, keep_err, dims, win_tiles);
    rc = check_launch("node units launch");
    if (rc) return rc;
    hipLaunchKernelGGL(node_records_kernel, dim3((unsigned)(pv.cap + 1)), dim3(128), 0, stream, p.node_kv, p.node_kv_offset,
                       p.node_kv_len, p.node_q, p.node_q_offset, p.node_q_len, p.G, (int)p.rows, p.q_st, p.q_sh, p.kv_ss, ul,
                       pv.hdr, pv.records, pv.row_q, ap.cache_loc, ap.n_new, ap.new_st * 2, np_nt_passes_knob(), NE, dims, win_tiles,
                       win_tab);
    rc = check_launch("node records launch");
    if (rc) return rc;
    return launch_qrows(pv, stream);
}

size_t deft_node_plan_bytes(int NE, int P, int64_t total_kv, int Hq, int Hkv) {
    if (Hq <= 0 || Hkv <= 0 || Hq % Hkv) return 0;
    const int64_t tiles = node_max_tiles(NE, total_kv);
    return plan_view(nullptr, tiles * (Hq / Hkv), tiles * node_rows_per_tile(P)).bytes;
}

static int node_build_plan_impl(const int64_t* node_kv, const int64_t* node_kv_offset, const int64_t* node_kv_len,
                                const int64_t* node_q, const int64_t* node_q_offset, const int64_t* node_q_len, int NE, int P,
                                int64_t total_kv, const int32_t* dims, int Hq, int Hkv, int64_t q_stride_tok,
                                int64_t q_stride_head, int64_t kv_stride_slot, const int32_t* cache_loc, int n_new,
                                int64_t new_stride_tok, void* plan, size_t plan_bytes, void* stream) {
    if (NE < 0 || P < 0 || total_kv < 0 || total_kv > 0x7fffffffLL || !plan || Hq <= 0 || Hkv <= 0 || Hq % Hkv ||
        (NE > 0 && (!node_kv || !node_kv_offset || !node_kv_len || !node_q || !node_q_offset || !node_q_len))) {
        set_error("bad node plan arguments (NE=%d P=%d total_kv=%lld)", NE, P, (long long)total_kv);
        return DEFT_EINVAL;
    }
    const int64_t tiles = node_max_tiles(NE, total_kv);
    const int64_t rows = tiles * node_rows_per_tile(P);
    const PlanView pv = plan_view(plan, tiles * (Hq / Hkv), rows);
    if (plan_bytes < pv.bytes) {
        set_error("plan buffer too small: %zu < %zu", plan_bytes, pv.bytes);
        return DEFT_EWORKSPACE;
    }
    Stage1Params p{};
    p.node_kv = node_kv;
    p.node_kv_offset = node_kv_offset;
    p.node_kv_len = node_kv_len;
    p.node_q = node_q;
    p.node_q_offset = node_q_offset;
    p.node_q_len = node_q_len;
    p.rows = rows;
    p.G = Hq / Hkv;
    p.Hkv = Hkv;
    p.q_st = q_stride_tok;
    p.q_sh = q_stride_head;
    p.kv_ss = kv_stride_slot;
    AppendArgs ap;  // only the slots of this step's new rows and their stride matter to the plan
    if (cache_loc) {
        if (n_new < 0 || new_stride_tok < 0) {
            set_error("bad node plan append arguments (n_new=%d)", n_new);
            return DEFT_EINVAL;
        }
        ap.cache_loc = cache_loc;
        ap.n_new = n_new;
        ap.new_st = new_stride_tok;
    }
    return launch_node_plan(p, NE, rows, pv, ap, static_cast<hipStream_t>(stream), 0, dims);
}

int deft_node_build_plan(const int64_t* node_kv, const int64_t* node_kv_offset, const int64_t* node_kv_len,
                         const int64_t* node_q, const int64_t* node_q_offset, const int64_t* node_q_len, int NE, int P,
                         int64_t total_kv, int Hq, int Hkv, int64_t q_stride_tok, int64_t q_stride_head,
                         int64_t kv_stride_slot, const int32_t* cache_loc, int n_new, int64_t new_stride_tok, void* plan,
                         size_t plan_bytes, void* stream) {
    return node_build_plan_impl(node_kv, node_kv_offset, node_kv_len, node_q, node_q_offset, node_q_len, NE, P, total_kv, nullptr, Hq,
                                Hkv, q_stride_tok, q_stride_head, kv_stride_slot, cache_loc, n_new, new_stride_tok, plan, plan_bytes,
                                stream);
}

// Node-mode counterpart of deft_flatten_build_plan_dims: NE, P, total_kv are the CAPACITIES of the device-built arrays, this
// step's entry count is read from dims[1] on the device.
int deft_node_build_plan_dims(const int64_t* node_kv, const int64_t* node_kv_offset, const int64_t* node_kv_len,
                              const int64_t* node_q, const int64_t* node_q_offset, const int64_t* node_q_len, int NE, int P,
                              int64_t total_kv, const int32_t* dims, int Hq, int Hkv, int64_t q_stride_tok,
                              int64_t q_stride_head, int64_t kv_stride_slot, const int32_t* cache_loc, int n_new,
                              int64_t new_stride_tok, void* plan, size_t plan_bytes, void* stream) {
    if (!dims) {
        set_error("deft_node_build_plan_dims: null dims");
        return DEFT_EINVAL;
    }
    return node_build_plan_impl(node_kv, node_kv_offset, node_kv_len, node_q, node_q_offset, node_q_len, NE, P, total_kv, dims, Hq,
                                Hkv, q_stride_tok, q_stride_head, kv_stride_slot, cache_loc, n_new, new_stride_tok, plan, plan_bytes,
                                stream);
}

}  // extern "C"

static int node_decode_impl(const void* q, int64_t q_stride_tok, int64_t q_stride_head, const void* k_base,
                            const void* v_base, int64_t kv_stride_slot, int64_t kv_stride_head, void* out,
                            int64_t o_stride_tok, int64_t o_stride_head, const int64_t* node_kv,
                            const int64_t* node_kv_offset, const int64_t* node_kv_len, const int64_t* node_q,
                            const int64_t* node_q_offset, const int64_t* node_q_len, int NE, int P, int64_t total_kv, int nq,
                            int Hq, int Hkv, int D, float scale, const void* plan, void* workspace, size_t workspace_bytes,
                            void* stream, const AppendArgs& ap, int rows_per_tile = 0 /* 0: from P; 1: the sequential comparator */) {
    int rc = check_common(q, q_stride_tok, q_stride_head, k_base, v_base, kv_stride_slot, kv_stride_head, out,
                          o_stride_tok, o_stride_head, nq, Hq, Hkv, D);
    if (rc) return rc;
    if (NE < 0 || P < 0 || total_kv < 0 || total_kv > 0x7fffffffLL ||
        (NE > 0 && (!node_kv || !node_kv_offset || !node_kv_len || !node_q || !node_q_offset || !node_q_len))) {
        set_error("bad Node metadata (NE=%d P=%d total_kv=%lld)", NE, P, (long long)total_kv);
        return DEFT_EINVAL;
    }
    if (!workspace) {
        set_error("null workspace");
        return DEFT_EINVAL;
    }
    rc = check_append(ap, Hkv, D);
    if (rc) return rc;
    const int G = Hq / Hkv;
    const int64_t tiles = node_max_tiles(NE, total_kv);
    const int64_t rows = tiles * (rows_per_tile > 0 ? rows_per_tile : node_rows_per_tile(P));  // every entry has at most that many queries
    const Workspace ws = carve(workspace, Hq, D, rows, tiles, plan_view(nullptr, tiles * G, rows).bytes);
    if (workspace_bytes < ws.bytes) {
        set_error("workspace too small: %zu < %zu", workspace_bytes, ws.bytes);
        return DEFT_EWORKSPACE;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    Stage1Params p{};
    p.q = static_cast<const _Float16*>(q);
    p.q_st = q_stride_tok;
    p.q_sh = q_stride_head;
    p.k = static_cast<const _Float16*>(k_base);
    p.v = static_cast<const _Float16*>(v_base);
    p.kv_ss = kv_stride_slot;
    p.kv_sh = kv_stride_head;
    p.node_kv = node_kv;
    p.node_kv_offset = node_kv_offset;
    p.node_kv_len = node_kv_len;
    p.node_q = node_q;
    p.node_q_offset = node_q_offset;
    p.node_q_len = node_q_len;
    p.partial_o = ws.partial_o;
    p.partial_lse = ws.partial_lse;
    p.row_q = ws.row_q;
    p.rows = rows;
    p.Hkv = Hkv;
    p.G = G;
    p.scale_log2e = scale * LOG2E;
    const bool hd2 = hd2_geometry(D, Hkv, kv_stride_head) && !ap.cos_sin && rows_per_tile != 1;
    if (D == 128 || hd2) {
        PlanView pv;
        if (plan) {
            pv = plan_view(const_cast<void*>(plan), tiles * G, rows);
        } else {
            pv = plan_view(ws.plan, tiles * G, rows);
            rc = launch_node_plan(p, NE, rows, pv, ap, st);
            if (rc) return rc;
        }
        rc = launch_stage1_np(p, tiles * G, pv, ap, st, nq, /*reread=*/rows_per_tile == 1, hd2);
        if (rc) return rc;
        return launch_merge(D, ws, &pv, pv.row_q, rows, out, o_stride_tok, o_stride_head, nq, Hq, G * (hd2 ? 2 : 1), st);
    }
    // tile-per-workgroup form (head_dim 64)
    if (ap.k_new) {  // separate append launch first
        rc = deft_kv_append_f16(const_cast<void*>(k_base), const_cast<void*>(v_base), kv_stride_slot, kv_stride_head,
                                ap.cache_loc, ap.k_new, ap.v_new, ap.new_st, ap.n_new, Hkv, D, stream);
        if (rc) return rc;
    }
    const size_t prep_lds = sizeof(int) * 2 * (size_t)(NE + 1);
    if (prep_lds > 64 * 1024) {
        set_error("node mode: %d entries exceed the prep kernel's LDS scan", NE);
        return DEFT_EUNSUPPORTED;
    }
    hipLaunchKernelGGL(node_prep_kernel, dim3(1), dim3(256), prep_lds, st, node_kv_offset, node_kv_len, node_q_offset,
                       node_q_len, NE, tiles, rows, ws.desc, ws.row_q);
    rc = check_launch("node prep launch");
    if (rc) return rc;
    p.desc = ws.desc;
    rc = launch_stage1_d64<1>(D, p, tiles, st);
    if (rc) return rc;
    return launch_merge(D, ws, nullptr, ws.row_q, rows, out, o_stride_tok, o_stride_head, nq, Hq, 1, st);
}

extern "C" {

int deft_node_decode_f16(const void* q, int64_t q_stride_tok, int64_t q_stride_head, const void* k_base,
                         const void* v_base, int64_t kv_stride_slot, int64_t kv_stride_head, void* out,
                         int64_t o_stride_tok, int64_t o_stride_head, const int64_t* node_kv,
                         const int64_t* node_kv_offset, const int64_t* node_kv_len, const int64_t* node_q,
                         const int64_t* node_q_offset, const int64_t* node_q_len, int NE, int P, int64_t total_kv, int nq,
                         int Hq, int Hkv, int D, float scale, const void* plan, void* workspace, size_t workspace_bytes,
                         void* stream) {
    return node_decode_impl(q, q_stride_tok, q_stride_head, k_base, v_base, kv_stride_slot, kv_stride_head, out,
                            o_stride_tok, o_stride_head, node_kv, node_kv_offset, node_kv_len, node_q, node_q_offset,
                            node_q_len, NE, P, total_kv, nq, Hq, Hkv, D, scale, plan, workspace, workspace_bytes, stream,
                            AppendArgs());
}

int deft_node_decode_append_f16(const void* q, int64_t q_stride_tok, int64_t q_stride_head, void* k_base, void* v_base,
                                int64_t kv_stride_slot, int64_t kv_stride_head, void* out, int64_t o_stride_tok,
                                int64_t o_stride_head, const int64_t* node_kv, const int64_t* node_kv_offset,
                                const int64_t* node_kv_len, const int64_t* node_q, const int64_t* node_q_offset,
                                const int64_t* node_q_len, int NE, int P, int64_t total_kv, int nq, int Hq, int Hkv, int D,
                                float scale, const int32_t* cache_loc, const void* k_new, const void* v_new,
                                int64_t new_stride_tok, int n_new, const void* plan, void* workspace,
                                size_t workspace_bytes, void* stream) {
    AppendArgs ap;
    ap.k_new = static_cast<const _Float16*>(k_new);
    ap.v_new = static_cast<const _Float16*>(v_new);
    ap.cache_loc = cache_loc;
    ap.new_st = new_stride_tok;
    ap.n_new = n_new;
    if (!k_new || !v_new || !cache_loc) {
        set_error("fused append needs k_new, v_new and cache_loc");
        return DEFT_EINVAL;
    }
    return node_decode_impl(q, q_stride_tok, q_stride_head, k_base, v_base, kv_stride_slot, kv_stride_head, out,
                            o_stride_tok, o_stride_head, node_kv, node_kv_offset, node_kv_len, node_q, node_q_offset,
                            node_q_len, NE, P, total_kv, nq, Hq, Hkv, D, scale, plan, workspace, workspace_bytes, stream, ap);
}

int deft_node_decode_rope_append_f16(const void* q, int64_t q_stride_tok, int64_t q_stride_head, void* k_base, void* v_base,
                                     int64_t kv_stride_slot, int64_t kv_stride_head, void* out, int64_t o_stride_tok,
                                     int64_t o_stride_head, const int64_t* node_kv, const int64_t* node_kv_offset,
                                     const int64_t* node_kv_len, const int64_t* node_q, const int64_t* node_q_offset,
                                     const int64_t* node_q_len, int NE, int P, int64_t total_kv, int nq, int Hq, int Hkv, int D,
                                     float scale, const int32_t* cache_loc, const void* k_new, const void* v_new,
                                     int64_t new_stride_tok, int n_new, const float* cos_sin_rows, int rotary_dim,
                                     int is_neox_style, const void* plan, void* workspace, size_t workspace_bytes,
                                     void* stream) {
    AppendArgs ap;
    ap.k_new = static_cast<const _Float16*>(k_new);
    ap.v_new = static_cast<const _Float16*>(v_new);
    ap.cache_loc = cache_loc;
    ap.new_st = new_stride_tok;
    ap.n_new = n_new;
    if (!k_new || !v_new || !cache_loc) {
        set_error("fused append needs k_new, v_new and cache_loc");
        return DEFT_EINVAL;
    }
    const int rc = fill_rope(ap, cos_sin_rows, rotary_dim, is_neox_style, q_stride_tok, q_stride_head, Hq, D);
    if (rc) return rc;
    return node_decode_impl(q, q_stride_tok, q_stride_head, k_base, v_base, kv_stride_slot, kv_stride_head, out,
                            o_stride_tok, o_stride_head, node_kv, node_kv_offset, node_kv_len, node_q, node_q_offset,
                            node_q_len, NE, P, total_kv, nq, Hq, Hkv, D, scale, plan, workspace, workspace_bytes, stream, ap);
}

int deft_kv_append_f16(void* k_base, void* v_base, int64_t kv_stride_slot, int64_t kv_stride_head,
                       const int32_t* cache_loc, const void* k_new, const void* v_new, int64_t new_stride_tok, int n,
                       int Hkv, int D, void* stream) {
    if (n == 0) return DEFT_OK;
    if (!k_base || !v_base || !cache_loc || !k_new || !v_new || n < 0 || Hkv <= 0 || D <= 0 || (D % 8)) {
        set_error("bad kv_append arguments (n=%d Hkv=%d D=%d)", n, Hkv, D);
        return DEFT_EINVAL;
    }
    if (!aligned16(k_base) || !aligned16(v_base) || !aligned16(k_new) || !aligned16(v_new) || (kv_stride_slot % 8) ||
        (kv_stride_head % 8) || (new_stride_tok % 8)) {
        set_error("kv_append rows must be 16-byte aligned");
        return DEFT_EINVAL;
    }
    const int64_t total = (int64_t)n * Hkv * (D / 8);
    const int64_t blocks = (total + 255) / 256;
    const unsigned grid = (unsigned)(blocks < 2048 ? blocks : 2048);
    hipLaunchKernelGGL(kv_append_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<_Float16*>(k_base), static_cast<_Float16*>(v_base), kv_stride_slot, kv_stride_head,
                       cache_loc, static_cast<const _Float16*>(k_new), static_cast<const _Float16*>(v_new),
                       new_stride_tok, n, Hkv, D);
    return check_launch("kv_append launch");
}

int deft_flatten_read_partials(const void* workspace, size_t workspace_bytes, int NB, int P, int nq, int Hq, int Hkv, int D,
                               float* partial_o_dev, float* partial_lse_dev, void* stream) {
    (void)nq;
    (void)Hkv;
    if (!workspace || !partial_o_dev || !partial_lse_dev) {
        set_error("null pointer");
        return DEFT_EINVAL;
    }
    const Workspace ws = carve(const_cast<void*>(workspace), Hq, D, P, 0, plan_view(nullptr, flatten_unit_cap(NB, Hq / Hkv), P).bytes);
    if (workspace_bytes < ws.bytes) {
        set_error("workspace too small");
        return DEFT_EWORKSPACE;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemcpyAsync(partial_o_dev, ws.partial_o, sizeof(float) * (size_t)Hq * P * D, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess)
        e = hipMemcpyAsync(partial_lse_dev, ws.partial_lse, sizeof(float) * (size_t)Hq * P, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) {
        set_error("hipMemcpyAsync: %s", hipGetErrorString(e));
        return DEFT_EHIP;
    }
    return DEFT_OK;
}

}  // extern "C"


// ---------------------------------------------------------------------------
// Sequential (per-request) paged attention: the reference's comparator,
// token_attention_fwd (DeFT/deft/layers/attention/token_attention.py:297-335) behind
// DeFTAttention.radix_attention_forward (deft_attention.py:153-188).  Every request attends
// to its own full path through the page table; shared prefixes are re-read per request -- that
// IS the baseline DeFT is measured against.  Implemented on the Node machinery: request i is
// an entry with node_kv = req_to_token[b_req_idx[i], :b_seq_len[i]] and the single query row i,
// so the same stage-1 / merge kernels run (the reference materialises a [Hq, total_tokens]
// logit matrix instead, token_attention.py:312-314).
// ---------------------------------------------------------------------------
namespace deft {

__global__ __launch_bounds__(256) void seq_to_node_kernel(const int32_t* req_to_token, int64_t req_stride,
                                                           const int32_t* b_req_idx, const int32_t* b_start_loc,
                                                           const int32_t* b_seq_len, int nq, int64_t total, int64_t* node_kv,
                                                           int64_t* node_kv_offset, int64_t* node_kv_len, int64_t* node_q,
                                                           int64_t* node_q_offset, int64_t* node_q_len, int32_t* hdr) {
    const int i = blockIdx.x;
    if (i >= nq) return;
    // start / len come from device memory, the buffer was sized from the host's total_num_tokens: a request that does
    // not fit is cut short (and flagged in the plan header) instead of writing past the end
    int64_t start = b_start_loc[i];
    int64_t len = b_seq_len[i];
    if (start < 0 || len < 0 || start + len > total) {
        if (blockIdx.y == 0 && threadIdx.x == 0) atomicOr(hdr + HDR_ERR, 2);
        start = start < 0 ? 0 : (start > total ? total : start);
        len = len < 0 ? 0 : (start + len > total ? total - start : len);
    }
    const int32_t* row = req_to_token + (int64_t)b_req_idx[i] * req_stride;
    for (int64_t j = blockIdx.y * blockDim.x + threadIdx.x; j < len; j += gridDim.y * blockDim.x) node_kv[start + j] = row[j];
    if (blockIdx.y == 0 && threadIdx.x == 0) {
        node_kv_offset[i] = start;
        node_kv_len[i] = len;
        node_q[i] = i;
        node_q_offset[i] = i;
        node_q_len[i] = 1;
    }
}

struct SeqPlan {  // [node plan | node_kv | node_kv_offset | node_kv_len | node_q | node_q_offset | node_q_len]
    void* node_plan;
    size_t node_plan_bytes;
    int64_t *node_kv, *node_kv_offset, *node_kv_len, *node_q, *node_q_offset, *node_q_len;
    size_t bytes;
};
static SeqPlan seq_plan_view(void* base, int nq, int64_t total, int Hq, int Hkv) {
    SeqPlan v;
    char* p = static_cast<char*>(base);
    const int64_t tiles = node_max_tiles(nq, total);
    v.node_plan = p;
    v.node_plan_bytes = plan_view(nullptr, tiles * (Hq / Hkv), tiles).bytes;
    size_t off = align_up(v.node_plan_bytes, 256);
    auto take = [&](int64_t n) {
        int64_t* r = reinterpret_cast<int64_t*>(p + off);
        off = align_up(off + sizeof(int64_t) * (size_t)(n > 0 ? n : 1), 256);
        return r;
    };
    v.node_kv = take(total);
    v.node_kv_offset = take(nq);
    v.node_kv_len = take(nq);
    v.node_q = take(nq);
    v.node_q_offset = take(nq);
    v.node_q_len = take(nq);
    v.bytes = off;
    return v;
}

}  // namespace deft

extern "C" {

size_t deft_seq_plan_bytes(int nq, int64_t total_tokens, int Hq, int Hkv) {
    if (nq < 0 || total_tokens < 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv) return 0;
    return seq_plan_view(nullptr, nq, total_tokens, Hq, Hkv).bytes;
}

size_t deft_seq_workspace_bytes(int nq, int64_t total_tokens, int Hq, int Hkv, int D) {
    if (nq < 0 || total_tokens < 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv) return 0;
    const int64_t tiles = node_max_tiles(nq, total_tokens);
    return carve(nullptr, Hq, D, tiles, tiles, plan_view(nullptr, tiles * (Hq / Hkv), tiles).bytes).bytes;
}

int deft_seq_build_plan(const int32_t* req_to_token, int64_t req_stride, const int32_t* b_req_idx,
                        const int32_t* b_start_loc, const int32_t* b_seq_len, int nq, int64_t total_tokens, int Hq,
                        int Hkv, int64_t q_stride_tok, int64_t q_stride_head, int64_t kv_stride_slot,
                        const int32_t* cache_loc, int n_new, int64_t new_stride_tok, void* plan, size_t plan_bytes,
                        void* stream) {
    if (nq < 0 || total_tokens < 0 || total_tokens > 0x7fffffffLL || !plan || Hq <= 0 || Hkv <= 0 || Hq % Hkv ||
        (nq > 0 && (!req_to_token || !b_req_idx || !b_start_loc || !b_seq_len))) {
        set_error("bad sequential plan arguments (nq=%d total_tokens=%lld)", nq, (long long)total_tokens);
        return DEFT_EINVAL;
    }
    const SeqPlan sp = seq_plan_view(plan, nq, total_tokens, Hq, Hkv);
    if (plan_bytes < sp.bytes) {
        set_error("plan buffer too small: %zu < %zu", plan_bytes, sp.bytes);
        return DEFT_EWORKSPACE;
    }
    if (nq == 0) return DEFT_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(static_cast<int32_t*>(sp.node_plan) + HDR_ERR, 0, sizeof(int32_t), st) != hipSuccess) {
        set_error("hipMemsetAsync(plan header) failed");
        return DEFT_EHIP;
    }
    const unsigned gy = (unsigned)((total_tokens / nq + 255) / 256 < 1 ? 1 : ((total_tokens / nq + 255) / 256 > 64 ? 64 : (total_tokens / nq + 255) / 256));
    hipLaunchKernelGGL(seq_to_node_kernel, dim3((unsigned)nq, gy), dim3(256), 0, st, req_to_token, req_stride, b_req_idx,
                       b_start_loc, b_seq_len, nq, total_tokens, sp.node_kv, sp.node_kv_offset, sp.node_kv_len, sp.node_q,
                       sp.node_q_offset, sp.node_q_len, static_cast<int32_t*>(sp.node_plan));
    int rc = check_launch("seq_to_node launch");
    if (rc) return rc;
    const int64_t tiles = node_max_tiles(nq, total_tokens);
    const PlanView pv = plan_view(sp.node_plan, tiles * (Hq / Hkv), tiles);
    Stage1Params p{};
    p.node_kv = sp.node_kv;
    p.node_kv_offset = sp.node_kv_offset;
    p.node_kv_len = sp.node_kv_len;
    p.node_q = sp.node_q;
    p.node_q_offset = sp.node_q_offset;
    p.node_q_len = sp.node_q_len;
    p.rows = tiles;
    p.G = Hq / Hkv;
    p.Hkv = Hkv;
    p.q_st = q_stride_tok;
    p.q_sh = q_stride_head;
    p.kv_ss = kv_stride_slot;
    AppendArgs ap;
    if (cache_loc) {
        ap.cache_loc = cache_loc;
        ap.n_new = n_new;
        ap.new_st = new_stride_tok;
    }
    return launch_node_plan(p, nq, tiles, pv, ap, st, 1);
}

static int seq_decode_impl(const void* q, int64_t q_stride_tok, int64_t q_stride_head, const void* k_base,
                           const void* v_base, int64_t kv_stride_slot, int64_t kv_stride_head, void* out,
                           int64_t o_stride_tok, int64_t o_stride_head, const void* plan, int nq, int64_t total_tokens,
                           int Hq, int Hkv, int D, float scale, void* workspace, size_t workspace_bytes, void* stream,
                           const AppendArgs& ap) {
    if (!plan || nq < 0 || total_tokens < 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv) {
        set_error("bad sequential decode arguments (nq=%d total_tokens=%lld)", nq, (long long)total_tokens);
        return DEFT_EINVAL;
    }
    const SeqPlan sp = seq_plan_view(const_cast<void*>(plan), nq, total_tokens, Hq, Hkv);
    return node_decode_impl(q, q_stride_tok, q_stride_head, k_base, v_base, kv_stride_slot, kv_stride_head, out,
                            o_stride_tok, o_stride_head, sp.node_kv, sp.node_kv_offset, sp.node_kv_len, sp.node_q,
                            sp.node_q_offset, sp.node_q_len, nq, nq, total_tokens, nq, Hq, Hkv, D, scale,
                            D == 128 ? sp.node_plan : nullptr, workspace, workspace_bytes, stream, ap, 1);
}

int deft_seq_decode_f16(const void* q, int64_t q_stride_tok, int64_t q_stride_head, const void* k_base, const void* v_base,
                        int64_t kv_stride_slot, int64_t kv_stride_head, void* out, int64_t o_stride_tok,
                        int64_t o_stride_head, const void* plan, int nq, int64_t total_tokens, int Hq, int Hkv, int D,
                        float scale, void* workspace, size_t workspace_bytes, void* stream) {
    return seq_decode_impl(q, q_stride_tok, q_stride_head, k_base, v_base, kv_stride_slot, kv_stride_head, out,
                           o_stride_tok, o_stride_head, plan, nq, total_tokens, Hq, Hkv, D, scale, workspace,
                           workspace_bytes, stream, AppendArgs());
}

int deft_seq_decode_append_f16(const void* q, int64_t q_stride_tok, int64_t q_stride_head, void* k_base, void* v_base,
                               int64_t kv_stride_slot, int64_t kv_stride_head, void* out, int64_t o_stride_tok,
                               int64_t o_stride_head, const void* plan, int nq, int64_t total_tokens, int Hq, int Hkv,
                               int D, float scale, const int32_t* cache_loc, const void* k_new, const void* v_new,
                               int64_t new_stride_tok, int n_new, void* workspace, size_t workspace_bytes, void* stream) {
    if (!k_new || !v_new || !cache_loc) {
        set_error("fused append needs k_new, v_new and cache_loc");
        return DEFT_EINVAL;
    }
    AppendArgs ap;
    ap.k_new = static_cast<const _Float16*>(k_new);
    ap.v_new = static_cast<const _Float16*>(v_new);
    ap.cache_loc = cache_loc;
    ap.new_st = new_stride_tok;
    ap.n_new = n_new;
    return seq_decode_impl(q, q_stride_tok, q_stride_head, k_base, v_base, kv_stride_slot, kv_stride_head, out,
                           o_stride_tok, o_stride_head, plan, nq, total_tokens, Hq, Hkv, D, scale, workspace,
                           workspace_bytes, stream, ap);
}

}  // extern "C"


// ---------------------------------------------------------------------------
// Rotary position embedding of this step's q and k rows, in place: the op right in front of the attention path
// (LlamaAttention.forward, DeFT/deft/models/llama2.py:108-110 -> RotaryEmbedding.forward_cuda,
// DeFT/deft/layers/rotary_embedding.py:157-177, which calls flashinfer.rope.apply_rope_with_cos_sin_cache_inplace:
// NeoX pairing (d, d + rot/2), cos|sin cache in fp32 (llama2.py:86-93 passes dtype=float32), arithmetic in fp32,
// one rounding back to fp16).  flashinfer is not part of the reference tree; its published algorithm is restated
// in oracle/rope.py.  No FMA contraction, so the fp32 products and sums are the oracle's.
// ---------------------------------------------------------------------------
namespace deft {

__global__ __launch_bounds__(256) void rope_qk_kernel(_Float16* q, int64_t q_st, int64_t q_sh, int Hq, _Float16* k,
                                                       int64_t k_st, int64_t k_sh, int Hk, const int64_t* positions,
                                                       const float* cos_sin, int64_t cache_stride, int n, int D, int rot,
                                                       int neox) {
#pragma clang fp contract(off)
    const int half = rot / 2;
    const int per_tok = (Hq + Hk) * half;
    const int64_t total = (int64_t)n * per_tok;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int tok = (int)(i / per_tok);
        const int rem = (int)(i - (int64_t)tok * per_tok);
        const int hd = rem / half;
        const int j = rem - hd * half;
        _Float16* row = hd < Hq ? q + (int64_t)tok * q_st + (int64_t)hd * q_sh : k + (int64_t)tok * k_st + (int64_t)(hd - Hq) * k_sh;
        const float* cs = cos_sin + positions[tok] * cache_stride;
        const float c = cs[j], sn = cs[half + j];
        const int i1 = neox ? j : 2 * j, i2 = neox ? j + half : 2 * j + 1;
        const float x1 = (float)row[i1], x2 = (float)row[i2];
        const float o1 = x1 * c - x2 * sn;
        const float o2 = x2 * c + x1 * sn;
        row[i1] = (_Float16)o1;
        row[i2] = (_Float16)o2;
    }
}

}  // namespace deft

extern "C" int deft_rope_qk_f16(void* q, int64_t q_stride_tok, int64_t q_stride_head, int Hq, void* k, int64_t k_stride_tok,
                                int64_t k_stride_head, int Hk, const int64_t* positions, const float* cos_sin_cache,
                                int64_t cache_stride, int n, int D, int rotary_dim, int is_neox_style, void* stream) {
    if (n == 0) return DEFT_OK;
    if (!q || !k || !positions || !cos_sin_cache || n < 0 || Hq <= 0 || Hk < 0 || D <= 0 || rotary_dim <= 0 ||
        rotary_dim > D || (rotary_dim & 1) || cache_stride < rotary_dim) {
        set_error("bad rope arguments (n=%d Hq=%d Hk=%d D=%d rotary_dim=%d)", n, Hq, Hk, D, rotary_dim);
        return DEFT_EINVAL;
    }
    const int64_t total = (int64_t)n * (Hq + Hk) * (rotary_dim / 2);
    const int64_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(deft::rope_qk_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<_Float16*>(q), q_stride_tok, q_stride_head, Hq,
                       static_cast<_Float16*>(k), k_stride_tok, k_stride_head, Hk, positions, cos_sin_cache, cache_stride, n,
                       D, rotary_dim, is_neox_style ? 1 : 0);
    return deft::check_launch("rope launch");
}

namespace deft {
__global__ __launch_bounds__(256) void rope_gather_kernel(const int64_t* positions, const float* cos_sin, int64_t cache_stride,
                                                         int n, int rot, float* out) {
    const int64_t total = (int64_t)n * rot;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / rot), j = (int)(i - (int64_t)r * rot);
        out[i] = cos_sin[positions[r] * cache_stride + j];
    }
}
}  // namespace deft

extern "C" int deft_rope_gather_rows(const int64_t* positions, const float* cos_sin_cache, int64_t cache_stride, int n,
                                     int rotary_dim, float* rows_out, void* stream) {
    if (n == 0) return DEFT_OK;
    if (!positions || !cos_sin_cache || !rows_out || n < 0 || rotary_dim <= 0 || cache_stride < rotary_dim) {
        set_error("bad rope-gather arguments (n=%d rotary_dim=%d)", n, rotary_dim);
        return DEFT_EINVAL;
    }
    const int64_t blocks = ((int64_t)n * rotary_dim + 255) / 256;
    hipLaunchKernelGGL(deft::rope_gather_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), positions, cos_sin_cache, cache_stride, n, rotary_dim, rows_out);
    return deft::check_launch("rope gather launch");
}


// ---------------------------------------------------------------------------
// Causal prefill attention (prefill.h): context_attention_fwd
// (DeFT/deft/layers/attention/context_flashattention_nopad.py:130-195)
// ---------------------------------------------------------------------------
extern "C" int deft_prefill_f16(const void* q, int64_t q_stride_tok, int64_t q_stride_head, const void* k,
                                int64_t k_stride_tok, int64_t k_stride_head, const void* v, int64_t v_stride_tok,
                                int64_t v_stride_head, void* out, int64_t o_stride_tok, int64_t o_stride_head,
                                const int32_t* b_start_loc, const int32_t* b_seq_len, int batch, int max_input_len, int Hq,
                                int Hkv, int D, float scale, void* stream) {
    using namespace deft;
    if (batch == 0 || max_input_len == 0) return DEFT_OK;
    if (!q || !k || !v || !out || !b_start_loc || !b_seq_len || batch < 0 || max_input_len < 0 || Hq <= 0 || Hkv <= 0 ||
        Hq % Hkv) {
        set_error("bad prefill arguments (batch=%d max_input_len=%d Hq=%d Hkv=%d)", batch, max_input_len, Hq, Hkv);
        return DEFT_EINVAL;
    }
    if (D != 128 && D != 64 && D != 32 && D != 16) {
        set_error("prefill: unsupported head_dim %d (supported: 16, 32, 64, 128)", D);
        return DEFT_EUNSUPPORTED;
    }
    if (!aligned16(q) || !aligned16(k) || !aligned16(v) || (reinterpret_cast<uintptr_t>(out) & 7u) || (q_stride_tok % 8) ||
        (q_stride_head % 8) || (k_stride_tok % 8) || (k_stride_head % 8) || (v_stride_tok % 8) || (v_stride_head % 8) ||
        (o_stride_tok % 4) || (o_stride_head % 4)) {
        set_error("prefill: q/k/v rows must be 16-byte aligned, out rows 8-byte aligned");
        return DEFT_EINVAL;
    }
    if (D == 128) {
        const int rc = raise_lds(reinterpret_cast<const void*>(&prefill_kernel<128>), PrefillSmem<128>::BYTES, ATTR_PREFILL, "prefill");
        if (rc) return rc;
    } else if (D == 64) {
        const int rc = raise_lds(reinterpret_cast<const void*>(&prefill_kernel<64>), PrefillSmem<64>::BYTES, ATTR_PREFILL_64, "prefill (head_dim 64)");
        if (rc) return rc;
    }
    PrefillParams p{};
    p.q = static_cast<const _Float16*>(q);
    p.k = static_cast<const _Float16*>(k);
    p.v = static_cast<const _Float16*>(v);
    p.o = static_cast<_Float16*>(out);
    p.q_st = q_stride_tok;
    p.q_sh = q_stride_head;
    p.k_st = k_stride_tok;
    p.k_sh = k_stride_head;
    p.v_st = v_stride_tok;
    p.v_sh = v_stride_head;
    p.o_st = o_stride_tok;
    p.o_sh = o_stride_head;
    p.b_start_loc = b_start_loc;
    p.b_seq_len = b_seq_len;
    p.G = Hq / Hkv;
    p.scale_log2e = scale * LOG2E;
    p.nblk = (max_input_len + 255) / 256;
    p.Hq = Hq;
    p.batch = batch;
    p.dbg = g_dbg;
    if ((int64_t)p.nblk * Hq * batch > 0x7fffffffLL) {
        set_error("prefill grid too large (%d query blocks x %d heads x %d sequences)", p.nblk, Hq, batch);
        return DEFT_EINVAL;
    }
    const dim3 grid((unsigned)((int64_t)p.nblk * Hq * batch));
    if (D == 128) hipLaunchKernelGGL((prefill_kernel<128>), grid, dim3(512), PrefillSmem<128>::BYTES, static_cast<hipStream_t>(stream), p);
    else if (D == 64) hipLaunchKernelGGL((prefill_kernel<64>), grid, dim3(512), PrefillSmem<64>::BYTES, static_cast<hipStream_t>(stream), p);
    else {  // head_dim 32 / 16: a wave per (token, head)
        const dim3 g2((unsigned)((((int64_t)p.nblk * 256 + 3) / 4) * batch), (unsigned)Hq);
        if (D == 32) hipLaunchKernelGGL((prefill_small_kernel<32>), g2, dim3(256), 0, static_cast<hipStream_t>(stream), p);
        else hipLaunchKernelGGL((prefill_small_kernel<16>), g2, dim3(256), 0, static_cast<hipStream_t>(stream), p);
    }
    return check_launch("prefill launch");
}


// ---------------------------------------------------------------------------
// Device-side TreeMetadata (tree_plan.h): TreeMetadata.from_tree_cache (tree_cache.py:618-881) from the compact tree
// that lives on the GPU.  The host-side tree (tree.cpp) lays the tree out and computes the sizes; these calls
// only launch.
// ---------------------------------------------------------------------------
namespace deft {
static TreeScratch tree_scratch_view(void* base, int n, int nqw, int nbp_cap, size_t* bytes) {
    TreeScratch s;
    char* p = static_cast<char*>(base);
    size_t off = 0;
    auto take = [&](size_t count, size_t elem) {
        char* r = p + off;
        off = align_up(off + count * elem, 256);
        return r;
    };
    s.dims = reinterpret_cast<int32_t*>(take(TREE_DIMS, 4));
    s.pos = reinterpret_cast<int32_t*>(take((size_t)n + 1, 4));
    s.e_off = reinterpret_cast<int32_t*>(take((size_t)n + 1, 4));
    s.q_off = reinterpret_cast<int32_t*>(take((size_t)n + 1, 4));
    s.kv_off = reinterpret_cast<int32_t*>(take((size_t)n + 1, 4));
    s.b_first = reinterpret_cast<int32_t*>(take((size_t)nbp_cap + 1, 4));
    s.b_eoff = reinterpret_cast<int32_t*>(take((size_t)nbp_cap + 1, 4));
    s.b_poff = reinterpret_cast<int32_t*>(take((size_t)nbp_cap + 1, 4));
    s.b_union = reinterpret_cast<unsigned long long*>(take((size_t)(nbp_cap + 1) * (size_t)nqw, 8));
    *bytes = off;
    return s;
}
}  // namespace deft

extern "C" {

/* scratch of deft_tree_dev_build_md; its first 16 int32 are dims[]: query_num, NE, total_kv, len(node_q), len(node_kv),
 * NB, P, len(block_kv), physical blocks, error flags (bit 0: a leaf outgrew its room, bit 1: more blocks than nbp_cap) --
 * zero them when a layout is uploaded */
size_t deft_tree_dev_scratch_bytes(int n_nodes, int nqw, int nbp_cap) {
    if (n_nodes < 0 || nqw < 1 || nbp_cap < 0) return 0;
    size_t bytes = 0;
    tree_scratch_view(nullptr, n_nodes, nqw, nbp_cap, &bytes);
    return bytes;
}

int deft_tree_dev_advance(int n_nodes, int nq, int nqw, const int32_t* node_start, int32_t* node_len, const int32_t* node_cap,
                          const uint64_t* refs, const int32_t* leaf_node, int32_t* slots, const int32_t* cache_loc,
                          void* scratch, void* stream) {
    if (nq == 0) return DEFT_OK;
    if (n_nodes <= 0 || nq < 0 || nqw < 1 || !node_start || !node_len || !node_cap || !refs || !leaf_node || !slots || !cache_loc ||
        !scratch) {
        set_error("deft_tree_dev_advance: bad arguments (nodes=%d nq=%d)", n_nodes, nq);
        return DEFT_EINVAL;
    }
    TreeDev t{n_nodes, nq, nqw, node_start, node_len, node_cap, reinterpret_cast<const unsigned long long*>(refs), leaf_node, slots};
    hipLaunchKernelGGL(tree_advance_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), t,
                       cache_loc, static_cast<int32_t*>(scratch) + TREE_ERR);
    return check_launch("tree advance launch");
}

// Replay the host tree's journal of absorbed changes (deft_tree_journal_take: words[0 .. n)) on the device copy: `ops` is a DEVICE
// buffer {n, words ...}.  deft_tree_dev_build_md_ops does the same inside its first kernel.
int deft_tree_dev_apply_ops(int n_nodes, int nq, int nqw, const int32_t* node_start, int32_t* node_len, const int32_t* node_cap,
                            const uint64_t* refs, const int32_t* leaf_node, int32_t* slots, const int32_t* ops, void* scratch,
                            void* stream) {
    if (n_nodes <= 0 || nq < 0 || nqw < 1 || !node_start || !node_len || !node_cap || !refs || !leaf_node || !slots || !ops || !scratch) {
        set_error("deft_tree_dev_apply_ops: bad arguments (nodes=%d nq=%d)", n_nodes, nq);
        return DEFT_EINVAL;
    }
    TreeDev t{n_nodes, nq, nqw, node_start, node_len, node_cap, reinterpret_cast<const unsigned long long*>(refs), leaf_node, slots};
    hipLaunchKernelGGL(tree_ops_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), t, ops,
                       static_cast<int32_t*>(scratch) + TREE_ERR);
    return check_launch("tree ops launch");
}

static int tree_dev_build_md_impl(int n_nodes, int nq, int nqw, const int32_t* node_start, int32_t* node_len, const int32_t* node_cap,
                           const uint64_t* refs, const int32_t* leaf_node, int32_t* slots, int max_q_len, int block_len,
                           int max_block_len, int nbp_cap, void* scratch, size_t scratch_bytes, int64_t* node_q, int64_t* node_kv,
                           int64_t* node_q_len, int64_t* node_kv_len, int64_t* node_q_offset, int64_t* node_kv_offset,
                           int64_t* block_q, int64_t* block_q_cnts, int64_t* block_q_offset, int64_t* block_bitmasks,
                           int64_t* block_kv, int64_t* block_lens, const int32_t* advance_loc, const int32_t* ops, PageWrite pw,
                           void* stream) {
    if (n_nodes <= 0 || nq < 0 || nqw < 1 || nbp_cap < 0 || !node_start || !node_len || !node_cap || !refs || !leaf_node || !slots ||
        !scratch) {
        set_error("deft_tree_dev_build_md: bad arguments (nodes=%d nq=%d)", n_nodes, nq);
        return DEFT_EINVAL;
    }
    // either group of six arrays may be omitted as a whole (all six pointers null): a DeFT-Flatten step reads only the
    // block arrays, a DeFT-Node step only the node arrays, and the kernel that writes the other six is then not launched
    const int n_node_ptrs = !!node_q + !!node_kv + !!node_q_len + !!node_kv_len + !!node_q_offset + !!node_kv_offset;
    const int n_block_ptrs = !!block_q + !!block_q_cnts + !!block_q_offset + !!block_bitmasks + !!block_kv + !!block_lens;
    if ((n_node_ptrs != 0 && n_node_ptrs != 6) || (n_block_ptrs != 0 && n_block_ptrs != 6) || n_node_ptrs + n_block_ptrs == 0) {
        set_error("deft_tree_dev_build_md: the node arrays and the block arrays are each given as a whole or not at all");
        return DEFT_EINVAL;
    }
    if (max_q_len < 1 || max_q_len > 63 || block_len < 1 || block_len > 1024 || (max_block_len < 1 && max_block_len != -1)) {
        set_error("deft_tree_dev_build_md: bad config max_q_len=%d block_len=%d max_block_len=%d", max_q_len, block_len, max_block_len);
        return DEFT_EINVAL;
    }
    size_t need = 0;
    const TreeScratch sc = tree_scratch_view(scratch, n_nodes, nqw, nbp_cap, &need);
    if (scratch_bytes < need) {
        set_error("tree scratch too small: %zu < %zu", scratch_bytes, need);
        return DEFT_EWORKSPACE;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    TreeDev t{n_nodes, nq, nqw, node_start, node_len, node_cap, reinterpret_cast<const unsigned long long*>(refs), leaf_node, slots};
    TreeMdOut o{node_q, node_kv, node_q_len, node_kv_len, node_q_offset, node_kv_offset,
                block_q, block_q_cnts, block_q_offset, block_bitmasks, block_kv, block_lens};
    // node tables in LDS when the tree fits: 6 x (n + 1) words + one per block
    size_t scan_lds = 0;
    if (n_nodes <= TREE_LDS_NODES) {
        scan_lds = sizeof(int32_t) * 6 * ((size_t)n_nodes + 1);
        if (nbp_cap <= TREE_LDS_BLOCKS) scan_lds += sizeof(int32_t) * ((size_t)nbp_cap + 1);
    }
    // (dynamic part: at most 6 x 4097 + 8193 words = 131 KB of tables, next to 16 KB of static LDS for the journal replay)
    int rc = raise_lds(reinterpret_cast<const void*>(&tree_md_scan_kernel), 136 * 1024, ATTR_TREE, "tree_md_scan");
    if (rc) return rc;
    hipLaunchKernelGGL(tree_md_scan_kernel, dim3(1), dim3(1024), scan_lds, st, t, sc, max_q_len, block_len, max_block_len, nbp_cap,
                       advance_loc, ops, pw);
    rc = check_launch("tree scan launch");
    if (rc) return rc;
    if (nbp_cap > 0 && n_block_ptrs) {
        hipLaunchKernelGGL(tree_md_blocks_kernel, dim3((unsigned)nbp_cap), dim3(128), 0, st, t, sc, o, max_q_len, block_len);
        rc = check_launch("tree blocks launch");
        if (rc) return rc;
    }
    if (!n_node_ptrs) return DEFT_OK;
    hipLaunchKernelGGL(tree_md_nodes_kernel, dim3((unsigned)n_nodes, 4), dim3(256), 0, st, t, sc, o, max_q_len, max_block_len);
    return check_launch("tree nodes launch");
}

int deft_tree_dev_build_md(int n_nodes, int nq, int nqw, const int32_t* node_start, int32_t* node_len, const int32_t* node_cap,
                           const uint64_t* refs, const int32_t* leaf_node, int32_t* slots, int max_q_len, int block_len,
                           int max_block_len, int nbp_cap, void* scratch, size_t scratch_bytes, int64_t* node_q, int64_t* node_kv,
                           int64_t* node_q_len, int64_t* node_kv_len, int64_t* node_q_offset, int64_t* node_kv_offset,
                           int64_t* block_q, int64_t* block_q_cnts, int64_t* block_q_offset, int64_t* block_bitmasks,
                           int64_t* block_kv, int64_t* block_lens, const int32_t* advance_loc, void* stream) {
    return tree_dev_build_md_impl(n_nodes, nq, nqw, node_start, node_len, node_cap, refs, leaf_node, slots, max_q_len, block_len,
                                  max_block_len, nbp_cap, scratch, scratch_bytes, node_q, node_kv, node_q_len, node_kv_len,
                                  node_q_offset, node_kv_offset, block_q, block_q_cnts, block_q_offset, block_bitmasks, block_kv,
                                  block_lens, advance_loc, nullptr, PageWrite{nullptr, 0, nullptr, nullptr}, stream);
}

// deft_tree_dev_build_md with the journal replay (deft_tree_dev_apply_ops) folded into its first kernel, in front of the
// advance: `ops` = device buffer {n, words ...}, read at run time -- n = 0 on a step without absorbed changes -- so that the
// launch has the same arguments on every step of an epoch (a captured decode step).  `page_table` (nullable, int32
// [requests][page_stride]): the same kernel also writes the page-table entries of the step's new tokens,
// page_table[page_rows[r]][page_cols[r]] = advance_loc[r] -- what TreeCache.alloc does with an index_put.
int deft_tree_dev_build_md_ops(int n_nodes, int nq, int nqw, const int32_t* node_start, int32_t* node_len, const int32_t* node_cap,
                               const uint64_t* refs, const int32_t* leaf_node, int32_t* slots, int max_q_len, int block_len,
                               int max_block_len, int nbp_cap, void* scratch, size_t scratch_bytes, int64_t* node_q,
                               int64_t* node_kv, int64_t* node_q_len, int64_t* node_kv_len, int64_t* node_q_offset,
                               int64_t* node_kv_offset, int64_t* block_q, int64_t* block_q_cnts, int64_t* block_q_offset,
                               int64_t* block_bitmasks, int64_t* block_kv, int64_t* block_lens, const int32_t* advance_loc,
                               const int32_t* ops, int32_t* page_table, int64_t page_stride, const int64_t* page_rows,
                               const int64_t* page_cols, void* stream) {
    if (page_table && (!advance_loc || !page_rows || !page_cols || page_stride <= 0)) {
        set_error("deft_tree_dev_build_md_ops: the page-table write needs advance_loc, rows, cols and a row stride");
        return DEFT_EINVAL;
    }
    return tree_dev_build_md_impl(n_nodes, nq, nqw, node_start, node_len, node_cap, refs, leaf_node, slots, max_q_len, block_len,
                                  max_block_len, nbp_cap, scratch, scratch_bytes, node_q, node_kv, node_q_len, node_kv_len,
                                  node_q_offset, node_kv_offset, block_q, block_q_cnts, block_q_offset, block_bitmasks, block_kv,
                                  block_lens, advance_loc, ops, PageWrite{page_table, page_stride, page_rows, page_cols}, stream);
}

}  // extern "C"

// ---------------------------------------------------------------------------
// Window plans (window.h): the incremental per-step head of deft_amd.DecodeSession.
// ---------------------------------------------------------------------------
extern "C" {

/* 1 when a tree step of this shape can run on a window plan: every (query chunk, 32-row pass) pair that hosts the overflow tiles
 * has a place in the tables. */
int deft_window_supported(int nq, int max_q_len, int Hq, int Hkv) {
    if (nq <= 0 || max_q_len < 1 || max_q_len > MQ || Hq <= 0 || Hkv <= 0 || Hq % Hkv) return 0;  // (at most one 32-query tile per chunk)
    const int G = Hq / Hkv;
    const int chunks = (nq + max_q_len - 1) / max_q_len;
    const int passes = (std::min(nq, max_q_len) * G + MQ - 1) / MQ;
    return passes <= WIN_PASSES && chunks * passes <= 64 ? 1 : 0;
}

int deft_flatten_build_plan_window(int64_t* block_q, int64_t* block_q_cnts, int64_t* block_q_offset, int64_t* block_bitmasks,
                                   int64_t* block_kv, int64_t* block_lens, int NB, int P, int32_t* dims, int nq, int max_q_len,
                                   int win_tiles, int32_t* win_tab, int Hq, int Hkv, int64_t q_stride_tok, int64_t q_stride_head,
                                   int64_t kv_stride_slot, void* plan, size_t plan_bytes, void* stream) {
    if (NB <= 0 || P <= 0 || !plan || !dims || !win_tab || win_tiles < 1 || !deft_window_supported(nq, max_q_len, Hq, Hkv) || !block_q ||
        !block_q_cnts || !block_q_offset || !block_bitmasks || !block_kv || !block_lens) {
        set_error("bad window plan arguments (NB=%d P=%d nq=%d max_q_len=%d tiles=%d)", NB, P, nq, max_q_len, win_tiles);
        return DEFT_EINVAL;
    }
    const PlanView pv = plan_view(plan, flatten_unit_cap(NB, Hq / Hkv), P);
    if (plan_bytes < pv.bytes) {
        set_error("plan buffer too small: %zu < %zu", plan_bytes, pv.bytes);
        return DEFT_EWORKSPACE;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    TreeMdOut o{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, block_q, block_q_cnts, block_q_offset, block_bitmasks, block_kv, block_lens};
    hipLaunchKernelGGL(window_blocks_kernel, dim3(1), dim3(256), 0, st, o, dims, nq, max_q_len, win_tiles, TILE, NB, P);
    int rc = check_launch("window blocks launch");
    if (rc) return rc;
    Stage1Params p{};
    p.block_q = block_q;
    p.block_q_cnts = block_q_cnts;
    p.block_q_offset = block_q_offset;
    p.block_bitmasks = block_bitmasks;
    p.block_kv = block_kv;
    p.block_lens = block_lens;
    p.rows = P;
    p.G = Hq / Hkv;
    p.Hkv = Hkv;
    p.q_st = q_stride_tok;
    p.q_sh = q_stride_head;
    p.kv_ss = kv_stride_slot;
    return launch_plan(p, NB, pv, AppendArgs(), st, dims, win_tiles, win_tab);
}

int deft_node_build_plan_window(int64_t* node_kv, int64_t* node_kv_offset, int64_t* node_kv_len, int64_t* node_q,
                                int64_t* node_q_offset, int64_t* node_q_len, int NE, int P, int64_t total_kv, int32_t* dims, int nq,
                                int max_q_len, int win_tiles, int32_t* win_tab, int Hq, int Hkv, int64_t q_stride_tok,
                                int64_t q_stride_head, int64_t kv_stride_slot, void* plan, size_t plan_bytes, void* stream) {
    if (NE <= 0 || P <= 0 || total_kv <= 0 || total_kv > 0x7fffffffLL || !plan || !dims || !win_tab || win_tiles < 1 ||
        !deft_window_supported(nq, max_q_len, Hq, Hkv) || !node_kv || !node_kv_offset || !node_kv_len || !node_q || !node_q_offset ||
        !node_q_len) {
        set_error("bad window plan arguments (NE=%d P=%d total_kv=%lld nq=%d)", NE, P, (long long)total_kv, nq);
        return DEFT_EINVAL;
    }
    const int64_t tiles = node_max_tiles(NE, total_kv);
    const int64_t rows = tiles * node_rows_per_tile(P);
    const PlanView pv = plan_view(plan, tiles * (Hq / Hkv), rows);
    if (plan_bytes < pv.bytes) {
        set_error("plan buffer too small: %zu < %zu", plan_bytes, pv.bytes);
        return DEFT_EWORKSPACE;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    TreeMdOut o{node_q, node_kv, node_q_len, node_kv_len, node_q_offset, node_kv_offset, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipLaunchKernelGGL(window_entries_kernel, dim3(1), dim3(256), 0, st, o, dims, nq, max_q_len, win_tiles, NE, P, (int)total_kv);
    int rc = check_launch("window entries launch");
    if (rc) return rc;
    Stage1Params p{};
    p.node_kv = node_kv;
    p.node_kv_offset = node_kv_offset;
    p.node_kv_len = node_kv_len;
    p.node_q = node_q;
    p.node_q_offset = node_q_offset;
    p.node_q_len = node_q_len;
    p.rows = rows;
    p.G = Hq / Hkv;
    p.Hkv = Hkv;
    p.q_st = q_stride_tok;
    p.q_sh = q_stride_head;
    p.kv_ss = kv_stride_slot;
    return launch_node_plan(p, NE, rows, pv, AppendArgs(), st, 0, dims, win_tiles, win_tab);
}

int deft_window_patch(int n_nodes, int nq, int nqw, const int32_t* node_start, int32_t* node_len, const int32_t* node_cap,
                      const uint64_t* refs, const int32_t* leaf_node, int32_t* slots, const int32_t* ops, const int32_t* cache_loc,
                      int32_t* page_table, int64_t page_stride, const int64_t* page_rows, const int64_t* page_cols,
                      const int32_t* patch, const int32_t* win_tab, void* plan, int max_q_len, int win_tiles, int Hq, int Hkv,
                      int64_t kv_stride_slot, int64_t new_stride_tok, void* scratch, const void* fetch_ring, int fetch_slot_bytes,
                      int fetch_ring_n, void* fetch_dst, int32_t* fetch_counter, void* stream) {
    if (n_nodes <= 0 || nq <= 0 || nqw < 1 || !node_start || !node_len || !node_cap || !refs || !leaf_node || !slots || !cache_loc ||
        !patch || !win_tab || !plan || !scratch || win_tiles < 1 || !deft_window_supported(nq, max_q_len, Hq, Hkv)) {
        set_error("deft_window_patch: bad arguments (nodes=%d nq=%d tiles=%d)", n_nodes, nq, win_tiles);
        return DEFT_EINVAL;
    }
    if (page_table && (!page_rows || !page_cols || page_stride <= 0)) {
        set_error("deft_window_patch: the page-table write needs rows, cols and a row stride");
        return DEFT_EINVAL;
    }
    TreeDev t{n_nodes, nq, nqw, node_start, node_len, node_cap, reinterpret_cast<const unsigned long long*>(refs), leaf_node, slots};
    WindowPatch w{ops, cache_loc, patch, win_tab, static_cast<char*>(plan) + PLAN_HDR, static_cast<int32_t*>(scratch) + TREE_ERR,
                  max_q_len, Hq / Hkv, win_tiles, kv_stride_slot, new_stride_tok * 2};
    if (fetch_ring && (!fetch_dst || !fetch_counter || fetch_slot_bytes < 32 || fetch_slot_bytes % 16 || fetch_ring_n < 1)) {
        set_error("deft_window_patch: bad fetch arguments");
        return DEFT_EINVAL;
    }
    hipLaunchKernelGGL(window_patch_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), t, w,
                       PageWrite{page_table, page_stride, page_rows, page_cols},
                       StageFetch{static_cast<const char*>(fetch_ring), static_cast<char*>(fetch_dst), fetch_counter, fetch_slot_bytes, fetch_ring_n});
    return check_launch("window patch launch");
}

/* The step's host-written words fetched from a ring of pinned host slots by a kernel (window.h, StageFetch) -- what a
 * hipMemcpyAsync in front of the step's graph did, without the idle queue around a stand-alone copy.  Slot k (k = *counter modulo
 * ring_n, then *counter += 1) = {uint32 used bytes, 12 bytes of padding, payload}: the payload's first `used` bytes go to dst. */
int deft_stage_fetch(const void* ring, int slot_bytes, int ring_n, void* dst, int32_t* counter, void* stream) {
    if (!ring || !dst || !counter || slot_bytes < 32 || slot_bytes % 16 || ring_n < 1) {
        set_error("deft_stage_fetch: bad arguments");
        return DEFT_EINVAL;
    }
    hipLaunchKernelGGL(stage_fetch_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream),
                       StageFetch{static_cast<const char*>(ring), static_cast<char*>(dst), counter, slot_bytes, ring_n});
    return check_launch("stage fetch launch");
}

/* The default form of the same hand-over: slot `slot` of the ring copied to dst by ONE hipMemcpyAsync on `stream`, `used` read from
 * the slot's header by the host.  (What deft_amd.DecodeSession did through two tensor slices and Tensor.copy_: ~10 us of host time.) */
int deft_stage_copy(const void* ring, int slot_bytes, int slot, void* dst, size_t dst_bytes, void* stream) {
    if (!ring || !dst || slot_bytes < 32 || slot_bytes % 16 || slot < 0) {
        set_error("deft_stage_copy: bad arguments");
        return DEFT_EINVAL;
    }
    const char* src = static_cast<const char*>(ring) + static_cast<size_t>(slot) * slot_bytes;
    uint32_t used;
    memcpy(&used, src, 4);
    if (used > dst_bytes || used + 16 > static_cast<size_t>(slot_bytes)) {
        set_error("deft_stage_copy: the slot's header names more bytes than the slot or the destination holds");
        return DEFT_EINVAL;
    }
    if (used == 0) return DEFT_OK;
    hipError_t e = hipMemcpyAsync(dst, src + 16, used, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) {
        set_error("deft_stage_copy: %s", hipGetErrorString(e));
        return DEFT_EHIP;
    }
    return DEFT_OK;
}

}  // extern "C"
