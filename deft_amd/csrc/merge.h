// Stage 2: log-sum-exp merge of the partial rows of one (query, head) pair by ONE wave
// (the reference: tree_attention.py:296-546, two kernels with fp32 / fp16 atomics and a host-side div_).
//
// Deterministic gather: the wave lists the partial rows of its query in ascending row order (scan of row_q, ordered
// ballots), takes the TRUE maximum of their log-sum-exps, accumulates in fp32 in list order and rounds to fp16 once.
// Used by the stand-alone merge kernel (merge_kernel, deft_kernels.hip) and by the merge waves at the end of the
// single-launch decode kernel (stage1_np.h) -- same code, same bits.
//
// Included by deft_kernels.hip after plan_records.h.
#pragma once

namespace deft {

// Running state of one pair's merge; lists longer than the wave's LDS area are merged piecewise (online rescale).
struct MergeState {
    float m = -INFINITY;  // running maximum of the log-sum-exps seen
    float L = 0.f;        // sum of exp(lse - m)
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
};

// Rows [from, to) of row_q that belong to query q, ascending, appended to list[0 .. cap); returns how many were found
// (only the first `cap` are stored).  Eight independent loads per lane in flight: one L2 round trip per 512 rows.
__device__ inline int scan_rows_wave(const int32_t* row_q, int64_t from, int64_t to, int q, int* list, int cap, int lane) {
    int n = 0;
    for (int64_t base = from; base < to; base += 512) {
        int val[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t i = base + 64 * u + lane;
            val[u] = (i < to) ? row_q[i] : -1;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool hit = val[u] == q;
            const unsigned long long mask = __ballot(hit);
            const int pos = n + __popcll(mask & ((1ull << lane) - 1ull));
            if (hit && pos < cap) list[pos] = (int)(base + 64 * u + lane);
            n += __popcll(mask);
        }
    }
    return n;
}

// Fold rows list[0 .. n) of NH heads into their states -- one query, so ONE row list; the heads only differ in the
// base of their partial rows, and the loads of all NH heads are requested together (one memory round trip for up to
// NH * NU * RPS rows).  CP = cache policy of the loads (0: ordinary; CP_SYS: the rows were written by other workgroups
// of this launch).  po / lse: buffer resources over each head's partial rows.
template <int D, int CP, int NH>
__device__ inline void merge_accumulate(MergeState (&st)[NH], const __amdgpu_buffer_rsrc_t (&po)[NH],
                                        const __amdgpu_buffer_rsrc_t (&lse)[NH], const int* list, int n, int lane) {
    constexpr int LPR = D / 4;     // lanes per row (16 bytes each)
    constexpr int RPS = 64 / LPR;  // rows per wave-wide load
    // Loads in flight per lane and head.  4 (= 8 rows of head_dim 128 per round trip) measured best: 8 cost EVERY layer
    // 1.2-2.2 us (north-star 36.2 -> 34.9 us per layer, Medusa-64 16.2 -> 15.0, ToT-50 23.6 -> 22.3, one 8k x 8 tree 18.8 ->
    // 16.6; 198 -> 70 VGPRs: the zero-fills, predicated loads and weighted adds of the unused slots are per-batch fixed cost
    // for the usual 3-11 rows), 16 cost 3-8 us, 3 ties, 2 and 1 lose 0.2-1.5 us on the long lists
    constexpr int NU = 4;
    const int sub = lane / LPR, col = lane % LPR;
    for (int cbase = 0; cbase < n; cbase += 64) {
        const int cn = n - cbase < 64 ? n - cbase : 64;
        float x[NH];
        const int my_row = lane < cn ? list[cbase + lane] : 0;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            x[h] = -INFINITY;
            if (lane < cn) x[h] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lse[h], my_row * 4, 0, CP));
        }
        uintx4 v[NH][NU];
        auto load_batch = [&](int b) {
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int j = b * NU * RPS + u * RPS + sub;
                const int off = j < cn ? list[cbase + j] * (D * 4) + col * 16 : 0;
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    v[h][u] = uintx4{0u, 0u, 0u, 0u};
                    if (j < cn) v[h][u] = __builtin_amdgcn_raw_buffer_load_b128(po[h], off, 0, CP);
                }
            }
        };
        load_batch(0);  // requested together with the log-sum-exps
        float wl[NH];
        bool any = false;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            float cm = x[h];
#pragma unroll
            for (int sft = 32; sft > 0; sft >>= 1) cm = fmaxf(cm, __shfl_xor(cm, sft));
            wl[h] = 0.f;
            if (cm == -INFINITY) continue;  // wave-uniform: nothing live for this head in this chunk
            any = true;
            const float m_new = fmaxf(st[h].m, cm);
            const float scale = (st[h].m == -INFINITY) ? 0.f : __expf(st[h].m - m_new);
            wl[h] = (x[h] == -INFINITY) ? 0.f : __expf(x[h] - m_new);
            float ws = wl[h];
#pragma unroll
            for (int sft = 32; sft > 0; sft >>= 1) ws += __shfl_xor(ws, sft);
            st[h].L = st[h].L * scale + ws;
            st[h].acc *= scale;
            st[h].m = m_new;
        }
        if (!any) continue;
        for (int b = 0; b * NU * RPS < cn; ++b) {
            if (b > 0) load_batch(b);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int j = b * NU * RPS + u * RPS + sub;
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    const float wj = __shfl(wl[h], j & 63);
                    if (j < cn && wj > 0.f) st[h].acc += __builtin_bit_cast(floatx4, v[h][u]) * wj;
                }
            }
        }
    }
}

// Sum the row groups of the wave and write the pair's output row (fp16, one rounding).
template <int D>
__device__ inline void merge_finish(const MergeState& st, _Float16* dst, int lane) {
    constexpr int LPR = D / 4;
    floatx4 a = st.acc;
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] += __shfl_xor(a[e], off);
    if (lane < LPR) {
        const float inv = st.L > 0.f ? 1.f / st.L : 0.f;
        half2v lo = {(_Float16)(a[0] * inv), (_Float16)(a[1] * inv)};
        half2v hi = {(_Float16)(a[2] * inv), (_Float16)(a[3] * inv)};
        *reinterpret_cast<half2v*>(dst + 4 * lane) = lo;  // output rows are only 4-byte aligned by contract
        *reinterpret_cast<half2v*>(dst + 4 * lane + 2) = hi;
    }
}

// (Round 5 built a specialised form of the above for the usual case -- one head, head_dim 128, up to eight rows: wave-uniform row
//  ids, every lane reading all log-sum-exps itself so that maximum / weights / sum need no shuffles, one v_permlane32_swap per value
//  -- bit-identical by construction and 0.4-0.8 us SLOWER per launch: profiles/r5_merge_small_lists_negative.txt.  And NH = 2 / 4
//  heads of one XCD per wave, a quarter of the workgroups: +1.5 / +4.6 us, r5_merge_heads_per_wave_negative.txt.)

// One query, NH heads (hq0, hq0 + hq_step, ...; those >= Hq are skipped), start to finish: rows of the query from
// row_q (list area of `cap` ints in LDS, longer lists in further passes), merge, store.  `have` >= 0:
// list[0 .. min(have, cap)) already holds the scan.
template <int D, int CP, int NH>
__device__ inline void merge_heads_wave(const float* partial_o, const float* partial_lse, const int32_t* row_q, int64_t rows,
                                        int q, int hq0, int hq_step, int Hq, int* list, int cap, int have, _Float16* out_q,
                                        int64_t o_sh, int lane) {
    __amdgpu_buffer_rsrc_t po[NH], ls[NH];
    MergeState st[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const int hq = hq0 + h * hq_step < Hq ? hq0 + h * hq_step : hq0;  // (a skipped head re-reads the first one)
        po[h] = make_rsrc(partial_o + (int64_t)hq * rows * D);
        ls[h] = make_rsrc(partial_lse + (int64_t)hq * rows);
    }
    if (have >= 0 && have <= cap) {  // the usual case: the whole list was scanned ahead of time
        merge_accumulate<D, CP, NH>(st, po, ls, list, have, lane);
    } else {
        // a query with more rows than the list area: window by window over row_q (each window holds at most cap rows)
        for (int64_t from = 0; from < rows; from += cap) {
            const int64_t to = from + cap < rows ? from + cap : rows;
            const int n = scan_rows_wave(row_q, from, to, q, list, cap, lane);
            __builtin_amdgcn_wave_barrier();
            merge_accumulate<D, CP, NH>(st, po, ls, list, n, lane);
            __builtin_amdgcn_wave_barrier();
        }
    }
#pragma unroll
    for (int h = 0; h < NH; ++h)
        if (hq0 + h * hq_step < Hq) merge_finish<D>(st[h], out_q + (int64_t)(hq0 + h * hq_step) * o_sh, lane);
}

}  // namespace deft
