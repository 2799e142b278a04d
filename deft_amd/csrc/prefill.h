// Causal prefill attention over the prompt (TTFT): the caller of the decode path on its time axis.
//
// Included by deft_kernels.hip (needs its typedefs, dma16 / wait_vm / lds_barrier from stage1_stream.h).
//
// Replaces context_attention_fwd (DeFT/deft/layers/attention/context_flashattention_nopad.py:130-195; kernel :12-127)
// behind DeFTAttention.prefill_forward_triton (deft_attention.py:50-70): a batch of sequences packed without padding,
// q[T, Hq, D], k / v[T, Hkv, D], token i of sequence b attends to tokens 0..i of b.  Compute-bound (MFMA), unlike the
// decode path; arithmetic as there: fp16 operands, fp32 accumulation, scale 1/sqrt(D) after the dot, fp32 online
// softmax, P rounded to fp16 for the PV product with row sums over the ROUNDED values, one fp16 rounding at the end.
//
//   * one workgroup = 8 waves = 256 consecutive queries of one (sequence, query head); wave w owns queries 32w..32w+31;
//   * K / V tiles of 128 keys are staged in LDS by LDS-DMA, double buffered: while a tile is consumed the next one
//     lands (every wave stages 16 of its 128 keys); one barrier per tile;
//   * S^T = K Q^T on v_mfma_f32_32x32x16_f16 with the keys on M: a lane holds 16 key scores of ONE query, so the
//     softmax is in-lane (+ one half-wave swap), P^T stays in registers and is the B operand of O^T += V^T P^T, V^T
//     fragments read with ds_read_b64_tr_b16 -- the decode kernel's inner loop (stage1_np.h), four 32-key blocks per tile;
//   * one online-softmax step per 128-key tile: 32 QK^T MFMAs on four independent accumulators, one max / rescale,
//     32 PV MFMAs;
//   * causal structure: query block m needs key tiles 0 .. 2m+1; only the last two touch the diagonal and are masked,
//     a wave skips tiles that lie entirely above its queries; workgroups are launched longest first over the whole grid;
//   * head_dim 128 and 64 (the reference takes 16 / 32 / 64 / 128, context_flashattention_nopad.py:134; 32 and 16 go through
//     prefill_small_kernel below): a row is D / 8 16-byte chunks, O^T has D / 32 column blocks, QK^T D / 16 k-steps; the LDS
//     swizzles keep their form with the chunk count as modulus.
#pragma once

namespace deft {

struct PrefillParams {
    const _Float16* q;
    const _Float16* k;
    const _Float16* v;
    _Float16* o;
    int64_t q_st, q_sh, k_st, k_sh, v_st, v_sh, o_st, o_sh;  // elements
    const int32_t* b_start_loc;
    const int32_t* b_seq_len;
    int G;  // query heads per KV head
    float scale_log2e;
    int nblk;  // query blocks per sequence in the grid: ceil(max_input_len / 256)
    int Hq, batch;
    unsigned long long* dbg;  // experiments build: per-workgroup wall-clock stamps [workgroup][8], or null
};

template <int D>
struct PrefillSmem {
    static constexpr int STAGE = TILE * D * 2;  // one K (or V) tile of 128 keys
    static constexpr int K_OFF = 0;             // two stages
    static constexpr int V_OFF = 2 * STAGE;     // two stages
    static constexpr int BYTES = 4 * STAGE;     // 128 KB
    static_assert(BYTES <= 160 * 1024, "LDS budget");
};

template <int D>
__global__ __launch_bounds__(512, 1) void prefill_kernel(PrefillParams p) {
    constexpr int KS = D / 16;   // k-steps of QK^T
    constexpr int CH = D / 8;    // 16-byte chunks per row
    constexpr int NB = D / 32;   // 32-wide column blocks of O^T
    constexpr int RPI = 64 / CH; // rows one 64-lane request instruction covers (4 at head_dim 128, 8 at 64)
    constexpr int NREQ = TILE / (8 * RPI);  // request instructions per wave for a K (or V) tile: 4 / 2
    constexpr int QB = 256;  // queries per workgroup
    static_assert(D == 128 || D == 64, "prefill_kernel is instantiated for head_dim 128 and 64");
    using SM = PrefillSmem<D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63;
    const int c = l & 31;
    const int h = l >> 5;
    // One linear grid, query block major, LONGEST BLOCKS FIRST ACROSS ALL HEADS AND SEQUENCES: the dispatcher hands
    // workgroups out in index order, so this is longest-processing-time-first list scheduling (with the block index
    // fastest and the head slowest -- the first form -- a 4k-token prompt ran 64 tile times on its busiest CU for a mean of
    // 34: the CU that finished a head's shortest block was handed the next head's LONGEST one).  Within a block row the
    // q heads of one KV head land on one XCD (workgroup index mod 8), whose L2 then holds 4 KV heads, not all of them.
    const int per_row = p.Hq * p.batch;
    const int L = (int)blockIdx.x;
    const int m = p.nblk - 1 - L / per_row;
    const int rem = L - (L / per_row) * per_row;
    const int b = rem / p.Hq;
    const int hi = rem - b * p.Hq;
    const int Hkv = p.Hq / p.G;
    const int head = (hi % Hkv) * p.G + hi / Hkv;
    const int len = p.b_seq_len[b];
    const int64_t start = p.b_start_loc[b];
    if (m * QB >= len) return;
#ifdef DEFT_EXPERIMENTS
    unsigned long long t_start = 0, t_loop = 0, t_epi = 0;
    if (p.dbg) t_start = wall_clock64();
#endif
    const int kvh = head / p.G;

    // ---- lane constants (LDS layouts of stage1_np.h: K chunks XOR-ed by key & (CH - 1), V chunks by 4*(key & (NB - 1))) ------
    const int dpos = l & (CH - 1), dkey = l / CH;
    const int tg = l >> 4, tx = l & 15;
    int vtr_col_b[NB];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) vtr_col_b[blk] = (4 * (blk ^ ((tx >> 2) & (NB - 1))) + 2 * (tg & 1) + ((tx & 3) >> 1)) * 16;
    const int vtr_row_b = (4 * (tg >> 1) + (tx >> 2)) * D * 2 + (tx & 1) * 8;
    const int krow_b = c * D * 2;
    const int kcol_b = ((h ^ c) & (CH - 1)) * 16;
    int kfrag_b[KS], vfrag_b[NB];  // per-lane fragment bases inside a stage
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kfrag_b[ks] = krow_b + (kcol_b ^ (32 * ks));
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) vfrag_b[blk] = vtr_row_b + vtr_col_b[blk];

    const char* kbase = reinterpret_cast<const char*>(p.k + start * p.k_st + (int64_t)kvh * p.k_sh);
    const char* vbase = reinterpret_cast<const char*>(p.v + start * p.v_st + (int64_t)kvh * p.v_sh);
    // (round 4, tools/prefill_phases.py: the CU's request path takes the workgroup's 64 request instructions one wave at a time and
    //  the older half of the workgroup wins the arbitration -- waves 0-3 are through their eight in 0.43 us, waves 4-7 in 1.13 us,
    //  and the older half then waits 1.24 us per tile at the barrier.  Dealing the older half 5 .. 8 of every 8 pieces instead of 4
    //  moves that wait to the other half and changes nothing: 979 / 969 / 960 / 966 / 967 TFLOP/s at 16k tokens for 4 .. 8.)
    // (spreading a tile's eight request instructions over the k-steps' MFMAs instead -- one behind every four, or K among QK^T and V
    //  among PV -- measured 0.98x / 1.003x: the time moves from one phase into the other, profiles/r4_prefill64_negative.txt)
    auto issue_tile = [&](int t, int stg) {  // NREQ K + NREQ V instructions per wave: keys 16w + RPI i + dkey of tile t
#pragma unroll
        for (int i = 0; i < NREQ; ++i) {
            const int key = 16 * w + RPI * i + dkey;
            int tok = TILE * t + key;
            tok = tok < len ? tok : len - 1;  // padding aliases the last token; masked by the causal test
            const int kc = (dpos ^ (key & (CH - 1))) * 16;
            const int vc = (dpos ^ (4 * (key & (NB - 1)))) * 16;
            dma16(kbase + (int64_t)tok * p.k_st * 2 + kc, SM::K_OFF + (uint32_t)stg * SM::STAGE + (uint32_t)(16 * w + RPI * i) * (uint32_t)(D * 2));
            dma16(vbase + (int64_t)tok * p.v_st * 2 + vc, SM::V_OFF + (uint32_t)stg * SM::STAGE + (uint32_t)(16 * w + RPI * i) * (uint32_t)(D * 2));
        }
    };

    const int ntiles = min(2 * m + 2, (len + TILE - 1) / TILE);
    issue_tile(0, 0);

    // ---- this lane's query and its Q fragments (B operand: 8 halves at d = 16 ks + 8 h) ----------------------
    const int qi = m * QB + 32 * w + c;  // query index inside the sequence
    const int qrow = qi < len ? qi : len - 1;
    half8 qf[KS];
    {
        const _Float16* qp = p.q + (start + qrow) * p.q_st + (int64_t)head * p.q_sh + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const half8*>(qp + 16 * ks);
    }
    // Q fragments and tile 0.  The empty asm statements make the compiler itself wait for the Q loads HERE: otherwise its
    // counter model carries "8 loads outstanding" into the tile loop and puts s_waitcnt vmcnt(7) .. vmcnt(0) in front of
    // the first use of qf[0] .. qf[7] in EVERY iteration, where the only loads in flight are the next tile's DMA
    // (inline asm, invisible to it).  Measured neutral (a tile lasts far longer than the DMA), kept for the cleaner loop.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));

    float m_run = -INFINITY, l_run = 0.f;
    floatx16 o[NB];
#pragma unroll
    for (int bk = 0; bk < NB; ++bk)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[bk][r] = 0.f;
    const int q_lo = m * QB + 32 * w;  // first query of this wave

#ifdef DEFT_EXPERIMENTS
    if (p.dbg) t_loop = wall_clock64();
    // per-phase wall-clock sums of this wave over its tiles (tools/prefill_phases.py): 0 wait + barrier | 1 next tile's requests |
    // 2 QK^T | 3 softmax | 4 PV
    unsigned long long ph[5] = {0, 0, 0, 0, 0}, ph_t = 0;
    int ph_tiles = 0;
#define PF_PHASE(k)                                      \
    if (p.dbg) {                                         \
        const unsigned long long now_ = wall_clock64(); \
        ph[k] += now_ - ph_t;                            \
        ph_t = now_;                                     \
    }
#else
#define PF_PHASE(k)
#endif
    for (int t = 0; t < ntiles; ++t) {
        const int stg = t & 1;
#ifdef DEFT_EXPERIMENTS
        if (p.dbg) ph_t = wall_clock64();
#endif
        wait_vm<0>();   // tile t landed (this wave's part)
        lds_barrier();  // ... everyone's part; and every wave is done with tile t-1's stage
        PF_PHASE(0);
        if (t + 1 < ntiles) issue_tile(t + 1, stg ^ 1);
        PF_PHASE(1);
        const int key0 = TILE * t;
        if (key0 > q_lo + 31) continue;  // the whole tile lies above this wave's queries (wave-uniform)
        const bool diag = key0 + TILE - 1 > q_lo;  // some key of the tile is beyond some query of the wave
        // ---- S^T for all 128 keys of the tile: four independent accumulator chains (32 keys each) -----------
        floatx16 acc[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[kb][r] = 0.f;
        // (fragment addresses = a per-lane base kept in a register + the stage + an immediate: one add per k-step)
        const int kstage = SM::K_OFF + stg * SM::STAGE;
        // K fragments double-buffered by k-step: the four reads of k-step ks + 1 are issued BEFORE the four MFMAs of ks (left to
        // itself the compiler reads two fragments, waits, issues two MFMAs -- every pair of MFMAs behind an LDS round trip)
        half8 af[2][4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) af[0][kb] = *reinterpret_cast<const half8*>(smem + (kfrag_b[0] + kstage) + 32 * kb * D * 2);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) {
                const char* kp = smem + (kfrag_b[ks + 1] + kstage);
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) af[(ks + 1) & 1][kb] = *reinterpret_cast<const half8*>(kp + 32 * kb * D * 2);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) acc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks & 1][kb], qf[ks], acc[kb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        PF_PHASE(2);
#ifdef DEFT_EXPERIMENTS
        ++ph_tiles;
#endif
        // ---- one online-softmax step per tile (one rescale of O per 128 keys) ----------------------------------
        // (VALU-bound: per score one max, one fma feeding exp2, one conversion and half a dot2 -- the scale rides in the
        //  fma; the row sum is a chain of fp32 adds, see below)
        float mx = -INFINITY;
        if (diag) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + 32 * kb + 8 * (r >> 2) + 4 * h + (r & 3);
                    acc[kb][r] = key <= qi ? acc[kb][r] : -INFINITY;  // causal; keys >= len are > every valid query
                    mx = fmaxf(mx, acc[kb][r]);
                }
        } else {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, acc[kb][r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32)) * p.scale_log2e;  // (scaling is monotonic: the max of the scaled scores)
        const float m_new = fmaxf(m_run, mx);
        const float msafe = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - msafe);
        half8 pb[4][2];
        float sum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                // Row sums in fp32 over the UNROUNDED probabilities, as ONE dependent chain of single adds (the reference sums the
                // unrounded p too, context_flashattention_nopad.py:96-103).  Round 1 summed the rounded values with packed
                // fp16 dots (v_dot2c_f32_f16): fewer instructions, but that opcode competes with the other wave's MFMAs for
                // the matrix pipe -- plain adds: Llama-2-7B 779 -> 820 TFLOP/s at 4k tokens, 962 -> 985 at 16k (two side-by-side
                // chains would be packed into v_pk_add_f32, which is as bad)
                const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[kb][r], p.scale_log2e, -msafe));
                const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[kb][r + 1], p.scale_log2e, -msafe));
                pb[kb][r >> 3][r & 7] = (_Float16)e0;
                pb[kb][r >> 3][(r & 7) + 1] = (_Float16)e1;
                sum += e0;
                sum += e1;
            }
        sum += __shfl_xor(sum, 32);
        l_run = l_run * alpha + sum;
        m_run = m_new;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0ull) {
#pragma unroll
            for (int bk = 0; bk < NB; ++bk)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[bk][r] *= alpha;  // (packed by the compiler; 64 single multiplies measured the same)
        }
        PF_PHASE(3);
        // ---- O^T += V^T P^T: 32 MFMAs on four independent accumulators -------------------------------------------
        const int vstage = SM::V_OFF + stg * SM::STAGE;
        int vfrag[NB];
#pragma unroll
        for (int bk = 0; bk < NB; ++bk) vfrag[bk] = vfrag_b[bk] + vstage;
        // V^T fragments double-buffered by (key block, k-step) group: the transpose reads of group g + 1 go out before the
        // MFMAs of group g
        typedef __attribute__((address_space(3))) short4v* lds_s4;
        union VFrag {
            short4v s4[2];
            half8 h8;
        };
        VFrag vf[2][NB];
        auto load_group = [&](int g, VFrag (&dst)[NB]) {
            const int kb = g >> 1, tt = g & 1;
#pragma unroll
            for (int bk = 0; bk < NB; ++bk) {
                const int vb = vfrag[bk] + (32 * kb * D * 2 + (16 * tt) * D * 2);
                dst[bk].s4[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(uintptr_t)(vb));
                dst[bk].s4[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(uintptr_t)(vb + 8 * D * 2));
            }
        };
        load_group(0, vf[0]);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (g + 1 < 8) load_group(g + 1, vf[(g + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int bk = 0; bk < NB; ++bk)
                o[bk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[g & 1][bk].h8, pb[g >> 1][g & 1], o[bk], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        PF_PHASE(4);
    }
    wait_vm<0>();
#ifdef DEFT_EXPERIMENTS
    if (p.dbg) t_epi = wall_clock64();
    if (p.dbg && l == 0 && L < 1024) {  // [8192 x 8 workgroup stamps][1024 workgroups][8 waves][8]
        unsigned long long* d2 = p.dbg + (int64_t)8192 * 8 + ((int64_t)L * 8 + w) * 8;
#pragma unroll
        for (int k2 = 0; k2 < 5; ++k2) d2[k2] = ph[k2];
        d2[5] = (unsigned long long)ph_tiles;
        d2[6] = (unsigned long long)ntiles;
    }
#endif
    // ---- normalise and store: lane (c, h) holds d = 32 bk + 8 j + 4 h + (0..3) of query c ---------------------
    if (qi < len) {
        const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
        _Float16* op = p.o + (start + qi) * p.o_st + (int64_t)head * p.o_sh + 4 * h;
#pragma unroll
        for (int bk = 0; bk < NB; ++bk)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                half4 v4 = {(_Float16)(o[bk][4 * j] * inv), (_Float16)(o[bk][4 * j + 1] * inv),
                            (_Float16)(o[bk][4 * j + 2] * inv), (_Float16)(o[bk][4 * j + 3] * inv)};
                *reinterpret_cast<half4*>(op + 32 * bk + 8 * j) = v4;
            }
    }
#ifdef DEFT_EXPERIMENTS
    if (p.dbg && tid == 0 && L < 8192) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* d = p.dbg + (int64_t)L * 8;
        d[0] = t_start;
        d[1] = t_loop;
        d[2] = t_epi;
        d[3] = wall_clock64();
        d[4] = (unsigned long long)ntiles;
        d[5] = (unsigned long long)xcc;
    }
#endif
}
#undef PF_PHASE

// head_dim 32 and 16 (context_flashattention_nopad.py:134 takes them; no model the reference ships has them): one wave per
// (query token, query head), the lanes stride over the keys 0 .. i with a private online softmax in fp32 and merge at the end.
// Correctness path, not a tuned one.  Grid: (ceil(max_input_len / 4) * batch, Hq), 256 threads.
template <int D>
__global__ __launch_bounds__(256) void prefill_small_kernel(PrefillParams p) {
    static_assert(D == 32 || D == 16, "small head dimensions");
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int per_seq = (p.nblk * 256 + 3) / 4;  // (nblk * 256 >= max_input_len)
    const int b = (int)blockIdx.x / per_seq;
    const int qi = ((int)blockIdx.x - b * per_seq) * 4 + w;
    const int head = (int)blockIdx.y;
    if (b >= p.batch) return;  // (before the load: robust to a grid larger than per_seq * batch)
    const int len = p.b_seq_len[b];
    if (qi >= len) return;
    const int64_t start = p.b_start_loc[b];
    const int kvh = head / p.G;
    float q[D];
    const _Float16* qp = p.q + (start + qi) * p.q_st + (int64_t)head * p.q_sh;
#pragma unroll
    for (int d = 0; d < D; ++d) q[d] = (float)qp[d];
    float m = -INFINITY, s = 0.f, acc[D];
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] = 0.f;
    for (int j = l; j <= qi; j += 64) {
        const _Float16* kp = p.k + (start + j) * p.k_st + (int64_t)kvh * p.k_sh;
        const _Float16* vp = p.v + (start + j) * p.v_st + (int64_t)kvh * p.v_sh;
        float dot = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) dot += q[d] * (float)kp[d];
        const float x = dot * p.scale_log2e;
        const float mn = fmaxf(m, x);
        const float a = __builtin_amdgcn_exp2f(m - mn), e = __builtin_amdgcn_exp2f(x - mn);  // (exp2(-inf) = 0 the first time)
        s = s * a + e;
#pragma unroll
        for (int d = 0; d < D; ++d) acc[d] = acc[d] * a + e * (float)vp[d];
        m = mn;
    }
    // merge the 64 lanes' states (lanes beyond the keys hold m = -inf, s = 0)
    for (int off = 32; off > 0; off >>= 1) {
        const float mo = __shfl_xor(m, off, 64), so = __shfl_xor(s, off, 64);
        const float mn = fmaxf(m, mo);
        const float a = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - mn), bfac = (mo == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mo - mn);
        s = s * a + so * bfac;
#pragma unroll
        for (int d = 0; d < D; ++d) acc[d] = acc[d] * a + __shfl_xor(acc[d], off, 64) * bfac;
        m = mn;
    }
    if (l == 0) {
        const float inv = s > 0.f ? 1.f / s : 0.f;
        _Float16* op = p.o + (start + qi) * p.o_st + (int64_t)head * p.o_sh;
#pragma unroll
        for (int d = 0; d < D; ++d) op[d] = (_Float16)(acc[d] * inv);
    }
}

}  // namespace deft
