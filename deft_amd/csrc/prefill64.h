// Causal prefill attention, one wave per SIMD: the form of prefill.h in which a wave's own VALU work hides under its own MFMAs.
//
// Included by deft_kernels.hip after prefill.h (PrefillParams, the LDS layouts and the lane constants are prefill.h's).
//
// Why a second form (round 4; tools/probes/issue_cost.hip, profiles/r4_prefill64.txt).  With ONE wave on a SIMD a
// v_mfma_f32_32x32x16_f16 occupies the matrix pipe for 32 cycles and the wave can issue about five other instructions in its shadow
// for free (MFMA + 4 v_fma: 32.5 cycles; + 6: 35.6; an LDS fragment read counts as one, a v_exp_f32 as one and a half); everything
// beyond that is serial.  prefill.h's loop needs ~8.5 other instructions per MFMA -- every MFMA reads its own 1-KB A fragment from
// the LDS, every score costs max / fma / exp2 / conversion / add -- and its two waves per SIMD do not hide each other's (five
// arrangements measured, all ~68 cycles per MFMA).  So:
//   * 4 waves, 512 registers each; every wave owns TWO 32-query blocks A and B, and every K / V^T fragment read from the LDS feeds
//     two MFMAs, A's and B's (0.67 reads per MFMA instead of 1.5);
//   * the row sums come out of the matrix pipe: one more MFMA per 16-key step with an all-ones A operand accumulates
//     L[d][q] = sum_k P[k][q] next to O^T (and is rescaled with it), instead of 64 v_add per step;
//   * a step = sub-tile u's QK^T for both blocks (HALF Q: 32 MFMAs, the sub-tile's eight request instructions, nothing else), then
//     sub-tile u-1's PV for both blocks (HALF P: 40 MFMAs) woven with the softmax of sub-tile u (~210 VALU instructions): written out
//     piece by piece -- one MFMA, the fragment reads three fragments ahead, ~5 softmax instructions, a scheduling fence -- because
//     left to itself the compiler hoists every fragment read to the top (261 spilled registers) and runs the softmax as one block;
//   * P(u) overwrites P(u-1) in place: the conversions of the scores of 16-key step g are placed behind the PV MFMAs of step g.
//
// The MFMAs are inline asm so that the accumulators live where they are used.  O^T, L and the Q fragments (MFMA operands only) are
// accumulation registers this file OWNS -- a[0:63] / a[64:127] O^T of block A / B, a[128:159] / a[160:191] their Q fragments,
// a[192:207] / a[208:223] L -- never C++ values (an "a" operand that is an ordinary value elsewhere is kept in ordinary registers
// and copied around every MFMA), named and listed as clobbered by every asm statement of the loop so that the compiler keeps its
// own values out of them (checked in the ISA: it uses none below a224).  S (read by the softmax) is in ordinary registers: with 512
// registers per wave the compiler's own MFMAs all write accumulation registers and every score would cost a v_accvgpr_read.
// What an asm MFMA hides from the compiler's hazard recogniser is kept apart by the structure: S is written in half Q and read
// in half P (a no-op pad between them); O / L are read only by the rare rescale and the final store, each behind its own pad.
//
// Arithmetic as in prefill.h (fp16 operands, fp32 accumulation, scale inside the exp2 argument, one fp16 rounding at the end) with
// three differences: the online-softmax step is 64 keys (128 there); the row sums are over the ROUNDED probabilities (the weights
// sum to 1 exactly, as in the decode kernel); and the reference maximum of a row only moves when some row of the wave outgrew its
// own by more than 2^8 (a wave-uniform choice): until then the probabilities are taken against the old reference (<= 256, exact in
// fp16's range) and O needs no rescale -- with a moving maximum a wave of 64 rows rescales in almost every step of a 4k-token prompt.
//
//   * a workgroup = 4 waves = 256 consecutive queries of one (sequence, query head); wave w owns queries 64w .. 64w+63;
//   * K / V arrive in sub-tiles of 64 keys by LDS-DMA (scalar base + 32-bit lane offsets) into a ring of four stages (128 KB), two
//     sub-tiles ahead; one barrier per step;
//   * causal structure: wave w of query block m folds sub-tiles 0 .. 4m+w; only the last one touches the diagonal.
#pragma once

#include <type_traits>
#include <utility>

#define PF64_A16(b) "a" #b "0", "a" #b "1", "a" #b "2", "a" #b "3", "a" #b "4", "a" #b "5", "a" #b "6", "a" #b "7", "a" #b "8", "a" #b "9"
// a[0:223]
#define PF64_ACC_ALL                                                                                                              \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", PF64_A16(1), PF64_A16(2), PF64_A16(3), PF64_A16(4), PF64_A16(5),   \
        PF64_A16(6), PF64_A16(7), PF64_A16(8), PF64_A16(9), PF64_A16(10), PF64_A16(11), PF64_A16(12), PF64_A16(13), PF64_A16(14), \
        PF64_A16(15), PF64_A16(16), PF64_A16(17), PF64_A16(18), PF64_A16(19), PF64_A16(20), PF64_A16(21), "a220", "a221", "a222",  \
        "a223"

#ifndef PF64_CUT
#define PF64_CUT 72  // experiments: the full step stops after this many pieces (where does a step's time go)
#endif
#ifndef PF64_ABL
#define PF64_ABL 0  // experiments: 1 no softmax instructions, 2 no fragment reads, 4 no MFMAs, 8 no requests inside half Q, 16 no fences
#endif

namespace deft {
namespace pf64 {

constexpr int ACC_O = 0, ACC_Q = 128, ACC_L = 192, ACC_END = 224;

// S^T = K Q^T: result in ordinary registers, Q fragment a[QLO : QLO+3]
template <int QLO>
__device__ __forceinline__ void mfma_s0(floatx16& acc, const half8& a) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[%2:%3], 0" : "=&v"(acc) : "v"(a), "n"(QLO), "n"(QLO + 3) : PF64_ACC_ALL);
}
template <int QLO>
__device__ __forceinline__ void mfma_s(floatx16& acc, const half8& a) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[%2:%3], %0" : "+v"(acc) : "v"(a), "n"(QLO), "n"(QLO + 3) : PF64_ACC_ALL);
}
// a[LO : LO+15] += A B
template <int LO>
__device__ __forceinline__ void mfma_acc(const half8& a, const half8& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(a), "v"(b), "n"(LO), "n"(LO + 15) : PF64_ACC_ALL);
}
// four dwords of a Q fragment into a[LO : LO+3]
template <int LO>
__device__ __forceinline__ void acc_put4(const half8& q) {
    union {
        half8 h8;
        uint32_t u[4];
    } x;
    x.h8 = q;
    asm volatile("v_accvgpr_write_b32 a[%4], %0\n\tv_accvgpr_write_b32 a[%4 + 1], %1\n\tv_accvgpr_write_b32 a[%4 + 2], %2\n\t"
                 "v_accvgpr_write_b32 a[%4 + 3], %3" ::"v"(x.u[0]), "v"(x.u[1]), "v"(x.u[2]), "v"(x.u[3]), "n"(LO) : PF64_ACC_ALL);
}
// a[LO : LO+N-1] = 0
template <int LO, int N>
__device__ __forceinline__ void acc_zero() {
    asm volatile(".set .Lpf64_i, %0\n\t.rept %1\n\tv_accvgpr_write_b32 a[.Lpf64_i], 0\n\t.set .Lpf64_i, .Lpf64_i + 1\n\t.endr" ::"n"(LO), "n"(N) : PF64_ACC_ALL);
}
// a[LO : LO+N-1] *= alpha (N even).  Rare: the reference maximum of some row of the wave moved.  (pads: the last MFMAs into these
// registers may have been issued just before, and an MFMA may read them just after -- the compiler knows of neither)
template <int LO, int N>
__device__ __forceinline__ void acc_scale(float alpha) {
    float t0, t1;
    asm volatile("s_nop 15\n\ts_nop 15\n\t.set .Lpf64_i, %3\n\t.rept %4\n\tv_accvgpr_read_b32 %0, a[.Lpf64_i]\n\tv_accvgpr_read_b32 %1, a[.Lpf64_i + 1]\n\t"
                 "v_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %2\n\tv_accvgpr_write_b32 a[.Lpf64_i], %0\n\t"
                 "v_accvgpr_write_b32 a[.Lpf64_i + 1], %1\n\t.set .Lpf64_i, .Lpf64_i + 2\n\t.endr\n\ts_nop 7"
                 : "=&v"(t0), "=&v"(t1) : "v"(alpha), "n"(LO), "n"(N / 2) : PF64_ACC_ALL);
}
template <int N>
__device__ __forceinline__ float acc_read() {
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(x) : "n"(N) : PF64_ACC_ALL);
    return x;
}

template <class F, int... I>
__device__ __forceinline__ void unroll_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void unroll(F&& f) {
    unroll_impl(f, std::make_integer_sequence<int, N>{});
}

// The softmax of one 64-key sub-tile for BOTH query blocks as a flat list of single instructions (per block 32 scores per lane = 16
// pairs; arg = 16 * block + pair):
//   0 max3 of a pair into chain (pair & 1) | 1 reduce + new reference | 2 alpha | 3 / 4 fma of the pair's scores | 5 / 6 exp2 |
//   7 packed conversion into P
// -- the two blocks alternate, the fma / exp2 / conversion of consecutive pairs are staggered so that no instruction directly
// follows its producer.  piece[k]: the piece of half P (40 MFMAs: 16-key step g = pieces 10g .. 10g+9) behind whose MFMA the
// instruction goes; a conversion of a pair of step g's scores not before piece 10(g+1) (P(u-1) of that step has been read by
// then); 40 = behind the half.
constexpr int HP_PIECES = 40;
struct SoftmaxOps {
    int n = 0, nslots = 0;
    int type[256] = {}, arg[256] = {}, slot[256] = {}, piece[256] = {};
    constexpr void add(int t, int a, int w = 1) {
        type[n] = t, arg[n] = a, slot[n] = nslots;
        ++n;
        nslots += w;
    }
};
constexpr SoftmaxOps make_softmax_ops() {
    SoftmaxOps o;
    // maxima: pieces 0 .. 5
    for (int q = 0; q < 16; ++q)
        for (int b = 0; b < 2; ++b) {
            o.add(0, 16 * b + q);
            o.piece[o.n - 1] = (2 * q + b) * 6 / 32;
        }
    // the dependent chain from the maxima to the new reference (types 10 .. 14) and on to alpha (15, 16): ONE link per piece.  A
    // dependent VALU instruction issues ~22 cycles after its producer (tools/probes/issue_cost.hip: chains four apart run at 5.6
    // cycles an instruction, free ones at 4.25) and an in-order wave issues nothing else meanwhile, MFMAs included: the chain as
    // one block cost 900 cycles a step.  Every link's input is pinned behind its piece's MFMA (an empty asm volatile: the selection
    // DAG moves plain arithmetic across scheduling fences).
    for (int t = 10; t <= 16; ++t)
        for (int b = 0; b < 2; ++b) {
            o.add(t, 16 * b);
            o.piece[o.n - 1] = 6 + (t - 10);
        }
    // exponentials: pieces 12 .. 39 (the reference is known after type 14 = piece 10)
    for (int g = 0; g < 19; ++g)
        for (int b = 0; b < 2; ++b) {
            const int pc = 12 + g * 28 / 19;
            if (g < 16) {
                o.add(3, 16 * b + g), o.piece[o.n - 1] = pc;
                o.add(4, 16 * b + g), o.piece[o.n - 1] = pc;
            }
            if (g >= 1 && g - 1 < 16) {
                o.add(5, 16 * b + g - 1), o.piece[o.n - 1] = pc;
                o.add(6, 16 * b + g - 1), o.piece[o.n - 1] = pc;
            }
            if (g >= 3) {
                const int pair = g - 3, st = pair >> 2;  // pairs 4 st .. 4 st + 3 hold the scores of 16-key step st
                o.add(7, 16 * b + pair);
                o.piece[o.n - 1] = pc < 10 * (st + 1) ? 10 * (st + 1) : pc;
            }
        }
    return o;
}
static constexpr SoftmaxOps SM_OPS = make_softmax_ops();

// The same for ONE block inside the 72-piece full step (pieces 0-15 S^T of block A, 16-31 S^T of block B, 32-71 the 40 PV pieces):
// maxima from piece p_max (three pieces), the chain one link a piece behind them, exponentials spread over [p_exp, p_end]; a
// conversion of 16-key step st's scores not before piece 32 + 10 (st + 1); 72 = behind the step.  arg = pair.
constexpr int FS_PIECES = 72;
constexpr void add_exp_ops(SoftmaxOps& o, int pair_lo, int pair_hi, int p0, int p1, bool rolling) {
    const int np = pair_hi - pair_lo;
    for (int g = 0; g < np + 3; ++g) {
        const int pc = p0 + g * (p1 - p0) / (np + 2);
        if (g < np) {
            o.add(3, pair_lo + g), o.piece[o.n - 1] = pc;
            o.add(4, pair_lo + g), o.piece[o.n - 1] = pc;
        }
        if (g >= 1 && g - 1 < np) {
            o.add(5, pair_lo + g - 1), o.piece[o.n - 1] = pc;
            o.add(6, pair_lo + g - 1), o.piece[o.n - 1] = pc;
        }
        if (g >= 3) {
            const int pair = pair_lo + g - 3, st = pair >> 2, lo = rolling ? 32 + 10 * (st + 1) : 0;
            o.add(7, pair);
            o.piece[o.n - 1] = pc < lo ? lo : pc;
        }
    }
}
constexpr SoftmaxOps make_block_ops(int p_max, int pair_hi, int p_exp, int p_end) {
    SoftmaxOps o;
    for (int q = 0; q < 16; ++q) {
        o.add(0, q);
        o.piece[o.n - 1] = p_max + (q >> 3) + (q >= 12 ? 1 : 0);  // pairs 0-7 (the first 32 keys) first: their MFMAs are the older
    }
    for (int t = 10; t <= 16; ++t) {
        o.add(t, 0);
        o.piece[o.n - 1] = p_max + 3 + (t - 10);
    }
    add_exp_ops(o, 0, pair_hi, p_exp, p_end, true);
    return o;
}
constexpr SoftmaxOps make_tail_ops(int pair_lo, int p0, int p1) {
    SoftmaxOps o;
    add_exp_ops(o, pair_lo, 16, p0, p1, false);
    return o;
}
// Block B's last four pairs (the scores of its last 16 keys) are exponentiated in the NEXT step's pieces 0-15 -- S^T of block B is
// only overwritten from piece 16 on, their conversions could not be placed before piece 72 anyway, and those pieces carry little else
constexpr int FS_TAIL_PAIR = 12;
static constexpr SoftmaxOps FS_OPS_A = make_block_ops(17, 16, 27, 60);
static constexpr SoftmaxOps FS_OPS_B = make_block_ops(33, FS_TAIL_PAIR, 43, 71);
static constexpr SoftmaxOps FS_OPS_BT = make_tail_ops(FS_TAIL_PAIR, 0, 15);

}  // namespace pf64

template <int D>
struct Prefill64Smem {
    static constexpr int SUB = 64;                 // keys per sub-tile
    static constexpr int STAGE = SUB * D * 2;      // one K (or V) sub-tile: 16 KB
    static constexpr int NS = 4;                   // ring depth
    static constexpr int K_OFF = 0;
    static constexpr int V_OFF = NS * STAGE;
    static constexpr int BYTES = 2 * NS * STAGE;   // 128 KB
    static_assert(BYTES <= 160 * 1024, "LDS budget");
};

// the kernel addresses a sequence's K / V rows with 32-bit byte offsets from its first row
static inline bool prefill64_fits(int64_t max_input_len, int64_t k_st, int64_t v_st) {
    const int64_t st = k_st > v_st ? k_st : v_st;
    return max_input_len * st * 2 + 4096 < ((int64_t)1 << 32);
}

template <int D>
__global__ __launch_bounds__(256, 1) void prefill64_kernel(PrefillParams p) {
    using namespace pf64;
    constexpr int KS = D / 16;
    constexpr int QB = 256;  // queries per workgroup
    static_assert(D == 128, "prefill is instantiated for head_dim 128");
    using SM = Prefill64Smem<D>;
    constexpr int SUB = SM::SUB;
    constexpr float THR = 8.f;  // a row's reference maximum moves when some row of the wave is more than 2^THR above its own
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63;
    const int c = l & 31;
    const int h = l >> 5;
    // grid order: prefill.h (longest query blocks first over all heads and sequences; the q heads of a KV head on one XCD)
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
    const int kvh = head / p.G;

    // ---- lane constants (the LDS layouts of stage1_np.h / prefill.h: K chunks XOR-ed by key & 15, V chunks by 4*(key & 3)) --------
    const int dpos = l & 15, dkey = l >> 4;
    const int tg = l >> 4, tx = l & 15;
    const int vtr_row_b = (4 * (tg >> 1) + (tx >> 2)) * D * 2 + (tx & 1) * 8;
    int vfrag_b[4], kfrag_b[KS];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) vfrag_b[blk] = vtr_row_b + (4 * (blk ^ (tx >> 2)) + 2 * (tg & 1) + ((tx & 3) >> 1)) * 16;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kfrag_b[ks] = c * D * 2 + ((((h ^ c) & 15) * 16) ^ (32 * ks));

    // K / V rows of this sequence and KV head: scalar bases, 32-bit byte offsets per lane (prefill64_fits)
    const char* kbase = reinterpret_cast<const char*>(p.k + start * p.k_st + (int64_t)kvh * p.k_sh);
    const char* vbase = reinterpret_cast<const char*>(p.v + start * p.v_st + (int64_t)kvh * p.v_sh);
    const uint32_t krow = (uint32_t)p.k_st * 2u, vrow = (uint32_t)p.v_st * 2u;  // bytes per token
    const uint32_t klast = (uint32_t)(len - 1) * krow, vlast = (uint32_t)(len - 1) * vrow;
    const int nsub = (len + SUB - 1) / SUB;
    const int nt_all = min(4 * m + 4, nsub);     // sub-tiles the workgroup stages
    const int nt_w = min(4 * m + w + 1, nsub);   // ... of which this wave folds the first nt_w
    // one of this wave's 4 K + 4 V request instructions of sub-tile t: keys 16w + 4i + dkey (two adds and a min per request;
    // padding aliases the last token and is masked by the causal test)
    uint32_t kro[4], kch[4], vro[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int key = 16 * w + 4 * i + dkey;
        kro[i] = (uint32_t)key * krow;
        vro[i] = (uint32_t)key * vrow;
        kch[i] = (uint32_t)(dpos ^ (key & 15)) * 16u;
    }
    const uint32_t vch = (uint32_t)(dpos ^ (4 * (dkey & 3))) * 16u;
    auto issue_k1 = [&](int t, int i) __attribute__((always_inline)) {
        const uint32_t off = min((uint32_t)(SUB * t) * krow + kro[i], klast) + kch[i];
        dma16s(kbase, off, SM::K_OFF + (uint32_t)(t & (SM::NS - 1)) * SM::STAGE + (uint32_t)(16 * w + 4 * i) * 256u);
    };
    auto issue_v1 = [&](int t, int i) __attribute__((always_inline)) {
        const uint32_t off = min((uint32_t)(SUB * t) * vrow + vro[i], vlast) + vch;
        dma16s(vbase, off, SM::V_OFF + (uint32_t)(t & (SM::NS - 1)) * SM::STAGE + (uint32_t)(16 * w + 4 * i) * 256u);
    };
    auto issue_tile = [&](int t) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_k1(t, i);
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_v1(t, i);
    };
    // (every wave issues its eight requests of sub-tile u+2 in EVERY step, beyond the last sub-tile as harmless re-reads of the last
    //  token into a stage nobody reads any more: no branch inside the woven halves -- the compiler sinks the softmax into the last of
    //  the blocks a branch cuts a half into -- and one wait count for all steps)
    issue_tile(0);
    issue_tile(1);

    // ---- the two query blocks of this wave; their Q fragments (B operand: 8 halves at d = 16 ks + 8 h) into a[128:191] ---------
    const int q0 = m * QB + 64 * w;
    const int qiA = q0 + c, qiB = q0 + 32 + c;
    acc_zero<ACC_O, 128>();
    acc_zero<ACC_L, 32>();
    {
        half8 qfA[KS], qfB[KS];
        const int ra = qiA < len ? qiA : len - 1, rb = qiB < len ? qiB : len - 1;
        const _Float16* qa = p.q + (start + ra) * p.q_st + (int64_t)head * p.q_sh + 8 * h;
        const _Float16* qb = p.q + (start + rb) * p.q_st + (int64_t)head * p.q_sh + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qfA[ks] = *reinterpret_cast<const half8*>(qa + 16 * ks);
            qfB[ks] = *reinterpret_cast<const half8*>(qb + 16 * ks);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unroll<KS>([&](auto ic) __attribute__((always_inline)) {
            constexpr int ks = decltype(ic)::value;
            acc_put4<ACC_Q + 4 * ks>(qfA[ks]);
            acc_put4<ACC_Q + 32 + 4 * ks>(qfB[ks]);
        });
    }

    float mref[2] = {-INFINITY, -INFINITY};  // reference maxima of this lane's query in block A / B (log2 domain, scaled)
    floatx16 sc[2][2];                        // S^T of the current sub-tile: [block][32-key half]
    half8 pp[2][2][2];                        // P^T (fp16): [block][32-key half][16-key step]
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[x][kb][r] = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 8; ++e) pp[x][kb][t][e] = (_Float16)0.f;
        }
    half8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (_Float16)1.f;
    asm volatile("" : "+v"(ones));

    typedef __attribute__((address_space(3))) short4v* lds_s4;
    union VFrag {
        short4v s4[2];
        half8 h8;
    };
    const float scale = p.scale_log2e;
    constexpr int PD = 3;  // fragments are read PD fragments (six to eight MFMAs) ahead of their first MFMA

    // ---- HALF Q of step u: S^T(u) of both blocks, 16 K fragments, 32 MFMAs (piece 2j = block A, 2j+1 = block B on fragment j =
    //      (k-step ks = j >> 1, 32-key half kb = j & 1)), and the eight request instructions of sub-tile u+2 ------------------------
    auto half_q = [&](int u) __attribute__((always_inline)) {
        const int kstage = SM::K_OFF + (u & (SM::NS - 1)) * SM::STAGE;
        half8 fr[16];
        auto load = [&](auto jc) __attribute__((always_inline)) -> half8 {
            constexpr int j = decltype(jc)::value, ks = j >> 1, kb = j & 1;
            if constexpr ((PF64_ABL & 2) != 0) return ones;
            else return *reinterpret_cast<const half8*>(smem + (kfrag_b[ks] + kstage) + 32 * kb * D * 2);
        };
        unroll<PD>([&](auto jc) __attribute__((always_inline)) { fr[decltype(jc)::value] = load(jc); });
        unroll<32>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value, j = i >> 1, x = i & 1, ks = j >> 1, kb = j & 1;
            if constexpr (x == 0 && j + PD < 16) fr[j + PD] = load(std::integral_constant<int, j + PD>{});
            if constexpr ((PF64_ABL & 4) != 0) asm volatile("" ::"v"(fr[j]));
            else if constexpr (ks == 0) mfma_s0<ACC_Q + 32 * x>(sc[x][kb], fr[j]);
            else mfma_s<ACC_Q + 32 * x + 4 * ks>(sc[x][kb], fr[j]);
            if constexpr ((i & 3) == 2 && !(PF64_ABL & 8)) {
                if constexpr (i < 16) issue_k1(u + 2, i >> 2);
                else issue_v1(u + 2, (i >> 2) - 4);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("s_nop 7" ::: "memory");  // (S^T's last MFMAs against the softmax's first reads: the compiler does not know)
    };

    // ---- HALF P of step u: O^T, L += V(u-1)^T / ones x P(u-1) for both blocks (has_pv: 16 V^T fragments, 40 MFMAs: 16-key step
    //      g = (kb = g >> 1, t = g & 1) is pieces 10g .. 10g+9 = column blocks 0..3 x (A, B), then the two row-sum MFMAs), woven with
    //      the softmax of S^T(u) -> P(u), reference maxima (has_sm).  Returns through alpha[] what O^T / L of each block still owe.
    auto half_p = [&](auto has_pv, auto has_sm, auto diag, int u, float (&alpha)[2]) __attribute__((always_inline)) {
        constexpr bool PV = decltype(has_pv)::value, SMX = decltype(has_sm)::value, DIAG = decltype(diag)::value;
        constexpr int NM = PV ? HP_PIECES : 0;
        const int vstage = SM::V_OFF + ((u - 1) & (SM::NS - 1)) * SM::STAGE;
        half8 fr[16];
        auto load = [&](auto fc) __attribute__((always_inline)) -> half8 {
            constexpr int f = decltype(fc)::value, g = f >> 2, bk = f & 3, kb = g >> 1, t = g & 1;
            if constexpr ((PF64_ABL & 2) != 0) return ones;
            const int vb = vfrag_b[bk] + vstage + (32 * kb + 16 * t) * D * 2;
            VFrag vf;
            vf.s4[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(uintptr_t)(vb));
            vf.s4[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(uintptr_t)(vb + 8 * D * 2));
            return vf.h8;
        };
        // softmax state of the two blocks
        // (no row is ever without a visible key -- key 0, or the row's own key on the diagonal -- so the reference is finite after a
        //  row's first sub-tile, masked scores become exp2(-inf) = 0 and nothing else needs a special case)
        float mx2[2][2] = {{-INFINITY, -INFINITY}, {-INFINITY, -INFINITY}}, m_new[2] = {mref[0], mref[1]};
        float tv[2][32], ev[2][32], ch0[2] = {0.f, 0.f}, ch1[2] = {0.f, 0.f};
        half8 pq[2][2][2];
        if constexpr ((PF64_ABL & 1024) != 0) {
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int t = 0; t < 2; ++t) pq[x][kb][t] = ones;
        }
        if constexpr (SMX && DIAG) {  // the wave's last sub-tile: causal mask (keys >= len are > every valid query); not woven
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = SUB * u + 32 * kb + 8 * (r >> 2) + 4 * h + (r & 3);
                        sc[x][kb][r] = key <= (x ? qiB : qiA) ? sc[x][kb][r] : -INFINITY;
                    }
        }
        auto sm_op = [&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value;
            constexpr int type = SM_OPS.type[k], x = SM_OPS.arg[k] >> 4, a = SM_OPS.arg[k] & 15;
            constexpr int e0 = 2 * a, e1 = 2 * a + 1;  // the pair's scores: sc[x][e >> 4][e & 15]
            if constexpr (type == 0) mx2[x][a & 1] = fmaxf(fmaxf(mx2[x][a & 1], sc[x][e0 >> 4][e0 & 15]), sc[x][e1 >> 4][e1 & 15]);
            else if constexpr (type == 10) {
                asm volatile("" : "+v"(mx2[x][0]), "+v"(mx2[x][1]));
                ch0[x] = fmaxf(mx2[x][0], mx2[x][1]);
            } else if constexpr (type == 11) {
                asm volatile("" : "+v"(ch0[x]));
                const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(ch0[x]), __float_as_uint(ch0[x]), false, false);
                ch0[x] = __uint_as_float(r2[0]), ch1[x] = __uint_as_float(r2[1]);
            } else if constexpr (type == 12) {
                asm volatile("" : "+v"(ch0[x]), "+v"(ch1[x]));
                ch0[x] = fmaxf(ch0[x], ch1[x]);
            } else if constexpr (type == 13) {
                asm volatile("" : "+v"(ch0[x]));
                ch0[x] = ch0[x] * scale;  // (scaling is monotonic: the max of the scaled scores)
            } else if constexpr (type == 14) {
                asm volatile("" : "+v"(ch0[x]));
                const bool move = __builtin_amdgcn_ballot_w64(ch0[x] > mref[x] + THR) != 0ull;  // wave-uniform
                m_new[x] = move ? fmaxf(mref[x], ch0[x]) : mref[x];
            } else if constexpr (type == 15) {
                asm volatile("" : "+v"(m_new[x]));
                ch0[x] = mref[x] - m_new[x];
            } else if constexpr (type == 16) {
                asm volatile("" : "+v"(ch0[x]));
                alpha[x] = __builtin_amdgcn_exp2f(ch0[x]);  // (exp2(-inf) = 0: the first sub-tile)
                mref[x] = m_new[x];
            } else if constexpr (type == 3) tv[x][e0] = __builtin_fmaf(sc[x][e0 >> 4][e0 & 15], scale, -m_new[x]);
            else if constexpr (type == 4) tv[x][e1] = __builtin_fmaf(sc[x][e1 >> 4][e1 & 15], scale, -m_new[x]);
            else if constexpr (type == 5) ev[x][e0] = __builtin_amdgcn_exp2f(tv[x][e0]);
            else if constexpr (type == 6) ev[x][e1] = __builtin_amdgcn_exp2f(tv[x][e1]);
            else if constexpr ((PF64_ABL & 1024) != 0) {  // (timing experiment: the conversions go to registers no MFMA reads)
                pq[x][e0 >> 4][(e0 & 15) >> 3][e0 & 7] = (_Float16)ev[x][e0];
                pq[x][e0 >> 4][(e0 & 15) >> 3][(e0 & 7) + 1] = (_Float16)ev[x][e1];
            } else {
                // (pinned on both sides: a conversion has no consumer in this half, so the selection DAG is free to put it anywhere
                //  between its exponentials and the end -- early it overlaps P(u-1)'s last readers, late all 32 sit behind the last MFMA)
                asm volatile("" : "+v"(ev[x][e0]), "+v"(ev[x][e1]));
                typedef _Float16 half2v __attribute__((ext_vector_type(2)));
                union {
                    half2v h2;
                    uint32_t u;
                } cv;
                cv.h2 = half2v{(_Float16)ev[x][e0], (_Float16)ev[x][e1]};
                asm volatile("" : "+v"(cv.u));
                union {
                    half8 h8;
                    uint32_t u[4];
                } pv;
                pv.h8 = pp[x][e0 >> 4][(e0 & 15) >> 3];
                pv.u[(e0 & 7) >> 1] = cv.u;
                pp[x][e0 >> 4][(e0 & 15) >> 3] = pv.h8;
            }
        };
        auto sm_piece = [&](auto ic) __attribute__((always_inline)) {  // the softmax instructions placed behind MFMA i (NM: the rest)
            constexpr int i = decltype(ic)::value;
            if constexpr (SMX && NM > 0 && i > 11 && i < NM) asm volatile("" : "+v"(m_new[0]), "+v"(m_new[1]));
            if constexpr (SMX && !(PF64_ABL & 1))
                unroll<SM_OPS.n>([&](auto kc) __attribute__((always_inline)) {
                    constexpr int k = decltype(kc)::value;
                    constexpr int ty = SM_OPS.type[k];
                    constexpr bool skip = ((PF64_ABL & 32) && ty == 0) || ((PF64_ABL & 64) && ty >= 10) || ((PF64_ABL & 128) && (ty == 3 || ty == 4)) ||
                                          ((PF64_ABL & 256) && (ty == 5 || ty == 6)) || ((PF64_ABL & 512) && ty == 7);
                    if constexpr ((SM_OPS.piece[k] < NM ? SM_OPS.piece[k] : NM) == i && !skip) sm_op(kc);
                });
        };
        if constexpr (PV) unroll<PD>([&](auto fc) __attribute__((always_inline)) { fr[decltype(fc)::value] = load(fc); });
        unroll<NM>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value, g = i / 10, r = i % 10, kb = g >> 1, t = g & 1;
            if constexpr (r < 8) {
                constexpr int bk = r >> 1, x = r & 1, f = 4 * g + bk;
                if constexpr (x == 0 && f + PD < 16) fr[f + PD] = load(std::integral_constant<int, f + PD>{});
                if constexpr ((PF64_ABL & 4) != 0) asm volatile("" ::"v"(fr[f]), "v"(pp[x][kb][t]));
                else mfma_acc<ACC_O + 64 * x + 16 * bk>(fr[f], pp[x][kb][t]);
            } else {
                constexpr int x = r - 8;
                if constexpr ((PF64_ABL & 4) == 0) mfma_acc<ACC_L + 16 * x>(ones, pp[x][kb][t]);
            }
            sm_piece(ic);
            if constexpr (!(PF64_ABL & 16)) __builtin_amdgcn_sched_barrier(0);
        });
        sm_piece(std::integral_constant<int, NM>{});
        if constexpr ((PF64_ABL & 1024) != 0) {
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int t = 0; t < 2; ++t) asm volatile("" ::"v"(pq[x][kb][t]));
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // ---- THE FULL STEP (1 <= u < nt_w - 1), 72 pieces in one block: S^T(u) of block A (pieces 0-15, with the sub-tile's requests),
    //      S^T(u) of block B (16-31) and the PV of sub-tile u-1 (32-71, as in half P).  Block A's softmax starts behind piece 16, as
    //      soon as its S^T is complete, block B's behind piece 32: the ~210 softmax instructions are spread over 56 MFMAs instead of
    //      40 (half Q's MFMAs had nothing beside them, half P's pieces took 49 cycles).  The K fragments are read twice for that (once
    //      per block), the V^T fragments once.
    auto full_step = [&](auto tail_in, int u, float (&alpha)[2]) __attribute__((always_inline)) {
        constexpr bool TAIL_IN = decltype(tail_in)::value;  // the previous step was a full step: block B's last pairs are still to do
        const int kstage = SM::K_OFF + (u & (SM::NS - 1)) * SM::STAGE;
        const int vstage = SM::V_OFF + ((u - 1) & (SM::NS - 1)) * SM::STAGE;
        constexpr int PDK = 5, PDV = 3;
        half8 fk[32], fv[16];
        auto load_k = [&](auto jc) __attribute__((always_inline)) -> half8 {
            constexpr int j = decltype(jc)::value & 15, ks = j >> 1, kb = j & 1;
            return *reinterpret_cast<const half8*>(smem + (kfrag_b[ks] + kstage) + 32 * kb * D * 2);
        };
        auto load_v = [&](auto fc) __attribute__((always_inline)) -> half8 {
            constexpr int f = decltype(fc)::value, g = f >> 2, bk = f & 3, kb = g >> 1, t = g & 1;
            const int vb = vfrag_b[bk] + vstage + (32 * kb + 16 * t) * D * 2;
            VFrag vf;
            vf.s4[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(uintptr_t)(vb));
            vf.s4[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(uintptr_t)(vb + 8 * D * 2));
            return vf.h8;
        };
        float mx2[2][2] = {{-INFINITY, -INFINITY}, {-INFINITY, -INFINITY}}, m_new[2] = {mref[0], mref[1]};
        float tv[2][32], ev[2][32], ch0[2] = {0.f, 0.f}, ch1[2] = {0.f, 0.f};
        // Every softmax instruction is its own asm volatile statement: the statements keep their source order among themselves and
        // among the MFMAs, which is the whole point -- plain arithmetic is placed by the selection DAG wherever it likes between its
        // operands and its users (fences do not bind it), and what came out had five MFMAs back to back here and twenty VALU
        // instructions there: SQ_VALU_MFMA_COEXEC_CYCLES 35 % of the MFMA time.  The registers are still the compiler's.  (An asm
        // statement that reads what the asm statement right in front of it wrote gets an s_nop 0 from the hazard recogniser; the
        // order below never does that.)
        float negm[2] = {-mref[0], -mref[1]};
        auto sm_op = [&](auto xc, auto tc, auto ac) __attribute__((always_inline)) {
            constexpr int x = decltype(xc)::value, type = decltype(tc)::value, a = decltype(ac)::value;
            constexpr int e0 = 2 * a, e1 = 2 * a + 1;  // the pair's scores: sc[x][e >> 4][e & 15]
            if constexpr (type == 0) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(mx2[x][a & 1]) : "v"(sc[x][e0 >> 4][e0 & 15]), "v"(sc[x][e1 >> 4][e1 & 15]));
            else if constexpr (type == 10) asm volatile("v_max_f32 %0, %1, %2" : "=v"(ch0[x]) : "v"(mx2[x][0]), "v"(mx2[x][1]));
            else if constexpr (type == 11) asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(ch0[x]), "=&v"(ch1[x]));
            else if constexpr (type == 12) asm volatile("v_max_f32 %0, %0, %1" : "+v"(ch0[x]) : "v"(ch1[x]));
            else if constexpr (type == 13) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(ch0[x]) : "s"(scale));
            else if constexpr (type == 14) {
                asm volatile("" : "+v"(ch0[x]));
                const bool move = __builtin_amdgcn_ballot_w64(ch0[x] > mref[x] + THR) != 0ull;  // wave-uniform
                m_new[x] = move ? fmaxf(mref[x], ch0[x]) : mref[x];
                negm[x] = -m_new[x];
                asm volatile("" : "+v"(negm[x]));
            } else if constexpr (type == 15) asm volatile("v_add_f32 %0, %1, %2" : "=v"(ch0[x]) : "v"(mref[x]), "v"(negm[x]));
            else if constexpr (type == 16) {
                asm volatile("v_exp_f32 %0, %1" : "=v"(alpha[x]) : "v"(ch0[x]));  // (exp2(-inf) = 0: a row's first sub-tile)
                mref[x] = m_new[x];
            } else if constexpr (type == 3) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(tv[x][e0]) : "v"(sc[x][e0 >> 4][e0 & 15]), "s"(scale), "v"(negm[x]));
            else if constexpr (type == 4) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(tv[x][e1]) : "v"(sc[x][e1 >> 4][e1 & 15]), "s"(scale), "v"(negm[x]));
            else if constexpr (type == 5) asm volatile("v_exp_f32 %0, %1" : "=v"(ev[x][e0]) : "v"(tv[x][e0]));
            else if constexpr (type == 6) asm volatile("v_exp_f32 %0, %1" : "=v"(ev[x][e1]) : "v"(tv[x][e1]));
            else {
                union {
                    half8 h8;
                    uint32_t u[4];
                } pv;
                pv.h8 = pp[x][e0 >> 4][(e0 & 15) >> 3];
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pv.u[(e0 & 7) >> 1]) : "v"(ev[x][e0]), "v"(ev[x][e1]));
                pp[x][e0 >> 4][(e0 & 15) >> 3] = pv.h8;
            }
        };
        auto sm_piece = [&](auto ic) __attribute__((always_inline)) {  // the softmax instructions behind MFMA i (72: the rest)
            constexpr int i = decltype(ic)::value;
            unroll<FS_OPS_A.n>([&](auto kc) __attribute__((always_inline)) {
                constexpr int k = decltype(kc)::value;
                if constexpr (FS_OPS_A.piece[k] == i)
                    sm_op(std::integral_constant<int, 0>{}, std::integral_constant<int, FS_OPS_A.type[k]>{}, std::integral_constant<int, FS_OPS_A.arg[k]>{});
            });
            unroll<FS_OPS_B.n>([&](auto kc) __attribute__((always_inline)) {
                constexpr int k = decltype(kc)::value;
                if constexpr (FS_OPS_B.piece[k] == i)
                    sm_op(std::integral_constant<int, 1>{}, std::integral_constant<int, FS_OPS_B.type[k]>{}, std::integral_constant<int, FS_OPS_B.arg[k]>{});
            });
            if constexpr (TAIL_IN)
                unroll<FS_OPS_BT.n>([&](auto kc) __attribute__((always_inline)) {
                    constexpr int k = decltype(kc)::value;
                    if constexpr (FS_OPS_BT.piece[k] == i)
                        sm_op(std::integral_constant<int, 1>{}, std::integral_constant<int, FS_OPS_BT.type[k]>{}, std::integral_constant<int, FS_OPS_BT.arg[k]>{});
                });
        };
        unroll<PDK>([&](auto jc) __attribute__((always_inline)) { fk[decltype(jc)::value] = load_k(jc); });
        unroll<(PF64_CUT < FS_PIECES ? PF64_CUT : FS_PIECES)>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            if constexpr (i < 32) {
                constexpr int x = i >> 4, j = i & 15, ks = j >> 1, kb = j & 1;
                if constexpr (i + PDK < 32) fk[i + PDK] = load_k(std::integral_constant<int, i + PDK>{});
                if constexpr (i >= 32 - 2 * PDV && !(i & 1)) fv[(i - (32 - 2 * PDV)) >> 1] = load_v(std::integral_constant<int, ((i - (32 - 2 * PDV)) >> 1)>{});
                if constexpr (ks == 0) mfma_s0<ACC_Q + 32 * x>(sc[x][kb], fk[i]);
                else mfma_s<ACC_Q + 32 * x + 4 * ks>(sc[x][kb], fk[i]);
                if constexpr (i < 16 && (i & 1)) {
                    if constexpr (i < 8) issue_k1(u + 2, i >> 1);
                    else issue_v1(u + 2, (i >> 1) - 4);
                }
            } else {
                constexpr int q = i - 32, g = q / 10, r = q % 10, kb = g >> 1, t = g & 1;
                if constexpr (r < 8) {
                    constexpr int bk = r >> 1, x = r & 1, f = 4 * g + bk;
                    if constexpr (x == 0 && f + PDV < 16) fv[f + PDV] = load_v(std::integral_constant<int, f + PDV>{});
                    mfma_acc<ACC_O + 64 * x + 16 * bk>(fv[f], pp[x][kb][t]);
                } else {
                    constexpr int x = r - 8;
                    mfma_acc<ACC_L + 16 * x>(ones, pp[x][kb][t]);
                }
            }
            sm_piece(ic);
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (PF64_CUT >= FS_PIECES) sm_piece(std::integral_constant<int, FS_PIECES>{});
        else alpha[0] = alpha[1] = 1.f;
        __builtin_amdgcn_sched_barrier(0);
    };
    // block B's deferred pairs, all at once (the step after a full step is not a full step)
    auto flush_tail = [&]() __attribute__((always_inline)) {
        typedef _Float16 half2v __attribute__((ext_vector_type(2)));
        unroll<16 - FS_TAIL_PAIR>([&](auto ac) __attribute__((always_inline)) {
            constexpr int a = FS_TAIL_PAIR + decltype(ac)::value, e0 = 2 * a, e1 = 2 * a + 1;
            const float x0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[1][e0 >> 4][e0 & 15], scale, -mref[1]));
            const float x1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[1][e1 >> 4][e1 & 15], scale, -mref[1]));
            pp[1][e0 >> 4][(e0 & 15) >> 3][e0 & 7] = (_Float16)x0;
            pp[1][e0 >> 4][(e0 & 15) >> 3][(e0 & 7) + 1] = (_Float16)x1;
        });
    };
    auto rescale = [&](const float (&alpha)[2]) __attribute__((always_inline)) {
        if (__builtin_amdgcn_ballot_w64(alpha[0] != 1.f) != 0ull) {
            acc_scale<ACC_O, 64>(alpha[0]);
            acc_scale<ACC_L, 16>(alpha[0]);
        }
        if (__builtin_amdgcn_ballot_w64(alpha[1] != 1.f) != 0ull) {
            acc_scale<ACC_O + 64, 64>(alpha[1]);
            acc_scale<ACC_L + 16, 16>(alpha[1]);
        }
    };
    using T = std::true_type;
    using F = std::false_type;

#ifdef DEFT_EXPERIMENTS
    // shader cycles of this wave (tools/prefill64_cycles.py): 0 wait + barrier | 1 woven steps | 2 their count | 3 whole loop
    unsigned long long cyc[4] = {0, 0, 0, 0};
    const unsigned long long c_loop = __builtin_readcyclecounter();
#endif
    // step u of this wave: half Q for sub-tile u (u < nt_w), half P with sub-tile u-1's PV (u >= 1) and sub-tile u's softmax (u < nt_w)
    bool tail_pending = false;  // (wave-uniform) block B's last pairs of the previous step are still to be exponentiated
    for (int u = 0; u <= nt_all; ++u) {
        const bool has_q = u < nt_w;  // (wave-uniform)
#ifdef DEFT_EXPERIMENTS
        const unsigned long long c0 = p.dbg ? __builtin_readcyclecounter() : 0;
#endif
        if (u < nt_all) {
            wait_vm<8>();   // sub-tile u landed (this wave's part); younger: the eight requests of sub-tile u+1
            lds_barrier();  // ... everyone's part; and every wave is done with step u-1, i.e. with the stage sub-tile u+2 goes to
            if (!has_q) issue_tile(u + 2);  // (half Q carries them otherwise)
        }
#ifdef DEFT_EXPERIMENTS
        const unsigned long long c1 = p.dbg ? __builtin_readcyclecounter() : 0;
        cyc[0] += c1 - c0;
        if (p.dbg && u > 1 && u < nt_w) cyc[1] += c0 - cyc[3], ++cyc[2];  // (the previous step, if it was a full one too)
        cyc[3] = c1;
#endif
        if (u > nt_w) continue;  // (wave-uniform) this wave's sub-tiles are done: it only stages and keeps the barriers
        float alpha[2] = {1.f, 1.f};
        // S^T is dead between steps, which the compiler cannot see (a softmax without a half Q in front of it is impossible, not
        // unreachable): without this "definition" it carries all 64 registers around the loop, 32 v_mov_b64 a step
        auto kill_s = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) asm volatile("" : "=v"(sc[x][kb]));
        };
        if (u >= 1 && u < nt_w - 1) {  // the full step, one straight-line block
            if (tail_pending) full_step(T{}, u, alpha);
            else full_step(F{}, u, alpha);
            tail_pending = true;
            rescale(alpha);
            // (S^T of block B lives on into the next step's first pieces; block A's is dead)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) asm volatile("" : "=v"(sc[0][kb]));
            continue;
        }
        if (tail_pending) {
            flush_tail();
            tail_pending = false;
        }
        if (has_q) half_q(u);
        if (u == 0) {
            if (nt_w == 1) half_p(F{}, T{}, T{}, u, alpha);
            else half_p(F{}, T{}, F{}, u, alpha);
        } else if (u < nt_w) {
            if (u == nt_w - 1) half_p(T{}, T{}, T{}, u, alpha);
            else half_p(T{}, T{}, F{}, u, alpha);
        } else {
            half_p(T{}, F{}, F{}, u, alpha);
        }
        rescale(alpha);
        kill_s();
    }
    wait_vm<0>();
#ifdef DEFT_EXPERIMENTS
    if (p.dbg && l == 0 && L < 1024) {  // [8192 x 8 workgroup stamps][1024 workgroups][8 waves][8]
        unsigned long long* d2 = p.dbg + (int64_t)8192 * 8 + ((int64_t)L * 8 + w) * 8;
        d2[0] = cyc[0], d2[1] = cyc[1], d2[2] = cyc[2], d2[3] = __builtin_readcyclecounter() - c_loop;
        d2[4] = (unsigned long long)nt_w, d2[5] = (unsigned long long)nt_all;
    }
#endif
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs' results (asm MFMAs: the compiler does not know)
    // ---- normalise and store: lane (c, h) holds d = 32 bk + 8 j + 4 h + (0..3) of its query; every element of L is the row sum ---
    auto store = [&](auto xblock, int qi) __attribute__((always_inline)) {
        constexpr int XB = decltype(xblock)::value;
        float o[64];
        unroll<64>([&](auto ic) __attribute__((always_inline)) { o[decltype(ic)::value] = acc_read<ACC_O + 64 * XB + decltype(ic)::value>(); });
        const float l_run = acc_read<ACC_L + 16 * XB>();
        if (qi < len) {
            const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
            _Float16* op = p.o + (start + qi) * p.o_st + (int64_t)head * p.o_sh + 4 * h;
#pragma unroll
            for (int bk = 0; bk < 4; ++bk)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    half4 v4 = {(_Float16)(o[16 * bk + 4 * j] * inv), (_Float16)(o[16 * bk + 4 * j + 1] * inv),
                                (_Float16)(o[16 * bk + 4 * j + 2] * inv), (_Float16)(o[16 * bk + 4 * j + 3] * inv)};
                    *reinterpret_cast<half4*>(op + 32 * bk + 8 * j) = v4;
                }
        }
    };
    store(std::integral_constant<int, 0>{}, qiA);
    store(std::integral_constant<int, 1>{}, qiB);
}

}  // namespace deft
