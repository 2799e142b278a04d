// Plan kernels (once per decode step, shared by all layers): TreeMetadata arrays -> unit list -> one 2 KiB record per
// 128-slot tile, in the order the stage-1 kernels consume them.
//
// Included by deft_kernels.hip after plan_records.h (record layout PLAN_*, header words) and before stage1_np.h.  Flatten (tree_cache.py:618-881 blocks): flatten_units_kernel + flatten_records_kernel; Node
// (tree_attention.py:14-293 entries): node_units_kernel + node_records_kernel.  The unit kernels are one workgroup:
// wave 0 decides the runs, all waves write units and the tile-parallel record order ("Plan kernels").
#pragma once

namespace deft {

// ---------------------------------------------------------------------------
// Plan kernels (once per decode step): metadata -> unit list -> records
// ---------------------------------------------------------------------------
// Byte offset of a pool slot's row, or (bit 63 | offset into k_new / v_new) when the slot is one of
// this step's new tokens and the caller uses the fused append.  A record workgroup first hashes the step's new slots
// into LDS (open addressing, slot -> new row), so the test is O(1) per slot instead of a scan over all new tokens
// (a batched forest has hundreds); more new tokens than the table holds fall back to the scan.
constexpr int NEWMAP_SIZE = 2048;  // entries (power of two); holds up to NEWMAP_SIZE / 2 new tokens
struct NewMap {
    int* keys;  // [NEWMAP_SIZE] slot or -1
    int* vals;  // [NEWMAP_SIZE] new-row index
    bool hashed;
};
__device__ inline NewMap newmap_build(int* keys, int* vals, const int32_t* cache_loc, int n_new) {
    NewMap m{keys, vals, n_new > 0 && n_new <= NEWMAP_SIZE / 2};
    if (m.hashed) {
        for (int i = threadIdx.x; i < NEWMAP_SIZE; i += blockDim.x) keys[i] = -1;
        __syncthreads();
        for (int i = threadIdx.x; i < n_new; i += blockDim.x) {
            const int s = cache_loc[i];
            unsigned h = ((unsigned)s * 2654435761u) & (NEWMAP_SIZE - 1);
            while (true) {
                const int prev = atomicCAS(&keys[h], -1, s);
                if (prev == -1 || prev == s) {
                    if (prev == -1) vals[h] = i;
                    else atomicMin(&vals[h], i);  // (a slot listed twice: the first row, as the scan would find)
                    break;
                }
                h = (h + 1) & (NEWMAP_SIZE - 1);
            }
        }
        __syncthreads();
    }
    return m;
}
__device__ __forceinline__ int64_t plan_rowoff(int64_t slot, int64_t kv_stride_slot, const int32_t* cache_loc, int n_new,
                                               int64_t new_row_bytes, const NewMap& m) {
    if (m.hashed) {
        unsigned h = ((unsigned)(int)slot * 2654435761u) & (NEWMAP_SIZE - 1);
        while (true) {
            const int k = m.keys[h];
            if (k == -1) break;
            if (k == (int)slot) return ((int64_t)1 << 63) | ((int64_t)m.vals[h] * new_row_bytes);
            h = (h + 1) & (NEWMAP_SIZE - 1);
        }
        return slot * kv_stride_slot * 2;
    }
    for (int i = 0; i < n_new; ++i)
        if ((int64_t)cache_loc[i] == slot) return ((int64_t)1 << 63) | ((int64_t)i * new_row_bytes);
    return slot * kv_stride_slot * 2;  // fp16 bytes
}

struct UnitList {   // all int32, capacity `cap` each
    int32_t* src;   // Flatten: block index; Node: entry index
    int32_t* aux;   // Flatten: 0;           Node: 128-slot tile index within the entry
    int32_t* pass;  // 32-row pass of the unit's virtual query rows
    int32_t* flags; // bit 0: opens a run (query list differs from the previous unit's); bits 1..: unit index of the run's first unit
    int32_t* prow;  // first partial row of the unit's tile
    // tile-parallel record order (stage1_np.h), indexed by RECORD: which unit the record packs, and its chunk
    int32_t* perm;   // record -> unit
    int32_t* ch_n;   // tiles of the chunk this record leads (itself included); 0 = follower
    int32_t* ch_fb;  // record index of the chunk's first follower (followers are consecutive records)
    // union groups (Flatten, tile-parallel order), indexed by GROUP id = aux - 1: consecutive leaf tiles whose query
    // lists differ but are small are folded by ONE workgroup over the union of their queries
    int32_t* gn;    // queries in the union (<= UNION_CAP)
    int32_t* gq;    // [UNION_CAP][cap] query rows of the union, ascending
    int32_t* grow;  // [UNION_CAP][cap] the partial row that carries each union query (its first occurrence in the group)
};
constexpr int UNION_CAP = DEFT_UNION_CAP;

// Record order for the tile-parallel stage 1 (one workgroup per chunk, stage1_np.h).  A run of `nt` units with
// one query list is cut into S = ceil(nt / C) chunks; chunk p folds units p, p + S, p + 2S, ... of the run
// (interleaved: the chunks of a run advance through the pool side by side, one contiguous front).  Records are
// renumbered: the leaders of all chunks first (run by run, so the long shared-prefix chunks are dispatched first),
// then the followers, those of one chunk consecutive.  hdr[1] = number of leaders.  One thread.
// The runs come from a table the emitting thread kept in LDS (run k = units r0[k] .. r0[k] + nt[k] - 1, uni[k] = it is
// a union group): walking the unit arrays in global memory instead costs one dependent load per unit (~100 us for the
// north-star tree, once per decode step).
struct RunTable {
    int* r0;   // first unit of the run
    int* nt;   // units (tiles) in the run
    int* uni;  // non-zero: a union group = one chunk whatever its length (the Flatten kernel keeps the group id + 1 here)
    int n;    // runs recorded
    int cap;  // capacity; n > cap = overflow, fall back to scanning the unit arrays
};

constexpr int LONG_CHUNK = 4;  // tiles per chunk from which a chunk's leader is dispatched ahead of the short ones

// `Hkv` = stage-1 work items per chunk leader: KV heads -- or, NEGATED, head PAIRS (head_dim 64 runs two heads to a pool row,
// stage1_np.h HD2: half as many items as heads, each with two softmaxes' worth of arithmetic per tile, so its launches are
// short of workgroups where the head_dim-128 launch of the same tree is not -- chunk lengths follow the GQA rule and then
// shrink until the launch has 3/4 of a workgroup per resident slot).
__device__ inline void np_record_order(const UnitList& ul, int R, int Hkv, int G, int slots, int chunk_c, int32_t* hdr,
                                       RunTable rt) {
    const bool pairs = Hkv < 0;
    Hkv = pairs ? -Hkv : Hkv;
    if (rt.n > rt.cap) {  // rebuild the table is impossible: scan (slow path, huge trees only)
        rt.n = 0;
        rt.cap = 0;
    }
    const bool have = rt.cap > 0;
    // run iteration: either the LDS table or a scan over flags / aux
    auto for_runs = [&](auto&& fn) {
        if (have) {
            for (int k = 0; k < rt.n; ++k) fn(rt.r0[k], rt.nt[k], rt.uni[k]);
        } else {
            for (int r = 0; r < R;) {
                const int id = ul.flags[r] >> 1;
                int e = r + 1;
                while (e < R && (ul.flags[e] >> 1) == id) ++e;
                fn(r, e - r, (ul.aux[r] != 0 || (ul.pass[r] >> 16)) ? 1 : 0);  // (union group; a window plan's overflow run: Flatten aux -1, Node bit 16 of pass; a Node pack (< 0) is one unit anyway)
                r = e;
            }
        }
    };
    int C = chunk_c;
    if (C <= 0) {
        // Measured on MI355X (tools/np_sweep.sh): per-workgroup cost (descriptor round trip, 32-row epilogue) favours
        // long chunks, the critical path and the number of resident slots bound them.  8 tiles for long shared
        // prefixes, 4 from 8 tiles on, halved while fewer than ~0.3 workgroups per slot would be left.
        int lmax = 0;
        for_runs([&](int, int nt, int uni) {
            if (!uni && nt > lmax) lmax = nt;
        });
        C = lmax >= 32 ? 8 : (lmax >= 8 ? 4 : (lmax >= 4 ? 2 : 1));
        // ... and with GQA a chunk should not outlast the launch: the passes of a shared tile are separate chunks that
        // hit L2, so with T tiles per resident slot in all, chunks longer than ~T leave a few workgroups running
        // alone at the end (Llama-3 north-star tree: 2.8 tiles per slot, 8-tile chunks ran 18 us of a 25 us launch
        // with a quarter of the slots occupied; 4-tile chunks: 19.9 us).  MHA, every tile from HBM, measured the
        // other way (1-token branches, 2 tiles per slot: 8-tile chunks 19.1 us, 4-tile chunks 21.0).
        if (G > 1 || pairs) {
            int64_t tiles_all = 0;
            for_runs([&](int, int nt, int) { tiles_all += nt; });
            int cmax = 1;
            while (cmax < 8 && (int64_t)cmax * slots < tiles_all * Hkv) cmax <<= 1;
            if (C > cmax) C = cmax;
        }
        for (; C > 1; C >>= 1) {
            int64_t n = 0;
            for_runs([&](int, int nt, int uni) { n += uni ? 1 : (nt + C - 1) / C; });
            if ((pairs ? 4 : 10) * n * Hkv >= 3LL * slots) break;
        }
        // ... but no run is cut into more than 16 chunks while chunks may still grow (<= 8 tiles): every chunk of a
        // shared prefix is one more partial row for EVERY query below it, and the merge reads its rows 16 at a time
        // (one 8192-token prefix under 8 branches, Llama-3-8B: 32 chunks of 2 tiles 23.7 us per layer, 16 of 4 tiles 21.1).
        while (C < 8 && lmax > 16 * C) C <<= 1;
        // MHA, a shared prefix that DOMINATES the tree (round 6, profiles/r6_chunk_sweep_short.txt): with 8-tile chunks a 4096-token
        // prefix is 4 chunks per KV head -- 128 workgroups for 80 % of the launch's bytes when the branches are short, half the CUs
        // pulling on them, each through a serial chain of 8 tiles.  While the rest of the tree is no larger than the prefix (T_other
        // <= lmax) and the prefix's chunks do not fill the CUs, it is cut into 6 chunks per KV head, into 7 while the rest is at most a
        // quarter of it (4k prefix: 5-6 and 4-5 tiles per chunk; a 6k prefix: 8 and 7 -- measured there too, profiles/r6_chunk_sweep_short.txt): north-star tree at 1 / 25 / 50 / 100 / 125 tokens per branch 20.7 -> 18.9, 23.9 -> 21.3,
        // 25.0 -> 23.4, 29.6 -> 27.7, 30.7 -> 29.3 us per layer; from 150 tokens on (and for every other BASELINE shape) nothing
        // changes -- there the leaf items keep the other CUs busy and longer chunks mean fewer partial rows.
        if (G == 1 && !pairs && C == 8 && chunk_c == 0) {  // (chunk_c = -1: the rules WITHOUT this one -- A/B builds, DEFT_NP_CHUNK=-1)
            int64_t tiles_all = 0, n8 = 0;
            for_runs([&](int, int nt, int uni) {
                tiles_all += uni == -1 ? 1 : nt;  // (a window plan's overflow run: mostly dormant tiles -- counted as one)
                n8 += uni ? 1 : (nt + 7) / 8;
            });
            const int64_t t_other = tiles_all - lmax;
            if ((int64_t)((lmax + 7) / 8) * Hkv * 2 < slots && t_other <= lmax && n8 * Hkv * 4 <= 5LL * slots) {
                // chunks per KV head for 7/16 (the rest is small) or 3/8 of the resident slots: 224 or 192 workgroups on this part --
                // 7 or 6 chunks under Llama-2-7B's 32 KV heads, where the rule was measured
                const int S = max(1, (int)(((4 * t_other <= lmax) ? 7LL * slots / 16 : 3LL * slots / 8) / Hkv));
                C = max(3, min(8, (lmax + S - 1) / S));
            }
        }
        // A launch that leaves CUs empty (fewer chunks than CUs = slots / 2) takes the next shorter chunk length -- powers
        // of two or not -- as long as that still fits one workgroup per CU: Medusa-64 (an 8-tile root under two query
        // chunks, 32 KV heads) 192 workgroups of 4 tiles -> 256 of 3: 15.8 -> 14.9 us per layer.  Such a launch may also cut its
        // longest run into up to 24 chunks (round 4: one 8192-token prefix under 8 branches, Llama-3-8B, 192 workgroups of 4
        // tiles -> 240 of 3: 15.4 -> 14.35 us per layer, stage 1 12.4 -> 11.5; tools/ab.py DEFT_NP_CHUNK=1..8: 14.7 12.6 11.3 12.4 - 15.2 - 18.0).
        if (C > 2 && lmax <= 24 * (C - 1)) {
            int64_t n0 = 0, n1 = 0;
            for_runs([&](int, int nt, int uni) {
                n0 += uni ? 1 : (nt + C - 1) / C;
                n1 += uni ? 1 : (nt + C - 2) / (C - 1);
            });
            if (2 * n0 * Hkv < slots && 2 * n1 * Hkv <= slots) --C;
        }
        // MHA, ONE short shared prefix (<= 8 tiles: configs[1]'s 1024-token prompt) under one query chunk, the rest of the tree no
        // more than four times its tiles (branches of up to 128 tokens): 2-tile chunks.  The 4-tile chain of the prefix is the
        // launch's critical path there -- everything else is single tiles, and the prefix's chunks take at most half the CUs either
        // way (late round 6, forced chunk lengths, profiles/r6_chunk_sweep_short.txt block 5: 1k x 32 at 10 / 25 / 50 / 75 tokens per
        // branch 12.6 -> 11.9, 15.2 -> 13.8, 16.9 -> 14.8, 18.9 -> 17.3 us per layer; 100: 19.7 -> 20.0; from 129 on nothing
        // changes).  Two query chunks (Medusa-64: two runs over the root) measured the other way and keep their rule.
        if (G == 1 && !pairs && chunk_c == 0 && C > 2 && lmax <= 8) {
            int64_t tiles_all = 0;
            int long_runs = 0;
            for_runs([&](int, int nt, int uni) {
                tiles_all += uni == -1 ? 1 : nt;
                long_runs += (!uni && nt >= 3) ? 1 : 0;
            });
            if (long_runs == 1 && (int64_t)((lmax + 1) / 2) * Hkv * 4 <= slots && tiles_all - lmax <= 4 * lmax) C = 2;
        }
        // GQA: the launch should fit the resident slots -- chunks grow (up to 8 tiles) until there are fewer workgroups than slots:
        // a second round of workgroups, every one with its ramp and its epilogue, costs more than longer chunks do (late round 4,
        // rule variants of the shipped build, tools/ab_rules.sh: ToT-50 on Llama-3-8B, 6 passes over 32 root tiles + 28 leaf
        // tiles, 8 KV heads: 608 workgroups of 4 tiles 20.5 us per layer, 560 of 5 21.0, 512 of 6 20.9, 464 of 7 19.65, 416 of 8 20.0).
        // (Only if some length up to 8 does fit: a launch with more leaf tiles than slots keeps its short chunks.)
        if (G > 1 && !pairs)
            for (int c2 = C; c2 <= 8; ++c2) {
                int64_t n = 0;
                for_runs([&](int, int nt, int uni) { n += uni ? 1 : (nt + c2 - 1) / c2; });
                if (n * Hkv < slots) {
                    C = c2;
                    break;
                }
            }
    }
    // Leaders of LONG chunks (>= LONG_CHUNK tiles: the shared prefixes) come before all others, whatever run they belong
    // to: a capped grid hands item b + W to the workgroup that finishes item b, so with the leaders run by run a batch of
    // trees (prefix chunks, leaf tiles, prefix chunks, leaf tiles, ...) gave the workgroups that already held one
    // 8-tile chunk a second one (8 trees of 8k x 8 as one tree object: 71 -> 57 us per layer).  Longest first, as in prefill.
    int NL = 0, NLong = 0;
    for_runs([&](int, int nt, int uni) {
        const int S = uni ? 1 : (nt + C - 1) / C;
        NL += S;
        if ((!uni || uni == -1) && nt / S >= LONG_CHUNK) NLong += S;  // (-1: a window plan's overflow run -- one chunk of up to nt tiles, dispatched early)
    });
    int liL = 0, liS = NLong, fi = NL;
    for_runs([&](int r, int nt, int uni) {
        const int S = uni ? 1 : (nt + C - 1) / C;  // a union group is one chunk
        int& li = ((!uni || uni == -1) && nt / S >= LONG_CHUNK) ? liL : liS;
        for (int p = 0; p < S; ++p) {
            const int cnt = (nt - p + S - 1) / S;
            ul.perm[li] = r + p;
            ul.ch_n[li] = cnt;
            ul.ch_fb[li] = fi;
            ++li;
            for (int j = 1; j < cnt; ++j, ++fi) {
                ul.perm[fi] = r + p + j * S;
                ul.ch_n[fi] = 0;
                ul.ch_fb[fi] = 0;
            }
        }
    });
    hdr[1] = NL;
}

// Record order of the tile-parallel stage 1, from the unit kernel's LDS run table (the rules of np_record_order above, same
// result), in two parts.  record_order_wave0: ONE wave decides the chunk length C from sums / maxima over the runs, then each
// run's first leader and first follower record by prefix sums over the runs' chunk counts, into rT0 / rSp -- two arrays of run_cap
// words OF ITS OWN (round 5: it used to reuse the callers' run fields, so the units had to be written first and the barrier in
// between drained their stores: 2-4 us of a 17 us kernel; now the other waves write the units WHILE wave 0 decides, and the
// barrier between the two parts is an LDS barrier).  record_order_write: all waves, after that barrier.  sMeta[2..3]: two shared words.
__device__ inline void record_order_wave0(const RunTable& rt, int NR, int* rT0, int* rSp, int* sMeta, int32_t* hdr, int Hkv, int G,
                                          int slots, int chunk_c) {
    const int lane = threadIdx.x & 63;
    const bool pairs = Hkv < 0;  // (np_record_order: head pairs)
    Hkv = pairs ? -Hkv : Hkv;
    {
        auto wave_sum = [&](auto&& f) {
            int acc = 0;
            for (int k = lane; k < NR; k += 64) acc += f(rt.nt[k], rt.uni[k]);
            for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m, 64);
            return acc;
        };
        int C = chunk_c;
        if (C <= 0) {
            int lmax = 0;
            for (int k = lane; k < NR; k += 64)
                if (!rt.uni[k] && rt.nt[k] > lmax) lmax = rt.nt[k];
            for (int m = 32; m > 0; m >>= 1) lmax = max(lmax, __shfl_xor(lmax, m, 64));
            C = lmax >= 32 ? 8 : (lmax >= 8 ? 4 : (lmax >= 4 ? 2 : 1));
            if (G > 1 || pairs) {
                const int64_t tiles_all = wave_sum([](int nt, int) { return nt; });
                int cmax = 1;
                while (cmax < 8 && (int64_t)cmax * slots < tiles_all * Hkv) cmax <<= 1;
                if (C > cmax) C = cmax;
            }
            for (; C > 1; C >>= 1) {
                const int64_t n = wave_sum([C](int nt, int uni) { return uni ? 1 : (nt + C - 1) / C; });
                if ((pairs ? 4 : 10) * n * Hkv >= 3LL * slots) break;
            }
            while (C < 8 && lmax > 16 * C) C <<= 1;  // (np_record_order: at most 16 chunks per run while C < 8)
            if (G == 1 && !pairs && C == 8 && chunk_c == 0) {  // (np_record_order: a shared prefix that dominates an MHA tree is cut shorter)
                const int64_t tiles_all = wave_sum([](int nt, int uni) { return uni == -1 ? 1 : nt; });  // (np_record_order: overflow runs count as one tile)
                const int64_t n8 = wave_sum([](int nt, int uni) { return uni ? 1 : (nt + 7) / 8; });
                const int64_t t_other = tiles_all - lmax;
                if ((int64_t)((lmax + 7) / 8) * Hkv * 2 < slots && t_other <= lmax && n8 * Hkv * 4 <= 5LL * slots) {
                    const int S = max(1, (int)(((4 * t_other <= lmax) ? 7LL * slots / 16 : 3LL * slots / 8) / Hkv));
                    C = max(3, min(8, (lmax + S - 1) / S));
                }
            }
            if (C > 2 && lmax <= 24 * (C - 1)) {       // (np_record_order: fill the CUs of a launch that leaves some empty)
                const int64_t n0 = wave_sum([C](int nt, int uni) { return uni ? 1 : (nt + C - 1) / C; });
                const int64_t n1 = wave_sum([C](int nt, int uni) { return uni ? 1 : (nt + C - 2) / (C - 1); });
                if (2 * n0 * Hkv < slots && 2 * n1 * Hkv <= slots) --C;
            }
            if (G == 1 && !pairs && chunk_c == 0 && C > 2 && lmax <= 8) {  // (np_record_order: one short shared prefix under single-tile branches)
                const int64_t tiles_all = wave_sum([](int nt, int uni) { return uni == -1 ? 1 : nt; });
                const int long_runs = wave_sum([](int nt, int uni) { return (!uni && nt >= 3) ? 1 : 0; });
                if (long_runs == 1 && (int64_t)((lmax + 1) / 2) * Hkv * 4 <= slots && tiles_all - lmax <= 4 * lmax) C = 2;
            }
            if (G > 1 && !pairs)  // (np_record_order: a GQA launch should fit the resident slots)
                for (int c2 = C; c2 <= 8; ++c2) {
                    const int64_t n = wave_sum([c2](int nt, int uni) { return uni ? 1 : (nt + c2 - 1) / c2; });
                    if (n * Hkv < slots) {
                        C = c2;
                        break;
                    }
                }
        }
        // leaders: long chunks first (np_record_order), each class in run order; followers in run order
        int leadL = 0, leadS = 0, foll = 0;
        for (int base = 0; base < NR; base += 64) {
            const int k = base + lane;
            const int nt = k < NR ? rt.nt[k] : 0;
            const int S = k < NR ? (rt.uni[k] ? 1 : (nt + C - 1) / C) : 0;
            const bool lng = k < NR && (!rt.uni[k] || rt.uni[k] == -1) && S > 0 && nt / S >= LONG_CHUNK;  // (-1: an overflow run of a window plan)
            int aL = lng ? S : 0, aS = lng ? 0 : S, b = nt - S;  // inclusive scans over the lanes
            for (int d = 1; d < 64; d <<= 1) {
                const int uL = __shfl_up(aL, d, 64), uS = __shfl_up(aS, d, 64), ub = __shfl_up(b, d, 64);
                if (lane >= d) {
                    aL += uL;
                    aS += uS;
                    b += ub;
                }
            }
            if (k < NR) {
                rT0[k] = lng ? leadL + aL - S : -(leadS + aS - S) - 1;  // short runs: position inside their class, resolved below
                rSp[k] = foll + b - (nt - S);
            }
            leadL += __shfl(aL, 63, 64);
            leadS += __shfl(aS, 63, 64);
            foll += __shfl(b, 63, 64);
        }
        for (int k = lane; k < NR; k += 64)
            if (rT0[k] < 0) rT0[k] = leadL + (-rT0[k] - 1);
        if (lane == 0) {
            sMeta[2] = C;
            sMeta[3] = leadL + leadS;
            hdr[1] = leadL + leadS;
        }
    }
}

__device__ inline void record_order_write(const UnitList& ul, const RunTable& rt, int NR, const int* rT0, const int* rSp,
                                          const int* sMeta) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    {
        const int C = sMeta[2], NL = sMeta[3];
        for (int k = wave; k < NR; k += nwaves) {
            const int first = rt.r0[k], nt = rt.nt[k];
            const int S = rt.uni[k] ? 1 : (nt + C - 1) / C;  // a union group is one chunk
            const int li = rT0[k], fi = NL + rSp[k];
            const int q = nt / S, rem = nt - q * S;  // chunk p folds units p, p + S, ...: q + 1 of them for p < rem, else q
            for (int u = lane; u < nt; u += 64) {
                const int j = u / S, pc = u - j * S;
                const int fb = fi + pc * (q - 1) + min(pc, rem);  // followers of the chunks before pc
                if (j == 0) {
                    ul.perm[li + pc] = first + pc;
                    ul.ch_n[li + pc] = q + (pc < rem ? 1 : 0);
                    ul.ch_fb[li + pc] = fb;
                } else {
                    ul.perm[fb + j - 1] = first + u;
                    ul.ch_n[fb + j - 1] = 0;
                    ul.ch_fb[fb + j - 1] = 0;
                }
            }
        }
    }
}

// Union group of leaf tiles starting at block t (Flatten, tile-parallel order): up to `ulen` consecutive blocks whose
// query lists hold at most `ucap` queries each and at most `ucap` distinct queries together.  Returns the number of
// blocks taken; uq / urow / un = the union's queries in order of first occurrence and the partial row (block_q
// position) of each first occurrence.  sQ: the [NB][UNION_CAP] LDS table of the small blocks' lists, or nullptr
// (lists read from block_q).  Everything lives in registers, every loop is unrolled over UNION_CAP.
__device__ inline int union_group(int t, int NB, int ulen, int ucap, const int* sCnt, const int* sOff, const int* sQ,
                                  const int64_t* block_q, int (&uq)[UNION_CAP], int (&urow)[UNION_CAP], int& un) {
    static_assert(UNION_CAP % 4 == 0, "the query table is read as int4s");
    un = 0;
#pragma unroll
    for (int j = 0; j < UNION_CAP; ++j) uq[j] = 0, urow[j] = 0;
    int te = t;
    while (te < NB && te - t < ulen) {
        const int cnt = sCnt[te];
        if (cnt > ucap) break;
        const int off = sOff[te];
        int qv[UNION_CAP];
        if (sQ) {
#pragma unroll
            for (int i4 = 0; i4 < UNION_CAP / 4; ++i4) {
                const int4 v = *reinterpret_cast<const int4*>(sQ + te * UNION_CAP + 4 * i4);
                qv[4 * i4] = v.x, qv[4 * i4 + 1] = v.y, qv[4 * i4 + 2] = v.z, qv[4 * i4 + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < UNION_CAP; ++i) qv[i] = i < cnt ? (int)block_q[off + i] : 0;
        }
        int add = 0;  // queries of block te that are new to the union
#pragma unroll
        for (int i = 0; i < UNION_CAP; ++i) {
            bool found = false;
#pragma unroll
            for (int j = 0; j < UNION_CAP; ++j) found |= (j < un) & (uq[j] == qv[i]);
            add += (i < cnt && !found) ? 1 : 0;
        }
        if (un + add > ucap) break;
#pragma unroll
        for (int i = 0; i < UNION_CAP; ++i) {
            bool found = i >= cnt;
#pragma unroll
            for (int j = 0; j < UNION_CAP; ++j) found |= (j < un) & (uq[j] == qv[i]);
            if (!found) {  // first occurrence: this tile's row carries the query's partial
#pragma unroll
                for (int j = 0; j < UNION_CAP; ++j)
                    if (j == un) {
                        uq[j] = qv[i];
                        urow[j] = off + i;
                    }
                ++un;
            }
        }
        ++te;
    }
    return te - t;
}

// Flatten: one workgroup.  Phase 1 (parallel over blocks): does block t open a run, how many passes.
// Phase 2 (one thread): emit units run by run, pass-major inside a run so that consecutive units fold.
// Tile-parallel order only (union_len > 1): short runs of leaf tiles with small, different query lists -- a branch's
// tail shares a block with the next branch's head, so their lists go {a}, {a,b}, {b}, {b,c} ... -- are grouped, up
// to union_len tiles and UNION_CAP queries, into ONE run over the union of their queries (per-slot masks keep a
// query away from keys that are not on its path), so that one workgroup folds them and writes one partial per query.
__global__ __launch_bounds__(1024) void flatten_units_kernel(const int64_t* block_q, const int64_t* block_q_cnts,
                                                            const int64_t* block_q_offset, int NBc, int G, int cap,
                                                            UnitList ul, int32_t* hdr, int Hkv, int slots, int chunk_c,
                                                            int union_len, int run_cap, int qtab, int par,
                                                            const int32_t* dims, int32_t* row_q, int rows, int win_tiles,
                                                            const int64_t* block_lens, int solo_full) {
    constexpr int np = 1;  // records in the tile-parallel order (leaders first); the only stage-1 form
    // NBc = block CAPACITY (sizes the tables); with `dims` (device-side metadata, tree_plan.h) the block count of this
    // step is read from the device, so that one captured launch serves every step of a structural epoch
    const int NB = dims ? min(dims[5], NBc) : NBc;
    // window plan (window.h): blocks NBreg .. NB - 1 are overflow tiles -- `win_tiles` consecutive blocks per query chunk, each set
    // ONE run kept in ONE chunk (run-table marker -1), never part of a union group or of an interleaved run
    const int NBreg = (dims && win_tiles > 0) ? min(dims[WIN_DIM_NB], NB) : NB;
    if (dims)  // rows beyond this step's partial rows must read "dead" (the row lists are built over the capacity)
        for (int i = threadIdx.x; i < rows; i += blockDim.x) row_q[i] = -1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* sOpen = reinterpret_cast<int*>(smem);  // [NBc]
    int* sPass = sOpen + NBc;                   // [NBc]
    int* sCnt = sPass + NBc;                    // [NBc] block_q_cnts
    int* sOff = sCnt + NBc;                     // [NBc] block_q_offset
    // [NBc][UNION_CAP] query lists of the blocks small enough to join a union group (qtab: the table fits in LDS):
    // the one-thread phase below otherwise waits for a global load per leaf tile
    int* sQ = sOff + NBc;
    int* sRun = sQ + (qtab ? UNION_CAP * NBc : 0);
    RunTable rt{sRun, sRun + run_cap, sRun + 2 * run_cap, 0, run_cap};
    __shared__ int sEst[2];  // GQA: leaf-like blocks (at most three queries); 32-row passes of all other blocks
    if (threadIdx.x == 0) sEst[0] = sEst[1] = 0;
    for (int t = threadIdx.x; t < NB; t += blockDim.x) {
        const int cnt = (int)block_q_cnts[t];
        sCnt[t] = cnt;
        sOff[t] = (int)block_q_offset[t];
        sPass[t] = (cnt * G + MQ - 1) / MQ;
    }
    __syncthreads();
    if (G > 1 || Hkv > 0) {  // (MHA: blocks a union group may hold -- at most four queries -- and all others, among the plan's own blocks)
        int small = 0, passes = 0;
        for (int t = threadIdx.x; t < (G > 1 ? NB : NBreg); t += blockDim.x) {
            if (sCnt[t] <= (G > 1 ? 3 : 4)) ++small;
            else passes += sPass[t];
        }
        for (int m = 32; m > 0; m >>= 1) {
            small += __shfl_xor(small, m, 64);
            passes += __shfl_xor(passes, m, 64);
        }
        if ((threadIdx.x & 63) == 0 && (small | passes)) {
            atomicAdd(&sEst[0], small);
            atomicAdd(&sEst[1], passes);
        }
    }
    // Does block t's query list differ from the one 1 / 2 / 3 / 4 blocks back?  Half a wave per block, lane i on
    // list entry i (and i + 32, ... for longer lists), the five loads of an entry independent of each other: one
    // round trip per block instead of one per list entry (a shared-prefix block has 32 entries, and a thread walking
    // them with early exit waited ~1 us for each).
    //   bit 0: opens a run;  bits 1..3: the list also differs from the one 2 / 3 / 4 blocks back (a node with more
    //   than 32 queries is emitted by the reference as alternating blocks -- queries 0..31 / 32..63 / ... of the same
    //   128 slots, tree_cache.py:763-799 -- so its blocks repeat with period ceil(queries / 32)); inside an ordinary
    //   run the longer periods are never looked at and read as "differs".
    {
        const int lane = threadIdx.x & 63, half = lane >> 5, li = lane & 31;
        const int nhw = (blockDim.x >> 6) * 2;
        for (int t0 = (threadIdx.x >> 6) * 2; t0 < NB; t0 += nhw) {
            const int t = t0 + half;
            const bool live = t < NB;
            const int cnt = live ? sCnt[t] : 0;
            const int a = live ? sOff[t] : 0;
            int diff = 0;  // bit pd-1: some entry of this lane differs from the block pd back
            bool same_cnt[4];
            for (int pd = 1; pd <= 4; ++pd) same_cnt[pd - 1] = live && t >= pd && sCnt[t - pd] == cnt;
            for (int i = li; i < cnt; i += 32) {
                const int64_t mine = block_q[a + i];
                int64_t other[4];
                for (int pd = 1; pd <= 4; ++pd) other[pd - 1] = same_cnt[pd - 1] ? block_q[sOff[t - pd] + i] : mine;
                for (int pd = 1; pd <= 4; ++pd) diff |= (other[pd - 1] != mine) ? (1 << (pd - 1)) : 0;
                if (qtab && cnt <= UNION_CAP) sQ[t * UNION_CAP + i] = (int)mine;
            }
            int bits = 0;
            for (int pd = 1; pd <= 4; ++pd) {
                const unsigned long long b = __ballot((diff >> (pd - 1)) & 1);
                const bool any = ((half ? (b >> 32) : b) & 0xffffffffull) != 0;
                bits |= (!same_cnt[pd - 1] || any) ? (1 << (pd - 1)) : 0;
            }
            if (!(bits & 1)) bits = 0xe;  // not opening a run: longer periods read as "differs"
            if (t >= NBreg) bits = ((t - NBreg) % win_tiles == 0) ? 0xf : 0xe;  // (overflow tiles: a run of their own per query chunk)
            // bit 4 (round 6, MHA): a FULL tile of ONE query.  A union group that would hold nothing but such tiles is not formed (phase
            // 1b): at branch lengths that are multiples of 128 every leaf tile is one, the launch ends with the eight-tile root chunks'
            // serial chains, and many one-tile items keep the other slots streaming until then where groups of three all start at
            // t = 0 and leave them idle (profiles/r5_flatten_vs_node_aligned.txt: DeFT-Node's one-item-per-tile plan beat Flatten by
            // 1.0-1.5 us per layer there).  Groups that MIX full tiles with the straddling tiles around them stay: un-grouping every
            // full tile cost the 200- / 300- / 400-token trees 1.7-2.5 us per layer (profiles/r6_solo_full_tiles.txt)
            if (solo_full && live && li == 0 && cnt == 1 && t < NBreg && (int)block_lens[t] == TILE) bits |= 16;
            if (live && li == 0) sOpen[t] = bits;
        }
    }
    __syncthreads();
    // Phase 1b (tile-parallel order): for every block, how many blocks a union group starting there would take
    // (0 = none), one thread per block, into bits 8.. of sOpen -- the walk below then only looks the answer up.
    // union queries whose virtual rows fit one pass; four unless the group-length rule below (or the experiments knob) asks for more
    int ucap = (4 * G <= MQ) ? 4 : MQ / G;
    // GQA (late round 4, tools/knob_layer.sh DEFT_NP_UNION=2 | cap << 8, us per layer): pairs of tiles with at most THREE queries
    // between them -- a branch end straddling a block boundary, [a, b] + [b, c] -- fold as one group: the north-star tree on
    // Llama-3-8B 20.4 -> 19.5 (400 one-tile leaf workgroups become 200); pairs of four queries (ToT-50's [l1, l2] + [l3, l4]) cost
    // 1.7 us there and 1-2 on the 8-tree forest, so the cap is three.
    // ... and only where pairing is what makes the launch FIT the resident slots (rule variants of the shipped build at other
    // branch lengths, tools/ab_rules.sh BENCH_EXTRA="--branch-len N", us per layer with / without pairs: 100 tokens 17.75 / 16.15 --
    // 456 workgroups fit anyway, pairs only make them longer; 200 tokens 18.77 / 19.75 -- 656 become 456; 400 tokens 25.0 / 25.05 and
    // 1000 tokens 38.3 / 37.55 -- far more leaf tiles than slots either way).  Estimated per KV head from the blocks alone: the
    // leaf-like blocks (at most three queries) are a workgroup each, everything else a workgroup per four passes.
    // The group length is the SHORTEST (2, 3 or 4 tiles) with which the launch fits (300-token branches: 75 leaf tiles -- pairs
    // leave 560 workgroups, triples 456: 22.8 -> 20.8 us per layer; at 200 tokens pairs fit, and triples cost 1.2 us).
    int gqa_ulen = 1;
    if (G > 1) {
        ucap = min(ucap, 3);
        const int hk = Hkv < 0 ? -Hkv : Hkv;
        const int64_t rest = (sEst[1] + 3) / 4;
        if ((rest + sEst[0]) * hk > slots)
            for (int u = 2; u <= 4; ++u)
                if ((rest + (sEst[0] + u - 1) / u) * hk <= slots) {
                    gqa_ulen = u;
                    break;
                }
    }
    // MHA, large launches (where groups of three were the measured rule): three or four leaf blocks per group by LIST SCHEDULING the
    // launch's work items in tile times on the slots one KV head has (late round 6, profiles/r6_union_len_sweep.txt).  The shared
    // prefix's chunks start first and run C tiles; the groups fill the other slots round by round (the prefix's slots join when it is
    // done), the remainder -- a work item of its own -- takes the first free slot.  Fours are taken where threes finish BEFORE the prefix
    // does -- leaving its few workgroups alone with the memory system -- and fours keep the slots busy for longer without a longer
    // makespan: the north-star tree from ~205 to ~270 tokens per branch (forced lengths, shipped build: 254 tokens 41.4 -> 37.9 us per
    // layer, 262 41.6 -> 39.8, 220 / 240 within 1-2 %; 150 / 180 / 200 / 300 / 400 stay with three, which measures 1-5 % faster there).
    // A frozen step used to jump +11 % from 240 to 254 tokens per branch.
    int mha_ulen = 3;
    if (G == 1 && Hkv > 0) {
        const int S = slots / Hkv, nL = sEst[0], nP = NBreg - nL;
        const int Cest = nP >= 32 ? 8 : (nP >= 8 ? 4 : (nP >= 4 ? 2 : 1));
        const int npre = (nP + Cest - 1) / Cest;
        // (only where the model was validated: a prefix of >= 32 blocks in 8-tile chunks.  On the 1k prefix it picks four at 400 tokens per
        //  branch where three measure 3 % faster.)
        if (nP >= 32 && npre < S && nL > 0) {
            auto sched = [&](int u, int& M, int& F) {
                int g = nL / u, t = 0, last_take = 0, last_avail = 0;
                const int r = nL - g * u;
                F = 0;
                while (g > 0) {
                    const int avail = (S - npre) + (t >= Cest ? npre : 0);
                    const int take = min(avail, g);
                    g -= take;
                    last_take = take;
                    last_avail = avail;
                    t += u;
                    F = t;
                }
                if (r > 0) F = (last_take < last_avail || F == 0) ? max(F, (F > 0 ? F - u : 0) + r) : F + r;
                M = max(F, Cest);
            };
            int M3, F3, M4, F4;
            sched(3, M3, F3);
            sched(4, M4, F4);
            // (four only where threes leave the prefix alone, F3 < C, and fours do so for less time: at 300 tokens per branch threes
            //  already end a round behind the prefix, the model's makespan says four, and the shipped build measures three 3 % faster)
            if (F3 < Cest && F4 > F3 && M4 <= M3) mha_ulen = 4;
        }
    }
    if (((union_len >> 8) & 0xff) > 0) ucap = min(min(UNION_CAP, MQ / G), (union_len >> 8) & 0xff);  // (experiments: bits 8..15 of the knob SET the union's query cap)
    const int taper = (union_len >> 16) & 0xff;  // (experiments: bits 16..23 -- the last `taper` % of the blocks stay single, the `taper` % in front of them pair)
    auto union_len_at = [&](int t) {
        int ulen = union_len & 0xff;
        if (taper && G == 1 && Hkv > 0) {
            const int k1 = NBreg * taper / 100;
            if (t >= NBreg - k1) return 1;
            if (t >= NBreg - 2 * k1) return 2;
        }
        // (head pairs, Hkv < 0: groups of two -- their launches want workgroups, tools/ab_step.py on the head_dim-64 north-star tree;
        //  three measured 0.9 us per layer faster in the experiments build and 0.4 slower in the shipped one, tools/ab_lib.sh)
        if (ulen <= 0) ulen = G > 1 ? gqa_ulen : (Hkv < 0 ? 2 : ((int64_t)NB * Hkv < 2048 ? 4 : mha_ulen));  // measured, tools/np_sweep.sh / tools/ab.py
        return ulen;
    };
    if (np && ucap >= 2)
        for (int t = threadIdx.x; t < NBreg; t += blockDim.x) {
            const int ulen = union_len_at(t);
            int g = 0;
            if (ulen > 1 && sCnt[t] <= ucap) {
                bool short_run = NBreg - t < ulen;  // the run that opens at t is shorter than a group
                for (int u = t + 1; u < t + ulen && u < NBreg; ++u) short_run |= (sOpen[u] & 1) != 0;
                if (short_run) {
                    int uq[UNION_CAP], urow[UNION_CAP], un;
                    g = union_group(t, NBreg, ulen, ucap, sCnt, sOff, qtab ? sQ : nullptr, block_q, uq, urow, un);
                    if (g < 2) g = 0;
                    bool all_solo = g >= 2;  // (nothing but full tiles of one query each: no group)
                    for (int u = t; u < t + g; ++u) all_solo &= (sOpen[u] & 16) != 0;
                    if (all_solo) g = 0;
                }
            }
            sOpen[t] = (sOpen[t] & 0x1f) | (g << 8);
        }
    __syncthreads();
    // Phase 2: wave 0 walks the blocks and decides the runs (every lane takes the same decisions; the lanes only split
    // the searches for the next run boundary).  `par`: every run gets an entry of the LDS table, and the units and the
    // record order are then written by all waves (phases 3 and 4) -- one thread emitting them costs ~0.4 us per unit
    // and pass, 75 us per decode step for the north-star tree and 2 ms for a 100k-token prefix under 48 branches.
    // Otherwise (tables beyond the LDS) lane 0 emits as it walks.
    int* sMeta = sRun + (par ? 7 : 3) * run_cap;  // [8]: units, runs, chunk length, leaders, "written by all waves"
    int* rT0 = sRun + 3 * run_cap;    // par: first block of the run
    int* rSp = sRun + 4 * run_cap;    // par: block stride | pass << 8
    int* rLead = sRun + 5 * run_cap;  // par: the run's first leader record            (record_order_wave0)
    int* rFoll = sRun + 6 * run_cap;  // par: the run's first follower record - leaders
    const int lane = threadIdx.x & 63;
    const int par_req = par;
    if (threadIdx.x < 64) {
      // (a second walk, lane 0 emitting, if the runs did not fit the table: trees with very many small runs)
      for (int attempt = 0; attempt < 2; ++attempt) {
        par = par_req && attempt == 0;
        rt.n = 0;
        int r = 0, ng = 0;
        // first block >= from whose sOpen has a bit of `mask` set (NB if none)
        auto find = [&](int from, int mask) {
            for (int base = from; base < NB; base += 64) {
                const int t = base + lane;
                const unsigned long long b = __ballot(t < NB && (sOpen[t] & mask));
                if (b) return base + __ffsll((long long)b) - 1;
            }
            return NB;
        };
        // units of one run: blocks t0, t0 + st, ... < te, pass ps, aux (0, or union group + 1)
        auto emit_run = [&](int t0, int te, int st, int aux, int ps) {
            int n = st == 1 ? te - t0 : (te - t0 + st - 1) / st;
            if (n > cap - r) n = cap - r;
            if (n <= 0) return;
            const int first = r;
            if (par) {
                if (lane == 0 && rt.n < rt.cap) {
                    rt.r0[rt.n] = first;
                    rt.nt[rt.n] = n;
                    rt.uni[rt.n] = aux;
                    rT0[rt.n] = t0;
                    rSp[rt.n] = st | (ps << 8);
                }
                ++rt.n;
            } else {
                if (lane == 0)
                    for (int j = 0; j < n; ++j) {
                        const int t = t0 + j * st;
                        ul.src[first + j] = t;
                        ul.aux[first + j] = aux;
                        ul.pass[first + j] = ps;
                        ul.flags[first + j] = (first << 1) | (j == 0 ? 1 : 0);
                        ul.prow[first + j] = sOff[t];
                    }
                if (rt.n < rt.cap && lane == 0) {
                    rt.r0[rt.n] = first;
                    rt.nt[rt.n] = n;
                    rt.uni[rt.n] = aux;
                }
                ++rt.n;
            }
            r += n;
        };
        for (int ta = 0; ta < NB;) {
            // ---- overflow tiles of one query chunk (window plan): one run per pass, one chunk each ---------------
            if (ta >= NBreg) {
                const int te = min(NB, ta + win_tiles);
                for (int ps = 0; ps < sPass[ta]; ++ps) emit_run(ta, te, 1, -1, ps);
                ta = te;
                continue;
            }
            // ---- union group starting at ta (phase 1b) ------------------------------------------------------
            const int glen = sOpen[ta] >> 8;
            if (glen >= 2 && r < cap) {
                if (!par && lane == 0) {  // (the parallel form writes the group in phase 3)
                    int uq[UNION_CAP], urow[UNION_CAP], un;
                    union_group(ta, NB, glen, ucap, sCnt, sOff, qtab ? sQ : nullptr, block_q, uq, urow, un);
                    ul.gn[ng] = un;
                    for (int j = 0; j < UNION_CAP; ++j) {
                        ul.gq[j * cap + ng] = j < un ? uq[j] : 0;
                        ul.grow[j * cap + ng] = j < un ? urow[j] : 0;
                    }
                }
                emit_run(ta, ta + glen, 1, ng + 1, 0);
                ++ng;
                ta += glen;
                continue;
            }
            const int tb = find(ta + 1, 1);
            const int passes = sPass[ta];
            const int cnt_a = sCnt[ta];
            // ---- P query chunks of one node, alternating block by block: P interleaved runs ------------------
            if (tb - ta == 1 && cnt_a == MQ) {  // (the first chunk of such a node is always full: cheap filter)
                int P = 0;
                for (int pd = 2; pd <= 4 && !P; ++pd) {
                    bool rep = ta + 2 * pd <= NB;
                    for (int t = ta + pd; rep && t < ta + 2 * pd; ++t) rep = !(sOpen[t] & (1 << (pd - 1)));
                    if (rep) P = pd;
                }
                if (P) {
                    const int te = find(ta + P, 1 << (P - 1));
                    for (int par_ = 0; par_ < P; ++par_) {
                        const int pp = sPass[ta + par_];
                        for (int ps = 0; ps < pp; ++ps) emit_run(ta + par_, te, P, 0, ps);
                    }
                    ta = te;
                    continue;
                }
            }
            for (int ps = 0; ps < passes; ++ps) emit_run(ta, tb, 1, 0, ps);
            ta = tb;
        }
        if (lane == 0) {
            hdr[0] = r;
            hdr[1] = 0;
            hdr[HDR_ERR] = 0;
            hdr[HDR_QLISTS] = 0;
            if (par && rt.n <= rt.cap) {
                sMeta[0] = r;
                sMeta[1] = rt.n;
                sMeta[4] = 1;
            } else if (!par) {
                sMeta[4] = 0;
                if (np) np_record_order(ul, r, Hkv, G, slots, chunk_c, hdr, rt);
            }
        }
        if (!par || rt.n <= rt.cap) break;
      }
    }
    __syncthreads();
    if (!sMeta[4]) return;
    // Phases 3 and 4 side by side: wave 0 decides the record order of the tile-parallel stage 1 (record_order_wave0) while the
    // other waves write the units of run k, one wave per run, one lane per unit; an LDS barrier; all waves write the order.
    const int NR = sMeta[1];
    const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    if (np && wave == 0) record_order_wave0(rt, NR, rLead, rFoll, sMeta, hdr, Hkv, G, slots, chunk_c);
    for (int k = np ? wave - 1 : wave; k < NR && k >= 0; k += np ? nwaves - 1 : nwaves) {
        const int first = rt.r0[k], n = rt.nt[k], aux = rt.uni[k], t0 = rT0[k], st = rSp[k] & 0xff, ps = rSp[k] >> 8;
        if (aux > 0 && lane == 0) {  // a union group: its queries and the rows that carry their partials
            int uq[UNION_CAP], urow[UNION_CAP], un;
            union_group(t0, NB, sOpen[t0] >> 8, ucap, sCnt, sOff, qtab ? sQ : nullptr, block_q, uq, urow, un);
            ul.gn[aux - 1] = un;
            for (int j = 0; j < UNION_CAP; ++j) {
                ul.gq[j * cap + aux - 1] = j < un ? uq[j] : 0;
                ul.grow[j * cap + aux - 1] = j < un ? urow[j] : 0;
            }
        }
        for (int j = lane; j < n; j += 64) {
            const int t = t0 + j * st;
            ul.src[first + j] = t;
            ul.aux[first + j] = aux;
            ul.pass[first + j] = ps;
            ul.flags[first + j] = (first << 1) | (j == 0 ? 1 : 0);
            ul.prow[first + j] = sOff[t];
        }
    }
    if (!np) return;
    lds_barrier();  // (wave 0's rLead / rFoll / sMeta; NOT the unit stores above: nobody in this kernel reads them)
    record_order_write(ul, rt, NR, rLead, rFoll, sMeta);
}

// One workgroup of 128 threads per unit (+ the sentinel): pack its record.
__global__ __launch_bounds__(128) void flatten_records_kernel(const int64_t* block_q, const int64_t* block_q_cnts,
                                                              const int64_t* block_bitmasks, const int64_t* block_kv,
                                                              const int64_t* block_lens, int G, int rows, int64_t q_st,
                                                              int64_t q_sh, int64_t kv_stride_slot, UnitList ul,
                                                              const int32_t* hdr, char* plan, int32_t* row_q,
                                                              const int32_t* cache_loc, int n_new, int64_t new_row_bytes,
                                                              int nt_passes, int NBc, const int32_t* dims, int win_tiles,
                                                              int32_t* win_tab) {
    constexpr int np = 1;
    const int NB = dims ? min(dims[5], NBc) : NBc;  // (device-side metadata: this step's block count, see flatten_units_kernel)
    __shared__ int sKeys[NEWMAP_SIZE], sVals[NEWMAP_SIZE];
    const int r = blockIdx.x;
    const int k = threadIdx.x;
    const int R = hdr[0];
    char* rec = plan + (int64_t)r * PLAN_BYTES;
    int64_t* ro = reinterpret_cast<int64_t*>(rec + PLAN_ROWOFF);
    uint32_t* mk = reinterpret_cast<uint32_t*>(rec + PLAN_MASK);
    int32_t* desc = reinterpret_cast<int32_t*>(rec + PLAN_DESC);
    if (r > R) {  // unused capacity: a tile-parallel workgroup that lands here must see "not a chunk leader"
        if (k == 0) desc[4] = 0;
        return;
    }
    if (r == R) {  // sentinel
        ro[k] = 0;
        mk[k] = 0u;
        if (k == 0) {
            desc[0] = 0;
            desc[1] = 0;
            desc[2] = 1;
            desc[3] = -1;
            desc[4] = 0;
        }
        return;
    }
    const NewMap nm = newmap_build(sKeys, sVals, cache_loc, n_new);
    const int u = np ? ul.perm[r] : r;  // unit packed into this record
    const int t = ul.src[u];
    const int ps = ul.pass[u];
    const int prow = ul.prow[u];
    const int len = (int)block_lens[t];
    const int cnt = (int)block_q_cnts[t];
    const bool live = k < len;
    ro[k] = plan_rowoff(block_kv[(int64_t)t * TILE + (live ? k : 0)], kv_stride_slot, cache_loc, n_new, new_row_bytes, nm);
    const uint32_t qmask = live ? (uint32_t)block_bitmasks[(int64_t)t * TILE + k] : 0u;
    if (ul.aux[u] > 0) {
        // ---- member of a union group: virtual row v = (union query v / G, head v % G); a query that is not in
        //      this block's own list sees none of its slots ------------------------------------------------
        const int gid = ul.aux[u] - 1;
        const int cap = (int)(ul.gq - ul.gn);  // arrays are `cap` apart
        const int un = ul.gn[gid];
        const int nvu = un * G;
        uint32_t vmu = 0u;
        for (int j = 0; j < un; ++j) {
            const int qv = ul.gq[j * cap + gid];
            int idx = -1;
            for (int i = 0; i < cnt; ++i)
                if ((int)block_q[prow + i] == qv) idx = i;
            if (idx >= 0 && ((qmask >> idx) & 1u)) vmu |= ((G >= 32 ? 0xffffffffu : ((1u << G) - 1u)) << (j * G));
        }
        mk[k] = vmu;
        if (k < MQ) {
            const int kk = k < nvu ? k : 0;  // rows beyond the union alias its first row (their mask bits are 0)
            const int j = kk / G, g = kk % G;
            reinterpret_cast<int32_t*>(rec + PLAN_QSRC)[k] = (int)(ul.gq[j * cap + gid] * q_st + g * q_sh);
            reinterpret_cast<int32_t*>(rec + PLAN_OROW)[k] = k < nvu ? g * rows + ul.grow[j * cap + gid] : 0;
            if (k < cnt) {  // this tile's own partial rows: live iff the group parks a query's partial there
                int qlive = -1;
                for (int j2 = 0; j2 < un; ++j2)
                    if (ul.grow[j2 * cap + gid] == prow + k) qlive = ul.gq[j2 * cap + gid];
                row_q[prow + k] = qlive;
            }
        }
        if (k == 0) {
            desc[0] = nvu;
            desc[1] = prow;
            desc[2] = ul.flags[u] & 1;
            desc[3] = ul.flags[u] >> 1;
            desc[4] = ul.ch_n[r];
            desc[5] = ul.ch_fb[r];
            desc[6] = 0;  // one pass over the group's tiles: non-temporal
        }
        return;
    }
    const int nv = min(MQ, cnt * G - MQ * ps);  // virtual rows of this pass
    uint32_t vm = 0u;
    for (int v = 0; v < nv; ++v) vm |= ((qmask >> ((MQ * ps + v) / G)) & 1u) << v;
    mk[k] = vm;
    if (k < MQ) {
        int qs = 0, orow = 0;
        if (k < nv) {
            const int qi = (MQ * ps + k) / G, g = (MQ * ps + k) % G;
            qs = (int)(block_q[prow + qi] * q_st + g * q_sh);
            orow = g * rows + prow + qi;
        } else if (nv > 0) {
            qs = (int)(block_q[prow + (MQ * ps) / G] * q_st + ((MQ * ps) % G) * q_sh);  // alias a real row
        }
        reinterpret_cast<int32_t*>(rec + PLAN_QSRC)[k] = qs;
        reinterpret_cast<int32_t*>(rec + PLAN_OROW)[k] = orow;
        if (!np) {
            if (ps == 0 && k < cnt) row_q[prow + k] = (int32_t)block_q[prow + k];
        } else if (k < nv && (MQ * ps + k) % G == 0) {
            // tile-parallel order: folding is decided here, so only a chunk leader's rows are live
            const int qi = (MQ * ps + k) / G;
            row_q[prow + qi] = ul.ch_n[r] > 0 ? (int32_t)block_q[prow + qi] : -1;
        }
    }
    if (k == 0) {
        desc[0] = nv;
        desc[1] = prow;
        desc[2] = ul.flags[u] & 1;
        desc[3] = ul.flags[u] >> 1;  // run id: tiles with equal ids share one query list and may fold
        desc[4] = np ? ul.ch_n[r] : 0;
        desc[5] = np ? ul.ch_fb[r] : 0;
        // How many 32-row passes fold this tile's rows in the whole launch: its own, plus those of the neighbouring blocks over the
        // same slots (the query chunks of a node of more than max_q_len queries alternate over its tiles).  Many: the later ones
        // want the rows in L2, so the chunk asks for them with the temporal policy (launch_stage1_np).
        int passes = (cnt * G + MQ - 1) / MQ;
        const int64_t slot0 = block_kv[(int64_t)t * TILE];
        for (int o = t - 1; o >= 0 && passes <= nt_passes && block_kv[(int64_t)o * TILE] == slot0; --o)
            passes += ((int)block_q_cnts[o] * G + MQ - 1) / MQ;
        for (int o = t + 1; o < NB && passes <= nt_passes && block_kv[(int64_t)o * TILE] == slot0; ++o)
            passes += ((int)block_q_cnts[o] * G + MQ - 1) / MQ;
        desc[6] = passes > nt_passes;
        // window plan: the leader of an overflow run tells window_patch_kernel where the run's records are -- per (query chunk, pass)
        // {leader record, first follower record} (followers are consecutive: tile j of the run is record first + j - 1)
        if (win_tiles > 0 && dims && t >= dims[WIN_DIM_NB] && np && ul.ch_n[r] > 0 && ps < WIN_PASSES) {
            const int c = (t - dims[WIN_DIM_NB]) / win_tiles;
            win_tab[(c * WIN_PASSES + ps) * 2] = r;
            win_tab[(c * WIN_PASSES + ps) * 2 + 1] = ul.ch_fb[r];
        }
    }
}

// Node mode (tree_attention.py:14-293): every entry (a node's KV x up to 32 of its queries) is cut into
// 128-slot tiles; all live slots are visible to all of the entry's queries.  Consecutive tiles of one
// entry fold, which is the reference's serial online-softmax walk (:230-276) without the serialisation.
// Small entries (one tile, one pass) that follow each other are PACKED into one tile as long as their slots fit
// in 128 and their virtual query rows in 32 -- per-slot row masks make that the same arithmetic (a Medusa step
// has 64 one-token nodes: 2 tiles instead of 64).  A packed unit has aux = -(entries in it).
// Round 5: consecutive ONE-TILE entries over the same query list -- what `--mode node_chunk` (MAX_BLOCK_LEN = 128,
// examples/run_DeFT_llama_paged.py:145-150; tree_cache.py:744-760) makes of every node longer than 128 tokens: a 4096-token
// prompt arrives as 32 entries of 128 slots under one list -- form ONE run (per 32-row pass), so that they fold like the tiles of
// one entry: the north-star tree through node_chunk 47.2 -> 36 us per layer (32 partial rows per query and head became 4).
// Run-table aux = 2: unit j of the run is tile 0 of entry e + j.
// One workgroup.  Wave 0 walks the entries and decides the runs (packs are sequential by nature), all waves then write the units and the record order from the LDS run table -- as in
// flatten_units_kernel; `par` = 0 (tables beyond the LDS) or an overflowing table: lane 0 emits as it walks.
__global__ __launch_bounds__(1024) void node_units_kernel(const int64_t* node_kv_len, const int64_t* node_q_len,
                                                          const int64_t* node_q, const int64_t* node_q_offset, int NEc, int G,
                                                          int cap, int64_t rows_cap, UnitList ul, int32_t* hdr,
                                                          int32_t* row_q, int Hkv, int slots, int chunk_c, int run_cap,
                                                          int par, int keep_err, const int32_t* dims, int win_tiles) {
    constexpr int np = 1;
    const int NE = dims ? min(dims[1], NEc) : NEc;  // (device-side metadata: this step's entry count, see flatten_units_kernel)
    // window plan (window.h): entries NEreg .. NE - 1 are overflow entries, one per query chunk -- each ONE run in ONE chunk, never packed or folded
    const int NEreg = (dims && win_tiles > 0) ? min(dims[WIN_DIM_NB], NE) : NE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* sRun = reinterpret_cast<int*>(smem);
    RunTable rt{sRun, sRun + run_cap, sRun + 2 * run_cap, 0, run_cap};
    int* rT0 = sRun + 3 * run_cap;    // par: the run's entry;               later: its first leader record
    int* rSp = sRun + 4 * run_cap;    // par: the run's pass;                later: its first follower record - leaders
    int* rProw = sRun + 5 * run_cap;  // par: partial row of the run's first tile
    int* rQl = sRun + 6 * run_cap;    // par: partial rows per tile
    int* rAux = sRun + 7 * run_cap;   // par: 1 = tiles of one entry (aux = tile index), <= 0 = a pack (aux = -entries), 2 = one-tile entries
    int* rLead = sRun + 8 * run_cap;  // par: the run's first leader record            (record_order_wave0)
    int* rFoll = sRun + 9 * run_cap;  // par: the run's first follower record - leaders
    int* sMeta = sRun + (par ? 10 : 3) * run_cap;  // [8]
    for (int64_t i = threadIdx.x; i < rows_cap; i += blockDim.x) row_q[i] = -1;
    const int lane = threadIdx.x & 63;
    const int par_req = par;
    __shared__ const int64_t* sListPtr[2];  // node_q, node_q_offset for the walk's rare fold test (written and read by wave 0 only)
    if (threadIdx.x == 0) {
        sListPtr[0] = node_q;
        sListPtr[1] = node_q_offset;
    }
    __builtin_amdgcn_wave_barrier();  // (lane 0's stores in front of the other lanes' reads of wave 0 -- made explicit)
    if (threadIdx.x < 64) {
        for (int attempt = 0; attempt < 2; ++attempt) {
            par = par_req && attempt == 0;
            rt.n = 0;
            int r = 0, rowbase = 0;
            int pack_r = -1, pack_k = -1, pack_n = 0, pack_keys = 0, pack_rows = 0;  // open pack: unit, run, entries, slots, virtual rows
            auto emit_run = [&](int e, int n, int ps, int prow0, int ql, int aux, int uni = 0) {
                if (n > cap - r) n = cap - r;
                if (n <= 0) return;
                const int first = r;
                if (par) {
                    if (lane == 0 && rt.n < rt.cap) {
                        rt.r0[rt.n] = first;
                        rt.nt[rt.n] = n;
                        rt.uni[rt.n] = uni;
                        rT0[rt.n] = e;
                        rSp[rt.n] = ps;
                        rProw[rt.n] = prow0;
                        rQl[rt.n] = ql;
                        rAux[rt.n] = aux;
                    }
                } else if (lane == 0) {
                    for (int j = 0; j < n; ++j) {
                        ul.src[first + j] = aux == 2 ? e + j : e;
                        ul.aux[first + j] = aux == 2 ? 0 : (aux > 0 ? j : aux);
                        ul.pass[first + j] = ps | (uni ? 1 << 16 : 0);  // (bit 16: an overflow run of a window plan -- one chunk)
                        ul.flags[first + j] = (first << 1) | (j == 0 ? 1 : 0);
                        ul.prow[first + j] = prow0 + j * ql;
                    }
                    if (rt.n < rt.cap) {
                        rt.r0[rt.n] = first;
                        rt.nt[rt.n] = n;
                        rt.uni[rt.n] = uni;
                    }
                }
                ++rt.n;
                r += n;
            };
            // 64 entries at a time: lane i loads the lengths of entry base + i (one round trip per batch), the walk
            // reads them with v_readlane -- scalar code, no memory access per entry (a Medusa step has 65 entries)
            for (int base = 0; base < NE; base += 64) {
              const int mine = base + lane;
              const int vlen = mine < NE ? (int)node_kv_len[mine] : 0;
              const int vql = mine < NE ? (int)node_q_len[mine] : 0;
              const int lim = min(64, NE - base);
              // lane i: does entry base + i continue a run of one-tile entries -- the entry before it (same batch) is a FULL tile,
              // this one is one tile, and both have the same query list?  Nothing is read, shuffled or counted unless some entry of
              // the batch is exactly one full tile (uniform test: ordinary Node metadata -- whole nodes -- skips all of it; computed
              // per entry inside the walk this cost a Medusa step's plan 3-5 us).  vmore: entries that continue lane i's run.
              // (A run is detected INSIDE a batch of 64 entries: lane 0 of a batch never continues the previous batch's run, so a
              //  node_chunk run that straddles a batch boundary is cut there -- two runs, two partial rows per query instead of one:
              //  it costs folding, not correctness.  ADVICE r5.)
              int vmore = 0;
              if (__ballot(mine < NE && vlen == TILE) != 0ull) {
                  bool cont = false;
                  const int plen = __shfl_up(vlen, 1, 64), pql = __shfl_up(vql, 1, 64);
                  // (the two list pointers come back from the LDS: as kernel arguments they would stay in SGPRs through the whole
                  //  walk -- 15 more spilled SGPRs, +1.5 ... +3.5 us on every Node plan that never takes this branch)
                  const int64_t* nq = sListPtr[0];
                  const int64_t* nqo = sListPtr[1];
                  if (lane >= 1 && mine < NEreg && plen == TILE && vlen >= 1 && vlen <= TILE && vql == pql && vql > 0 && nq) {
                      const int64_t a = nqo[mine], b = nqo[mine - 1];
                      cont = true;
                      for (int t = 0; t < vql; ++t) cont &= nq[a + t] == nq[b + t];
                  }
                  const unsigned long long contm = __ballot(cont);
                  const unsigned long long after = lane < 63 ? ~(contm >> (lane + 1)) : ~0ull;
                  vmore = (contm && vlen >= 1 && vlen <= TILE) ? (after ? __builtin_ctzll(after) : 63 - lane) : 0;
              }
              for (int i = 0; i < lim; ++i) {
                const int e = base + i;
                const int len = __builtin_amdgcn_readlane(vlen, i);
                const int nt = (len + TILE - 1) / TILE;
                const int ql = __builtin_amdgcn_readlane(vql, i);
                const int npass = (ql * G + MQ - 1) / MQ;
                // entries e + 1 .. e + more continue e's run: `1 + more` one-tile entries emitted like the tiles of one entry (no third
                // emit site: on a lone workgroup every cold instruction line is a dependent round trip, and the ordinary walk
                // must not pay for this case)
                const int more = __builtin_amdgcn_readlane(vmore, i);
                const int nrun = more > 0 ? 1 + more : nt;
                if (e >= NEreg) {  // an overflow entry of a window plan: its tiles are one run per pass, each kept in one chunk
                    pack_r = -1;
                    for (int ps = 0; ps < npass; ++ps) emit_run(e, nt, ps, rowbase, ql, 1, -1);
                    rowbase += nt * ql;
                    continue;
                }
                if (more == 0 && nt == 1 && npass == 1 && r < cap) {
                    if (pack_r >= 0 && pack_n < MQ && pack_keys + len <= TILE && pack_rows + ql * G <= MQ) {
                        ++pack_n;
                        pack_keys += len;
                        pack_rows += ql * G;
                        if (lane == 0) {
                            if (!par) ul.aux[pack_r] = -pack_n;
                            else if (pack_k < rt.cap) rAux[pack_k] = -pack_n;
                        }
                    } else {
                        pack_r = r;
                        pack_k = rt.n;
                        pack_n = 1;
                        pack_keys = len;
                        pack_rows = ql * G;
                        emit_run(e, 1, 0, rowbase, 0, 0);  // a pack of one is an ordinary unit
                    }
                } else {
                    pack_r = -1;
                    for (int ps = 0; ps < npass; ++ps) emit_run(e, nrun, ps, rowbase, ql, more > 0 ? 2 : 1);
                }
                rowbase += nrun * ql;
                i += more;
              }
            }
            if (lane == 0) {
                hdr[0] = r;
                hdr[1] = 0;
                if (!keep_err) hdr[HDR_ERR] = 0;  // (the sequential plan's slot-list kernel has already run and may have flagged)
                hdr[HDR_QLISTS] = 0;
                if (par && rt.n <= rt.cap) {
                    sMeta[0] = r;
                    sMeta[1] = rt.n;
                    sMeta[4] = 1;
                } else if (!par) {
                    sMeta[4] = 0;
                    if (np) np_record_order(ul, r, Hkv, G, slots, chunk_c, hdr, rt);
                }
            }
            if (!par || rt.n <= rt.cap) break;
        }
    }
    __syncthreads();
    if (!sMeta[4]) return;
    const int NR = sMeta[1];
    const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    // (flatten_units_kernel: wave 0 decides the record order while the other waves write the units)
    if (np && wave == 0) record_order_wave0(rt, NR, rLead, rFoll, sMeta, hdr, Hkv, G, slots, chunk_c);
    for (int k = np ? wave - 1 : wave; k < NR && k >= 0; k += np ? nwaves - 1 : nwaves) {
        const int first = rt.r0[k], n = rt.nt[k], e = rT0[k], ps = rSp[k], prow0 = rProw[k], ql = rQl[k], aux = rAux[k];
        const int ovf = rt.uni[k] ? 1 << 16 : 0;
        for (int j = lane; j < n; j += 64) {
            ul.src[first + j] = aux == 2 ? e + j : e;
            ul.aux[first + j] = aux == 2 ? 0 : (aux > 0 ? j : aux);
            ul.pass[first + j] = ps | ovf;
            ul.flags[first + j] = (first << 1) | (j == 0 ? 1 : 0);
            ul.prow[first + j] = prow0 + j * ql;
        }
    }
    if (!np) return;
    lds_barrier();
    record_order_write(ul, rt, NR, rLead, rFoll, sMeta);
}

__global__ __launch_bounds__(128) void node_records_kernel(const int64_t* node_kv, const int64_t* node_kv_offset,
                                                           const int64_t* node_kv_len, const int64_t* node_q,
                                                           const int64_t* node_q_offset, const int64_t* node_q_len, int G,
                                                           int rows, int64_t q_st, int64_t q_sh, int64_t kv_stride_slot,
                                                           UnitList ul, const int32_t* hdr, char* plan, int32_t* row_q,
                                                           const int32_t* cache_loc, int n_new, int64_t new_row_bytes,
                                                           int nt_passes, int NEc, const int32_t* dims, int win_tiles,
                                                           int32_t* win_tab) {
    const int NE = dims ? min(dims[1], NEc) : NEc;
    const int NEreg = (dims && win_tiles > 0) ? min(dims[WIN_DIM_NB], NE) : NE;  // window plan: entries from here on are overflow entries
    __shared__ int sPasses;
    constexpr int np = 1;
    __shared__ int sKeys[NEWMAP_SIZE], sVals[NEWMAP_SIZE];
    const int r = blockIdx.x;
    const int k = threadIdx.x;
    const int R = hdr[0];
    char* rec = plan + (int64_t)r * PLAN_BYTES;
    int64_t* ro = reinterpret_cast<int64_t*>(rec + PLAN_ROWOFF);
    uint32_t* mk = reinterpret_cast<uint32_t*>(rec + PLAN_MASK);
    int32_t* desc = reinterpret_cast<int32_t*>(rec + PLAN_DESC);
    if (r > R) {  // unused capacity (see flatten_records_kernel)
        if (k == 0) desc[4] = 0;
        return;
    }
    if (r == R) {  // sentinel
        ro[k] = 0;
        mk[k] = 0u;
        if (k == 0) {
            desc[0] = 0;
            desc[1] = 0;
            desc[2] = 1;
            desc[3] = -1;
            desc[4] = 0;
        }
        return;
    }
    const NewMap nm = newmap_build(sKeys, sVals, cache_loc, n_new);
    const int u = np ? ul.perm[r] : r;  // unit packed into this record
    if (k == 0) {
        desc[4] = np ? ul.ch_n[r] : 0;
        desc[5] = np ? ul.ch_fb[r] : 0;
    }
    const int e0 = ul.src[u], aux = ul.aux[u], ps = ul.pass[u] & 0xffff, prow = ul.prow[u];  // (bit 16 of pass: node_units_kernel)
    if (aux < 0) {
        // ---- packed unit: entries e0 .. e0 - aux - 1, each one tile and one pass -------------------------
        const int cnt = -aux;
        // slot k -> (entry j, position inside it); virtual row k -> (entry j, query, head of the group)
        int kj = -1, kpos = 0, kvb = 0, knv = 0;  // for the slot role
        int vj = -1, vloc = 0, vrowbase = 0;      // for the row role (k < MQ)
        int keys = 0, vrows = 0, rowsum = 0;
        int first_slot_e = e0;
        for (int j = 0; j < cnt; ++j) {
            const int e = e0 + j;
            const int len = (int)node_kv_len[e];
            const int nv = (int)node_q_len[e] * G;
            if (kj < 0 && k < keys + len) {
                kj = e;
                kpos = k - keys;
                kvb = vrows;
                knv = nv;
            }
            if (vj < 0 && k < vrows + nv) {
                vj = e;
                vloc = k - vrows;
                vrowbase = rowsum;
            }
            keys += len;
            vrows += nv;
            rowsum += (int)node_q_len[e];
        }
        const bool live = kj >= 0;
        const int64_t slot = live ? node_kv[node_kv_offset[kj] + kpos] : node_kv[node_kv_offset[first_slot_e]];
        ro[k] = plan_rowoff(slot, kv_stride_slot, cache_loc, n_new, new_row_bytes, nm);
        mk[k] = live ? ((knv >= 32 ? 0xffffffffu : ((1u << knv) - 1u)) << kvb) : 0u;
        if (k < MQ) {
            int qs, orow = 0;
            if (vj >= 0) {
                const int qi = vloc / G, g = vloc % G;
                const int64_t qrow = node_q[node_q_offset[vj] + qi];
                qs = (int)(qrow * q_st + g * q_sh);
                orow = g * rows + prow + vrowbase + qi;
                if (g == 0) row_q[prow + vrowbase + qi] = (int32_t)qrow;
            } else {
                qs = (int)(node_q[node_q_offset[e0]] * q_st);  // alias a real row
            }
            reinterpret_cast<int32_t*>(rec + PLAN_QSRC)[k] = qs;
            reinterpret_cast<int32_t*>(rec + PLAN_OROW)[k] = orow;
        }
        if (k == 0) {
            desc[0] = vrows;
            desc[1] = prow;
            desc[2] = 1;
            desc[3] = ul.flags[u] >> 1;
            desc[6] = 0;
        }
        return;
    }
    const int e = e0, tt = aux;
    const int64_t kv0 = node_kv_offset[e] + (int64_t)tt * TILE;
    {   // How many 32-row passes fold this node's rows in the whole launch (see flatten_records_kernel): the entries of one node --
        // up to 32 of its queries over one stretch of its slots each (tree_cache.py:744-760) -- repeat the slots; thread k looks at
        // entry e - 64 + k.
        if (k == 0) sPasses = 0;
        __syncthreads();
        const int o = e - 64 + k;
        if (o >= 0 && o < NE && node_kv_len[o] == node_kv_len[e] && node_kv[node_kv_offset[o]] == node_kv[node_kv_offset[e]])
            atomicAdd(&sPasses, ((int)node_q_len[o] * G + MQ - 1) / MQ);
        __syncthreads();
    }
    const int len = (int)min((int64_t)TILE, node_kv_len[e] - (int64_t)tt * TILE);
    const int64_t q0 = node_q_offset[e];
    const int ql = (int)node_q_len[e];
    const int nv = min(MQ, ql * G - MQ * ps);
    const bool live = k < len;
    ro[k] = plan_rowoff(node_kv[kv0 + (live ? k : 0)], kv_stride_slot, cache_loc, n_new, new_row_bytes, nm);
    // (an overflow entry's slots start masked: window_patch_kernel gives every token it places there its own rows)
    mk[k] = (live && e < NEreg) ? (nv >= 32 ? 0xffffffffu : ((1u << nv) - 1u)) : 0u;
    if (k == 0 && e >= NEreg && tt == 0 && np && ul.ch_n[r] > 0 && ps < WIN_PASSES) {
        win_tab[((e - NEreg) * WIN_PASSES + ps) * 2] = r;
        win_tab[((e - NEreg) * WIN_PASSES + ps) * 2 + 1] = ul.ch_fb[r];
    }
    if (k < MQ) {
        int qs = 0, orow = 0;
        if (k < nv) {
            const int qi = (MQ * ps + k) / G, g = (MQ * ps + k) % G;
            qs = (int)(node_q[q0 + qi] * q_st + g * q_sh);
            orow = g * rows + prow + qi;
        } else if (nv > 0) {
            qs = (int)(node_q[q0 + (MQ * ps) / G] * q_st + ((MQ * ps) % G) * q_sh);
        }
        reinterpret_cast<int32_t*>(rec + PLAN_QSRC)[k] = qs;
        reinterpret_cast<int32_t*>(rec + PLAN_OROW)[k] = orow;
        if (!np) {
            if (ps == 0 && k < ql) row_q[prow + k] = (int32_t)node_q[q0 + k];
        } else if (k < nv && (MQ * ps + k) % G == 0) {
            const int qi = (MQ * ps + k) / G;
            row_q[prow + qi] = ul.ch_n[r] > 0 ? (int32_t)node_q[q0 + qi] : -1;
        }
    }
    if (k == 0) {
        desc[0] = nv;
        desc[1] = prow;
        desc[2] = ul.flags[u] & 1;
        desc[3] = ul.flags[u] >> 1;
        desc[6] = sPasses > nt_passes;
    }
}

}  // namespace deft

namespace deft {

// ---------------------------------------------------------------------------
// Per-query row lists: which partial rows the merge of query q reads, ascending.  row_q (partial row -> query, -1 =
// dead) is written record by record; every layer's merge would otherwise scan it again (two to six dependent round
// trips per launch, 32 times per step).  Two small kernels once per step: a histogram + exclusive scan (one workgroup),
// then one wave per query collecting its rows in order (ordered ballots, no atomics: the lists -- and the order of
// the merge's fp32 additions -- are a function of the plan alone).
// ---------------------------------------------------------------------------
constexpr int QROWS_MAX = 15 * 1024;  // rows whose histogram fits the LDS; larger workspaces keep the scanning merge

__global__ __launch_bounds__(1024) void qrows_hist_kernel(const int32_t* row_q, int rows, int32_t* qoff, int32_t* hdr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* sCnt = reinterpret_cast<int*>(smem);  // [rows] then scanned in place
    __shared__ int sWave[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < rows; i += 1024) sCnt[i] = 0;
    __syncthreads();
    for (int i = tid; i < rows; i += 1024) {
        const int q = row_q[i];
        if (q >= 0 && q < rows) atomicAdd(&sCnt[q], 1);  // LDS integer adds: the counts do not depend on the order
    }
    __syncthreads();
    // exclusive scan of sCnt[0 .. rows): thread t owns the consecutive span [t * per, (t + 1) * per)
    const int per = (rows + 1023) / 1024;
    const int lo = tid * per, hi = min(rows, lo + per);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += sCnt[i];
    int inc = sum;
    for (int d = 1; d < 64; d <<= 1) {
        const int u = __shfl_up(inc, d, 64);
        if (lane >= d) inc += u;
    }
    if (lane == 63) sWave[wave] = inc;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < wave; ++k) base += sWave[k];
    int run = base + inc - sum;
    for (int i = lo; i < hi; ++i) {
        const int c = sCnt[i];
        qoff[i] = run;
        run += c;
    }
    if (tid == 1023) {
        qoff[rows] = run;
        hdr[HDR_QLISTS] = 1;
    }
}

// one wave per query (four per workgroup): the query's rows, ascending, into qlist[qoff[q] ..)
__global__ __launch_bounds__(256) void qrows_fill_kernel(const int32_t* row_q, int rows, const int32_t* qoff, int32_t* qlist,
                                                         int32_t* qinl) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= rows) return;
    const int o = qoff[q], n = qoff[q + 1] - o;
    if (lane == 0) qinl[q * 16] = n;  // {count, first 15 rows}: what the merge of the usual query needs, in one 64-byte line
    if (n <= 0) return;  // (not a query of this step)
    int found = 0;
    for (int base = 0; base < rows && found < n; base += 512) {
        int val[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + 64 * u + lane;
            val[u] = (i < rows) ? row_q[i] : -1;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool hit = val[u] == q;
            const unsigned long long mask = __ballot(hit);
            if (hit) {
                const int pos = found + __popcll(mask & ((1ull << lane) - 1ull));
                qlist[o + pos] = base + 64 * u + lane;
                if (pos < 15) qinl[q * 16 + 1 + pos] = base + 64 * u + lane;
            }
            found += __popcll(mask);
        }
    }
}

// Both steps in ONE workgroup for the usual row counts (<= QROWS_FUSED_MAX partial rows: every single tree of BASELINE.json; a
// launch boundary alone costs ~2.5 us, and the row lists are one of six launches in front of every decode step's first layer):
// row_q is staged in LDS once, counted, scanned, and every wave then fills the lists of the queries q = wave, wave + 16, ...
// up to the largest query that has a row.  Same qoff / qlist / qinl as the two kernels above, word for word.
constexpr int QROWS_FUSED_MAX = 4096;
__global__ __launch_bounds__(1024) void qrows_fused_kernel(const int32_t* row_q, int rows, int32_t* qoff, int32_t* qlist, int32_t* qinl,
                                                           int32_t* hdr) {
    __shared__ int sRow[QROWS_FUSED_MAX], sCnt[QROWS_FUSED_MAX + 1];
    __shared__ int sWave[16], sQmax;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) sQmax = -1;
    for (int i = tid; i < rows; i += 1024) sCnt[i] = 0;
    __syncthreads();
    int my_max = -1;
    for (int i = tid; i < rows; i += 1024) {
        const int q = row_q[i];
        sRow[i] = q;
        if (q >= 0 && q < rows) {
            atomicAdd(&sCnt[q], 1);  // LDS integer adds: the counts do not depend on the order
            my_max = max(my_max, q);
        }
    }
    for (int sft = 32; sft > 0; sft >>= 1) my_max = max(my_max, __shfl_xor(my_max, sft));
    if (lane == 0 && my_max >= 0) atomicMax(&sQmax, my_max);  // (one atomic per wave: 1024 of them on one word serialise)
    __syncthreads();
    // exclusive scan of sCnt[0 .. rows) in place (qrows_hist_kernel's), offsets also to qoff
    const int per = (rows + 1023) / 1024;
    const int lo = tid * per, hi = min(rows, lo + per);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += sCnt[i];
    int inc = sum;
    for (int d = 1; d < 64; d <<= 1) {
        const int u = __shfl_up(inc, d, 64);
        if (lane >= d) inc += u;
    }
    if (lane == 63) sWave[wave] = inc;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < wave; ++k) base += sWave[k];
    int run = base + inc - sum;
    for (int i = lo; i < hi; ++i) {
        const int c = sCnt[i];
        sCnt[i] = run;
        qoff[i] = run;
        qinl[i * 16] = c;  // {count, first 15 rows} of query i (qrows_fill_kernel)
        run += c;
    }
    if (tid == 1023) {
        sCnt[rows] = run;
        qoff[rows] = run;
        hdr[HDR_QLISTS] = 1;
    }
    __syncthreads();
    const int qmax = sQmax;
    for (int q = wave; q <= qmax; q += 16) {
        const int o = sCnt[q], n = sCnt[q + 1] - o;
        if (n <= 0) continue;  // (wave-uniform)
        int found = 0;
        for (int b0 = 0; b0 < rows && found < n; b0 += 64) {
            const int i = b0 + lane;
            const bool hit = i < rows && sRow[i] == q;
            const unsigned long long mask = __ballot(hit);
            if (hit) {
                const int pos = found + __popcll(mask & ((1ull << lane) - 1ull));
                qlist[o + pos] = i;
                if (pos < 15) qinl[q * 16 + 1 + pos] = i;
            }
            found += __popcll(mask);
        }
    }
}

}  // namespace deft
