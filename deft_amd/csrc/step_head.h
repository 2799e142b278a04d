// The single-workgroup part of a DeFT-Flatten decode step in ONE launch (round 4).
//
// A captured step used to run five dependent launches in front of its first attention layer -- tree_md_scan (1 workgroup),
// tree_md_blocks (a workgroup per block), flatten_units (1), flatten_records (a workgroup per record), qrows_fused (1): 55-66 us,
// of which ~35 us are the three single-workgroup kernels and ~2.5 us each of the four boundaries between dependent launches.
// The three single-workgroup kernels do not need the two wide ones in between:
//   * the unit phase reads the blocks' QUERY LISTS only (block_q, block_q_cnts, block_q_offset) -- which the scan can write itself
//     from the per-block leaf sets it has just OR-ed together (tree_md_scan_body's tail);
//   * the merge's per-query row lists need row_q only -- which follows from the unit list (flatten_rowq_body), not from the records.
// So: this kernel = scan -> query lists -> units + record order -> row_q -> row lists, one workgroup of 1024 threads, the phases
// separated by a fence and a barrier, the dynamic LDS reused from phase to phase; then tree_md_blocks_kernel (slots + masks) and
// flatten_records_kernel as before.  Three launches instead of five, and the plan / TreeMetadata bytes are the same ones
// (tests/test_session.py::test_fused_step_head_builds_the_same_plan).
//
// Included by deft_kernels.hip after tree_plan.h and plan_kernels.h.
#pragma once

namespace deft {

struct StepHeadArgs {
    // scan
    int max_q_len, block_len, max_block_len, nbp_cap;
    const int32_t* cache_loc;  // this step's slots (nullable: the image already holds them)
    const int32_t* ops;        // journal {n, words ...} (nullable)
    PageWrite pw;
    // units
    int NBc, G, cap, Hkv_items, slots, chunk_c, union_len, run_cap, qtab, par, rows;
};

__global__ __launch_bounds__(1024) void flatten_step_head_kernel(TreeDev t, TreeScratch s, TreeMdOut o, UnitList ul, int32_t* hdr,
                                                                 int32_t* row_q, int32_t* qoff, int32_t* qlist, int32_t* qinl,
                                                                 StepHeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef DEFT_EXPERIMENTS
    // (experiments build: the phases' end times, 10 ns ticks from the kernel's start, in dims[10 .. 14] -- tools/step_head_phases.py)
    const unsigned long long t0_ = wall_clock64();
#define HEAD_STAMP(k) \
    if (threadIdx.x == 0) s.dims[10 + (k)] = (int32_t)(wall_clock64() - t0_);
#else
#define HEAD_STAMP(k)
#endif
    // ---- phase 1: journal, page table, leaf append, scans, per-block leaf sets; the emitted blocks' query lists ----------------
    tree_md_scan_body(t, s, a.max_q_len, a.block_len, a.max_block_len, a.nbp_cap, a.cache_loc, a.ops, a.pw, smem, &o);
    __threadfence();
    __syncthreads();
    HEAD_STAMP(0);
    if (s.dims[TREE_ERR]) {  // (uniform) the scan gave up -- out of room, too many blocks: an empty plan, the flags say why
        if (threadIdx.x == 0) {
            hdr[0] = 0;
            hdr[1] = 0;
            hdr[HDR_ERR] = 0;
            hdr[HDR_QLISTS] = 0;
        }
        return;
    }
    // ---- phase 2: units, union groups, record order (hdr[0], hdr[1], the unit arrays) ------------------------------------------
    flatten_units_body(o.block_q, o.block_q_cnts, o.block_q_offset, a.NBc, a.G, a.cap, ul, hdr, a.Hkv_items, a.slots, a.chunk_c,
                       a.union_len, a.run_cap, a.qtab, a.par, s.dims, row_q, a.rows, smem);
    __threadfence();
    __syncthreads();
    HEAD_STAMP(1);
    // ---- phase 3: row_q from the units, then the merge's per-query row lists --------------------------------------------------
    flatten_rowq_body(o.block_q, o.block_q_cnts, a.G, ul, hdr, row_q);
    __threadfence();
    __syncthreads();
    HEAD_STAMP(2);
    int* sRow = reinterpret_cast<int*>(smem);
    qrows_fused_body(row_q, a.rows, qoff, qlist, qinl, hdr, sRow, sRow + QROWS_FUSED_MAX);
    __syncthreads();
    HEAD_STAMP(3);
}
#undef HEAD_STAMP

}  // namespace deft
