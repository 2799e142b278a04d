// Per-step plan: record layout, header words and the LDS-DMA / wait helpers shared by the plan kernels
// (plan_kernels.h, tree_plan.h) and the stage-1 kernel (stage1_np.h).
//
// Included by deft_kernels.hip (needs its typedefs and Stage1Params).
#pragma once

namespace deft {

typedef int32_t intx4 __attribute__((ext_vector_type(4)));
typedef short short4v __attribute__((ext_vector_type(4)));

// One plan record per work unit of ONE KV head (+ a sentinel record after the last one).  A unit is a
// 128-slot KV tile together with up to 32 "virtual query rows": row v = (query qi, head g of the GQA
// group), g fastest.  A tile whose cnt * G rows exceed 32 appears once per 32-row pass; passes of a run
// of tiles with one query list are ordered pass-major so that consecutive records fold.
constexpr int PLAN_BYTES = 2048;
constexpr int PLAN_ROWOFF = 0;   // int64[128]  byte offset of each slot's row in the pool (pads alias slot 0)
constexpr int PLAN_MASK = 1024;  // uint32[128] bit v set <=> virtual row v sees the slot (0 for pads)
constexpr int PLAN_DESC = 1536;  // int32[8]    n_vrows, prow, opens_run, run_id, chunk tiles (0 = follower), first follower record,
                                 //             temporal (the tile is folded by so many passes that its rows should stay in L2), -
constexpr int PLAN_QSRC = 1600;  // int32[32]   element offset of row v's Q vector from q + kvh*G*q_stride_head
constexpr int PLAN_OROW = 1728;  // int32[32]   partial row of row v, relative to kvh*G*rows: g*rows + prow + qi

// Plan header (4 KB in front of the records), int32 words:
//   hdr[0]  R   records (units) per KV head          hdr[1]  NL  chunk leaders (work items per KV head)
//   hdr[2]  error flags raised by the plan kernels (bit 1: sequential plan overflow)
//   hdr[3]  1 = the per-query row lists (qoff / qlist) are valid; 0 = the merge scans row_q itself
constexpr int PLAN_HDR = 4096;
constexpr int HDR_ERR = 2;
constexpr int HDR_QLISTS = 3;

// Window plans (window.h): per query chunk and 32-row pass, where the overflow run's records are
constexpr int WIN_PASSES = 16;   // 32-row passes per query chunk the table has room for (max_q_len * G / 32)
constexpr int WIN_DIM_NB = 10;   // dims[10]: blocks (Flatten) / entries (Node) in front of the overflow ones
constexpr int WIN_DIM_P = 11;    // dims[11]: block_q / node_q elements in front of the overflow lists

// 64 lanes x 16 bytes, global (per-lane address) -> LDS (lds_dst + 16*lane).  M0 is not
// otherwise used by the kernels (checked in the .s), so it is written, not saved.
__device__ __forceinline__ void dma16(const void* gsrc, uint32_t lds_dst) {
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, off"
        :
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}

// The same with the non-temporal cache policy: a row that exactly one workgroup of the launch reads (MHA decode) need not
// displace anything in L2 / Infinity Cache on its way in
__device__ __forceinline__ void dma16nt(const void* gsrc, uint32_t lds_dst) {
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, off nt"
        :
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}

// The same with a scalar base and a 32-bit per-lane byte offset (saddr form: no 64-bit address per lane)
__device__ __forceinline__ void dma16s(const void* sbase, uint32_t voff, uint32_t lds_dst) {
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %1"
        :
        : "v"(voff), "s"(sbase), "s"(lds_dst)
        : "memory");
}

// 64 lanes x 4 bytes, global (per-lane address) -> LDS (lds_dst + 4*lane)
__device__ __forceinline__ void dma4(const void* gsrc, uint32_t lds_dst) {
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dword %0, off"
        :
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}

// All DMA issue and all waits on it are inline asm: hipcc neither counts asm VMEM operations nor drains them at a
// raw s_barrier, which is what lets loads stay in flight across barriers (cdna_hip_programming.md section 5.7).
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Buffer resource over a float array: the merge addresses partial rows as descriptor (per head, in SGPRs) + 32-bit offset.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}

}  // namespace deft
