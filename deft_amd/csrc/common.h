// Shared declarations of libdeft_amd.so (gfx950 only).
#pragma once

#include <stdarg.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/deft_amd.h"

namespace deft {

// thread-local last-error text behind deft_last_error()
void set_error(const char* fmt, ...);
const char* get_error();

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Workspace carve shared by the Flatten and Node entry points:
//   partial_o   [Hq][rows][D] f32   normalised stage-1 outputs
//   partial_lse [Hq][rows]    f32   log-sum-exp of each partial row (natural log)
//   row_q       [rows]        i32   query row of each partial row, -1 = unused
//   desc        [tiles][8]    i32   tile descriptors (Node mode only)
//   plan        [NB+1][2048]  bytes Flatten plan records: row byte offsets, 32-bit masks, {cnt, prow,
//                                   run_start, len} per block (stage1_stream.h)
struct Workspace {
    float* partial_o;
    float* partial_lse;
    int32_t* row_q;
    int32_t* desc;
    char* plan;          // [NB+1][2048] Flatten streaming plan records (stage1_stream.h)
    size_t bytes;
};

inline Workspace carve(void* base, int Hq, int D, int64_t rows, int64_t tiles, int64_t NB = 0) {
    Workspace w;
    char* p = static_cast<char*>(base);
    size_t off = 0;
    w.partial_o = reinterpret_cast<float*>(p + off);
    off = align_up(off + sizeof(float) * (size_t)Hq * (size_t)rows * (size_t)D, 256);
    w.partial_lse = reinterpret_cast<float*>(p + off);
    off = align_up(off + sizeof(float) * (size_t)Hq * (size_t)rows, 256);
    w.row_q = reinterpret_cast<int32_t*>(p + off);
    off = align_up(off + sizeof(int32_t) * (size_t)rows, 256);
    w.desc = reinterpret_cast<int32_t*>(p + off);
    off = align_up(off + sizeof(int32_t) * 8 * (size_t)tiles, 256);
    w.plan = p + off;  // a whole PlanView when the caller passes no plan
    off = align_up(off + 2048 * (size_t)(NB + 1) + 768 + sizeof(int32_t) * (size_t)(rows > 0 ? rows : 1), 256);
    w.bytes = off;
    return w;
}

inline int64_t node_max_tiles(int NE, int64_t total_kv) { return (int64_t)NE + total_kv / DEFT_BLOCK_LEN; }

// Flatten plan buffer: [NB+1] records of 2048 B, row_q[P] (i32), then the stream scheduler's two
// words {ticket counter, workgroups done} (zero between launches)
struct PlanView {
    char* records;
    int32_t* row_q;
    int32_t* sched;
    size_t bytes;
};
inline PlanView plan_view(void* base, int64_t NB, int64_t P) {
    PlanView v;
    char* p = static_cast<char*>(base);
    v.records = p;
    size_t off = align_up(2048 * (size_t)(NB + 1), 256);
    v.row_q = reinterpret_cast<int32_t*>(p + off);
    off = align_up(off + sizeof(int32_t) * (size_t)(P > 0 ? P : 1), 256);
    v.sched = reinterpret_cast<int32_t*>(p + off);
    off += 256;
    v.bytes = off;
    return v;
}

}  // namespace deft
