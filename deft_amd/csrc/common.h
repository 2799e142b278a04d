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

// Queries a union group of leaf tiles may hold (plan_kernels.h; sizes the unit list of a plan buffer).  A multiple of four.
// (Round 5 ran it at 8 to let groups of 5-8 leaf tiles form -- would a launch whose items all fit the resident slots at once do
//  better?  No: the north-star tree 35.9 us per layer with groups of 3 tiles, 37.3 with 5, 38.1 with 6, 41.6 with 8
//  (profiles/r5_union_groups_longer_negative.txt) -- a longer item is a longer serial chain at the end of the launch.)
#define DEFT_UNION_CAP 4

// Workspace carve shared by the Flatten and Node entry points:
//   partial_o   [Hq][rows][D] f32   normalised stage-1 outputs
//   partial_lse [Hq][rows]    f32   log-sum-exp of each partial row (natural log)
//   row_q       [rows]        i32   query row of each partial row, -1 = unused
//   desc        [tiles][8]    i32   tile descriptors (Node mode only)
//   plan        [NB+1][2048]  bytes Flatten plan records: row byte offsets, 32-bit masks, {cnt, prow,
//                                   run_start, len} per block (plan_records.h)
struct Workspace {
    float* partial_o;
    float* partial_lse;
    int32_t* row_q;
    int32_t* desc;
    char* plan;          // a whole plan buffer when the caller passes no plan
    size_t bytes;
};

inline Workspace carve(void* base, int Hq, int D, int64_t rows, int64_t tiles, size_t plan_bytes = 0) {
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
    off = align_up(off + sizeof(int32_t) * 17 * (size_t)tiles, 256);
    w.plan = p + off;  // a whole plan buffer when the caller passes no plan
    off = align_up(off + plan_bytes, 256);
    w.bytes = off;
    return w;
}

inline int64_t node_max_tiles(int NE, int64_t total_kv) { return (int64_t)NE + total_kv / DEFT_BLOCK_LEN; }

// Plan buffer (built once per decode step, read by every layer's call):
//   header      4 KB  : int32 hdr[0] = records per KV head, hdr[1] = chunk leaders, hdr[2] = error flags,
//                       hdr[3] = 1 when the per-query row lists below are valid (plan_records.h)
//   records     (cap+1) x 2048 B (plan_records.h PLAN_*), cap = units-per-head capacity
//   unit list   (9 + 2 DEFT_UNION_CAP) x cap int32 (src, aux, pass, flags, prow; tile-parallel order: perm, chunk tiles, first follower;
//               union groups: n, DEFT_UNION_CAP queries, DEFT_UNION_CAP rows)
//   row_q       rows int32 : partial row -> query row (-1 = dead row)
//   qoff, qlist rows + 1, rows int32 : per query, its live partial rows in ascending order (what the merge reads)
struct PlanView {
    int32_t* hdr;
    char* records;
    int32_t* units;  // 17 arrays of `cap`
    int32_t* row_q;
    int32_t* qoff;   // [rows + 1] first entry of every query's row list (queries are < rows: each has a partial row)
    int32_t* qlist;  // [rows] live partial rows grouped by query, ascending within a query
    int32_t* qinl;   // [rows][16] per query one 64-byte line: {count, first 15 rows} -- count and rows in ONE load for the usual query
    int64_t cap;
    int64_t rows;
    size_t bytes;
};
inline PlanView plan_view(void* base, int64_t cap, int64_t rows) {
    PlanView v;
    char* p = static_cast<char*>(base);
    v.cap = cap;
    v.rows = rows;
    v.hdr = reinterpret_cast<int32_t*>(p);
    size_t off = 4096;
    v.records = p + off;
    off = align_up(off + 2048 * (size_t)(cap + 1), 256);
    v.units = reinterpret_cast<int32_t*>(p + off);
    off = align_up(off + sizeof(int32_t) * (9 + 2 * DEFT_UNION_CAP) * (size_t)(cap > 0 ? cap : 1), 256);
    v.row_q = reinterpret_cast<int32_t*>(p + off);
    off = align_up(off + sizeof(int32_t) * (size_t)(rows > 0 ? rows : 1), 256);
    v.qoff = reinterpret_cast<int32_t*>(p + off);
    off = align_up(off + sizeof(int32_t) * (size_t)((rows > 0 ? rows : 1) + 1), 256);
    v.qlist = reinterpret_cast<int32_t*>(p + off);
    off = align_up(off + sizeof(int32_t) * (size_t)(rows > 0 ? rows : 1), 256);
    v.qinl = reinterpret_cast<int32_t*>(p + off);
    off = align_up(off + sizeof(int32_t) * 16 * (size_t)(rows > 0 ? rows : 1), 256);
    v.bytes = off;
    return v;
}
inline int64_t flatten_unit_cap(int NB, int G) { return (int64_t)NB * G; }

}  // namespace deft
