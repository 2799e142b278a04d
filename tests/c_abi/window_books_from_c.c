/* The window plan's host-side books (include/deft_amd.h "Window plans": deft_window_create / _step / _free) driven from plain C: the
 * call sequence a runner written against the C ABI makes around every decode step -- hand over the journal and the step's slots, get
 * the patch list deft_window_patch reads, or -1 = "run a replan step".  Host-only: runs without a GPU (tests/test_host_logic.py). */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "deft_amd.h"

#define CHECK(c)                                                        \
    do {                                                                \
        if (!(c)) {                                                     \
            fprintf(stderr, "%s:%d: %s failed (%s)\n", __FILE__, __LINE__, #c, deft_last_error()); \
            return 1;                                                   \
        }                                                               \
    } while (0)

int main(void) {
    /* a root (DFS 0) above two leaves (DFS 1, 2): query rows 0 and 1; Llama-3 geometry, 4 query heads per KV head */
    const int32_t leaf_node[2] = {1, 2};
    const uint64_t refs[3] = {0x3, 0x1, 0x2};
    int32_t out[1 + 64 + 3 * 32];
    CHECK(deft_abi_version() == 2);
    CHECK(deft_window_supported(2, 32, 32, 8) == 1 && deft_window_supported(2, 48, 32, 8) == 0);
    const int64_t w = deft_window_create(3, 2, 1, 1, leaf_node, refs, 32, 4, 32);
    CHECK(w > 0);
    const int32_t loc0[2] = {100, 101}, loc1[2] = {102, 103};
    CHECK(deft_window_step(w, 0, NULL, 0, loc0, out, sizeof out / 4) == -1); /* no window yet: the caller replans */
    int64_t n = deft_window_step(w, 1, NULL, 0, loc0, out, sizeof out / 4);
    CHECK(n == 65 + 3 * 2 && out[0] == 2 && out[1] == 1);
    /* both leaves' tokens in region 0 (8 rows of its 32): masks = 4 rows each, values = "new row r" */
    CHECK(out[65] == 0 && out[66] == 0xf && out[67] == -1 && out[68] == 1 && out[69] == 0xf0 && out[70] == -2);
    n = deft_window_step(w, 0, NULL, 0, loc1, out, sizeof out / 4);
    CHECK(n == 65 + 3 * 4 && out[0] == 4); /* last step's rows get their pool slots, this step's go behind them */
    int seen_pool = 0, seen_new = 0;
    for (int i = 0; i < 4; ++i) {
        const int32_t pos = out[65 + 3 * i] & 0xfffff, val = out[67 + 3 * i];
        if (pos < 2) seen_pool += (val == loc0[pos]);
        else seen_new += (val == -1 - (pos - 2));
    }
    CHECK(seen_pool == 2 && seen_new == 2);
    /* a RESET of a leaf whose first token sits in the static part of the plan: not expressible, replan */
    const int32_t journal[3] = {2, 1, 0};
    CHECK(deft_window_step(w, 0, journal, 3, loc1, out, sizeof out / 4) == -1);
    CHECK(deft_window_step(w, 1, journal, 3, loc1, out, sizeof out / 4) == 65 + 3 * 2);
    CHECK(deft_window_free(w) == 0 && deft_window_free(w) != 0);
    printf("window books from C: ok\n");
    return 0;
}
