/*
 * deft_amd.h — C ABI of libdeft_amd.so: MI355X (gfx950) tree-attention decode path.
 *
 * This is the drop-in boundary for ONE path of LINs-lab/DeFT: paged DeFT-Flatten /
 * DeFT-Node attention at decode time.  The reference has no native code for this
 * path (its kernels are Triton, DeFT/deft/layers/attention/tree_attention.py); the
 * functions below are what the reference's two Python operators and their
 * companions would bind over ctypes (INTEGRATION.md shows the stubs):
 *
 *   deft_flatten_decode_f16   replaces tree_attention_subtree_fwd
 *                             (tree_attention.py:551-667: kernel2 :859-976 + stage 2 :296-546)
 *   deft_node_decode_f16      replaces tree_attention_fwd
 *                             (tree_attention.py:14-68: stage 1 :81-293 + stage 2 :296-546)
 *   deft_kv_append_f16        replaces KVCacheUpdater.update, paged branch
 *                             (DeFT/deft/tree_decoding/tree_cache.py:67-76, called from
 *                              DeFTAttention.store_kv_cache, deft_attention.py:390-403)
 *   deft_md_*                 replaces TreeMetadata.from_tree_cache
 *                             (tree_cache.py:618-881), host side, no GPU needed
 *
 * Conventions
 *   - plain pointers and sizes only; device pointers are raw HIP device addresses;
 *     `stream` is a hipStream_t passed as void* (NULL = the null stream).
 *   - every function returns 0 on success or a negative DEFT_E* code and never
 *     throws; deft_last_error() returns a thread-local message for the last failure.
 *   - all launches are asynchronous on `stream`; no function synchronises, allocates
 *     device memory or keeps state between calls.  The caller owns every buffer,
 *     including `workspace` (size it with the *_workspace_bytes functions).
 *   - strides are in ELEMENTS of the pointed-to type.
 *   - fp16 tensors are IEEE binary16; index tensors are int64 exactly as the
 *     reference's TreeMetadata holds them (tree_cache.py:813-857); cache_loc is int32
 *     (memory_pool.py:80).
 *   - KV pool layout is the reference's: per layer [slot][2][Hkv][D] fp16
 *     (memory_pool.py:61-66).  k_base = &kv_data[0][0][0][0], v_base = &kv_data[0][1][0][0],
 *     kv_stride_slot = 2*Hkv*D, kv_stride_head = D.  Any other strides are accepted
 *     as long as rows of D elements are contiguous and 16-byte aligned.
 */
#ifndef DEFT_AMD_H
#define DEFT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: the functions declared in this header -- and nothing else -- are its
 * dynamic symbols (tests/test_host_logic.py::test_library_exports_nothing_but_the_declared_symbols compares `nm -D` with the declarations). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define DEFT_OK 0
#define DEFT_EINVAL (-1)       /* bad argument (null pointer, negative size, misalignment) */
#define DEFT_EUNSUPPORTED (-2) /* geometry the kernels do not cover (see deft_supported) */
#define DEFT_EHIP (-3)         /* a HIP runtime call failed; message has hipGetErrorString */
#define DEFT_EWORKSPACE (-4)   /* workspace too small */

#define DEFT_BLOCK_LEN 128 /* BLOCK_CONFIG["BLOCK_LEN"], tree_cache.py:587; kernel tile, tree_attention.py:655 */
#define DEFT_MAX_Q_LEN 32  /* from_tree_cache(max_q_len=32), tree_cache.py:623; BLOCK_M, tree_attention.py:656 */

int deft_abi_version(void);
const char* deft_last_error(void);

/* Key for callers that cache plans (deft_*_build_plan) across calls: everything a plan's layout depends on
 * besides the arguments it was built from.  Always 0 in the shipped library, which reads no environment; the
 * experiments build (`make -C deft_amd/csrc exp`, A/B measurements only) folds its plan knobs into it. */
int deft_plan_variant(void);

/* 1 if (Hq, Hkv, D) is covered: Hq % Hkv == 0, D in {16, 32, 64, 128} -- the head dims the reference asserts
 * (tree_attention.py:100,305,582).  D = 128 (Llama) runs the LDS-DMA chunk kernel; the smaller ones a tile-per-workgroup kernel. */
int deft_supported(int Hq, int Hkv, int D);

/* ---- DeFT-Flatten ------------------------------------------------------- */

/* Bytes of device scratch deft_flatten_decode_f16 needs for this shape. */
size_t deft_flatten_workspace_bytes(int NB, int P, int nq, int Hq, int Hkv, int D);

/*
 * Optional per-step plan.  The Flatten metadata is the same for every layer of one
 * decode step (the reference builds TreeMetadata once per step and all 32 layers read
 * it through a module global, tree_cache.py:1021-1037), so the device-side repack of it
 * (row byte offsets in the pool, 32-bit query masks, run boundaries, partial-row -> query
 * map) can be built once per step and handed to every layer's call.  `plan` is caller-
 * owned device memory of deft_flatten_plan_bytes(NB, P, Hq, Hkv) bytes; it depends on the
 * six metadata arrays, the head counts and the q / pool strides only (not on the layer).
 * Its header holds the arrival counters of the single-launch decode, which every launch leaves zeroed again, so a
 * plan serves one launch at a time (the layers of a step run in stream order).  Passing plan = NULL to the decode call
 * makes it build the plan itself into the workspace (one more small kernel per call).
 */
size_t deft_flatten_plan_bytes(int NB, int P, int Hq, int Hkv);
int deft_flatten_build_plan(
    const int64_t* block_q, const int64_t* block_q_cnts, const int64_t* block_q_offset,
    const int64_t* block_bitmasks, const int64_t* block_kv, const int64_t* block_lens,
    int NB, int P, int Hq, int Hkv, int64_t q_stride_tok, int64_t q_stride_head, int64_t kv_stride_slot,
    const int32_t* cache_loc /* nullable */, int n_new, int64_t new_stride_tok,
    void* plan, size_t plan_bytes, void* stream);

/* The plan for metadata built ON THE DEVICE (deft_tree_dev_build_md below): NB and P are the CAPACITIES of the arrays --
 * they size the plan, the grids and the partial-row stride; pass the same two values to the decode call -- and the block
 * count of the current step is read by the kernel from dims[5] (the first words of deft_tree_dev_build_md's scratch).
 * The launch therefore has identical arguments on every decode step of a structural epoch of the tree and can be part
 * of a captured hipGraph of the whole step (deft_amd/session.py). */
int deft_flatten_build_plan_dims(
    const int64_t* block_q, const int64_t* block_q_cnts, const int64_t* block_q_offset,
    const int64_t* block_bitmasks, const int64_t* block_kv, const int64_t* block_lens,
    int NB, int P, const int32_t* dims, int Hq, int Hkv, int64_t q_stride_tok, int64_t q_stride_head, int64_t kv_stride_slot,
    const int32_t* cache_loc /* nullable */, int n_new, int64_t new_stride_tok,
    void* plan, size_t plan_bytes, void* stream);

/*
 * out[nq,Hq,D] = tree attention of q over the flattened-tree blocks.
 *   q, out              fp16, [nq][Hq][D] with the given token/head strides
 *   block_q[P]          query row of every partial row, grouped per block
 *   block_q_cnts[NB]    queries per block (1..32)
 *   block_q_offset[NB]  exclusive prefix sum of block_q_cnts
 *   block_bitmasks[NB*128]  bit r set <=> r-th query of the block sees the slot
 *   block_kv[NB*128]    pool slot per position, -1 padded
 *   block_lens[NB]      valid positions per block
 *   scale               1/sqrt(D) in the reference (tree_attention.py:601)
 * `out` is overwritten (the reference accumulates into a pre-zeroed tensor,
 * tree_attention.py:546; pre-zeroing is tolerated, not required).
 */
int deft_flatten_decode_f16(
    const void* q, int64_t q_stride_tok, int64_t q_stride_head,
    const void* k_base, const void* v_base, int64_t kv_stride_slot, int64_t kv_stride_head,
    void* out, int64_t o_stride_tok, int64_t o_stride_head,
    const int64_t* block_q, const int64_t* block_q_cnts, const int64_t* block_q_offset,
    const int64_t* block_bitmasks, const int64_t* block_kv, const int64_t* block_lens,
    int NB, int P, int nq, int Hq, int Hkv, int D, float scale,
    const void* plan, void* workspace, size_t workspace_bytes, void* stream);

/*
 * DeFTAttention.deft_flatten_forward in one call (deft_attention.py:110-151): the paged append of
 * this step's K/V rows (store_kv_cache, :390-403 -> tree_cache.py:67-76) fused into the attention
 * launch.  k_base / v_base are written at cache_loc[i] (i < n_new) with k_new[i] / v_new[i]
 * (rows of Hkv*D fp16, token stride new_stride_tok) and the attention sees the new rows.
 * `plan` may be NULL (built per call) or a plan built with the same cache_loc arguments.
 */
int deft_flatten_decode_append_f16(
    const void* q, int64_t q_stride_tok, int64_t q_stride_head,
    void* k_base, void* v_base, int64_t kv_stride_slot, int64_t kv_stride_head,
    void* out, int64_t o_stride_tok, int64_t o_stride_head,
    const int64_t* block_q, const int64_t* block_q_cnts, const int64_t* block_q_offset,
    const int64_t* block_bitmasks, const int64_t* block_kv, const int64_t* block_lens,
    int NB, int P, int nq, int Hq, int Hkv, int D, float scale,
    const int32_t* cache_loc, const void* k_new, const void* v_new, int64_t new_stride_tok, int n_new,
    const void* plan, void* workspace, size_t workspace_bytes, void* stream);

/* ---- DeFT-Node ---------------------------------------------------------- */

size_t deft_node_workspace_bytes(int NE, int P, int64_t total_kv, int nq, int Hq, int Hkv, int D);

/* Optional per-step plan for Node mode; same contract as the Flatten plan above. */
size_t deft_node_plan_bytes(int NE, int P, int64_t total_kv, int Hq, int Hkv);
int deft_node_build_plan(
    const int64_t* node_kv, const int64_t* node_kv_offset, const int64_t* node_kv_len,
    const int64_t* node_q, const int64_t* node_q_offset, const int64_t* node_q_len,
    int NE, int P, int64_t total_kv, int Hq, int Hkv,
    int64_t q_stride_tok, int64_t q_stride_head, int64_t kv_stride_slot,
    const int32_t* cache_loc /* nullable */, int n_new, int64_t new_stride_tok,
    void* plan, size_t plan_bytes, void* stream);

/* For metadata built on the device (see deft_flatten_build_plan_dims): NE, P, total_kv are capacities, dims[1] = this step's entries. */
int deft_node_build_plan_dims(
    const int64_t* node_kv, const int64_t* node_kv_offset, const int64_t* node_kv_len,
    const int64_t* node_q, const int64_t* node_q_offset, const int64_t* node_q_len,
    int NE, int P, int64_t total_kv, const int32_t* dims, int Hq, int Hkv,
    int64_t q_stride_tok, int64_t q_stride_head, int64_t kv_stride_slot,
    const int32_t* cache_loc /* nullable */, int n_new, int64_t new_stride_tok,
    void* plan, size_t plan_bytes, void* stream);

/*
 *   node_kv[total_kv]       pool slots of every entry, concatenated
 *   node_kv_offset/len[NE]  slice of node_kv per entry
 *   node_q[P]               query rows of every entry, concatenated
 *   node_q_offset/len[NE]   slice of node_q per entry (len 1..32)
 * Long entries are split internally into 128-slot tiles (the reference walks a
 * node serially in 16-token tiles, tree_attention.py:230); the result is the same
 * attention output.
 */
int deft_node_decode_f16(
    const void* q, int64_t q_stride_tok, int64_t q_stride_head,
    const void* k_base, const void* v_base, int64_t kv_stride_slot, int64_t kv_stride_head,
    void* out, int64_t o_stride_tok, int64_t o_stride_head,
    const int64_t* node_kv, const int64_t* node_kv_offset, const int64_t* node_kv_len,
    const int64_t* node_q, const int64_t* node_q_offset, const int64_t* node_q_len,
    int NE, int P, int64_t total_kv, int nq, int Hq, int Hkv, int D, float scale,
    const void* plan, void* workspace, size_t workspace_bytes, void* stream);

/*
 * DeFTAttention.deft_node_forward in one call (deft_attention.py:72-108): the paged append of this step's K/V rows
 * fused into the Node attention launch; same contract as deft_flatten_decode_append_f16.
 */
int deft_node_decode_append_f16(
    const void* q, int64_t q_stride_tok, int64_t q_stride_head,
    void* k_base, void* v_base, int64_t kv_stride_slot, int64_t kv_stride_head,
    void* out, int64_t o_stride_tok, int64_t o_stride_head,
    const int64_t* node_kv, const int64_t* node_kv_offset, const int64_t* node_kv_len,
    const int64_t* node_q, const int64_t* node_q_offset, const int64_t* node_q_len,
    int NE, int P, int64_t total_kv, int nq, int Hq, int Hkv, int D, float scale,
    const int32_t* cache_loc, const void* k_new, const void* v_new, int64_t new_stride_tok, int n_new,
    const void* plan, void* workspace, size_t workspace_bytes, void* stream);

/* ---- rotary embedding + paged append + attention in ONE stage-1 launch ---------------
 *
 * What LlamaAttention.forward does between the qkv projection and the output projection (llama2.py:108-111:
 * rotary_emb(positions, q, k) -> attn(q, k, v)), as one call: q and k_new are the UNROTATED rows of the fused qkv;
 * the kernel rotates a query row while it becomes an MFMA operand, this step's key rows on their way into the pool
 * and inside the tiles that attend to them.  q / k_new are left unrotated (the reference rotates them in place, then
 * only reads them here).  Arithmetic = deft_rope_qk_f16 (fp32, no contraction, one rounding): outputs and pool bytes
 * are bit-identical to deft_rope_qk_f16 followed by the *_decode_append_f16 call.
 * cos_sin_rows[n_new][head_dim] fp32 = cos_sin_cache[positions[j]] for row j of q / k_new (query row j), gathered
 * ONCE per decode step and shared by all layers (deft_rope_gather_rows: positions and cache as in deft_rope_qk_f16) --
 * so that no positions -> cache indirection stands in front of a workgroup's first MFMA.
 * Supported: head_dim 128, rotary_dim == head_dim, NeoX pairing, token-major q (DEFT_EUNSUPPORTED otherwise: call
 * deft_rope_qk_f16 and the append entry point instead).
 */
int deft_flatten_decode_rope_append_f16(
    const void* q, int64_t q_stride_tok, int64_t q_stride_head, void* k_base, void* v_base, int64_t kv_stride_slot,
    int64_t kv_stride_head, void* out, int64_t o_stride_tok, int64_t o_stride_head,
    const int64_t* block_q, const int64_t* block_q_cnts, const int64_t* block_q_offset, const int64_t* block_bitmasks,
    const int64_t* block_kv, const int64_t* block_lens, int NB, int P, int nq, int Hq, int Hkv, int D, float scale,
    const int32_t* cache_loc, const void* k_new, const void* v_new, int64_t new_stride_tok, int n_new,
    const float* cos_sin_rows, int rotary_dim, int is_neox_style,
    const void* plan, void* workspace, size_t workspace_bytes, void* stream);
int deft_rope_gather_rows(const int64_t* positions, const float* cos_sin_cache, int64_t cache_stride, int n, int rotary_dim,
                          float* rows_out, void* stream);
int deft_node_decode_rope_append_f16(
    const void* q, int64_t q_stride_tok, int64_t q_stride_head, void* k_base, void* v_base, int64_t kv_stride_slot,
    int64_t kv_stride_head, void* out, int64_t o_stride_tok, int64_t o_stride_head,
    const int64_t* node_kv, const int64_t* node_kv_offset, const int64_t* node_kv_len, const int64_t* node_q,
    const int64_t* node_q_offset, const int64_t* node_q_len, int NE, int P, int64_t total_kv, int nq, int Hq, int Hkv, int D,
    float scale, const int32_t* cache_loc, const void* k_new, const void* v_new, int64_t new_stride_tok, int n_new,
    const float* cos_sin_rows, int rotary_dim, int is_neox_style,
    const void* plan, void* workspace, size_t workspace_bytes, void* stream);

/* ---- causal prefill attention over the prompt (TTFT) -------------------------------------------------------
 *
 * Replaces context_attention_fwd (DeFT/deft/layers/attention/context_flashattention_nopad.py:130-195) behind
 * DeFTAttention.prefill_forward_triton (deft_attention.py:50-70): sequences packed without padding,
 * q[T][Hq][D], k / v[T][Hkv][D] fp16, token i of sequence b attends to tokens 0..i of b;
 * b_start_loc / b_seq_len int32 [batch] (model_runner.py:110-114).  head_dim 16 / 32 / 64 / 128 as upstream
 * (context_flashattention_nopad.py:134): 128 and 64 on the MFMA kernel, 32 and 16 on a plain one.
 */
int deft_prefill_f16(const void* q, int64_t q_stride_tok, int64_t q_stride_head,
                     const void* k, int64_t k_stride_tok, int64_t k_stride_head,
                     const void* v, int64_t v_stride_tok, int64_t v_stride_head,
                     void* out, int64_t o_stride_tok, int64_t o_stride_head,
                     const int32_t* b_start_loc, const int32_t* b_seq_len, int batch, int max_input_len,
                     int Hq, int Hkv, int D, float scale, void* stream);

/* ---- rotary position embedding of this step's q / k rows (the op in front of the path) -----------------
 *
 * In place, like RotaryEmbedding.forward_cuda (DeFT/deft/layers/rotary_embedding.py:157-177 ->
 * flashinfer.rope.apply_rope_with_cos_sin_cache_inplace; LlamaAttention.forward, llama2.py:108-110):
 * q[n][Hq][D], k[n][Hk][D] fp16 (strided views of the fused qkv are fine), positions[n] int64,
 * cos_sin_cache[max_pos][rotary_dim] fp32 = cos(rotary_dim/2) | sin(rotary_dim/2)  (rotary_embedding.py:119-127,
 * created with dtype=float32 at llama2.py:86-93).  fp32 arithmetic, one rounding to fp16.
 */
int deft_rope_qk_f16(void* q, int64_t q_stride_tok, int64_t q_stride_head, int Hq,
                     void* k, int64_t k_stride_tok, int64_t k_stride_head, int Hk,
                     const int64_t* positions, const float* cos_sin_cache, int64_t cache_stride,
                     int n, int D, int rotary_dim, int is_neox_style, void* stream);

/* ---- sequential (per-request) paged attention: the reference's comparator ---------------------------------
 *
 * Replaces token_attention_fwd (DeFT/deft/layers/attention/token_attention.py:297-335) behind
 * DeFTAttention.radix_attention_forward (deft_attention.py:153-188), i.e. `--mode seq --mem paged`: request i attends
 * to req_to_token[b_req_idx[i], 0 : b_seq_len[i]] (ReqToTokenPool, memory_pool.py:11-45; int32 like the reference),
 * b_start_loc = exclusive prefix sum of b_seq_len (model_runner.py:177-178), one query row per request.
 * The plan (page table -> per-request slot lists + tile records) is built once per decode step and shared by all
 * layers, like the Flatten / Node plans.
 */
size_t deft_seq_plan_bytes(int nq, int64_t total_tokens, int Hq, int Hkv);
size_t deft_seq_workspace_bytes(int nq, int64_t total_tokens, int Hq, int Hkv, int D);
int deft_seq_build_plan(
    const int32_t* req_to_token, int64_t req_stride /* elements per request row */,
    const int32_t* b_req_idx, const int32_t* b_start_loc, const int32_t* b_seq_len, int nq, int64_t total_tokens,
    int Hq, int Hkv, int64_t q_stride_tok, int64_t q_stride_head, int64_t kv_stride_slot,
    const int32_t* cache_loc /* nullable */, int n_new, int64_t new_stride_tok,
    void* plan, size_t plan_bytes, void* stream);
int deft_seq_decode_f16(
    const void* q, int64_t q_stride_tok, int64_t q_stride_head,
    const void* k_base, const void* v_base, int64_t kv_stride_slot, int64_t kv_stride_head,
    void* out, int64_t o_stride_tok, int64_t o_stride_head,
    const void* plan, int nq, int64_t total_tokens, int Hq, int Hkv, int D, float scale,
    void* workspace, size_t workspace_bytes, void* stream);
/* store_kv_cache + token_attention_fwd in one call (radix_attention_forward, deft_attention.py:153-188) */
int deft_seq_decode_append_f16(
    const void* q, int64_t q_stride_tok, int64_t q_stride_head,
    void* k_base, void* v_base, int64_t kv_stride_slot, int64_t kv_stride_head,
    void* out, int64_t o_stride_tok, int64_t o_stride_head,
    const void* plan, int nq, int64_t total_tokens, int Hq, int Hkv, int D, float scale,
    const int32_t* cache_loc, const void* k_new, const void* v_new, int64_t new_stride_tok, int n_new,
    void* workspace, size_t workspace_bytes, void* stream);

/* ---- paged KV append ---------------------------------------------------- */

/* k_base[cache_loc[i]] = k_new[i], v_base[cache_loc[i]] = v_new[i] for i < n
 * (rows of Hkv*D fp16; new rows have token stride new_stride_tok and head stride D). */
int deft_kv_append_f16(
    void* k_base, void* v_base, int64_t kv_stride_slot, int64_t kv_stride_head,
    const int32_t* cache_loc, const void* k_new, const void* v_new, int64_t new_stride_tok,
    int n, int Hkv, int D, void* stream);

/* ---- stage-level entry points (benchmarks / profiling) ------------------- */

/* Measurement aid, not on the product path: ONE launch of `workgroups` x 256 threads that reads `bytes` bytes at `base` (16-byte
 * aligned) with coalesced 16-byte loads, four in flight per thread, and does nothing else.  What a cold launch of that size can
 * draw from HBM on this part: the ceiling a stage-1 launch over the same K/V bytes is judged against (bench.py `ceiling_us`;
 * tools/probes/launch_ceiling.hip is the stand-alone form). */
int deft_probe_stream_read(const void* base, size_t bytes, int workgroups, void* stream);

/* Flatten stage 1 only: writes normalised fp32 partials + LSE into the workspace
 * (same layout deft_flatten_decode_f16 uses).  For roofline timing of the dominant
 * kernel in isolation. */
int deft_flatten_stage1_f16(
    const void* q, int64_t q_stride_tok, int64_t q_stride_head,
    const void* k_base, const void* v_base, int64_t kv_stride_slot, int64_t kv_stride_head,
    const int64_t* block_q, const int64_t* block_q_cnts, const int64_t* block_q_offset,
    const int64_t* block_bitmasks, const int64_t* block_kv, const int64_t* block_lens,
    int NB, int P, int nq, int Hq, int Hkv, int D, float scale,
    const void* plan, void* workspace, size_t workspace_bytes, void* stream);

/* Copy stage-1 partials out of a workspace (tests): partial_o[Hq][P][D] fp32, partial_lse[Hq][P] fp32. */
int deft_flatten_read_partials(
    const void* workspace, size_t workspace_bytes, int NB, int P, int nq, int Hq, int Hkv, int D,
    float* partial_o_dev, float* partial_lse_dev, void* stream);

/* ---- host-side metadata builder (no GPU) --------------------------------- */

/*
 * TreeMetadata.from_tree_cache (tree_cache.py:618-881) on a compact tree description.
 * Nodes are given in any order; children are visited in ascending node id, which is
 * the reference's dict-insertion (= creation) order (tree_cache.py:251-256, :790).
 *   node_id[n]       unique ids; the root is the node with parent_id < 0
 *   parent_id[n]     id of the parent, -1 for the root
 *   is_leaf[n]       1 for live leaves (TreeCache.leaves)
 *   kv_offset[n+1]   slice of kv_slots per node (unsorted, as appended)
 *   kv_slots[...]    pool slots
 * Query rows are the live leaves sorted by id (tree_cache.py:650-652).
 * Returns a handle (>0) or a negative error.
 */
int64_t deft_md_build(
    int n_nodes, const int64_t* node_id, const int64_t* parent_id, const uint8_t* is_leaf,
    const int64_t* kv_offset, const int64_t* kv_slots,
    int max_q_len, int block_len, int max_block_len);

/* sizes[0..7] = query_num, node_num (NE), total_kv_len, len(node_q), len(node_kv),
 *               NB, len(block_q) (P), len(block_kv) (= NB*block_len) */
int deft_md_sizes(int64_t handle, int64_t sizes[8]);

/* Copy the arrays into caller buffers of the sizes reported above (host memory). */
int deft_md_fetch(
    int64_t handle,
    int64_t* node_q, int64_t* node_kv, int64_t* node_q_len, int64_t* node_kv_len,
    int64_t* node_q_offset, int64_t* node_kv_offset,
    int64_t* block_q, int64_t* block_q_cnts, int64_t* block_q_offset,
    int64_t* block_bitmasks, int64_t* block_kv, int64_t* block_lens,
    int64_t* leaf_ids /* [query_num] node id of each query row */);

int deft_md_free(int64_t handle);

/*
 * The decoding tree over the paged pool (host side; deft_amd/csrc/tree.cpp).  The state machine behind TreeCache
 * (tree_cache.py:147-403, :504-516): nodes with a parent and children in creation order, their pool slots, which nodes are
 * live leaves and how many live leaves sit below every node (the reference's `refs` sets).  deft_amd.TreeCache / TreeNode are
 * attribute views over this object.  Query row r of a decode step = the r-th live leaf by id (tree_cache.py:650-652).
 */
int64_t deft_tree_create(void);
int deft_tree_free(int64_t tree);
int deft_tree_add_node(int64_t tree, int64_t id, int64_t parent_id /* -1 = root */);
int deft_tree_remove_node(int64_t tree, int64_t id);
int deft_tree_set_leaf(int64_t tree, int64_t id, int is_leaf);
/* TreeCache.branch (:338-370): `id` stops being a leaf; cnt new live leaves first_id .. first_id + cnt - 1 below it */
int deft_tree_branch(int64_t tree, int64_t id, int cnt, int64_t first_id);
/* TreeCache.cut (:373-403): the leaf and every ancestor left without a live leaf; returns their ids (leaf first) and slots.
 * *n_ids / *n_slots always report the counts; DEFT_EWORKSPACE (nothing changed) if a buffer is too small. */
int deft_tree_cut(int64_t tree, int64_t id, int64_t* deleted_ids, int cap_ids, int* n_ids, int64_t* freed_slots,
                  int64_t cap_slots, int64_t* n_slots);
/* TreeCache.alloc (:261-283): one new slot per live leaf, ascending leaf id */
int deft_tree_alloc_step(int64_t tree, int n, const int64_t* slots);
int deft_tree_append_slots(int64_t tree, int n, const int64_t* ids, const int64_t* slots); /* one slot per id */
int deft_tree_extend_node(int64_t tree, int64_t id, int n, const int64_t* slots);
int deft_tree_set_node_kv(int64_t tree, int64_t id, int n, const int64_t* slots);
int deft_tree_clear_node_kv(int64_t tree, int64_t id);
int64_t deft_tree_take_nodes_kv(int64_t tree, int n, const int64_t* ids, int64_t* out, int64_t cap); /* slots of n nodes moved out (lists left empty); returns the total */
int64_t deft_tree_node_len(int64_t tree, int64_t id);
int64_t deft_tree_node_kv(int64_t tree, int64_t id, int64_t* out, int64_t cap);    /* returns the length */
int64_t deft_tree_node_refs(int64_t tree, int64_t id, int64_t* out, int64_t cap);  /* live leaves below, ascending */
int64_t deft_tree_path_slots(int64_t tree, int64_t id, int64_t* out, int64_t cap); /* root -> node slots */
int deft_tree_leaf_ids(int64_t tree, int64_t* out, int cap);                       /* query-row order; returns the count */
int deft_tree_stats(int64_t tree, int64_t stats[4]); /* nodes, live leaves, total KV slots, structure epoch */
/* == deft_md_build(<the tree>); handle consumed with deft_md_sizes / _fetch / _free */
int64_t deft_tree_build_md(int64_t tree, int max_q_len, int block_len, int max_block_len);

/*
 * The same tree on the GPU: KV-guided grouping and flattened split as HIP kernels (deft_amd/csrc/tree_plan.h), so a decode
 * step uploads nothing but its nq new slot numbers.
 *   deft_tree_layout        nodes in DFS pre-order, room for `slack` more tokens per live leaf; sizes = {nodes, queries,
 *                           64-bit words per leaf set, total slot capacity, epoch}.  The layout (and the device copy made from
 *                           deft_tree_layout_fetch) stays valid while the epoch does: deft_tree_alloc_step keeps it as long as
 *                           every leaf has room; every other mutation bumps it.  SIDE EFFECT (ABI version 2): the caller is about
 *                           to fetch an image that holds every change made so far, so the journal of absorbed changes
 *                           (deft_tree_journal_take) is CLEARED here -- and when it was not empty the epoch is bumped first (the
 *                           layout stays valid): other device copies of the old epoch never saw those changes and must upload.
 *                           sizes[4] is the epoch the image will carry.
 *   deft_tree_layout_fetch  fills the upload image of that layout (and swallows, the same way, whatever was journalled between the
 *                           two calls: a caller that mutates the tree in between reads the epoch again, deft_tree_stats).
 *   deft_tree_md_sizes      sizes of the metadata for the current lengths + `grow` tokens per leaf: sizes[0..7] as
 *                           deft_md_sizes, sizes[8] = physical 128-slot blocks (capacity planning and tensor shapes; no slot touched)
 *   deft_tree_dev_advance   device: append cache_loc[r] to query row r's leaf (kept ascending inside the node)
 *   deft_tree_dev_build_md  device: the twelve int64 arrays of TreeMetadata, bit for bit deft_md_build's; with `advance_loc`
 *                           (this step's cache_loc, nullable) the append of deft_tree_dev_advance is folded into its first kernel;
 *                           the six node_* pointers or the six block_* pointers may all be NULL: that group is not written
 *                           (a Flatten step reads only the block arrays, a Node step only the node arrays)
 */
int deft_tree_layout(int64_t tree, int slack, int64_t sizes[5]);
int deft_tree_layout_fetch(int64_t tree, int32_t* node_start, int32_t* node_len, int32_t* node_cap, uint64_t* refs,
                           int32_t* leaf_node, int32_t* slots);
int deft_tree_md_sizes(int64_t tree, int max_q_len, int block_len, int max_block_len, int grow, int64_t sizes[9]);
int deft_tree_md_sizes_upto(int64_t tree, int max_q_len, int block_len, int max_block_len, int grow_max, int64_t sizes[9]); /* element-wise max over every growth 0..grow_max: epoch capacities (the block arrays are not monotone in the growth) */
int deft_tree_md_caps(int64_t tree, int max_q_len, int block_len, int max_block_len, int grow_max, int64_t sizes[9]); /* O(nodes) upper bounds of the same maxima: what an epoch's buffers are sized with (_upto is the exact checker) */
size_t deft_tree_dev_scratch_bytes(int n_nodes, int nqw, int nbp_cap);
int deft_tree_dev_advance(int n_nodes, int nq, int nqw, const int32_t* node_start, int32_t* node_len, const int32_t* node_cap,
                          const uint64_t* refs, const int32_t* leaf_node, int32_t* slots, const int32_t* cache_loc,
                          void* scratch, void* stream);
int deft_tree_dev_build_md(int n_nodes, int nq, int nqw, const int32_t* node_start, int32_t* node_len, const int32_t* node_cap,
                           const uint64_t* refs, const int32_t* leaf_node, int32_t* slots, int max_q_len, int block_len,
                           int max_block_len, int nbp_cap, void* scratch, size_t scratch_bytes, int64_t* node_q, int64_t* node_kv,
                           int64_t* node_q_len, int64_t* node_kv_len, int64_t* node_q_offset, int64_t* node_kv_offset,
                           int64_t* block_q, int64_t* block_q_cnts, int64_t* block_q_offset, int64_t* block_bitmasks,
                           int64_t* block_kv, int64_t* block_lens, const int32_t* advance_loc /* nullable */, void* stream);

/* Changes a structural epoch ABSORBS besides a decode step's one slot per leaf: slots appended to a node that has room in the
 * layout (deft_tree_extend_node) and a node's slots dropped (deft_tree_take_nodes_kv, deft_tree_set_node_kv(n = 0)) -- what the
 * reference's speculative-decoding mock does every step (branch_func_example.py:420-437: merge_nodes of the accepted leaves into
 * the root, reset_node_KV of every leaf).  The host tree journals them instead of bumping the epoch:
 *   deft_tree_journal_take   hands the journal over ONCE (int32 words, oldest first: {1 = EXTEND, DFS index, n, n slots} |
 *                            {2 = RESET, DFS index, 0}); returns the words written, 0 = none, -5 = longer than `cap` (the call has
 *                            then started a new epoch, whose upload carries everything)
 *   deft_tree_dev_apply_ops  replays it on the device copy: `ops` = DEVICE buffer {words, journal ...}
 *   deft_tree_dev_build_md_ops  = deft_tree_dev_build_md with that replay folded into its first kernel (in front of the advance),
 *                            the word count read at run time: identical arguments on every step of an epoch (captured decode steps) */
int64_t deft_tree_journal_take(int64_t tree, int32_t* out, int64_t cap);
int deft_tree_dev_apply_ops(int n_nodes, int nq, int nqw, const int32_t* node_start, int32_t* node_len, const int32_t* node_cap,
                            const uint64_t* refs, const int32_t* leaf_node, int32_t* slots, const int32_t* ops, void* scratch,
                            void* stream);
int deft_tree_dev_build_md_ops(int n_nodes, int nq, int nqw, const int32_t* node_start, int32_t* node_len, const int32_t* node_cap,
                               const uint64_t* refs, const int32_t* leaf_node, int32_t* slots, int max_q_len, int block_len,
                               int max_block_len, int nbp_cap, void* scratch, size_t scratch_bytes, int64_t* node_q,
                               int64_t* node_kv, int64_t* node_q_len, int64_t* node_kv_len, int64_t* node_q_offset,
                               int64_t* node_kv_offset, int64_t* block_q, int64_t* block_q_cnts, int64_t* block_q_offset,
                               int64_t* block_bitmasks, int64_t* block_kv, int64_t* block_lens,
                               const int32_t* advance_loc /* nullable */, const int32_t* ops /* nullable */,
                               /* optional: page_table[page_rows[r] * page_stride + page_cols[r]] = advance_loc[r], r < nq -- the
                                  page-table write of TreeCache.alloc (tree_cache.py:270-283) in the same kernel */
                               int32_t* page_table /* nullable */, int64_t page_stride, const int64_t* page_rows,
                               const int64_t* page_cols, void* stream);

/* Window plans: the INCREMENTAL per-step head of a captured decode loop (deft_amd.DecodeSession; deft_amd/csrc/window.h).
 * Replaces, for most steps of a structural epoch, what the reference rebuilds whole on every decode step --
 * TreeMetadata.from_tree_cache, DeFT/deft/tree_decoding/tree_cache.py:619-881, timed at tree_generate.py:123-131.
 *   deft_window_supported            1 iff steps of this shape can run on a window plan
 *   deft_flatten_build_plan_window   a REPLAN step's plan: the device-built block arrays (deft_tree_dev_build_md_ops WITHOUT the
 *                                    step's advance: the tree as it stands before the step's tokens) get `win_tiles` overflow blocks
 *                                    per chunk of max_q_len query rows appended (NB, P = capacities INCLUDING them; dims[5..7] are
 *                                    raised, dims[10] / dims[11] keep the counts in front of them), then units, records, row lists.
 *                                    `win_tab` (int32 [chunks][16][2], device) receives {leader record, first follower record} of
 *                                    every overflow run
 *   deft_node_build_plan_window      the same for the node arrays: one overflow entry of win_tiles * 128 slots per query chunk
 *   deft_window_patch                EVERY step of a window, replan steps included (ONE workgroup): journal replay (`ops`, nullable:
 *                                    a replan step's scan already did), page-table write, the step's slots appended to the device
 *                                    tree, and the host's patch list applied to the plan: int32 {entries, active overflow tiles of
 *                                    regions 0 .. 63, {region << 20 | position, row mask (0 = cleared), pool slot | -1 - new row} ...}
 *                                    (deft_window_step writes it: an entry's row mask is its node's leaf set within the region) */
int deft_window_supported(int nq, int max_q_len, int Hq, int Hkv);
int deft_flatten_build_plan_window(int64_t* block_q, int64_t* block_q_cnts, int64_t* block_q_offset, int64_t* block_bitmasks,
                                   int64_t* block_kv, int64_t* block_lens, int NB, int P, int32_t* dims, int nq, int max_q_len,
                                   int win_tiles, int32_t* win_tab, int Hq, int Hkv, int64_t q_stride_tok, int64_t q_stride_head,
                                   int64_t kv_stride_slot, void* plan, size_t plan_bytes, void* stream);
int deft_node_build_plan_window(int64_t* node_kv, int64_t* node_kv_offset, int64_t* node_kv_len, int64_t* node_q,
                                int64_t* node_q_offset, int64_t* node_q_len, int NE, int P, int64_t total_kv, int32_t* dims, int nq,
                                int max_q_len, int win_tiles, int32_t* win_tab, int Hq, int Hkv, int64_t q_stride_tok,
                                int64_t q_stride_head, int64_t kv_stride_slot, void* plan, size_t plan_bytes, void* stream);
int deft_window_patch(int n_nodes, int nq, int nqw, const int32_t* node_start, int32_t* node_len, const int32_t* node_cap,
                      const uint64_t* refs, const int32_t* leaf_node, int32_t* slots, const int32_t* ops /* nullable */,
                      const int32_t* cache_loc, int32_t* page_table /* nullable */, int64_t page_stride, const int64_t* page_rows,
                      const int64_t* page_cols, const int32_t* patch, const int32_t* win_tab, void* plan, int max_q_len,
                      int win_tiles, int Hq, int Hkv, int64_t kv_stride_slot, int64_t new_stride_tok, void* scratch,
                      /* optional (fetch_ring nullable): deft_stage_fetch folded into the kernel's opening */
                      const void* fetch_ring, int fetch_slot_bytes, int fetch_ring_n, void* fetch_dst, int32_t* fetch_counter,
                      void* stream);
/* A decode step's host-written words (slot numbers, page-table coordinates, journal, patch list) FETCHED by a kernel from a ring of
 * pinned, device-accessible host slots -- slot (*counter mod ring_n) = {uint32 used bytes, 12 bytes padding, payload}; its first `used`
 * payload bytes go to `dst`, then *counter += 1 -- instead of a hipMemcpyAsync in front of the captured step.  One workgroup; identical
 * arguments on every step, so it sits in the step's hipGraph.  OPTIONAL: deft_amd.DecodeSession copies by default -- the two forms
 * measure equal, and about one run in twenty ran 2.6 x slower with kernel-side PCIe reads (profiles/r6_staging_kernel_vs_copy.txt). */
int deft_stage_fetch(const void* ring, int slot_bytes, int ring_n, void* dst, int32_t* counter, void* stream);
/* The default hand-over of the same words: slot `slot` of the ring -- the host wrote `used` into its header -- copied to `dst` (room
 * for dst_bytes) by one hipMemcpyAsync on `stream`.  DEFT_EINVAL when the header names more than the slot or `dst` holds. */
int deft_stage_copy(const void* ring, int slot_bytes, int slot, void* dst, size_t dst_bytes, void* stream);

/* The host-side books of a window plan: which overflow position holds which node's slot (deft_amd/csrc/window_host.cpp).
 *   deft_window_create   books for one structural epoch: `leaf_node[r]` = DFS index of query row r's leaf, `refs` = the nodes' leaf sets
 *                        (both as deft_tree_layout_fetch returns them), `group` = Hq / Hkv, `tiles` overflow tiles per region -- a
 *                        region = one (chunk of max_q_len query rows, 32-row pass) pair --, at most `max_entries` patch entries per
 *                        step; returns a handle (< 0: error)
 *   deft_window_step     one decode step: the journal deft_tree_journal_take handed over and the step's nq slots -> the step's patch
 *                        list in `out` (what deft_window_patch reads); returns the int32 words written, or -1: the window cannot
 *                        express this step (replan = 0: run a replan step and call again with replan = 1; replan = 1: run the step
 *                        without a window) */
int64_t deft_window_create(int n_nodes, int nq, int nqw, int tiles, const int32_t* leaf_node, const uint64_t* refs, int max_q_len,
                           int group, int max_entries);
int deft_window_free(int64_t window);
int64_t deft_window_step(int64_t window, int replan, const int32_t* journal, int64_t journal_words, const int32_t* loc, int32_t* out,
                         int64_t out_cap);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif /* DEFT_AMD_H */
