"""Oracle: DeFT-Flatten / DeFT-Node two-stage attention and ground truth (numpy).

Test infrastructure only (see oracle/__init__.py).  Restates, on CPU:

  flatten_stage1   tree_attention_subtree_fwd_kernel2
                   DeFT/deft/layers/attention/tree_attention.py:859-976
  node_stage1      DeFT_splitBynode_Triton_stage1_kernel          :170-293
  merge_reference  DeFT_splitBynode_Triton_stage2 (+ _2_1, _2_3)  :296-416, :419-445, :484-546
                   including its quirks: row max initialised to 0 and
                   accumulation into the fp16 output before the division
  merge_exact      the mathematically identical merge with the true max and
                   fp32 accumulation (what the HIP path computes)
  token_attention_forward  the sequential comparator, token_attention.py:297-335
                   (fp16 logits and all), bit-exact on tests/golden/seq_*.npz
  sequential_truth per-leaf softmax(q K^T / sqrt(D)) V over the leaf's
                   root->leaf slots, fp64 — the recipe of
                   DeFT/tests/model/test_DeFT_kernel.py:212-276

Array conventions follow the reference operator (tree_attention.py:14-25, :552-568):
  q        [nq, Hq, D]   fp16
  kv_data  [slots, 2, Hkv, D] fp16   (memory_pool.py:61-66; K = [:,0], V = [:,1])
  out      [nq, Hq, D]   fp16
GQA: kv_head = head // (Hq // Hkv)   (tree_attention.py:894)
"""
from __future__ import annotations

from typing import Dict, Sequence, Tuple

import numpy as np


def _kv_views(kv_data: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    return kv_data[:, 0], kv_data[:, 1]


def _expand_heads(x: np.ndarray, group: int) -> np.ndarray:
    """[len, Hkv, D] -> [Hq, len, D] with each KV head repeated `group` times."""
    return np.repeat(np.transpose(x, (1, 0, 2)), group, axis=0)


# ---------------------------------------------------------------------------
# stage 1, Flatten (tree_attention.py:859-976)
# ---------------------------------------------------------------------------
def flatten_stage1(q, kv_data, md: Dict[str, object], block_n: int = 128):
    nq, Hq, D = q.shape
    kbuf, vbuf = _kv_views(kv_data)
    group = Hq // kbuf.shape[1]
    scale = np.float32(1.0 / (D ** 0.5))
    block_q = md["block_q"]
    P = block_q.shape[0]
    NB = md["block_q_cnts"].shape[0]
    partial_o = np.zeros((Hq, P, D), dtype=np.float32)
    partial_lse = np.zeros((Hq, P), dtype=np.float32)
    for b in range(NB):
        cnt = int(md["block_q_cnts"][b])
        off = int(md["block_q_offset"][b])
        cur_len = int(md["block_lens"][b])
        rows = block_q[off : off + cnt]
        slots = md["block_kv"][b * block_n : b * block_n + cur_len]
        masks = md["block_bitmasks"][b * block_n : b * block_n + cur_len]
        qt = np.transpose(q[rows].astype(np.float32), (1, 0, 2))  # [Hq, cnt, D]
        k = _expand_heads(kbuf[slots].astype(np.float32), group)  # [Hq, len, D]
        v = _expand_heads(vbuf[slots].astype(np.float32), group)
        s = np.matmul(qt, np.transpose(k, (0, 2, 1))) * scale  # [Hq, cnt, len]  (:944-945)
        visible = ((masks[None, :] >> np.arange(cnt, dtype=np.int64)[:, None]) & 1).astype(bool)
        s = np.where(visible[None], s, -np.inf)  # (:946-950)
        m = s.max(axis=2)  # single-tile softmax (:952-953)
        p = np.exp(s - m[..., None])
        acc = np.matmul(p, v)  # (:955)
        l = p.sum(axis=2)
        partial_o[:, off : off + cnt] = acc / l[..., None]  # (:964-968)
        partial_lse[:, off : off + cnt] = m + np.log(l)  # (:972-975)
    return partial_o, partial_lse


# ---------------------------------------------------------------------------
# stage 1, Node (tree_attention.py:170-293): 16-token tiles, running max
# ---------------------------------------------------------------------------
def node_stage1(q, kv_data, md: Dict[str, object], tile: int = 16):
    nq, Hq, D = q.shape
    kbuf, vbuf = _kv_views(kv_data)
    group = Hq // kbuf.shape[1]
    scale = np.float32(1.0 / (D ** 0.5))
    node_q = md["node_q"]
    P = node_q.shape[0]
    NE = md["node_q_len"].shape[0]
    partial_o = np.zeros((Hq, P, D), dtype=np.float32)
    partial_lse = np.zeros((Hq, P), dtype=np.float32)
    for e in range(NE):
        q_off, q_len = int(md["node_q_offset"][e]), int(md["node_q_len"][e])
        kv_off, kv_len = int(md["node_kv_offset"][e]), int(md["node_kv_len"][e])
        rows = node_q[q_off : q_off + q_len]
        qt = np.transpose(q[rows].astype(np.float32), (1, 0, 2))  # [Hq, q_len, D]
        m = np.full((Hq, q_len), -np.inf, dtype=np.float32)
        l = np.zeros((Hq, q_len), dtype=np.float32)
        acc = np.zeros((Hq, q_len, D), dtype=np.float32)
        for t0 in range(0, kv_len, tile):  # (:230)
            slots = md["node_kv"][kv_off + t0 : kv_off + min(t0 + tile, kv_len)]
            k = _expand_heads(kbuf[slots].astype(np.float32), group)
            v = _expand_heads(vbuf[slots].astype(np.float32), group)
            s = np.matmul(qt, np.transpose(k, (0, 2, 1))) * scale
            m_new = np.maximum(m, s.max(axis=2))  # (:262-263)
            p = np.exp(s - m_new[..., None])
            alpha = np.exp(m - m_new)  # (:268)
            acc = acc * alpha[..., None] + np.matmul(p, v)  # (:270-272)
            l = l * alpha + p.sum(axis=2)  # (:274)
            m = m_new
        partial_o[:, q_off : q_off + q_len] = acc / l[..., None]  # (:283-287)
        partial_lse[:, q_off : q_off + q_len] = m + np.log(l)  # (:289-293)
    return partial_o, partial_lse


# ---------------------------------------------------------------------------
# stage 2 (tree_attention.py:296-416)
# ---------------------------------------------------------------------------
def merge_reference(row_to_q, partial_o, partial_lse, nq: int) -> np.ndarray:
    """The reference merge with its quirks: m = max(0, max lse) (:307-309, :445),
    fp16 accumulation in partial-row order (:546), division afterwards (:416).
    (On a GPU the atomic order is nondeterministic; index order is what the
    Triton interpreter does.)"""
    Hq, P, D = partial_o.shape
    row_max = np.zeros((Hq, nq), dtype=np.float32)
    np.maximum.at(row_max, (slice(None), row_to_q), partial_lse)
    L = np.zeros((Hq, nq), dtype=np.float32)
    o = np.zeros((Hq, nq, D), dtype=np.float16)
    for i in range(P):
        qi = int(row_to_q[i])
        w = np.exp(partial_lse[:, i] - row_max[:, qi]).astype(np.float32)
        L[:, qi] += w
        o[:, qi] = (o[:, qi] + (w[:, None] * partial_o[:, i]).astype(np.float16)).astype(np.float16)
    o = (o / L[..., None].astype(np.float16)).astype(np.float16)
    return np.transpose(o, (1, 0, 2))  # [nq, Hq, D]


def merge_exact(row_to_q, partial_o, partial_lse, nq: int) -> np.ndarray:
    """Same merge with the true row max and fp32 accumulation, one fp16 rounding."""
    Hq, P, D = partial_o.shape
    row_max = np.full((Hq, nq), -np.inf, dtype=np.float32)
    np.maximum.at(row_max, (slice(None), row_to_q), partial_lse)
    w = np.exp(partial_lse - row_max[:, row_to_q])  # [Hq, P]
    L = np.zeros((Hq, nq), dtype=np.float32)
    np.add.at(L, (slice(None), row_to_q), w)
    o = np.zeros((Hq, nq, D), dtype=np.float32)
    np.add.at(o, (slice(None), row_to_q), w[..., None] * partial_o)
    with np.errstate(invalid="ignore", divide="ignore"):
        o = o / L[..., None]
    return np.transpose(o, (1, 0, 2)).astype(np.float16)


def flatten_forward(q, kv_data, md, merge: str = "exact") -> np.ndarray:
    po, pl = flatten_stage1(q, kv_data, md)
    fn = merge_exact if merge == "exact" else merge_reference
    return fn(md["block_q"], po, pl, q.shape[0])


def node_forward(q, kv_data, md, merge: str = "exact") -> np.ndarray:
    po, pl = node_stage1(q, kv_data, md)
    fn = merge_exact if merge == "exact" else merge_reference
    return fn(md["node_q"], po, pl, q.shape[0])


# ---------------------------------------------------------------------------
# ground truth: sequential attention per leaf (test_DeFT_kernel.py:212-276)
# ---------------------------------------------------------------------------
def sequential_truth(q, kv_data, paths: Sequence[Sequence[int]], dtype=np.float64) -> np.ndarray:
    """paths[i] = root->leaf pool slots of query row i.  Returns [nq, Hq, D] in `dtype`."""
    nq, Hq, D = q.shape
    kbuf, vbuf = _kv_views(kv_data)
    group = Hq // kbuf.shape[1]
    out = np.zeros((nq, Hq, D), dtype=dtype)
    scale = 1.0 / (D ** 0.5)
    for i, slots in enumerate(paths):
        slots = np.asarray(slots, dtype=np.int64)
        k = _expand_heads(kbuf[slots].astype(dtype), group)  # [Hq, S, D]
        v = _expand_heads(vbuf[slots].astype(dtype), group)
        s = np.einsum("hd,hsd->hs", q[i].astype(dtype), k) * scale
        s -= s.max(axis=1, keepdims=True)
        p = np.exp(s)
        p /= p.sum(axis=1, keepdims=True)
        out[i] = np.einsum("hs,hsd->hd", p, v)
    return out


def kv_append(kv_data: np.ndarray, cache_loc, k_new, v_new) -> None:
    """KVCacheUpdater.update, paged branch (tree_cache.py:70-76): index_put of one
    K row and one V row per leaf."""
    loc = np.asarray(cache_loc, dtype=np.int64)
    kv_data[loc, 0] = k_new
    kv_data[loc, 1] = v_new


# ---------------------------------------------------------------------------
# sequential comparator: token_attention_fwd
# (DeFT/deft/layers/attention/token_attention.py:297-335; stage 1 :12-80, stage 2 :83-150)
# ---------------------------------------------------------------------------
def token_attention_forward(q, kv_data, req_rows, b_seq_len, block_n: int = 64):
    """Restates the reference's two kernels bit for bit (pinned on tests/golden/seq_*.npz: max |diff| = 0).

    Stage 1 (:51-79): `q[None, :] * k` multiplies two fp16 tensors, so every product is rounded to fp16; `tl.sum`
    accumulates them in fp32 and hands back fp16; `att_value *= sm_scale` is again an fp16 multiply; the logits are
    stored in `att_m`, which has the dtype of q (:312-314).  Stage 2 (:121-146) walks each request's logits in blocks
    of 64 with an fp32 online softmax and fp32 P.V.  The fp16 logits cost the reference ~1.3e-3 against fp64 truth on
    the 40-leaf tree; the HIP path keeps fp32 logits, so it is compared to these vectors at 2.5e-3 and to the truth
    at 5e-4.  req_rows[i] = page-table row of request i (req_to_token[b_req_idx[i]])."""
    nq, Hq, D = q.shape
    kbuf, vbuf = _kv_views(kv_data)
    group = Hq // kbuf.shape[1]
    scale16 = np.float16(1.0 / (D ** 0.5))
    out = np.zeros((nq, Hq, D), dtype=np.float16)
    for i in range(nq):
        n = int(b_seq_len[i])
        slots = np.asarray(req_rows[i][:n], dtype=np.int64)
        k16 = np.repeat(np.transpose(kbuf[slots], (1, 0, 2)), group, axis=0)  # [Hq, n, D] fp16
        v = _expand_heads(vbuf[slots].astype(np.float32), group)
        prod = q[i][:, None, :] * k16  # fp16 x fp16 -> fp16
        dots = prod.astype(np.float32).sum(axis=2, dtype=np.float32).astype(np.float16)
        logits = (dots * scale16).astype(np.float16).astype(np.float32)
        e_max = np.full((Hq,), -np.inf, dtype=np.float32)
        e_sum = np.zeros((Hq,), dtype=np.float32)
        acc = np.zeros((Hq, D), dtype=np.float32)
        for s in range(0, n, block_n):
            qk = logits[:, s : s + block_n]
            n_e_max = np.maximum(qk.max(axis=1), e_max)
            old = np.exp(e_max - n_e_max).astype(np.float32)
            p = np.exp(qk - n_e_max[:, None]).astype(np.float32)
            e_sum = e_sum * old + p.sum(axis=1, dtype=np.float32)
            acc = acc * old[:, None] + np.einsum("hn,hnd->hd", p, v[:, s : s + block_n], dtype=np.float32)
            e_max = n_e_max
        out[i] = (acc / e_sum[:, None]).astype(np.float16)
    return out


# ---------------------------------------------------------------------------
# causal prefill: context_attention_fwd
# (DeFT/deft/layers/attention/context_flashattention_nopad.py:130-195; kernel :12-127)
# ---------------------------------------------------------------------------
def context_attention_forward(q, k, v, b_start_loc, b_seq_len, block: int = 128):
    """Restates the reference kernel: per (sequence, head, 128-query block) a loop over 128-key blocks with the
    lightllm-style online softmax that keeps `acc` NORMALISED after every block (:89-101: p is scaled by
    beta / l_new before it is rounded to fp16 for the PV dot, acc by l_old / l_new * alpha).  fp16 operands, fp32
    dots.  q [T, Hq, D], k / v [T, Hkv, D] fp16 -> out [T, Hq, D] fp16."""
    T, Hq, D = q.shape
    group = Hq // k.shape[1]
    scale = np.float32(1.0 / (D ** 0.5))
    out = np.zeros((T, Hq, D), dtype=np.float16)
    for b in range(len(b_seq_len)):
        s0, n = int(b_start_loc[b]), int(b_seq_len[b])
        for hq in range(Hq):
            kh = hq // group
            kk = k[s0 : s0 + n, kh].astype(np.float32)
            vv = v[s0 : s0 + n, kh].astype(np.float32)
            for m0 in range(0, n, block):
                rows = np.arange(m0, min(m0 + block, n))
                qq = q[s0 + rows, hq].astype(np.float32)
                m_i = np.full(len(rows), -np.inf, dtype=np.float32)
                l_i = np.zeros(len(rows), dtype=np.float32)
                acc = np.zeros((len(rows), D), dtype=np.float32)
                for n0 in range(0, m0 + block, block):
                    cols = np.arange(n0, min(n0 + block, n))
                    if len(cols) == 0:
                        break
                    qk = (qq @ kk[cols].T).astype(np.float32) * scale
                    qk = np.where(rows[:, None] >= cols[None, :], qk, -np.inf).astype(np.float32)
                    m_ij = qk.max(axis=1)
                    p = np.exp(qk - m_ij[:, None]).astype(np.float32)
                    l_ij = p.sum(axis=1, dtype=np.float32)
                    m_new = np.maximum(m_i, m_ij)
                    alpha = np.exp(m_i - m_new).astype(np.float32)
                    beta = np.exp(m_ij - m_new).astype(np.float32)
                    l_new = alpha * l_i + beta * l_ij
                    p = (p * (beta / l_new)[:, None]).astype(np.float16).astype(np.float32)
                    acc = acc * (l_i / l_new * alpha)[:, None] + p @ vv[cols]
                    l_i, m_i = l_new, m_new
                out[s0 + rows, hq] = acc.astype(np.float16)
    return out


def causal_truth(q, k, v, b_start_loc, b_seq_len):
    """fp64 causal attention per sequence (ground truth for the prefill path)."""
    T, Hq, D = q.shape
    group = Hq // k.shape[1]
    out = np.zeros((T, Hq, D), dtype=np.float64)
    for b in range(len(b_seq_len)):
        s0, n = int(b_start_loc[b]), int(b_seq_len[b])
        mask = np.tril(np.ones((n, n), dtype=bool))
        for hq in range(Hq):
            kh = hq // group
            s = q[s0 : s0 + n, hq].astype(np.float64) @ k[s0 : s0 + n, kh].astype(np.float64).T / np.sqrt(D)
            s = np.where(mask, s, -np.inf)
            p = np.exp(s - s.max(axis=1, keepdims=True))
            out[s0 : s0 + n, hq] = (p / p.sum(axis=1, keepdims=True)) @ v[s0 : s0 + n, kh].astype(np.float64)
    return out
