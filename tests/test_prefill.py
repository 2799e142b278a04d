"""Causal prefill attention (deft_prefill_f16 / deft_amd.context_attention_fwd).

tests/golden/prefill.npz = the reference's context_attention_fwd under the Triton interpreter
(tools/gen_golden_prefill.py).  The oracle restates that kernel bit-exactly (its per-block renormalisation with P
rounded to fp16 AFTER scaling costs the reference ~1.4e-3 against fp64 truth); the HIP kernel accumulates unnormalised
like the decode path, so it is compared to the reference's vectors at 2.5e-3 and to the truth at 5e-4."""
import os

import numpy as np
import pytest
import torch

import deft_amd
from deft_amd.utils.synthetic import dyadic_normal
from oracle import attention as oa

def _close_to_truth(out, truth):
    """5e-4 absolute plus half an fp16 ulp of the value: the first tokens of a sequence attend to one or two keys, so
    their outputs are V rows of magnitude 1..4, where rounding to fp16 alone costs up to 9.8e-4."""
    err = np.abs(out.astype(np.float64) - truth)
    return bool((err <= 5e-4 + np.abs(truth) * 2.0 ** -11).all())


GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "prefill.npz"))
CASES = ("single_300", "single_513_gqa", "batch_ragged", "d64_gqa_385", "d64_batch_ragged", "d32_gqa_200", "d16_150")


def _inputs(name):
    lens = GOLD[name + "_lens"]
    Hq, Hkv, D = (int(x) for x in GOLD[name + "_geom"])
    T = int(lens.sum())
    q, k, v = dyadic_normal((T, Hq, D), 101), dyadic_normal((T, Hkv, D), 102), dyadic_normal((T, Hkv, D), 103)
    start = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int32)
    return q, k, v, start, lens.astype(np.int32)


@pytest.mark.parametrize("name", CASES)
def test_oracle_is_bit_exact_on_reference_outputs(name):
    q, k, v, start, lens = _inputs(name)
    o = oa.context_attention_forward(q, k, v, start, lens)
    rows = GOLD[name + "_rows"]
    assert np.array_equal(o[rows], GOLD[name + "_o"])
    assert np.abs(oa.causal_truth(q, k, v, start, lens)[rows] - o[rows].astype(np.float64)).max() < 2.5e-3


def test_operator_refuses_cpu_tensors():
    q = torch.zeros(4, 2, 128, dtype=torch.float16)
    with pytest.raises(deft_amd.DeftLibraryError, match="no CPU path"):
        deft_amd.context_attention_fwd(q, q, q, q.clone(), torch.zeros(1, dtype=torch.int32), torch.tensor([4], dtype=torch.int32), 4)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_prefill_matches_reference_and_truth(name):
    q, k, v, start, lens = _inputs(name)
    o = torch.full(q.shape, float("nan"), dtype=torch.float16, device="cuda")
    deft_amd.context_attention_fwd(torch.from_numpy(q).cuda(), torch.from_numpy(k).cuda(), torch.from_numpy(v).cuda(), o,
                                   torch.from_numpy(start).cuda(), torch.from_numpy(lens).cuda(), int(lens.max()))
    torch.cuda.synchronize()
    out = o.cpu().numpy()
    assert np.isfinite(out.astype(np.float32)).all()
    rows = GOLD[name + "_rows"]
    assert np.abs(out[rows].astype(np.float64) - GOLD[name + "_o"].astype(np.float64)).max() < 2.5e-3
    assert _close_to_truth(out, oa.causal_truth(q, k, v, start, lens))


@pytest.mark.gpu
@pytest.mark.parametrize("lens,geom", [([1], (4, 4, 128)), ([255, 256, 257], (8, 2, 128)), ([1000], (32, 32, 128)),
                                       ([640, 3], (32, 8, 128)), ([1], (4, 4, 64)), ([255, 256, 257], (8, 2, 64)),
                                       ([1000, 130], (32, 8, 64)), ([257, 1, 64], (4, 2, 32)), ([300], (2, 1, 16))])
def test_gpu_prefill_edges_and_strided_views(lens, geom):
    """Block edges (255/256/257), a single token, GQA, and q/k/v as strided views of the fused qkv (llama2.py:108-109)."""
    Hq, Hkv, D = geom
    T = sum(lens)
    qkv_np = dyadic_normal((T, (Hq + 2 * Hkv) * D), 7)
    qkv = torch.from_numpy(qkv_np).cuda()
    q, k, v = (t.view(T, -1, D) for t in qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1))
    start = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int32)
    o = torch.full((T, Hq, D), float("nan"), dtype=torch.float16, device="cuda")
    deft_amd.context_attention_fwd(q, k, v, o, torch.from_numpy(start).cuda(), torch.tensor(lens, dtype=torch.int64).cuda(), max(lens))
    torch.cuda.synchronize()
    qn = qkv_np[:, : Hq * D].reshape(T, Hq, D)
    kn = qkv_np[:, Hq * D: (Hq + Hkv) * D].reshape(T, Hkv, D)
    vn = qkv_np[:, (Hq + Hkv) * D:].reshape(T, Hkv, D)
    truth = oa.causal_truth(qn, kn, vn, start, np.asarray(lens))
    assert _close_to_truth(o.cpu().numpy(), truth)


@pytest.mark.gpu
def test_prefill_then_decode_through_the_module():
    """ForwardMode.PREFILL stores the prompt's K/V in the pool; a sequential decode step over that pool then equals
    causal attention of one more token."""
    Hq, Hkv, D, n = 8, 2, 128, 200
    req = deft_amd.ReqToTokenPool(8, 512, device="cuda")
    pool = deft_amd.TokenToKVPool(512, torch.float16, Hkv, D, 1, device="cuda")
    tree = deft_amd.TreeCache(torch.float16, Hkv, D, 1, req, pool, None, True, False)
    updater = tree.init_prompt(torch.arange(1, n + 1, dtype=torch.int32))
    x = dyadic_normal((n + 1, (Hq + 2 * Hkv) * D), 9)
    qkv = torch.from_numpy(x).cuda()
    q, k, v = qkv[:n].split([Hq * D, Hkv * D, Hkv * D], dim=-1)
    meta = deft_amd.InputMetadata(deft_amd.ForwardMode.PREFILL, updater, pool, req,
                                  start_loc=torch.zeros(1, dtype=torch.int32, device="cuda"),
                                  seq_lens=torch.tensor([n], dtype=torch.int64, device="cuda"), max_seq_len=n)
    attn = deft_amd.DeFTAttention(Hq, D, D ** -0.5, Hkv, 0)
    out = attn(q, k, v, meta)
    torch.cuda.synchronize()
    qn, kn, vn = (x[:, a:b].reshape(n + 1, -1, D) for a, b in ((0, Hq * D), (Hq * D, (Hq + Hkv) * D), ((Hq + Hkv) * D, None)))
    truth = oa.causal_truth(qn[:n], kn[:n], vn[:n], [0], [n])
    assert _close_to_truth(out.view(n, Hq, D).cpu().numpy(), truth)
    slots = tree.root.kv_indices
    assert np.array_equal(pool.kv_data[0][slots, 0].cpu().numpy(), kn[:n]) and np.array_equal(pool.kv_data[0][slots, 1].cpu().numpy(), vn[:n])


@pytest.mark.gpu
def test_prefill_full_size_properties():
    """Llama-2-7B geometry, 4096-token prompt (the north-star prefix): the last token's row equals decode attention of
    that token over the prompt's K/V (the Node operator on one entry), rows only depend on earlier tokens (changing
    the second half of K/V leaves the first half of the output bit-identical), and the operator is linear in V."""
    Hq, Hkv, D, S = 32, 32, 128, 4096
    g = torch.Generator(device="cuda").manual_seed(11)
    q = torch.randn((S, Hq, D), dtype=torch.float16, device="cuda", generator=g)
    k = torch.randn((S, Hkv, D), dtype=torch.float16, device="cuda", generator=g)
    v = torch.randn((S, Hkv, D), dtype=torch.float16, device="cuda", generator=g)
    start = torch.zeros(1, dtype=torch.int32, device="cuda")
    lens = torch.tensor([S], dtype=torch.int32, device="cuda")

    def run(kk, vv):
        o = torch.full_like(q, float("nan"))
        deft_amd.context_attention_fwd(q, kk, vv, o, start, lens, S)
        return o

    o = run(k, v)
    torch.cuda.synchronize()
    assert torch.isfinite(o.float()).all()
    # decode attention of the last token over all S keys
    idx = torch.arange(S, device="cuda")
    one = torch.zeros(1, dtype=torch.int64, device="cuda")
    o_dec = torch.empty((1, Hq, D), dtype=torch.float16, device="cuda")
    deft_amd.tree_attention_fwd(q[S - 1:], k, v, o_dec, idx, one, torch.tensor([S], device="cuda"), one, one, one + 1)
    torch.cuda.synchronize()
    assert (o[S - 1].float() - o_dec[0].float()).abs().max().item() < 5e-4
    # causality: garbage in the second half of K/V cannot reach the first half of the output
    k2, v2 = k.clone(), v.clone()
    k2[S // 2:] = 7.0
    v2[S // 2:] = -3.0
    o2 = run(k2, v2)
    torch.cuda.synchronize()
    assert torch.equal(o2[: S // 2], o[: S // 2])
    # linearity in V (exactly representable scaling): attention(2 V) = 2 attention(V) up to fp16 rounding of the output
    o3 = run(k, v * 2)
    torch.cuda.synchronize()
    assert (o3.float() - 2 * o.float()).abs().max().item() < 2e-3


@pytest.mark.gpu
def test_gpu_prefill_random_ragged_batches():
    rng = np.random.default_rng(5)
    for trial in range(6):
        Hq, Hkv = [(8, 8), (8, 2), (16, 1)][trial % 3]
        D = 128
        lens = [int(x) for x in rng.integers(1, 700, size=int(rng.integers(1, 6)))]
        T = sum(lens)
        q, k, v = dyadic_normal((T, Hq, D), 20 + trial), dyadic_normal((T, Hkv, D), 40 + trial), dyadic_normal((T, Hkv, D), 60 + trial)
        start = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int32)
        o = torch.full((T, Hq, D), float("nan"), dtype=torch.float16, device="cuda")
        deft_amd.context_attention_fwd(torch.from_numpy(q).cuda(), torch.from_numpy(k).cuda(), torch.from_numpy(v).cuda(), o,
                                       torch.from_numpy(start).cuda(), torch.tensor(lens, dtype=torch.int32).cuda(), max(lens))
        torch.cuda.synchronize()
        assert _close_to_truth(o.cpu().numpy(), oa.causal_truth(q, k, v, start, np.asarray(lens))), (trial, lens)
