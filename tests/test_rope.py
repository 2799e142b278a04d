"""Rotary embedding in front of the path (deft_rope_qk_f16 / deft_amd.RotaryEmbedding).

flashinfer, which the reference's decode path calls for this op (rotary_embedding.py:31, :157-177), is not part of the
reference tree: parity is against the restated algorithm (oracle/rope.py, fp32 arithmetic) -- bit-exact -- and
against an fp64 evaluation of the same rotation (<= 1 fp16 ulp)."""
import numpy as np
import pytest
import torch

import deft_amd
from deft_amd.utils.synthetic import dyadic_normal
from oracle import rope as orope


def test_oracle_cache_matches_the_reference_formula():
    rot, maxpos, base = 128, 512, 10000.0
    cache = orope.cos_sin_cache(rot, maxpos, base)
    inv = 1.0 / (base ** (torch.arange(0, rot, 2, dtype=torch.float) / rot))  # rotary_embedding.py:109-116
    freqs = torch.einsum("i,j -> ij", torch.arange(maxpos, dtype=torch.float), inv)
    ref = torch.cat((freqs.cos(), freqs.sin()), dim=-1).numpy()
    assert np.abs(cache - ref).max() < 1e-4  # fp32 angles up to 511 rad: numpy's and torch's cos differ by ulps of the ANGLE
    mod = deft_amd.RotaryEmbedding(128, rot, maxpos, base, True, torch.float32)
    assert np.array_equal(mod.cos_sin_cache.numpy(), ref)


@pytest.mark.parametrize("neox", [True, False])
def test_oracle_rotation_is_the_rotation(neox):
    n, H, D = 5, 3, 128
    x = dyadic_normal((n, H, D), 11)
    pos = np.array([0, 1, 17, 300, 511])
    cache = orope.cos_sin_cache(D, 512)
    out = orope.apply_rope(x, pos, cache, D, neox).astype(np.float64)
    xf = x.astype(np.float64)
    ang = pos[:, None] * (1.0 / (10000.0 ** (np.arange(0, D, 2) / D)))[None, :]
    c, s = np.cos(ang)[:, None, :], np.sin(ang)[:, None, :]
    x1, x2 = (xf[..., : D // 2], xf[..., D // 2:]) if neox else (xf[..., 0::2], xf[..., 1::2])
    o1, o2 = x1 * c - x2 * s, x2 * c + x1 * s
    ref = np.concatenate([o1, o2], -1) if neox else np.stack([o1, o2], -1).reshape(n, H, D)
    assert np.abs(out - ref).max() < 4e-3  # fp16 rounding of values up to ~4, fp32 cache angles up to 511 rad
    norm_in = (xf ** 2).sum(-1)
    assert np.abs((out ** 2).sum(-1) - norm_in).max() / norm_in.max() < 2e-3  # a rotation keeps the norm


def test_module_refuses_cpu_tensors_and_scaled_variants():
    mod = deft_amd.get_rope(128, 128, 64, 10000.0)
    q = torch.zeros(2, 4 * 128, dtype=torch.float16)
    with pytest.raises(deft_amd.DeftLibraryError, match="no CPU path"):
        mod(torch.zeros(2, dtype=torch.int64), q, q.clone())
    with pytest.raises(NotImplementedError):
        deft_amd.get_rope(128, 128, 64, 10000.0, rope_scaling={"type": "linear", "factor": 2.0})


@pytest.mark.gpu
@pytest.mark.parametrize("Hq,Hkv,D,rot,neox", [(32, 32, 128, 128, True), (32, 8, 128, 128, True), (8, 2, 64, 64, True),
                                               (4, 4, 128, 64, True), (4, 4, 128, 128, False)])
def test_gpu_rope_is_bit_exact_on_fused_qkv_views(Hq, Hkv, D, rot, neox):
    n, maxpos = 37, 4400
    qkv_np = dyadic_normal((n, (Hq + 2 * Hkv) * D), 5)
    pos_np = np.random.default_rng(0).integers(0, maxpos, size=n)
    qkv = torch.from_numpy(qkv_np).cuda()
    q, k, v = qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)  # strided views, as llama2.py:108-109 hands them over
    mod = deft_amd.RotaryEmbedding(D, rot, maxpos, 10000.0, neox, torch.float32).cuda()
    q2, k2 = mod(torch.from_numpy(pos_np).cuda(), q, k)
    torch.cuda.synchronize()
    assert q2.data_ptr() == q.data_ptr() and k2.data_ptr() == k.data_ptr()  # in place
    cache = mod.cos_sin_cache.cpu().numpy()
    q_ref = orope.apply_rope(qkv_np[:, : Hq * D].reshape(n, Hq, D), pos_np, cache, rot, neox)
    k_ref = orope.apply_rope(qkv_np[:, Hq * D: (Hq + Hkv) * D].reshape(n, Hkv, D), pos_np, cache, rot, neox)
    out = qkv.cpu().numpy()
    assert np.array_equal(out[:, : Hq * D].reshape(n, Hq, D), q_ref)
    assert np.array_equal(out[:, Hq * D: (Hq + Hkv) * D].reshape(n, Hkv, D), k_ref)
    assert np.array_equal(out[:, (Hq + Hkv) * D:], qkv_np[:, (Hq + Hkv) * D:])  # v untouched


# ---- pinned on the reference's own rotary embedding (tools/gen_golden_rope.py: get_rope + forward_native run in the build
#      container) -----------------------------------------------------------------------------------------------------------
def _rope_cases():
    import os

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "rope.npz"))
    for ci, row in enumerate(g["cases"]):
        Hq, Hkv, D, rot, neox, maxpos, base = int(row[0]), int(row[1]), int(row[2]), int(row[3]), bool(row[4]), int(row[5]), float(row[6])
        yield ci, g, (Hq, Hkv, D, rot, neox, maxpos, base)


def _rope_inputs(ci, Hq, Hkv, D, n):
    return dyadic_normal((n, Hq * D), 100 + ci), dyadic_normal((n, Hkv * D), 200 + ci)


def test_oracle_rope_is_bit_exact_on_reference_outputs():
    """oracle.rope.apply_rope on the reference's cache rows == the reference's forward_native in fp32 arithmetic, bit for bit;
    the reference's fp16-arithmetic result (cos / sin and products rounded to fp16) is within a few fp16 ulps of it."""
    for ci, g, (Hq, Hkv, D, rot, neox, maxpos, base) in _rope_cases():
        pos = g[f"pos_{ci}"]
        n = len(pos)
        q, k = _rope_inputs(ci, Hq, Hkv, D, n)
        full = np.zeros((maxpos, rot), dtype=np.float32)
        full[pos] = g[f"cache_{ci}"]
        for x, H, name in ((q, Hq, "q"), (k, Hkv, "k")):
            out = orope.apply_rope(x.reshape(n, H, D), pos, full, rot, neox).reshape(n, H * D)
            assert np.array_equal(out, g[f"{name}_f32_{ci}"]), (ci, name)
            assert np.abs(out.astype(np.float32) - g[f"{name}_f16_{ci}"].astype(np.float32)).max() < 1.2e-2
        # the product module builds the same cache as the reference's _compute_cos_sin_cache, bit for bit
        mod = deft_amd.RotaryEmbedding(D, rot, maxpos, base, neox, torch.float32)
        assert np.array_equal(mod.cos_sin_cache.numpy()[pos], g[f"cache_{ci}"])
        assert np.abs(orope.cos_sin_cache(rot, maxpos, base)[pos] - g[f"cache_{ci}"]).max() < 2e-3  # numpy vs torch cos of fp32 angles


@pytest.mark.gpu
def test_gpu_rope_is_bit_exact_on_reference_outputs():
    for ci, g, (Hq, Hkv, D, rot, neox, maxpos, base) in _rope_cases():
        pos = g[f"pos_{ci}"]
        n = len(pos)
        q_np, k_np = _rope_inputs(ci, Hq, Hkv, D, n)
        q, k = torch.from_numpy(q_np).cuda(), torch.from_numpy(k_np).cuda()
        mod = deft_amd.get_rope(D, rot, maxpos, base, neox).cuda()
        mod(torch.from_numpy(pos).cuda(), q, k)
        torch.cuda.synchronize()
        assert np.array_equal(q.cpu().numpy(), g[f"q_f32_{ci}"]), ci
        assert np.array_equal(k.cpu().numpy(), g[f"k_f32_{ci}"]), ci


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["flatten", "node"])
@pytest.mark.parametrize("name,geom", [("multilevel", (8, 2, 128)), ("spec_mock", (4, 4, 128)), ("after_cut", (32, 8, 128)), ("wide40", (4, 4, 128))])
def test_fused_rope_equals_rope_then_attention(name, geom, mode):
    """rotary embedding + paged append + attention in ONE stage-1 launch
    (deft_{flatten,node}_decode_rope_append_f16, DeFTAttention.forward(..., rotary_emb=, positions=)) is bit-identical --
    outputs AND pool bytes -- to the reference's sequence rotary_emb(positions, q, k) -> attn(q, k, v)
    (llama2.py:108-111), q / k being rows of a fused qkv projection; the fused form leaves q and k unrotated."""
    from product_helpers import product_tree
    Hq, Hkv, D = geom
    outs, pools = [], []
    rope = deft_amd.get_rope(D, D, 2048, 10000.0, True)
    for fused in (False, True):
        tree = product_tree(name, device="cuda", heads=(Hkv, D))
        for leaf in list(tree.leaves.values()):
            leaf.append_token(9)
        updater = tree.alloc()
        md = deft_amd.TreeMetadata.from_tree_cache(tree)
        nq = md.query_num
        g = torch.Generator(device="cuda")
        g.manual_seed(5)
        pool = tree.token_to_kv_pool
        pool.kv_data[0].normal_(generator=g)
        qkv = torch.randn((nq, (Hq + 2 * Hkv) * D), dtype=torch.float16, device="cuda", generator=g)
        q, k, v = qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)  # strided views, like llama2.py:107
        # positions: the length of every leaf's path (distinct values, some far apart)
        positions = torch.tensor([len(tree.leaf_path_slots(lf)) - 1 + 37 * i for i, lf in
                                  enumerate(sorted(tree.leaves.values(), key=lambda n: n.id))], dtype=torch.int64, device="cuda")
        assert positions.shape[0] == nq
        before = qkv.clone()
        attn = deft_amd.DeFTAttention(Hq, D, D ** -0.5, Hkv, layer_id=0)
        fm = deft_amd.ForwardMode.TREE_DECODE_FLATTEN if mode == "flatten" else deft_amd.ForwardMode.TREE_DECODE_NODE
        meta = deft_amd.InputMetadata(fm, updater, pool)
        deft_amd.register_tree_metadata(md)
        try:
            if fused:
                o = attn(q, k, v, meta, rotary_emb=rope, positions=positions, fuse_rope=True)
                torch.cuda.synchronize()
                assert torch.equal(qkv, before)  # nothing rotated in place
            else:
                rope(positions, q, k)
                o = attn(q, k, v, meta)
                torch.cuda.synchronize()
                assert not torch.equal(qkv, before)
        finally:
            deft_amd.unregister_tree_metadata()
        outs.append(o.cpu())
        pools.append(pool.kv_data[0].cpu())
    assert torch.equal(pools[0], pools[1])
    assert torch.equal(outs[0], outs[1])
    assert torch.isfinite(outs[1].float()).all() and outs[1].float().abs().max() > 0


@pytest.mark.gpu
def test_fused_rope_refuses_what_it_does_not_cover():
    """GPT-J pairing / partial rotary dims are not fused: the C ABI says DEFT_EUNSUPPORTED, the module runs the rotation
    as its own launch first (same results as the reference's sequence by construction)."""
    i64 = torch.zeros(256, dtype=torch.int64, device="cuda")
    f16 = torch.zeros(4 * 128 * 8, dtype=torch.float16, device="cuda")
    f32 = torch.zeros(1024, dtype=torch.float32, device="cuda")
    i32 = torch.zeros(8, dtype=torch.int32, device="cuda")
    ws = torch.zeros(1 << 22, dtype=torch.uint8, device="cuda")
    def call(rot, neox):
        return deft_amd.lib.deft_flatten_decode_rope_append_f16(
            f16.data_ptr(), 512, 128, f16.data_ptr(), f16.data_ptr(), 1024, 128, f16.data_ptr(), 512, 128,
            *[i64.data_ptr()] * 6, 1, 1, 1, 4, 4, 128, 128 ** -0.5, i32.data_ptr(), f16.data_ptr(), f16.data_ptr(), 512, 1,
            f32.data_ptr(), rot, neox, None, ws.data_ptr(), ws.numel(), None)
    assert call(64, 1) == -2 and call(128, 0) == -2  # DEFT_EUNSUPPORTED
