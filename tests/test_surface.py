"""The drop-in surface, checked mechanically (VERDICT r4 "what's missing" #4): every public callable, dataclass field, enum
member and configuration key of the reference modules the path's callers import -- recorded from the reference by
tools/gen_surface.py into tests/golden/surface.json (names, parameter kinds, defaults; no source) -- against deft_amd's
counterparts.  A counterpart must accept every call the reference's accepts: the same parameters, in the same order, with
the same names and defaults; what it takes BEYOND that must be optional and is listed here, by name, in EXTRA.  What the
package does not cover is listed in OUT_OF_SCOPE with the reason, so a name that silently went missing fails."""
import ast
import dataclasses
import inspect
import json
import os

import pytest

import deft_amd
import deft_amd.branch_func_example
import deft_amd.context_attention
import deft_amd.data_loader
import deft_amd.deft_attention
import deft_amd.forward_mode
import deft_amd.memory_pool
import deft_amd.rotary_embedding
import deft_amd.token_attention
import deft_amd.tree_attention
import deft_amd.tree_cache
import deft_amd.tree_generate

SURFACE = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "surface.json")))

COUNTERPART = {
    "deft.layers.attention.tree_attention": deft_amd.tree_attention,
    "deft.layers.attention.deft_attention": deft_amd.deft_attention,
    "deft.layers.attention.token_attention": deft_amd.token_attention,
    "deft.layers.attention.context_flashattention_nopad": deft_amd.context_attention,
    "deft.layers.rotary_embedding": deft_amd.rotary_embedding,
    "deft.tree_decoding.tree_cache": deft_amd.tree_cache,
    "deft.memory_pool": deft_amd.memory_pool,
    "deft.model_runner": deft_amd.forward_mode,
    "deft.data_loader": deft_amd.data_loader,
    "deft.tree_decoding.generation.branch_func_example": deft_amd.branch_func_example,
    "deft.tree_decoding.branch_controller": deft_amd.branch_func_example,
    "deft.tree_decoding.generation.tree_generate": deft_amd.tree_generate,
}

# optional parameters deft_amd's counterparts take beyond the reference's (each must have a default)
EXTRA = {
    "deft.tree_decoding.tree_cache.TreeMetadata.from_tree_cache": {"device", "copy", "device_build"},
    "deft.memory_pool.ReqToTokenPool.__init__": {"device"},
    "deft.memory_pool.TokenToKVPool.__init__": {"device"},
    "deft.layers.attention.deft_attention.DeFTAttention.forward": {"rotary_emb", "positions", "fuse_rope"},
}

OUT_OF_SCOPE = {
    # --mem unpaged modes (SURVEY §2 "OUT OF SCOPE")
    "deft.layers.attention.deft_attention.DeFTAttention.deft_flatten_unpaged_forward": "unpaged",
    "deft.layers.attention.deft_attention.DeFTAttention.deft_node_unpaged_forward": "unpaged",
    "deft.layers.attention.deft_attention.DeFTAttention.flash_decoding_unpaged_forward": "unpaged",
    "deft.layers.attention.deft_attention.DeFTAttention.medusa_unpaged_forward": "unpaged",
    "deft.tree_decoding.tree_cache.TreeCache.get_kv_seq": "unpaged",
    "deft.tree_decoding.tree_cache.TreeCache.get_kv_tree": "unpaged",
    "deft.tree_decoding.tree_cache.TreeCache.get_kv_tree_with_mask": "unpaged",
    # --mode tree_index
    "deft.tree_decoding.tree_cache.TreeMetadata.from_tree_cache_node": "tree_index",
    # the model runner's batch bookkeeping for prefill / extend: model side, not the attention path
    "deft.model_runner.InputMetadata.create": "model runner",
    # the CUDA custom-op spelling of forward_native (vllm _C ops); forward() is the entry point
    "deft.layers.rotary_embedding.RotaryEmbedding.forward_cuda": "cuda custom op",
    # internals of the template loader: deft_amd.templates keeps a template as arrays, not as linked node objects
    "deft.data_loader.build_tree": "loader internals", "deft.data_loader.build_trees": "loader internals",
    "deft.data_loader.build_tree_SD": "loader internals", "deft.data_loader.ExecuteTreeNode.__init__": "loader internals",
    "deft.data_loader.ExecuteTree.__init__": "loader internals", "deft.data_loader.ExecuteTree.build_tree_metadata": "loader internals",
}


def _simple(default_src):
    try:
        return ast.literal_eval(default_src)
    except Exception:
        return default_src  # an expression: compared as text below


def _check(qual, ref_params, obj):
    ref = [p for p in ref_params if p["name"] not in ("self", "cls")]
    got = [p for p in inspect.signature(obj).parameters.values() if p.name not in ("self", "cls")]
    assert len(got) >= len(ref), f"{qual}: takes {len(got)} parameters, the reference's takes {len(ref)}"
    kinds = {inspect.Parameter.POSITIONAL_ONLY: "posonly", inspect.Parameter.POSITIONAL_OR_KEYWORD: "pos",
             inspect.Parameter.VAR_POSITIONAL: "vararg", inspect.Parameter.KEYWORD_ONLY: "kwonly", inspect.Parameter.VAR_KEYWORD: "varkw"}
    for i, r in enumerate(ref):
        g = got[i]
        assert g.name == r["name"], f"{qual}: parameter {i} is `{g.name}`, the reference's is `{r['name']}`"
        assert kinds[g.kind] == r["kind"], f"{qual}: `{g.name}` is {kinds[g.kind]}, the reference's is {r['kind']}"
        has = g.default is not inspect.Parameter.empty
        assert has == (r["default"] is not None), f"{qual}: `{g.name}` default presence differs"
        if has:
            want = _simple(r["default"])
            assert g.default == want or repr(g.default) == r["default"], f"{qual}: `{g.name}` defaults to {g.default!r}, the reference's to {r['default']}"
    extra = {p.name for p in got[len(ref):]}
    assert extra == EXTRA.get(qual, set()), f"{qual}: extra parameters {sorted(extra)} (listed: {sorted(EXTRA.get(qual, set()))})"
    for p in got[len(ref):]:
        assert p.default is not inspect.Parameter.empty or p.kind in (p.VAR_POSITIONAL, p.VAR_KEYWORD), f"{qual}: extra `{p.name}` is not optional"


def _cases():
    for mod, e in SURFACE.items():
        for name, params in e["functions"].items():
            yield f"{mod}.{name}", mod, (name,), params
        for cname, c in e["classes"].items():
            for mname, m in c["methods"].items():
                yield f"{mod}.{cname}.{mname}", mod, (cname, mname), m["params"]


@pytest.mark.parametrize("qual,mod,path,params", list(_cases()), ids=[c[0] for c in _cases()])
def test_callable_accepts_the_references_calls(qual, mod, path, params):
    if qual in OUT_OF_SCOPE:
        obj = COUNTERPART[mod]
        for p in path:
            obj = getattr(obj, p, None)
        if obj is None:
            pytest.skip(f"out of scope: {OUT_OF_SCOPE[qual]}")
        return  # (present although out of scope: nothing to hold it to)
    obj = COUNTERPART[mod]
    for p in path:
        assert hasattr(obj, p), f"deft_amd has no counterpart of {qual}"
        obj = inspect.getattr_static(obj, p) if inspect.isclass(obj) else getattr(obj, p)
    if isinstance(obj, (classmethod, staticmethod)):
        obj = obj.__func__
    _check(qual, params, obj)


def test_out_of_scope_list_names_real_things():
    known = {c[0] for c in _cases()}
    assert set(OUT_OF_SCOPE) <= known and set(EXTRA) <= known


def test_tree_metadata_fields_and_block_config():
    ref = SURFACE["deft.tree_decoding.tree_cache"]["classes"]["TreeMetadata"]
    assert ref["dataclass"] and dataclasses.is_dataclass(deft_amd.TreeMetadata)
    got = [f.name for f in dataclasses.fields(deft_amd.TreeMetadata)]
    want = [f["name"] for f in ref["fields"]]
    assert got[: len(want)] == want, "TreeMetadata's fields differ from the reference's (tree_cache.py:591-617)"
    for f in dataclasses.fields(deft_amd.TreeMetadata)[len(want):]:  # anything more must be optional
        assert f.default is not dataclasses.MISSING or f.default_factory is not dataclasses.MISSING
    assert deft_amd.BLOCK_CONFIG == ast.literal_eval(SURFACE["deft.tree_decoding.tree_cache"]["constants"]["BLOCK_CONFIG"])


def test_forward_mode_members_and_input_metadata_fields():
    ref = SURFACE["deft.model_runner"]["classes"]
    assert [m.name for m in deft_amd.ForwardMode] == ref["ForwardMode"]["members"]
    ref_fields = {f["name"] for f in ref["InputMetadata"]["fields"]}
    got = {f.name for f in dataclasses.fields(deft_amd.InputMetadata)}
    # the slice of InputMetadata the attention module reads: every field exists under the same name in the reference's
    assert got <= ref_fields, got - ref_fields
    assert {"forward_mode", "kv_updater", "token_to_kv_pool", "req_to_token_pool", "req_pool_indices", "start_loc", "seq_lens",
            "max_seq_len", "total_num_tokens", "other_kv_index"} <= got


def test_template_objects_answer_to_the_references_attribute_names():
    """ExecuteTree's attributes as the branch functions and the example script read them (data_loader.py:30-49;
    branch_func_example.py:304-311, :385-388; run_DeFT_llama_paged.py:245-263)."""
    from deft_amd.data_loader import ExecuteTree
    from deft_amd.templates import synthetic_reasoning_template, synthetic_speculative_template

    t = synthetic_reasoning_template(widths=(2, 2), lens=(3, 2))
    assert isinstance(t, ExecuteTree)
    for attr in ("root", "nodes", "prompt", "branch_record", "prune_record", "max_depth", "max_width", "width_per_depth", "node_num",
                 "accepted_len_list"):
        assert hasattr(t, attr), attr
    assert t.root is t.nodes[0] and t.root.id == 0 and len(t.nodes) == t.node_num
    n = t.nodes[1]
    assert (n.id, n.value, n.start_offset, n.end_offset, n.depth) == (1, 3, 1, 3, 1) and [c.id for c in t.root.children] == [1, 2]
    sd = synthetic_speculative_template(tree_size=4, steps=3)
    assert sd.accepted_len_list == sd.accept_lengths
    sd.accepted_len_list = [1, 2]
    assert sd.accept_lengths == [1, 2]
