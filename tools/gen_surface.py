#!/usr/bin/env python3
"""The reference's operator SURFACE for the hot path, as data: tests/golden/surface.json.

`north_star` says the path "drops into run_DeFT_llama_paged.py unchanged".  This script (build container only) records
what "unchanged" means -- the signature of every public callable, the dataclass fields, the enum members and the
configuration keys of the reference modules the path's callers import:

  DeFT/deft/layers/attention/tree_attention.py      tree_attention_fwd (:14-25), tree_attention_subtree_fwd (:551-568)
  DeFT/deft/layers/attention/deft_attention.py      DeFTAttention (:33-48, :72, :110, :349, :390)
  DeFT/deft/layers/attention/token_attention.py     token_attention_fwd
  DeFT/deft/layers/attention/context_flashattention_nopad.py   context_attention_fwd
  DeFT/deft/layers/rotary_embedding.py              RotaryEmbedding, get_rope
  DeFT/deft/tree_decoding/tree_cache.py             KVCacheUpdater, TreeNode, TreeCache, TreeMetadata, the registries (:52-130, :147-403, :591-623, :1025-1053)
  DeFT/deft/memory_pool.py                          ReqToTokenPool, TokenToKVPool (:11-108)
  DeFT/deft/model_runner.py                         ForwardMode members, InputMetadata fields (:31-42, :73-94)
  DeFT/deft/data_loader.py                          ExecuteTree, load_trees, load_prompts, generate_accepted_len_list
  DeFT/deft/tree_decoding/generation/branch_func_example.py, branch_controller.py   the three branch functions, Branch_Controller

Every module is read with `ast` (several cannot be imported without a GPU or flashinfer); where a module DOES import here
its callables are also read with `inspect.signature` and the two readings must agree.  The output holds names, parameter
kinds and default values -- no source text.  tests/test_surface.py checks deft_amd's counterparts against it.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/gen_surface.py
"""
from __future__ import annotations

import ast
import importlib
import inspect
import json
import os
import sys

sys.dont_write_bytecode = True
os.environ.setdefault("TRITON_INTERPRET", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/DeFT"
sys.path.insert(0, REF)

MODULES = {
    "deft.layers.attention.tree_attention": ("functions", ["tree_attention_fwd", "tree_attention_subtree_fwd"]),
    "deft.layers.attention.deft_attention": ("classes", ["DeFTAttention"]),
    "deft.layers.attention.token_attention": ("functions", ["token_attention_fwd"]),
    "deft.layers.attention.context_flashattention_nopad": ("functions", ["context_attention_fwd"]),
    "deft.layers.rotary_embedding": ("mixed", ["RotaryEmbedding", "get_rope"]),
    "deft.tree_decoding.tree_cache": ("mixed", ["KVCacheUpdater", "TreeNode", "TreeCache", "TreeMetadata", "register_tree_metadata",
                                                 "unregister_tree_metadata", "get_global_tree_metadata", "register_tree_cache",
                                                 "unregister_tree_cache", "get_global_tree_cache", "BLOCK_CONFIG"]),
    "deft.memory_pool": ("classes", ["ReqToTokenPool", "TokenToKVPool"]),
    "deft.model_runner": ("classes", ["ForwardMode", "InputMetadata"]),
    "deft.tree_decoding.generation.branch_func_example": ("functions", ["example_branch_Func1_SimpleTree",
                                                                         "example_branch_Func3_FromTreeTemplate",
                                                                         "example_branch_Func4_SpeculativeDecoding"]),
    "deft.tree_decoding.branch_controller": ("classes", ["Branch_Controller"]),
    "deft.tree_decoding.generation.tree_generate": ("functions", ["tree_generate"]),
    "deft.data_loader": ("mixed", ["ExecuteTreeNode", "ExecuteTree", "build_tree", "build_trees", "load_dataset", "load_trees",
                                   "load_prompts", "generate_accepted_len_list", "build_tree_SD"]),
}


def params_from_ast(fn: ast.FunctionDef) -> list:
    a = fn.args
    out = []
    pos = list(a.posonlyargs) + list(a.args)
    defaults = [None] * (len(pos) - len(a.defaults)) + list(a.defaults)
    for i, (arg, d) in enumerate(zip(pos, defaults)):
        kind = "posonly" if i < len(a.posonlyargs) else "pos"
        out.append({"name": arg.arg, "kind": kind, "default": None if d is None else ast.unparse(d)})
    if a.vararg:
        out.append({"name": a.vararg.arg, "kind": "vararg", "default": None})
    for arg, d in zip(a.kwonlyargs, a.kw_defaults):
        out.append({"name": arg.arg, "kind": "kwonly", "default": None if d is None else ast.unparse(d)})
    if a.kwarg:
        out.append({"name": a.kwarg.arg, "kind": "varkw", "default": None})
    return out


def params_from_inspect(obj) -> list:
    kinds = {inspect.Parameter.POSITIONAL_ONLY: "posonly", inspect.Parameter.POSITIONAL_OR_KEYWORD: "pos",
             inspect.Parameter.VAR_POSITIONAL: "vararg", inspect.Parameter.KEYWORD_ONLY: "kwonly",
             inspect.Parameter.VAR_KEYWORD: "varkw"}
    return [{"name": p.name, "kind": kinds[p.kind], "has_default": p.default is not inspect.Parameter.empty}
            for p in inspect.signature(obj).parameters.values()]


def decorators(node) -> list:
    return [ast.unparse(d) for d in node.decorator_list]


def class_surface(cls: ast.ClassDef) -> dict:
    out = {"bases": [ast.unparse(b) for b in cls.bases], "dataclass": any("dataclass" in d for d in decorators(cls)),
           "fields": [], "members": [], "methods": {}}
    for st in cls.body:
        if isinstance(st, ast.AnnAssign) and isinstance(st.target, ast.Name):  # dataclass fields, in order
            out["fields"].append({"name": st.target.id, "default": None if st.value is None else ast.unparse(st.value)})
        elif isinstance(st, ast.Assign) and all(isinstance(t, ast.Name) for t in st.targets):  # enum members
            out["members"] += [t.id for t in st.targets]
        elif isinstance(st, ast.FunctionDef) and (not st.name.startswith("_") or st.name in ("__init__", "__call__")):
            kind = "classmethod" if "classmethod" in decorators(st) else "staticmethod" if "staticmethod" in decorators(st) else "method"
            out["methods"][st.name] = {"kind": kind, "params": params_from_ast(st)}
    return out


def main() -> None:
    surface = {}
    for mod, (_, names) in MODULES.items():
        path = os.path.join(REF, *mod.split(".")) + ".py"
        tree = ast.parse(open(path).read())
        top = {}
        for st in tree.body:
            if isinstance(st, (ast.FunctionDef, ast.ClassDef)):
                top[st.name] = st
            elif isinstance(st, ast.Assign) and isinstance(st.targets[0], ast.Name):
                top[st.targets[0].id] = st
        try:
            live = importlib.import_module(mod)
        except Exception as e:  # no GPU / flashinfer / vllm here
            live = None
            print(f"  ({mod}: ast only -- {type(e).__name__}: {str(e)[:60]})")
        entry = {"importable_here": live is not None, "functions": {}, "classes": {}, "constants": {}}
        for name in names:
            node = top[name]
            if isinstance(node, ast.FunctionDef):
                entry["functions"][name] = params_from_ast(node)
                if live is not None:
                    fn = getattr(live, name)
                    fn = getattr(fn, "__wrapped__", fn)  # (torch.no_grad / inference_mode wrappers keep the signature anyway)
                    got = params_from_inspect(fn)
                    want = [{"name": p["name"], "kind": p["kind"], "has_default": p["default"] is not None} for p in entry["functions"][name]]
                    assert got == want, (mod, name, got, want)
            elif isinstance(node, ast.ClassDef):
                entry["classes"][name] = class_surface(node)
                if live is not None:
                    cls = getattr(live, name)
                    for mname, m in entry["classes"][name]["methods"].items():
                        got = params_from_inspect(inspect.getattr_static(cls, mname).__func__ if m["kind"] != "method"
                                                  else getattr(cls, mname))
                        want = [{"name": p["name"], "kind": p["kind"], "has_default": p["default"] is not None} for p in m["params"]]
                        assert got == want, (mod, name, mname, got, want)
            else:  # a module-level configuration dictionary
                entry["constants"][name] = ast.unparse(node.value)
                if live is not None:
                    assert getattr(live, name) == ast.literal_eval(node.value)
        surface[mod] = entry
    path = os.path.join(ROOT, "tests", "golden", "surface.json")
    json.dump(surface, open(path, "w"), indent=1, sort_keys=True)
    n_fn = sum(len(e["functions"]) + sum(len(c["methods"]) for c in e["classes"].values()) for e in surface.values())
    print(f"{path}: {len(surface)} modules, {n_fn} callables, {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
