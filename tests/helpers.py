"""Shared test helpers: replay a scenario on the oracle tree, build seeded inputs."""
from __future__ import annotations

import numpy as np

from oracle.metadata import build_metadata
from oracle.tree_model import OracleReqTable, OracleTokenPool, OracleTree
from scenarios import SCENARIOS, input_seeds

from deft_amd.utils.synthetic import dyadic_normal


def oracle_tree(name: str) -> OracleTree:
    sc = SCENARIOS[name]
    tree = OracleTree(OracleTokenPool(sc.pool_size), OracleReqTable(128, sc.pool_size + 8))
    sc.script(tree, lambda n: np.arange(1, n + 1, dtype=np.int32))
    return tree


def oracle_metadata(name: str, tree: OracleTree = None):
    sc = SCENARIOS[name]
    tree = tree if tree is not None else oracle_tree(name)
    return build_metadata(tree, sc.max_q_len, sc.block_len, sc.max_block_len)


def seeded_inputs(name: str, geom, nq: int):
    """(q [nq,Hq,D], kv_data [pool,2,Hkv,D]) exactly as tools/gen_golden.py made them."""
    Hq, Hkv, D = geom
    seeds = input_seeds(name, geom)
    q = dyadic_normal((nq, Hq, D), seeds["q"])
    kv = dyadic_normal((SCENARIOS[name].pool_size, 2, Hkv, D), seeds["kv"])
    return q, kv


def leaf_paths(tree: OracleTree):
    return [tree.path_slots(leaf) for leaf in tree.leaf_order()]


def max_abs(a, b) -> float:
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))
