"""Tree TEMPLATES: the shape a decoding tree takes over time, as flat arrays and per-iteration event lists.

The reference describes a workload's tree ahead of time and lets a branch controller consult it at every decode step
(DeFT/deft/data_loader.py:9-132 for the reasoning files, :181-235 for the speculative-decoding records;
`dataset/generation/**`).  A template here is a table -- one row per node, numpy columns -- and two event maps derived
from it in ONE iterative pass:

    value[i]            tokens generated inside node i
    start[i], end[i]    iterations at which node i appears / branches or finishes
    kids(i)             children of node i (CSR: child_off / child_ids), in file order
    level[i], rank[i]   depth of node i, and its index among the nodes of its depth in DFS order
    branch_at[it]       [(node, [children])]  nodes that branch at iteration `it`
    prune_at[it]        [node]                nodes released at iteration `it`: a childless node at its own end, an inner
                                              node when the last node of its subtree has ended

`branch_record` / `prune_record` give the same events in the reference's dictionary form; tests/golden/templates.json pins
them (and the level statistics) on what the reference's own loader derives from its shipped files.  The files themselves
belong to the reference repository and are read where the user has them (`read_reasoning_file`, `read_speculative_file`);
`synthetic_*` build templates of the same form.
"""
from __future__ import annotations

import json
import random
from typing import Any, Dict, Iterable, List, Mapping, Optional, Sequence, Tuple

import numpy as np

__all__ = ["TreeTemplate", "read_reasoning_file", "read_speculative_file", "fit_accept_lengths", "synthetic_reasoning_template",
           "synthetic_speculative_template", "synthetic_few_shot_template", "synthetic_beam_template", "default_prompt_len"]

OPEN_ENDED = 1 << 30  # `value` / `end` of a node that generates until the replay's own limit (few-shot leaves)


class TemplateNode:
    """One node of a template as the reference's loader presents it (data_loader.py:9-27)."""
    __slots__ = ("id", "value", "start_offset", "end_offset", "depth", "width", "children")

    def __init__(self, node_id: int, value: int, start_offset: int, end_offset: int, depth: int = 0, width: int = 0) -> None:
        self.id, self.value, self.start_offset, self.end_offset = node_id, value, start_offset, end_offset
        self.depth, self.width = depth, width
        self.children: List["TemplateNode"] = []

    def __repr__(self) -> str:
        return (f"TreeNode(id={self.id}, value={self.value}, start={self.start_offset}, end={self.end_offset}, "
                f"depth={self.depth}, width={self.width})")


class TreeTemplate:
    def __init__(self, value: Sequence[int], start: Sequence[int], end: Sequence[int], children: Sequence[Sequence[int]],
                 prompt: Optional[str] = None, accept_lengths: Optional[List[int]] = None) -> None:
        n = len(value)
        assert n > 0 and len(start) == n and len(end) == n and len(children) == n
        self.value = np.asarray(value, dtype=np.int64)
        self.start = np.asarray(start, dtype=np.int64)
        self.end = np.asarray(end, dtype=np.int64)
        self.child_off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum([len(c) for c in children], out=self.child_off[1:])
        self.child_ids = np.fromiter((int(c) for cs in children for c in cs), dtype=np.int64, count=int(self.child_off[-1]))
        self.prompt = prompt
        self.accept_lengths = accept_lengths  # speculative decoding only: accepted tokens per verification step
        self.level = np.zeros(n, dtype=np.int64)
        self.rank = np.zeros(n, dtype=np.int64)
        self.release = np.zeros(n, dtype=np.int64)  # iteration at which the node is pruned
        self.branch_at: Dict[int, List[Tuple[int, List[int]]]] = {}
        self.prune_at: Dict[int, List[int]] = {}
        self._nodes: Optional[List["TemplateNode"]] = None
        self._derive_events()

    # ---- shape ---------------------------------------------------------------------------------------------
    @property
    def node_num(self) -> int:
        return int(self.value.shape[0])

    def kids(self, i: int) -> List[int]:
        return self.child_ids[self.child_off[i] : self.child_off[i + 1]].tolist()

    @property
    def root_width(self) -> int:
        return int(self.child_off[1] - self.child_off[0])

    @property
    def width_per_depth(self) -> Dict[int, int]:
        lv, cnt = np.unique(self.level[self._reached], return_counts=True)
        return {int(a): int(b) for a, b in zip(lv, cnt)}

    @property
    def max_depth(self) -> int:
        return int(self.level[self._reached].max())

    @property
    def max_width(self) -> int:
        return max(self.width_per_depth.values())

    # ---- events --------------------------------------------------------------------------------------------
    def _derive_events(self) -> None:
        """One DFS from node 0 with an explicit stack.  Events are appended in the order the walk meets them: a childless
        node's release and an inner node's branch on the way DOWN, an inner node's release on the way back UP -- at the latest
        `end` of its subtree (the reference's recursion yields the same order, data_loader.py:51-77)."""
        n = self.node_num
        seen_at_level: Dict[int, int] = {}
        reached = np.zeros(n, dtype=bool)
        latest = self.end.copy()  # latest end inside the subtree, folded upwards on the way back
        stack: List[Tuple[int, int, int]] = [(0, 0, -1)]  # (node, level, parent); node < 0 marks the way back up of ~node
        while stack:
            i, lvl, par = stack.pop()
            if i < 0:
                i = ~i
                self.release[i] = latest[i]
                self.prune_at.setdefault(int(latest[i]), []).append(i)
                if par >= 0:
                    latest[par] = max(latest[par], latest[i])
                continue
            reached[i] = True
            self.level[i] = lvl
            self.rank[i] = seen_at_level.get(lvl, 0)
            seen_at_level[lvl] = int(self.rank[i]) + 1
            ks = self.kids(i)
            if not ks:
                self.release[i] = self.end[i]
                self.prune_at.setdefault(int(self.end[i]), []).append(i)
                if par >= 0:
                    latest[par] = max(latest[par], self.end[i])
                continue
            self.branch_at.setdefault(int(self.end[i]), []).append((i, ks))
            stack.append((~i, lvl, par))
            for c in reversed(ks):
                stack.append((c, lvl + 1, i))
        self._reached = reached

    @property
    def branch_record(self) -> Dict[int, Dict[int, List[int]]]:
        return {it: {node: list(ks) for node, ks in ev} for it, ev in self.branch_at.items()}

    @property
    def prune_record(self) -> Dict[int, List[int]]:
        return {it: list(v) for it, v in self.prune_at.items()}

    # ---- the reference's attribute names (data_loader.py:9-49) -------------------------------------------------
    @property
    def accepted_len_list(self) -> Optional[List[int]]:
        return self.accept_lengths

    @accepted_len_list.setter
    def accepted_len_list(self, v: Optional[List[int]]) -> None:
        self.accept_lengths = v

    @property
    def nodes(self) -> List["TemplateNode"]:
        """Node records in the reference's form (`id, value, start_offset, end_offset, children, depth, width`), built on first
        use from the arrays; `root` is `nodes[0]`."""
        if self._nodes is None:
            ns = [TemplateNode(i, int(self.value[i]), int(self.start[i]), int(self.end[i]), int(self.level[i]), int(self.rank[i]))
                  for i in range(self.node_num)]
            for i, nd in enumerate(ns):
                nd.children = [ns[c] for c in self.kids(i)]
            self._nodes = ns
        return self._nodes

    @property
    def root(self) -> "TemplateNode":
        return self.nodes[0]

    def token_budget(self) -> int:
        """Tokens all nodes but open-ended ones generate (pool sizing)."""
        return int(self.value[self.value < (OPEN_ENDED >> 1)].clip(min=0).sum())

    # ---- constructors --------------------------------------------------------------------------------------
    @classmethod
    def from_node_table(cls, table: Mapping[Any, Mapping[str, Any]], prompt: Optional[str] = None) -> "TreeTemplate":
        """`table`: the node map of a reasoning file -- {key: {"id", "value", "start", "end", "children"}} (keys are ignored,
        rows are placed by their "id")."""
        n = len(table)
        value, start, end = [0] * n, [0] * n, [0] * n
        children: List[List[int]] = [[] for _ in range(n)]
        for row in table.values():
            i = int(row["id"])
            value[i], start[i], end[i] = int(row["value"]), int(row["start"]), int(row["end"])
            children[i] = [int(c) for c in row["children"]]
        return cls(value, start, end, children, prompt)

    @classmethod
    def flat(cls, size: int, accept_lengths: Iterable[int], prompt: Optional[str] = None) -> "TreeTemplate":
        """The speculative-decoding form (data_loader.py:181-197): `size` unconnected nodes -- the branch function keeps
        `size` one-token leaves below the root -- and the accepted lengths of one record."""
        z = [0] * size
        return cls(z, z, z, [[] for _ in range(size)], prompt, [int(a) for a in accept_lengths])


def _read_json_or_pickle(path: str) -> Any:
    if path.endswith(".json"):
        with open(path, "r") as f:
            return json.load(f)
    if path.endswith(".pkl"):
        import pickle

        with open(path, "rb") as f:
            return pickle.load(f)
    raise NotImplementedError(f"Unsupported file format: {path}")


def read_reasoning_file(path: str) -> List[TreeTemplate]:
    """dataset/generation/Reasoning/*.json: a list of records, either a bare node table or {"data": table, "prompt",
    "incompleted"}; records flagged incompleted are skipped (data_loader.py:99-104)."""
    out = []
    for rec in _read_json_or_pickle(path):
        if "data" in rec:
            if rec.get("incompleted"):
                continue
            out.append(TreeTemplate.from_node_table(rec["data"], rec.get("prompt")))
        else:
            out.append(TreeTemplate.from_node_table(rec))
    return out


def read_speculative_file(path: str) -> List[TreeTemplate]:
    """dataset/generation/Speculative_Decoding/*.json: {"Token_Tree_size", "Records": [{"prompt", "Accept_length"}]}."""
    doc = _read_json_or_pickle(path)
    size = int(doc["Token_Tree_size"])
    return [TreeTemplate.flat(size, rec["Accept_length"], rec.get("prompt")) for rec in doc["Records"]]


def fit_accept_lengths(tpl: TreeTemplate, max_gen_len: int, rng: Optional[random.Random] = None) -> None:
    """Make the record's accepted lengths sum to exactly `max_gen_len`: keep the longest prefix that fits, then draw further
    lengths between the record's smallest and largest (clipped at the end) -- data_loader.py:200-235."""
    rng = rng or random
    acc = tpl.accept_lengths
    assert acc
    lo, hi = min(acc), max(acc)
    csum = np.cumsum(acc)
    keep = int(np.searchsorted(csum, max_gen_len, side="right"))
    out = [int(a) for a in acc[:keep]]
    total = int(csum[keep - 1]) if keep else 0
    while total < max_gen_len:
        a = min(rng.randint(lo, hi), max_gen_len - total)
        out.append(a)
        total += a
    tpl.accept_lengths = out


# ---- synthetic templates of the same form -------------------------------------------------------------------
def synthetic_reasoning_template(widths=(7, 6), lens=(128, 64)) -> TreeTemplate:
    """A tree-of-thoughts template: level d has widths[d] children per node, each generating lens[d] tokens
    (7 x 128 then 42 x 64 = 50 live nodes).  Node ids in creation (BFS) order, like the reference's files."""
    value, start, end, children = [0], [0], [0], [[]]
    frontier = [0]
    for width, n_tok in zip(widths, lens):
        nxt = []
        for parent in frontier:
            for _ in range(width):
                nid = len(value)
                value.append(n_tok)
                start.append(end[parent] + 1)
                end.append(end[parent] + n_tok)
                children.append([])
                children[parent].append(nid)
                nxt.append(nid)
        frontier = nxt
    return TreeTemplate(value, start, end, children)


def synthetic_beam_template(width: int = 10, depth: int = 10, seg_len: int = 8) -> TreeTemplate:
    """The shape of the reference's shipped Reasoning templates (dataset/generation/Reasoning/*: 31 / 61 / 91 / 101 lifetime nodes,
    width 10 per level): at every level the kept node branches into `width` candidates, each generates `seg_len` tokens, all but
    the first are pruned and the first branches again -- 1 + width x depth nodes, a branch (and width - 1 prunes) every `seg_len`
    decode steps: the workload whose tree changes shape every few steps."""
    value, start, end, children = [0], [0], [0], [[]]
    parent = 0
    for _ in range(depth):
        kids = []
        for _ in range(width):
            nid = len(value)
            value.append(seg_len)
            start.append(end[parent] + 1)
            end.append(end[parent] + seg_len)
            children.append([])
            children[parent].append(nid)
            kids.append(nid)
        parent = kids[0]
    return TreeTemplate(value, start, end, children)


def synthetic_speculative_template(tree_size: int = 64, steps: int = 100, accept=(1, 4), seed: int = 0) -> TreeTemplate:
    rng = random.Random(seed)
    return TreeTemplate.flat(tree_size, [rng.randint(accept[0], accept[1]) for _ in range(steps)])


def synthetic_few_shot_template(width: int = 32) -> TreeTemplate:
    """SimpleTree (branch_func_example.py:12-62): the root branches into `width` leaves after the prefill."""
    n = width + 1
    return TreeTemplate([0] + [OPEN_ENDED] * width, [0] + [1] * width, [0] + [OPEN_ENDED] * width,
                        [list(range(1, n))] + [[] for _ in range(width)])


def default_prompt_len(tpl: TreeTemplate, task: str, from_file: bool = False) -> int:
    """Prompt length of a replay that was given none: a reasoning template READ FROM A FILE carries it as the root's token count
    (data_loader.py:31-49: the root node of a Reasoning tree is the prompt); otherwise BASELINE's shapes -- a ~1k-token root under a
    speculative-decoding tree, a 4k prompt under everything else."""
    root_tokens = int(tpl.value[0])
    if from_file and task == "reasoning" and 0 < root_tokens < OPEN_ENDED:
        return root_tokens
    return 1016 if task == "speculative_decoding" else 4096
