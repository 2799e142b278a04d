"""Token-granular paged KV pools with the reference's layout and interface.

Mirrors DeFT/deft/memory_pool.py:
  ReqToTokenPool (:11-45)   leaf -> slot page table [requests, ctx] int32
  TokenToKVPool  (:48-108)  per layer kv_data[layer] = [size, 2, Hkv, D] fp16;
                            K view = [:, 0], V view = [:, 1]; int16 refcount per
                            slot; alloc = the lowest free slots (:74-80)

MI355X-side differences (interface unchanged):
  * slot bookkeeping lives in host memory (numpy), so alloc/free never launch a
    kernel or synchronise; the reference runs `torch.nonzero` on the GPU and a
    `.item()` per leaf every step (tree_cache.py:265-270).
  * `device` is a constructor argument (the reference hard-codes "cuda"), so the
    host logic is testable without a GPU.
  * all layers live in ONE allocation [layers, size, 2, Hkv, D]; 288 GB of HBM3E
    per GPU holds e.g. 32 layers x 500k slots of Llama-2-7B KV (262 GB).
    `kv_data[layer]` still returns the reference-shaped per-layer tensor.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch


class ReqToTokenPool:
    def __init__(self, size: int, max_context_len: int, device: str = "cuda") -> None:
        self._free = np.ones(size, dtype=bool)
        self.can_use_mem_size = size
        self.req_to_token = torch.zeros((size, max_context_len), dtype=torch.int32, device=device)

    def alloc(self, need_size: int) -> Optional[torch.Tensor]:
        if need_size > self.can_use_mem_size:
            return None
        idx = np.flatnonzero(self._free)[:need_size]
        self._free[idx] = False
        self.can_use_mem_size -= need_size
        return torch.from_numpy(idx.astype(np.int32))

    def free(self, free_index) -> None:
        if isinstance(free_index, int):
            self.can_use_mem_size += 1
            self._free[free_index] = True
        else:
            idx = np.asarray(torch.as_tensor(free_index).cpu(), dtype=np.int64).reshape(-1)
            self.can_use_mem_size += idx.shape[0]
            self._free[idx] = True

    def copy(self, from_req: int, to_req: int, copy_len: int) -> None:
        self.req_to_token[to_req, :copy_len] = self.req_to_token[from_req, :copy_len]

    def clear(self) -> None:
        self._free[:] = True
        self.can_use_mem_size = len(self._free)


class TokenToKVPool:
    def __init__(
        self,
        size: int,
        dtype: torch.dtype,
        head_num: int,
        head_dim: int,
        layer_num: int,
        device: str = "cuda",
    ) -> None:
        self.size = size
        self.head_num = head_num
        self.head_dim = head_dim
        self.device = torch.device(device)
        self.mem_state = np.zeros(size, dtype=np.int16)  # refcounts, host side
        self.alloc_ct = 0
        self._lo = 0  # every slot below this index is in use: where the search for free slots starts
        # [layer][size, key/value, head_num, head_dim] — memory_pool.py:61-66
        self._storage = torch.empty((layer_num, size, 2, head_num, head_dim), dtype=dtype, device=device)
        self.device = self._storage.device  # (WITH its index: torch.device("cuda") != torch.device("cuda:0"))
        self.kv_data = [self._storage[i] for i in range(layer_num)]

    def get_key_buffer(self, layer_id: int) -> torch.Tensor:
        return self.kv_data[layer_id][:, 0]

    def get_value_buffer(self, layer_id: int) -> torch.Tensor:
        return self.kv_data[layer_id][:, 1]

    # -- host-side slot allocator -------------------------------------------------
    def alloc_host(self, need_size: int) -> Optional[np.ndarray]:
        """The `need_size` LOWEST free slots (what `torch.nonzero(mem_state == 0)[:need_size]` picks, memory_pool.py:75-84).
        The scan starts at the lowest slot that can be free and stops when it has enough: a decode step's handful of slots
        costs microseconds whatever the pool's size (a full scan of a 500k-slot pool is ~0.4 ms, per step)."""
        if need_size <= 0:
            return np.zeros(0, dtype=np.int32)
        lo, ms = self._lo, self.mem_state
        if lo + need_size <= self.size and not ms[lo : lo + need_size].any():
            # (the usual decode step: the lowest free slots are the next ones in a row)
            ms[lo : lo + need_size] = 1
            self._lo = lo + need_size
            self.alloc_ct += need_size
            return np.arange(lo, lo + need_size, dtype=np.int32)
        idx = self._lowest_free(self._lo, need_size)
        if idx is None and self._lo > 0:
            idx = self._lowest_free(0, need_size)  # (entries of mem_state cleared directly, below the hint)
        if idx is None:
            return None
        self._lo = int(idx[-1]) + 1
        self.mem_state[idx] = 1  # (distinct slots that were free: no ufunc.at -- 10 us per call for a step's 64 slots)
        self.alloc_ct += len(idx)
        return idx.astype(np.int32)

    def _lowest_free(self, start: int, need: int) -> Optional[np.ndarray]:
        ms, size = self.mem_state, self.size
        chunk = max(1024, 4 * need)
        found, cnt = [], 0
        while start < size and cnt < need:
            seg = np.flatnonzero(ms[start : start + chunk] == 0)
            if seg.size:
                seg = seg[: need - cnt] + start
                found.append(seg)
                cnt += seg.size
            start += chunk
        if cnt < need:
            return None
        return found[0] if len(found) == 1 else np.concatenate(found)

    def alloc(self, need_size: int) -> Optional[torch.Tensor]:
        idx = self.alloc_host(need_size)
        if idx is None:
            return None
        return torch.from_numpy(idx).to(self.device, non_blocking=True)

    @staticmethod
    def _as_index(token_index) -> np.ndarray:
        if isinstance(token_index, torch.Tensor):
            token_index = token_index.detach().cpu().numpy()
        return np.asarray(token_index, dtype=np.int64).reshape(-1)

    def _add(self, idx: np.ndarray) -> None:
        self.alloc_ct += len(idx)
        if len(idx) <= 8:  # (ufunc.at costs microseconds per call whatever the length: a leaf's one or two slots go by hand)
            ms = self.mem_state
            for i in idx.tolist():
                ms[i] += 1
            return
        np.add.at(self.mem_state, idx, 1)

    def add_refs(self, token_index) -> None:
        self._add(self._as_index(token_index))

    def decrease_refs(self, token_index) -> int:
        idx = self._as_index(token_index)
        self.alloc_ct -= len(idx)
        if len(idx) <= 8:
            ms, lst = self.mem_state, idx.tolist()
            for i in lst:
                ms[i] -= 1
                if ms[i] == 0 and i < self._lo:
                    self._lo = i
            return sum(int(ms[i] == 0) for i in lst)  # (per occurrence, like the vector form below)
        np.subtract.at(self.mem_state, idx, 1)
        freed = self.mem_state[idx] == 0
        n_free = int(np.sum(freed))
        if n_free:
            self._lo = min(self._lo, int(idx[freed].min()))
        return n_free

    def free(self, free_index) -> int:
        return self.decrease_refs(free_index)

    def used_size(self) -> int:
        return int(np.count_nonzero(self.mem_state))

    def available_size(self) -> int:
        return int(np.sum(self.mem_state == 0))

    def clear(self) -> None:
        self.mem_state[:] = 0
        self.alloc_ct = 0
        self._lo = 0
