"""RotaryEmbedding: the op in front of the attention path (LlamaAttention.forward, llama2.py:108-110).

Mirrors DeFT/deft/layers/rotary_embedding.py:75-180 for the configuration the reference's Llama uses
(`get_rope(head_dim, rotary_dim=head_dim, max_position, base, rope_scaling=None, dtype=float32)`, llama2.py:86-93):
same constructor arguments, same fp32 `cos_sin_cache` buffer, `forward(positions, query, key)` rotates query and key
IN PLACE and returns them, like `forward_cuda` (:157-177).  The rotation runs in libdeft_amd.so (deft_rope_qk_f16);
there is no PyTorch fallback.  Scaled variants (linear / dynamic NTK / YaRN, :183-650) are out of scope.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import nn

from ._lib import DeftLibraryError, check, lib

__all__ = ["RotaryEmbedding", "get_rope"]


class RotaryEmbedding(nn.Module):
    def __init__(self, head_size: int, rotary_dim: int, max_position_embeddings: int, base: float, is_neox_style: bool,
                 dtype: torch.dtype) -> None:
        super().__init__()
        if dtype != torch.float32:
            raise NotImplementedError("the cos/sin cache is fp32, as the reference's Llama builds it (llama2.py:86-93)")
        self.head_size, self.rotary_dim = head_size, rotary_dim
        self.max_position_embeddings, self.base, self.is_neox_style, self.dtype = max_position_embeddings, base, is_neox_style, dtype
        inv_freq = 1.0 / (base ** (torch.arange(0, rotary_dim, 2, dtype=torch.float) / rotary_dim))  # :103-117
        freqs = torch.einsum("i,j -> ij", torch.arange(max_position_embeddings, dtype=torch.float), inv_freq)
        self.register_buffer("cos_sin_cache", torch.cat((freqs.cos(), freqs.sin()), dim=-1), persistent=False)

    @torch.inference_mode()
    def forward(self, positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor,
                offsets: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        if not query.is_cuda:
            raise DeftLibraryError("RotaryEmbedding needs CUDA (ROCm) tensors; deft_amd has no CPU path")
        if query.dtype != torch.float16 or key.dtype != torch.float16:
            raise TypeError("query / key must be float16")
        if offsets is not None:
            positions = positions + offsets
        positions = positions.flatten().to(torch.int64)
        n = positions.shape[0]
        if self.cos_sin_cache.device != query.device:
            self.cos_sin_cache = self.cos_sin_cache.to(query.device)  # :165
        q = query.view(n, -1, self.head_size)
        k = key.view(n, -1, self.head_size)
        if q.stride(2) != 1 or k.stride(2) != 1:
            raise ValueError("query / key must have a contiguous head dimension")
        check(lib.deft_rope_qk_f16(q.data_ptr(), q.stride(0), q.stride(1), q.shape[1], k.data_ptr(), k.stride(0), k.stride(1),
                                   k.shape[1], positions.data_ptr(), self.cos_sin_cache.data_ptr(),
                                   self.cos_sin_cache.stride(0), n, self.head_size, self.rotary_dim,
                                   1 if self.is_neox_style else 0, torch.cuda.current_stream(query.device).cuda_stream),
              "deft_rope_qk_f16")
        return query, key

    def forward_native(self, positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor,
                       offsets: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """The reference's PyTorch spelling (rotary_embedding.py:119-155) returns NEW tensors and leaves its inputs alone; same
        here -- the same kernel on copies (its results are bit-equal to the reference's forward_native: tests/golden/rope.npz)."""
        return self.forward(positions, query.clone(), key.clone(), offsets)


def get_rope(head_size: int, rotary_dim: int, max_position: int, base: float, is_neox_style: bool = True,
             rope_scaling=None, dtype: Optional[torch.dtype] = None, partial_rotary_factor: float = 1.0) -> RotaryEmbedding:
    """rotary_embedding.py:647-690 for rope_scaling=None (the reference's Llama passes dtype=float32, llama2.py:86-93; with no
    dtype the reference takes torch's default dtype, which this class accepts only when it is float32)."""
    if rope_scaling is not None:
        raise NotImplementedError("scaled rotary embeddings are out of scope (rotary_embedding.py:183-650)")
    if dtype is None:
        dtype = torch.get_default_dtype()
    if partial_rotary_factor < 1.0:
        rotary_dim = int(rotary_dim * partial_rotary_factor)
    return RotaryEmbedding(head_size, rotary_dim, max_position, base, is_neox_style, dtype)
