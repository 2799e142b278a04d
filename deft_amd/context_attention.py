"""Causal prefill attention over the prompt — `context_attention_fwd`.

Same name, positional signature and in-place output as
DeFT/deft/layers/attention/context_flashattention_nopad.py:130-195, the operator behind
`DeFTAttention.prefill_forward_triton` (deft_attention.py:50-70): sequences packed without padding, token i of a
sequence attends to tokens 0..i of it.  Backed by libdeft_amd.so (deft_prefill_f16; head_dim 128 and 64 on the MFMA
kernel, 32 and 16 on a plain one); no PyTorch or CPU fallback.
"""
from __future__ import annotations

import torch

from ._lib import DeftLibraryError, check, lib

__all__ = ["context_attention_fwd"]


@torch.inference_mode()
def context_attention_fwd(q, k, v, o, b_start_loc, b_seq_len, max_input_len) -> None:
    for name, t in (("q", q), ("k", k), ("v", v), ("o", o)):
        if not t.is_cuda:
            raise DeftLibraryError(f"{name} must be a CUDA (ROCm) tensor; deft_amd has no CPU path")
        if t.dtype != torch.float16:
            raise TypeError(f"{name} must be float16, got {t.dtype}")
        if t.dim() != 3 or t.stride(2) != 1:
            raise ValueError(f"{name} must be [tokens, heads, head_dim] with a contiguous head_dim")
    Lq, Lk, Lv = q.shape[-1], k.shape[-1], v.shape[-1]
    assert Lq == Lk and Lk == Lv  # context_flashattention_nopad.py:136-138
    assert Lk in {16, 32, 64, 128}
    start = b_start_loc if b_start_loc.dtype == torch.int32 else b_start_loc.to(torch.int32)
    lens = b_seq_len if b_seq_len.dtype == torch.int32 else b_seq_len.to(torch.int32)  # seq_lens is int64 upstream
    start, lens = start.contiguous(), lens.contiguous()
    rc = lib.deft_prefill_f16(q.data_ptr(), q.stride(0), q.stride(1), k.data_ptr(), k.stride(0), k.stride(1),
                              v.data_ptr(), v.stride(0), v.stride(1), o.data_ptr(), o.stride(0), o.stride(1),
                              start.data_ptr(), lens.data_ptr(), lens.shape[0], int(max_input_len), q.shape[1], k.shape[1],
                              Lk, 1.0 / (Lq ** 0.5), torch.cuda.current_stream(q.device).cuda_stream)
    check(rc, "deft_prefill_f16")
