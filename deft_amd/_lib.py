"""ctypes binding of libdeft_amd.so (the C ABI declared in include/deft_amd.h).

The library is built in-tree by `deft_amd/csrc/Makefile` (see `__graft_entry__.build`).
There is no fallback: if the shared object is missing, importing this module
raises, and every operator in the package fails with it.
"""
from __future__ import annotations

import ctypes as C
import os

# PyTorch-ROCm ships its own libamdhip64; load it FIRST so that libdeft_amd.so's DT_NEEDED entry resolves to the
# runtime torch has initialised.  Loaded the other way round the process holds two HIP runtimes and the one this
# library is bound to reports "no ROCm-capable device is detected" at the first launch.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DEFT_AMD_LIB") or os.path.join(_HERE, "lib", "libdeft_amd.so")  # override: A/B of two builds


class DeftLibraryError(RuntimeError):
    pass


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise DeftLibraryError(
            f"{LIB_PATH} not found: build it with `make -C deft_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "deft_amd has no CPU or PyTorch fallback for its HIP kernels."
        )
    return C.CDLL(LIB_PATH)


lib = _load()

_vp, _i64, _i32, _f32, _sz = C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_size_t

lib.deft_abi_version.restype = C.c_int
lib.deft_plan_variant.restype = C.c_int
if hasattr(lib, "deft_debug_plan_form"):  # experiments build only (DEFT_AMD_LIB=.../libdeft_amd_exp.so): a test hook
    lib.deft_debug_plan_form.argtypes = [C.c_int, C.c_int]
    lib.deft_debug_plan_form.restype = None
lib.deft_last_error.restype = C.c_char_p
lib.deft_supported.argtypes = [_i32, _i32, _i32]
lib.deft_supported.restype = C.c_int

lib.deft_flatten_workspace_bytes.argtypes = [_i32] * 6
lib.deft_flatten_workspace_bytes.restype = _sz
lib.deft_node_workspace_bytes.argtypes = [_i32, _i32, _i64, _i32, _i32, _i32, _i32]
lib.deft_node_workspace_bytes.restype = _sz

_QKV = [_vp, _i64, _i64, _vp, _vp, _i64, _i64]  # q, q strides, k, v, kv strides
_OUT = [_vp, _i64, _i64]
_MD6 = [_vp] * 6

lib.deft_flatten_decode_f16.argtypes = _QKV + _OUT + _MD6 + [_i32] * 6 + [_f32, _vp, _vp, _sz, _vp]
lib.deft_flatten_decode_f16.restype = C.c_int
lib.deft_flatten_decode_append_f16.argtypes = (_QKV + _OUT + _MD6 + [_i32] * 6 + [_f32, _vp, _vp, _vp, _i64, _i32] +
                                               [_vp, _vp, _sz, _vp])
lib.deft_flatten_decode_append_f16.restype = C.c_int
lib.deft_flatten_stage1_f16.argtypes = _QKV + _MD6 + [_i32] * 6 + [_f32, _vp, _vp, _sz, _vp]
lib.deft_flatten_stage1_f16.restype = C.c_int
lib.deft_probe_stream_read.argtypes = [_vp, _sz, _i32, _vp]
lib.deft_probe_stream_read.restype = C.c_int
lib.deft_flatten_plan_bytes.argtypes = [_i32, _i32, _i32, _i32]
lib.deft_flatten_plan_bytes.restype = _sz
lib.deft_flatten_build_plan.argtypes = _MD6 + [_i32, _i32, _i32, _i32, _i64, _i64, _i64, _vp, _i32, _i64, _vp, _sz, _vp]
lib.deft_flatten_build_plan.restype = C.c_int
lib.deft_flatten_build_plan_dims.argtypes = _MD6 + [_i32, _i32, _vp, _i32, _i32, _i64, _i64, _i64, _vp, _i32, _i64, _vp, _sz, _vp]
lib.deft_flatten_build_plan_dims.restype = C.c_int
lib.deft_node_decode_f16.argtypes = _QKV + _OUT + _MD6 + [_i32, _i32, _i64, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _sz, _vp]
lib.deft_node_plan_bytes.argtypes = [_i32, _i32, _i64, _i32, _i32]
lib.deft_node_plan_bytes.restype = _sz
lib.deft_node_build_plan.argtypes = _MD6 + [_i32, _i32, _i64, _i32, _i32, _i64, _i64, _i64, _vp, _i32, _i64, _vp, _sz, _vp]
lib.deft_node_build_plan.restype = C.c_int
lib.deft_node_build_plan_dims.argtypes = _MD6 + [_i32, _i32, _i64, _vp, _i32, _i32, _i64, _i64, _i64, _vp, _i32, _i64, _vp, _sz, _vp]
lib.deft_node_build_plan_dims.restype = C.c_int
lib.deft_node_decode_f16.restype = C.c_int
lib.deft_node_decode_append_f16.argtypes = (_QKV + _OUT + _MD6 + [_i32, _i32, _i64, _i32, _i32, _i32, _i32, _f32]
                                            + [_vp, _vp, _vp, _i64, _i32] + [_vp, _vp, _sz, _vp])
lib.deft_node_decode_append_f16.restype = C.c_int
_ROPE = [_vp, _i32, _i32]  # cos_sin_rows, rotary_dim, is_neox_style
lib.deft_flatten_decode_rope_append_f16.argtypes = (_QKV + _OUT + _MD6 + [_i32] * 6 + [_f32] + [_vp, _vp, _vp, _i64, _i32] + _ROPE
                                                    + [_vp, _vp, _sz, _vp])
lib.deft_flatten_decode_rope_append_f16.restype = C.c_int
lib.deft_node_decode_rope_append_f16.argtypes = (_QKV + _OUT + _MD6 + [_i32, _i32, _i64, _i32, _i32, _i32, _i32, _f32]
                                                 + [_vp, _vp, _vp, _i64, _i32] + _ROPE + [_vp, _vp, _sz, _vp])
lib.deft_node_decode_rope_append_f16.restype = C.c_int
lib.deft_rope_gather_rows.argtypes = [_vp, _vp, _i64, _i32, _i32, _vp, _vp]
lib.deft_rope_gather_rows.restype = C.c_int
lib.deft_prefill_f16.argtypes = [_vp, _i64, _i64] * 4 + [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp]
lib.deft_prefill_f16.restype = C.c_int
lib.deft_rope_qk_f16.argtypes = [_vp, _i64, _i64, _i32, _vp, _i64, _i64, _i32, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp]
lib.deft_rope_qk_f16.restype = C.c_int
lib.deft_seq_plan_bytes.argtypes = [_i32, _i64, _i32, _i32]
lib.deft_seq_plan_bytes.restype = _sz
lib.deft_seq_workspace_bytes.argtypes = [_i32, _i64, _i32, _i32, _i32]
lib.deft_seq_workspace_bytes.restype = _sz
lib.deft_seq_build_plan.argtypes = [_vp, _i64, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _i64, _i64, _i64, _vp, _i32, _i64, _vp, _sz, _vp]
lib.deft_seq_build_plan.restype = C.c_int
lib.deft_seq_decode_f16.argtypes = _QKV + _OUT + [_vp, _i32, _i64, _i32, _i32, _i32, _f32, _vp, _sz, _vp]
lib.deft_seq_decode_f16.restype = C.c_int
lib.deft_seq_decode_append_f16.argtypes = (_QKV + _OUT + [_vp, _i32, _i64, _i32, _i32, _i32, _f32] + [_vp, _vp, _vp, _i64, _i32]
                                           + [_vp, _sz, _vp])
lib.deft_seq_decode_append_f16.restype = C.c_int
lib.deft_kv_append_f16.argtypes = [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp]
lib.deft_kv_append_f16.restype = C.c_int
lib.deft_flatten_read_partials.argtypes = [_vp, _sz] + [_i32] * 6 + [_vp, _vp, _vp]
lib.deft_flatten_read_partials.restype = C.c_int

lib.deft_md_build.argtypes = [_i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32]
lib.deft_md_build.restype = _i64
lib.deft_md_sizes.argtypes = [_i64, _vp]
lib.deft_md_sizes.restype = C.c_int
lib.deft_md_fetch.argtypes = [_i64] + [_vp] * 13
lib.deft_md_fetch.restype = C.c_int
lib.deft_md_free.argtypes = [_i64]
lib.deft_md_free.restype = C.c_int
lib.deft_tree_create.argtypes = []
lib.deft_tree_create.restype = _i64
_TREE_FUNCS = {
    "deft_tree_free": ([_i64], C.c_int),
    "deft_tree_add_node": ([_i64, _i64, _i64], C.c_int),
    "deft_tree_remove_node": ([_i64, _i64], C.c_int),
    "deft_tree_set_leaf": ([_i64, _i64, C.c_int], C.c_int),
    "deft_tree_branch": ([_i64, _i64, C.c_int, _i64], C.c_int),
    "deft_tree_cut": ([_i64, _i64, _vp, C.c_int, _vp, _vp, _i64, _vp], C.c_int),
    "deft_tree_alloc_step": ([_i64, C.c_int, _vp], C.c_int),
    "deft_tree_append_slots": ([_i64, C.c_int, _vp, _vp], C.c_int),
    "deft_tree_extend_node": ([_i64, _i64, C.c_int, _vp], C.c_int),
    "deft_tree_set_node_kv": ([_i64, _i64, C.c_int, _vp], C.c_int),
    "deft_tree_clear_node_kv": ([_i64, _i64], C.c_int),
    "deft_tree_take_nodes_kv": ([_i64, C.c_int, _vp, _vp, _i64], _i64),
    "deft_tree_node_len": ([_i64, _i64], _i64),
    "deft_tree_node_kv": ([_i64, _i64, _vp, _i64], _i64),
    "deft_tree_node_refs": ([_i64, _i64, _vp, _i64], _i64),
    "deft_tree_path_slots": ([_i64, _i64, _vp, _i64], _i64),
    "deft_tree_leaf_ids": ([_i64, _vp, C.c_int], C.c_int),
    "deft_tree_stats": ([_i64, _vp], C.c_int),
    "deft_tree_build_md": ([_i64, C.c_int, C.c_int, C.c_int], _i64),
    "deft_tree_layout": ([_i64, C.c_int, _vp], C.c_int),
    "deft_tree_layout_fetch": ([_i64, _vp, _vp, _vp, _vp, _vp, _vp], C.c_int),
    "deft_tree_md_sizes": ([_i64, C.c_int, C.c_int, C.c_int, C.c_int, _vp], C.c_int),
    "deft_tree_md_sizes_upto": ([_i64, C.c_int, C.c_int, C.c_int, C.c_int, _vp], C.c_int),
    "deft_tree_md_caps": ([_i64, C.c_int, C.c_int, C.c_int, C.c_int, _vp], C.c_int),
    "deft_tree_dev_scratch_bytes": ([C.c_int, C.c_int, C.c_int], _sz),
    "deft_tree_dev_advance": ([C.c_int, C.c_int, C.c_int] + [_vp] * 9, C.c_int),
    "deft_tree_dev_build_md": ([C.c_int, C.c_int, C.c_int] + [_vp] * 6 + [C.c_int] * 4 + [_vp, _sz] + [_vp] * 12 + [_vp, _vp], C.c_int),
    "deft_tree_dev_build_md_ops": ([C.c_int, C.c_int, C.c_int] + [_vp] * 6 + [C.c_int] * 4 + [_vp, _sz] + [_vp] * 12 + [_vp, _vp] +
                                   [_vp, _i64, _vp, _vp, _vp], C.c_int),
    "deft_tree_dev_apply_ops": ([C.c_int, C.c_int, C.c_int] + [_vp] * 6 + [_vp, _vp, _vp], C.c_int),
    "deft_tree_journal_take": ([_i64, _vp, _i64], _i64),
    "deft_window_supported": ([C.c_int] * 4, C.c_int),
    "deft_window_create": ([C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int], _i64),
    "deft_window_free": ([_i64], C.c_int),
    "deft_window_step": ([_i64, C.c_int, _vp, _i64, _vp, _vp, _i64], _i64),
    "deft_flatten_build_plan_window": ([_vp] * 6 + [C.c_int, C.c_int, _vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _i64, _i64, _i64,
                                        _vp, _sz, _vp], C.c_int),
    "deft_node_build_plan_window": ([_vp] * 6 + [C.c_int, C.c_int, _i64, _vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _i64, _i64,
                                     _i64, _vp, _sz, _vp], C.c_int),
    "deft_window_patch": ([C.c_int, C.c_int, C.c_int] + [_vp] * 6 + [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int,
                           C.c_int, _i64, _i64, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp], C.c_int),
    "deft_stage_fetch": ([_vp, C.c_int, C.c_int, _vp, _vp, _vp], C.c_int),
    "deft_stage_copy": ([_vp, C.c_int, C.c_int, _vp, C.c_size_t, _vp], C.c_int),
}
for _f, (_a, _r) in _TREE_FUNCS.items():
    getattr(lib, _f).argtypes = _a
    getattr(lib, _f).restype = _r

EXPORTED = (
    "deft_abi_version", "deft_last_error", "deft_supported", "deft_plan_variant",
    "deft_flatten_workspace_bytes", "deft_flatten_plan_bytes", "deft_flatten_build_plan", "deft_flatten_build_plan_dims",
    "deft_flatten_decode_f16", "deft_flatten_decode_append_f16", "deft_flatten_stage1_f16",
    "deft_flatten_read_partials", "deft_node_workspace_bytes", "deft_node_plan_bytes", "deft_node_build_plan", "deft_node_build_plan_dims",
    "deft_node_decode_f16", "deft_node_decode_append_f16", "deft_flatten_decode_rope_append_f16", "deft_node_decode_rope_append_f16", "deft_rope_gather_rows",
    "deft_prefill_f16", "deft_rope_qk_f16", "deft_seq_plan_bytes", "deft_seq_workspace_bytes", "deft_seq_build_plan", "deft_seq_decode_f16", "deft_seq_decode_append_f16",
    "deft_kv_append_f16", "deft_probe_stream_read", "deft_md_build", "deft_md_sizes", "deft_md_fetch", "deft_md_free",
    "deft_tree_create",
) + tuple(_TREE_FUNCS)

_ERR_NAMES = {-1: "DEFT_EINVAL", -2: "DEFT_EUNSUPPORTED", -3: "DEFT_EHIP", -4: "DEFT_EWORKSPACE"}


def tensor_version(t) -> int:
    """In-place version counter of a tensor, or -1 for inference tensors (created under torch.inference_mode(): they
    keep none, and `t._version` raises).  Callers that cache per-step plans by (data_ptr, version) must not cache on -1."""
    try:
        return t._version
    except RuntimeError:
        return -1


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib.deft_last_error().decode("utf-8", "replace")
        raise DeftLibraryError(f"{what} failed: {_ERR_NAMES.get(rc, rc)}: {msg}")
