"""deft_amd — MI355X-native paged tree-attention decode (DeFT-Flatten / DeFT-Node).

One path of LINs-lab/DeFT, rebuilt from scratch for gfx950 behind the reference's
own operator surface.  Importing the package loads libdeft_amd.so and fails loudly
if it has not been built; nothing here falls back to PyTorch or the CPU.
"""
from ._lib import LIB_PATH, DeftLibraryError, lib  # noqa: F401
from . import branch_func_example, data_loader, tree_generate  # noqa: F401  (modules under the reference's names: loader, branch functions, driver loop)
from .context_attention import context_attention_fwd  # noqa: F401
from .deft_attention import DeFTAttention  # noqa: F401
from .forest import Forest, concat_metadata_host  # noqa: F401
from .forward_mode import ForwardMode, InputMetadata, forward_mode_from_cli  # noqa: F401
from .memory_pool import ReqToTokenPool, TokenToKVPool  # noqa: F401
from .rotary_embedding import RotaryEmbedding, get_rope  # noqa: F401
from .session import DecodeSession, FlattenDecodeSession  # noqa: F401
from .token_attention import token_attention_fwd  # noqa: F401
from .tree_attention import kv_append, tree_attention_fwd, tree_attention_subtree_fwd  # noqa: F401
from .tree_cache import (  # noqa: F401
    BLOCK_CONFIG,
    KVCacheUpdater,
    TreeCache,
    TreeMetadata,
    TreeNode,
    get_global_tree_cache,
    get_global_tree_metadata,
    register_tree_cache,
    register_tree_metadata,
    unregister_tree_cache,
    unregister_tree_metadata,
)

__version__ = "0.1.0"
