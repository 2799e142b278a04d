"""The drop-in claim executed (tools/check_dropin_with_reference_driver.py): the REFERENCE's own `tree_generate` loop, branch
functions, `Branch_Controller`, template loader and `TreeMetadata.from_tree_cache`, imported unchanged from /root/reference, run on
deft_amd's TreeCache / TreeNode / pools and reproduce, step for step, what they did on the reference's own objects
(tests/golden/replay_*.npz).  Needs the reference checkout: runs in the build container, skips anywhere else (the GPU box)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_dropin_with_reference_driver as chk  # noqa: E402


@pytest.mark.skipif(not chk.available(), reason="/root/reference is not here (build container only)")
def test_the_references_own_driver_runs_unchanged_on_deft_amd_objects():
    done = dict(chk.run(chk.QUICK, verbose=False))
    assert done == {"simple_w6": 39, "keywordToT": 564, "set128ToT": 363, "speculative64": 48, "speculative256": 28}
