"""No-GPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/branchgan.h declares, and
the ctypes signature table covers exactly that set.  No compute call is made here."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "branchgan.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from branchgan import _capi
    assert os.path.exists(_capi.LIB_PATH), "build first: python __graft_entry__.py"
    decl = declared_symbols()
    assert len(decl) >= 17
    assert sorted(_capi.SIGNATURES) == decl                      # binding table == header
    assert sorted(_capi.exported_symbols()) == decl              # library == header
    lib = _capi.lib()
    assert lib.bg_version() >= 1
    assert lib.bg_last_error() is not None


def test_workspace_queries_are_host_only():
    from branchgan import _capi
    lib = _capi.lib()
    assert lib.bg_topn_workspace_bytes(8192, 50304, 16) == 0                       # enough rows: no split, no workspace
    assert lib.bg_topn_workspace_bytes(256, 50304, 4) == 256 * 8 * 4 * 8           # 8 splits x n x u64
    assert lib.bg_topp_workspace_bytes(256, 50304, 4) == (256 + 1) * 4             # list form: count + one id per row left over
    assert lib.bg_topp_workspace_bytes(256, 1 << 20, 4) == 0                       # beyond the list form's row length: none


def test_argument_validation_happens_before_any_launch():
    """Bad shapes are rejected on the host (no device needed): error code + message."""
    import ctypes
    from branchgan import _capi
    lib = _capi.lib()
    rc = lib.bg_topn_select(None, 0, 4, 100, 100, None, None, 0, None, 0, 0.0, 1, None, None, None, 0, None)
    assert rc == 1 and b"null" in lib.bg_last_error()
    sh = _capi.AttnShape()
    sh.dtype, sh.B, sh.H, sh.NQ, sh.K, sh.Dh, sh.S_text = 0, 1, 1, 1, 4, 48, 8
    rc = lib.bg_branch_attn_fwd(ctypes.byref(sh), None, None, None, None, None, None, None, None, None, None, None)
    assert rc == 1 and b"Dh=48" in lib.bg_last_error()


def test_ops_refuse_cpu_tensors():
    import torch
    from branchgan import ops
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.topn_select(torch.randn(2, 10), 1)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.softmax_stats(torch.randn(2, 10), torch.zeros(2, 1, dtype=torch.long))
