"""bg_topn_select_static: the entry point that allocates nothing (its partial lists live in a file-scope device array).

Called through the C-ABI and compared, bit for bit, with bg_topn_select (the existing entry point, which takes the same
partial lists in a caller workspace) on two calls in a row, the second with a smaller input, so the array still holds the
first call's lists when the second one runs.  Collected last (file name) so a failure here hides no other test."""
import pytest
import torch

from branchgan import _capi as C

pytestmark = pytest.mark.gpu


def _both(scores, noise, blocked, n):
    """ids / keys of the workspace entry point and of the static one, same inputs."""
    lib = C.lib()
    rows, V = scores.shape
    nb = 0 if blocked is None else blocked.shape[1]
    out = []
    for static in (False, True):
        ids = torch.full((rows, n), -7, dtype=torch.int64, device="cuda")
        keys = torch.full((rows, n), -7.0, dtype=torch.float32, device="cuda")
        common = (C.ptr(scores), C.dtype_code(scores), rows, V, scores.stride(0), None,
                  C.ptr(noise), noise.stride(0) if noise is not None else 0,
                  C.ptr(blocked), nb, 0.0, n, C.ptr(ids), C.ptr(keys))
        if static:
            C.check(lib.bg_topn_select_static(*common, C.stream_ptr()), "bg_topn_select_static")
        else:
            ws_bytes = lib.bg_topn_workspace_bytes(rows, V, n)
            ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device="cuda")
            C.check(lib.bg_topn_select(*common, C.ptr(ws), ws_bytes, C.stream_ptr()), "bg_topn_select")
        torch.cuda.synchronize()
        out.append((ids.cpu(), keys.cpu()))
    return out


def test_static_entry_matches_workspace_entry_two_calls_in_a_row():
    g = torch.Generator().manual_seed(11)
    # call 1: 300 rows x 50304 -> 7 splits, 300 * 7 * 16 partial items; noise + blocked ids (SAMPLE_NO_REPLACEMENT at step 1)
    # call 2: a SMALLER input (37 rows x 9001, odd V: unaligned rows, 2 splits), the array still holds call 1's lists
    for rows, V, n, dtype, use_noise in ((300, 50304, 16, torch.float32, True), (37, 9001, 5, torch.bfloat16, False)):
        scores = torch.softmax(3 * torch.randn(rows, V, generator=g), -1).to(dtype).cuda()
        noise = (-torch.log(torch.rand(rows, V, generator=g) + 1e-14)).cuda() if use_noise else None
        blocked = torch.randint(0, V, (rows, 1), generator=g).cuda()
        assert C.lib().bg_topn_workspace_bytes(rows, V, n) > 0          # the split path, i.e. the one that uses the array
        (ids_w, keys_w), (ids_s, keys_s) = _both(scores, noise, blocked, n)
        assert torch.equal(ids_w, ids_s)
        assert torch.equal(keys_w.view(torch.int32), keys_s.view(torch.int32))
        assert (ids_s >= 0).all() and (ids_s < V).all()
        # and both agree with a plain sort of the same keys (ties are absent with random data; blocked ids score 0)
        key = scores.float()
        key.scatter_(1, blocked, 0.0)
        if noise is not None:
            key = key / noise
        top = torch.sort(key, dim=-1, descending=True, stable=True)
        assert torch.equal(top.values[:, :n].cpu().view(torch.int32), keys_s.view(torch.int32))


def test_static_entry_without_split_and_size_limit():
    lib = C.lib()
    g = torch.Generator().manual_seed(12)
    scores = torch.randn(2048, 1031, generator=g).cuda()                # rows >= 2048: one workgroup per row, no partial lists
    (ids_w, _), (ids_s, _) = _both(scores, None, None, 3)
    assert torch.equal(ids_w, ids_s)
    # beyond the array: 2047 rows x 2 splits x 64 = 262016 items > BG_TOPN_STATIC_ITEMS -> refused before any launch
    big = torch.randn(2047, 8192, generator=g).cuda()
    ids = torch.empty((2047, 64), dtype=torch.int64, device="cuda")
    rc = lib.bg_topn_select_static(C.ptr(big), 0, 2047, 8192, 8192, None, None, 0, None, 0, 0.0, 64, C.ptr(ids), None, C.stream_ptr())
    assert rc == 3 and b"partial items need" in lib.bg_last_error()


def test_static_entry_runs_under_graph_capture():
    """What the entry point exists for: callers that cannot allocate.  A hipMalloc / hipFree, a synchronisation or an allocating
    library call inside a stream capture is an error, so the static call is captured into a HIP graph (after one eager call on
    the device, which takes the array's address with hipGetSymbolAddress), replayed on fresh inputs and compared bit for bit
    with the workspace entry point."""
    lib = C.lib()
    g = torch.Generator().manual_seed(13)
    rows, V, n = 128, 50304, 8                                           # 8 splits per row: the partial lists are in use
    scores = torch.randn(rows, V, generator=g).cuda()
    ids = torch.full((rows, n), -7, dtype=torch.int64, device="cuda")
    keys = torch.full((rows, n), -7.0, dtype=torch.float32, device="cuda")
    assert lib.bg_topn_workspace_bytes(rows, V, n) > 0

    def static_call():
        C.check(lib.bg_topn_select_static(C.ptr(scores), C.dtype_code(scores), rows, V, scores.stride(0), None, None, 0, None, 0, 0.0, n,
                                          C.ptr(ids), C.ptr(keys), C.stream_ptr()), "bg_topn_select_static")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        static_call()                                                    # eager: resolves the symbol address on this device
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):                           # default error mode: any allocation or sync raises
        static_call()
    scores.copy_(torch.randn(rows, V, generator=g))                      # new inputs at the captured addresses
    ids.fill_(-7)
    graph.replay()
    torch.cuda.synchronize()
    want_ids = torch.empty_like(ids)
    want_keys = torch.empty_like(keys)
    ws_bytes = lib.bg_topn_workspace_bytes(rows, V, n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    C.check(lib.bg_topn_select(C.ptr(scores), C.dtype_code(scores), rows, V, scores.stride(0), None, None, 0, None, 0, 0.0, n,
                               C.ptr(want_ids), C.ptr(want_keys), C.ptr(ws), ws_bytes, C.stream_ptr()), "bg_topn_select")
    torch.cuda.synchronize()
    assert torch.equal(ids, want_ids) and torch.equal(keys.view(torch.int32), want_keys.view(torch.int32))
    assert torch.equal(ids[:, 0], scores.argmax(-1))
