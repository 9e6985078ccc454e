"""cogaim_weighted_tokens_static: the entry point that allocates nothing (its partial sums live in a static device
array of the library) against cogaim_weighted_tokens with a caller workspace -- bit for bit, two calls in a row,
the second with a smaller input, plus the size limit."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from cognitive_aim_depth_estimation_amd._lib import call, lib, ptr, stream_ptr  # noqa: E402


def _static():
    return lib().cogaim_weighted_tokens_static            # raw entry: return codes are checked by hand below


def test_static_scratch_entry_point_matches_workspace_one():
    assert torch.cuda.is_available()
    fn = _static()
    rows = call("cogaim_weighted_tokens_rows")
    torch.manual_seed(0)
    calls = 0
    for B, N, D, with_pe in ((32, 1369, 768, True), (3, 100, 768, False)):          # second call: smaller input
        tok = torch.randn(B, N + 1, D, device="cuda")
        w = torch.softmax(torch.randn(B, N, device="cuda"), 1)
        scale = 1 + torch.rand(B, N, device="cuda") if with_pe else None
        pe = torch.randn(N, D, device="cuda") if with_pe else None
        nchunk = (N + rows - 1) // rows
        ws = torch.empty(B * nchunk * D, device="cuda")
        want = torch.empty(B, D, device="cuda")
        call("cogaim_weighted_tokens", tok.data_ptr() + D * 4, (N + 1) * D, ptr(scale), ptr(pe), ptr(w), ptr(want), ptr(ws),
             B, N, D, stream_ptr())
        got = torch.full((B, D), float("nan"), device="cuda")
        torch.cuda.synchronize()
        before = torch.cuda.memory_allocated()
        free_before = torch.cuda.mem_get_info()[0]          # hipMemGetInfo: sees a hipMalloc that torch's counters would not
        rc = fn(tok.data_ptr() + D * 4, (N + 1) * D, ptr(scale), ptr(pe), ptr(w), ptr(got), B, N, D, stream_ptr())
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated() == before
        if calls:       # the first call resolves the static array's address (the runtime may map the code object's data then)
            assert torch.cuda.mem_get_info()[0] >= free_before, (free_before, torch.cuda.mem_get_info()[0])
        calls += 1
        assert torch.equal(got, want), float((got - want).abs().max())
    # beyond the fixed 8 MiB of partial sums: refused, nothing launched
    B, N, D = 64, 1369, 768
    assert B * ((N + rows - 1) // rows) * D > 2 * 1024 * 1024
    out = torch.zeros(B, D, device="cuda")
    assert fn(tok.data_ptr(), 0, None, None, w.data_ptr(), out.data_ptr(), B, N, D, stream_ptr()) == -22
    assert fn(tok.data_ptr(), 0, None, None, w.data_ptr(), out.data_ptr(), 0, N, D, stream_ptr()) == -22
    torch.cuda.synchronize()


def test_static_scratch_entry_point_under_graph_capture():
    """The callers it exists for: a launch sequence under HIP graph capture, where hipMalloc is not allowed.  After one
    eager call (which resolves the static array's address), the entry point is captured and replayed on new inputs."""
    fn = _static()
    rows = call("cogaim_weighted_tokens_rows")
    B, N, D = 4, 333, 768
    tok = torch.randn(B, N + 1, D, device="cuda")
    w = torch.softmax(torch.randn(B, N, device="cuda"), 1)
    got = torch.zeros(B, D, device="cuda")
    assert fn(tok.data_ptr() + D * 4, (N + 1) * D, None, None, ptr(w), ptr(got), B, N, D, stream_ptr()) == 0
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        rc = fn(tok.data_ptr() + D * 4, (N + 1) * D, None, None, ptr(w), ptr(got), B, N, D, stream_ptr())
    assert rc == 0
    tok.normal_()
    w.copy_(torch.softmax(torch.randn(B, N, device="cuda"), 1))
    got.fill_(float("nan"))
    g.replay()
    torch.cuda.synchronize()
    want = torch.empty(B, D, device="cuda")
    ws = torch.empty(B * ((N + rows - 1) // rows) * D, device="cuda")
    call("cogaim_weighted_tokens", tok.data_ptr() + D * 4, (N + 1) * D, None, None, ptr(w), ptr(want), ptr(ws), B, N, D,
         stream_ptr())
    torch.cuda.synchronize()
    assert torch.equal(got, want)
