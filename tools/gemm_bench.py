#!/usr/bin/env python3
"""GPU micro-benchmark of the GEMM entry points at the shapes of the B=32 @518 forward.
Interleaved rounds in one process; prints TFLOP/s (median) per shape.  Tuning aid only."""
import statistics
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cognitive_aim_depth_estimation_amd._lib import call, ptr, stream_ptr  # noqa: E402

M = int(os.environ.get("M", "43840"))
dt, tdt = (0, torch.bfloat16) if os.environ.get("BF16") else (1, torch.float16)
dev = "cuda"
torch.manual_seed(0)


def mk(m, k):
    if os.environ.get("ZERO"):       # all-zero operands: same instruction stream at the lowest switching power
        return torch.zeros(m, k, device=dev, dtype=tdt)
    return (torch.randn(m, k, device=dev) * float(os.environ.get("SCALE", "0.5"))).to(tdt)


shapes = {
    "qk    [M,1536,768]": ("bias", 1536, 768),
    "fc1   [M,3072,768]": ("gelu", 3072, 768),
    "out   [M,768,768]": ("res", 768, 768),
    "fc2   [M,768,3072]": ("res", 768, 3072),
    "qkv   [M,2304,768]": ("bias3", 2304, 768),
}
if os.environ.get("EXTRA"):      # same shape, different epilogue: isolates the epilogue cost
    shapes["fc1b  [M,3072,768]"] = ("bias", 3072, 768)
    shapes["qkg   [M,1536,768]"] = ("gelu", 1536, 768)
WSB = torch.empty(call("cogaim_gemm_workspace_bytes"), device=dev, dtype=torch.uint8)
USE_WS = os.environ.get("NO_WS") is None
wsp = (WSB.data_ptr(), WSB.numel()) if USE_WS else (None, 0)
if os.environ.get("SHAPES"):
    shapes = {k: v for k, v in shapes.items() if k.split()[0] in os.environ["SHAPES"].split(",")}
ROUNDS = int(os.environ.get("ROUNDS", "7"))
bufs = {}
for name, (kind, N, K) in shapes.items():
    A, W = mk(M, K), mk(N, K)
    bias, ls = torch.randn(N, device=dev), torch.ones(N, device=dev)
    out16 = torch.empty(M, N, device=dev, dtype=tdt)
    x = torch.zeros(M, N, device=dev)
    bufs[name] = (A, W, bias, ls, out16, x)


def run(name):
    kind, N, K = shapes[name]
    A, W, bias, ls, out16, x = bufs[name]
    st = stream_ptr()
    if kind == "bias3":
        call("cogaim_gemm_bias_half", ptr(A), ptr(W), ptr(bias), ptr(out16), M, N, K, K, K, N, N // 12 * 4, 0.18, dt, *wsp, None, st)
    elif kind == "bias":
        call("cogaim_gemm_bias_half", ptr(A), ptr(W), ptr(bias), ptr(out16), M, N, K, K, K, N, N // 8 * 4, 0.18, dt, *wsp, None, st)
    elif kind == "gelu":
        call("cogaim_gemm_bias_gelu_half", ptr(A), ptr(W), ptr(bias), ptr(out16), M, N, K, K, K, N, dt, *wsp, None, st)
    elif kind == "res":
        call("cogaim_gemm_residual_f32", ptr(A), ptr(W), ptr(bias), ptr(ls), ptr(x), M, N, K, K, K, N, dt, *wsp, st)


def check(name):
    kind, N, K = shapes[name]
    A, W, bias, ls, out16, x = bufs[name]
    rows = torch.tensor([0, 1, 127, 128, 1369, 1370, 20000, M - 65, M - 1], device=dev)
    ref = A[rows].float() @ W.float().t() + bias
    x.zero_()
    run(name)
    torch.cuda.synchronize()
    if kind == "bias3":
        ref[:, : N // 3] *= 0.18
        got = out16[rows].float()
    elif kind == "bias":
        ref[:, : N // 2] *= 0.18
        got = out16[rows].float()
    elif kind == "gelu":
        ref = torch.nn.functional.gelu(ref)
        got = out16[rows].float()
    elif kind == "res":
        got = x[rows]
    err = float((got - ref).abs().max() / ref.abs().max())
    return err


algos = [int(a) for a in sys.argv[1:]] or [0]
for algo in algos:
    if algo or len(algos) > 1:
        call("cogaim_set_gemm_algo", algo)
    errs = {n: check(n) for n in shapes} if algo < 100 else {}
    print(f"algo {algo} max-rel-err:", {k.split()[0]: f"{v:.1e}" for k, v in errs.items()})
times = {(a, n): [] for a in algos for n in shapes}
for rnd in range(ROUNDS):
    for algo in algos:
        if algo or len(algos) > 1:
            call("cogaim_set_gemm_algo", algo)
        for name in shapes:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            run(name)
            e0.record()
            for _ in range(5):
                run(name)
            e1.record()
            torch.cuda.synchronize()
            times[(algo, name)].append(e0.elapsed_time(e1) / 5)
for algo in algos:
    tot = 0.0
    for name, (kind, N, K) in shapes.items():
        ms = statistics.median(times[(algo, name)])
        tot += ms * (12 if True else 1)
        print(f"algo {algo} {name}: {ms * 1000:7.1f} us  {2.0 * M * N * K / ms / 1e9:7.1f} TF")
    print(f"algo {algo} sum x12 layers: {tot:.2f} ms")
