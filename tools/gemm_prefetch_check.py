#!/usr/bin/env python3
"""Tools build only (tools/wt_build.sh): every GEMM entry point with the fragment-prefetch kernel taking the whole
rounds of tiles (cogaim_set_gemm_algo(5)) against the 256x256 ping-pong kernel alone (4) on the same operands --
outputs must be equal bit for bit."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cognitive_aim_depth_estimation_amd._lib import call, ptr, stream_ptr  # noqa: E402

dev = "cuda"
torch.manual_seed(0)
WSB = torch.empty(call("cogaim_gemm_workspace_bytes"), device=dev, dtype=torch.uint8)
WS = (WSB.data_ptr(), WSB.numel())
ok = True


def both(fn):
    outs = []
    for algo in (4, 5):
        call("cogaim_set_gemm_algo", algo)
        if algo == 5 and os.environ.get("WIDE"):          # the four-wave kernel in place of gemm256f_kernel
            call("cogaim_set_gemm_algo", 13)
        outs.append(fn())
        torch.cuda.synchronize()
    return outs


def report(name, outs):
    global ok
    same = all(torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a.view(torch.int16),
                           b.view(torch.int32) if b.dtype == torch.float32 else b.view(torch.int16))
               for a, b in zip(outs[0], outs[1]))
    nan = sum(int(torch.isnan(t.float()).sum()) for t in outs[1])
    print(f"{name}: bit-identical {same}, nan {nan}", flush=True)
    ok = ok and same and nan == 0


for dt, tdt in ((0, torch.bfloat16), (1, torch.float16)):
    def mk(m, k, s=0.5):
        return (torch.randn(m, k, device=dev) * s).to(tdt)
    for kind, M, N, K in (("bias", 21920, 2304, 768), ("gelu", 21920, 3072, 768), ("gelu", 43840, 3072, 768),
                          ("res", 21920, 768, 768), ("res", 21920, 768, 3072), ("res", 43840, 768, 3072),
                          ("bias", 43808, 768, 768), ("gelu", 33000, 3072, 832), ("bias", 70000, 1000, 256),
                          ("res", 17000, 1024, 4096)):
        A, W, bias, ls = mk(M, K), mk(N, K), torch.randn(N, device=dev), torch.rand(N, device=dev) + 0.5
        x0 = torch.randn(M, N, device=dev)

        def run():
            if kind == "res":
                x = x0.clone()
                call("cogaim_gemm_residual_f32", ptr(A), ptr(W), ptr(bias), ptr(ls), ptr(x), M, N, K, K, K, N, dt, *WS, stream_ptr())
                return [x]
            out = torch.full((M, N), float("nan"), device=dev, dtype=tdt)
            if kind == "bias":
                call("cogaim_gemm_bias_half", ptr(A), ptr(W), ptr(bias), ptr(out), M, N, K, K, K, N, (N // 12) * 4, 0.18, dt, *WS, None, stream_ptr())
            else:
                call("cogaim_gemm_bias_gelu_half", ptr(A), ptr(W), ptr(bias), ptr(out), M, N, K, K, K, N, dt, *WS, None, stream_ptr())
            return [out]
        report(f"dtype {dt} {kind} M={M} N={N} K={K}", both(run))
    # batched fp32 scores: S[z] = Y[z] X[z]^T, ragged N
    for Nn, B in ((1369, 16), (1369, 32), (5476, 2)):
        D = 768
        Y, X = mk(B * Nn, D), mk(B * Nn, D)
        ldS = (Nn + 3) & ~3

        def run_b():
            S = torch.zeros(B, Nn, ldS, device=dev)
            call("cogaim_gemm_batched_f32", ptr(Y), ptr(X), ptr(S), Nn, Nn, D, D, D, ldS, Nn * D, Nn * D, Nn * ldS, B, dt, stream_ptr())
            return [S[:, :, :Nn].contiguous()]        # (the row padding past N holds unspecified values)
        report(f"dtype {dt} batched scores N={Nn} B={B}", both(run_b))
    # focal scores reduced in registers (large grids)
    Nn, B, D = 5476, 2, 768
    Y, X = mk(B * Nn, D, 0.2), mk(B * Nn, D, 0.2)
    parts, blocks = call("cogaim_focal_rowstats_parts", Nn), call("cogaim_focal_scores_colsum_blocks", Nn)

    def run_f():
        pmax = torch.zeros(B, Nn, parts, device=dev); psum = torch.zeros(B, Nn, parts, device=dev)
        call("cogaim_focal_scores_rowstats", ptr(Y), ptr(X), ptr(pmax), ptr(psum), Nn, D, D, Nn * D, B, dt, stream_ptr())
        rmax = pmax.max(2).values.contiguous()
        rinv = (1.0 / (psum * torch.exp2(pmax - rmax[..., None])).sum(2)).contiguous()
        part = torch.zeros(B, blocks, Nn, device=dev)
        call("cogaim_focal_scores_colsum", ptr(Y), ptr(X), ptr(rmax), ptr(rinv), None, ptr(part), Nn, D, D, Nn * D, B, dt, stream_ptr())
        return [pmax, psum, part]
    report(f"dtype {dt} focal rowstats + colsum N={Nn}", both(run_f))
    # patch embedding (K = 640)
    Bp, P, Dm, Kp = 16, 1369, 768, 640
    cols, wp = mk(Bp * P, Kp), mk(Dm, Kp)
    bias, pos = torch.randn(Dm, device=dev), torch.randn(P + 1, Dm, device=dev)

    def run_p():
        x = torch.zeros(Bp, P + 1, Dm, device=dev)
        call("cogaim_gemm_patch_embed", ptr(cols), ptr(wp), ptr(bias), ptr(pos), ptr(x), Bp, P, Dm, Kp, dt, *WS, stream_ptr())
        return [x]
    report(f"dtype {dt} patch embed", both(run_p))
print("OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
