#!/usr/bin/env python3
"""Shader clock the chip holds INSIDE the persistent GEMM (library built with -DCOGAIM_STAMPS, selected by COGAIM_LIB):
every workgroup stamps the shader-cycle counter and the 100-MHz counter at entry and exit; the clock is the ratio,
taken after two seconds of back-to-back launches on random operands (median over workgroups).  Writes one JSON line.
Tuning / measurement aid only."""
import ctypes, json, os, statistics, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cognitive_aim_depth_estimation_amd import _lib
from cognitive_aim_depth_estimation_amd._lib import call, ptr, stream_ptr
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.cogaim_debug_clock.argtypes = [ctypes.c_void_p]
out = {}
for name, (M, N, K, kind) in {"fc1 [21920,3072,768] gelu": (21920, 3072, 768, "gelu"), "fc2 [21920,768,3072] residual": (21920, 768, 3072, "res"),
                              "qkv [21920,2304,768] bias": (21920, 2304, 768, "bias")}.items():
    for dname, tdt, di in (("bf16", torch.bfloat16, 0), ("fp16", torch.float16, 1)):
        for data in ("random", "zeros"):
            torch.manual_seed(0)
            A = (torch.randn(M, K, device="cuda") * 0.5).to(tdt) if data == "random" else torch.zeros(M, K, device="cuda", dtype=tdt)
            W = (torch.randn(N, K, device="cuda") * 0.05).to(tdt) if data == "random" else torch.zeros(N, K, device="cuda", dtype=tdt)
            bias, ls = torch.randn(N, device="cuda"), torch.ones(N, device="cuda")
            ROT = int(os.environ.get("ROTATE", "1"))       # >1: rotate operands / outputs through buffers no cache keeps
            As = [A] + [A.clone() for _ in range(ROT - 1)]
            o16s = [torch.empty(M, N, device="cuda", dtype=tdt) for _ in range(ROT)]
            o16 = o16s[0]
            state = {"i": 0}
            x = torch.zeros(M, N, device="cuda")
            ws = torch.empty(call("cogaim_gemm_workspace_bytes"), device="cuda", dtype=torch.uint8)
            def run():
                state["i"] = (state["i"] + 1) % ROT
                A, o16 = As[state["i"]], o16s[state["i"]]
                if kind == "gelu":
                    call("cogaim_gemm_bias_gelu_half", ptr(A), ptr(W), ptr(bias), ptr(o16), M, N, K, K, K, N, di, ws.data_ptr(), ws.numel(), None, stream_ptr())
                elif kind == "bias":
                    call("cogaim_gemm_bias_half", ptr(A), ptr(W), ptr(bias), ptr(o16), M, N, K, K, K, N, 0, 1.0, di, ws.data_ptr(), ws.numel(), None, stream_ptr())
                else:
                    call("cogaim_gemm_residual_f32", ptr(A), ptr(W), ptr(bias), ptr(ls), ptr(x), M, N, K, K, K, N, di, ws.data_ptr(), ws.numel(), stream_ptr())
            t_end = time.perf_counter() + (2.0 if data == "random" else 1.0)
            n = 0
            while time.perf_counter() < t_end:
                for _ in range(50):
                    run()
                torch.cuda.synchronize(); n += 50
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                run()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 50
            buf = np.zeros(4 * 256 * 4, dtype=np.int64)
            assert lib.cogaim_debug_clock(buf.ctypes.data) == 0
            b = buf.reshape(4, 256, 4)[{"gelu": 0, "bias": 1, "res": 2}[kind]]
            b = b[(b[:, 3] > b[:, 1])]
            ghz = [float((r[2] - r[0]) / ((r[3] - r[1]) * 10.0)) for r in b]       # cycles per 10 ns -> GHz
            clk = statistics.median(ghz)
            tf = 2.0 * M * N * K / us / 1e6
            out[f"{name} {dname} {data}"] = {"us_per_launch": round(us, 1), "tflops": round(tf, 1), "clock_ghz": round(clk, 3),
                                            "frac_of_2.5PF": round(tf / 2500, 4), "frac_of_peak_at_clock": round(tf / (2500 * clk / 2.4), 4)}
            print(name, dname, data, out[f"{name} {dname} {data}"], flush=True)
print(json.dumps(out))
