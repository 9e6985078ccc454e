#!/usr/bin/env python3
"""Per-workgroup timeline of one persistent-GEMM launch (library built with -DCOGAIM_STAMPS, selected by
COGAIM_LIB): for every work item the 100-MHz clock at item start, after its first K-tile, after its last
K-tile and after the epilogue was issued.  Tuning aid only."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cognitive_aim_depth_estimation_amd import _lib  # noqa: E402
from cognitive_aim_depth_estimation_amd._lib import call, ptr, stream_ptr  # noqa: E402

M, N, K = int(os.environ.get("M", "43840")), int(os.environ.get("N", "3072")), int(os.environ.get("K", "768"))
kind = os.environ.get("KIND", "gelu")
dev = "cuda"
torch.manual_seed(0)
A = (torch.randn(M, K, device=dev) * 0.5).half()
W = (torch.randn(N, K, device=dev) * 0.5).half()
bias = torch.randn(N, device=dev)
ls = torch.ones(N, device=dev)
out = torch.empty(M, N, device=dev, dtype=torch.float16)
x = torch.zeros(M, N, device=dev)
ws = torch.empty(call("cogaim_gemm_workspace_bytes"), device=dev, dtype=torch.uint8)
if len(sys.argv) > 1:
    call("cogaim_set_gemm_algo", int(sys.argv[1]))


def run():
    if kind == "gelu":
        call("cogaim_gemm_bias_gelu_half", ptr(A), ptr(W), ptr(bias), ptr(out), M, N, K, K, K, N, 1, ws.data_ptr(), ws.numel(), None, stream_ptr())
    elif kind == "bias":
        call("cogaim_gemm_bias_half", ptr(A), ptr(W), ptr(bias), ptr(out), M, N, K, K, K, N, 0, 1.0, 1, ws.data_ptr(), ws.numel(), None, stream_ptr())
    else:
        call("cogaim_gemm_residual_f32", ptr(A), ptr(W), ptr(bias), ptr(ls), ptr(x), M, N, K, K, K, N, 1, ws.data_ptr(), ws.numel(), None, stream_ptr())


for _ in range(3):
    run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
run()
e1.record()
torch.cuda.synchronize()
print(f"launch (kernel + fix-up): {e0.elapsed_time(e1) * 1e3:.1f} us")
buf = np.zeros(256 * 12 * 4, dtype=np.int64)
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.cogaim_debug_stamps.argtypes = [ctypes.c_void_p]
assert lib.cogaim_debug_stamps(buf.ctypes.data) == 0
s = buf.reshape(256, 12, 4).astype(np.float64) / 100.0   # us
s = s[s[:, 0, 0] > 0]                                    # workgroups that ran
print(len(s), "workgroups")
t0 = s[:, 0, 0].min()
s -= t0
items = int(os.environ.get("ITEMS", "8"))
print("item: start(min/max over WGs)  first-K-tile  rest-of-mainloop  epilogue-issue   [us, mean over WGs]")
for j in range(items):
    w = s[s[:, j, 0] >= 0]                               # workgroups that have an item j
    if not len(w):
        break
    st = w[:, j, 0]
    print(f"{j}: ({len(w):3d} WGs) start {st.min():7.2f}..{st.max():7.2f}  first {np.mean(w[:, j, 1] - w[:, j, 0]):6.2f}  "
          f"rest {np.mean(w[:, j, 2] - w[:, j, 1]):6.2f}  epi {np.mean(w[:, j, 3] - w[:, j, 2]):6.2f}  "
          f"end {w[:, j, 3].min():7.2f}..{w[:, j, 3].max():7.2f}")
last = s[:, :, 3].max(axis=1)
print(f"last epilogue issued: min {last.min():.1f} max {last.max():.1f} us")
print("WG 0, 1, 8, 100 starts:", [[round(float(v), 1) for v in s[b, :items, 0]] for b in (0, 1, 8, 100) if b < len(s)])
