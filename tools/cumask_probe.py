#!/usr/bin/env python3
"""Probe: streams restricted to a subset of the CUs (hipExtStreamCreateWithCUMask).  Does a kernel launched on such a
stream run on the masked CUs only, which mask bits make up an XCD, do two masked streams run side by side, and does a
captured graph replayed on a masked stream keep the restriction?"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cognitive_aim_depth_estimation_amd._lib import call, ptr  # noqa: E402

hip = None
for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
    try:
        hip = ctypes.CDLL(name)
        break
    except OSError:
        pass
assert hip is not None
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


torch.cuda.init()
x = torch.randn(21920, 768, device="cuda")
g, b = torch.ones(768, device="cuda"), torch.zeros(768, device="cuda")
h = torch.empty(21920, 768, device="cuda", dtype=torch.bfloat16)
A = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)


def work(st):
    with torch.cuda.stream(st):
        for _ in range(20):
            call("cogaim_layernorm", ptr(x), ptr(g), ptr(b), ptr(h), 21920, 768, 1e-6, 0, st.cuda_stream)


def mm(st, n=4):
    with torch.cuda.stream(st):
        for _ in range(n):
            torch.mm(A, A)


def timed(fn, *sts):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for st in sts:
        fn(st)
    for st in sts:
        st.synchronize()
    return (time.perf_counter() - t0) * 1e3


full = torch.cuda.Stream()
masks = {"first128": list(range(128)), "even": list(range(0, 256, 2)), "mod8<4": [i for i in range(256) if i % 8 < 4],
         "first32": list(range(32)), "mod8==0": list(range(0, 256, 8))}
for _ in range(2):
    mm(full)
torch.cuda.synchronize()
print("mm full: %.2f ms" % timed(mm, full))
for name, bits in masks.items():
    st = masked_stream(bits)
    mm(st, 1)
    print(f"mm {name} ({len(bits)} CUs): {timed(mm, st):.2f} ms;  layernorm x20: {timed(work, st):.2f} ms (full {timed(work, full):.2f})")
a, bb = masked_stream([i for i in range(256) if i % 8 < 4]), masked_stream([i for i in range(256) if i % 8 >= 4])
mm(a, 1); mm(bb, 1)
print("two complementary masked streams, mm on each: %.2f ms (one alone %.2f)" % (timed(mm, a, bb), timed(mm, a)))
# graph on a masked stream
gr = torch.cuda.CUDAGraph()
with torch.cuda.stream(a):
    mm(a, 1)
    a.synchronize()
    with torch.cuda.graph(gr, stream=a):
        for _ in range(4):
            torch.mm(A, A)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); gr.replay(); a.synchronize(); t1 = (time.perf_counter() - t0) * 1e3
with torch.cuda.stream(full):
    t0 = time.perf_counter(); gr.replay(); full.synchronize(); t2 = (time.perf_counter() - t0) * 1e3
print("graph of 4 mm captured on a masked stream: replay on it %.2f ms, replay on an unmasked stream %.2f ms" % (t1, t2))
