#!/usr/bin/env python3
"""GPU box: the weight-gradient kernels alone - the ring kernel (wgrad_ring.hip, forced: EEM_WGRAD_RING=all), the tile kernel with its
products as bf16 pieces (wgrad_enc.hip, EEM_NO_WGRAD_RING=1) and with fp32 MFMAs (+ EEM_NO_WGRAD_BX3=1) - at the shapes of
the EEMFlow training step (C3: 346x260 batch 32 -> 64 images; C4: 1280x720 batch 8 -> 16 images) and of E-RAFT's update block / encoders
at 640x480 batch 4.  tools/wgrad_bench.py [reps]   Prints us per launch, TFLOP/s, and the ratio."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eemflow_amd import _lib  # noqa: E402

DEV = torch.device("cuda", 0)
SHAPES = [
    # name, cin, cout, kh, kw, stride, n, hin, win
    ("C3 pconv1_2 16->16", 16, 16, 3, 3, 1, 64, 160, 192),
    ("C3 pconv2_1 16->32 s2", 16, 32, 3, 3, 2, 64, 160, 192),
    ("C3 pconv2_2 32->32", 32, 32, 3, 3, 1, 64, 80, 96),
    ("C3 pconv3_1 32->64 s2", 32, 64, 3, 3, 2, 64, 80, 96),
    ("C3 pconv3_2 64->64", 64, 64, 3, 3, 1, 64, 40, 48),
    ("C4 pconv1_2 16->16", 16, 16, 3, 3, 1, 16, 384, 640),
    ("C4 pconv2_1 16->32 s2", 16, 32, 3, 3, 2, 16, 384, 640),
    ("C4 pconv2_2 32->32", 32, 32, 3, 3, 1, 16, 192, 320),
    ("C4 pconv3_1 32->64 s2", 32, 64, 3, 3, 2, 16, 192, 320),
    ("C4 pconv3_2 64->64", 64, 64, 3, 3, 1, 16, 96, 160),
    ("ERAFTcall gru 1x5 128->128", 128, 128, 1, 5, 1, 4, 60, 80),
    ("ERAFTcall gru 5x1 128->128", 128, 128, 5, 1, 1, 4, 60, 80),
    ("ERAFTcall heads 128->256", 128, 256, 3, 3, 1, 4, 60, 80),
    ("ERAFTcall convc2 256->192", 256, 192, 3, 3, 1, 4, 60, 80),
    ("ERAFTcall conv 192->126", 192, 126, 3, 3, 1, 4, 60, 80),
    ("ERAFTcall conv 64->126", 64, 126, 3, 3, 1, 4, 60, 80),
    ("ERAFTcall convf2 128->64", 128, 64, 3, 3, 1, 4, 60, 80),
    ("ERAFTcall enc 64->64 b4", 64, 64, 3, 3, 1, 4, 240, 320),
    ("ERAFTcall enc 64->64 b8", 64, 64, 3, 3, 1, 8, 240, 320),
    ("ERAFTcall enc 96->96 b4", 96, 96, 3, 3, 1, 4, 120, 160),
    ("ERAFTcall enc 96->96 b8", 96, 96, 3, 3, 1, 8, 120, 160),
    ("ERAFTcall enc 128->128 b4", 128, 128, 3, 3, 1, 4, 60, 80),
    ("ERAFTcall enc 128->128 b8", 128, 128, 3, 3, 1, 8, 60, 80),
    ("ERAFTcall enc 64->96 s2 b4", 64, 96, 3, 3, 2, 4, 240, 320),
    ("ERAFTcall enc 64->96 s2 b8", 64, 96, 3, 3, 2, 8, 240, 320),
    ("ERAFTcall enc 96->128 s2 b4", 96, 128, 3, 3, 2, 4, 120, 160),
    ("ERAFTcall enc 96->128 s2 b8", 96, 128, 3, 3, 2, 8, 120, 160),
    ("ERAFT gru 1x5 384->128", 384, 128, 1, 5, 1, 4, 60, 80),
    ("ERAFT gru 5x1 384->128", 384, 128, 5, 1, 1, 4, 60, 80),
    ("ERAFT convc2 256->192", 256, 192, 3, 3, 1, 4, 60, 80),
    ("ERAFT conv 256->126", 256, 126, 3, 3, 1, 4, 60, 80),
    ("ERAFT head 128->256", 128, 256, 3, 3, 1, 4, 60, 80),
    ("ERAFT enc 64->64 240x320 b12", 64, 64, 3, 3, 1, 12, 240, 320),
    ("ERAFT enc 64->96 s2 b12", 64, 96, 3, 3, 2, 12, 240, 320),
    ("ERAFT enc 96->96 120x160 b12", 96, 96, 3, 3, 1, 12, 120, 160),
    ("ERAFT enc 96->128 s2 b12", 96, 128, 3, 3, 2, 12, 120, 160),
    ("ERAFT enc 128->128 60x80 b12", 128, 128, 3, 3, 1, 12, 60, 80),
]


def run(shape, reps):
    name, cin, cout, kh, kw, stride, n, hin, win = shape
    ph, pw = kh // 2, kw // 2
    hout, wout = (hin + 2 * ph - kh) // stride + 1, (win + 2 * pw - kw) // stride + 1
    x = torch.randn(n, cin, hin, win, device=DEV)
    dy = torch.randn(n, cout, hout, wout, device=DEV)
    dw = torch.zeros(cout, cin, kh, kw, device=DEV)
    db = torch.zeros(cout, device=DEV)
    L = _lib.lib()
    st = _lib.current_stream_ptr(DEV)

    def call():
        _lib.check(L.eemop_conv2d_bwd_weight(x.data_ptr(), dy.data_ptr(), n, hin, win, cin, 0, cin, cout, kh, kw, stride, ph, pw, dw.data_ptr(),
                                             db.data_ptr(), st))
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    gflop = 2.0 * n * hout * wout * cout * cin * kh * kw / 1e9
    return us, gflop / us * 1e3            # TFLOP/s


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    only = sys.argv[2] if len(sys.argv) > 2 else ""
    print(f"{'shape':34s} {'ring us':>9s} {'TFLOP/s':>8s} {'tile-bx3 us':>11s} {'TFLOP/s':>8s} {'tile-fp32 us':>12s} {'TFLOP/s':>8s}")
    for sh in SHAPES:
        if only and only not in sh[0]:
            continue
        os.environ.pop("EEM_NO_WGRAD_RING", None)
        os.environ.pop("EEM_NO_WGRAD_BX3", None)
        os.environ["EEM_WGRAD_RING"] = "all"
        ur, _ = run(sh, reps)
        os.environ.pop("EEM_WGRAD_RING", None)
        os.environ["EEM_NO_WGRAD_RING"] = "1"
        ub, _ = run(sh, reps)
        os.environ["EEM_NO_WGRAD_BX3"] = "1"
        uo, _ = run(sh, reps)
        name, cin, cout, kh, kw, stride, n, hin, win = sh
        hout, wout = (hin + 2 * (kh // 2) - kh) // stride + 1, (win + 2 * (kw // 2) - kw) // stride + 1
        gflop = 2.0 * n * hout * wout * cout * cin * kh * kw / 1e9
        print(f"{name:34s} {ur:9.1f} {gflop / ur * 1e3:8.1f} {ub:11.1f} {gflop / ub * 1e3:8.1f} {uo:12.1f} {gflop / uo * 1e3:8.1f}", flush=True)


if __name__ == "__main__":
    main()
