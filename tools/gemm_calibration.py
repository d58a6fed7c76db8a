#!/usr/bin/env python3
"""What the vendor's GEMM reaches on THIS box, as a yardstick for the fused kernels' fractions of the
dense MFMA peaks (2.5 PFLOP/s bf16, 157.3 TFLOP/s fp32: MI355X_MICROARCH.md).

torch.matmul (hipBLASLt / rocBLAS) on a large square problem and on the shapes of the tower's two 1x1
layers at 65 536 boards (rows = boards x 64 squares; 512 and 128 output channels), timed with events over
back-to-back launches like bench.py times the evaluator.  Not part of the product: nothing here is
loaded by the library.

    python3 tools/gemm_calibration.py --out gpurun_out/gemm_calibration.json
"""
import argparse
import json

import torch

PEAK = {"bf16": 2500.0e12, "f32": 157.3e12}


def time_matmul(m, k, n, dtype, reps):
    a = torch.randn(m, k, device="cuda", dtype=dtype)
    b = torch.randn(k, n, device="cuda", dtype=dtype)
    out = torch.empty(m, n, device="cuda", dtype=dtype)
    for _ in range(3):
        torch.matmul(a, b, out=out)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        torch.matmul(a, b, out=out)
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    torch.backends.cuda.matmul.allow_tf32 = False
    rows = 65536 * 64
    shapes = [
        ("square 8192", 8192, 8192, 8192),
        ("square 16384 x 8192 x 8192", 16384, 8192, 8192),
        ("1x1 layer 236 (240) -> 512 channels, 65 536 boards", rows, 240, 512),
        ("1x1 layer 512 -> 128 channels, 65 536 boards", rows, 512, 128),
    ]
    res = []
    for name, m, k, n in shapes:
        for tag, dt in (("bf16", torch.bfloat16), ("f32", torch.float32)):
            if tag == "f32" and m == rows and k * 4 * m > 12e9:
                continue
            s = time_matmul(m, k, n, dt, args.reps)
            flops = 2.0 * m * k * n
            r = {"shape": name, "m": m, "k": k, "n": n, "dtype": tag, "ms": s * 1e3,
                 "tflops": flops / s / 1e12, "frac_of_dense_peak": flops / s / PEAK[tag],
                 "bytes_min_GBps": (m * k + k * n + m * n) * (2 if tag == "bf16" else 4) / s / 1e9}
            res.append(r)
            print(json.dumps(r), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump({"device": torch.cuda.get_device_name(0), "torch": torch.__version__, "runs": res}, f, indent=1)


if __name__ == "__main__":
    main()
