"""BASELINE configs[2]: batch sweep 256 .. 262144 on one MI355X, latency versus throughput.

    python tools/sweep.py [--dtypes bf16,f32] [--out profiles/r1/sweep.json]

For every batch size N and precision:
  * device: record ingest + evaluation kernel on boards resident in HBM (torch ops, events on the
    current stream; median of `--reps` launches after warm-up) -> microseconds per call, boards/s;
  * callable: the reference's host callable, NumPy wire format in -> NumPy scores out
    (board_predictor(piece_filters, occupied_bbs), chestimator.py:38), wall clock, includes
    the PCIe copies both ways and the synchronisation -> microseconds per call, boards/s.
Prints one JSON object; the knee is the smallest N whose device boards/s reaches 90 % of the
largest N's.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtypes", default="bf16,f32,f32s,f32s3")
    ap.add_argument("--reps", type=int, default=15)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()

    import torch
    from batch_first_amd import chestimator, encoder, ops, positions, weights

    dev = torch.device("cuda", 0)
    w = weights.make_weights(seed=1234, bn="perturbed")
    sizes = [256 << i for i in range(11)]
    structs_all = positions.to_compact(positions.random_structs(sizes[-1], seed=7))
    empty = structs_all[:0]
    result = {"config": "BASELINE configs[2]: batch sweep on one MI355X", "sizes": sizes, "reps": args.reps, "runs": {}}
    for dtype in args.dtypes.split(","):
        ctx = ops.DeviceContext(w, device=0, dtype=dtype)
        bp, _, closer = chestimator.get_inference_functions(w, device=0, dtype=dtype)
        rows = []
        for n in sizes:
            structs = structs_all[:n]
            recs = ops.records_to_tensor(structs, dev)
            for _ in range(3):
                torch.ops.batchfirst.eval_structs(ctx.handle, recs)
            torch.cuda.synchronize()
            times = []
            for _ in range(args.reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                torch.ops.batchfirst.eval_structs(ctx.handle, recs)
                e1.record()
                torch.cuda.synchronize()
                times.append(e0.elapsed_time(e1) * 1e3)
            dev_us = float(np.median(times))
            nib, occ = encoder.struct_array_to_ann_inputs(structs, empty, np.ones(n, bool), np.zeros(0, bool), n)
            for _ in range(2):
                bp(nib, occ)
            reps_host = max(3, min(args.reps, 200000 // n + 3))
            t0 = time.perf_counter()
            for _ in range(reps_host):
                bp(nib, occ)
            host_us = (time.perf_counter() - t0) / reps_host * 1e6
            rows.append({"n": n, "device_us": dev_us, "device_boards_per_s": n / dev_us * 1e6,
                         "callable_us": host_us, "callable_boards_per_s": n / host_us * 1e6})
            print("%-5s N=%7d  device %9.1f us %11.3e boards/s   callable %9.1f us %11.3e boards/s"
                  % (dtype, n, dev_us, n / dev_us * 1e6, host_us, n / host_us * 1e6), flush=True)
        top = rows[-1]["device_boards_per_s"]
        knee = next(r["n"] for r in rows if r["device_boards_per_s"] >= 0.9 * top)
        result["runs"][dtype] = {"rows": rows, "knee_n_90pct": knee}
        closer()
        ctx.close()
    text = json.dumps(result, indent=1)
    if args.out:
        with open(os.path.join(ROOT, args.out), "w") as fh:
            fh.write(text)
    print(json.dumps({k: v["knee_n_90pct"] for k, v in result["runs"].items()}))


if __name__ == "__main__":
    main()
