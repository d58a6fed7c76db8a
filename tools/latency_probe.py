"""Latency of the host entry points the search calls, per call, at small batches:
    python tools/latency_probe.py [--dtype bf16]
value network: bf_eval_structs_host; move scoring: bf_policy_start_structs_host + bf_policy_finish_host."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--reps", type=int, default=300)
    args = ap.parse_args()
    from batch_first_amd import _lib, chestimator, search as S, weights as W
    from tests_fens import FENS
    bp, _, closer = chestimator.get_inference_functions(W.make_weights(seed=1234, bn="perturbed"), dtype=args.dtype,
                                                        policy_weights=W.make_policy_weights(seed=4321))
    lib = _lib.load()
    ctx = bp._handle()
    recs = np.concatenate([S.random_playout_records(FENS[i % 7], 60, 100 + i) for i in range(80)])
    print("records available", len(recs))
    for n in (1, 16, 64, 146, 256, 384, 512, 768, 1024, 4096):
        r = np.ascontiguousarray(recs[:n])
        scores = np.zeros(n, np.float32)
        done = C.c_size_t(0)
        moves = [S.legal_moves_of_record(x) for x in r]
        per = np.array([len(m) for m in moves], np.uint8)
        total = int(per.sum())
        frm = np.concatenate([m[:, 0] for m in moves]).astype(np.uint8) if total else np.zeros(0, np.uint8)
        filt = np.zeros(total, np.uint8)
        out = np.zeros(total, np.float32)

        def value():
            rc = lib.bf_eval_structs_host(ctx, r.ctypes.data, 88, None, n, scores.ctypes.data, C.byref(done))
            assert rc == 0

        def policy():
            assert lib.bf_policy_start_structs_host(ctx, r.ctypes.data, 88, None, n) == 0
            assert lib.bf_policy_finish_host(ctx, frm.ctypes.data, filt.ctypes.data, per.ctypes.data, n, total, out.ctypes.data) == 0

        for name, fn in (("value", value), ("policy start+finish", policy)):
            for _ in range(20):
                fn()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                fn()
            dt = (time.perf_counter() - t0) / args.reps
            print("n %5d  %-20s %8.1f us per call  (%d moves)" % (n, name, dt * 1e6, total), flush=True)
    closer()


if __name__ == "__main__":
    main()
