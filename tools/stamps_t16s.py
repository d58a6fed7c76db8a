"""Stage times of k_eval_t16s (one board per workgroup) from a BF_STAMPS build:
    tools/build_variant.sh s_stamps eval_t16s -DBF_STAMPS
    BATCHFIRST_LIB=build/variants/libbf_s_stamps.so python3 tools/stamps_t16s.py [--n 512]
Workgroup 0's LAST board (with n = 2 x the compute units: its second, the steady state), s_memtime ticks
between the stage boundaries, per wave, and the kernel's wall time for the scale."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    args = ap.parse_args()
    import torch
    from batch_first_amd import _lib, ops, positions as P, weights as W
    lib = _lib.load()
    lib.bf_set_split_boards(1 << 20)
    ctx = ops.DeviceContext(W.make_weights(seed=1234, bn="perturbed"), device=0, dtype="bf16")
    recs = ops.records_to_tensor(P.to_compact(P.test_set(1024, seed=0)), torch.device("cuda", 0))[:args.n].contiguous()
    mb = torch.empty(args.n, 32, dtype=torch.uint8, device="cuda")
    _lib.check(lib.bf_encode_mailbox_structs(ctx._ctx, recs.data_ptr(), 88, args.n, mb.data_ptr(), None))
    out = torch.zeros(4096, dtype=torch.float32, device="cuda")
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        _lib.check(lib.bf_eval_mailbox(ctx._ctx, mb.data_ptr(), args.n, out.data_ptr(), None))
    torch.cuda.synchronize()
    t0.record()
    for _ in range(100):
        _lib.check(lib.bf_eval_mailbox(ctx._ctx, mb.data_ptr(), args.n, out.data_ptr(), None))
    t1.record()
    torch.cuda.synchronize()
    print("n = %d: %.2f us per launch back to back" % (args.n, t0.elapsed_time(t1) * 10))
    st = out.cpu().numpy().view(np.uint32)[1024:1024 + 64].reshape(4, 16).astype(np.int64)
    names = ["image + barrier", "stage 1", "stage 2 (1x1 pair)", "3x3: 128->32", "3x3: five 32->32", "padding groups", "head"]
    for w in range(4):
        d = (st[w, 1:8] - st[w, 0:7]) & 0xFFFFFFFF
        total = (st[w, 7] - st[w, 0]) & 0xFFFFFFFF
        whole = (st[w, 9] - st[w, 8]) & 0xFFFFFFFF
        print("wave %d: board %6d ticks, kernel %6d ticks | " % (w, total, whole) + ", ".join("%s %d" % (nm, x) for nm, x in zip(names, d)))


if __name__ == "__main__":
    main()
