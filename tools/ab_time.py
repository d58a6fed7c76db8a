"""Interleaved A/B timing of k_eval_f32 across library builds, in one process (rule: perf deltas
come from interleaved rounds on one device).

    python tools/ab_time.py [--n 65536] [--rounds 7] name=path/to/lib.so[@VAR=value] ...

Each library gets its own context with the same weights; every round launches each library's
bf_eval_mailbox once, timed with events on torch's current stream.  Prints median/min ms and the
largest score difference against the first library.
"""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=65536)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--burst", type=int, default=1, help="launches back to back per timing (the chip lowers its clock under sustained load)")
    ap.add_argument("--dtype", default="f32", help="f32 or bf16: the precision every library's context is created with")
    ap.add_argument("libs", nargs="+")
    args = ap.parse_args()
    import torch
    from batch_first_amd import _lib, positions, weights
    from oracle import encoding

    blob = weights.to_blob(weights.make_weights(seed=1234, bn="perturbed"))
    if os.environ.get("BF_ZERO_WEIGHTS"):   # diagnostic: data-dependent power (operand toggling) against the same instruction stream
        blob = np.zeros_like(blob)
    structs = positions.random_structs(args.n, seed=5)
    nib, occ = None, None
    # mailbox via the first library's own encoder
    libs = []
    for spec in args.libs:
        name, path = spec.split("=", 1)
        env = {}
        if "@" in path:     # name=lib.so@VAR=value: environment for the context's creation (e.g. BF_T16_TWO_KERNELS=1)
            path, rest = path.split("@", 1)
            env = dict(kv.split("=", 1) for kv in rest.split("@"))
        os.environ.update(env)
        lib = ctypes.CDLL(os.path.join(ROOT, path) if not os.path.isabs(path) else path)
        for fn, (res, argt) in _lib.SIGNATURES.items():
            if hasattr(lib, fn):
                getattr(lib, fn).restype = res
                getattr(lib, fn).argtypes = argt
        ctx = ctypes.c_void_p()
        rc = lib.bf_ctx_create_ex(ctypes.byref(ctx), 0, blob.ctypes.data, blob.size, _lib.precision_code(args.dtype))
        assert rc == 0, lib.bf_last_error()
        for k in env:
            del os.environ[k]
        libs.append((name, lib, ctx))
    recs = torch.from_numpy(np.ascontiguousarray(positions.to_compact(structs)).view(np.uint8).reshape(args.n, -1).copy()).cuda()
    mailbox = torch.empty((args.n, 32), dtype=torch.uint8, device="cuda")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    name0, lib0, ctx0 = libs[0]
    assert lib0.bf_encode_mailbox_structs(ctx0, recs.data_ptr(), recs.shape[1], args.n, mailbox.data_ptr(), stream) == 0
    outs = {name: torch.zeros(args.n, dtype=torch.float32, device="cuda") for name, _, _ in libs}
    times = {name: [] for name, _, _ in libs}
    for name, lib, ctx in libs:  # warm-up
        assert lib.bf_eval_mailbox(ctx, mailbox.data_ptr(), args.n, outs[name].data_ptr(), stream) == 0, lib.bf_last_error()
    torch.cuda.synchronize()
    for _ in range(args.rounds):
        for name, lib, ctx in libs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _b in range(args.burst):
                lib.bf_eval_mailbox(ctx, mailbox.data_ptr(), args.n, outs[name].data_ptr(), stream)
            e1.record()
            torch.cuda.synchronize()
            times[name].append(e0.elapsed_time(e1) / args.burst)
    ref = outs[name0].cpu().numpy()
    for name, _, _ in libs:
        t = np.array(times[name])
        diff = float(np.max(np.abs(outs[name].cpu().numpy() - ref)))
        print("%-16s median %8.3f ms  min %8.3f ms  boards/s %.3e  max|score - %s| %.3e"
              % (name, np.median(t), t.min(), args.n / (np.median(t) * 1e-3), name0, diff), flush=True)
    for name, lib, ctx in libs:
        lib.bf_ctx_destroy(ctx)


if __name__ == "__main__":
    main()
