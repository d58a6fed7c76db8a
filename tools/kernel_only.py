"""One evaluation kernel launched back to back with nothing else in the process (for rocprofv3 --pmc passes whose
per-launch HBM bytes should not carry another kernel's dirty lines: VERDICT round 2, "make the traffic figure
reproducible").

    python tools/kernel_only.py f32|f32s|f32s3|bf16 [launches] [n]

f32: k_eval_f32 on mailboxes made by ONE k_to_mailbox launch before a two-second pause; bf16: k_eval_t16<4>, which
takes the records themselves (the kernel bench.py's timed step launches).  Prints the kernel's source hash, which
tools/pmc_summary.py stores beside the counters (bench.py says whether the quoted file matches the build it runs).
"""
import ctypes
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def source_sha16(dtype):
    f32 = ["eval_f32.hip", "bf_layout.h", "pack.cpp"]
    files = {"f32": f32, "f32s": f32, "f32s3": f32, "bf16": ["eval_t16.hip", "t16_common.h", "bf_layout_t16.h", "pack_t16.cpp"]}[dtype]
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(ROOT, "batch_first_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def main():
    dtype = sys.argv[1]
    launches = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
    import torch
    from batch_first_amd import _lib, positions, weights
    lib = _lib.load()
    blob = weights.to_blob(weights.make_weights(seed=1234, bn="perturbed"))
    ctx = ctypes.c_void_p()
    _lib.check(lib.bf_ctx_create_ex(ctypes.byref(ctx), 0, blob.ctypes.data, blob.size, _lib.precision_code(dtype)))
    recs = torch.from_numpy(np.ascontiguousarray(positions.to_compact(positions.random_structs(n, seed=1000))).view(np.uint8).reshape(n, -1).copy()).cuda()
    scores = torch.zeros(n, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    mb = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    if dtype != "bf16":
        _lib.check(lib.bf_encode_mailbox_structs(ctx, recs.data_ptr(), recs.shape[1], n, mb.data_ptr(), st))
    torch.cuda.synchronize()
    time.sleep(2.0)   # whatever the set-up left dirty in the caches has been written back
    for _ in range(launches):
        if dtype != "bf16":
            _lib.check(lib.bf_eval_mailbox(ctx, mb.data_ptr(), n, scores.data_ptr(), st))
        else:
            _lib.check(lib.bf_eval_structs(ctx, recs.data_ptr(), recs.shape[1], n, scores.data_ptr(), st))
    torch.cuda.synchronize()
    print("kernel_only", dtype, "launches", launches, "n", n, "source_sha16", source_sha16(dtype), "finite", bool(torch.isfinite(scores).all()))
    lib.bf_ctx_destroy(ctx)


if __name__ == "__main__":
    main()
