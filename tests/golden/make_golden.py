"""Generates tests/golden/positions_1024.npz with the oracle (run from the repo root).

The reference Python cannot run here (no TensorFlow / numba / python-chess), so these vectors
come from the CPU restatement in oracle/, which is itself pinned by the reference's test rule
(tests/test_oracle_encoding.py).  Inputs: BASELINE config 1 -- the reference's 7 test FENs, 3
hand positions and 1014 seeded structurally valid positions; weights seed 1234.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from batch_first_amd import positions, weights  # noqa: E402
from oracle import encoding, network  # noqa: E402


def main():
    structs = positions.test_set(1024, seed=0)
    compact = positions.to_compact(structs)
    nib, occ = encoding.struct_array_to_ann_inputs(structs)
    planes = encoding.parse_into_ann_input_inference(nib, occ)
    mailbox = encoding.planes_to_mailbox(planes)
    out = {
        "records": compact.view(np.uint8).reshape(len(compact), -1),
        "piece_filters": nib,
        "occupied_bbs": occ,
        "mailbox": mailbox,
    }
    for bn in ("perturbed", "fresh"):
        w = weights.make_weights(seed=1234, bn=bn)
        out["weights_sha256_" + bn] = np.frombuffer(weights.blob_sha256(weights.to_blob(w)).encode(), dtype=np.uint8)
        out["scores_f32_" + bn] = network.forward_torch(planes, w)
        out["scores_f64_" + bn] = network.forward_numpy64(planes, w)
    path = os.path.join(ROOT, "tests", "golden", "positions_1024.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
