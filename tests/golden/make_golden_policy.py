"""Generates tests/golden/policy_256.npz with the oracle (run from the repo root).

Move-scoring network (batch_first/anns/move_evaluation_ann.py) on the first 256 positions of
positions_1024.npz, weights seed 4321 (perturbed batch-norm), 0..40 seeded random requests per
board over all 64 from-squares and 73 filters (the network does not care whether a request is a
legal move).  Expected scores from the float32 torch restatement and a float64 run of the same.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from batch_first_amd import weights  # noqa: E402
from oracle import encoding, network  # noqa: E402


def main():
    with np.load(os.path.join(ROOT, "tests", "golden", "positions_1024.npz")) as z:
        planes = encoding.decode_planes_fast(z["piece_filters"], z["occupied_bbs"])[:256]
    rng = np.random.Generator(np.random.PCG64(77))
    per = rng.integers(0, 41, 256).astype(np.uint8)
    per[5] = 0
    per[17] = 255 % 41
    m = int(per.sum())
    frm = rng.integers(0, 64, m).astype(np.uint8)
    flt = rng.integers(0, 73, m).astype(np.uint8)
    w = weights.make_policy_weights(seed=4321, bn="perturbed")
    l32, x9 = network.policy_logits_torch(planes, w)
    l64, _ = network.policy_logits_torch(planes, w, double=True)
    out = {
        "moves_per_board": per, "from_squares": frm, "move_filters": flt,
        "scores_f32": network.policy_gather(l32, frm, flt, per).astype(np.float32),
        "scores_f64": network.policy_gather(l64, frm, flt, per).astype(np.float64),
        "weights_sha256": np.frombuffer(weights.blob_sha256(weights.to_policy_blob(w)).encode(), dtype=np.uint8),
    }
    path = os.path.join(ROOT, "tests", "golden", "policy_256.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes", m, "moves")


if __name__ == "__main__":
    main()
