"""Generates tests/golden/legal_1024.npz (run from the repo root, after building
libbatchfirst_search.so): BASELINE configs[0] -- 1024 random LEGAL positions.

Positions: random legal playouts (uniform over legal moves, the way the reference's own tests
draw positions with python-chess, code_testing.py:336-337) from the start position and the
reference's other six test FENs, by the native move generator that the reference's perft tables
pin (tests/test_search.py).  Expected scores: the oracle's float32 and float64 runs of the value
network, weights seed 1234 with perturbed batch-norm.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from batch_first_amd import positions, search, weights  # noqa: E402
from oracle import encoding, network  # noqa: E402
from tests_fens import FENS  # noqa: E402


def main():
    recs = []
    seed = 0
    while sum(len(r) for r in recs) < 1024:
        fen = FENS[0] if seed % 2 == 0 else FENS[1 + (seed // 2) % 6]
        recs.append(search.random_playout_records(fen, 120, seed=1000 + seed))
        seed += 1
    records = np.concatenate(recs)[:1024]
    structs = np.frombuffer(records.tobytes(), dtype=positions.COMPACT_DTYPE)
    nib, occ = encoding.struct_array_to_ann_inputs(structs)
    planes = encoding.parse_into_ann_input_inference(nib, occ)
    w = weights.make_weights(seed=1234, bn="perturbed")
    out = {"records": records, "piece_filters": nib, "occupied_bbs": occ,
           "scores_f32": network.forward_torch(planes, w), "scores_f64": network.forward_numpy64(planes, w),
           "weights_sha256": np.frombuffer(weights.blob_sha256(weights.to_blob(w)).encode(), dtype=np.uint8)}
    path = os.path.join(ROOT, "tests", "golden", "legal_1024.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", seed, "playouts; black to move in", int((structs["turn"] == 0).sum()),
          "; with ep square", int((structs["ep_square"] != 0).sum()), "; with castling rights", int((structs["castling_rights"] != 0).sum()))


if __name__ == "__main__":
    main()
