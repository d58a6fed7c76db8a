"""Generates tests/golden/siblings_128.npz (run from the repo root, after building
libbatchfirst_search.so): the ordering test set of SURVEY.md 8(d) config 3 -- 128 sibling groups.

A sibling group is what one node's expansion gives the search to rank: the children of one parent
(create_child_structs, numba_negamax_zero_window.py:315-347).  Parents: the first 128 positions of
legal_1024.npz (random legal playouts) that have 20 to 40 legal moves; children: EVERY legal move of
the parent, by the native move generator and make-move that the reference's perft tables pin
(bfs_legal_moves_record, bfs_push_record).  Expected value-network scores of the children: the oracle's
float32 and float64 runs (weights seed 1234, perturbed batch-norm).  For the move-scoring network the
same groups are the parents' legal moves as (from_square, move_filter) requests in the mover's frame
(numba_negamax_zero_window.py:485-527), scored by the oracle with policy weights seed 4321.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from batch_first_amd import moves, positions, search, weights  # noqa: E402
from oracle import encoding, network  # noqa: E402


def main():
    with np.load(os.path.join(ROOT, "tests", "golden", "legal_1024.npz")) as z:
        pool = z["records"]
    parents, groups, children, all_moves = [], [], [], []
    for i, rec in enumerate(pool):
        mv = search.legal_moves_of_record(rec)
        if not 20 <= len(mv) <= 40:
            continue
        parents.append(i)
        groups.append(len(mv))
        all_moves.append(mv)
        children.extend(search.push_record(rec, m) for m in mv)
        if len(parents) == 128:
            break
    assert len(parents) == 128
    children = np.stack(children)
    group_sizes = np.array(groups, np.uint8)
    all_moves = np.concatenate(all_moves)
    structs = np.frombuffer(children.tobytes(), dtype=positions.COMPACT_DTYPE)
    nib, occ = encoding.struct_array_to_ann_inputs(structs)
    planes = encoding.decode_planes_fast(nib, occ)
    w = weights.make_weights(seed=1234, bn="perturbed")
    # the move-scoring network on the parents
    pstructs = np.frombuffer(pool[parents].tobytes(), dtype=positions.COMPACT_DTYPE)
    pnib, pocc = encoding.struct_array_to_ann_inputs(pstructs)
    frm, flt, per = moves.move_info_for(pstructs["turn"], group_sizes, all_moves)
    pw = weights.make_policy_weights(seed=4321, bn="perturbed")
    pplanes = encoding.decode_planes_fast(pnib, pocc)
    l32, _ = network.policy_logits_torch(pplanes, pw)
    l64, _ = network.policy_logits_torch(pplanes, pw, double=True)
    out = {
        "parent_index": np.array(parents, np.int32), "group_sizes": group_sizes, "moves": all_moves,
        "records": children, "piece_filters": nib, "occupied_bbs": occ,
        "scores_f32": network.forward_torch(planes, w), "scores_f64": network.forward_numpy64(planes, w),
        "parent_records": pool[parents], "from_squares": frm, "move_filters": flt,
        "policy_scores_f32": network.policy_gather(l32, frm, flt, per).astype(np.float32),
        "policy_scores_f64": network.policy_gather(l64, frm, flt, per).astype(np.float64),
        "weights_sha256": np.frombuffer(weights.blob_sha256(weights.to_blob(w)).encode(), dtype=np.uint8),
        "policy_weights_sha256": np.frombuffer(weights.blob_sha256(weights.to_policy_blob(pw)).encode(), dtype=np.uint8),
    }
    path = os.path.join(ROOT, "tests", "golden", "siblings_128.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", len(children), "children in 128 groups of", group_sizes.min(), "to",
          group_sizes.max())


if __name__ == "__main__":
    main()
