"""Where the bf16 variant's error comes from, rounding by rounding (CPU only; VERDICT round 2, "bf16 error budget").

The bf16 arithmetic (oracle/network.py:forward_bf16_model, no reference counterpart) rounds 19 things: the folded
kernels of the nine input convolutions (one group), of the nine tower layers, and the activations x1..x9.  This
script evaluates the 3 734 sibling positions of tests/golden/siblings_128.npz with a model in which every one of
those roundings can be switched off or split into two bf16 terms (hi + lo, i.e. 16 significant bits), in float64
sums, and reports for each variant what the search consumes: worst |score - fp32| / max(|fp32|, rms), flipped
sibling pairs, moved arg-maxima.

    python tools/bf16_error_budget.py            # the table
    python tools/bf16_error_budget.py --plans    # + candidate mixed-precision plans with their MFMA cost

MFMA cost model (cycles per board on v_mfma_f32_16x16x32_bf16, bf_layout_t16.h): stage 1 4 544, 236->512 16 384,
512->128 8 192, 128->32 4 608, each 32->32 1 152; a two-term operand on one side doubles a layer's MFMAs, on both
sides triples them (hi*hi + hi*lo + lo*hi).
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from batch_first_amd import weights as W  # noqa: E402
from oracle import encoding as E, network as N  # noqa: E402

LAYERS = ["in"] + ["L%d" % i for i in range(2, 11)]          # kernels: input convs, x2 .. x10
ACTS = ["x%d" % i for i in range(1, 10)]                     # activations rounded after their ReLU
MFMA = {"in": 4544, "L2": 16384, "L3": 8192, "L4": 4608, "L5": 1152, "L6": 1152, "L7": 1152, "L8": 1152, "L9": 1152, "L10": 0}


def rnd(a, mode):
    """mode 1: one bf16 term; 2: two terms (hi + lo); 0: float32 as is."""
    a = np.asarray(a, np.float32)
    if mode == 0:
        return a
    hi = N.bf16_round(a)
    if mode == 1:
        return hi
    return hi + N.bf16_round(a - hi)


def forward(planes, weights, wmode, amode):
    """wmode[layer], amode[act] in {0, 1, 2}; float64 sums."""
    import torch
    import torch.nn.functional as F
    kernels, shift = N.fold_batch_norm(weights)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(torch.float64)

    def conv(x, w_hwio, mode, dil=(1, 1), padding="same"):
        w = t(np.transpose(rnd(w_hwio, mode), (3, 2, 0, 1)))
        k = w_hwio.shape[0]
        if padding == "same":
            pt, pb = N._same_pads(k, dil[0])
            pl, pr = N._same_pads(k, dil[1])
            x = F.pad(x, (pl, pr, pt, pb))
        return F.conv2d(x, w, dilation=dil)

    def act(x, sh, mode):
        y = torch.relu(x + t(sh).view(1, -1, 1, 1))
        return t(rnd(y.to(torch.float32).numpy(), mode)) if mode else y

    with torch.no_grad():
        x0 = t(planes)
        x = act(torch.cat([conv(x0, kernels[i], wmode["in"], dil) for i, (_, dil) in enumerate(N.INPUT_CONV_GEOMETRY)], dim=1),
                shift[0], amode["x1"])
        li = 9
        for section in N.CONVOLUTIONAL_MODULES[0][0]:
            padding = "valid" if isinstance(section, tuple) else "same"
            name = "x%d" % (li - 7)
            x = act(conv(x, kernels[li], wmode["L%d" % (li - 7)], (1, 1), padding), shift[li - 8], amode.get(name, 0))
            li += 1
        return F.conv2d(x, t(np.transpose(kernels[18], (3, 2, 0, 1)))).reshape(-1).numpy()


def report(got, ref, sizes):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    rms = float(np.sqrt(np.mean(ref ** 2)))
    err = np.abs(got - ref) / np.maximum(np.abs(ref), rms)
    pos, flipped, pairs, moved = 0, 0, 0, 0
    for k in sizes:
        k = int(k)
        r, s = ref[pos:pos + k], got[pos:pos + k]
        pos += k
        iu = np.triu_indices(k, 1)
        dr, ds = r[:, None] - r[None, :], s[:, None] - s[None, :]
        flipped += int(np.sum(np.sign(dr[iu]) != np.sign(ds[iu])))
        pairs += len(iu[0])
        moved += int(np.argmax(s) != np.argmax(r))
    return float(err.max()), float(np.sqrt(np.mean(err ** 2))), flipped, pairs, moved


def cost(wmode, amode):
    """MFMA cycles per board: a layer's input activation and its kernels each count their terms."""
    src = {"in": None, "L2": "x1", "L3": "x2", "L4": "x3", "L5": "x4", "L6": "x5", "L7": "x6", "L8": "x7", "L9": "x8", "L10": "x9"}
    total = 0
    for l in LAYERS:
        wt = max(1, wmode[l])
        at = max(1, amode[src[l]]) if src[l] else 1
        products = {(1, 1): 1, (1, 2): 2, (2, 1): 2, (2, 2): 3}[(wt, at)]
        total += MFMA[l] * products
    return total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--plans", action="store_true")
    args = ap.parse_args()
    z = np.load(os.path.join(ROOT, "tests", "golden", "siblings_128.npz"))
    w = W.make_weights(seed=1234, bn="perturbed")
    planes = E.decode_planes_fast(z["piece_filters"], z["occupied_bbs"])
    ref, sizes = z["scores_f32"], z["group_sizes"]
    base_cost = cost({l: 1 for l in LAYERS}, {a: 1 for a in ACTS})

    def run(label, wmode, amode):
        e, rmse, fl, pairs, mv = report(forward(planes, w, wmode, amode), ref, sizes)
        c = cost(wmode, amode)
        print("%-46s worst %.2e  rms %.2e  flipped %4d / %d (%.2f %%)  arg-max moved %2d / 128  MFMA x%.3f"
              % (label, e, rmse, fl, pairs, 100.0 * fl / pairs, mv, c / base_cost), flush=True)
        return e

    all1w, all1a = {l: 1 for l in LAYERS}, {a: 1 for a in ACTS}
    none_w, none_a = {l: 0 for l in LAYERS}, {a: 0 for a in ACTS}
    run("float64 sums, nothing rounded (sanity)", none_w, none_a)
    run("product arithmetic: all 19 roundings", all1w, all1a)
    print("-- ONE rounding alone (everything else float32):")
    for l in LAYERS:
        run("only kernels %s" % l, dict(none_w, **{l: 1}), none_a)
    for a in ACTS:
        run("only activation %s" % a, none_w, dict(none_a, **{a: 1}))
    print("-- all roundings but ONE (that one float32):")
    for l in LAYERS:
        run("all but kernels %s" % l, dict(all1w, **{l: 0}), all1a)
    for a in ACTS:
        run("all but activation %s" % a, all1w, dict(all1a, **{a: 0}))
    if args.plans:
        print("-- plans (2 = two bf16 terms):")
        tail = ["L5", "L6", "L7", "L8", "L9"]
        run("two-term kernels + activations, 32->32 layers", dict(all1w, **{l: 2 for l in tail}),
            dict(all1a, **{a: 2 for a in ("x4", "x5", "x6", "x7", "x8")}))
        run("two-term x3..x8 and kernels L4..L9", dict(all1w, **{l: 2 for l in ["L4"] + tail}),
            dict(all1a, **{a: 2 for a in ("x3", "x4", "x5", "x6", "x7", "x8")}))
        run("two-term x9 only (head is float32 already)", all1w, dict(all1a, x9=2))
        run("two-term kernels everywhere", {l: 2 for l in LAYERS}, all1a)
        run("two-term activations everywhere", all1w, {a: 2 for a in ACTS})
        run("two-term x1 + kernels L2", dict(all1w, L2=2), dict(all1a, x1=2))
        run("two-term x2 + kernels L3", dict(all1w, L3=2), dict(all1a, x2=2))
        run("two-term x1, x2", all1w, dict(all1a, x1=2, x2=2))
        run("two-term kernels L2, L3", dict(all1w, L2=2, L3=2), all1a)
        run("two-term everything (= 16-bit operands)", {l: 2 for l in LAYERS}, {a: 2 for a in ACTS})


if __name__ == "__main__":
    main()
