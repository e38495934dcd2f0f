#!/usr/bin/env python3
"""Offline comparison of launch-budget policies on the per-time-step iteration record of a real run
(scripts/step_iters_dump.py -> gpurun_out/r05/step_iters.npz: 64 maps x 183 time steps of config 2 at the production settings).
  (a) budgets per step CLASS (rounds 2-4) against budgets per TIME STEP: launches budgeted and maps that would overflow;
  (b) heads for the persistent tails: cost per step of the launches that find nothing to do (4.4 us each, 3 per pressure
      iteration) + the tail launch (4.4 us) + the extra cost of an iteration run in the tail (15 us pressure / 5 us velocity),
      for several predictors of the head.
    python3 scripts/step_budget_policies.py [step_iters.npz]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
z = np.load(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r05", "step_iters.npz"))
H, P = z["helm"], z["pres"]
nm, ns = H.shape
print("record: %d maps x %d time steps; mean iterations per step (maps 9..): velocity %.2f, pressure %.2f" % (nm, ns, H[8:].mean(), P[8:].mean()))


def nbr(A, nb, f):
    B = A.copy()
    for d in range(1, nb + 1):
        B[d:] = f(B[d:], A[:-d]); B[:-d] = f(B[:-d], A[d:])
    return B


cls = np.array([0, 1, 2] + [3] * 3 + [4] * 10 + [5] * (ns - 16))
tot_h = tot_p = ov = 0
for m in range(8, nm):
    bh = np.zeros(ns, int); bp = np.zeros(ns, int)
    for k in range(6):
        idx = cls == k
        mh, mp = H[m - 8:m][:, idx].max(), P[m - 8:m][:, idx].max()
        xh = max(4, mh // 2) if k <= 3 else (1 if k == 4 else 0); xp = max(4, mp) if k <= 3 else (1 if k == 4 else 0)
        bh[idx] = mh + 3 + xh; bp[idx] = mp + 3 + xp
    tot_h += bh.sum(); tot_p += bp.sum(); ov += int((H[m] > bh - 1).any() or (P[m] > bp).any())
print("(a) per step class (window 8, +3): velocity launches/step %.2f, pressure iterations/step %.2f, overflowing maps %d/%d" % (tot_h / (nm - 8) / ns, tot_p / (nm - 8) / ns, ov, nm - 8))
for W, nb, hh, hp in ((8, 2, 3, 2), (8, 2, 2, 2), (8, 1, 3, 2), (4, 2, 3, 2), (8, 0, 3, 2)):
    tot_h = tot_p = ov = 0
    for m in range(8, nm):
        bh = nbr(H[m - W:m].max(0), nb, np.maximum) + hh; bp = nbr(P[m - W:m].max(0), nb, np.maximum) + hp
        bh[:6] += 4; bp[:6] += 6
        tot_h += bh.sum(); tot_p += bp.sum(); ov += int((H[m] > bh - 1).any() or (P[m] > bp).any())
    print("    per time step, window %d, neighbours +-%d, head-room %d/%d: %.2f / %.2f, overflowing maps %d/%d" % (W, nb, hh, hp, tot_h / (nm - 8) / ns, tot_p / (nm - 8) / ns, ov, nm - 8))
print("(b) heads for the persistent tails: overhead per time step in us (idle launches + tail launch + tail surcharge)")
for name, A, NO, TL, TX, plus in (("pressure", P, 13.2, 4.4, 15.0, 0), ("velocity", H, 4.8, 4.4, 5.0, 1)):
    rows = []
    for kind in ("max", "min", "median", "last"):
        for W in (1, 4, 8):
            for off in (-2, -1, 0, 1, 2):
                tot = idle = tail = 0.0
                for m in range(8, nm):
                    Aw = A[m - W:m]
                    pred = {"max": Aw.max(0), "min": Aw.min(0), "median": np.median(Aw, 0), "last": Aw[-1]}[kind]
                    hd = np.maximum(np.round(pred + off).astype(int) + plus, 0)
                    u = A[m] + plus
                    no, tx = np.maximum(hd - u, 0), np.maximum(u - hd, 0)
                    tot += (no * NO + TL + tx * TX).sum(); idle += no.sum(); tail += tx.sum()
                n = (nm - 8) * ns
                rows.append((tot / n, kind, W, off, idle / n, tail / n))
    rows.sort()
    for r in rows[:5]:
        print("    %s: %.1f us  head = %s of the last %d maps %+d: %.2f idle iterations and %.2f tail iterations per step" % ((name,) + r))
