"""Round 6, a measurement for what could come after this repository (CPU only, the oracle's volumes): how much of the coarse lattice would
an EXACT branch-and-bound have to score?  Karto's results need the exact response only where it is within 0.1 of the best (ties for the
mean, the positional covariance's threshold; open_karto ComputePositionalCovariance).  Bound of a block of B x B hypotheses at one angle:
sum over the beams of the maximum grid byte in the 8 x 8-cell tiles the block's reads can fall into (a 1/8-resolution image of tile maxima,
2 x 2 of them for B = 4) -- an upper bound of every response in the block, penalties aside (they only lower it).  Lower bound of the best:
the exact response at the centre hypothesis of the four blocks with the highest bound.  Survivors: blocks whose bound reaches that - 0.1.
Prints, per problem, the hypotheses within 0.1 of the best, the surviving blocks and the angles that hold them."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as orc  # noqa: E402
from tests.util import PlainScan, cfg2_scans  # noqa: E402
from yag_slam_amd import synth  # noqa: E402


def sliding_max(Q, n):
    """out[t] = max(Q[t : t + n]) along both axes (zero beyond the edge)"""
    P = np.pad(Q, ((0, n), (0, n)))
    out = np.zeros_like(Q)
    for dy in range(n):
        for dx in range(n):
            out = np.maximum(out, P[dy:dy + Q.shape[0], dx:dx + Q.shape[1]])
    return out


def analyse(q, base, label):
    o = orc.Oracle(None, "karto")
    o.match_scan(q, base, True, False)
    R = o.responses(0)
    nt, ny, nx = R.shape
    G, info = o.grid_u8()
    G = G.astype(np.int64)
    ql = o.query_local()
    N = len(ql)
    best = R.max()
    res, scale = 0.01, 100.0
    pose = (q.corrected_pose.x, q.corrected_pose.y, q.corrected_pose.euler[-1])
    border, roi_w = info["roi"][0], info["roi"][2]
    offx, offy = pose[0] - 0.5 * (roi_w - 1) * res, pose[1] - 0.5 * (roi_w - 1) * res
    rnd = lambda v: np.where(v >= 0, np.floor(v + 0.5), np.ceil(v - 0.5)).astype(np.int64)
    cx = rnd(((pose[0] + (-0.25 + np.arange(nx) * 0.02)) - offx) * scale) + border
    cy = rnd(((pose[1] + (-0.25 + np.arange(ny) * 0.02)) - offy) * scale) + border
    T = 8
    H, W = G.shape
    Q = G[:H // T * T, :W // T * T].reshape(H // T, T, W // T, T).max(axis=(1, 3))
    line = ["%-22s best %.4f, within 0.1 of it: %3d of %d hypotheses |" % (label, best, int((R >= best - 0.1).sum()), R.size)]
    for B, ntile in ((4, 2), (8, 3)):
        QS = sliding_max(Q, ntile)
        nb = (nx + B - 1) // B
        UB = np.zeros((nt, nb, nb))
        for k in range(nt):
            ang = (pose[2] - 0.349) + k * 0.0349
            c, s = np.cos(ang), np.sin(ang)
            gx = rnd((((c * ql[:, 0] - s * ql[:, 1]) + offx) - offx) * scale)
            gy = rnd((((s * ql[:, 0] + c * ql[:, 1]) + offy) - offy) * scale)
            for by in range(nb):
                for bx in range(nb):
                    UB[k, by, bx] = QS[(cy[by * B] + gy) // T, (cx[bx * B] + gx) // T].sum() / (N * 100.0)
                    assert UB[k, by, bx] + 1e-12 >= R[k, by * B:(by + 1) * B, bx * B:(bx + 1) * B].max()  # it IS a bound
        lb = 0.0
        for f in np.argsort(UB.ravel())[::-1][:4]:
            k, by, bx = np.unravel_index(f, UB.shape)
            lb = max(lb, R[k, min(ny - 1, by * B + B // 2), min(nx - 1, bx * B + B // 2)])
        surv = UB >= lb - 0.1
        line.append("B = %d: %3d of %4d blocks in %2d of %d angles (lb %.3f) |" % (B, int(surv.sum()), surv.size, int(surv.any(axis=(1, 2)).sum()), nt, lb))
    print(" ".join(line), flush=True)


def main():
    q, base = cfg2_scans()
    analyse(q, base, "cfg2")
    analyse(q, base[:2], "cfg2, 2-scan chain")
    truth, prior = synth.loop_trajectory(400)
    scene = synth.Scene()
    mk = lambda rr, p: PlainScan(rr, synth.MIN_ANGLE, synth.ANGLE_INCREMENT, synth.MIN_RANGE, 20.0, p)
    for i in (30, 120, 260, 333):
        chain = [mk(scene.scan_ranges(tuple(truth[j]), index=j), tuple(truth[j])) for j in range(i - 10, i)]
        analyse(mk(scene.scan_ranges(tuple(truth[i]), index=i), tuple(prior[i])), chain, "trajectory step %d" % i)


if __name__ == "__main__":
    main()
