#!/usr/bin/env python3
"""Developer tool: fit / evaluate the cold-handle cost predictor on gpurun_out/pred/*.npz (scripts/dev/predictor_data.py).
Features come from the packed record alone; quality = makespan of longest-predicted-first list scheduling on the chip's
workgroup slots under a cost model base + c * iterations, against natural order and against the true iteration counts."""
import glob
import heapq
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def fields(rec, h, nc):
    nf = 73 if nc == 3 else 54
    f = rec[:, : 4 * (nf + 12 * h)].copy().view(np.float32)
    o = dict(p=f[:, 0:3], v=f[:, 3:6], q=f[:, 6:10], w=f[:, 10:13], r=f[:, 13:13 + 3 * nc], traj=f[:, nf:nf + 12 * h].reshape(-1, h, 12))
    g = rec[:, 4 * (nf + 12 * h): 4 * (nf + 12 * h) + nc * h].reshape(-1, h, nc)
    o["gait"] = g
    return o


def features(rec, h, nc):
    o = fields(rec, h, nc)
    qw, qx, qy, qz = o["q"].T.astype(np.float64)
    roll = np.arctan2(2 * (qw * qx + qy * qz), 1 - 2 * (qx * qx + qy * qy))
    pitch = np.arcsin(np.clip(2 * (qw * qy - qx * qz), -1, 1))
    v, w, tr = o["v"].astype(np.float64), o["w"].astype(np.float64), o["traj"].astype(np.float64)
    dvx = v[:, 0] - tr[:, 0, 9]
    dvy = v[:, 1] - tr[:, 0, 10]
    stance = o["gait"].reshape(len(rec), -1).sum(axis=1).astype(np.float64)
    F = dict(dvx=dvx, dvy=dvy, vz=v[:, 2], roll=roll, pitch=pitch, wx=w[:, 0], wy=w[:, 1], wz=w[:, 2], stance=stance)
    return F


def lpt_makespan(cost, order, slots):
    heap = [0.0] * slots
    heapq.heapify(heap)
    for i in order:
        t = heapq.heappop(heap)
        heapq.heappush(heap, t + cost[i])
    return max(heap)


if __name__ == "__main__":
    for path in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "pred", "*.npz"))):
        d = np.load(path)
        h, nc, it = int(d["h"]), int(d["nc"]), d["iters"].astype(np.float64)
        F = features(d["rec"], h, nc)
        base, c = (400.0, 20.0) if nc == 3 else (200.0, 7.0)   # k-cycles: ~ kernel cycles without iterations, per iteration
        cost = base + c * it
        slots = 512 if nc == 3 else 768
        nat = lpt_makespan(cost, range(len(it)), slots)
        best = lpt_makespan(cost, np.argsort(-it, kind="stable"), slots)
        line = f"{os.path.basename(path):28s} natural {nat:8.0f}  exact-order {best:8.0f} ({nat / best:.3f}x)"
        for name, score in (("|dvx|", np.abs(F["dvx"])),
                            ("|dvx|+.5|dvy|", np.abs(F["dvx"]) + 0.5 * np.abs(F["dvy"])),
                            ("stance*(.1+|dvx|)", F["stance"] * (0.1 + np.abs(F["dvx"])))):
            ms = lpt_makespan(cost, np.argsort(-score, kind="stable"), slots)
            line += f"  {name} {ms:8.0f} ({nat / ms:.3f}x, corr {np.corrcoef(score, it)[0, 1]:.2f})"
        print(line)


def design(F):
    a = lambda k: np.abs(F[k])
    cols = [a("dvx"), a("dvy"), a("vz"), a("roll"), a("pitch"), a("wx"), a("wy"), a("wz"),
            a("dvx") ** 2, a("dvy") ** 2, a("roll") ** 2, a("pitch") ** 2, a("wx") ** 2, a("wy") ** 2,
            a("dvx") * a("pitch"), a("dvx") * a("wy"), a("dvy") * a("roll"), a("dvy") * a("wx")]
    return np.stack(cols, axis=1)


def fit_and_report():
    P = lambda n: os.path.join(ROOT, "gpurun_out", "pred", n + ".npz")
    groups = {"2c": (["standing_h10_s6", "standing_h10_x3", "single_h20_s2", "standing_h20_s2", "mixed_h10_s3"],
                     ["standing_h10_s1006", "standing_h10_s2", "standing_h10_s2_b1024", "standing_h10_s6_next", "walking_h10_s6",
                      "standing_h10_x3", "standing_h20_s2", "single_h20_s2"]),
              "3c": (["3contact_s5"], ["3contact_s6", "3contact_s5_b2048"])}
    for gname, (train, test) in groups.items():
        X, y = [], []
        for n in train:
            d = np.load(P(n))
            F = features(d["rec"], int(d["h"]), int(d["nc"]))
            A = design(F) * (F["stance"][:, None] / 20.0)   # cost scales with the number of stance leg-steps
            X.append(A)
            y.append(d["iters"].astype(np.float64))
        X, y = np.concatenate(X), np.concatenate(y)
        coef, *_ = np.linalg.lstsq(np.c_[X, np.ones(len(X))], y, rcond=None)
        print(gname, "coef", np.round(coef, 2))
        for n in test:
            d = np.load(P(n))
            h, nc, it = int(d["h"]), int(d["nc"]), d["iters"].astype(np.float64)
            F = features(d["rec"], h, nc)
            score = (design(F) * (F["stance"][:, None] / 20.0)) @ coef[:-1]
            base, c = (400.0, 20.0) if nc == 3 else (200.0, 7.0)
            cost = base + c * it
            slots = 512 if nc == 3 else 768
            nat = lpt_makespan(cost, range(len(it)), slots)
            best = lpt_makespan(cost, np.argsort(-it, kind="stable"), slots)
            ms = lpt_makespan(cost, np.argsort(-score, kind="stable"), slots)
            print(f"  {n:26s} natural {nat:7.0f} exact {nat / best:.3f}x  fitted {nat / ms:.3f}x  corr {np.corrcoef(score, it)[0, 1]:.2f}")


if __name__ == "__main__":
    fit_and_report()


def shipped_score(rec, h, nc):
    """hmpc_builder.h predicted_cost_bucket, restated (the bucket is this score x 48, clamped to 0..63)."""
    nf = 73 if nc == 3 else 54
    f = rec[:, : 4 * (nf + 12 * h)].copy().view(np.float32)
    vx, qw, qx, qy, qz = f[:, 3], f[:, 6], f[:, 7], f[:, 8], f[:, 9]
    u = (f[:, nf + 9] - vx) + 2.0 * 0.5 * (f[:, 13] + f[:, 14])
    sr, sp = 2.0 * (qw * qx + qy * qz), 2.0 * (qw * qy - qx * qz)
    return np.where(u > 0, u, -0.05 * u) + 0.5 * (np.abs(sr) + np.abs(sp))


def report_shipped():
    print("== the shipped predictor: score = hinge((vx_cmd - vx) + 2 mean foot x; slope 1 / 0.05) + 0.5 (|sin roll| + |sin pitch|)")
    for path in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "pred", "*.npz"))):
        d = np.load(path)
        h, nc, it = int(d["h"]), int(d["nc"]), d["iters"].astype(np.float64)
        score = shipped_score(d["rec"], h, nc)
        base, c = (400.0, 20.0) if nc == 3 else (200.0, 7.0)
        cost = base + c * it
        slots = 512 if nc == 3 else 768
        nat = lpt_makespan(cost, range(len(it)), slots)
        best = lpt_makespan(cost, np.argsort(-it, kind="stable"), slots)
        ms = lpt_makespan(cost, np.argsort(-np.clip((score * 48).astype(int), 0, 63), kind="stable"), slots)
        print(f"  {os.path.basename(path):28s} corr(score, iterations) {np.corrcoef(score, it)[0, 1]:.2f}   simulated makespan: natural / "
              f"by true iterations / by the predictor's buckets = 1 / {best / nat:.3f} / {ms / nat:.3f}")


if __name__ == "__main__":
    report_shipped()
