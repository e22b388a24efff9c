#!/usr/bin/env python3
"""Developer tool (CPU only): binary64 emulation of the kernel's active-set strategy -- block start in rounds, window-row flips,
releases, then single-row Goldfarb-Idnani iterations -- on the oracle's reduced QPs, to COUNT what a change of the strategy is
worth before it is built: rounds, flips, releases inside rounds, single-row additions and drops, peak |W|.

    python scripts/dev/emulate_rounds.py [scale] [count] [gait] [h]

Schemes: (kbmax, max_rounds, min_new): the block start's row capacity, rounds at most, newly violated rows a further round needs.
The product kernel of round 5 is (48, 2, 3) with a 64-row working set."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hector_simulation_amd import records, synthetic  # noqa: E402
from oracle import oracle_py  # noqa: E402

FEAS = 1e-9


class QP:
    def __init__(self, o):
        self.H, self.g, self.A = o["H_red"].astype(float), o["g_red"].astype(float), o["A_red"].astype(float)
        self.H = np.triu(self.H) + np.triu(self.H, 1).T
        self.lb, self.ub = o["lb_red"].astype(float), o["ub_red"].astype(float)
        self.m = self.A.shape[0]
        self.rr = np.arange(self.m) % 8
        self.M = np.linalg.inv(self.H)
        self.xu = -self.M @ self.g
        self.hasl = (self.rr <= 4) | (self.rr == 7)
        self.hasu = self.rr >= 4
        self.sc = np.where((self.rr == 7) & (self.ub > 1.0), 1.0 / np.maximum(self.ub, 1e-30), 1.0)

    def slack(self, x):
        """(scaled slack on the tighter side, side: +1 lower / -1 upper)"""
        s = self.A @ x
        sl = np.where(self.hasl, s - np.where(self.hasl, self.lb, 0.0), np.inf)
        su = np.where(self.hasu, np.where(self.hasu, self.ub, 0.0) - s, np.inf) * self.sc
        side = np.where(sl <= su, 1, -1)
        return np.minimum(sl, su), side


def solve_on(qp, act):
    """x, u for the working set act (signed: +1 lower, -1 upper, 0 out); rows in index order"""
    W = np.flatnonzero(act)
    if W.size == 0:
        return qp.xu.copy(), np.zeros(0), W
    N = act[W, None] * qp.A[W]
    b = np.where(act[W] > 0, qp.lb[W], -qp.ub[W])
    S0 = N @ qp.M @ N.T
    u = np.linalg.solve(S0, b - N @ qp.xu)
    x = qp.xu + qp.M @ N.T @ u
    return x, u, W


def emulate(qp, kbmax, max_rounds, min_new, qcap, batch_release=False):
    cnt = dict(rounds=0, flips=0, rel=0, adds=0, drops=0, peak=0, overflow=0, resolves=0)
    act = np.zeros(qp.m, dtype=int)
    flpc = np.zeros(qp.m, dtype=bool)
    x = qp.xu.copy()
    for r in range(max_rounds):
        sl, side = qp.slack(x if r > 0 else qp.xu)
        viol = (sl < -FEAS) & (qp.rr <= 6) & (act == 0)
        # friction rows: at most one per axis (partner 0<->1, 2<->3 violated or active -> skipped)
        occupied = viol | (act != 0)
        partner = np.arange(qp.m) ^ 1
        fr = qp.rr < 4
        fresh = viol & ~(fr & occupied[np.where(fr, partner, 0)])
        take = (act != 0) | fresh
        k0 = int(take.sum())
        if r > 0 and (int(fresh.sum()) < min_new or k0 > kbmax):
            break
        if k0 > kbmax:  # round 0: the first kbmax candidates
            idx = np.flatnonzero(take)[kbmax:]
            take[idx] = False
        new_act = np.where(take, np.where(act != 0, act, side), 0)
        if not take.any():
            break
        act = new_act
        cnt["rounds"] += 1
        # negative multipliers: window rows switch sides (all at once, each once) when the most negative one is such a row,
        # otherwise the most negative row is released
        while True:
            x, u, W = solve_on(qp, act)
            if W.size == 0 or u.min() >= -1e-12:
                break
            l = int(np.argmin(u))
            c = W[l]
            if qp.rr[c] == 4 and not flpc[c]:
                cand = (u < -1e-12) & (qp.rr[W] == 4) & ~flpc[W]
                act[W[cand]] *= -1
                flpc[W[cand]] = True
                cnt["flips"] += 1
            elif batch_release:
                # every row with a negative multiplier leaves at once (window rows that may still switch excepted); one more solve
                neg = (u < -1e-12) & ~((qp.rr[W] == 4) & ~flpc[W])
                act[W[neg]] = 0
                cnt["resolves"] += 1
                cnt["relb"] = cnt.get("relb", 0) + int(neg.sum())
            else:
                act[c] = 0
                cnt["rel"] += 1
        cnt["peak"] = max(cnt["peak"], int((act != 0).sum()))
        if (act != 0).sum() == 0:
            break
    # single-row dual active set from here
    x, u, W = solve_on(qp, act)
    ufull = np.zeros(qp.m)
    ufull[W] = u
    guard = 0
    while True:
        guard += 1
        if guard > 2000:
            cnt["stuck"] = 1
            break
        sl, side = qp.slack(x)
        sl = np.where(act == 0, sl, np.inf)
        p = int(np.argmin(sl))
        if not sl[p] < -FEAS:
            break
        if (act != 0).sum() >= qcap:
            cnt["overflow"] = 1
            qcap = 10 ** 9  # (keep counting as if handed over)
        sg = side[p]
        npl = sg * qp.A[p]
        sp = (qp.A[p] @ x - qp.lb[p]) if sg > 0 else (qp.ub[p] - qp.A[p] @ x)
        up = 0.0
        while True:
            W = np.flatnonzero(act)
            if W.size:
                N = act[W, None] * qp.A[W]
                E = np.linalg.inv(N @ qp.M @ N.T)
                d = N @ qp.M @ npl
                rvec = E @ d
                z = qp.M @ (npl - N.T @ rvec)
            else:
                rvec = np.zeros(0)
                z = qp.M @ npl
            delta = npl @ z
            gamma = npl @ qp.M @ npl
            pos = rvec > 1e-14
            t1, l = (np.inf, -1)
            if pos.any():
                ratios = np.where(pos, ufull[W] / np.where(pos, rvec, 1.0), np.inf)
                l = int(np.argmin(ratios))
                t1 = ratios[l]
            dep = not (delta > 1e-12 * gamma)
            t2 = np.inf if dep else -sp / delta
            t = min(t1, t2)
            if t == np.inf:
                act[p] = 0
                break
            if not dep:
                x = x + t * z
                sp = sp + t * delta
            if W.size:
                ufull[W] -= t * rvec
            up += t
            if not dep and not (t1 < t2):
                act[p] = sg
                ufull[p] = up
                cnt["adds"] += 1
                break
            c = W[l]
            act[c] = 0
            ufull[c] = 0.0
            cnt["drops"] += 1
        cnt["peak"] = max(cnt["peak"], int((act != 0).sum()))
    cnt["final"] = int((act != 0).sum())
    cnt["x"] = x
    return cnt


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 6
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    gait = sys.argv[3] if len(sys.argv) > 3 else "standing"
    h = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    f = synthetic.hard_batch(count, h, gait, 17, scale)
    rec = records.pack_records(f, h)
    ref = oracle_py.solve_records(rec, h, synthetic.DT_MPC, synthetic.F_MAX)
    qps = [QP(oracle_py.assemble_record(rec[k], h, synthetic.DT_MPC, synthetic.F_MAX)) for k in range(count)]
    print(f"{gait} h={h} x{scale:g}, {count} instances; qpOASES nWSR mean {ref['nwsr'].mean():.1f} max {ref['nwsr'].max()}")
    for kb, mr, mn, qc, br in ((48, 2, 3, 64, False), (64, 2, 3, 64, False), (64, 2, 3, 64, True), (64, 3, 3, 64, True), (120, 4, 3, 120, False), (120, 4, 3, 120, True), (120, 8, 1, 120, True)):
        rs = [emulate(q, kb, mr, mn, qc, br) for q in qps]
        err = 0.0
        for k, r in enumerate(rs):
            o = oracle_py.assemble_record(rec[k], h, synthetic.DT_MPC, synthetic.F_MAX)
            xx = np.zeros(12 * h)
            xx[o["var_ind"]] = r["x"]
            err = max(err, np.abs(xx - ref["q_soln"][k]).max() / max(1.0, np.abs(ref["q_soln"][k]).max()))
        mean = lambda key: np.mean([r[key] for r in rs])
        single = mean("adds") + mean("drops")
        inround = mean("rounds") + mean("flips") + mean("rel")
        print(f"  kbmax {kb:3d} rounds<= {mr:2d} min_new {mn}: rounds {mean('rounds'):.2f} flips {mean('flips'):.2f} releases {mean('rel'):.2f} | "
              f"single-row adds {mean('adds'):.1f} drops {mean('drops'):.1f} | kernel 'iters' {inround + single:.1f} | peak |W| {mean('peak'):.1f} "
              f"batch-release solves {mean('resolves'):.2f} | max {max(r['peak'] for r in rs)} final {mean('final'):.1f} | overflow(>{qc}) {mean('overflow'):.2f} | err {err:.1e} stuck {sum(r.get('stuck', 0) for r in rs)}")


if __name__ == "__main__":
    main()
