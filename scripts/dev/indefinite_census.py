#!/usr/bin/env python3
"""Developer tool: does a fast variant ever report HMPC_S_OK for an instance whose Hessian is not positive definite?
Double support over h = 20 at `scale` x the input ranges, `nb` instances: smallest eigenvalue of the oracle's reduced Hessian per
instance (CPU, a process pool) against the status the fast pass alone returns, and against the final status / error vs qpOASES.
    python scripts/dev/indefinite_census.py [nb] [scale] [h] [gait]"""
import os
import sys

os.environ.setdefault("OMP_NUM_THREADS", "1")  # (one BLAS thread per pool process: 16 processes x 256 threads each is what an unset value gives on the GPU box)
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
scale = int(sys.argv[2]) if len(sys.argv) > 2 else 10
h = int(sys.argv[3]) if len(sys.argv) > 3 else 20
gait = sys.argv[4] if len(sys.argv) > 4 else "standing"


def min_eigs(args):
    lo, hi = args
    from hector_simulation_amd import records, synthetic
    from oracle import oracle_py

    rec = records.pack_records(synthetic.hard_batch(nb, h, gait, 17, scale), h)
    out = np.zeros(hi - lo)
    for i in range(lo, hi):
        a = oracle_py.assemble_record(rec[i], h, synthetic.DT_MPC, synthetic.F_MAX)
        out[i - lo] = np.linalg.eigvalsh(a["H_red"])[0] if a["n"] > 0 else 1.0
    return out


def main():
    cache = os.path.join(ROOT, "scripts", "dev", "_census_eig.npy")  # (computed beforehand on the CPU box: python scripts/dev/indefinite_census.py ... eig-only)
    if os.path.exists(cache) and np.load(cache).shape[0] == nb:
        eig = np.load(cache)
    else:
        workers = min(8, os.cpu_count() or 1)
        chunks = [(k * nb // workers, (k + 1) * nb // workers) for k in range(workers)]
        with ProcessPoolExecutor(workers) as ex:
            eig = np.concatenate(list(ex.map(min_eigs, chunks)))
        if len(sys.argv) > 5 and sys.argv[5] == "eig-only":
            np.save(cache, eig)
            print("smallest eigenvalues written to", cache, "-- not positive definite:", int((eig < 0).sum()), "of", nb)
            return
    import torch

    torch.zeros(1, device="cuda")
    from hector_simulation_amd import interface, records, synthetic
    from oracle import pool

    rec = records.pack_records(synthetic.hard_batch(nb, h, gait, 17, scale), h)
    m = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, nb)
    m.set_auto_resolve(False)
    m.upload(rec)
    m.solve()
    _, s0 = m.download()
    m.resolve_failed()
    f1, s1 = m.download()
    m.close()
    indef = eig < 0
    c0, c1 = interface.status_code(s0), interface.status_code(s1)
    print(f"{gait} h={h} x{scale} nb={nb}: Hessian not positive definite in {int(indef.sum())} instances (smallest eigenvalue {eig.min():.2e} .. {eig[indef].max() if indef.any() else 0:.2e})")
    print(f"  fast pass on those: codes {dict(zip(*np.unique(c0[indef], return_counts=True)))}  -> reported HMPC_S_OK by a fast variant: {int((c0[indef] == 0).sum())}")
    print(f"  after hmpc_resolve_failed: codes of those {dict(zip(*np.unique(c1[indef], return_counts=True)))}; of all {dict(zip(*np.unique(c1, return_counts=True)))}")
    ref = pool.solve_records_parallel(rec, h, synthetic.DT_MPC, synthetic.F_MAX)
    rbad = np.asarray(ref["bad"], dtype=bool)
    q = ref["q_soln"]
    err = np.abs(f1 - q).max(axis=1) / np.maximum(1, np.abs(q).max(axis=1))
    ok = c1 == 0
    print(f"  qpOASES fails on {int(rbad.sum())} ({int((rbad & indef).sum())} of them indefinite); max error where both solve: indefinite {err[ok & indef & ~rbad].max() if (ok & indef & ~rbad).any() else 0:.1e}, "
          f"definite {err[ok & ~indef & ~rbad].max():.1e}; GPU flagged & qpOASES ok: {int((~ok & ~rbad).sum())}")


if __name__ == "__main__":
    main()
