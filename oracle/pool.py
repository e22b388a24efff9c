"""Process pool over the CPU oracle.  TEST INFRASTRUCTURE ONLY (tests/ and scripts/; never the product).

qpOASES has a process-global message handler (QP/src/MessageHandling.cpp:615-624), so parallel reference solves must be
PROCESSES, not threads; and the calling test process already holds a HIP context, so the workers are fresh interpreters
(``python -m oracle.pool worker ...``), not forks."""
from __future__ import annotations

import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def solve_records_parallel(rec: np.ndarray, horizon: int, dt: float, f_max: float, nc: int = 2, nproc: int | None = None):
    """``oracle_py.solve_records`` over all rows of ``rec`` with ``nproc`` worker processes (default: all host cores,
    at most 64).  Returns dict(q_soln [n, 6 nc h], nwsr [n], obj [n], n_bad, bad [n] bool)."""
    n = rec.shape[0]
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    nproc = max(1, min(nproc or cores, 64, n))
    bounds = [(i * n // nproc, (i + 1) * n // nproc) for i in range(nproc)]
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "rec.npy")
        np.save(path, np.ascontiguousarray(rec))
        procs = [subprocess.Popen([sys.executable, "-m", "oracle.pool", "worker", path, str(horizon), repr(float(dt)),
                                   repr(float(f_max)), str(nc), str(lo), str(hi), os.path.join(td, f"out{i}.npz")],
                                  cwd=ROOT, stdout=subprocess.DEVNULL)
                 for i, (lo, hi) in enumerate(bounds)]
        for p in procs:
            if p.wait() != 0:
                raise RuntimeError("oracle pool worker failed")
        parts = [np.load(os.path.join(td, f"out{i}.npz")) for i in range(nproc)]
        return dict(q_soln=np.concatenate([p["q_soln"] for p in parts]), nwsr=np.concatenate([p["nwsr"] for p in parts]),
                    obj=np.concatenate([p["obj"] for p in parts]), n_bad=int(sum(int(p["n_bad"]) for p in parts)),
                    bad=np.concatenate([p["bad"] for p in parts]))


def _worker(argv):
    path, h, dt, fmax, nc, lo, hi, out = argv[0], int(argv[1]), float(argv[2]), float(argv[3]), int(argv[4]), int(argv[5]), int(argv[6]), argv[7]
    from oracle import oracle_py

    rec = np.load(path)
    r = oracle_py.solve_records(rec, h, dt, fmax, first=lo, count=hi - lo, nc=nc)
    np.savez(out, q_soln=r["q_soln"], nwsr=r["nwsr"], obj=r["obj"], n_bad=np.int64(r["n_bad"]), bad=r["bad"])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        _worker(sys.argv[2:])
