#!/usr/bin/env python3
"""Per-phase cycle profile of the fused kernel (developer tool).  Builds a -DHMPC_PROFILE copy of the library next to
the product one, runs one batch and prints the mean shader-clock cycles thread 0 spent in each phase."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hector_simulation_amd import _lib, build, interface, records, synthetic  # noqa: E402

PH = ["asm", "H+g", "sweep", "xu", "select", "d", "E*d", "w", "matvec", "sel:slack", "update", "polish", "final", "TOTAL", "blk:x", "blk:S0", "blk:inv", "blk:drop", "sel:a", "asm:load", "asm:trig", "asm:scalar", "H+g:g"]


def main():
    gait = sys.argv[1] if len(sys.argv) > 1 else "standing"
    h = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    nb = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
    nc = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    prof_lib = os.path.join(ROOT, "gpurun_out", "libhector_mpc_hip_prof.so")
    os.makedirs(os.path.dirname(prof_lib), exist_ok=True)
    build.build_to(prof_lib, ["-DHMPC_PROFILE"])
    build.LIB = prof_lib
    build.needs_build = lambda: False
    if nc == 3:
        f = synthetic.make_batch3(nb, h, gait, seed=5, hand="contact")
        rec = records.pack_records(f, h, 3)
    else:
        f = synthetic.make_batch(nb, h, gait, seed=6, phase="random")
        rec = records.pack_records(f, h)
    mpc = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, nb, contacts=nc)
    mpc.upload(rec)
    mpc.solve()
    _, status = mpc.download()
    cyc = np.zeros((nb, 32), dtype=np.int64)
    interface._check(mpc.L.hmpc_debug_phase_cycles(mpc.h, cyc.ctypes.data), "phase_cycles")
    it = interface.status_iters(status)
    mean = cyc.mean(axis=0)
    print(f"gait={gait} h={h} batch={nb} contacts={nc} iters median {np.median(it)} mean {it.mean():.1f}")
    for i, name in enumerate(PH):
        print(f"  {name:8s} {mean[i]:12.0f} cycles  {100 * mean[i] / mean[13]:5.1f}%   per-iter {mean[i] / max(it.mean(), 1):9.0f}")


if __name__ == "__main__":
    main()
