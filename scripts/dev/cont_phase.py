#!/usr/bin/env python3
"""Developer tool: per-phase cycle profile (-DHMPC_PROFILE build, as scripts/phase_profile.py) of an OFF-NOMINAL batch through the
device-side chain fast -> continuation (hmpc_set_device_repair 2): the instances the fast variant finished and the ones it handed
over apart -- for the latter the numbers are the CONTINUATION variant's (it overwrites the instance's profile slot).
    python scripts/dev/cont_phase.py [scale] [batch]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hector_simulation_amd import _lib, build, interface, records, synthetic  # noqa: E402,F401

PH = ["asm", "H+g", "sweep", "xu", "select", "d", "E*d", "w", "matvec", "sel:slack", "update", "polish", "final", "TOTAL", "blk:x", "blk:S0", "blk:inv", "blk:drop", "sel:a",
      "asm:load", "asm:trig", "asm:scalar", "H+g:g"]


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    h = 10
    prof_lib = os.path.join(ROOT, "gpurun_out", "libhector_mpc_hip_prof.so")
    os.makedirs(os.path.dirname(prof_lib), exist_ok=True)
    build.build_to(prof_lib, ["-DHMPC_PROFILE"])
    build.LIB = prof_lib
    build.needs_build = lambda: False
    rec = records.pack_records(synthetic.hard_batch(nb, h, "standing", 17, scale), h)
    out = {}
    for mode in (0, 2):
        mpc = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, nb)
        mpc.set_auto_resolve(False)
        mpc.set_device_repair(mode)
        mpc.upload(rec)
        mpc.solve()
        _, status = mpc.download()
        cyc = np.zeros((nb, 32), dtype=np.int64)
        interface._check(mpc.L.hmpc_debug_phase_cycles(mpc.h, cyc.ctypes.data), "phase_cycles")
        out[mode] = (status.copy(), cyc)
        mpc.close()
    st0, cyc0 = out[0]
    st2, cyc2 = out[2]
    handed = interface.status_code(st0) == 5
    print(f"standing h={h} x{scale} batch={nb}: handed over {int(handed.sum())}; after the continuation codes {dict(zip(*np.unique(interface.status_code(st2), return_counts=True)))}")
    for name, sel, cyc, st in (("fast variant, finished there", ~handed, cyc0, st0), ("fast variant, handed over (its part)", handed, cyc0, st0),
                               ("continuation variant (handed-over instances)", handed, cyc2, st2)):
        if not sel.any():
            continue
        it = interface.status_iters(st)[sel]
        mean = cyc[sel].mean(axis=0)
        print(f"-- {name}: {int(sel.sum())} instances, status iterations mean {it.mean():.1f} p90 {np.percentile(it, 90):.0f}; |W| mean {interface.status_nactive(st)[sel].mean():.1f}")
        for i, ph in enumerate(PH):
            if mean[i] > 0:
                print(f"   {ph:10s} {mean[i]:12.0f} cycles  {100 * mean[i] / max(mean[13], 1):5.1f}%")


if __name__ == "__main__":
    main()
