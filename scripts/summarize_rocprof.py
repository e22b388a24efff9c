#!/usr/bin/env python3
"""Condense gpurun_out/rocprof (written by scripts/gpu_rocprof.sh) into the tracked summaries under profiles/<round>/:
kernel_stats.csv (rocprofv3 --stats, copied), pmc.csv (per-launch means of every counter of the hmpc kernel, one row per
counter and pass) and profiles/hbm_traffic.json (what bench.py reports as roofline.traffic)."""
import csv
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "rocprof")


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
    dst = os.path.join(ROOT, "profiles", rnd)
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(SRC, "kt", "kt_kernel_stats.csv"), os.path.join(dst, "kernel_stats.csv"))
    for name in ("bench_standing.json", "bench_walking.json", "bench_h20_single.json"):
        if os.path.exists(os.path.join(SRC, name)):
            shutil.copy(os.path.join(SRC, name), os.path.join(dst, name))
    rows = []
    means = {}
    for pas in sorted(os.listdir(SRC)):
        f = os.path.join(SRC, pas, "pmc_counter_collection.csv")
        if not os.path.exists(f):
            continue
        acc = defaultdict(list)
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if "hmpc_kernel" in r["Kernel_Name"] and int(r["Grid_Size"]) >= 8192 * 128:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, v in sorted(acc.items()):
            rows.append((pas, c, len(v), sum(v) / len(v)))
            means[c] = sum(v) / len(v)
    with open(os.path.join(dst, "pmc.csv"), "w") as fh:
        fh.write("pass,counter,dispatches,mean_per_dispatch\n")
        for r in rows:
            fh.write("%s,%s,%d,%.3f\n" % r)
    if "FETCH_SIZE" in means and "WRITE_SIZE" in means:
        batch = 8192
        raw = (means["FETCH_SIZE"] + means["WRITE_SIZE"]) * 1024 / batch
        cor = (2 * means["FETCH_SIZE"] + means["WRITE_SIZE"]) * 1024 / batch
        ctr = {}
        if "SQ_WAVE_CYCLES" in means:
            wc = means["SQ_WAVE_CYCLES"]
            ctr = {
                "sq_wait_any_frac_of_wave_cycles": means.get("SQ_WAIT_ANY", 0) / wc,
                "sq_active_inst_any_frac_of_wave_cycles": means.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                "valu_insts_per_solve": means.get("SQ_INSTS_VALU", 0) / batch,
                "salu_insts_per_solve": means.get("SQ_INSTS_SALU", 0) / batch,
                "lds_insts_per_solve": means.get("SQ_INSTS_LDS", 0) / batch,
                "mfma_insts_per_solve": means.get("SQ_INSTS_MFMA", 0) / batch,
                "lds_bank_conflict_frac_of_lds_active": means.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, means.get("SQ_LDS_IDX_ACTIVE", 1.0)),
            }
            if "GRBM_GUI_ACTIVE" in means and "SQ_VALU_MFMA_BUSY_CYCLES" in means:
                cyc = means["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
                ctr["kernel_cycles_per_launch"] = cyc
                ctr["mfma_pipe_busy_frac"] = means["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)  # 256 CUs x 4 SIMDs
        json.dump({
            "horizon": 10, "gait": "standing", "batch": batch, "counters": ctr,
            "FETCH_SIZE_KB_per_launch": means["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": means["WRITE_SIZE"],
            "bytes_per_solve_raw": raw, "bytes_per_solve": cor,
            "correction": "FETCH_SIZE doubled (gfx950 rocprofv3 reports half the bytes of a coalesced stream, "
                          "MI355X_MICROARCH.md section HBM; upper bound for our 4-B/lane burst), WRITE_SIZE as reported; units KB",
            "algorithmic_bytes_per_solve": 1200,
            "source": "profiles/%s/pmc.csv (separate --pmc passes of: python bench.py --steps 10 --warmup 2 "
                      "--no-cpu-baseline --no-side-configs --check 0)" % rnd,
        }, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
    for r in rows:
        print("%-10s %-28s n=%3d mean %.4g" % r)


if __name__ == "__main__":
    main()
