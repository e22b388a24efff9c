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


sys.path.insert(0, ROOT)


def main():
    from hector_simulation_amd import build as hip_build

    rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
    dst = os.path.join(ROOT, "profiles", rnd)
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(SRC, "kt", "kt_kernel_stats.csv"), os.path.join(dst, "kernel_stats.csv"))
    for name in ("bench_standing.json", "bench_walking.json", "bench_h20_single.json", "bench_gpus2_gloo.json", "bench_gpus8_refused.txt"):
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
                # a wave64 VALU instruction occupies its 16-lane SIMD for (at least) 4 cycles
                ctr["valu_issue_frac"] = 4.0 * means.get("SQ_INSTS_VALU", 0) / (1024.0 * cyc)
            for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64"):
                if k in means:
                    ctr[k.lower() + "_per_solve"] = means[k] / batch
        json.dump({
            "horizon": 10, "gait": "standing", "batch": batch, "counters": ctr,
            "source_hash": hip_build.source_hash(), "profile_dir": "profiles/%s" % rnd,
            "FETCH_SIZE_KB_per_launch": means["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": means["WRITE_SIZE"],
            "bytes_per_solve_raw": raw, "bytes_per_solve": cor,
            "correction": "FETCH_SIZE doubled (gfx950 rocprofv3 reports half the bytes of a coalesced stream, "
                          "MI355X_MICROARCH.md section HBM; upper bound for our 4-B/lane burst), WRITE_SIZE as reported; units KB",
            "algorithmic_bytes_per_solve": 1200,
            "source": "profiles/%s/pmc.csv (separate --pmc passes of: python bench.py --steps 10 --warmup 2 "
                      "--no-cpu-baseline --no-side-configs --check 0 --streams 1)" % rnd,
        }, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
    for r in rows:
        print("%-10s %-28s n=%3d mean %.4g" % r)
    for name in ("phase_cycles.txt", "latency_vs_batch.txt", "soak.txt", "dispatch_order.txt", "stress.txt", "gpu_tests.txt", "range_scale.txt", "continuation_pass.txt"):
        if os.path.exists(os.path.join(SRC, name)):
            shutil.copy(os.path.join(SRC, name), os.path.join(dst, name))
    extra = []
    ts = two_stream_trace(dst)
    if ts:
        extra.append(ts)
    extra += variant_lines(dst)
    write_readme(dst, rnd, means, extra)


def _hmpc_dispatches(path, min_grid=0):
    out = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if "hmpc_kernel" in r["Kernel_Name"] and int(r["Grid_Size_X"]) >= min_grid:
                out.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r.get("Stream_Id", 0) or 0), int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])))
    return sorted(out)


def two_stream_trace(dst):
    """The DEFAULT bench mode alternates two launch streams: the kernel trace of that run (start / end of consecutive
    dispatches) shows the overlap the headline value comes from, and the value follows from the trace alone."""
    src = os.path.join(SRC, "kt2", "kt2_kernel_trace.csv")
    if not os.path.exists(src):
        return None
    d = _hmpc_dispatches(src, 8192 * 128)
    if len(d) < 24:
        return None
    timed = d[3:23]  # 3 warm-up steps, then the 20 timed ones (bench.py --steps 20 --warmup 3)
    with open(os.path.join(dst, "two_stream_kernel_trace.csv"), "w") as fh:
        fh.write("dispatch,stream,start_us,end_us,duration_us,overlap_with_previous_us\n")
        t0 = timed[0][0]
        prev_end = None
        ov_total = 0.0
        for i, (a, b, st, wgs) in enumerate(timed):
            ov = max(0.0, (prev_end - a) / 1e3) if prev_end is not None else 0.0
            ov_total += ov
            fh.write("%d,%d,%.1f,%.1f,%.1f,%.1f\n" % (i, st, (a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3, ov))
            prev_end = b
    span = (timed[-1][1] - timed[0][0]) / 1e9
    dur = sum(b - a for a, b, _, _ in timed) / len(timed) / 1e6
    nb = timed[0][3]
    return ("* `two_stream_kernel_trace.csv` (rocprofv3 --kernel-trace of the DEFAULT run, `python bench.py --steps 20 --warmup 3`): "
            "the 20 timed dispatches span %.3f ms = **%.3f M solves/s** from the trace alone (%d instances each); a dispatch lasts "
            "%.3f ms on average while a new one starts every %.3f ms -- consecutive dispatches (alternating streams) overlap by "
            "%.3f ms on average: the partly filled last rounds of workgroups of one launch run under the first rounds of the next"
            % (span * 1e3, len(timed) * nb / span / 1e6, nb, dur, span * 1e3 / len(timed), ov_total / (len(timed) - 1) / 1e3))


def variant_lines(dst):
    """One line per other kernel variant: rocprofv3 --stats duration + one SQ counter pass (scripts/gpu_rocprof.sh part 4)."""
    lines = []
    rowsv = []
    for case in sorted(os.listdir(SRC)):
        if not case.startswith("var_") or not os.path.isdir(os.path.join(SRC, case)):
            continue
        ks = os.path.join(SRC, case, "kt", "kt_kernel_stats.csv")
        pm = os.path.join(SRC, case, "pmc", "pmc_counter_collection.csv")
        if not os.path.exists(ks):
            continue
        name, avg, calls = None, 0.0, 0
        with open(ks) as fh:
            for r in csv.DictReader(fh):
                if "hmpc_kernel" in r["Name"] and int(r["Calls"]) > calls:
                    name, avg, calls = r["Name"].split("(")[0].replace("void hmpc::", ""), float(r["AverageNs"]) / 1e6, int(r["Calls"])
        nb = int(case.split("_b")[-1])
        acc = defaultdict(list)
        if os.path.exists(pm):
            with open(pm) as fh:
                for r in csv.DictReader(fh):
                    if "hmpc_kernel" in r["Kernel_Name"]:
                        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        m = {k: sum(v) / len(v) for k, v in acc.items()}
        wc = m.get("SQ_WAVE_CYCLES", 0.0)
        cyc = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        line = "* `%s` (%s, %d instances per launch): rocprofv3 --stats %d calls, average **%.4f ms** = %.3f M solves/s" % (
            case[4:], name, nb, calls, avg, nb / avg / 1e3)
        if wc:
            line += "; per solve %.0f VALU / %.0f SALU / %.0f LDS wave-instructions, SQ_WAIT_ANY %.1f %% / SQ_ACTIVE_INST_ANY %.1f %% of wave cycles" % (
                m.get("SQ_INSTS_VALU", 0) / nb, m.get("SQ_INSTS_SALU", 0) / nb, m.get("SQ_INSTS_LDS", 0) / nb,
                100 * m.get("SQ_WAIT_ANY", 0) / wc, 100 * m.get("SQ_ACTIVE_INST_ANY", 0) / wc)
        if cyc:
            line += ", VALU issue occupancy %.1f %%" % (100 * 4.0 * m.get("SQ_INSTS_VALU", 0) / (1024.0 * cyc))
        lines.append(line)
        rowsv.append((case[4:], name, nb, calls, avg) + tuple(m.get(k, 0.0) for k in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE")))
    if rowsv:
        with open(os.path.join(dst, "variants.csv"), "w") as fh:
            fh.write("case,kernel,instances,calls,avg_ms,SQ_WAVES,SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_WAVE_CYCLES,SQ_WAIT_ANY,SQ_ACTIVE_INST_ANY,GRBM_GUI_ACTIVE\n")
            for r in rowsv:
                fh.write('%s,"%s",%d,%d,%.4f,' % r[:5] + ",".join("%.0f" % x for x in r[5:]) + "\n")
        lines.insert(0, "* other kernel variants (`variants.csv`; `python scripts/quick_times.py <case>` under rocprofv3, counters are per-launch means):")
    return lines


def write_readme(dst, rnd, means, extra=()):
    """profiles/<round>/README.md generated from the CSV/JSON files next to it -- no hand-typed figures."""
    lines = ["# profiles/%s -- generated by scripts/summarize_rocprof.py from the files in this directory" % rnd, "",
             "Collected with `scripts/gpu_rocprof.sh` (one `gpurun` call; rocprofv3 passes kept separate: kernel trace + "
             "stats, then one `--pmc` pass per counter group).  Command profiled: `python bench.py --steps 10 --warmup 2 "
             "--no-cpu-baseline --no-side-configs --check 0 --streams 1` (serial launches on one stream, so that per-kernel durations are not stretched by the overlap of the default two-stream mode; 8192 randomized 2-contact h=10 instances per launch, records "
             "resident in HBM).", ""]
    ks = os.path.join(dst, "kernel_stats.csv")
    with open(ks) as fh:
        for r in csv.DictReader(fh):
            if "hmpc_kernel" in r["Name"]:
                avg_ms = float(r["AverageNs"]) / 1e6
                lines.append("* `kernel_stats.csv`: `%s` -- %s calls, average **%.4f ms** per 8192-instance launch = %.3f M "
                             "solves/s (kernel only)" % (r["Name"].split("(")[0], r["Calls"], avg_ms, 8192 / avg_ms / 1e3))
    bj = os.path.join(dst, "bench_standing.json")
    if os.path.exists(bj):
        try:
            b = json.loads(open(bj).read().strip().splitlines()[-1])
            lines.append("* `bench_standing.json`: value %.4g %s, ms_per_step %.4f, HIP-event kernel_ms %.4f, roofline.frac %.3g, "
                         "roofline_mfma.frac %.3g, fp64_valu_frac %.3g, iterations/solve %.2f; cpu_baseline (%s, %d cores) %.0f solves/s"
                         % (b["value"], b["unit"], b["ms_per_step"], b["roofline"]["kernel_ms"], b["roofline"]["frac"],
                            b["roofline_mfma"]["frac"], b.get("fp64_valu_frac", float("nan")), b.get("iterations_per_solve", float("nan")),
                            b.get("cpu_baseline", {}).get("kind", "-"), b.get("cpu_baseline", {}).get("cores", 0),
                            b.get("cpu_baseline", {}).get("value", float("nan"))))
        except Exception as exc:  # keep the README honest rather than failing the summary
            lines.append("* `bench_standing.json`: could not be parsed (%r)" % (exc,))
    tj = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tj):
        t = json.load(open(tj))
        c = t.get("counters", {})
        lines.append("* `pmc.csv` -> `../hbm_traffic.json`: FETCH_SIZE %.0f KB + WRITE_SIZE %.0f KB per launch = %.0f B/solve raw, "
                     "%.0f B/solve with the gfx950 x2 FETCH correction, against 1200 B/solve algorithmic"
                     % (t["FETCH_SIZE_KB_per_launch"], t["WRITE_SIZE_KB_per_launch"], t["bytes_per_solve_raw"], t["bytes_per_solve"]))
        if c:
            lines.append("* per solve: %.0f VALU / %.0f SALU / %.0f LDS / %.0f MFMA wave-instructions; SQ_WAIT_ANY %.1f %% and "
                         "SQ_ACTIVE_INST_ANY %.1f %% of wave cycles; LDS bank conflicts %.1f %% of LDS-active cycles"
                         % (c.get("valu_insts_per_solve", 0), c.get("salu_insts_per_solve", 0), c.get("lds_insts_per_solve", 0),
                            c.get("mfma_insts_per_solve", 0), 100 * c.get("sq_wait_any_frac_of_wave_cycles", 0),
                            100 * c.get("sq_active_inst_any_frac_of_wave_cycles", 0),
                            100 * c.get("lds_bank_conflict_frac_of_lds_active", 0)))
            if "valu_issue_frac" in c:
                lines.append("* kernel %.0f cycles per launch; VALU issue occupancy (4 cycles per wave instruction / SIMD cycles) "
                             "%.1f %%; matrix pipe busy %.2f %%" % (c["kernel_cycles_per_launch"], 100 * c["valu_issue_frac"],
                                                                     100 * c["mfma_pipe_busy_frac"]))
            f64 = [k for k in c if k.endswith("_f64_per_solve")]
            if f64:
                lines.append("* binary64 VALU instructions per solve: " + ", ".join("%s %.0f" % (k[len("sq_insts_valu_"):-len("_per_solve")], c[k]) for k in sorted(f64)))
        lines.append("* library source hash of the profiled build: `%s`" % t.get("source_hash", "?"))
    lines += list(extra)
    for other in sorted(os.listdir(dst)):
        if other.endswith(".txt") or (other.endswith(".json") and other != "bench_standing.json"):
            lines.append("* `%s`" % other)
    open(os.path.join(dst, "README.md"), "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
