#!/usr/bin/env python3
"""bench.py -- MPC QP solves/sec (horizon=10, 2 contacts) of the HIP path, with roofline and CPU-baseline legs.

    python bench.py --gpus N --steps K --warmup W
    python bench.py --gpus 8 --gait walking                  # BASELINE config 3: 8 x 8 192 walking sweep
    python bench.py --gpus 4 --contacts 3 --batch 2048       # BASELINE config 5: 4 x 2 048 three-contact QPs
    python bench.py --gpus N --exchange none                 # diagnosis: N ranks, NO collective -- per-rank kernel-only rates
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Both forms run N ranks, one per GPU: started WITHOUT torchrun (no WORLD_SIZE in the environment) and N > 1, bench.py
re-executes itself under `python -m torch.distributed.run --nproc-per-node N` (127.0.0.1, a free port) and passes the
ranks' exit code on -- it never silently runs one GPU.  It exits non-zero, with the reason on stderr, when fewer than N
devices are visible or when an explicit --gpus disagrees with the WORLD_SIZE it was launched with (no --gpus under
torchrun = WORLD_SIZE ranks).  Rank bring-up (process group over RCCL + the first collective) runs under a 120-second
watchdog: a rank that cannot bring RCCL up prints its device list, the RCCL version and the HSA_* / NCCL_* / RCCL_*
environment to stderr and the job exits non-zero within two minutes instead of hanging; `--exchange none` keeps the
control plane (barriers, the max over ranks) on gloo and runs no data-path collective at all, so that a broken collective
still yields a diagnosable per-rank line.
(`--backend gloo` is a TEST transport that lets the N ranks share GPUs: `python bench.py --gpus 2 --backend gloo` on a
one-GPU box runs the N = 2 code path, every rank checking its own shard.)

A "step" is one pass of the hot path (assembly + QP solve, one kernel launch) over one batch of synthetic MPC
instances whose packed records already live in HBM (consecutive steps alternate between two launch streams / two output
blocks, --streams 1 for strictly serial launches); for N > 1 every rank owns a contiguous shard of the global batch
(weak scaling, per-GPU batch fixed) and the step ends with the all_gather (RCCL) of the solved forces.
The timed region is K steps between barrier + synchronize on both sides, max over ranks; it is repeated --windows times
(default 5) and `value` / `ms_per_step` are the MEDIAN window's, with every window's value, min and max in `windows`.
Rank 0 prints ONE JSON line.  Workload = the case BASELINE.json's metric string names: randomized 2-contact
(standing gait) instances, horizon 10 -> 120 x 160 QPs (SURVEY.md section 8d "metric_2contact").
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_SOLVE = {10: 1200, 20: 2180}   # (54+12h)*4 + 2h in, 12h*4 + 4 out
MFLOP_PER_SOLVE = {10: 3.744, 20: 29.952}  # 2*(12h)^2*(13h) dense B'SB contraction
HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_F32_PEAK_TF = 157.3
FP64_VALU_PEAK_TF = 78.6                   # vector fp64 = half the fp32 vector rate of MI355X_MICROARCH.md (157.3 / 2)


def cpu_worker(path: str, horizon: int, first: int, count: int, kind: str = "reference", repeat: int = 1) -> None:
    """One CPU-baseline process over instances [first, first+count) of the saved field arrays, `repeat` times over (the timed
    region covers all passes: a sample of the wanted wall time from a slice of the batch).

    kind "reference": the reference's OWN source end to end -- setup_problem / update_problem_data / get_solution of
    oracle/_ref/libsolvempc_ref.so (ConvexMPC/SolverMPC.cpp + RobotState.cpp + convexMPC_interface.cpp compiled
    unmodified against the Eigen stand-in, linked with the reference's vendored qpOASES; oracle/Makefile).  Its three
    printed lines per solve go to /dev/null.
    kind "port": the oracle's C restatement of the assembly + the same qpOASES (prints removed)."""
    from hector_simulation_amd import records, synthetic

    f = dict(np.load(path))
    if kind == "reference":
        from oracle import ref_py

        ref_py.lib()
        # the reference reports a failed solve only by printing "failed to solve!" (SolverMPC.cpp:714-715): its stdout goes
        # to a scratch file whose matching lines are counted afterwards (the prints stay part of the timed work)
        log = path + f".stdout.{first}"
        ref_py.silence_forever(log)
        t0 = time.perf_counter()
        for _ in range(max(1, repeat)):
            q = ref_py.solve_fields(f, horizon, synthetic.DT_MPC, 0.25, synthetic.F_MAX, first=first, count=count)
        t1 = time.perf_counter()
        ref_py.flush_stdio()
        with open(log, "rb") as fh:
            n_failed_lines = fh.read().count(b"failed to solve!")
        os.unlink(log)
        os.write(2, (json.dumps(dict(count=count * max(1, repeat), wall=t1 - t0,
                                     n_bad=int(n_failed_lines + np.isnan(q).any(axis=1).sum()))) + "\n").encode())
        return
    from oracle import oracle_py

    rec = records.pack_records(f, horizon)
    t0 = time.perf_counter()
    r = oracle_py.solve_records(rec, horizon, synthetic.DT_MPC, synthetic.F_MAX, first=first, count=count)
    t1 = time.perf_counter()
    os.write(2, (json.dumps(dict(count=count, wall=t1 - t0, t_assemble=r["t_assemble"], t_solve=r["t_solve"],
                                 n_bad=int(r["n_bad"]), nwsr_med=float(np.median(r["nwsr"])), nwsr_max=int(r["nwsr"].max()),
                                 nwsr_hist=np.bincount(np.minimum(r["nwsr"] // 10, 9), minlength=10).tolist())) + "\n").encode())


def _run_worker(path, horizon, first, count, kind, repeat=1):
    return subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", path, str(horizon), str(first),
                             str(count), kind, str(repeat)], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)


def device_description(torch, local_rank: int) -> dict:
    """Which physical device a rank ran on (the scaling line lists one per rank: two ranks on one device would show)."""
    pr = torch.cuda.get_device_properties(local_rank)
    d = {"local_rank_device": int(local_rank), "name": pr.name}
    for k in ("pci_bus_id", "pci_device_id", "pci_domain_id", "uuid", "gcnArchName", "multi_processor_count"):
        v = getattr(pr, k, None)
        if v is not None:
            d[k] = v if isinstance(v, (int, str)) else str(v)
    d["visible_devices_env"] = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES")
    return d


def rccl_version(torch):
    try:
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception as exc:  # the line must not depend on it
        return repr(exc)


def host_description() -> dict:
    """What the GPU box's host is ('core count stated'): logical CPUs the process may run on, and -- where
    /proc/cpuinfo tells -- sockets, physical cores and threads per core."""
    d = {"nproc_affinity": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
         "os_cpu_count": os.cpu_count()}
    try:
        phys, model, logical = set(), None, 0
        cur = {}
        for line in open("/proc/cpuinfo"):
            if ":" in line:
                k, v = [t.strip() for t in line.split(":", 1)]
                cur[k] = v
                if k == "model name" and model is None:
                    model = v
            elif cur:
                logical += 1
                if "physical id" in cur and "core id" in cur:
                    phys.add((cur["physical id"], cur["core id"]))
                cur = {}
        if cur:
            logical += 1
            if "physical id" in cur and "core id" in cur:
                phys.add((cur["physical id"], cur["core id"]))
        d.update(model=model, logical_cpus=logical, physical_cores=len(phys) or None,
                 sockets=len({p for p, _ in phys}) or None,
                 threads_per_core=(logical // len(phys)) if phys else None)
    except OSError:
        pass
    try:
        # a cgroup CPU quota (container limit) caps the usable CPU time below the visible CPU count
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        d["cgroup_cpu_max"] = None if q[0] == "max" else float(q[0]) / float(q[1])
    except (OSError, ValueError, IndexError):
        d["cgroup_cpu_max"] = None
    return d


def cpu_baseline(fields: dict, horizon: int, per_core: int, target_s: float = 10.0, sweep_s: float = 3.0, reps: int = 3) -> dict:
    """Reference CPU path timed on the host cores as PROCESSES (qpOASES has a process-global message handler), with a
    sweep over the process count P so that the line says where the box saturates.

    Quotable: the point at P = `cores` is the MEDIAN of `reps` runs of >= `target_s` seconds of
    wall time each (every process passes over its own slice of the batch as often as that takes; the passes are inside the
    timed region), min and max beside it; every sweep point runs >= `sweep_s` seconds.  A 1.8-second sample swung 2x from
    run to run."""
    from oracle import ref_py

    host = host_description()
    avail = host["nproc_affinity"] or (os.cpu_count() or 1)
    # the CPUs this container may actually use: a cgroup quota (cpu.max) below the visible CPU count is the real core count --
    # more processes than that only time-slice (measured on the GPU box: 256 logical CPUs visible, quota 16, best aggregate
    # rate at P = 16, 64 processes 20 % slower)
    usable = int(np.ceil(host["cgroup_cpu_max"])) if host.get("cgroup_cpu_max") else avail
    cores = max(1, min(avail, usable, 64))
    nb = int(np.asarray(fields["p"]).shape[0])
    per_core = max(1, min(per_core, nb // cores))
    kind = "reference" if ref_py.available() else "port"
    last = lambda p: json.loads(p.communicate()[1].strip().splitlines()[-1])

    def run_p(path, P, per, repeat):
        t0 = time.perf_counter()
        procs = [_run_worker(path, horizon, c * per, per, kind, repeat) for c in range(P)]
        res = [last(p) for p in procs]
        return res, time.perf_counter() - t0

    def timed_point(path, P, per, want_s, rate_guess):
        """one run of about want_s seconds at P processes: the repeat count comes from the current rate estimate (per process)"""
        repeat = max(1, int(np.ceil(want_s * max(rate_guess, 1e-9) / per)))
        res, wall = run_p(path, P, per, repeat)
        inner = max(r["wall"] for r in res)  # slowest worker, excluding interpreter start-up
        return dict(processes=P, solves=P * per * repeat, inner_s=inner, wall_s=wall, solves_per_s=P * per * repeat / inner,
                    per_process=per * repeat / inner, n_bad=sum(r["n_bad"] for r in res))

    t_start = time.perf_counter()
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "fields.npz")
        n_saved = min(nb, max(per_core * cores, 64 * 64))
        np.savez(path, **{k: np.asarray(v)[:n_saved] for k, v in fields.items()})
        # P = 1: one process alone on the box (the reference's own operating point: one controller, one core)
        solo_n = min(96, n_saved)
        cal = last(_run_worker(path, horizon, 0, solo_n, kind))         # calibration pass (also warms the page cache)
        solo_rate0 = solo_n / cal["wall"]
        solo = timed_point(path, 1, solo_n, sweep_s, solo_rate0)
        solo_port = last(_run_worker(path, horizon, 0, solo_n, "port"))
        solo_rate = solo["solves_per_s"]
        # P-sweep (same instances for every P: process c takes slice c), >= sweep_s seconds per point
        sweep = []
        guess = solo_rate
        for P in (2, 4, 8, 16, 32, 64):
            if P == cores or P > min(avail, 4 * cores):  # up to 4x oversubscription of the usable CPUs: shows the plateau
                continue
            per = max(8, min(per_core, 128, n_saved // P))
            pt = timed_point(path, P, per, sweep_s, guess * (0.9 if P <= cores else 0.9 * cores / P))
            guess = pt["per_process"] if P < cores else guess
            sweep.append(pt)
        # the quoted point: P = cores, `reps` runs of >= target_s seconds, median
        runs = []
        guess_full = min([x["per_process"] for x in sweep if x["processes"] <= cores] or [solo_rate * 0.5])
        for _ in range(reps):
            pt = timed_point(path, cores, per_core, target_s, guess_full)
            guess_full = pt["per_process"]
            runs.append(pt)
    rates = sorted(r["solves_per_s"] for r in runs)
    value = float(np.median(rates))
    med = min(runs, key=lambda r: abs(r["solves_per_s"] - value))
    sweep_pts = sorted([{"processes": 1, "solves_per_s": solo_rate, "per_process": solo_rate, "wall_s": solo["inner_s"]}] +
                       [{"processes": x["processes"], "solves_per_s": x["solves_per_s"], "per_process": x["per_process"],
                         "wall_s": x["inner_s"]} for x in sweep] +
                       [{"processes": cores, "solves_per_s": value, "per_process": value / cores, "wall_s": med["inner_s"]}],
                       key=lambda x: x["processes"])
    # saturation point: the smallest P that already delivers 90 % of the best aggregate rate of the sweep
    best = max(x["solves_per_s"] for x in sweep_pts)
    sat = next(x["processes"] for x in sweep_pts if x["solves_per_s"] >= 0.9 * best)
    eff = value / (cores * solo_rate)
    why = (f"{cores} processes deliver {value / solo_rate:.1f}x one process alone (parallel efficiency {eff:.2f}); the sweep "
           f"reaches 90 % of its best aggregate rate at P = {sat}.  ")
    pc, lc = host.get("physical_cores"), host.get("logical_cpus")
    if host.get("cgroup_cpu_max"):
        why += (f"The container's CPU quota (cgroup cpu.max) is {host['cgroup_cpu_max']:.0f} CPUs of the {lc} logical ones visible "
                f"({pc} physical cores): `cores` = {cores} is what the baseline can really use; more processes than that "
                "time-slice the same quota (the sweep's points above it).  ")
    elif pc and lc and pc < lc:
        why += f"The box has {pc} physical cores behind {lc} logical CPUs (SMT).  "
    why += ("Below the quota the per-process rate falls with P as well: every reference solve frees and re-allocates its 12 "
            "qpOASES/Eigen buffers (resize_qp_mats, SolverMPC.cpp:196-299) and streams ~0.6 MB of dense matrices per tick, so "
            "the processes also contend for the allocator, memory bandwidth and shared caches.")
    what = ("the reference's own SolverMPC.cpp/RobotState.cpp/convexMPC_interface.cpp compiled unmodified against the Eigen "
            "stand-in oracle/mini_eigen (naive k-ascending products: its assembly is slower than real Eigen's would be) + "
            "the reference's vendored qpOASES 3.2.0, driven through setup_problem/update_problem_data/get_solution; its "
            "three printed lines per solve go to a scratch file") if kind == "reference" else \
           "oracle C restatement of the fp32 assembly + the reference's own vendored qpOASES 3.2.0 (oracle/_ref), prints removed"
    return dict(value=value, unit="QP solves/s", cores=cores, kind=kind,
                sample=(f"median of {reps} runs of >= {target_s:.0f} s wall each at {cores} processes: {med['solves']} solves per run "
                        f"({per_core} of the bench's 2-contact h={horizon} instances per process, passed over "
                        f"{med['solves'] // (per_core * cores)} times inside the timed region); ") + what,
                runs_solves_per_s=[r["solves_per_s"] for r in runs], runs_wall_s=[r["inner_s"] for r in runs],
                value_min=rates[0], value_max=rates[-1], spread=(rates[-1] - rates[0]) / value,
                host=host, process_sweep=sweep_pts, saturation_processes=sat, parallel_efficiency_at_all_cores=eff,
                scaling_note=why,
                single_process_alone_value=solo_rate, single_process_alone_ms=1e3 / solo_rate,
                per_process_value_under_full_load=value / cores,
                port_single_process_alone_value=solo_n / (solo_port["t_assemble"] + solo_port["t_solve"]),
                port_single_process_alone_ms={"assemble": 1e3 * solo_port["t_assemble"] / solo_n,
                                              "solve": 1e3 * solo_port["t_solve"] / solo_n},
                nwsr_median=solo_port["nwsr_med"], nwsr_max=solo_port["nwsr_max"], nwsr_hist_by_10=solo_port["nwsr_hist"],
                n_failed=sum(r["n_bad"] for r in runs), wall_s=time.perf_counter() - t_start)


def bench_shard(rank: int, batch: int, horizon: int, gait: str, contacts: int = 2):
    """The synthetic shard rank `rank` of a bench run owns: (fields, packed records).  Seed 6 + 1000 rank, random gait
    phase -- tests/test_gpu_full_batch.py solves exactly these shards of BASELINE configs 3 and 5, every instance."""
    from hector_simulation_amd import records, synthetic

    if contacts == 3:
        fields = synthetic.make_batch3(batch, horizon, gait, seed=6 + 1000 * rank, phase="random", hand="contact")
    else:
        fields = synthetic.make_batch(batch, horizon, gait, seed=6 + 1000 * rank, phase="random")
    return fields, records.pack_records(fields, horizon, contacts)


def bench_builder(args, torch, local_rank) -> None:
    """Rows f1-f3 on their own: device-side record builder and body-frame wrench kernels.  Streaming
    kernels: algorithmic bytes = 408 B tick in + 720 B record out (+16 B) and 480+72 B in / 96 B out per instance."""
    import ctypes as C

    from hector_simulation_amd import interface, synthetic

    h, B = args.horizon, args.batch
    dev = torch.device("cuda", local_rank)
    ticks = synthetic.make_ticks(B, h, "walking", seed=11)
    d_ticks = torch.from_numpy(ticks.view(np.uint8).reshape(B, -1)).to(dev)
    d_wpd = torch.zeros((B, 2), dtype=torch.float64, device=dev)
    d_rb = torch.from_numpy(np.ascontiguousarray(ticks["rBody"])).to(dev)
    d_fff = torch.zeros((B, 12), dtype=torch.float64, device=dev)
    mpc = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, B, device=local_rank)
    stream = torch.cuda.current_stream().cuda_stream
    L = mpc.L

    def step():
        interface._check(L.hmpc_build_records_device(mpc.h, C.c_void_p(d_ticks.data_ptr()), B, synthetic.DT_MPC,
                                                     C.c_void_p(d_wpd.data_ptr()), C.c_void_p(stream)), "build")
        interface._check(L.hmpc_body_wrench_device(mpc.h, C.c_void_p(d_rb.data_ptr()), C.c_void_p(d_fff.data_ptr()),
                                                   C.c_void_p(stream)), "wrench")

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kms = e0.elapsed_time(e1) / args.steps
    stride = mpc.stride
    bytes_per = 408 + stride + 16 + 48 * h + 72 + 96
    out = {"metric": "MPC tick records built + wrenches rotated per second (rows f1-f3)", "value": B * args.steps / elapsed,
           "unit": "instances/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"updateMPCIfNeeded input builder + gait table + body-frame wrench, horizon {h}",
                      "batch_per_gpu": B},
           "roofline": {"bound": "hbm", "achieved": B * bytes_per / (kms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": B * bytes_per / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "kernel": "build_records_kernel + body_wrench_kernel", "kernel_ms": kms,
                        "algorithmic_bytes_per_instance": bytes_per}}
    print(json.dumps(out), flush=True)
    mpc.close()


def parity_of_rank(args, rank, rec, fields, h, nc, mpc, d_forces, status) -> dict:
    """THIS rank's first `--check` instances of the timed batch against the oracle (checker only, after the timed region):
    forces and objective against qpOASES on the oracle's (bit-identical) QP data, KKT quantities of the kernel's binary64
    solution, and -- two contacts, reference library present -- the same instances END TO END through the reference's own
    source (its update_problem_data -> get_solution), as numbers measured in this run."""
    from hector_simulation_amd import interface, synthetic
    from oracle import oracle_py

    B = rec.shape[0]
    nchk = min(args.check, B)
    ref = oracle_py.solve_records(rec, h, synthetic.DT_MPC, synthetic.F_MAX, first=0, count=nchk, nc=nc)
    f = d_forces[:nchk].cpu().numpy().astype(np.float64)
    q = ref["q_soln"]
    err = np.abs(f - q).max(axis=1) / np.maximum(1.0, np.abs(q).max(axis=1))
    # objective and KKT quantities (SolverMPC.cpp:699-712 -> qpOASES getObjVal): the kernel's own
    # binary64 solution and objective against qpOASES on the oracle's (bit-identical) reduced QP
    x64, obj64 = mpc.download_f64()
    gap = np.abs(obj64[:nchk] - ref["obj"]) / np.maximum(1.0, np.abs(ref["obj"]))
    nk = min(nchk, 32)
    sub = viol = stat = 0.0
    for k in range(nk):
        o = oracle_py.assemble_record(rec[k], h, synthetic.DT_MPC, synthetic.F_MAX, nc=nc)
        x = x64[k][o["var_ind"]]
        Hk, gk, Ak = o["H_red"], o["g_red"], o["A_red"]
        den = max(1.0, abs(ref["obj"][k]))
        sub = max(sub, (0.5 * x @ Hk @ x + gk @ x - ref["obj"][k]) / den)
        ax = Ak @ x
        scale = max(1.0, np.abs(x).max())
        viol = max(viol, max(0.0, (o["lb_red"] - ax).max(), (ax - o["ub_red"]).max()) / scale)
        act = (np.abs(ax - o["lb_red"]) <= 1e-7 * scale) | (np.abs(ax - o["ub_red"]) <= 1e-7 * scale)
        grad = Hk @ x + gk  # stationarity: grad in the span of the active rows (least-squares multipliers)
        if act.any():
            lam = np.linalg.lstsq(Ak[act].T, grad, rcond=None)[0]
            grad = grad - Ak[act].T @ lam
        stat = max(stat, np.abs(grad).max() / max(1.0, np.abs(gk).max()))
    out = {"rank": rank, "checked": nchk, "max_rel_force_err_vs_qpoases": float(err.max()),
           "max_rel_objective_gap": float(gap.max()),
           "not_ok_in_shard": int((interface.status_code(status) != 0).sum()),
           "kkt": {"checked": nk, "max_rel_suboptimality": float(sub), "max_rel_row_violation": float(viol),
                   "max_rel_stationarity_residual": float(stat)},
           "qpoases_failed": int(ref["n_bad"])}
    # end to end against the reference's OWN source on the same inputs (oracle/_ref/libsolvempc_ref.so; h = 10, two contacts:
    # the shapes its code can run).  The QP data differ from the kernel's by binary32 round-off (HMPC-A1 vs the reference's
    # SSE2 build), which cond(H) ~ 2.4e6 turns into force differences well above the solver's own error: reported as
    # measured, next to the bit-identical-QP figure above -- the caveat every headline must carry.
    try:
        from oracle import ref_py

        if nc == 2 and h == 10 and ref_py.available():
            qs = ref_py.solve_fields(fields, h, synthetic.DT_MPC, 0.25, synthetic.F_MAX, first=0, count=nchk)
            e2 = np.abs(f - qs).max(axis=1) / np.maximum(1.0, np.abs(qs).max(axis=1))
            out["vs_reference_source_end_to_end"] = {
                "checked": nchk, "max_rel_force_err": float(e2.max()), "median_rel_force_err": float(np.median(e2)),
                "fraction_above_1e-4": float((e2 > 1e-4).mean()),
                "what": "HIP forces vs the reference's own SolverMPC.cpp + qpOASES (compiled unmodified against the Eigen "
                        "stand-in) on the same inputs; differences are QP-data round-off amplified by cond(H), see "
                        "tests/test_reference_source.py (objective, feasibility and suboptimality in the reference's own QP)"}
    except Exception as exc:  # the checker library is optional on a box without oracle/_ref
        out["vs_reference_source_end_to_end"] = {"error": repr(exc)}
    return out


def merge_parity(per_rank: list) -> dict:
    """max over ranks of every rank's own check (a wrong shard on any rank shows here)."""
    per_rank = [p for p in per_rank if p]
    worst = lambda key: max(p[key] for p in per_rank)
    out = {"ranks_checked": len(per_rank), "checked_per_rank": per_rank[0]["checked"],
           "checked": sum(p["checked"] for p in per_rank),
           "max_rel_force_err_vs_qpoases": worst("max_rel_force_err_vs_qpoases"),
           "max_rel_objective_gap": worst("max_rel_objective_gap"),
           "not_ok_over_all_shards": sum(p["not_ok_in_shard"] for p in per_rank),
           "kkt": {"checked": sum(p["kkt"]["checked"] for p in per_rank),
                   **{k: max(p["kkt"][k] for p in per_rank) for k in
                      ("max_rel_suboptimality", "max_rel_row_violation", "max_rel_stationarity_residual")},
                   "note": "binary64 solution of the kernel on the oracle's reduced QP (bit-identical QP data); "
                           "stationarity = |Hx + g - A_act' lambda|_inf / max(1, |g|_inf), least-squares lambda"},
           "qpoases_failed": sum(p["qpoases_failed"] for p in per_rank),
           "per_rank_max_rel_force_err": [p["max_rel_force_err_vs_qpoases"] for p in per_rank]}
    e2e = [p["vs_reference_source_end_to_end"] for p in per_rank if "max_rel_force_err" in p.get("vs_reference_source_end_to_end", {})]
    if e2e:
        out["vs_reference_source_end_to_end"] = {
            "checked": sum(e["checked"] for e in e2e), "max_rel_force_err": max(e["max_rel_force_err"] for e in e2e),
            "median_rel_force_err": float(np.median([e["median_rel_force_err"] for e in e2e])),
            "fraction_above_1e-4": float(np.mean([e["fraction_above_1e-4"] for e in e2e])), "what": e2e[0]["what"]}
    else:
        out["vs_reference_source_end_to_end"] = None  # three contacts / h != 10: the reference has no code for the shape
    return out


DIAG_ENV_PREFIXES = ("HSA_", "NCCL_", "RCCL_", "HIP_", "ROCR_", "CUDA_VISIBLE", "TORCH_NCCL", "TORCH_DISTRIBUTED", "MASTER_", "GLOO_")
DIAG_ENV_NAMES = ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK")


def rank_diagnostics(rank, local_rank, what: str, exc=None) -> str:
    """What a maintainer needs when a rank cannot be brought up (first contact with an N-GPU node): devices visible to the
    rank, the RCCL torch was built with, and every environment variable that steers HSA / RCCL / the rendezvous."""
    lines = [f"[bench] rank {rank} (LOCAL_RANK {local_rank}): {what}" + (f": {exc!r}" if exc is not None else "")]
    try:
        import torch

        nd = torch.cuda.device_count()
        lines.append(f"[bench]   torch {torch.__version__}, hip {getattr(torch.version, 'hip', None)}, RCCL {rccl_version(torch)}, "
                     f"{nd} device(s) visible")
        for i in range(nd):
            pr = torch.cuda.get_device_properties(i)
            lines.append(f"[bench]   device {i}: {pr.name} {getattr(pr, 'gcnArchName', '')} pci {getattr(pr, 'pci_bus_id', '?')} "
                         f"{pr.total_memory >> 30} GiB")
    except Exception as e2:  # diagnostics must never raise
        lines.append(f"[bench]   (device query failed: {e2!r})")
    env = {k: v for k, v in sorted(os.environ.items()) if k.startswith(DIAG_ENV_PREFIXES) or k in DIAG_ENV_NAMES}
    lines.append("[bench]   env: " + (" ".join(f"{k}={v}" for k, v in env.items()) or "(none of HSA_* NCCL_* RCCL_* HIP_* ROCR_* set)"))
    lines.append("[bench]   try: `python bench.py --gpus N --exchange none` (no data-path collective: per-rank kernel-only rates), "
                 "NCCL_DEBUG=INFO, HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC is the only mode this driver supports)")
    return "\n".join(lines)


class BringUpWatchdog:
    """Bounds rank bring-up: if the guarded region (init_process_group + the first collective) is still running after
    `seconds`, the rank prints its diagnostics and the PROCESS exits with code 3 -- torchrun then tears the other ranks down
    -- instead of sitting in the default 10-minute store / communicator time-outs."""

    def __init__(self, seconds: float, rank, local_rank, what: str):
        self.t = threading.Timer(seconds, self._fire)
        self.t.daemon = True
        self.args = (rank, local_rank, what, seconds)

    def _fire(self):
        rank, local_rank, what, seconds = self.args
        sys.stderr.write(rank_diagnostics(rank, local_rank, f"ERROR: {what} did not finish within {seconds:.0f} s") + "\n")
        sys.stderr.flush()
        os._exit(3)

    def __enter__(self):
        self.t.start()
        return self

    def __exit__(self, *exc):
        self.t.cancel()
        return False


def bring_up_process_group(torch, dist, backend: str, rank: int, local_rank: int, timeout_s: float):
    """init_process_group with a bounded time-out, then ONE collective on the transport the exchange will use, all under the
    watchdog.  Any failure: diagnostics on stderr, exit code 3."""
    from datetime import timedelta

    try:
        with BringUpWatchdog(timeout_s + 30.0, rank, local_rank, f"bring-up of the {backend} process group"):
            if backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank),
                                        timeout=timedelta(seconds=timeout_s))
                probe = torch.ones(1, device=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend="gloo", timeout=timedelta(seconds=timeout_s))
                probe = torch.ones(1)
            dist.all_reduce(probe)  # the first collective creates the communicator: a broken transport fails HERE
            if probe.is_cuda:
                torch.cuda.synchronize()
            if int(probe.item()) != dist.get_world_size():
                raise RuntimeError(f"first all_reduce returned {probe.item()} for a world of {dist.get_world_size()}")
    except SystemExit:
        raise
    except BaseException as exc:
        sys.stderr.write(rank_diagnostics(rank, local_rank, f"ERROR: could not bring up the {backend} process group", exc) + "\n")
        sys.stderr.flush()
        os._exit(3)


def self_launch(args, argv) -> int:
    """`python bench.py --gpus N` without torchrun: N ranks all the same.  Re-executes this file under
    torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1 at a free port) and returns its exit code.
    Fewer than N visible devices is an error (exit code 2), not a one-GPU run -- unless the ranks are allowed to share
    devices (--backend gloo, the test transport)."""
    import socket

    import torch

    if not torch.cuda.is_available():
        print("bench.py needs a GPU: the solve path has no CPU fallback", file=sys.stderr)
        return 2
    ndev = torch.cuda.device_count()
    if ndev < args.gpus and args.backend != "gloo":
        print(f"[bench] ERROR: --gpus {args.gpus} but only {ndev} GPU(s) visible (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES?): "
              f"refusing to report an n_gpus={args.gpus} line from fewer devices.  (--backend gloo is the TEST transport that "
              "lets ranks share a device.)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HMPC_BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs across processes on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print("[bench] --gpus %d without torchrun: launching %s" % (args.gpus, " ".join(cmd)), file=sys.stderr)
    return subprocess.call(cmd, env=env)


def main() -> None:
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
        cpu_worker(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6] if len(sys.argv) > 6 else "reference",
                   int(sys.argv[7]) if len(sys.argv) > 7 else 1)
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks = GPUs of the job.  Default: WORLD_SIZE when launched under torchrun, else 1")
    ap.add_argument("--windows", type=int, default=5,
                    help="the timed region (K steps between barrier + synchronize) is repeated this many times; value / ms_per_step "
                         "are the median window's, every window is in the line")
    ap.add_argument("--bringup-timeout", type=float, default=120.0,
                    help="seconds a rank may take to bring up its process group and first collective before it prints its "
                         "diagnostics and exits 3")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8192, help="MPC instances per GPU per step")
    ap.add_argument("--horizon", type=int, default=10)
    ap.add_argument("--gait", default="standing", help="standing = the metric's 2-contact case; walking = BASELINE config 3's sweep")
    ap.add_argument("--contacts", type=int, default=2, choices=[2, 3],
                    help="3 = BASELINE config 5, the loco-manipulation extension (two feet + hand, 180 x 240 QPs; its 4-GPU "
                         "split is --gpus 4 --batch 2048 --contacts 3)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the exchange.  nccl (= RCCL) is the product transport.  gloo is a TEST "
                         "transport: the packed wrench block is staged through host memory and ranks may share a GPU "
                         "(rank r runs on device r %% device_count) -- it lets the N>1 code path run with two ranks on a one-GPU box")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-configs", action="store_true",
                    help="skip the other_configs side measurements (profiling runs: only the headline kernel launches)")
    ap.add_argument("--streams", type=int, default=2, help="launch streams used alternately by consecutive steps (1..4; 2 measured best but for +0.8 %% at 3)")
    ap.add_argument("--exchange", default="wrench", choices=["wrench", "full", "none"],
                    help="N>1: what the ranks all_gather per solve (wrench = step-0 wrench + status, SURVEY 8e).  none = a DIAGNOSTIC "
                         "mode without any data-path collective: the control plane (barriers, max over ranks) runs on gloo, the line "
                         "carries every rank's kernel-only rate -- what to run when the RCCL exchange of the default mode fails")
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the N>1 code path (process group, posted all_gather per solve, stream ordering) with whatever "
                         "WORLD_SIZE is -- also 1: what the one-GPU test of the torchrun path uses")
    ap.add_argument("--cpu-per-core", type=int, default=384)
    ap.add_argument("--cpu-seconds", type=float, default=10.0,
                    help="wall time of each of the three CPU-baseline runs at P = cores (the quoted value is their median); the "
                         "P-sweep points run 0.3x this")
    ap.add_argument("--path", default="solve", choices=["solve", "builder"],
                    help="solve = the metric (default); builder = rows f1-f3 only (record builder + wrench kernels)")
    ap.add_argument("--check", type=int, default=256, help="instances checked against the oracle after the timed region")
    args = ap.parse_args()
    gpus_given = args.gpus is not None
    if not gpus_given:  # (an external `torchrun --nproc-per-node N bench.py` without --gpus runs N ranks)
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    args.windows = max(1, args.windows)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started without torchrun: N ranks all the same (never a silent one-GPU run that prints n_gpus = 1)
        raise SystemExit(self_launch(args, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    from hector_simulation_amd import interface, records, sharding, synthetic

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the solve path has no CPU fallback")
    if gpus_given and args.gpus != world:
        # launched under torchrun with another rank count than --gpus asks for: a line whose n_gpus contradicts its command
        raise SystemExit(f"[bench] ERROR: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                         "(or plain `python bench.py --gpus N`, which starts the N ranks itself)")
    ndev = torch.cuda.device_count()
    # control-plane / exchange backend: RCCL (product), gloo (TEST transport), or -- --exchange none -- gloo for the barriers only
    ctl_backend = "gloo" if args.exchange == "none" else args.backend
    if args.backend == "gloo":
        local_rank = local_rank % ndev  # test transport: ranks may share a device
    elif local_rank >= ndev:
        sys.stderr.write(rank_diagnostics(rank, local_rank, f"ERROR: no GPU for this rank: {ndev} device(s) visible for {world} ranks -- "
                                          "RCCL needs one device per rank (--backend gloo is the device-sharing TEST transport)") + "\n")
        raise SystemExit(2)
    torch.cuda.set_device(local_rank)
    if world > 1 or args.force_exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:  # stand-alone --force-exchange run: any free port (torchrun sets its own)
            import socket

            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        bring_up_process_group(torch, dist, ctl_backend, rank, local_rank, args.bringup_timeout)
    n_gpus = world

    h, B, nc = args.horizon, args.batch, args.contacts
    W = 6 * nc  # forces per horizon step = width of the step-0 wrench the exchange carries
    if args.path == "builder":
        return bench_builder(args, torch, local_rank)
    # every rank owns the contiguous shard [rank*B, (rank+1)*B) of the global batch (seed offset by rank)
    assert sharding.shard_bounds(world * B, world, rank) == (rank * B, (rank + 1) * B)
    fields, rec = bench_shard(rank, B, h, args.gait, nc)
    n_red = 6 * int(np.asarray(fields["gait"]).reshape(B, -1).sum(axis=1).max())

    dev = torch.device("cuda", local_rank)
    d_rec = torch.from_numpy(rec).to(dev)                      # inputs resident in HBM before the timed region
    # ... and the SAME instances one 5 ms MPC tick later (synthetic.advance_tick: body moved, velocities drifted, feet stayed):
    # every handle alternates between the two tick batches, so that whatever the library carries from one solve to the next --
    # by default the dispatch order, hmpc_set_dispatch_order: longest previous solve first -- is one tick old, as in an MPC
    # loop, and never the exact answer to the same data
    rec_next = records.pack_records(synthetic.advance_tick(fields, h, seed=7 + 1000 * rank), h, nc)
    d_rec_next = torch.from_numpy(rec_next).to(dev)
    # Two handles on two streams, each with its own output block, used alternately: the tail of one step's launch (the last,
    # partly filled round of workgroups) and the launch gap overlap the head of the next step's.  Every step still is one
    # complete pass of the hot path over the whole batch; --streams 1 times strictly back-to-back launches on one stream.
    nstream = max(1, min(4, args.streams))
    if world > 1 or args.force_exchange:
        nstream = min(nstream, 2)  # the posted exchange is double buffered: one slot per launch stream
    streams = [torch.cuda.Stream(device=dev) for _ in range(nstream)]
    d_forces_l = [torch.zeros((B, W * h), dtype=torch.float32, device=dev) for _ in range(nstream)]
    d_status_l = [torch.zeros((B,), dtype=torch.int32, device=dev) for _ in range(nstream)]
    mpcs = []
    for k in range(nstream):
        m = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, B, device=local_rank, contacts=nc)
        m.set_device_records(d_rec.data_ptr(), B, max_reduced_vars=n_red, keepalive=d_rec)
        m.set_device_outputs(d_forces_l[k].data_ptr(), d_status_l[k].data_ptr(), keepalive=(d_forces_l[k], d_status_l[k]))
        mpcs.append(m)
    mpc, d_forces, d_status = mpcs[0], d_forces_l[0], d_status_l[0]
    stream = streams[0].cuda_stream
    torch.cuda.synchronize()

    # the path's only exchange: all_gather of the step-0 wrenches + status words, posted after every
    # solve on the communicator's stream so that it overlaps the next solve; --exchange full gathers all 12h forces
    # synchronously instead
    xch = sharding.WrenchExchange(B, W, dev, always_collective=args.force_exchange) \
        if ((world > 1 or args.force_exchange) and args.exchange == "wrench") else None
    nstep = [0]
    uses = [0] * nstream

    def solve_on(k):  # handle k's next solve: the other tick batch than its last one
        d_in = d_rec_next if uses[k] % 2 else d_rec
        uses[k] += 1
        mpcs[k].set_device_records(d_in.data_ptr(), B, max_reduced_vars=n_red, keepalive=(d_rec, d_rec_next))
        mpcs[k].solve(streams[k].cuda_stream)

    def step():
        k = nstep[0] % nstream
        with torch.cuda.stream(streams[k]):
            solve_on(k)
            if xch is not None:
                xch.post(k, d_forces_l[k], d_status_l[k])  # slot = launch stream: never reused while that stream's step owns it
            elif world > 1 and args.exchange == "full":
                sharding.gather_forces(d_forces_l[k], world * B)
        nstep[0] += 1

    def fence():
        if xch is not None:
            xch.wait_all()   # every posted exchange is complete inside the timed region
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # the timed region: EXACTLY K steps between barrier + synchronize on both sides, max over ranks -- repeated `--windows` times;
    # the line's value is the median window's, every window is reported
    window_s = []
    for _ in range(args.windows):
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev if ctl_backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        window_s.append(el)
    elapsed = float(np.median(window_s))
    if args.windows % 2 == 0:  # (the median of an even count is no window's own time: take the upper middle one)
        elapsed = sorted(window_s)[args.windows // 2]
    ms_per_step = 1e3 * elapsed / args.steps

    exchange_check = None
    if xch is not None:  # what the last posted exchange delivered == this rank's own slice of the last step's outputs
        klast = (nstep[0] - 1) % nstream
        gw, gs = xch.result(klast)
        mine = slice(rank * B, (rank + 1) * B)
        exchange_check = bool(torch.equal(gw[mine], d_forces_l[klast][:, :W]) and torch.equal(gs[mine], d_status_l[klast].view(torch.int32)))
        if world > 1:
            # ... and the other ranks' slices are what THEY computed: every rank's last-step wrench, gathered once more
            # through the plain (synchronous) path, must equal the posted exchange's block bit for bit
            ref_all = sharding.gather_forces(d_forces_l[klast][:, :W].contiguous(), world * B)
            exchange_check = exchange_check and bool(torch.equal(gw, ref_all))
    # the same K passes strictly back to back on ONE stream (no overlap of a launch's tail with the next launch's head),
    # reported beside the headline value
    torch.cuda.synchronize()
    ts0 = time.perf_counter()
    with torch.cuda.stream(streams[0]):
        for _ in range(args.steps):
            solve_on(0)
    torch.cuda.synchronize()
    single_stream_s = time.perf_counter() - ts0

    # the headline loop once more in natural dispatch order (hmpc_set_dispatch_order(0)): what the ordering is worth
    natural_s = predicted_s = None
    if xch is None and world == 1:
        for m in mpcs:
            m.set_dispatch_order(False)
        for _ in range(max(2, args.warmup)):
            step()
        torch.cuda.synchronize()
        tn0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        natural_s = time.perf_counter() - tn0
        # ... and ordered by the record-only cost predictor (mode 2: what a cold handle gets)
        for m in mpcs:
            m.set_dispatch_order(2)
        for _ in range(max(2, args.warmup)):
            step()
        torch.cuda.synchronize()
        tp0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        predicted_s = time.perf_counter() - tp0
        for m in mpcs:
            m.set_dispatch_order(1)
        for _ in range(2):   # (back to the default mode: the next solves are ordered by a previous one again)
            step()
        torch.cuda.synchronize()

    # dominant kernel's own duration: HIP events on the launch stream, kernel launches only (no collective); tick batch `rec`
    # ordered by the solve of the other tick batch before it, as in the timed loop
    mpc.set_device_records(d_rec_next.data_ptr(), B, max_reduced_vars=n_red, keepalive=(d_rec, d_rec_next))
    mpc.solve(stream)
    mpc.set_device_records(d_rec.data_ptr(), B, max_reduced_vars=n_red, keepalive=(d_rec, d_rec_next))
    mpc.solve(stream)
    torch.cuda.synchronize()
    kernel_ms = mpc.time_solve(max(5, args.steps), stream)   # (leaves handle 0's outputs = the solution of `rec`: checked below)

    status = d_status.cpu().numpy().astype(np.uint32)
    n_fail = int((interface.status_code(status) != 0).sum())
    iters = interface.status_iters(status)
    nact = interface.status_nactive(status)

    # every rank checks ITS OWN shard against the oracle; the line carries the worst over the ranks
    parity = None
    mine_summary = {"parity": parity_of_rank(args, rank, rec, fields, h, nc, mpc, d_forces, status) if args.check > 0 else None,
                    "exchange_ok": exchange_check, "failed": n_fail, "kernel_ms": kernel_ms,
                    "device": device_description(torch, local_rank), "iters_mean": float(iters.mean()), "iters_max": int(iters.max())}
    summaries = [mine_summary]
    if world > 1:
        summaries = [None] * world
        dist.all_gather_object(summaries, mine_summary)
    if args.check > 0:
        parity = merge_parity([sm["parity"] for sm in summaries])
    if exchange_check is not None:
        exchange_check = all(bool(sm["exchange_ok"]) for sm in summaries)
    n_fail_all = sum(sm["failed"] for sm in summaries)

    if rank == 0:
        total = world * B * args.steps
        value = total / elapsed
        if nc == 2:
            bps = BYTES_PER_SOLVE.get(h, (54 + 12 * h) * 4 + 2 * h + 48 * h + 4)
            mfl = MFLOP_PER_SOLVE.get(h, 2 * (12 * h) ** 2 * (13 * h) / 1e6)
        else:  # extension record (SURVEY.md 8d's formulas with 18 inputs per step): (73+12h)*4 + 3h in, 18h*4 + 4 out
            bps = (73 + 12 * h) * 4 + 3 * h + W * h * 4 + 4
            mfl = 2 * (W * h) ** 2 * (13 * h) / 1e6
        ach_gbs = B * bps / (kernel_ms * 1e-3) / 1e9
        ach_tf = B * mfl * 1e6 / (kernel_ms * 1e-3) / 1e12
        # HBM traffic and issue counters come from rocprofv3 --pmc passes, which cannot run inside this process; the
        # committed summary is used ONLY when it was taken on exactly this build of the library (source hash) and this
        # workload -- otherwise the keys are null rather than stale
        traffic = counters = valu_issue_frac = None
        traffic_note = "no committed PMC profile for this build"
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                from hector_simulation_amd import build as hip_build

                tj = json.load(open(tpath))
                if tj.get("source_hash") != hip_build.source_hash():
                    traffic_note = "profiles/hbm_traffic.json was taken on another build of the kernel: ignored"
                elif tj.get("horizon") == h and tj.get("gait") == args.gait:
                    traffic = tj["bytes_per_solve"] * B
                    counters = tj.get("counters")
                    valu_issue_frac = (counters or {}).get("valu_issue_frac")
                    traffic_note = f"rocprofv3 --pmc passes of {tj.get('profile_dir', 'profiles/')} on this build (source hash matches)"
            except Exception:
                traffic = None
        # useful binary64 work of one solve: the inverse by n symmetric sweeps over the upper
        # triangle, n^2 (n+1) flop, plus per active-set iteration the products z = M w and r = E d, ~4 n^2 flop
        it_mean = float(iters.mean())
        fp64_flop = n_red * n_red * (n_red + 1) + it_mean * 4.0 * n_red * n_red
        fp64_tf = B * fp64_flop / (kernel_ms * 1e-3) / 1e12
        out = {
            "metric": "MPC QP solves/sec (horizon=10, 2 contacts)" if (h == 10 and args.gait == "standing" and nc == 2)
                      else f"MPC QP solves/sec (horizon={h}, gait={args.gait}, {nc} contacts)",
            "value": value, "unit": "QP solves/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 assembly / f64 solve", "data": "synthetic",
            "config": {"workload": (f"{'2-contact standing' if args.gait == 'standing' else args.gait} randomized MPC ticks, " if nc == 2 else
                                    f"3-contact (two feet + hand, BASELINE config 5 extension) randomized MPC ticks, feet {args.gait}, ") +
                                   f"horizon {h}, reduced QP up to {n_red}x{n_red // 6 * 8}; records and forces device-resident in/out; "
                                   "two batches of the same instances one 5 ms tick apart, solved alternately",
                       "batch_per_gpu": B, "global_batch": world * B, "horizon": h, "contacts": nc,
                       "exchange_backend": (("none (--exchange none: NO data-path collective; barriers and the max over ranks on gloo)" if args.exchange == "none" else
                                             args.backend + (" (TEST transport: host-staged, ranks may share a GPU)" if args.backend == "gloo" else " (RCCL)"))
                                            if (world > 1 or args.force_exchange) else None),
                       "launch_streams": nstream,
                       "world": world, "launcher": ("bench.py re-executed itself under torch.distributed.run (--gpus N without torchrun)"
                                                    if os.environ.get("HMPC_BENCH_SELF_LAUNCHED") else
                                                    ("torch.distributed.run (external)" if "TORCHELASTIC_RUN_ID" in os.environ or world > 1 else "python (single process)")),
                       "devices_visible": ndev, "rank_devices": [sm["device"] for sm in summaries],
                       "rccl_version": rccl_version(torch) if (world > 1 or args.force_exchange) else None,
                       "parallelism": ((f"batch shards x{world}, all_gather of "
                                        f"{'step-0 wrench + status (overlapped with the next solve)' if xch is not None else 'all forces'}")
                                       if args.exchange != "none" else
                                       f"batch shards x{world}, NO exchange (diagnostic mode: every rank solves its shard, nothing is gathered)")
                       if world > 1 else ("single GPU" + (", exchange code path forced on (group of one)" if xch is not None else "")),
                       **({"exchange_selfcheck_ok": exchange_check} if exchange_check is not None else {})},
            "roofline": {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
                         "pmc_counters": counters,
                         "kernel": "hmpc_kernel (fused assembly + QP solve)", "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_solve": bps,
                         "note": "neither HBM nor MFMA binds this path (SURVEY.md 8d): the limiter is the serial "
                                 "active-set iteration inside one workgroup (LDS/VALU fp64 latency)"},
            "roofline_mfma": {"bound": "mfma", "achieved": ach_tf, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                              "frac": ach_tf / MFMA_F32_PEAK_TF, "algorithmic_mflop_per_solve": mfl},
            "windows": {"n": args.windows, "steps_each": args.steps,
                        "values": [world * B * args.steps / w for w in window_s],
                        "value_min": world * B * args.steps / max(window_s), "value_max": world * B * args.steps / min(window_s),
                        "spread": (max(window_s) - min(window_s)) / elapsed,
                        "note": "the timed region (K steps between barrier + synchronize, max over ranks) repeated n times back to "
                                "back; value / ms_per_step are the median window's"},
            "single_stream": {"value": B * args.steps / single_stream_s, "ms_per_step": 1e3 * single_stream_s / args.steps,
                              "note": "this rank's K passes launched back to back on one stream, no exchange; the headline value "
                                      "alternates two streams (config.launch_streams) so that the partly filled last round of "
                                      "workgroups of one launch overlaps the next launch"},
            "dispatch_order": {
                "mode": "longest previous solve first (hmpc_set_dispatch_order, the library's default for 512 < batch <= 32768)",
                "hint": "each handle's solve is ordered on the device by the iteration counts of ITS previous solve, which was of "
                        "the same instances one tick apart (the other of the two tick batches) -- never of the same data; the sort "
                        "(one small launch) is inside the timed region; results do not depend on the order",
                "natural_order": None if natural_s is None else {"value": world * B * args.steps / natural_s,
                                                                 "ms_per_step": 1e3 * natural_s / args.steps},
                "predicted_order": None if predicted_s is None else {
                    "value": world * B * args.steps / predicted_s, "ms_per_step": 1e3 * predicted_s / args.steps,
                    "what": "hmpc_set_dispatch_order(2): every solve ordered by the cost predicted from its records alone "
                            "(what a cold handle gets; hmpc_builder.h predicted_cost_bucket)"}},
            "fp64_valu_frac": fp64_tf / FP64_VALU_PEAK_TF,
            "fp64_valu": {"achieved": fp64_tf, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s",
                          "useful_flop_per_solve": fp64_flop,
                          "formula": "n^2 (n+1) [inverse by symmetric sweeps] + iterations_mean * 4 n^2 [z = M w, r = E d]"},
            "valu_issue_frac": valu_issue_frac,
            "iterations_per_solve": it_mean,
            "solver": {"failed": n_fail, "failed_over_all_ranks": n_fail_all,
                       "kernel_ms_per_rank": [sm["kernel_ms"] for sm in summaries],
                       "iters_mean_per_rank": [sm["iters_mean"] for sm in summaries], "iters_max_per_rank": [sm["iters_max"] for sm in summaries], "iters_median": float(np.median(iters)), "iters_max": int(iters.max()), "active_median": float(np.median(nact)), "active_max": int(nact.max()),
                       "kernel_solves_per_s": B / (kernel_ms * 1e-3)},
        }
        # the other BASELINE.json shapes on the same kernel family, a few launches each (not the headline value)
        extra = {}
        try:
            if args.no_side_configs or world > 1 or nc != 2:  # (side measurements belong to the default single-GPU line)
                raise StopIteration
            # Every side config in the three dispatch orders (hmpc_set_dispatch_order), with ONE protocol: solve tick k, then time
            # ONE solve of the same instances one 5 ms tick later (best of 4) -- never a batch ordered by the answer to its own data:
            #   natural   = mode 0;  predicted = mode 2 (what a COLD handle gets: cost predicted from the records alone);
            #   next_tick = mode 1 (the default in an MPC loop: ordered by the previous tick's iteration counts)
            def three_orders(fields2, hh, bb, ncs):
                rec_a = records.pack_records(fields2, hh, ncs)
                rec_b = records.pack_records(synthetic.advance_tick(fields2, hh, seed=9), hh, ncs)
                res2 = {}
                for oname, mode in (("natural", 0), ("predicted", 2), ("next_tick", 1)):
                    mm = interface.BatchedMPC(synthetic.DT_MPC, hh, synthetic.F_MAX, bb, device=local_rank, contacts=ncs)
                    mm.set_dispatch_order(mode)
                    ts2 = []
                    for _ in range(4):
                        mm.upload(rec_a)
                        mm.solve(stream)
                        torch.cuda.synchronize()
                        mm.upload(rec_b)
                        ts2.append(mm.time_solve(1, stream))
                    _, stq = mm.download()
                    mm.close()
                    res2[oname] = {"solves_per_s": bb / (min(ts2) * 1e-3), "kernel_ms": min(ts2)}
                    res2["failed"] = int((interface.status_code(stq) != 0).sum())
                    res2["iters_mean"] = float(interface.status_iters(stq).mean())
                res2["solves_per_s"] = res2["next_tick"]["solves_per_s"]   # (the order an MPC loop runs in)
                res2["kernel_ms"] = res2["next_tick"]["kernel_ms"]
                return res2

            for name, gait2, hh, bb in (("cfg2_walking_b1024_fixed_phase", "walking", 10, 1024),
                                        ("metric_2contact_b1024", "standing", 10, 1024),
                                        ("cfg3_walking_sweep_b8192_per_gpu", "walking", 10, 8192),
                                        ("cfg4_h20_single_support_b4096", "single", 20, 4096),
                                        ("h20_double_support_240x320_b2048_wide_variant", "standing", 20, 2048)):
                f2 = synthetic.make_batch(bb, hh, gait2, seed=2, phase=(0 if "fixed" in name else "random"))
                extra[name] = three_orders(f2, hh, bb, 2)
            # BASELINE configs[4]: two feet + hand, 180 variables x 240 rows (the three-contact extension)
            for name, bb in (("cfg5_3contact_180x240_b2048_per_gpu", 2048), ("cfg5_3contact_180x240_b8192", 8192)):
                extra[name] = three_orders(synthetic.make_batch3(bb, 10, "standing", seed=5, hand="contact"), 10, bb, 3)
            # OFF-NOMINAL INPUT RANGES: the metric's workload (2-contact standing, h = 10, this batch size)
            # with attitude / velocity / angular-velocity / joint / command ranges at 1x, 3x and 6x SURVEY 8d's
            # (synthetic.hard_batch), device-resident, INCLUDING the device-side repair of whatever the fast variant flags
            # (hmpc_set_device_repair: continuation of overflowed working sets + safe pass, no host round trip).  Same protocol as
            # above: solve tick k, time ONE solve of tick k + 1, best of 4.
            def range_scale(scale):
                from oracle import oracle_py

                fs = synthetic.hard_batch(B, h, "standing", 17, scale)
                rec_a = records.pack_records(fs, h)
                rec_b = records.pack_records(synthetic.advance_tick(fs, h, seed=9), h)
                m0 = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, B, device=local_rank)
                m0.set_auto_resolve(False)      # what the fast pass alone leaves flagged
                m0.upload(rec_b)
                m0.solve(stream)
                torch.cuda.synchronize()
                t_fast = m0.time_solve(1, stream)
                _, st0 = m0.download()
                m0.close()
                c0 = interface.status_code(st0)
                def device_pipeline(mode):
                    m1 = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, B, device=local_rank)
                    m1.set_auto_resolve(False)
                    m1.set_device_repair(mode)
                    tsx = []
                    for _ in range(4):
                        m1.upload(rec_a)
                        m1.solve(stream)
                        torch.cuda.synchronize()
                        m1.upload(rec_b)
                        tsx.append(m1.time_solve(1, stream))
                    fx, stx = m1.download()          # (auto-resolve off: exactly what the device-side passes left)
                    m1.close()
                    return tsx, fx, stx

                ts3, f1, st1 = device_pipeline(1)    # fast -> continuation -> cold safe pass, all on the device
                ts2, _, st2 = device_pipeline(2)     # fast -> continuation; what is left stays flagged for hmpc_download

                def two_stream_throughput(mode, nsteps=8):
                    """the headline's protocol on this workload: two handles on two streams, the two tick batches alternately, K whole
                    solves (every repair launch included) back to back -- the tail of one solve's safe pass (a few dozen workgroups)
                    runs under the next solve's fast pass instead of idling the chip"""
                    hs2, d_a, d_b = [], torch.from_numpy(rec_a).to(dev), torch.from_numpy(rec_b).to(dev)
                    for k2 in range(2):
                        mm = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, B, device=local_rank)
                        mm.set_auto_resolve(False)
                        mm.set_device_repair(mode)
                        hs2.append(mm)

                    def run(nrun):
                        for i2 in range(nrun):
                            k2 = i2 % 2
                            d_in = d_b if (i2 // 2) % 2 else d_a
                            hs2[k2].set_device_records(d_in.data_ptr(), B, max_reduced_vars=120, keepalive=(d_a, d_b))
                            hs2[k2].solve(streams[k2 % len(streams)].cuda_stream)
                        torch.cuda.synchronize()

                    run(4)
                    t20 = time.perf_counter()
                    run(nsteps)
                    t2s = time.perf_counter() - t20
                    for mm in hs2:
                        mm.close()
                    return B * nsteps / t2s

                thr1 = two_stream_throughput(1)
                c1 = interface.status_code(st1)
                flagged = np.flatnonzero(c0 != 0)
                others = np.flatnonzero(c0 == 0)
                idx = np.concatenate([flagged[:32], others[:64 - min(32, flagged.size)]])
                refq = oracle_py.solve_records(np.ascontiguousarray(rec_b[idx]), h, synthetic.DT_MPC, synthetic.F_MAX, first=0, count=idx.size)
                qq = refq["q_soln"]
                ee = np.abs(f1[idx].astype(np.float64) - qq).max(axis=1) / np.maximum(1.0, np.abs(qq).max(axis=1))
                okc = (c1[idx] == 0)
                return {"scale": scale, "solves_per_s": B / (min(ts3) * 1e-3), "kernel_ms": min(ts3),
                        "fast_pass_only_kernel_ms": t_fast,
                        "flagged_fraction_fast_pass": float((c0 != 0).mean()),
                        "flag_codes_fast_pass": {interface.STATUS_NAMES[int(k)]: int(v) for k, v in zip(*np.unique(c0, return_counts=True))},
                        "not_ok_after_device_repair": int(((c1 != 0) & (c1 != 6)).sum()),
                        "two_stream_throughput": {"solves_per_s": thr1,
                                                  "what": "the headline's protocol on this workload (two handles on two streams, 8 whole solves incl. every "
                                                          "repair launch, wall clock): the tail of one solve's safe pass runs under the next solve"},
                        "continuation_only": {"solves_per_s": B / (min(ts2) * 1e-3), "kernel_ms": min(ts2),
                                              "left_flagged_for_the_host": int((interface.status_code(st2) != 0).sum()),
                                              "what": "hmpc_set_device_repair(2): the continuation pass only; the cold safe pass of the few "
                                                      "instances it does not finish (hundreds of iterations on ONE workgroup each: "
                                                      "milliseconds at the tail of the stream) is left to hmpc_download"},
                        "iters_mean": float(interface.status_iters(st1).mean()), "iters_max": int(interface.status_iters(st1).max()),
                        "active_max": int(interface.status_nactive(st1).max()),
                        "checked_vs_qpoases": int(idx.size), "checked_flagged": int(min(32, flagged.size)),
                        "max_rel_force_err_vs_qpoases": float(ee[okc].max()) if okc.any() else None,
                        "qpoases_failed_in_checked": int(refq["n_bad"])}

            # COMMAND SWEEPS: 128 robot states x 64 velocity / yaw-rate commands = 8 192 instances, the
            # trajectories built from the commands as ConvexMPCLocomotion.cpp:351-406 builds them; H and its inverse formed once per
            # state (hmpc_solve_command_sweep) against the same records as 8 192 independent instances (hmpc_solve) -- bit-identical
            # outputs asserted, kernel times by HIP events, best of 4
            def command_sweep(groups, kk, gait2, floors=0):
                basef = synthetic.make_batch(groups, h, gait2, seed=12, phase="random")
                fs = {key: np.repeat(np.asarray(v), kk, axis=0) for key, v in basef.items()}
                rng2 = np.random.default_rng(13)
                bb = groups * kk
                vx, vy, yr = rng2.uniform(-0.5, 0.5, bb), rng2.uniform(-0.2, 0.2, bb), rng2.uniform(-0.3, 0.3, bb)
                tr = fs["traj"].reshape(bb, h, 12).copy()
                stp = np.arange(h)[None, :]
                tr[:, :, 9], tr[:, :, 10], tr[:, :, 8] = vx[:, None], vy[:, None], yr[:, None]
                tr[:, :, 3] = fs["p"][:, 0:1] + stp * synthetic.DT_MPC * vx[:, None]
                tr[:, :, 4] = fs["p"][:, 1:2] + stp * synthetic.DT_MPC * vy[:, None]
                tr[:, 1:, 2] = tr[:, 0:1, 2] + stp[:, 1:] * synthetic.DT_MPC * yr[:, None]
                fs["traj"] = tr.reshape(bb, 12 * h)
                recs = records.pack_records(fs, h)
                d_mu = None
                if floors:  # terrain sweep: a friction parameter per instance (hmpc_set_instance_mu), `floors` values inside every group
                    d_mu = torch.from_numpy(np.array([0.8, 1.25, 2.0, 3.0], dtype=np.float32)[np.arange(bb) % floors]).to(dev)
                mi = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, bb, device=local_rank)
                if d_mu is not None:
                    mi.set_instance_mu(d_mu.data_ptr(), keepalive=d_mu)
                mi.upload(recs)
                mi.solve(stream)
                fi, si = mi.download()
                t_ind = min(mi.time_solve(1, stream) for _ in range(4))
                mi.close()
                ms = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, bb, device=local_rank)
                if d_mu is not None:
                    ms.set_instance_mu(d_mu.data_ptr(), keepalive=d_mu)
                ms.upload(recs)
                ts4 = []
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for _ in range(5):
                    with torch.cuda.stream(streams[0]):
                        ev0.record()
                        ms.solve_command_sweep(kk, streams[0].cuda_stream)
                        ev1.record()
                    torch.cuda.synchronize()
                    ts4.append(ev0.elapsed_time(ev1))
                fsw, ssw = ms.download()
                ms.close()
                return {"states": groups, "commands_per_state": kk, "instances": bb, "gait": gait2,
                        **({"floors_per_state": floors, "commands_per_floor": kk // floors} if floors else {}),
                        "independent": {"solves_per_s": bb / (t_ind * 1e-3), "kernel_ms": t_ind},
                        "sweep": {"solves_per_s": bb / (min(ts4[1:]) * 1e-3), "kernel_ms": min(ts4[1:]),
                                  "note": "both launches (one workgroup per state forms H^-1, one per instance solves with it), torch events on the launch stream"},
                        "speedup": t_ind / min(ts4[1:]),
                        "bit_identical_forces": bool(np.array_equal(fsw.view(np.uint32), fi.view(np.uint32))),
                        "identical_status_words": bool(np.array_equal(ssw, si)),
                        "failed": int((interface.status_code(ssw) != 0).sum())}

            extra["command_sweep_128_states_x_64_commands"] = command_sweep(128, 64, "standing")
            extra["command_sweep_1024_states_x_8_commands"] = command_sweep(1024, 8, "standing")
            extra["terrain_command_sweep_128_states_x_4_floors_x_16_commands"] = command_sweep(128, 64, "standing", floors=4)
            extra["command_sweep_128_states_x_64_commands_walking"] = command_sweep(128, 64, "walking")
            rs = {f"range_scale_{sc}": range_scale(sc) for sc in (1, 3, 6)}
            for sc in (3, 6):
                rs[f"range_scale_{sc}"]["fraction_of_range_scale_1"] = rs[f"range_scale_{sc}"]["solves_per_s"] / rs["range_scale_1"]["solves_per_s"]
                rs[f"range_scale_{sc}"]["two_stream_throughput"]["fraction_of_range_scale_1"] = \
                    rs[f"range_scale_{sc}"]["two_stream_throughput"]["solves_per_s"] / rs["range_scale_1"]["two_stream_throughput"]["solves_per_s"]
                rs[f"range_scale_{sc}"]["continuation_only"]["fraction_of_range_scale_1"] = \
                    rs[f"range_scale_{sc}"]["continuation_only"]["solves_per_s"] / rs["range_scale_1"]["solves_per_s"]
            extra.update(rs)
            extra["range_scale_note"] = ("2-contact standing h=10 instances with the SURVEY 8d input ranges multiplied by `scale` "
                                         "(synthetic.hard_batch, seed 17, random gait phase, yaw-rate command), this batch size, ONE "
                                         "device-resident solve of tick k+1 after tick k (best of 4) with hmpc_set_device_repair on: the "
                                         "time includes every launch the library enqueues -- dispatch-order sort, fast variant, the "
                                         "continuation of instances whose working set outgrew the fast variant's 64 rows, the safe pass")
            extra["dispatch_order_note"] = ("every side config: solve tick k, then ONE timed solve of the same instances one 5 ms tick later "
                                            "(best of 4), in three dispatch orders -- `natural` (hmpc_set_dispatch_order 0), `predicted` (2: "
                                            "cost predicted from the records alone, what a cold handle gets), `next_tick` (1, the default: "
                                            "ordered by the previous tick's iteration counts); `solves_per_s` = next_tick.  Batches of "
                                            "single-support QPs only (walking) are never reordered.")
            # rows f1+f2 -> solve -> f3 as ONE device-resident entry (hmpc_tick_solve_device): tick structs in HBM in, joint
            # torques in HBM out, no host call between the launches; every instance routed on the device to the smallest
            # kernel variant that holds it (walking ticks run on the 60-variable kernel without any host hint)
            for gait2 in ("walking", "standing"):
                tk = synthetic.make_ticks(B, h, gait2, seed=11)
                d_tk = torch.from_numpy(tk.view(np.uint8).reshape(B, -1)).to(dev)
                d_tau = torch.zeros((B, 10), dtype=torch.float64, device=dev)
                d_ff = torch.zeros((B, 12), dtype=torch.float64, device=dev)
                d_wp = torch.zeros((B, 2), dtype=torch.float64, device=dev)
                mp = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, B, device=local_rank)
                one = lambda: mp.tick_solve_device(d_tk.data_ptr(), B, synthetic.DT_MPC, d_tau.data_ptr(), d_ff.data_ptr(),
                                                   d_wp.data_ptr(), stream)
                for _ in range(3):
                    one()
                torch.cuda.synchronize()
                tt0 = time.perf_counter()
                nrep = 20
                for _ in range(nrep):
                    one()
                torch.cuda.synchronize()
                tt = (time.perf_counter() - tt0) / nrep
                _, stt = mp.download()
                io_bytes = tk.dtype.itemsize + 8 * (10 + 12 + 2)          # tick struct in, tau + f_ff + wpd out
                inner_bytes = 2 * (mp.stride + 4 * 12 * h + 4) + 1        # record, forces, status written and read once, class byte
                extra[f"ticks_in_torques_out_b{B}_{gait2}"] = {
                    "ticks_per_s": B / tt, "ms_per_tick_batch": 1e3 * tt, "failed": int((interface.status_code(stt) != 0).sum()),
                    "algorithmic_bytes_per_tick": io_bytes, "hbm_bytes_per_tick_incl_intermediates": io_bytes + inner_bytes,
                    "hbm_gbs_incl_intermediates": B * (io_bytes + inner_bytes) / tt / 1e9,
                    "hbm_frac": B * (io_bytes + inner_bytes) / tt / 1e9 / HBM_PEAK_GBS,
                    "note": "hmpc_tick_solve_device: build_records_kernel (+ size classes) -> hmpc_kernel per size class -> "
                            "leg_torque_kernel on one stream, device-resident in and out"}
                mp.close()
            # warm start across ticks (off in the headline): second tick of a synthetic tick pair, each timed launch
            # starts from the sets the FIRST tick left (the sequence is replayed per repetition)
            rec1 = records.pack_records(synthetic.advance_tick(fields, h, seed=7 + 1000 * rank), h)
            rec0 = rec
            mt = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, B, device=local_rank)
            res = {}
            for mode in ("cold", "tick_warm"):
                mt.set_tick_warm_start(mode == "tick_warm")
                tms = []
                for _ in range(4):
                    mt.upload(rec0)
                    mt.solve(stream)
                    torch.cuda.synchronize()
                    mt.upload(rec1)
                    tms.append(mt.time_solve(1, stream))
                _, stt = mt.download()
                res[mode] = {"solves_per_s": B / (min(tms) * 1e-3), "kernel_ms": min(tms),
                             "iters_mean": float(interface.status_iters(stt).mean()),
                             "failed": int((interface.status_code(stt) != 0).sum())}
            mt.close()
            extra["second_tick_%s_b%d" % (args.gait, B)] = res
            # host records in, host forces out through the blocking batched API (PCIe-inclusive; never the headline value)
            mh = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, B, device=local_rank)
            th = []
            for _ in range(5):
                th0 = time.perf_counter()
                mh.upload(rec)
                mh.solve(stream)
                mh.download()
                th.append(time.perf_counter() - th0)
            mh.close()
            extra["host_in_host_out_b%d" % B] = {"solves_per_s": B / min(th[1:]), "ms": 1e3 * min(th[1:]),
                                                 "note": "hmpc_upload_records + hmpc_solve + hmpc_download, pageable host memory, no overlap"}
            # the same host-to-host work pipelined: two handles on two streams, pinned host buffers, the copies of one
            # batch under the solve of the other (hmpc_upload_records_async / hmpc_download_async)
            nbuf = 2
            streams = [torch.cuda.Stream(device=dev) for _ in range(nbuf)]
            hs = [interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, B, device=local_rank) for _ in range(nbuf)]
            h_rec = [torch.from_numpy(rec.copy()).pin_memory() for _ in range(nbuf)]
            h_for = [torch.empty((B, 12 * h), dtype=torch.float32).pin_memory() for _ in range(nbuf)]
            h_st = [torch.empty((B,), dtype=torch.int32).pin_memory() for _ in range(nbuf)]

            def pipelined(nrounds):
                for r in range(nrounds):
                    k2 = r % nbuf
                    st2 = streams[k2].cuda_stream
                    hs[k2].upload_async(h_rec[k2].data_ptr(), B, st2)
                    hs[k2].solve(st2)
                    hs[k2].download_async(h_for[k2].data_ptr(), h_st[k2].data_ptr(), st2)
                for st3 in streams:
                    st3.synchronize()

            pipelined(4)
            tp0 = time.perf_counter()
            nr = 12
            pipelined(nr)
            tp = time.perf_counter() - tp0
            okp = int((interface.status_code(h_st[0].numpy().astype(np.uint32)) != 0).sum())
            for hh2 in hs:
                hh2.close()
            extra["host_in_host_out_pipelined_b%d" % B] = {"solves_per_s": nr * B / tp, "ms_per_batch": 1e3 * tp / nr,
                                                           "failed": okp,
                                                           "note": "two handles / two streams, pinned host buffers, copies overlapped with the other batch's solve"}
            # the reference's own call sequence for ONE robot (setup_problem / update_problem_data / get_solution,
            # ConvexMPCLocomotion.cpp:410-429): host-to-host latency of a blocking tick through the legacy interface
            row = {k2: np.asarray(v2)[0] for k2, v2 in fields.items()}
            lat = []
            for rep in range(60):
                tl0 = time.perf_counter()
                interface.setup_problem(synthetic.DT_MPC, h, 0.25, synthetic.F_MAX)
                interface.update_problem_data(row["p"], row["v"], row["q"], row["w"], row["r"], row["joint_angles"],
                                              float(row["yaw"]), row["weights"], row["traj"], row["Alpha_K"], row["gait"])
                u0 = [interface.get_solution(i2) for i2 in range(12)]
                lat.append(time.perf_counter() - tl0)
            extra["legacy_single_tick"] = {"median_ms": 1e3 * float(np.median(lat[10:])), "min_ms": 1e3 * float(min(lat[10:])),
                                           "note": "blocking host call incl. H2D record, launch, D2H forces (ctypes overhead included)"}
        except StopIteration:
            extra["skipped"] = True
        except Exception as exc:  # never let the side measurements break the headline line
            extra["error"] = repr(exc)
        out["other_configs"] = extra
        if parity is not None:
            out["parity"] = parity
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(fields, h, args.cpu_per_core, target_s=args.cpu_seconds, sweep_s=0.3 * args.cpu_seconds)
        print(json.dumps(out), flush=True)
    for m in mpcs:
        m.close()
    if world > 1 or args.force_exchange:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
