"""Robustness outside the nominal input distribution: 3x the SURVEY 8d ranges must solve 100 %; at 6x (roll/pitch up
to 0.6 rad, 3 rad/s body rates: far outside what the controller lets happen) working sets outgrow the fast kernel's 80
rows -> the safe pass (hmpc_resolve_failed, capacity = variable count, cold start) must pick those up; instances that
still cycle at a degenerate vertex get the last-resort pass with bounds relaxed by 1e-7 / 1e-6 (status OK_RELAXED, still
within 1e-4 of qpOASES), and whatever is left must be FLAGGED, never silently wrong."""
import numpy as np
import pytest

from hector_simulation_amd import interface, records, synthetic

pytestmark = pytest.mark.gpu


hard_batch = synthetic.hard_batch  # (nb, h, gait, seed, scale): the off-nominal stress rows


@pytest.mark.parametrize("gait,h,scale,min_ok", [("standing", 10, 3, 1.0), ("walking", 10, 3, 1.0), ("single", 20, 3, 1.0),
                                                  ("standing", 10, 6, 1.0), ("single", 20, 6, 1.0), ("walking", 10, 10, 1.0),
                                                  ("standing", 10, 10, 1.0), ("single", 20, 10, 1.0)])
def test_hard_inputs(oracle, gait, h, scale, min_ok):
    """1x .. 10x the nominal input ranges (10x: +-1 rad of tilt, +-5 rad/s): "solved" here <=> solved by qpOASES.  Since round 4
    the safe pass rebuilds E every 48 working-set changes (its round-off used to add up over the 300-470 changes such inputs
    take -- a run ended 3.5 N off its constraints with every multiplier looking fine), the final check also holds the working
    set's rows to their bounds, and a relaxed last-resort pass ends with an exact re-solve on its working set: every instance
    is HMPC_S_OK, exact."""
    nb = 192
    rec = records.pack_records(hard_batch(nb, h, gait, 17, scale), h)
    mpc = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, nb)
    mpc.set_auto_resolve(False)
    mpc.upload(rec)
    mpc.solve()
    _, st_fast = mpc.download()
    n_flagged = int((interface.status_code(st_fast) != 0).sum())
    assert mpc.resolve_failed() == n_flagged
    forces, status = mpc.download()
    mpc.close()
    code = interface.status_code(status)
    ok = (code == 0) | (code == 6)  # 6 = solved with bounds relaxed by <= 2e-6 (hmpc_status_code HMPC_S_OK_RELAXED)
    assert (code == 0).all(), np.unique(code, return_counts=True)  # exact: not even HMPC_S_OK_RELAXED
    assert ok.mean() >= min_ok, (ok.mean(), np.unique(interface.status_code(status), return_counts=True))
    if scale >= 6:
        assert n_flagged > 0  # the regime really exercises the safe pass
        assert (interface.status_code(status) != 5).all()  # ... which cannot run out of working-set room
    ref = oracle.solve_records(rec, h, synthetic.DT_MPC, synthetic.F_MAX)
    q = ref["q_soln"]
    err = np.abs(forces - q).max(axis=1) / np.maximum(1.0, np.abs(q).max(axis=1))
    assert ref["n_bad"] == 0 and err[ok].max() < 1e-4  # everything reported ok matches qpOASES


@pytest.mark.parametrize("gait,h,nc,scale", [("standing", 10, 2, 10), ("walking", 10, 2, 10), ("mixed", 10, 2, 10), ("single", 20, 2, 10),
                                             ("standing", 16, 2, 10), ("standing", 20, 2, 10), ("3contact", 10, 3, 10),
                                             ("standing", 20, 2, 6)])
def test_ok_iff_qpoases_ok_on_the_stress_rows(gait, h, nc, scale):
    """The 10x rows of scripts/stress.py (profiles/r05/stress.txt) as assertions, 256 instances per shape against the qpOASES
    pool: on the set where the reference has an answer at all (its failures masked out and counted separately, VERDICT round 4
    item 7) every instance the kernel reports solved agrees with it, and "ok <=> qpOASES ok" holds."""
    from oracle import pool

    nb = 256
    if nc == 3:
        f = synthetic.make_batch3(nb, 10, "standing", seed=19, phase="random", hand="window")
        g = hard_batch(nb, 10, "standing", 19, scale)
        for k in ("q", "v", "w", "joint_angles", "traj"):
            f[k] = g[k]
    else:
        f = hard_batch(nb, h, gait, 17, scale)
    rec = records.pack_records(f, h, nc)
    mpc = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, nb, contacts=nc)
    mpc.upload(rec)
    mpc.solve()
    forces, status = mpc.download()  # with the safe / last-resort passes
    mpc.close()
    code = interface.status_code(status)
    ok = (code == 0) | (code == 6)
    ref = pool.solve_records_parallel(rec, h, synthetic.DT_MPC, synthetic.F_MAX, nc=nc)
    rbad = np.asarray(ref["bad"], dtype=bool)
    q = ref["q_soln"]
    err = np.abs(forces - q).max(axis=1) / np.maximum(1.0, np.abs(q).max(axis=1))
    both = ok & ~rbad
    assert both.any() and err[both].max() < 1e-4, (err[both].max(), int(np.argmax(np.where(both, err, 0))))
    assert np.isin(code[~ok], (1, 4, 5, 8)).all()  # whatever is not solved is FLAGGED
    n_gpu_only_fails = int((~ok & ~rbad).sum())
    # (round 6: no exception left -- double support over h = 20 at 10x used to end 2-3 % flagged HMPC_S_KKT: instances whose
    #  binary32 Hessian is not positive definite, which qpOASES regularises; see test_indefinite_hessian_is_regularised_like_qpoases)
    assert n_gpu_only_fails == 0, (n_gpu_only_fails, np.unique(code, return_counts=True))


@pytest.mark.parametrize("gait,h,nb", [("standing", 10, 4096), ("mixed", 10, 2048), ("single", 20, 1024)])
def test_repeated_solves_are_bitwise_identical(gait, h, nb):
    """No data race anywhere in the kernel: the same batch solved three times (other workgroups resident in different
    phases each time) gives the same forces and the same status words -- iteration counts included -- bit for bit."""
    f = synthetic.make_batch(nb, h, gait, seed=77, phase="random")
    rec = records.pack_records(f, h)
    mpc = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, nb)
    mpc.upload(rec)
    outs = []
    for _ in range(3):
        mpc.solve()
        forces, status = mpc.download()
        outs.append((forces.copy(), status.copy()))
    mpc.close()
    for forces, status in outs[1:]:
        np.testing.assert_array_equal(status, outs[0][1])
        np.testing.assert_array_equal(forces.view(np.uint32), outs[0][0].view(np.uint32))


@pytest.mark.parametrize("gait,h", [("standing", 10), ("single", 20)])
def test_device_side_safe_pass_equals_the_host_driven_one(gait, h):
    """hmpc_set_device_repair: the safe variant runs behind the fast launch on the same stream over the device-resident
    list of flagged instances -- the outputs in HBM equal those of hmpc_resolve_failed's first (exact) pass, bit for bit,
    with no host round trip; nominal batches are unaffected."""
    nb = 256
    rec = records.pack_records(hard_batch(nb, h, gait, 23, 6), h)
    a = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, nb)
    a.set_auto_resolve(False)
    a.upload(rec)
    a.solve()
    _, st_fast = a.download()
    n_flagged = int(np.isin(interface.status_code(st_fast), (1, 2, 4, 5)).sum())
    assert n_flagged > 0
    b = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, nb)
    b.set_auto_resolve(False)
    b.set_device_repair(True)
    b.upload(rec)
    b.solve()
    f_dev, st_dev = b.download()          # auto-resolve off: exactly what the two launches left in HBM
    b.close()
    a.set_auto_resolve(True)
    f_host, st_host = a.download()
    a.close()
    # what the safe variant itself solves is the same bits either way; the few instances that also need the host-driven
    # last-resort passes (perturbed bounds + exact re-solve: exact on the host side since round 4) stay flagged on the device
    exact = (interface.status_code(st_host) == 0) & (interface.status_code(st_dev) == 0)
    assert (interface.status_code(st_host) == 0).all()
    assert exact.sum() >= nb - 8 and exact.sum() >= (interface.status_code(st_fast) == 0).sum() + n_flagged - 8
    np.testing.assert_array_equal(st_dev[exact], st_host[exact])
    np.testing.assert_array_equal(f_dev[exact].view(np.uint32), f_host[exact].view(np.uint32))
    assert (interface.status_code(st_dev) != 5).all()
    assert np.isin(interface.status_code(st_dev)[~exact], (1, 4)).all()  # flagged (max-iter / KKT), never silently wrong
    # a nominal batch: nothing flagged, same bits with and without the extra (empty) launch
    rec2 = records.pack_records(synthetic.make_batch(nb, h, gait, seed=3, phase="random"), h)
    outs = []
    for on in (False, True):
        m = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, nb)
        m.set_device_repair(on)
        m.upload(rec2)
        m.solve()
        outs.append(m.download())
        m.close()
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    np.testing.assert_array_equal(outs[0][0].view(np.uint32), outs[1][0].view(np.uint32))


def _hard3(nb, seed, scale):
    """make_batch3 with the body state drawn from `scale` times the nominal ranges (as hard_batch does for two contacts)"""
    f = synthetic.make_batch3(nb, 10, "standing", seed=seed, hand="contact")
    g = hard_batch(nb, 10, "standing", seed, scale)
    for k in ("q", "v", "w", "joint_angles", "traj"):
        f[k] = g[k]
    return f


@pytest.mark.parametrize("scale,min_ok", [(3, 1.0), (6, 1.0), (10, 1.0)])
def test_hard_inputs_three_contacts(oracle, scale, min_ok):
    """The three-contact variant (256 threads, two register blocks per thread, working set 96 rows; safe pass 140 rows in LDS,
    then 180 with E in global memory) outside the nominal ranges: every instance solved, exactly as qpOASES solves it."""
    nb = 128
    rec = records.pack_records(_hard3(nb, 19, scale), 10, 3)
    mpc = interface.BatchedMPC(synthetic.DT_MPC, 10, synthetic.F_MAX, nb, contacts=3)
    mpc.upload(rec)
    mpc.solve()
    forces, status = mpc.download()  # with the safe pass
    mpc.close()
    code = interface.status_code(status)
    ok = (code == 0) | (code == 6)
    assert ok.mean() >= min_ok, np.unique(code, return_counts=True)
    ref = oracle.solve_records(rec, 10, synthetic.DT_MPC, synthetic.F_MAX, nc=3)
    q = ref["q_soln"]
    err = np.abs(forces - q).max(axis=1) / np.maximum(1.0, np.abs(q).max(axis=1))
    assert ref["n_bad"] == 0 and err[ok].max() < 1e-4


@pytest.mark.parametrize("h,scale,min_ok", [(20, 3, 1.0), (16, 6, 1.0), (20, 6, 1.0), (16, 10, 1.0)])
def test_hard_inputs_wide_variant(oracle, h, scale, min_ok):
    """Double support over more than ten steps (wide variant, working set 152 rows) outside the nominal ranges: instances that
    outgrow the working set go to the safe pass whose packed Schur inverse lives in global memory (240 rows: cannot overflow;
    VERDICT round 3 item 4) -- 100 % solved where round 3 accepted 1 % / 10 % flagged."""
    nb = 64
    rec = records.pack_records(hard_batch(nb, h, "standing", 29, scale), h)
    mpc = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, nb)
    mpc.upload(rec)
    mpc.solve()
    forces, status = mpc.download()
    mpc.close()
    code = interface.status_code(status)
    ok = (code == 0) | (code == 6)
    assert ok.mean() >= min_ok, np.unique(code, return_counts=True)
    ref = oracle.solve_records(rec, h, synthetic.DT_MPC, synthetic.F_MAX)
    q = ref["q_soln"]
    err = np.abs(forces - q).max(axis=1) / np.maximum(1.0, np.abs(q).max(axis=1))
    assert ref["n_bad"] == 0 and err[ok].max() < 1e-4


@pytest.mark.parametrize("device_repair", [False, True])
def test_unsized_device_records_at_h20_route_the_safe_pass_by_size_class(oracle, device_repair):
    """a two-contact batch at h = 20 handed in by DEVICE pointer without a size hint holds single-support
    (<= 120 variables) and double-support (up to 240) instances side by side.  The fast launches route them by the size classes
    counted on the device; the safe pass must do the same -- a wide instance flagged by the wide variant used to be re-solved by
    the 120-variable safe variant, which ends as HMPC_S_TOO_LARGE with zero forces that nothing re-solves.  Hard inputs (6x the
    nominal ranges) so that both classes do get flagged."""
    import torch

    nb, h = 96, 20
    fa = hard_batch(nb // 2, h, "standing", 29, 6)
    fb = hard_batch(nb // 2, h, "single", 31, 6)
    rec = np.concatenate([records.pack_records(fa, h), records.pack_records(fb, h)])
    perm = np.random.default_rng(3).permutation(nb)
    rec = np.ascontiguousarray(rec[perm])
    d_rec = torch.from_numpy(rec).cuda()
    mpc = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, nb)
    mpc.set_auto_resolve(False)
    mpc.set_device_repair(device_repair)
    mpc.set_device_records(d_rec.data_ptr(), nb, keepalive=d_rec)  # max_reduced_vars = -1: sized on the device
    mpc.solve()
    _, st_fast = mpc.download()
    code_fast = interface.status_code(st_fast)
    assert (code_fast != 3).all(), np.unique(code_fast, return_counts=True)  # never TOO_LARGE: every instance met its variant
    if not device_repair:
        assert np.isin(code_fast, (1, 4, 5)).sum() > 0  # the regime exercises the safe pass
    mpc.set_auto_resolve(True)
    forces, status = mpc.download()
    mpc.close()
    code = interface.status_code(status)
    assert (code != 3).all(), np.unique(code, return_counts=True)
    ok = (code == 0) | (code == 6)
    assert ok.mean() >= 0.97, np.unique(code, return_counts=True)
    ref = oracle.solve_records(rec, h, synthetic.DT_MPC, synthetic.F_MAX)
    q = ref["q_soln"]
    err = np.abs(forces - q).max(axis=1) / np.maximum(1.0, np.abs(q).max(axis=1))
    both = ok & ~np.asarray(ref["bad"], dtype=bool)
    assert err[both].max() < 1e-4
    assert (np.abs(forces[ok]).max(axis=1) > 0).all()  # no "solved" instance with all-zero forces


def test_device_repair_leaves_the_callers_iteration_cap_alone():
    """with hmpc_set_device_repair on, an instance that ran into the CALLER'S cap (hmpc_set_max_iterations)
    is the caller's answer -- status HMPC_S_MAXITER, last iterate in the force buffer -- and is not re-solved cold by the
    on-stream safe launch (whose variants skip the block start: the re-solve would overwrite the forces with a worse iterate)."""
    nb = 256
    rec = records.pack_records(synthetic.make_batch(nb, 10, "standing", seed=77, phase="random"), 10)
    outs = []
    for repair in (False, True):
        m = interface.BatchedMPC(synthetic.DT_MPC, 10, synthetic.F_MAX, nb)
        m.set_auto_resolve(False)
        m.set_device_repair(repair)
        m.set_max_iterations(1)
        m.upload(rec)
        m.solve()
        outs.append(m.download())
        m.close()
    code = interface.status_code(outs[0][1])
    assert (code == 1).sum() > nb // 8 and set(np.unique(code)) <= {0, 1}
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    np.testing.assert_array_equal(outs[0][0].view(np.uint32), outs[1][0].view(np.uint32))


@pytest.mark.parametrize("gait,h,scale", [("standing", 10, 6), ("standing", 10, 10), ("single", 20, 10)])
def test_full_working_set_is_handed_over_not_resolved_cold(oracle, gait, h, scale):
    """hmpc_set_handover (round 6, default on): a fast variant whose working set is full hands its live state (x, u, W, E, M) to
    the safe variant, which CONTINUES -- against the round-5 behaviour (flag, re-solve cold without the block start): the same
    optimum (vs qpOASES and vs the cold path), every instance ok either way, and far fewer iterations in the second pass."""
    nb = 512
    rec = records.pack_records(hard_batch(nb, h, gait, 17, scale), h)
    out = {}
    for mode in ("handover", "cold"):
        m = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, nb)
        m.set_handover(mode == "handover")
        m.set_auto_resolve(False)
        m.upload(rec)
        m.solve()
        _, st_fast = m.download()
        full = interface.status_code(st_fast) == 5
        assert full.sum() >= 0.1 * nb                      # the regime really overflows the fast variant's 64 rows
        assert (interface.status_nactive(st_fast)[full] <= 64).all()  # (64 = capacity reached; fewer = handed over early, right after the block rounds)
        assert m.resolve_failed() == int((interface.status_code(st_fast) != 0).sum())
        f, st = m.download()
        m.close()
        assert (interface.status_code(st) == 0).all(), np.unique(interface.status_code(st), return_counts=True)
        out[mode] = (f, st, full)
    fh, sth, full = out["handover"]
    fc, stc, full_c = out["cold"]
    # with the hand-over on, the fast pass also hands over EARLY (far more candidate rows than its block start takes): a superset
    assert (full | ~full_c).all() and full.sum() >= full_c.sum()
    # instances the fast pass solved in both modes are untouched by either safe pass: bit-identical
    np.testing.assert_array_equal(fh[~full].view(np.uint32), fc[~full].view(np.uint32))
    ref = oracle.solve_records(rec, h, synthetic.DT_MPC, synthetic.F_MAX)
    q = ref["q_soln"]
    for f in (fh, fc):
        err = np.abs(f - q).max(axis=1) / np.maximum(1.0, np.abs(q).max(axis=1))
        assert ref["n_bad"] == 0 and err.max() < 2e-6, err.max()   # (two orders inside the 1e-4 bar)
    # the status word of a continued solve counts the iterations of BOTH passes, a re-solve's its own only (which, since the safe pass
    # got the block start too, is no longer a cold run's): the continuation must not need more in total than starting over does
    it_h, it_c = interface.status_iters(sth)[full_c], interface.status_iters(stc)[full_c]
    assert np.median(it_h) < 1.25 * np.median(it_c), (np.median(it_h), np.median(it_c))
    assert (interface.status_nactive(sth) <= 120).all() and (interface.status_nactive(sth)[full_c] > 64).mean() > 0.5  # (a set may shrink again after its peak)
    assert (interface.status_nactive(sth) == interface.status_nactive(stc)).mean() > 0.9  # the same optimum: (nearly always) the same final set


def test_handover_on_the_device_equals_the_host_driven_one():
    """The continuation behind the fast launch on the same stream (hmpc_set_device_repair) and the one hmpc_resolve_failed
    launches are the same kernel on the same handed-over state: bit-identical forces and status words, no host round trip."""
    nb, h = 1024, 10
    rec = records.pack_records(hard_batch(nb, h, "standing", 17, 6), h)
    a = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, nb)
    a.set_auto_resolve(False)
    a.set_device_repair(True)
    a.upload(rec)
    a.solve()
    fa, sta = a.download()
    a.close()
    b = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, nb)
    b.upload(rec)
    b.solve()
    fb, stb = b.download()   # host-driven safe pass (continuation first, then whatever is still flagged)
    b.close()
    assert (interface.status_code(stb) == 0).all()
    same = interface.status_code(sta) == 0
    assert same.sum() >= nb - 4          # (the device side has no last-resort passes)
    np.testing.assert_array_equal(sta[same], stb[same])
    np.testing.assert_array_equal(fa[same].view(np.uint32), fb[same].view(np.uint32))
    assert (interface.status_nactive(sta) > 64).sum() > 0.1 * nb  # working sets beyond the fast variant's capacity were completed


def test_a_stale_handover_slot_is_never_resumed(oracle):
    """Slots are per instance index and outlive a solve: batch A overflows and is NOT repaired; batch B (other data, same
    indices) is then solved with the hand-over switched off, and again with it on -- the safe pass must continue only from
    state that THIS solve of THIS batch left (slot table rewritten by every fast launch, status word still 'working set full')."""
    nb, h = 256, 10
    rec_a = records.pack_records(hard_batch(nb, h, "standing", 17, 10), h)
    rec_b = records.pack_records(hard_batch(nb, h, "standing", 31, 6), h)
    ref = oracle.solve_records(rec_b, h, synthetic.DT_MPC, synthetic.F_MAX)
    q = ref["q_soln"]
    for second_mode in (False, True):
        m = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, nb)
        m.set_auto_resolve(False)
        m.upload(rec_a)
        m.solve()
        _, st_a = m.download()
        assert (interface.status_code(st_a) == 5).sum() > 0.3 * nb   # slots of A are filled and never consumed
        m.set_handover(second_mode)
        m.upload(rec_b)
        m.solve()
        m.resolve_failed()
        f, st = m.download()
        m.close()
        assert (interface.status_code(st) == 0).all()
        err = np.abs(f - q).max(axis=1) / np.maximum(1.0, np.abs(q).max(axis=1))
        assert err.max() < 2e-6, (second_mode, err.max())


def test_indefinite_hessian_is_regularised_like_qpoases():
    """Double support over h = 20 at 10x the nominal input ranges: in ~2 % of the instances the reduced Hessian, assembled in binary32
    as the contract demands (SolverMPC.cpp:560-570), is NOT positive definite (smallest eigenvalue ~ -5e-5 against entries of 1e3).
    qpOASES answers the failed Cholesky factorisation by regularising -- H + rho I, rho = |H|_F sqrt(1e3 eps), then one more QP with
    the gradient g - rho x_1 (QProblem.cpp:1753-1860, QProblemB.cpp:1418-1431, 1999-2031; Options::setToMPC) -- and reports success;
    the fast variants diverge on such an instance and flag it (KKT), the safe variants find the non-positive sweep pivot
    (HMPC_S_INDEFINITE) and hmpc_resolve_failed runs the same two regularised QPs: every instance ends HMPC_S_OK within 1e-6 of qpOASES
    (until round 6: 24 of 1 024 ended flagged).  The device-side chain (hmpc_set_device_repair) runs the same two launches."""
    from oracle import oracle_py, pool

    nb, h = 96, 20
    rec = records.pack_records(hard_batch(nb, h, "standing", 17, 10), h)
    # which instances are indefinite: eigenvalues of the oracle's reduced Hessian (binary32 assembly, widened)
    indef = np.zeros(nb, dtype=bool)
    for i in range(nb):
        a = oracle_py.assemble_record(rec[i], h, synthetic.DT_MPC, synthetic.F_MAX)
        indef[i] = np.linalg.eigvalsh(a["H_red"])[0] < -1e-7
    assert 2 <= indef.sum() <= 12, int(indef.sum())
    ref = pool.solve_records_parallel(rec, h, synthetic.DT_MPC, synthetic.F_MAX)
    rbad = np.asarray(ref["bad"], dtype=bool)
    q = ref["q_soln"]
    both = indef & ~rbad
    assert both.sum() >= 2
    # fast pass alone: never "solved"
    mpc = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, nb)
    mpc.set_auto_resolve(False)
    mpc.upload(rec)
    mpc.solve()
    _, st_fast = mpc.download()
    assert (interface.status_code(st_fast)[indef] != 0).all()
    # host-driven repair: the reference's regularisation steps inside hmpc_resolve_failed
    assert mpc.resolve_failed() >= int(indef.sum())
    forces, status = mpc.download()
    mpc.close()
    code = interface.status_code(status)
    assert (code == 0).all(), np.unique(code, return_counts=True)
    err = np.abs(forces - q).max(axis=1) / np.maximum(1.0, np.abs(q).max(axis=1))
    assert err[both].max() < 1e-6, err[indef]
    assert err[~rbad].max() < 1e-4
    # device-side chain: the same two launches behind the safe pass, no host in the loop -- the same answers
    mpc = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, nb)
    mpc.set_auto_resolve(False)
    mpc.set_device_repair(True)
    mpc.upload(rec)
    mpc.solve()
    f_dev, st_dev = mpc.download()
    mpc.close()
    assert (interface.status_code(st_dev)[indef] == 0).all(), interface.status_code(st_dev)[indef]
    np.testing.assert_array_equal(f_dev[indef].view(np.uint32), forces[indef].view(np.uint32))
