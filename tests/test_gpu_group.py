"""Device groups behind the C ABI (include/hector_mpc.h ``hmpc_group_*``, csrc/hmpc_group.hip):
contiguous slices per member, no data-path collective, one exchange step = the gather of step-0 wrench + status on every
member and on the host.  A one-GPU box exercises (a) the RCCL transport with a group of one (ncclCommInitAll +
ncclAllGather on hardware) and (b) the multi-member slicing / packing / stream ordering / layout with device 0 listed
several times over the P2P transport.  No run with more than one physical GPU exists."""
import ctypes as C

import numpy as np
import pytest

from hector_simulation_amd import interface, records, synthetic

pytestmark = pytest.mark.gpu
H = 10


def _single(rec):
    mpc = interface.BatchedMPC(synthetic.DT_MPC, H, synthetic.F_MAX, rec.shape[0])
    mpc.upload(rec)
    mpc.solve()
    f, s = mpc.download()
    mpc.close()
    return f, s


def _device_words(ptr, nwords):
    import torch  # plumbing only: a device -> host copy of the gathered block

    out = np.zeros(nwords, dtype=np.uint32)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(out.ctypes.data, C.c_void_p(ptr), nwords * 4, 2) == 0
    return out


@pytest.mark.parametrize("devices,transport,nb", [([0], "rccl", 96), ([0], "auto", 1), ([0, 0, 0], "p2p", 100),
                                                 ([0, 0], "p2p", 64), ([0, 0, 0, 0], "auto", 3)])
def test_group_gather_equals_single_handle(devices, transport, nb):
    f = synthetic.make_batch(nb, H, "walking", seed=31, phase="random")
    rec = records.pack_records(f, H)
    ref_f, ref_s = _single(rec)
    grp = interface.DeviceGroup(synthetic.DT_MPC, H, synthetic.F_MAX, nb, devices, transport)
    assert grp.transport == ("rccl" if len(set(devices)) == len(devices) else "p2p")
    grp.upload(rec)
    covered = []
    for i in range(grp.size):
        _, dev, lo, n, _ = grp.member(i)
        assert (lo, lo + n) == interface.shard_bounds(nb, grp.size, i)
        covered += list(range(lo, lo + n))
    assert covered == list(range(nb))
    grp.solve()
    wrench, status = grp.gather_wrench()
    np.testing.assert_array_equal(status, ref_s)
    np.testing.assert_array_equal(wrench.view(np.uint32), ref_f[:, :12].view(np.uint32))
    # every member holds the same gathered block in HBM
    blocks = []
    for i in range(grp.size):
        ptr, rows = grp.device_gathered(i)
        blk = _device_words(ptr, grp.size * rows * 13).reshape(grp.size, rows, 13)
        blocks.append(blk)
        for s in range(grp.size):
            _, _, lo, n, _ = grp.member(s)
            np.testing.assert_array_equal(blk[s, :n, :12], ref_f[lo:lo + n, :12].view(np.uint32))
            np.testing.assert_array_equal(blk[s, :n, 12], ref_s[lo:lo + n])
    for b in blocks[1:]:
        np.testing.assert_array_equal(b, blocks[0])
    full_f, full_s = grp.download()
    np.testing.assert_array_equal(full_f.view(np.uint32), ref_f.view(np.uint32))
    np.testing.assert_array_equal(full_s, ref_s)
    grp.close()


def test_group_pipelined_exchange_under_the_next_solve():
    """solve k+1 is enqueued before the gather of solve k is waited for; each gather must still carry solve k's data."""
    nb = 90
    batches = [records.pack_records(synthetic.make_batch(nb, H, g, seed=40 + i, phase="random"), H)
               for i, g in enumerate(("walking", "standing", "mixed", "walking"))]
    refs = [_single(r) for r in batches]
    grp = interface.DeviceGroup(synthetic.DT_MPC, H, synthetic.F_MAX, nb, [0, 0, 0], "p2p")
    grp.upload(batches[0])
    grp.solve()
    for k in range(len(batches)):
        grp.post_gather()              # exchange of solve k on the comm streams ...
        if k + 1 < len(batches):
            grp.upload(batches[k + 1])
            grp.solve()                # ... under solve k+1 on the solve streams
        wrench, status = grp.gather_wrench()
        np.testing.assert_array_equal(status, refs[k][1])
        np.testing.assert_array_equal(wrench.view(np.uint32), refs[k][0][:, :12].view(np.uint32))
    grp.close()


def test_group_rejects_repeated_devices_over_rccl():
    with pytest.raises(interface.HmpcError):
        interface.DeviceGroup(synthetic.DT_MPC, H, synthetic.F_MAX, 8, [0, 0], "rccl")


def test_group_device_resident_records():
    """hmpc_group_set_device_records: every member solves its slice straight from records that already live in HBM."""
    import torch

    nb = 77
    f = synthetic.make_batch(nb, H, "standing", seed=12)
    rec = records.pack_records(f, H)
    ref_f, ref_s = _single(rec)
    d_rec = torch.from_numpy(rec).cuda()
    grp = interface.DeviceGroup(synthetic.DT_MPC, H, synthetic.F_MAX, nb, [0, 0, 0], "p2p")
    ptrs = []
    for i in range(grp.size):
        lo, hi = interface.shard_bounds(nb, grp.size, i)
        ptrs.append(d_rec.data_ptr() + lo * rec.shape[1])
    torch.cuda.synchronize()
    grp.set_device_records(ptrs, nb, max_reduced_vars=120, keepalive=d_rec)
    grp.solve()
    wrench, status = grp.gather_wrench()
    np.testing.assert_array_equal(status, ref_s)
    np.testing.assert_array_equal(wrench.view(np.uint32), ref_f[:, :12].view(np.uint32))
    grp.close()


def test_gather_after_a_collected_exchange_posts_a_fresh_one():
    """solve, post, wait (collected through the device copy), solve again, gather_wrench: the second solve's data -- and
    the host unpack follows the slices of the batch that was POSTED, also when the batch size changed in between."""
    nb1, nb2 = 60, 45
    rec1 = records.pack_records(synthetic.make_batch(nb1, H, "walking", seed=51, phase="random"), H)
    rec2 = records.pack_records(synthetic.make_batch(nb2, H, "standing", seed=52), H)
    ref1, ref2 = _single(rec1), _single(rec2)
    grp = interface.DeviceGroup(synthetic.DT_MPC, H, synthetic.F_MAX, nb1, [0, 0, 0], "p2p")
    grp.upload(rec1)
    grp.solve()
    grp.post_gather()
    grp.wait_gather()                      # collected: the caller reads hmpc_group_device_gathered
    ptr, rows = grp.device_gathered(1)
    blk = _device_words(ptr, grp.size * rows * 13).reshape(grp.size, rows, 13)
    lo, hi = interface.shard_bounds(nb1, 3, 2)
    np.testing.assert_array_equal(blk[2, :hi - lo, :12], ref1[0][lo:hi, :12].view(np.uint32))
    grp.upload(rec2)
    grp.solve()
    wrench, status = grp.gather_wrench()   # must post anew and return the SECOND solve
    np.testing.assert_array_equal(status, ref2[1])
    np.testing.assert_array_equal(wrench.view(np.uint32), ref2[0][:, :12].view(np.uint32))
    # pipelined with a batch-size change between post and collection: rows land at the posted batch's slices
    grp.upload(rec1)
    grp.solve()
    grp.post_gather()
    grp.upload(rec2)
    grp.solve()
    w1 = np.zeros((nb1, 12), dtype=np.float32)
    s1 = np.zeros(nb1, dtype=np.uint32)
    grp._check(grp.L.hmpc_group_gather_wrench(grp.g, w1.ctypes.data, s1.ctypes.data), "gather")
    np.testing.assert_array_equal(s1, ref1[1])
    np.testing.assert_array_equal(w1.view(np.uint32), ref1[0][:, :12].view(np.uint32))
    grp.close()


def test_exchange_carries_repaired_rows():
    """6x the nominal input ranges: the fast variant flags some instances (working set full).  With the members'
    device-side safe pass (default) the gathered wrench/status are the REPAIRED ones -- equal to what hmpc_download returns
    after its host-driven safe pass; with it off the exchange shows the flags."""
    from tests.test_gpu_robustness import hard_batch

    nb = 384
    rec = records.pack_records(hard_batch(nb, H, "standing", 17, 6), H)
    mpc = interface.BatchedMPC(synthetic.DT_MPC, H, synthetic.F_MAX, nb)
    mpc.set_auto_resolve(False)
    mpc.upload(rec)
    mpc.solve()
    _, st_fast = mpc.download()
    assert (interface.status_code(st_fast) == 5).any()      # the regime exercises the safe pass
    mpc.set_auto_resolve(True)
    ref_f, ref_s = mpc.download()                            # host-driven safe pass (+ relaxed passes)
    mpc.close()
    grp = interface.DeviceGroup(synthetic.DT_MPC, H, synthetic.F_MAX, nb, [0, 0, 0], "p2p")
    grp.upload(rec)
    grp.solve()
    wrench, status = grp.gather_wrench()
    code, ref_code = interface.status_code(status), interface.status_code(ref_s)
    assert (code != 5).all()                                  # nothing is left "working set full"
    same = ref_code == 0                                      # instances the first safe pass solves exactly
    np.testing.assert_array_equal(code[same], ref_code[same])
    np.testing.assert_array_equal(wrench[same].view(np.uint32), ref_f[same, :12].view(np.uint32))
    assert (code[~same] != 0).all()                           # what needs the relaxed passes stays flagged, never silent
    grp.set_exchange_repair(False)
    grp.solve()
    _, status_raw = grp.gather_wrench()
    np.testing.assert_array_equal(status_raw, st_fast)
    grp.close()


@pytest.mark.parametrize("devices,transport,nb", [([0], "rccl", 40), ([0, 0, 0, 0], "p2p", 50)])
def test_group_of_three_contact_handles_cfg5_split(devices, transport, nb):
    """BASELINE config 5 is the three-contact extension split over 4 GPUs: ``hmpc_group_create_ex(..., n_contacts = 3)``
    -- member handles of the extension, records of its stride, and an exchange that carries the 18 step-0 values
    [F_L F_R F_H M_L M_R M_H] (ConvexMPCLocomotion.cpp:419-440 with one more contact) + status per instance."""
    f = synthetic.make_batch3(nb, H, "standing", seed=33, phase="random", hand="window")
    rec = records.pack_records(f, H, 3)
    one = interface.BatchedMPC(synthetic.DT_MPC, H, synthetic.F_MAX, nb, contacts=3)
    one.upload(rec)
    one.solve()
    ref_f, ref_s = one.download()
    one.close()
    grp = interface.DeviceGroup(synthetic.DT_MPC, H, synthetic.F_MAX, nb, devices, transport, contacts=3)
    assert grp.wrench_width == 18 and grp.stride == records.record_stride(H, 3)
    assert int(grp.L.hmpc_group_contacts(grp.g)) == 3
    grp.upload(rec)
    grp.solve()
    wrench, status = grp.gather_wrench()
    assert wrench.shape == (nb, 18)
    np.testing.assert_array_equal(status, ref_s)
    np.testing.assert_array_equal(wrench.view(np.uint32), ref_f[:, :18].view(np.uint32))
    for i in range(grp.size):
        ptr, rows = grp.device_gathered(i)
        blk = _device_words(ptr, grp.size * rows * 19).reshape(grp.size, rows, 19)
        for s in range(grp.size):
            _, _, lo, n, _ = grp.member(s)
            np.testing.assert_array_equal(blk[s, :n, :18], ref_f[lo:lo + n, :18].view(np.uint32))
            np.testing.assert_array_equal(blk[s, :n, 18], ref_s[lo:lo + n])
    full_f, full_s = grp.download()
    assert full_f.shape == (nb, 18 * H)
    np.testing.assert_array_equal(full_f.view(np.uint32), ref_f.view(np.uint32))
    np.testing.assert_array_equal(full_s, ref_s)
    grp.close()


def _predicted_cost(rec, h):
    """host restatement of the kernel-side predictor's score (hmpc_builder.h predicted_cost_bucket), two contacts"""
    f = np.ascontiguousarray(rec[:, : 4 * (54 + 12 * h)]).view(np.float32)
    u = (f[:, 54 + 9] - f[:, 3]) + 2.0 * 0.5 * (f[:, 13] + f[:, 14])
    return np.where(u > 0, u, -0.05 * u)


def test_striped_deal_balances_a_skewed_batch():
    """group-level balance.  A parameter sweep is usually ORDERED, so its hard instances sit together; cut
    into contiguous slices one member gets all of them and the gather waits for it.  Batch sorted hardest-first (by the record-only
    cost predictor, i.e. the worst case for contiguous slices), four members on the box's one GPU, each member's solve timed ALONE:
      contiguous slices: the first member's kernel takes much longer than the last one's;
      striped deal (hmpc_group_set_deal): all members within 10 % of each other;
    and the host-facing results are the same, in instance order, bit for bit, either way."""
    h, nb, G = 10, 8192, 4
    f = synthetic.make_batch(nb, h, "standing", seed=6, phase="random")
    rec = records.pack_records(f, h)
    rec = np.ascontiguousarray(rec[np.argsort(-_predicted_cost(rec, h), kind="stable")])  # an ordered sweep: hardest first
    grp = interface.DeviceGroup(synthetic.DT_MPC, h, synthetic.F_MAX, nb, [0] * G, transport="p2p")
    L = grp.L
    out, times = {}, {}
    for striped in (False, True):
        grp.set_deal(striped)
        grp.upload(rec)
        grp.solve()
        grp.synchronize()
        ms = []
        for i in range(G):
            hdl, _, lo, n, st = grp.member(i)
            assert grp.member_step(i) == (G if striped else 1) and lo == (i if striped else i * (nb // G)) and n == nb // G
            L.hmpc_set_dispatch_order(hdl, 0)   # natural order inside a member: what is compared is the DEAL
            best = 1e9
            for _ in range(3):
                t = C.c_float(0)
                assert L.hmpc_time_solve(hdl, C.c_void_p(st), 5, C.byref(t)) == 0
                best = min(best, t.value)
            ms.append(best)
        times[striped] = ms
        grp.solve()
        wrench, wstat = grp.gather_wrench()
        forces, status = grp.download()
        out[striped] = (wrench.copy(), wstat.copy(), forces.copy(), status.copy())
    grp.close()
    for a, b in zip(out[False], out[True]):
        np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
    np.testing.assert_array_equal(out[True][0], out[True][2][:, :12])     # the gathered wrench = step 0 of the forces
    assert (interface.status_code(out[True][3]) == 0).all()
    cont, strp = np.array(times[False]), np.array(times[True])
    print("member kernel ms, contiguous slices:", np.round(cont, 4), " striped:", np.round(strp, 4))
    assert cont.max() / cont.min() > 1.15, cont        # the skew is real ...
    assert strp.max() / strp.min() < 1.10, strp        # ... and the striped deal removes it
    assert strp.max() < cont.max()                     # the member the gather waits for got faster


@pytest.mark.parametrize("nb,G", [(101, 3), (7, 4), (2, 4), (0, 2)])
def test_striped_deal_ragged_batches_equal_a_single_handle(nb, G):
    """striped deal with batch sizes that do not divide (and members left empty): instance order on the host side, bit for bit."""
    rec = records.pack_records(synthetic.make_batch(max(nb, 1), H, "mixed", seed=13, phase="random"), H)[:nb]
    grp = interface.DeviceGroup(synthetic.DT_MPC, H, synthetic.F_MAX, max(nb, 1), [0] * G, transport="p2p")
    grp.set_deal(True)
    grp.upload(rec)
    seen = []
    for i in range(G):
        _, _, lo, n, _ = grp.member(i)
        seen += [lo + k * grp.member_step(i) for k in range(n)]
    assert sorted(seen) == list(range(nb))
    grp.solve()
    wrench, wstat = grp.gather_wrench()
    forces, status = grp.download()
    grp.close()
    if nb:
        ref_f, ref_s = _single(rec)
        np.testing.assert_array_equal(status, ref_s)
        np.testing.assert_array_equal(forces.view(np.uint32), ref_f.view(np.uint32))
        np.testing.assert_array_equal(wrench.view(np.uint32), ref_f[:, :12].view(np.uint32))
        np.testing.assert_array_equal(wstat, ref_s)


def test_group_command_sweep_equals_a_single_handle():
    """hmpc_group_solve_command_sweep: every member runs the command sweep on its slice (whole groups per slice) -- the gathered
    wrench and the full download are the bits of the independent solves of a single handle; slices that would tear a group apart,
    and the striped deal, are refused before anything is enqueued."""
    from tests.test_gpu_command_sweep import sweep_fields

    groups, k = 12, 8                      # 96 instances: three members x four groups x eight commands
    rec = records.pack_records(sweep_fields(groups, k, H, "standing", seed=51), H)
    ref_f, ref_s = _single(rec)
    grp = interface.DeviceGroup(synthetic.DT_MPC, H, synthetic.F_MAX, groups * k, [0, 0, 0], "p2p")
    grp.upload(rec)
    grp.solve_command_sweep(k)
    wrench, status = grp.gather_wrench()
    np.testing.assert_array_equal(status, ref_s)
    np.testing.assert_array_equal(wrench.view(np.uint32), ref_f[:, :12].view(np.uint32))
    full_f, full_s = grp.download()
    np.testing.assert_array_equal(full_f.view(np.uint32), ref_f.view(np.uint32))
    with pytest.raises(interface.HmpcError):
        grp.solve_command_sweep(5)         # 32 instances per member are not whole groups of five
    grp.set_deal(True)
    grp.upload(rec)
    with pytest.raises(interface.HmpcError):
        grp.solve_command_sweep(k)         # a striped deal scatters every group over the members
    grp.close()
