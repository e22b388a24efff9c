"""Longest-first dispatch (include/hector_mpc.h hmpc_set_dispatch_order; the library's default): workgroup b takes the instance
with the b-th most iterations in the PREVIOUS solve.  What is under test: it is a pure scheduling choice -- every instance is solved exactly once
and bit-identically to the natural order, also when the previous solve was of other data, of another batch size, or absent."""
import numpy as np
import pytest

from hector_simulation_amd import interface, records, synthetic

pytestmark = pytest.mark.gpu


def _batches(nb, contacts, seeds):
    out = []
    for sd in seeds:
        if contacts == 3:
            f = synthetic.make_batch3(nb, 10, "standing", seed=sd, hand="contact")
            out.append(records.pack_records(f, 10, 3))
        else:
            f = synthetic.make_batch(nb, 10, "mixed", seed=sd, phase="random")
            out.append(records.pack_records(f, 10))
    return out


@pytest.mark.parametrize("contacts,nb", [(2, 1500), (3, 700)])
def test_longest_first_is_only_a_schedule(contacts, nb):
    rec1, rec2 = _batches(nb, contacts, (41, 42))
    nat = interface.BatchedMPC(synthetic.DT_MPC, 10, synthetic.F_MAX, nb, contacts=contacts)
    lpt = interface.BatchedMPC(synthetic.DT_MPC, 10, synthetic.F_MAX, nb, contacts=contacts)
    nat.set_dispatch_order(False)
    want = {}
    for key, rec in (("r1", rec1), ("r2", rec2)):
        nat.upload(rec)
        nat.solve()
        want[key] = nat.download()
    # first solve of the handle: natural order; second: ordered by the iterations of OTHER data (rec1's); third: by its own
    for key, rec in (("r1", rec1), ("r2", rec2), ("r2", rec2)):
        lpt.upload(rec)
        lpt.solve()
        forces, status = lpt.download()
        np.testing.assert_array_equal(status, want[key][1])
        np.testing.assert_array_equal(forces, want[key][0])
    assert (interface.status_code(want["r2"][1]) == 0).all()
    assert interface.status_iters(want["r2"][1]).max() > interface.status_iters(want["r2"][1]).min()  # there is something to sort
    # another batch size: natural order again (the previous statuses describe other instances), then ordered
    half = nb // 2
    nat.upload(rec1[:half])
    nat.solve()
    f_half, s_half = nat.download()
    for _ in range(2):
        lpt.upload(rec1[:half])
        lpt.solve()
        forces, status = lpt.download()
        np.testing.assert_array_equal(status, s_half)
        np.testing.assert_array_equal(forces, f_half)
    # and off again
    lpt.set_dispatch_order(False)
    lpt.upload(rec2)
    lpt.solve()
    forces, status = lpt.download()
    np.testing.assert_array_equal(forces, want["r2"][0])
    nat.close()
    lpt.close()


def test_longest_first_with_device_built_walking_batch():
    """Size-class routing (every variant launched over the whole batch) takes the same order list."""
    import torch

    nb = 1024
    rec = np.concatenate([records.pack_records(synthetic.make_batch(nb // 2, 10, "walking", seed=7, phase="random"), 10),
                          records.pack_records(synthetic.make_batch(nb // 2, 10, "standing", seed=8), 10)])
    d_rec = torch.from_numpy(rec).cuda()
    torch.cuda.synchronize()
    out = []
    for mode in (False, True):
        mpc = interface.BatchedMPC(synthetic.DT_MPC, 10, synthetic.F_MAX, nb)
        mpc.set_dispatch_order(mode)
        mpc.set_device_records(d_rec.data_ptr(), nb, keepalive=d_rec)  # no size hint: classes on the device
        mpc.solve()
        mpc.solve()
        out.append(mpc.download())
        mpc.close()
    np.testing.assert_array_equal(out[0][1], out[1][1])
    np.testing.assert_array_equal(out[0][0], out[1][0])
    assert (interface.status_code(out[0][1]) == 0).all()


def _predicted_bucket(rec, h, nc):
    """numpy restatement of hmpc_builder.h predicted_cost_bucket (binary32 as on the device)."""
    nf = 73 if nc == 3 else 54
    f = np.ascontiguousarray(rec[:, : 4 * (nf + 12 * h)]).view(np.float32)
    vx, qw, qx, qy, qz = f[:, 3], f[:, 6], f[:, 7], f[:, 8], f[:, 9]
    mrx = np.float32(0.5) * (f[:, 13] + f[:, 14])
    u = (f[:, nf + 9] - vx) + np.float32(2.0) * mrx
    sr = np.float32(2.0) * (qw * qx + qy * qz)
    sp = np.float32(2.0) * (qw * qy - qx * qz)
    score = np.where(u > 0, u, np.float32(-0.05) * u) + np.float32(0.5) * (np.abs(sr) + np.abs(sp))
    return np.clip((score * np.float32(48.0)).astype(np.int32), 0, 63)


@pytest.mark.parametrize("contacts,nb", [(2, 2048), (3, 1024)])
def test_cold_handle_is_ordered_by_the_predictor(contacts, nb):
    """an order hint that needs no previous solve.  (1) the predictor says something about the solve it
    orders: its bucket correlates with the iteration counts the instances then take (0.8 on these sets; asserted at 0.6);
    (2) mode 2 (always predicted) and the cold first solve of mode 1 are pure scheduling: bit-identical to natural order."""
    if contacts == 3:
        rec = records.pack_records(synthetic.make_batch3(nb, 10, "standing", seed=5, hand="contact"), 10, 3)
    else:
        rec = records.pack_records(synthetic.make_batch(nb, 10, "standing", seed=6, phase="random"), 10)
    outs = {}
    for mode in (0, 1, 2):
        m = interface.BatchedMPC(synthetic.DT_MPC, 10, synthetic.F_MAX, nb, contacts=contacts)
        m.set_dispatch_order(mode)
        m.upload(rec)
        m.solve()          # the FIRST solve of the handle: mode 1 has no previous solve to go by
        outs[mode] = m.download()
        if mode == 2:
            m.solve()      # ... and mode 2 never uses one
            again = m.download()
            np.testing.assert_array_equal(again[1], outs[mode][1])
        m.close()
    for mode in (1, 2):
        np.testing.assert_array_equal(outs[mode][1], outs[0][1])
        np.testing.assert_array_equal(outs[mode][0].view(np.uint32), outs[0][0].view(np.uint32))
    it = interface.status_iters(outs[0][1]).astype(np.float64)
    b = _predicted_bucket(rec, 10, contacts).astype(np.float64)
    assert np.corrcoef(b, it)[0, 1] > 0.6, np.corrcoef(b, it)[0, 1]
    # the instances the predictor starts first really are the long ones: mean iterations of its top decile vs the rest
    top = b >= np.quantile(b, 0.9)
    assert it[top].mean() > 2.0 * it[~top].mean()
