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
