"""Host-side logic: packed record layout and the synthetic input generators (no GPU, no oracle)."""
import numpy as np

from hector_simulation_amd import records, synthetic


def test_stride_and_payload():
    assert records.payload_bytes(10) == 716 and records.record_stride(10) == 720  # 716 B at h=10
    assert records.payload_bytes(20) == 1216 and records.record_stride(20) == 1216
    for h in range(1, 21):
        assert records.record_stride(h) % 16 == 0 and records.record_stride(h) >= records.payload_bytes(h)


def test_pack_unpack_roundtrip():
    f = synthetic.make_batch(5, 10, "mixed", seed=1, phase="random")
    rec = records.pack_records(f, 10)
    assert rec.shape == (5, 720) and rec.dtype == np.uint8
    u = records.unpack_records(rec, 10)
    for k in ("p", "v", "q", "w", "r", "joint_angles", "weights", "Alpha_K", "traj"):
        np.testing.assert_array_equal(u[k], np.asarray(f[k], dtype=np.float64).astype(np.float32).reshape(5, -1))
    np.testing.assert_array_equal(u["gait"], np.asarray(f["gait"]).astype(np.uint8))


def test_reference_gait_tables():
    # GaitGenerator.cpp:85-103 with Gait(10,(0,5),(5,5)) (ConvexMPCLocomotion.cpp:16): left stance 0-4, right 5-9
    t = synthetic.mpc_gait(10, (0, 5), (5, 5), 0).reshape(10, 2)
    assert t[:5, 0].all() and not t[5:, 0].any() and not t[:5, 1].any() and t[5:, 1].all()
    t3 = synthetic.mpc_gait(10, (0, 5), (5, 5), 3).reshape(10, 2)
    np.testing.assert_array_equal(t3, np.roll(t, -3, axis=0))
    assert synthetic.mpc_gait(10, (0, 0), (10, 10), 4).all()  # standing: both feet always in stance
    # every walking phase is single support
    for ph in range(10):
        assert (synthetic.mpc_gait(10, (0, 5), (5, 5), ph).reshape(10, 2).sum(axis=1) == 1).all()


def test_generator_is_seeded_and_shaped():
    a = synthetic.make_batch(7, 10, "walking", seed=3, phase="random")
    b = synthetic.make_batch(7, 10, "walking", seed=3, phase="random")
    for k in a:
        np.testing.assert_array_equal(a[k], b[k])
    assert a["traj"].shape == (7, 120) and a["gait"].shape == (7, 20) and a["r"].shape == (7, 6)
    # r[2*axis+leg]: left foot +y, right foot -y, both below the body
    assert (a["r"][:, 2] > 0).all() and (a["r"][:, 3] < 0).all() and (a["r"][:, 4:] < 0).all()
    n = synthetic.make_batch(1, 10, "standing", seed=1, randomize=False)
    assert np.allclose(n["q"], [[1, 0, 0, 0]]) and np.allclose(n["p"], [[0, 0, 0.55]])
    for name, cfg in synthetic.CONFIGS.items():
        assert cfg["horizon"] in (10, 20) and cfg["batch"] >= 1
