"""Full-batch oracle parity at BASELINE.json's sizes: EVERY instance of configs 2 and 4, of the metric's b1024 2-contact
case, and -- with the seeds bench.py gives its ranks (seed 6 + 1000 rank) -- of ALL 8 shards of config 3 (65 536 walking
instances) and ALL 4 shards of config 5 (8 192 three-contact instances), against the reference's own qpOASES on the oracle's (bit-identical)
QP data.  The oracle runs as a pool of processes over the host cores (oracle/pool.py) where one core would take more
than a few seconds.  Bar: forces within north_star's 1e-4 relative of qpOASES, every instance reported ok."""
import numpy as np
import pytest

from hector_simulation_amd import interface, records, synthetic

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _gpu(rec, h, contacts=2):
    nb = rec.shape[0]
    mpc = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, nb, contacts=contacts)
    mpc.upload(rec)
    mpc.solve()
    forces, status = mpc.download()
    mpc.close()
    return forces, status


def _compare(name, forces, status, ref):
    assert ref["n_bad"] == 0
    code = interface.status_code(status)
    assert (code == 0).all(), (name, np.bincount(code))
    q = ref["q_soln"]
    err = np.abs(forces.astype(np.float64) - q).max(axis=1) / np.maximum(1.0, np.abs(q).max(axis=1))
    print(f"{name}: {len(err)} instances vs qpOASES: max rel force err {err.max():.2e}, median {np.median(err):.2e}; "
          f"qpOASES nWSR median {np.median(ref['nwsr']):.0f}, GPU iterations median "
          f"{np.median(interface.status_iters(status)):.0f}")
    assert err.max() < TOL, (name, err.max(), int(err.argmax()))


def test_cfg2_all_1024_walking_fixed_phase(oracle):
    kw = synthetic.CONFIGS["cfg2_walk_1024"]
    rec = records.pack_records(synthetic.make_batch(**kw), kw["horizon"])
    forces, status = _gpu(rec, kw["horizon"])
    _compare("cfg2", forces, status, oracle.solve_records(rec, kw["horizon"], synthetic.DT_MPC, synthetic.F_MAX))


def test_metric_case_all_1024_two_contact(oracle):
    from oracle import pool

    kw = synthetic.CONFIGS["metric_2contact_1024"]
    rec = records.pack_records(synthetic.make_batch(**kw), kw["horizon"])
    forces, status = _gpu(rec, kw["horizon"])
    _compare("metric b1024", forces, status, pool.solve_records_parallel(rec, kw["horizon"], synthetic.DT_MPC, synthetic.F_MAX))


def test_cfg4_all_4096_horizon_20_single_support(oracle):
    from oracle import pool

    kw = synthetic.CONFIGS["cfg4_h20_single_4096"]
    rec = records.pack_records(synthetic.make_batch(**kw), kw["horizon"])
    forces, status = _gpu(rec, kw["horizon"])
    _compare("cfg4", forces, status, pool.solve_records_parallel(rec, kw["horizon"], synthetic.DT_MPC, synthetic.F_MAX))


def test_cfg3_one_gpu_shard_8192_of_the_random_sweep(oracle):
    """BASELINE config 3 is 65 536 instances over 8 GPUs: one GPU's shard, every instance."""
    from oracle import pool

    kw = dict(synthetic.CONFIGS["cfg3_walk_sweep_65536"], batch=8192)
    rec = records.pack_records(synthetic.make_batch(**kw), kw["horizon"])
    forces, status = _gpu(rec, kw["horizon"])
    _compare("cfg3[:8192]", forces, status, pool.solve_records_parallel(rec, kw["horizon"], synthetic.DT_MPC, synthetic.F_MAX))


def test_cfg5_one_gpu_shard_2048_three_contact_extension(oracle):
    """BASELINE config 5 is 8 192 instances over 4 GPUs: one GPU's shard, every instance."""
    from oracle import pool

    kw = dict(synthetic.CONFIG5, batch=2048)
    rec = records.pack_records(synthetic.make_batch3(**kw), 10, 3)
    forces, status = _gpu(rec, 10, contacts=3)
    _compare("cfg5[:2048] (oracle extension)", forces, status,
             pool.solve_records_parallel(rec, 10, synthetic.DT_MPC, synthetic.F_MAX, nc=3))


def _bench_shards(n_ranks, per_rank, gait, contacts):
    """every rank's shard of a `bench.py --gpus n_ranks --batch per_rank` run, solved one after the other on this GPU"""
    import bench
    from oracle import pool

    worst, not_ok, total = 0.0, 0, 0
    for rank in range(n_ranks):
        _, rec = bench.bench_shard(rank, per_rank, 10, gait, contacts)
        forces, status = _gpu(rec, 10, contacts=contacts)
        ref = pool.solve_records_parallel(rec, 10, synthetic.DT_MPC, synthetic.F_MAX, nc=contacts)
        assert ref["n_bad"] == 0, (rank, ref["n_bad"])
        q = ref["q_soln"]
        err = np.abs(forces.astype(np.float64) - q).max(axis=1) / np.maximum(1.0, np.abs(q).max(axis=1))
        worst = max(worst, float(err.max()))
        not_ok += int((interface.status_code(status) != 0).sum())
        total += len(err)
    return worst, not_ok, total


def test_cfg3_all_eight_shards_65536_with_the_bench_seeds(oracle):
    """BASELINE config 3 as `bench.py --gpus 8 --gait walking` shards it: 8 x 8 192 instances, every one against qpOASES."""
    worst, not_ok, total = _bench_shards(8, 8192, "walking", 2)
    print(f"cfg3, all 8 bench shards: {total} instances, {not_ok} not ok, max rel force err vs qpOASES {worst:.2e}")
    assert total == 65536 and not_ok == 0 and worst < TOL


def test_cfg5_all_four_shards_8192_three_contact_with_the_bench_seeds(oracle):
    """BASELINE config 5 as `bench.py --gpus 4 --batch 2048 --contacts 3` shards it: 4 x 2 048 instances, every one."""
    worst, not_ok, total = _bench_shards(4, 2048, "standing", 3)
    print(f"cfg5, all 4 bench shards (oracle extension): {total} instances, {not_ok} not ok, max rel force err vs qpOASES {worst:.2e}")
    assert total == 8192 and not_ok == 0 and worst < TOL
