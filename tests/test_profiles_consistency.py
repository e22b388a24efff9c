"""The committed PMC summary that bench.py reports as roofline.traffic only counts for the build it was taken on
(bench.py ignores it otherwise and prints null): keep it in step with the sources.  If this fails after a change under
hector_simulation_amd/csrc or include/, re-run `gpurun -- 'bash scripts/gpu_rocprof.sh <round>'` and
`python scripts/summarize_rocprof.py <round>` (scripts/README.md)."""
import json
import os

from hector_simulation_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_hbm_traffic_summary_belongs_to_this_build():
    tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    assert tj["source_hash"] == build.source_hash(), "profiles/hbm_traffic.json was taken on another build of the library"
    assert tj["bytes_per_solve"] > 0 and tj["horizon"] == 10 and tj["gait"] == "standing"
    prof = os.path.join(ROOT, tj["profile_dir"])
    for name in ("kernel_stats.csv", "pmc.csv", "bench_standing.json", "README.md"):
        assert os.path.exists(os.path.join(prof, name)), name
