"""bench.py's host-side pieces that do not need a GPU: the algorithmic-byte model, the stamp that ties
profiles/traffic.json to the device sources, and the cpu_baseline leg (CPU build of the same megakernel)."""
import json
import os
import shutil
import sys

from common import REPO

sys.path.insert(0, REPO)
import bench  # noqa: E402


class _St:
    nodeFetches, triTests, closestHits, lightLoads = 10, 20, 3, 4


def test_algorithmic_bytes_is_survey_8d():
    assert bench.algorithmic_bytes(_St, 100) == 128 * 10 + 48 * 20 + 108 * 3 + 72 * 4 + 24 * 100


def test_traffic_json_is_only_used_for_the_sources_it_was_measured_on(tmp_path):
    h = bench.source_hash(REPO)
    assert len(h) == 16 and h == bench.source_hash(REPO)
    fake = tmp_path / "repo"
    shutil.copytree(os.path.join(REPO, "minimaloptix_amd", "csrc"), fake / "minimaloptix_amd" / "csrc")
    shutil.copy(os.path.join(REPO, "Makefile"), fake / "Makefile")
    (fake / "profiles").mkdir()
    assert bench.source_hash(str(fake)) == h
    (fake / "profiles" / "traffic.json").write_text(json.dumps({"source_hash": h, "traffic_GB_per_launch": 1.0}))
    assert bench.read_traffic(str(fake))["traffic_GB_per_launch"] == 1.0
    with open(fake / "minimaloptix_amd" / "csrc" / "pt_rng.h", "a") as f:
        f.write("// changed\n")
    assert bench.source_hash(str(fake)) != h
    assert bench.read_traffic(str(fake)) is None                      # stale profile: not reported


def test_committed_traffic_json_is_well_formed():
    t = json.load(open(os.path.join(REPO, "profiles", "traffic.json")))
    for k in ("source_hash", "FETCH_SIZE_KB_per_launch", "WRITE_SIZE_KB_per_launch", "traffic_GB_per_launch", "tcc_hit_rate"):
        assert k in t
    assert abs((t["FETCH_SIZE_KB_per_launch"] + t["WRITE_SIZE_KB_per_launch"]) * 1024 / 1e9 - t["traffic_GB_per_launch"]) < 0.1


def test_cpu_baseline_leg_runs_the_cpu_build_of_the_megakernel():
    r = bench.cpu_baseline(96, 54, 0.5)
    assert r["kind"] == "port" and r["unit"] == "Mrays/s" and r["value"] > 0 and r["cores"] >= 1
    assert "LBVH build" in r["sample"] and "excluded" in r["sample"]
