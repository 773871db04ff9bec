"""bench.py's host-side pieces that do not need a GPU: the algorithmic-byte model, the stamp that ties
profiles/traffic.json to the device sources, and the cpu_baseline leg (CPU build of the same megakernel)."""
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from common import REPO

sys.path.insert(0, REPO)
import bench  # noqa: E402


class _St:
    nodeFetches, triTests, closestHits, lightLoads = 10, 20, 3, 4


def test_algorithmic_bytes_is_survey_8d():
    assert bench.algorithmic_bytes(_St, 100) == 64 * 10 + 48 * 20 + 108 * 3 + 72 * 4 + 24 * 100


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


def test_gather_ceilings_come_from_the_committed_micro_benchmark_output():
    c = bench.gather_ceilings(REPO)
    assert c is not None, bench.GATHER_CEILING_FILE
    # per-lane 16-byte gathers are bounded by one L1 tag lookup per clock per CU: 64 B x 256 CUs x 2.4 GHz = 9.8 TB/s for
    # whole-line records; a 64-byte record is half a line and does better per byte
    assert 5000 < c["l2_128_at_12_waves"] <= c["l2_128"] < 9900 < c["l2_64"] < 13000
    traffic = {"traffic_GB_per_launch": 829.4, "tcc_hit_rate": 0.63}
    r = bench.roofline_block(5000.0, 440.0, 3, 2.2e12, 800.0, 2.77e9, 1.0, "k", traffic, c)
    # head of the block = the contract's form: SURVEY 8(d) algorithmic bytes per second against the HBM peak; without a PMC pass
    # for the device code there is no valu_issue part
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["frac"] == 0.625 and r["achieved"] == 5000.0
    assert r["valu_issue"] is None and r["gather_frac"] == round(5000.0 / c["l2_64"], 4) and r["traffic"] == 829.4
    assert abs(r["hbm_frac"] - 829.4 / 0.44 / 8000.0) < 1e-3
    assert "achieved" in r["live_fields"] and "traffic" in r["replayed_fields"]


def test_valu_issue_is_priced_against_the_measured_ceiling():
    """the vector-ALU peak is a number a file under profiles/ contains (tools/micro/valu_issue.hip on
    MI355X), read by bench.py exactly as the gather ceiling is; C2's FP32 peak comes from the same file."""
    v = bench.valu_ceilings(REPO)
    assert v is not None, bench.VALU_CEILING_FILE
    fma, mn = v["v_fma_f32"], v["v_min_f32"]
    assert set(fma) >= {1, 2, 3, 4, 8}
    # the guide's "2 cycles at >= 2 waves per SIMD, 4 for one wave alone" holds for fma / mul / add (1,229 G/s at 2.4 GHz would be
    # exactly 2): the chip delivers 750-950 because it lowers its clock under that load; one wave alone half of that
    assert 700 < fma[3] < 1300 and fma[1] < 0.65 * fma[2] and max(fma.values()) < 1300
    # min / max / convert / compare / select issue at half the fma rate whatever the occupancy
    for k in ("v_min_f32", "v_max_f32", "v_cvt_f32_ubyte0", "v_cmp_lt_f32(vcc)", "v_cndmask_b32_e64(sgpr)", "v_min_f64"):
        assert 0.45 * 1229 * 0.9 < v[k][3] < 0.55 * 1229, (k, v[k])
    assert 250 < v["v_rcp_f32"][3] < 320                                      # quarter rate
    assert abs(bench.fp32_vector_peak_tflops(REPO) - fma[8] * 0.128) < 1e-9 and 90 < bench.fp32_vector_peak_tflops(REPO) < 160
    c = bench.gather_ceilings(REPO)
    traffic = {"traffic_GB_per_launch": 600.0, "tcc_hit_rate": 0.65, "valu_active_frac": 0.74,
               "SQ": {"SQ_INSTS_VALU": 1.5e11, "SQ_ACTIVE_INST_VALU": 1.53e11, "SQ_THREAD_CYCLES_VALU": 6.7e12}}
    r = bench.roofline_block(7000.0, 320.0, 3, 2.2e12, 800.0, 2.77e9, 1.0, "k", traffic, c, 64, v)
    vi = r["valu_issue"]
    assert r["bound"] == "hbm" and r["frac"] == 0.875
    assert vi["peak"] == round(fma[3], 1) and os.path.basename(bench.VALU_CEILING_FILE) in vi["peak_source"] and bench.VALU_CEILING_FILE >= os.path.join("profiles", "r05_valu_ceiling.txt") and vi["peak_half_rate_class"] == round(mn[3], 1)
    assert abs(vi["achieved"] - 1.5e11 / 0.32 / 1e9) < 0.1 and abs(vi["frac"] - vi["achieved"] / fma[3]) < 1e-3
    assert abs(vi["lane_utilisation"] - 6.7e12 / 64 / 1.53e11) < 1e-3 and abs(vi["useful_lane_frac"] - vi["frac"] * vi["lane_utilisation"]) < 1e-3


def test_gpus_2_without_devices_fails_with_a_device_count_not_a_usage_message():
    # `python bench.py --gpus 2` as typed (no WORLD_SIZE): the launcher path; this container has no GPU
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.device_count() < 2:
        assert p.returncode != 0 and "2 devices needed" in p.stderr, p.stderr[-500:]


@pytest.mark.parametrize("split", ["tile", "sample"])
def test_gpus_2_launcher_path_on_cpu(tmp_path, split):
    """bench.launch_workers -> torch.distributed.run -> 2 ranks of bench.run_rank over gloo, with the CPU build of the
    per-lane code standing in for the GPU renderer (tests/bench_stub_worker.py): one JSON line from rank 0, the gathered
    frame equal to the single-rank frame."""
    from common import M, hostsim_render
    w, h, spp = 40, 24, 4
    frame_file = str(tmp_path / "frame.npy")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["BENCH_STUB_FRAME"] = frame_file
    env["OMP_NUM_THREADS"] = "2"
    argv = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--width", str(w), "--height", str(h), "--spp", str(spp),
            "--scene", "spheres", "--no-cpu-baseline", "--split", split]
    out_file = tmp_path / "out.txt"
    code = ("import sys; sys.path.insert(0, %r); import bench; sys.exit(bench.launch_workers(2, %r, script=%r))"
            % (REPO, argv, os.path.join(REPO, "tests", "bench_stub_worker.py")))
    with open(out_file, "w") as fh:
        rc = subprocess.call([sys.executable, "-c", code], env=env, stdout=fh, stderr=subprocess.STDOUT, timeout=600)
    text = out_file.read_text()
    assert rc == 0, text[-2000:]
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert len(lines) == 1, text[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["value"] > 0 and d["scaling"] == "strong"
    assert d["config"]["split"] == split and "STUB" in d["data"]
    ref, _ = hostsim_render(M.HostScene("spheres", w, h), M.launch_seeds(spp))
    got = np.load(frame_file)
    if split == "tile":
        assert np.array_equal(got, ref)
    else:
        assert np.allclose(got, ref, rtol=0, atol=2e-6 * spp)


def _stub_two_ranks(tmp_path, extra_env, split="tile"):
    w, h, spp = 40, 24, 2
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["BENCH_STUB_FRAME"] = str(tmp_path / "frame.npy"); env["OMP_NUM_THREADS"] = "2"
    env.update(extra_env)
    argv = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--width", str(w), "--height", str(h), "--spp", str(spp), "--scene", "spheres", "--no-cpu-baseline", "--split", split]
    code = ("import sys; sys.path.insert(0, %r); import bench; sys.exit(bench.launch_workers(2, %r, script=%r))"
            % (REPO, argv, os.path.join(REPO, "tests", "bench_stub_worker.py")))
    out_file = tmp_path / "out.txt"
    with open(out_file, "w") as fh:
        rc = subprocess.call([sys.executable, "-c", code], env=env, stdout=fh, stderr=subprocess.STDOUT, timeout=600)
    text = out_file.read_text()
    return rc, [json.loads(l) for l in text.splitlines() if l.startswith("{")], text


def test_two_ranks_on_cpu_both_modes_in_one_line(tmp_path):
    """bench.run_rank's round-6 control flow on CPU (gloo, two ranks, the stand-in renderer): both modes timed in one run, K steps each; the line
    carries both and reports the better one."""
    rc, lines, text = _stub_two_ranks(tmp_path, {"BENCH_STUB_MODES": "2"})
    assert rc == 0 and len(lines) == 1, text[-2000:]
    d = lines[0]
    modes = d["config"]["modes"]
    assert set(modes) == {"one_frame", "two_in_flight"} and all(m["ms_per_frame"] > 0 for m in modes.values())
    assert d["config"]["mode"] == min(modes, key=lambda m: modes[m]["ms_per_frame"]) and d["pmc_live"] is False
    assert d["config"]["communicator"] == "stub, non-blocking" and d["config"]["communicator_note"] is None


def test_two_ranks_on_cpu_a_failing_second_mode_leaves_the_first_standing(tmp_path):
    rc, lines, text = _stub_two_ranks(tmp_path, {"BENCH_STUB_MODES": "2", "BENCH_STUB_FAIL_MODE_B": "1"})
    assert rc == 0 and len(lines) == 1, text[-2000:]
    d = lines[0]
    assert d["config"]["mode"] == "one_frame" and d["value"] > 0
    assert "rank 1" in d["config"]["modes"]["two_in_flight"]["error"] and "ms_per_frame" in d["config"]["modes"]["one_frame"]


def test_two_ranks_on_cpu_preflight_falls_back_once_then_gives_up_with_a_line(tmp_path):
    """A pre-flight collective that fails on one rank: every rank makes the other kind of communicator once (the note says so); if that fails too,
    rank 0 prints ONE line with "error" and each rank's diagnosis and the launcher exits non-zero."""
    rc, lines, text = _stub_two_ranks(tmp_path, {"BENCH_STUB_FAIL_PREFLIGHT": "once"})
    assert rc == 0 and len(lines) == 1, text[-2000:]
    assert "pre-flight failed with the non-blocking communicator" in lines[0]["config"]["communicator_note"] and lines[0]["config"]["communicator"] == "stub, blocking"
    rc, lines, text = _stub_two_ranks(tmp_path, {"BENCH_STUB_FAIL_PREFLIGHT": "always"})
    assert rc != 0 and len(lines) == 1, text[-2000:]
    d = lines[0]
    assert d["value"] is None and "pre-flight" in d["error"] and d["config"]["ranks"]["diagnosis"][0].endswith("ok") and "rank 1" in d["config"]["ranks"]["diagnosis"][1]
