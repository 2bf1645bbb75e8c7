"""bench.py's one JSON line at N = 1 (a small workload, every leg on): the keys the driver and the judge read."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_bench_line_carries_the_contract():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--points", "300000", "--frames", "16", "--steps", "3", "--warmup", "1",
           "--roofline-points", "4000000", "--roofline-launches", "8", "--mls-points", "300000", "--cpu-points", "100000",
           "--cpu-frames", "4"]
    proc = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, proc.stdout[-2000:]  # stdout carries the line and nothing else
    line = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["warmup"] == 1 and line["higher_is_better"] is True
    assert line["vs_baseline"] is None and line["data"] == "synthetic" and line["scaling"] == "weak"
    assert "workload" in line["config"] and "model" not in line["config"]
    assert line["value"] > 0 and abs(line["value"] - 300000 * 16 / (line["ms_per_step"] * 1e-3) / 1e6) <= 0.02 * line["value"]
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    cpu = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cpu, key
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0
    # the two boundaries side by side, and the side legs
    assert line["value_resident"] == line["value"]
    assert line["value_host_images"] == line["host_images"]["value"] > 0
    assert line["host_images"]["pcie_floor_ms"] > 0 and line["camera_ref"]["value"] > 0
    assert line["mls"]["value"] > 0 and line["mls"]["sor_mls_sor"]["outputs"] > 0 and line["nid"]["valid"] is True
    # hidden_points_removal, the cull the reference binary runs: whole hull pass, and keyframe 0 against the oracle's quickhull
    assert line["hpr"]["hull_pass_s"] > 0 and line["hpr"]["cpu_baseline"]["equal_to_gpu"] is True
    # the timed region: an untimed settle phase in front of it, per-step statistics inside it
    assert line["settle"]["steps"] > 0 and line["settle"]["ms"] >= 400.0
    st = line["step_ms"]
    assert 0 < st["min"] <= st["median"] <= st["max"] and st["host_enqueue_median"] > 0
    assert "pruned" in line["value_note"]
    # the reference's own upsampling configuration (1 mm x 4): whole chain on a sub-sample, whole map in chunks
    assert line["mls"]["reference_config_chain"]["outputs"] > 0
    rs = line["mls"]["reference_config_stream"]
    assert rs["voxels"] >= rs["outputs"] > 0 and rs["chunks"] >= 1
    # round 5: the streamed chain on the whole map (kept == what the begin call reported), the command line end to end with its
    # own phase split, and the counter summaries' build check (a summary of another library is dropped, never divided by this
    # run's durations)
    wm = line["mls"]["reference_config_chain_whole_map"]
    assert wm["rows_before_last_filter"] > wm["outputs"] == wm["kept_reported"] > 0 and wm["chunks"] >= 1
    assert wm["min_margin_mm"] > wm["max_displacement_mm"] >= 0.0
    cli = line["cli_e2e"]
    for form in ("skip_filtered_dumps_off", "skip_filtered_dumps_on", "cull_hpr_skip_filtered_dumps_on"):
        assert cli[form]["wall_s"] > 0 and "images_decode_and_upload_wall_s" in cli[form]["phases_s"], cli
    assert "roofline kernel" in line["value_note"] or "k_project_frame" in line["value_note"]
    prof = line["profiles"]
    assert len(prof["lib_sha256"]) == 64
    for name, st_ in prof["summaries"].items():
        assert "stale" in st_, name
    if prof["summaries"].get("r05_pmc.json", {}).get("stale", True):
        assert line["roofline"]["traffic"] is None and line["roofline"]["traffic_stale"] is True
    assert ("kernels_ms" in line) != ("kernels_ms_timing_pass" in line)
