"""The bench line's contract, checked on the newest committed line (profiles/rNN_bench_steps20.json =
`python bench.py --gpus 1 --steps 20 --warmup 5` on an MI355X): the keys the driver and the judge read, their units and the
arithmetic that ties them together.  No GPU needed -- it reads the recorded line; bench.py's own control flow is covered by
tests/test_distributed_cpu.py with the solver stubbed."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    """The newest committed `rNN_bench_steps20.json` (scripts/profile_all.sh writes one per round)."""
    import bench

    tag = bench.profile_tags("bench_steps20.json")[0]
    path = os.path.join(ROOT, "profiles", f"{tag}_bench_steps20.json")
    return json.loads(open(path).read().strip().splitlines()[-1])


def test_recorded_bench_line_keeps_the_contract():
    import bench

    d = _line()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "sim steps/s" and d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    assert not any(k in d["config"] for k in ("model", "seq_len", "global_batch"))
    E = d["config"]["episodes_per_gpu"]
    assert d["value"] == pytest.approx(E * d["steps"] / (d["ms_per_step"] * d["steps"] * 1e-3), rel=1e-9)
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == bench.HBM_PEAK_GBS
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and r["saturated"] == (r["frac"] > 1.0)
    assert r["algorithmic_bytes_per_launch"] == bench.BYTES_PER_STEP * E == bench.algorithmic_bytes_per_step(64, 64) * E
    assert r["achieved"] == pytest.approx(r["algorithmic_bytes_per_launch"] / (r["kernel_ms_per_launch"] * 1e-3) / 1e9, rel=1e-9)
    assert r["kernel_ms_per_launch"] <= d["ms_per_step"] and r["traffic"] is not None and r["traffic"] < 0.2 * r["algorithmic_bytes_per_launch"]
    # the physical bound leads the line
    keys = list(d)
    assert keys.index("valu_roofline") < keys.index("roofline") and d["roofline_that_bounds_the_kernel"] == "valu_roofline"
    v = d["valu_roofline"]
    assert v["bound"] == "valu" and 0.5 < v["frac"] < 1.0 and v["frac"] == pytest.approx(v["achieved"] / v["peak"])
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample", "single_thread", "all_cores", "arithmetic"):
        assert key in c, key
    assert c["kind"] == "port" and c["value"] == c["all_cores"]["value"] and c["cores"] == c["all_cores"]["cores"] >= 1
    assert 0 < c["single_thread"]["value"] <= c["value"] and "ORC_EXACT_RSQRT" in c["arithmetic"]
    assert d["parity"]["bit_exact"] is True and d["parity_checked"] is True
    by_key = {e.get("key"): e for e in d["configs"]}
    for k in ("c2_fling_256", "c2_fling_64"):
        assert by_key[k]["parity"]["bit_exact"] is True and 0 < by_key[k]["roofline_frac_equivalent"] < 1
    assert d["fling_phase_ratio"] == pytest.approx(by_key["c2_fling_256"]["ratio_to_crumpled_sheet"])
    assert 0 < d["eval_loop"]["roofline_frac_equivalent"] < 1 and 0 < d["eval_loop"]["continuous"]["roofline_frac_equivalent"] < 1


def test_quoted_hbm_traffic_is_the_newest_pmc_pass():
    """`roofline.traffic` of the bench line is NOT measured in the run: bench.py quotes profiles/hbm_traffic.json, the PMC
    pass of the same command, scaled per episode.  That makes it the one number of the line nobody re-measures -- so it must
    be the newest committed pass (profiles/rNN_pmc.json with the highest NN), byte for byte; scripts/summarize_profile.py
    writes both files in one go, and this test fails when a round commits a new PMC pass without the quoted figure (or edits
    the figure by hand)."""
    import bench

    tags = bench.profile_tags("pmc.json")
    assert tags, "no profiles/rNN_pmc.json"
    newest = json.load(open(os.path.join(ROOT, "profiles", f"{tags[0]}_pmc.json")))
    quoted = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    assert quoted["tag"] == tags[0] == newest["tag"], (quoted["tag"], tags[0])
    assert quoted["episodes"] == newest["episodes"]
    assert quoted["bytes_per_launch"] == pytest.approx(newest["hbm_bytes_per_launch"]["total_corrected"], rel=1e-12)
    h = newest["hbm_bytes_per_launch"]        # the guide's gfx950 correction: FETCH_SIZE counts half of what was read
    assert h["total_corrected"] == pytest.approx(2 * h["fetch_reported"] + h["write"], rel=1e-12)
    traffic, source = bench.traffic_from_profile(newest["episodes"])
    assert traffic == pytest.approx(quoted["bytes_per_launch"]) and tags[0] in source
    assert bench.limiter_from_profile()["source"] == f"profiles/{tags[0]}_pmc.json"


def test_algorithmic_bytes_follow_surveys_table():
    """63 524 608 B per pyflex.step() of a 64 x 64 cloth, 15 576 832 B for 32 x 32; spring counts of section 8's
    table (23 938 / 5 826)."""
    import bench

    assert bench.algorithmic_bytes_per_step(64, 64) == 63524608 and bench.algorithmic_bytes_per_step(32, 32) == 15576832
    for dim, m in ((64, 23938), (32, 5826)):
        n = dim * dim
        assert bench.algorithmic_bytes_per_step(dim, dim) == 4 * (112 * n + 30 * (32 * n + 16 * m))


def test_eval_loop_limiter_reads_the_committed_counters():
    """The evaluation-loop entry of the bench line names what bounds its launches from the committed PMC pass of the loop-less
    reconstruction (profiles/rNN_eval_shapes.txt, case A): VALU per wave of the iterate / search / boundary kernels and the
    iterate kernel's VALU-busy fraction against the 0.2 ceiling of five waves per SIMD."""
    import bench

    lim = bench.eval_limiter_from_profile()
    assert lim and lim["bound"] == "valu-issue" and lim["source"].startswith("profiles/r")
    k = lim["kernels"]
    assert {lim["iterate_kernel"], "fs_k_find_neighbors", "fs_k_boundary"} <= set(k)
    assert 300 < k[lim["iterate_kernel"]]["valu_per_wave"] < 1500 and 0.5 < lim["iterate_valu_busy_fraction"] <= 1.05
