"""CPU: the bench line committed under profiles/ carries every field of the driver's contract, and bench.py's
argument surface / constants are what DESIGN.md and SURVEY.md §8d state."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest_bench_line():
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench_final.json")))
    assert files, "no committed bench line under profiles/"
    return json.load(open(files[-1])), os.path.basename(files[-1])


def test_round2_bench_line_fields():
    """Fields added in round 2: event timing beside the wall clock, executed training FLOPs,
    the traffic figure labelled as stored, CPU baselines for the train step and the crop."""
    line, name = _latest_bench_line()
    if name < "r02":
        import pytest
        pytest.skip("round-1 line")
    assert abs(line["ms_per_step_events"] - line["ms_per_step"]) / line["ms_per_step"] < 0.02
    assert "median" in line["timing"]
    t = line["train"]
    assert abs(t["ms_per_step_events"] - t["ms_per_step"]) / t["ms_per_step"] < 0.02
    assert 0.3 < t["tflops_executed_frac_of_fp32_mfma_peak"] < 1.0
    assert abs(t["value"] - t["global_batch"] / (t["ms_per_step"] * 1e-3)) / t["value"] < 1e-3
    r = line["roofline"]
    assert r["traffic"] is None or "STORED" in r["traffic_source"] or "measured" in r["traffic_source"]
    c = line["cpu_baseline"]
    assert c["train_step"]["value"] > 0 and c["crop"]["value"] > 0


def test_committed_bench_line_has_the_contract_fields():
    line, _ = _latest_bench_line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["unit"] == "grasps/s" and line["higher_is_better"] is True and line["scaling"] == "weak"
    assert line["vs_baseline"] is None and line["data"] == "synthetic" and "workload" in line["config"]
    assert "model" not in line["config"]
    r = line["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(r)
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = line["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(c) and c["kind"] in ("reference", "port")
    # value is whole-job throughput: batch x steps / time
    assert abs(line["value"] - 1024 * line["n_gpus"] / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-3


def test_flop_model_matches_survey():
    import bench
    # FLOPs fwd = N*557,842 + 2,627,072 (k=2) / 2,627,584 (k=3); trunk = 278,912 FLOP per point
    assert bench.flops_per_grasp(1024, 2) == 1024 * 557842 + 2627072
    assert bench.flops_per_grasp(1024, 3) == 1024 * 557842 + 2627584
    assert bench.FLOP_PER_POINT_TRUNK == 278912
    assert bench.PEAK_FP32_MFMA_TFLOPS == 157.3
