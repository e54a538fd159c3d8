"""The final stdout line of bench.py stays small enough for the driver to parse (a 27 KB line gave
``parsed: null``).  bench.compact_line is a pure function of the full result object; it is run here on the committed full
object of round 4 (profiles/r4d_bench_line.json: canned numbers) and on a stripped result (legs skipped)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402

CONTRACT_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "per_rank_ms_per_step", "dist"}


def _canned():
    with open(os.path.join(ROOT, "profiles", "r4d_bench_line.json")) as fh:
        return json.load(fh)


def test_compact_line_size_and_keys():
    full = _canned()
    assert len(json.dumps(full)) > 20000                       # the object that broke the driver's parser
    line = bench.compact_line(full, ["bench_detail.json"])
    assert "\n" not in line
    assert len(line) < bench.LINE_CAP_BYTES
    assert len(line) <= bench.LINE_TARGET_BYTES
    out = json.loads(line)
    assert CONTRACT_KEYS <= set(out)
    assert out["metric"] == full["metric"] and out["unit"] == "TOPS"
    assert abs(out["value"] - full["value"]) / full["value"] < 1e-5
    assert abs(out["ms_per_step"] - full["ms_per_step"]) / full["ms_per_step"] < 1e-5
    assert out["config"]["workload"].startswith("c2")
    rf = out["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "step_hbm"):
        assert k in rf, k
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    # the number carries its own verdict against north_star's target
    assert rf["step_hbm"]["target"] == 0.70 and rf["step_hbm"]["target_met"] is False
    assert rf["step_hbm"]["ceiling_claimed"] == 0.44
    cb = out["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cb) and cb["kind"] == "port"
    assert out["alexnet"]["images_per_s"] > 0 and out["alexnet"]["cpu_images_per_s"] > 0
    assert out["alexnet"]["xnor_images_per_s"] > 0
    assert "c4" in out["extra"] and "c5" in out["extra"]


def test_compact_line_without_optional_legs():
    full = _canned()
    for k in ("alexnet", "extra", "cpu_baseline", "reference_ops_on_gpu", "parity_vs_cpu_port"):
        full.pop(k, None)
    out = json.loads(bench.compact_line(full))
    assert (CONTRACT_KEYS - {"cpu_baseline"}) <= set(out)
    assert "alexnet" not in out and "extra" not in out


def test_compact_line_sheds_optional_objects_before_the_cap():
    full = _canned()
    full["extra"]["c4_dorefa_resnet18_w1a4"]["module_graph"]["images_per_s"] = 1.0
    # a pathological leg name explosion: many training legs with long error strings
    for key in ("n2_training_step_alexnet_bin", "n2_training_step_alexnet_xnor", "n2_training_step_dorefa_resnet18_w1a4"):
        full["extra"][key] = {"error": "x" * 4000}
    line = bench.compact_line(full)
    assert len(line) <= bench.LINE_TARGET_BYTES
    out = json.loads(line)
    assert "extra" not in out and CONTRACT_KEYS <= set(out)


def test_multi_rank_line_fits():
    full = _canned()
    full["n_gpus"] = 8
    full["per_rank_ms_per_step"] = [0.0624901504488662 + i * 1e-4 for i in range(8)]
    line = bench.compact_line(full)
    assert len(line) <= bench.LINE_TARGET_BYTES
    assert len(json.loads(line)["per_rank_ms_per_step"]) == 8
