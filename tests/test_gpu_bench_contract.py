"""bench.py's JSON line: the keys and types the driver and the judge read (the measurement contract), on the headline
configuration with a short CPU-baseline sample."""

import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_default_bench_line_carries_the_contract_fields():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--cpu-seconds", "3", "--no-alone"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly ONE JSON line on stdout"
    d = json.loads(lines[0])
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                     ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[key], typ), key
    assert "vs_baseline" in d and d["vs_baseline"] is None  # BASELINE.md holds no published number for this metric
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 2 and d["unit"] == "MP/s" and d["higher_is_better"] is True
    assert "100MP" in d["metric"].replace(" ", "") or "100 MP" in d["metric"]
    assert d["config"]["workload"].startswith("cfg4_100mp") and "model" not in d["config"]
    # value = whole-job throughput: the frame's megapixels over the step time
    assert d["value"] == pytest.approx(12288 * 8192 / 1e6 / (d["ms_per_step"] * 1e-3), rel=1e-6)
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and "traffic" in r
    assert r["peak"] == 8000.0 and r["frac"] == pytest.approx(r["achieved"] / r["peak"])
    assert r["achieved"] == pytest.approx(24.0 * 12288 * 8192 / (d["ms_per_step"] * 1e-3) / 1e9, rel=1e-6)  # 24 B/px
    if r["traffic"] is not None:  # only from a capture of these very sources
        assert r["traffic_provenance"]["match"] is True and r["traffic"] > 24.0 * 12288 * 8192
    k = r["dominant_kernel"]
    assert k["bound"] == "hbm" and 0.0 < k["frac"] < 1.0 and k["kernel_ms"] > 0
    assert "fft_cols_walk_kernel" in k["kernel"] or "tail_kernel" in k["kernel"]  # (the largest per-step class of the two: run dependent)
    # the headline frame's exposure spans max / min = 1.4e5: beyond the 12-byte element's guard (6.2e4 with the stand-in Portra curve:
    # the searched constant) -- but the choice is made per WINDOW PAIR, and a window of the noise frame spans ~2e4: nearly every pair
    # takes the element; and the eager stage-by-stage steps behind the breakdown made the same choices from the same record
    e = r["halation_scratch_element"]
    assert e["armed"] and 0 < e["min"] < 1e-2 and 10 < e["max_abs"] < 1e3 and 3e4 < e["bound"] < 1e5
    assert e["max_abs"] > e["bound"] * max(e["min"], e["floor"])
    assert e["pairs"] > 500 and 0.9 * e["pairs"] <= e["packed_pairs"] <= e["pairs"]
    assert e["eager_breakdown_took_the_same_element"] is True
    if "fft_cols_walk_kernel" in k["kernel"]:
        assert "fft_cols_walk_kernel<32, 3>" in k["kernel"] or "fft_cols_walk_kernel<32, 1>" in k["kernel"]  # instances the timed steps launch
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "MP/s" and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str)
    assert d["value"] / c["value"] > 100  # a reported ratio, not the target -- but the GPU path must not be the CPU path
