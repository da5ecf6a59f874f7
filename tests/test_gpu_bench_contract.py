"""bench.py prints ONE JSON line with the fields the driver reads (metric/value/unit/n_gpus/steps/warmup/ms_per_step/
higher_is_better/scaling/vs_baseline/dtype/data/config) plus the roofline and cpu_baseline objects."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_bench_json_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--reads", "2000", "--steps", "2", "--warmup", "1",
                          "--cpu-sample", "40"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "reads/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 2000 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-6
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0
    assert c["mismatched_fields"] == 0 and len(c["sample"]) <= 100
    assert c["affinity_cpus"] >= 1 and "cgroup_cpu_max" in c
    # the driver keeps the TAIL of the line: every single-GPU configuration's [reads/s, roofline.frac, whole_path_frac] ends it
    tail = lines[0][-2000:]
    assert '"summary"' in tail, tail[:200]
    sm = json.loads(tail[tail.index('"summary"') + len('"summary": '):-1])
    for k in ("headline", "cnn_200k", "cnn_200k_f32_stack", "cnn_default", "pareto", "llr_default_window"):
        assert k in sm and sm[k][0] > 0, k
        assert d["secondary"][k]["roofline"]["frac"] == pytest.approx(sm[k][1], abs=1e-4) if k != "headline" else True
        # no fraction of a peak above 1 anywhere on the line
        assert 0 < sm[k][1] <= 1.0 and 0 < sm[k][2] <= 1.0, (k, sm[k])
        if k != "headline":
            for name, v in d["secondary"][k]["roofline"].items():
                if name.endswith("frac"):
                    assert v <= 1.0, (k, name, v)
    # the PCIe-inclusive runs: [reads/s, GB/s across PCIe] each, below what PCIe Gen5 x16 can carry
    hp = sm["host_pipeline"]
    for k in ("f32_padded", "int16_padded", "int16_ragged_pareto"):
        assert hp[k] is not None and hp[k][0] > 0 and 0 < hp[k][1] < 64.0, (k, hp[k])
        assert d["secondary"]["host_pipeline"][k]["minibatches"] == 12
    assert hp["f32_padded"][0] < sm["headline"][0]  # (never the headline)
    assert len(json.dumps(sm)) < 1400 and "grouped" not in d["secondary"] and "int16" not in sm
