"""The several-ranks code paths on RCCL (backend "nccl"), executed on ONE GPU as a process group of one: RCCL initialises in
the process that also holds libadapted_hip.so (one HIP runtime mapped, adapted_amd.lib.hip_runtimes), the row gather / barrier /
all-reduce of bench.py's N > 1 branch and of the CLI run on device tensors, and the results equal the plain run's.  The curve
over 2/4/8 GPUs is the driver's to measure; this pins that the branch executes at all (reference: the process pool of
adapted/file_proc.py:738-784 is the only parallelism there)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _keep(name, text):
    d = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, name), "w") as fh:
            fh.write(text)
    except OSError:
        pass


def _bench(extra_env, launcher):
    args = ["--gpus", "1", "--reads", "4000", "--steps", "3", "--warmup", "1", "--cpu-sample", "0", "--no-secondary"]
    cmd = [sys.executable]
    if launcher:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(_port())]
    cmd += [os.path.join(ROOT, "bench.py")] + args
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), **extra_env)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, out.stdout[-2000:]
    if not launcher:  # (torchrun adds nothing to a rank's stdout either, but only the plain form is promised to be bare)
        # ONE line on stdout, nothing else: RCCL's version banner (printed on stdout when the first communicator comes up) goes to stderr
        assert [ln for ln in out.stdout.splitlines() if ln.strip()] == lines, out.stdout[:500]
    return json.loads(lines[0])


def test_bench_rccl_branch_on_a_group_of_one():
    plain = _bench({}, launcher=False)
    grouped = _bench({"ADP_BENCH_FORCE_DIST": "1", "ADP_BENCH_BACKEND": "nccl"}, launcher=True)
    bare = _bench({"ADP_BENCH_FORCE_DIST": "1", "ADP_BENCH_BACKEND": "nccl"}, launcher=False)  # (its own rendezvous; stdout checked bare)
    assert bare["rows_sha256"] == plain["rows_sha256"] and bare["dist"]["backend"] == "nccl"
    _keep("r04_rccl_group_of_one_bench.json", json.dumps({"plain": plain, "rccl_group_of_one": grouped}, indent=1))
    d = grouped["dist"]
    assert d["backend"] == "nccl" and d["world_size"] == 1 and d["forced_group_of_one"] is True
    assert len(d["hip_runtimes"]) == 1, d["hip_runtimes"]            # RCCL and the library share ONE runtime
    assert grouped["gathered_block_equals_local_rows"] is True       # dist.gather delivered the rows the kernels wrote
    assert grouped["rows_sha256"] == plain["rows_sha256"]            # and they are the plain run's rows, byte for byte
    assert grouped["n_gpus"] == 1 and grouped["scaling"] == "weak" and grouped["cpu_baseline"] is None
    assert grouped["config"]["pass_rate"] > 0.9


def test_cli_rccl_gather_on_a_group_of_one_equals_plain_run(tmp_path):
    from adapted_amd import synth
    from adapted_amd.config import get_chemistry_specific_config

    spc = get_chemistry_specific_config("RNA004")
    spc.llr_boundaries.llr_detect = True
    spc.cnn_boundaries.cnn_detect = False
    spc.update_primary_method()
    spc.update_sig_preload_size()
    m = spc.sig_preload_size
    mb, n = 8, 8 * 4 * 3 + 5
    lens = np.array([m if i % 4 else synth.pareto_length(3, i) for i in range(n)], dtype=np.int32)
    sig, _ = synth.synth_batch(37, 0, n, m, lens)
    for j in range(24):
        sig[4, 120 + 40 * j: 123 + 40 * j] = 260.0  # an open_pores list longer than a row holds travels beside the rows
    ids = np.array(["read_%04d" % i for i in range(n)], dtype=object)
    np.savez(tmp_path / "reads_0.npz", signals=sig, full_lengths=lens, read_ids=ids)
    cfg = tmp_path / "cfg.toml"
    spc.to_toml(str(cfg))

    def run(out, grouped):
        cmd = [sys.executable]
        env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        if grouped:
            cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", str(_port())]
            env.update(ADAPTED_DIST_FORCE="1", ADAPTED_DIST_BACKEND="nccl")
        cmd += ["-m", "adapted_amd.main", "detect", "-i", str(tmp_path / "reads_0.npz"), "-o", str(out), "--config", str(cfg), "-s", str(mb), "-b", "50"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        runs = [d for d in os.listdir(out) if d.startswith("adapted_")]
        assert len(runs) == 1, runs
        files = {}
        for sub in ("boundaries", "failed_reads"):
            d = out / runs[0] / sub
            for f in sorted(os.listdir(d)) if d.exists() else []:
                files[sub + "/" + f] = (d / f).read_text()
        return files, r.stdout

    one, _ = run(tmp_path / "plain", False)
    two, log = run(tmp_path / "rccl", True)
    _keep("r03_rccl_group_of_one_cli.log", log[-4000:])
    assert "process group: backend nccl, 1 rank(s)" in log
    line = [ln for ln in log.splitlines() if "HIP runtimes mapped" in ln][-1]
    assert line.count("libamdhip64") == 1, line
    assert one.keys() == two.keys() and len(one) >= 1
    for k in one:
        assert one[k] == two[k], k
