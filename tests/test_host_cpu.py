"""CPU-side tests of the host layer: config/TOML surface, CSV writer against the reference's
CSV text, C-ABI library load + exports, synthetic generator, sharding + gather (gloo, 2 ranks)."""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from util import GOLD

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))




def _free_port() -> str:
    """a free TCP port on 127.0.0.1 (bind to port 0), as bench.spawn_ranks picks its rendezvous port"""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return str(sk.getsockname()[1])

def test_presets_and_derived_fields():
    from adapted_amd.config import get_chemistry_specific_config

    r4 = get_chemistry_specific_config("RNA004")
    assert (r4.primary_method, r4.sig_preload_size) == ("cnn", 17500)
    assert r4.core.downscale_factor == 10 and r4.cnn_boundaries.polya_cand_k == 10
    r2 = get_chemistry_specific_config("RNA002")
    assert (r2.primary_method, r2.sig_preload_size) == ("llr", 26500)
    r4.core.max_obs_trace = 200000
    r4.update_sig_preload_size()
    assert r4.sig_preload_size == 201500
    r4.llr_boundaries.llr_detect = True
    with pytest.raises(ValueError, match="Exactly one primary method"):
        r4.update_primary_method()
    with pytest.raises(ValueError):
        get_chemistry_specific_config("RNA003")


def test_toml_roundtrip_and_unknown_keys(tmp_path):
    from adapted_amd.config import get_chemistry_specific_config, load_nested_config_from_file
    from adapted_amd.config import toml_io

    spc = get_chemistry_specific_config("RNA002")
    p = tmp_path / "config.toml"
    spc.to_toml(str(p))
    text = p.read_text()
    assert "pA_var_range = [ -inf, 20.0,]" in text
    back = load_nested_config_from_file(str(p))
    assert back.primary_method == "llr" and back.sig_preload_size == spc.sig_preload_size
    assert back.mvs_polya.pA_var_range == (-np.inf, 20.0)
    assert back.core.dict() == spc.core.dict()
    # the built-in parser (used when no toml library is importable) reads the same file
    d = toml_io._parse_builtin(text)
    assert d["core"]["max_obs_trace"] == 25000 and d["mvs_polya"]["median_shift_range"] == [5.0, np.inf]
    bad = tmp_path / "bad.toml"
    bad.write_text("[core]\nmax_obs_trace = 1\n[nonsense]\nx = 1\n")
    with pytest.raises(ValueError, match="Unknown key"):
        load_nested_config_from_file(str(bad))
    bad.write_text("[core]\nnot_a_key = 1\n")
    with pytest.raises(ValueError, match="Could not parse section"):
        load_nested_config_from_file(str(bad))


@pytest.mark.reference
def test_presets_equal_reference_tomls():
    ref = "/root/reference/adapted/config/config_files"
    if not os.path.isdir(ref):
        pytest.skip("reference not present")
    from adapted_amd.config import get_chemistry_specific_config, load_nested_config_from_file

    for chem, fn in (("RNA004", "rna004_130bps@v0.2.4.toml"), ("RNA002", "rna002_70bps@v0.2.4.toml")):
        a = get_chemistry_specific_config(chem)
        b = load_nested_config_from_file(os.path.join(ref, fn))
        for sec in ("core", "llr_boundaries", "mvs_polya", "real_range", "cnn_boundaries", "med_shift", "rna_start_peak"):
            assert getattr(a, sec).typed_dict() == getattr(b, sec).typed_dict(), (chem, sec)


def _results_from_golden(name):
    from adapted_amd.container_types import DetectResults, ReadResult

    with open(os.path.join(GOLD, name + ".rows.json")) as fh:
        g = json.load(fh)
    f32 = {"start_peak_pa", "start_peak_next_max_pa", "adapter_rna_median_shift"}
    out = []
    for i, r in enumerate(g["rows"]):
        d = DetectResults(success=r["success"])
        for k, v in r.items():
            if k == "success":
                continue
            if isinstance(v, list):
                v = np.array(v, dtype=np.int64)
            if k in f32 and v is not None:
                v = np.float32(v)
            setattr(d, k, v)
        out.append(ReadResult(read_id="read_%04d" % i, success=d.success, fail_reason=d.fail_reason, detect_results=d))
    return out


@pytest.mark.parametrize("name", ["rna002_llr_4k", "rna004_llr_default", "rna004_llr_mvs_overwrite_wide", "rna004_llr_open_pores", "rna004_cnn_default", "rna004_cnn_200k",
                                  "rna004_start_peak_blips", "rna004_llr_quantised", "rna004_llr_nan_holes"])
def test_csv_text_equals_reference(tmp_path, name):
    from adapted_amd.output import CSV_COLUMNS, save_detected_boundaries

    res = _results_from_golden(name)
    ok = [r for r in res if r.success]
    bad = [r for r in res if not r.success]
    p, f = tmp_path / "p.csv", tmp_path / "f.csv"
    save_detected_boundaries(ok, str(p), save_fail_reasons=False)
    save_detected_boundaries(bad, str(f), save_fail_reasons=True)
    with open(os.path.join(GOLD, name + ".pass.csv")) as fh:
        assert p.read_text() == fh.read()
    with open(os.path.join(GOLD, name + ".fail.csv")) as fh:
        assert f.read_text() == fh.read()
    header = p.read_text().splitlines()[0].split(",")
    assert header == CSV_COLUMNS and header[17] == "polya_truncated"  # scripts/get_truncated.sh: column 18


def test_abi_library_loads_and_exports_header_symbols():
    from adapted_amd import lib

    L = lib.load()
    assert L.adp_abi_version() == 3
    assert L.adp_sizeof_row() == lib.ROW_DTYPE.itemsize == 544
    assert L.adp_sizeof_cfg() == ctypes.sizeof(lib.AdpCfg)
    with open(os.path.join(ROOT, "include", "adapted_hip.h")) as fh:
        declared = set(re.findall(r"\b(adp_[a-z0-9_]+)\s*\(", fh.read()))
    assert declared == set(lib.EXPORTS)
    for sym in declared:
        assert hasattr(L, sym), sym


def test_no_cpu_fallback_without_gpu():
    """Without a GPU the operators must fail loudly, not fall back to anything."""
    from adapted_amd import lib

    L = lib.load()
    if L.adp_device_count() >= 1:
        pytest.skip("GPU present")
    from adapted_amd.config import get_chemistry_specific_config
    from adapted_amd.detect.combined import combined_detect_llr2

    spc = get_chemistry_specific_config("RNA002")
    with pytest.raises(lib.HipLibraryError):
        combined_detect_llr2(np.zeros((2, spc.sig_preload_size), np.float32), np.array([10, 10]), spc)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "adapted_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip")):
                with open(os.path.join(dirpath, f)) as fh:
                    src = fh.read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "liboracle" not in src, f


def test_nothing_of_the_reference_in_the_tree():
    """SURVEY 8(c) travel rule: the snapshot pushed to the GPU box (everything but .git/ and the paths of
    .gpurunignore) holds no reference-derived build output (pyximport's _c_llr.c quotes the reference's source)."""
    with open(os.path.join(ROOT, ".gpurunignore")) as fh:
        ignored = [ln.strip().rstrip("/") for ln in fh if ln.strip()]
    assert "oracle/_ref" in ignored
    from oracle import ref_harness

    assert os.path.commonpath([ref_harness.build_dir(), ROOT]) != ROOT
    for dirpath, dirs, files in os.walk(ROOT):
        rel = os.path.relpath(dirpath, ROOT)
        dirs[:] = [d for d in dirs if os.path.normpath(os.path.join(rel, d)) not in ignored]
        for f in files + dirs:
            # (adapted_amd/detect/_c_llr.py is this repository's own drop-in of that module's API; what must not be here is
            # the Cython source, the C file generated from it or the extension built from that)
            derived = f.startswith("_c_llr") and f.endswith((".c", ".so", ".pyx", ".pyd", ".o", ".html"))
            assert not derived and ".pyxbld" not in f, os.path.join(dirpath, f)


def test_synth_is_deterministic_and_shaped():
    from adapted_amd import synth

    a = synth.synth_read(3, 17, 20000)
    b = synth.synth_read(3, 17, 20000)
    assert a.dtype == np.float32 and np.array_equal(a, b)
    al, pl, *_ = synth.read_params(3, 17)
    assert abs(float(np.median(a[:al])) - 80) < 3 and abs(float(np.median(a[al:al + pl])) - 108) < 3
    c = synth.synth_read(3, 17, 20000, full_len=5000)
    assert np.isnan(c[5000:]).all() and np.array_equal(c[:5000], a[:5000])
    assert 10_000 <= synth.pareto_length(1, 5) <= 1_000_000


def test_shard_minibatches_cover_everything():
    from adapted_amd.parallel import shard_minibatches, shard_reads

    for n_mb in (1, 7, 8, 9, 64):
        for ws in (1, 2, 3, 8):
            got = [i for r in range(ws) for i in shard_minibatches(n_mb, ws, r)]
            assert got == list(range(n_mb))
    assert [shard_reads(9500, 1000, 4, r) for r in range(4)] == [(0, 3000), (3000, 6000), (6000, 8000), (8000, 9500)]


_GLOO_WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from adapted_amd import parallel
from adapted_amd.lib import ROW_DTYPE
dist.init_process_group("gloo")
rank, ws = dist.get_rank(), dist.get_world_size()
n_reads, mb = 7300, 1000
a, b = parallel.shard_reads(n_reads, mb, ws, rank)
rows = np.zeros(b - a, dtype=ROW_DTYPE)
rows["col"][:, 0] = np.arange(a, b)           # signal_len column carries the global read index
rows["success"] = 1
rows["cand"][:, 0] = np.arange(a, b) * 3
out = parallel.gather_rows(rows, dst=0)
if rank == 0:
    assert out.shape[0] == n_reads, out.shape
    assert np.array_equal(out["col"][:, 0], np.arange(n_reads))
    assert np.array_equal(out["cand"][:, 0], np.arange(n_reads) * 3)
    print("GATHER_OK", out.shape[0])
else:
    assert out is None
# a rank without any row (fewer minibatches than ranks) must not break the collective
a, b = parallel.shard_reads(500, mb, ws, rank)
rows = np.zeros(b - a, dtype=ROW_DTYPE)
rows["col"][:, 0] = np.arange(a, b)
out = parallel.gather_rows(rows, dst=0)
if rank == 0:
    assert out.shape[0] == 500 and np.array_equal(out["col"][:, 0], np.arange(500))
    print("EMPTY_RANK_OK", b - a)
else:
    assert b - a == 0 and out is None
dist.destroy_process_group()
"""


def test_row_gather_two_ranks_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_GLOO_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", _free_port(), str(script)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "GATHER_OK 7300" in r.stdout and "EMPTY_RANK_OK 500" in r.stdout


def test_minibatches_fill_caller_buffers_in_place(tmp_path):
    """io_utils.yield_minibatches(buffers=...) writes into the arrays the caller hands out (the pipeline's pinned staging
    slots) and asks for the next pair only when the next minibatch starts."""
    from adapted_amd import io_utils

    rng = np.random.default_rng(0)
    n, m = 7, 50
    lens = np.array([60, 50, 20, 55, 10, 70, 50], dtype=np.int32)
    sigs = np.full((n, 80), np.nan, dtype=np.float32)
    for i, L in enumerate(lens):
        sigs[i, :L] = rng.normal(90, 10, L)
    f = tmp_path / "a.npz"
    np.savez(f, signals=sigs, full_lengths=lens, read_ids=np.array(["r%d" % i for i in range(n)]))
    handed = []

    def buffers():
        pair = (np.zeros((3, m), dtype=np.float32), np.zeros(3, dtype=np.int32))
        handed.append(pair)
        return pair

    plain = list(io_utils.yield_minibatches([str(f)], set(), set(), 3, m))
    seen = 0
    for k, (sig, ln, ids) in enumerate(io_utils.yield_minibatches([str(f)], set(), set(), 3, m, buffers=buffers)):
        assert len(handed) == k + 1                      # one pair per minibatch, handed out lazily
        assert np.shares_memory(sig, handed[k][0]) and np.shares_memory(ln, handed[k][1])
        np.testing.assert_array_equal(sig, plain[k][0])
        np.testing.assert_array_equal(ln, plain[k][1])
        assert list(ids) == list(plain[k][2])
        seen += sig.shape[0]
    assert seen == n and len(handed) == 3


@pytest.mark.parametrize("mode", ["reader", "pool"])
def test_pod5_records_are_decoded_in_the_reader_thread_while_the_file_is_open(monkeypatch, mode):
    """By default the copy pool never touches a pod5 record: decoding happens in the thread that iterates the Reader and before
    the file is closed (groups that span two files, the final partial group).  A stand-in `pod5` whose records refuse any other
    use (wrong thread, closed reader) drives all three assemblers over three files.  ADAPTED_POD5_DECODE=pool (opt-in) hands the
    decode to the pool: other threads, but still never after the Reader has closed."""
    import sys
    import threading
    import types

    from adapted_amd import io_utils

    monkeypatch.setenv("ADAPTED_POD5_DECODE", mode)
    other_threads = set()
    m, per_file = 700, 5
    rng = np.random.default_rng(3)
    store = {"f%d.pod5" % f: [(("r%d_%d" % (f, i)), rng.integers(200, 900, int(rng.integers(300, 1000))).astype(np.int16))
                              for i in range(per_file)] for f in range(3)}

    class Cal:
        scale, offset = 0.25, -3.0

    class Rec:
        def __init__(self, owner, rid, raw):
            self._o, self.read_id, self._raw, self.num_samples, self.calibration = owner, rid, raw, raw.size, Cal()

        def _check(self):
            assert not self._o.closed, "record used after its Reader was closed"
            if mode == "reader":
                assert threading.get_ident() == self._o.thread, "record decoded outside the reader's thread"
            elif threading.get_ident() != self._o.thread:
                other_threads.add(threading.get_ident())

        @property
        def signal(self):
            self._check()
            return self._raw

        @property
        def signal_pa(self):
            self._check()
            return (np.float32(0.25) * (self._raw.astype(np.float32) + np.float32(-3.0))).astype(np.float32)

    class Reader:
        def __init__(self, fn):
            self.fn, self.closed, self.thread = os.path.basename(fn), False, threading.get_ident()

        def __enter__(self):
            return self

        def __exit__(self, *a):
            self.closed = True

        def reads(self, selection=None, missing_ok=True):
            for rid, raw in store[self.fn]:
                if selection is None or rid in selection:
                    yield Rec(self, rid, raw)

    monkeypatch.setitem(sys.modules, "pod5", types.SimpleNamespace(Reader=Reader))
    files = sorted(store)
    flat_all = [raw for f in files for _, raw in store[f]]
    ids_all = [rid for f in files for rid, _ in store[f]]
    N = 4  # groups of 4 over files of 5: every group but the first spans two files; 15 reads leave a partial group
    got_ids, k0 = [], 0
    for sig, lens, ids in io_utils.yield_minibatches(files, set(), set(), N, m, workers=4):
        for j in range(len(ids)):
            raw = flat_all[k0 + j]
            want = (np.float32(0.25) * (raw.astype(np.float32) + np.float32(-3.0)))[:m]
            assert np.array_equal(sig[j, :want.size], want) and np.isnan(sig[j, want.size:]).all() and lens[j] == raw.size
        got_ids += list(ids)
        k0 += len(ids)
    assert got_ids == ids_all
    k0 = 0
    for raw, lens, sc, of, ids in io_utils.yield_minibatches_i16(files, set(), set(), N, m, workers=4):
        for j in range(len(ids)):
            w = flat_all[k0 + j][:m]
            assert np.array_equal(raw[j, :w.size], w) and sc[j] == np.float32(0.25) and of[j] == np.float32(-3.0)
        k0 += len(ids)
    assert k0 == len(ids_all)
    for i16 in (False, True):
        bufs = []

        def buffers():
            b = (np.zeros(N * m, dtype=np.int16 if i16 else np.float32), np.zeros(N, np.int32), np.zeros(N + 1, np.int64),
                 np.zeros(N, np.float32), np.zeros(N, np.float32))
            bufs.append(b)
            return b

        k0 = 0
        for k, ids in io_utils.yield_minibatches_packed(files, set(), set(), N, m, buffers, int16=i16, workers=4):
            flat, lens, offs = bufs[-1][0], bufs[-1][1], bufs[-1][2]
            for j in range(k):
                raw = flat_all[k0 + j][:m]
                want = raw if i16 else (np.float32(0.25) * (raw.astype(np.float32) + np.float32(-3.0)))
                assert np.array_equal(flat[offs[j]:offs[j + 1]], want)
            k0 += k
        assert k0 == len(ids_all)
    assert bool(other_threads) == (mode == "pool")


def test_adapted_console_script_is_declared():
    """`adapted` (the reference's entry point, setup.py:49) resolves to this package's main()"""
    try:
        import tomllib
    except ImportError:
        import tomli as tomllib
    with open(os.path.join(ROOT, "pyproject.toml"), "rb") as fh:
        cfg = tomllib.load(fh)
    target = cfg["project"]["scripts"]["adapted"]
    mod, fn = target.split(":")
    import importlib

    main = getattr(importlib.import_module(mod), fn)
    p = __import__("adapted_amd.main", fromlist=["build_parser"]).build_parser()
    assert callable(main) and {"detect", "continue"} <= set(p._subparsers._group_actions[0].choices)


def test_one_hip_runtime_whatever_the_import_order():
    """adapted_amd.lib.load() before OR after `import torch` leaves exactly one libamdhip64 mapped (the torch wheel bundles a
    runtime with the SONAME this library links: INTEGRATION.md section 3); ADAPTED_HIP_RUNTIME=system keeps /opt/rocm's."""
    import subprocess
    import sys

    def run(code, **env):
        e = dict(os.environ, PYTHONPATH=ROOT, **env)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=e, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        return r.stdout.strip().splitlines()[-1]

    first = run("from adapted_amd import lib; lib.load(); import torch; print(len(lib.hip_runtimes()), lib.hip_runtimes())")
    after = run("import torch; from adapted_amd import lib; lib.load(); print(len(lib.hip_runtimes()), lib.hip_runtimes())")
    assert first.startswith("1 ") and after.startswith("1 "), (first, after)
    if "torch" in after:  # a wheel with a bundled runtime: both orders end on it
        assert "torch" in first
    system = run("from adapted_amd import lib; lib.load(); print(len(lib.hip_runtimes()), lib.hip_runtimes())", ADAPTED_HIP_RUNTIME="system")
    assert system.startswith("1 ") and "torch" not in system, system
    # a wheel whose bundled runtime carries ANOTHER SONAME than the one this library links (another ROCm major) is left alone:
    # the library runs on the system runtime it was built against, and nothing is raised while torch stays unimported
    other = run("from adapted_amd import lib\n"
                "real = lib._elf_dynamic\n"
                "lib._elf_dynamic = lambda p: (('libamdhip64.so.6', []) if '/torch/' in p else real(p))\n"
                "lib.load(); print(len(lib.hip_runtimes()), lib.hip_runtimes())")
    assert other.startswith("1 ") and "torch" not in other, other


def test_elf_dynamic_reads_soname_and_needed():
    from adapted_amd import build, lib

    soname, needed = lib._elf_dynamic(build.build())
    assert any(n.startswith("libamdhip64.so") for n in needed) and "libc.so.6" in needed
    import glob

    libc = [p for p in glob.glob("/lib/x86_64-linux-gnu/libm.so.6") + glob.glob("/usr/lib/x86_64-linux-gnu/libm.so.6")]
    if libc:
        assert lib._elf_dynamic(libc[0])[0] == "libm.so.6"


def test_cpu_baseline_worker_count_follows_mask_and_quota(monkeypatch):
    """bench.py's all-cores CPU baseline: every CPU the affinity mask and the cgroup quota grant, the documented 16-CPU share of a
    one-GPU lease only where neither restricts a big host, an explicit --cpu-procs above all"""
    import bench

    def grant(aff, quota):
        monkeypatch.setattr(bench, "cpu_grant", lambda: {"affinity_cpus": aff, "cgroup_cpu_max": "x", "cgroup_cpus": quota})

    grant(256, 16.0)
    assert bench.cpu_worker_count(None) == (16, "cgroup cpu.max")
    grant(8, None)
    assert bench.cpu_worker_count(None)[0] == 8
    grant(256, None)
    n, why = bench.cpu_worker_count(None)
    assert n == 16 and "unrestricted" in why
    grant(32, 48.0)
    assert bench.cpu_worker_count(None) == (32, "affinity mask")
    assert bench.cpu_worker_count(5) == (5, "--cpu-procs")
    g = bench.cpu_grant.__wrapped__() if hasattr(bench.cpu_grant, "__wrapped__") else None
    assert g is None or g["affinity_cpus"] >= 1


# ---------------------------------------------------------------------------------------------------------------------------
# Multi-GPU preflight without hardware (round 4): the pieces of the N > 1 path that do not need a GPU, at the rank counts the
# driver will launch (reference: the host pool of adapted/file_proc.py:738-784 -- here one process per GPU, whole groups of
# minibatches per rank, one row gather).
@pytest.mark.parametrize("ws", [2, 4, 8])
def test_group_sharder_on_heavy_tailed_lengths(ws):
    """io_utils.GroupSharder over 8000 Pareto lengths (BASELINE configs[4]): every rank computes the same assignment from the
    metadata alone, every group has exactly one owner, the stream order comes back from the groups' ordinals, and the preloaded
    samples per rank stay within 5 % of their mean (groups of 50 reads: a group is 1 / 20 of a rank's share at 8 ranks)."""
    from adapted_amd import synth
    from adapted_amd.io_utils import GroupSharder

    m, group = 201500, 50
    lens = [synth.pareto_length(2024, i) for i in range(8000)]
    owners = []
    for rank in range(ws):
        sh = GroupSharder(ws, rank, m)
        mine = []
        for g0 in range(0, len(lens), group):
            own = sh.start_group()
            for n in lens[g0:g0 + group]:
                sh.add(n)
            if own:
                mine.append(g0)
        owners.append((mine, list(sh.load)))
    loads = owners[0][1]
    assert all(o[1] == loads for o in owners)                      # the same bookkeeping on every rank
    allg = sorted(g for o in owners for g in o[0])
    assert allg == list(range(0, len(lens), group))                # every group owned exactly once
    order = sorted((g0, r) for r, o in enumerate(owners) for g0 in o[0])
    assert [g for g, _ in order] == allg                            # ordinals rebuild the stream order
    assert sum(loads) == sum(min(n, m) for n in lens)
    assert max(loads) / (sum(loads) / ws) <= 1.05, loads
    assert min(len(o[0]) for o in owners) >= 1


_GLOO_WORKER_8 = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from adapted_amd import parallel, lib
from adapted_amd.lib import ROW_DTYPE
dist.init_process_group("gloo")
rank, ws = dist.get_rank(), dist.get_world_size()
assert ws == 8
n_reads, mb = 5300, 1000                       # 6 minibatches over 8 ranks: ranks 6 and 7 own nothing
a, b = parallel.shard_reads(n_reads, mb, ws, rank)
rows = np.zeros(b - a, dtype=ROW_DTYPE)
rows["col"][:, 0] = np.arange(a, b)
rows["success"] = 1
rows["n_open_pores"] = 0
long_lists = {3: (7, 23), 5: (299, 39)}        # rank -> (local row, entries): lists beyond a row's 16 slots, on non-zero ranks
if rank in long_lists:
    i, k = long_lists[rank]
    rows[i]["n_open_pores"] = k
    rows[i]["open_pores_more"] = lib.register_open_pores(np.arange(k, dtype=np.int32) * 11 + rank)
out = parallel.gather_rows(rows, dst=0)
if rank == 0:
    assert out.shape[0] == n_reads and np.array_equal(out["col"][:, 0], np.arange(n_reads))
    for rk, (i, k) in long_lists.items():
        g = parallel.shard_reads(n_reads, mb, ws, rk)[0] + i
        assert out[g]["n_open_pores"] == k
        got = lib._OPEN_PORES_MORE[int(out[g]["open_pores_more"])]
        assert np.array_equal(got, np.arange(k, dtype=np.int32) * 11 + rk), (rk, got)
    assert int((out["n_open_pores"] > lib.MAX_OPEN_PORES).sum()) == 2
    print("GATHER8_OK", out.shape[0], [parallel.shard_reads(n_reads, mb, ws, r) for r in (5, 6, 7)])
else:
    assert out is None
    assert not lib._OPEN_PORES_MORE       # the senders' lists left their registries
dist.destroy_process_group()
"""


def test_row_gather_eight_ranks_gloo(tmp_path):
    script = tmp_path / "w8.py"
    script.write_text(_GLOO_WORKER_8 % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8",
                        "--master-addr", "127.0.0.1", "--master-port", _free_port(), str(script)],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "GATHER8_OK 5300 [(5000, 5300), (5300, 5300), (5300, 5300)]" in r.stdout


def test_bench_refuses_more_ranks_than_devices_before_touching_a_gpu():
    """`python bench.py --gpus 8` with fewer visible devices: exit code 2 from the launcher itself -- no rank is started, nothing
    that initialises the GPU is imported (counting devices does not) -- and the command it would start is torchrun's module form
    on 127.0.0.1 (the driver's own command line)."""
    import importlib.util

    import torch

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cmd = bench.launch_command(8, 29555, ["--gpus", "8", "--steps", "3"])
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29555"
    assert cmd[-5] == os.path.join(ROOT, "bench.py") and cmd[-4:] == ["--gpus", "8", "--steps", "3"]
    if torch.cuda.device_count() >= 8:
        pytest.skip("eight devices visible")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "ADP_BENCH_BACKEND")}
    probe = ("import sys, runpy; sys.argv = ['bench.py', '--gpus', '8', '--steps', '1', '--warmup', '0']\n"
             "try:\n    runpy.run_path(%r, run_name='__main__')\nexcept SystemExit as e:\n"
             "    import torch\n    print('EXIT', e.code, 'GPU_INITIALISED', torch.cuda.is_initialized(), 'LIB', 'adapted_amd.lib' in sys.modules)\n    raise\n"
             % os.path.join(ROOT, "bench.py"))
    r = subprocess.run([sys.executable, "-c", probe], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 2, (r.returncode, r.stdout[-500:], r.stderr[-500:])
    assert "EXIT 2 GPU_INITIALISED False LIB False" in r.stdout
    assert "--gpus 8 but only" in r.stderr


def test_assert_one_runtime_catches_torch_imported_after_a_system_runtime_load():
    """(advisor, round 4) the two-runtimes guard used to look only when the library was loaded: ADAPTED_HIP_RUNTIME=system (or a torch
    wheel with another SONAME), the library first, torch later -- two runtimes, silently.  assert_one_runtime() is called again at the
    entry points that hand torch's memory to the library (Engine calls with device pointers), parallel.gather_rows and bench.py."""
    code = ("from adapted_amd import lib\n"
            "lib.load()\n"                      # /opt/rocm's runtime, no torch yet: fine
            "lib.assert_one_runtime()\n"
            "import torch\n"                    # the wheel's bundled runtime joins it
            "n = len(lib.hip_runtimes())\n"
            "try:\n"
            "    lib.assert_one_runtime()\n"
            "    print('NO_ERROR', n)\n"
            "except lib.HipLibraryError as e:\n"
            "    print('RAISED', n)\n"
            "try:\n"
            "    lib._check_runtime_once_torch_is_here()\n"
            "    print('ENTRY_OK')\n"
            "except lib.HipLibraryError:\n"
            "    print('ENTRY_RAISED')\n")
    env = dict(os.environ, ADAPTED_HIP_RUNTIME="system")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.split()
    n = int(lines[1])
    if n > 1:  # (this image: the wheel bundles its own libamdhip64) the later import is caught, at the explicit check and at an entry point
        assert lines[0] == "RAISED" and "ENTRY_RAISED" in lines, out.stdout
    else:      # (a torch that links the system runtime: one runtime, nothing to catch)
        assert lines[0] == "NO_ERROR" and "ENTRY_OK" in lines, out.stdout


def test_bench_finds_a_kernels_traffic_whatever_its_template_arguments():
    """profiles/rNN_traffic*.json name a kernel with its template arguments (k_partition_stats<256, 5>), bench.py by its launch scope"""
    sys.path.insert(0, ROOT)
    import bench

    t = {"_reads_per_launch": 96000, "_note": "x", "k_partition_stats<256, 5>": {"hbm_bytes": 96.0e9}, "k_gains<1>": {"hbm_bytes": 1.0}, "k_gains<2>": {"hbm_bytes": 2.0},
         "k_validate": {"hbm_bytes": 30.0e9}}
    assert bench._traffic_of(t, "k_partition_stats", 48000) == 48.0e9
    assert bench._traffic_of(t, "k_validate", 96000) == 30.0e9
    assert bench._traffic_of(t, "k_gains<2>", 96000) == 2.0
    assert bench._traffic_of(t, "k_gains", 96000) is None          # ambiguous: two instantiations
    assert bench._traffic_of(t, "k_norm_pool", 96000) is None and bench._traffic_of(None, "k_validate", 1) is None


def test_every_lds_array_of_the_library_is_declared_16_byte_aligned():
    """Dynamic LDS starts behind a kernel's static words: a 4-byte-aligned array there leaves every 16-byte access of it misaligned --
    correct, and ~25 times slower (round 5: k_mvs_series_pipe's rings at byte 580, 11.9 instead of 6.2 ms).
    Every `__shared__` ARRAY (and the two scratch structs) in adapted_amd/csrc carries `__attribute__((aligned(16)))`."""
    import glob
    import re

    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "adapted_amd", "csrc")
    bad = []
    for path in sorted(glob.glob(os.path.join(root, "*.h")) + glob.glob(os.path.join(root, "*.hip"))):
        with open(path) as fh:
            for no, line in enumerate(fh, 1):
                code = line.split("//")[0]
                if "__shared__" not in code or "aligned(16)" in code:
                    continue
                if "[" in code or re.search(r"__shared__\s+(WaveScratch|N1Sel)\b", code):
                    bad.append("%s:%d: %s" % (os.path.basename(path), no, code.strip()))
    assert not bad, "\n".join(bad)
