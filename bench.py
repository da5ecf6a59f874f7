#!/usr/bin/env python3
"""bench.py -- reads/sec through adapter + poly(A) detect on MI355X.

Workload (BASELINE.json configs[1]): synthetic RNA004 reads, 200 000-sample trace window
(--max_obs_trace 200000 -> preload m = 201 500 float32 samples = 806 000 B per read), LLR
primary detector + start-peak scan + boundary validation, minibatches of 1000 reads
(normalisation is per minibatch, as in the reference).  A "step" is one pass of the hot path
over one HBM-resident batch of --reads reads per GPU (default 96 000: sized for the 288 GB HBM); inputs are generated on the device
before the timed region (bit-identical host twin: adapted_amd/synth.py).

    python bench.py --gpus 1 --steps 8 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU, each owns a contiguous block of whole minibatches (weak scaling,
no data-path collective); the fixed-width result rows are gathered to rank 0 over RCCL at the
end of every step (inside the timed region).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def make_spc(max_obs_trace: int, primary: str = "llr"):
    from adapted_amd.config import get_chemistry_specific_config

    spc = get_chemistry_specific_config("RNA004")
    spc.llr_boundaries.llr_detect = primary == "llr"
    spc.cnn_boundaries.cnn_detect = primary == "cnn"
    spc.core.max_obs_trace = max_obs_trace
    spc.update_primary_method()
    spc.update_sig_preload_size()
    return spc


# ---------------------------------------------------------------------------------------------- CPU baseline (the checker, timed)
def _cpu_worker(task):
    """one minibatch through the CPU oracle in a worker PROCESS (the reference's layout: adapted/file_proc.py:738-784)"""
    shm_name, off, n, m, max_obs_trace = task
    from multiprocessing import shared_memory

    from oracle import oracle

    oracle.lib()
    shm = shared_memory.SharedMemory(name=shm_name)
    try:
        sig = np.ndarray((n, m), dtype=np.float32, buffer=shm.buf, offset=off)
        lens = np.full(n, m, dtype=np.int32)
        spc = make_spc(max_obs_trace)
        t0 = time.perf_counter()
        res = oracle.detect_llr(sig, lens, spc, with_start_peak=True)
        dt = time.perf_counter() - t0
        return dt, sum(bool(r["success"]) for r in res)
    finally:
        shm.close()


def _cpu_pool_start(procs: int):
    """worker processes for the all-cores baseline, started BEFORE this process touches the GPU (spawned, not forked)"""
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    pool = ctx.Pool(procs)
    pool.map(_cpu_noop, range(procs))  # (workers up, modules imported)
    return pool


def _cpu_noop(i):
    from oracle import oracle

    oracle.lib()
    return i


def cpu_baseline(eng, spc, dsig, n_sample: int, m: int, gpu_rows, lens, pool, procs: int, n_all: int, max_obs_trace: int):
    """Time the CPU oracle (a C port of the reference path, oracle/adapted_oracle.c) on the first minibatches of the resident
    batch: ONE thread on n_sample reads (checked against the GPU rows of an identically composed minibatch), then `procs`
    worker processes with one minibatch of n_all reads each, the reference's own process layout.  The port / reference
    ratio measured in the build container (tools/time_reference.py -> profiles/r02_reference_timing.json) turns both into
    reference-equivalent figures (the reference cannot travel to the GPU box)."""
    from adapted_amd import lib
    from oracle import oracle

    oracle.lib()
    sig = np.zeros((n_sample, m), dtype=np.float32)
    eng.d2h(sig, dsig)
    lens1 = np.ascontiguousarray(lens[:n_sample], dtype=np.int32)
    t0 = time.perf_counter()
    want = oracle.detect_llr(sig, lens1, spc, with_start_peak=True)
    dt = time.perf_counter() - t0
    rows, _ = eng.detect_llr_rows(sig, lens1, n_sample, n_sample, with_start_peak=True)
    got = lib.rows_to_results(rows, "llr")
    mism = 0
    for g, w in zip(got, want):
        for k, v in w.items():
            if k.startswith("_"):
                continue
            a = getattr(g, k, None)
            if hasattr(a, "tolist"):
                a = a.tolist()
            if isinstance(v, float) and isinstance(a, (float, np.floating)):
                if not (a == v or (np.isnan(a) and np.isnan(v))):
                    mism += 1
            elif a != v:
                mism += 1
    out = {"value": n_sample / dt, "unit": "reads/s", "cores": 1, "kind": "port",
           "sample": "%d reads (1 minibatch) of the workload, C port (oracle/), 1 thread, %.1f s" % (n_sample, dt),
           "mismatched_fields": mism,  # GPU rows of the same minibatch against the port's, every field
           "host_cpus": os.cpu_count()}
    out.update(cpu_grant())
    del sig
    if pool is not None and procs > 1:
        from multiprocessing import shared_memory

        shm = shared_memory.SharedMemory(create=True, size=procs * n_all * m * 4)
        try:
            for k in range(procs):  # minibatch k of the resident batch -> worker k
                blk = np.ndarray((n_all, m), dtype=np.float32, buffer=shm.buf, offset=k * n_all * m * 4)
                eng.d2h(blk, dsig + k * n_all * m * 4)
            t0 = time.perf_counter()
            res = pool.map(_cpu_worker, [(shm.name, k * n_all * m * 4, n_all, m, max_obs_trace) for k in range(procs)])
            wall = time.perf_counter() - t0
        finally:
            shm.close()
            shm.unlink()
        out["cores_all"] = procs
        out["cores_all_from"] = _CPU_WHY
        out["value_all_cores"] = procs * n_all / wall
        out["sample_all_cores"] = "%d procs x %d reads (reference's pool layout), %.1f s wall" % (procs, n_all, wall)
    for tname in ("r04_reference_timing.json", "r03_reference_timing.json", "r02_reference_timing.json"):
        tfile = os.path.join(ROOT, "profiles", tname)
        if not os.path.exists(tfile):
            continue
        with open(tfile) as fh:
            t = json.load(fh)
        # the real reference beside the port, same reads, build container (tools/time_reference.py); numbers only, so
        # that the driver's record alone carries the GPU / reference ratio
        rp, ra = t["port_over_reference_per_proc"], t["port_over_reference_all_procs"]
        out["port_over_reference"] = rp
        out["port_over_reference_all_procs"] = ra
        out["port_over_reference_source"] = "profiles/" + tname
        out["reference_equivalent"] = out["value"] / rp  # reads/s of the reference on ONE core of this host
        if "value_all_cores" in out:
            out["reference_equivalent_all_cores"] = out["value_all_cores"] / ra  # ... on `cores_all` cores
        break
    return out


_CPU_WHY = None


def cpu_grant():
    """what this process may use of the host: the affinity mask, the cgroup CPU quota, and the worker count derived from
    them (a 1-GPU lease of the pool is a 16-CPU share of a 256-CPU host; nothing in the mask or the quota says so)"""
    aff = len(os.sched_getaffinity(0))
    quota = None
    raw = None
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(f) as fh:
                raw = fh.read().strip()
            break
        except OSError:
            continue
    if raw:
        parts = raw.split()
        try:
            if len(parts) == 2 and parts[0] != "max":
                quota = float(parts[0]) / float(parts[1])
            elif len(parts) == 1 and int(parts[0]) > 0:
                quota = int(parts[0]) / 100000.0
        except ValueError:
            pass
    return {"affinity_cpus": aff, "cgroup_cpu_max": raw, "cgroup_cpus": quota}


def cpu_worker_count(requested):
    """worker processes of the all-cores baseline: every CPU the mask and the quota grant; where neither limits anything
    (mask = the whole host, quota "max") the pool's documented share of a one-GPU box, 16"""
    if requested:
        return int(requested), "--cpu-procs"
    g = cpu_grant()
    n = g["affinity_cpus"]
    why = "affinity mask"
    if g["cgroup_cpus"]:
        q = max(1, int(g["cgroup_cpus"] + 0.5))
        if q < n:
            n, why = q, "cgroup cpu.max"
    if n > 64 and not g["cgroup_cpus"]:
        n, why = 16, "unrestricted mask (%d CPUs): the 1-GPU lease's 16-CPU share" % g["affinity_cpus"]
    return n, why


def host_pipeline_bench(args, spc, device):
    """PCIe-inclusive rate: minibatches assembled in host memory (a memcpy per minibatch out of a small pool stands in for
    the reader), pinned staging, H2D, detect, rows back.  Returns the JSON object (--host-pipeline prints it as a line of its
    own; the default run attaches three of them as secondary.host_pipeline)."""
    import threading

    import torch
    from adapted_amd import lib
    from adapted_amd.pipeline import HostPipeline

    m, mb = spc.sig_preload_size, args.minibatch
    i16 = bool(args.int16)
    G = max(1, args.group)
    ragged = bool(args.ragged)
    pipe = HostPipeline(spc, mb, m, device=device, primary="llr", with_start_peak=not args.no_start_peak, int16_input=i16, group=G,
                        ragged=ragged)
    mb = mb * G  # reads per slot from here on
    pool = []
    dev = torch.device("cuda", device)
    d = torch.empty((mb, m), dtype=torch.float32, device=dev)
    lens = np.full(mb, m, dtype=np.int32)
    if args.lens == "pareto":
        from adapted_amd import synth as _synth

        lens = np.array([_synth.pareto_length(args.seed, i) for i in range(mb)], dtype=np.int32)
    dl = torch.from_numpy(lens).to(dev)
    take = np.minimum(lens, m).astype(np.int64)
    offs = np.zeros(mb + 1, dtype=np.int64)
    np.cumsum(take, out=offs[1:])
    sc, of = np.float32(0.17), np.float32(-12.0)  # a typical pod5 calibration
    for k in range(3):
        pipe.eng.synth_fill(d.data_ptr(), dl.data_ptr(), mb, seed=args.seed, first_read=k * mb, decorate=True)
        torch.cuda.synchronize()
        if i16:  # the ADC codes whose calibrated values are (to the code's resolution) the synthetic pA samples
            pool.append(torch.clamp(torch.round(d / float(sc) - float(of)), -32768, 32767).to(torch.int16).cpu().numpy().copy())
        else:
            pool.append(d.cpu().numpy().copy())
        if ragged:  # the same reads packed back to back
            dense = pool[-1]
            flat = np.empty(int(offs[-1]), dtype=dense.dtype)
            for r in range(mb):
                flat[offs[r]:offs[r + 1]] = dense[r, :take[r]]
            pool[-1] = flat
    ids = np.arange(mb).astype(object)
    n_ok = [0]
    lock = threading.Lock()

    from concurrent.futures import ThreadPoolExecutor

    K = max(1, args.fill_threads)
    ex = ThreadPoolExecutor(K) if K > 1 else None

    def fill(get_buffers, count, assemble=True):
        for i in range(count):
            bufs = get_buffers()
            sig, ln = bufs[0], bufs[1]
            if i16:
                bufs[-2][:] = sc
                bufs[-1][:] = of
            if ragged:
                bufs[2][:] = offs
                if assemble:
                    src = pool[i % len(pool)]
                    if ex is None:
                        np.copyto(sig[:src.size], src)
                    else:
                        step = (src.size + K - 1) // K
                        list(ex.map(lambda a: np.copyto(sig[a:a + step], src[a:a + step]), range(0, src.size, step)))
            elif assemble:  # (host memcpy: the stand-in for a reader writing the minibatch)
                src = pool[i % len(pool)]
                if ex is None:
                    np.copyto(sig, src)
                else:
                    step = (mb + K - 1) // K
                    list(ex.map(lambda a: np.copyto(sig[a:a + step], src[a:a + step]), range(0, mb, step)))
            ln[:] = lens
            yield mb, ids

    def on_rows(_ids, rows):
        with lock:
            n_ok[0] += int(rows["success"].sum())

    pipe.run(lambda gb: fill(gb, 2), on_rows)  # warm-up
    n_ok[0] = 0
    t0 = time.perf_counter()
    total = pipe.run(lambda gb: fill(gb, args.host_pipeline), on_rows)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    total2 = pipe.run(lambda gb: fill(gb, args.host_pipeline, assemble=False), on_rows)  # staging slots already filled
    dt2 = time.perf_counter() - t1
    pipe.close()
    bps = 2 if i16 else 4
    per_read = (float(offs[-1]) / mb) if ragged else float(m)  # samples that cross PCIe per read
    gb = total * per_read * bps / 1e9
    del pool, d
    torch.cuda.empty_cache()
    return ({"metric": "reads/sec (adapter+polyA detect), RNA004 200k-sample reads, HOST buffers (PCIe-inclusive; not the headline)",
                      "value": total / dt, "unit": "reads/s", "n_gpus": 1, "minibatches": args.host_pipeline,
                      "h2d_GB_per_s": gb / dt, "pass_rate": n_ok[0] / max(total + total2, 1),
                      "without_host_assembly": {"value": total2 / dt2, "h2d_GB_per_s": total2 * per_read * bps / 1e9 / dt2},
                      "input": ("int16 ADC + device calibration" if i16 else "float32 pA") + (", reads packed back to back (ragged)" if ragged else ", padded [N, m] matrix"),
                      "lens": "%s (mean %.0f of m = %d samples preloaded)" % (args.lens, float(take.mean()), m), "fill_threads": K,
                      "config": {"workload": "LLR + start_peak + validate from pinned host staging, m=%d, minibatch=%d, %d minibatches per call" % (m, args.minibatch, G)}})


def _traffic_of(traffic, kernel, reads):
    """HBM bytes per launch of `kernel` from a rocprofv3 --pmc summary (profiles/rNN_traffic*.json), scaled to `reads` reads; the profile
    names a kernel with its template arguments (k_partition_stats<256, 5>), the bench by the name of its launch scope"""
    if not traffic:
        return None
    hit = [k for k in traffic if not k.startswith("_") and (k == kernel or k.split("<")[0] == kernel)]
    if len(hit) != 1:
        return None
    return traffic[hit[0]]["hbm_bytes"] * (reads / traffic["_reads_per_launch"])


CNN_FLOP_PER_POS = 2.0 * (64 * 7 + 2 * 64 * 64 * 7 + 64 * 2 * 7)  # F_alg = this x L1 per read
F32_MFMA_PEAK_TF = 157.3  # MI355X float32 matrix peak (/opt/skills/guides/MI355X_MICROARCH.md)
F16_MFMA_PEAK_TF = 2500.0  # dense float16 / bfloat16 matrix peak (same guide: ~2.5 PF, sparsity not counted)


def run_workload(w, rank, world, local, dist, backend, cpu_pool=None, cpu_procs=0):
    """One workload (w: argparse-like namespace) on this rank's GPU; rank 0 returns the JSON object, the others None.
    dist is None on the plain one-GPU path; with a process group (N > 1, or ADP_BENCH_FORCE_DIST=1 at N = 1: the same code
    on a group of one) every step ends with the row gather and the clock is the max over ranks."""
    import threading

    import torch
    from adapted_amd import lib

    dev = torch.device("cuda", local)
    multi = dist is not None
    comm_dev = dev if backend == "nccl" else torch.device("cpu")
    spc = make_spc(w.max_obs_trace, w.primary)
    cnn_mod = None
    if w.primary == "cnn":
        from adapted_amd.detect import cnn as cnn_mod
    m = spc.sig_preload_size
    R, mb = w.reads, w.minibatch
    assert R % mb == 0, "--reads must be a whole number of minibatches"
    NS = max(1, w.streams)
    assert (R // mb) % NS == 0, "--reads must split into whole minibatches per stream"
    Rs = R // NS
    engines = [lib.Engine(spc, Rs, m, device=local) for _ in range(NS)]
    eng = engines[0]
    sig_t = torch.empty((R, m), dtype=torch.float32, device=dev)
    lens_host = np.full(R, m, dtype=np.int32)
    if w.lens == "pareto":
        from adapted_amd import synth as _synth

        lens_host = np.array([_synth.pareto_length(w.seed, rank * R + i) for i in range(R)], dtype=np.int32)
    len_t = torch.from_numpy(lens_host).to(dev)
    # two row buffers: with several ranks the gather of step k (RCCL, its own stream) runs beside the kernels of step k + 1
    rows_bufs = [torch.empty((R, lib.ROW_DTYPE.itemsize), dtype=torch.uint8, device=dev) for _ in range(2 if multi else 1)]
    rows_t = rows_bufs[0]
    torch.cuda.synchronize()
    # rank r owns reads [r*R, (r+1)*R) of the global stream: contiguous whole minibatches
    for k, e in enumerate(engines):
        e.synth_fill(sig_t.data_ptr() + k * Rs * m * 4, len_t.data_ptr() + k * Rs * 4, Rs, seed=w.seed,
                     first_read=rank * R + k * Rs, decorate=True)
    if w.adc_step > 0:
        for s0 in range(0, R, 1000):
            sig_t[s0:s0 + 1000].div_(w.adc_step).round_().mul_(w.adc_step)
        torch.cuda.synchronize()
    raw_t = cal_t = None
    i16_check = None
    if getattr(w, "int16", False):
        # the same reads as the sequencer stores them: int16 ADC codes + a per-read calibration (pA = scale * (adc + offset));
        # the float32 matrix is dropped, every kernel reads the raw samples (adp_detect_llr_i16)
        sc, of = 0.17, -12.0
        raw_t = torch.empty((R, m), dtype=torch.int16, device=dev)
        for s0 in range(0, R, 500):
            raw_t[s0:s0 + 500] = torch.clamp(torch.round(torch.nan_to_num(sig_t[s0:s0 + 500]) / sc - of), -32768, 32767).to(torch.int16)
        cal_t = torch.cat([torch.full((R,), sc, dtype=torch.float32, device=dev), torch.full((R,), of, dtype=torch.float32, device=dev)])
        torch.cuda.synchronize()
        # parity of the first minibatch: raw path against adp_calibrate_i16 + the float32 path
        eng.calibrate_i16(raw_t.data_ptr(), len_t.data_ptr(), cal_t.data_ptr(), cal_t.data_ptr() + R * 4, mb, sig_t.data_ptr())
        a, _ = eng.detect_llr_rows(sig_t.data_ptr(), len_t.data_ptr(), mb, mb, with_start_peak=not w.no_start_peak, device_ptrs=True, tails_nan=True)
        b, _ = eng.detect_llr_rows_i16(raw_t.data_ptr(), len_t.data_ptr(), cal_t.data_ptr(), cal_t.data_ptr() + R * 4, mb, mb,
                                       with_start_peak=not w.no_start_peak)
        a["open_pores_more"] = 0; b["open_pores_more"] = 0
        i16_check = bool(a.tobytes() == b.tobytes())
        sig_t = None
        torch.cuda.empty_cache()
    gathered = None
    if multi and rank == 0:
        gathered = [torch.empty_like(rows_t, device=comm_dev) for _ in range(world)]

    state = {"rows": rows_t, "pending": None, "i": 0}

    def run_part(k):
        s0 = k * Rs
        rows_t = state["rows"]
        if w.primary == "cnn":  # one library call (adp_detect_cnn) per engine; find_peaks / row compaction per minibatch
            rows = cnn_mod.detect_rows_device(engines[k], sig_t.data_ptr() + s0 * m * 4, len_t.data_ptr() + s0 * 4, Rs,
                                              lens_host[s0:s0 + Rs], None, spc, minibatch=mb)
            engines[k].h2d(rows_t.data_ptr() + s0 * lib.ROW_DTYPE.itemsize, rows)
            return
        if raw_t is not None:
            engines[k].detect_llr_rows_i16(raw_t.data_ptr() + s0 * m * 2, len_t.data_ptr() + s0 * 4, cal_t.data_ptr() + s0 * 4,
                                           cal_t.data_ptr() + (R + s0) * 4, Rs, mb, with_start_peak=not w.no_start_peak,
                                           rows_dev=rows_t.data_ptr() + s0 * lib.ROW_DTYPE.itemsize)
            return
        engines[k].detect_llr_rows(sig_t.data_ptr() + s0 * m * 4, len_t.data_ptr() + s0 * 4, Rs, mb,
                                   with_start_peak=not w.no_start_peak, device_ptrs=True,
                                   rows_dev=rows_t.data_ptr() + s0 * lib.ROW_DTYPE.itemsize,
                                   tails_nan=True)  # (the generator pads with NaN; no padding at all in the headline workload)

    def step():
        if NS == 1:
            run_part(0)
        else:  # the C ABI call blocks (and releases the GIL): one host thread per engine/stream
            ths = [threading.Thread(target=run_part, args=(k,)) for k in range(NS)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
        if multi:
            # every detect call ends with its stream drained, so the rows are complete here; the gather is left running
            # and waited for before the next one starts (and before the clock stops)
            if state["pending"] is not None:
                state["pending"].wait()
                if backend == "nccl":  # (an RCCL wait orders torch's stream, not the host: the buffer that gather read is
                    torch.cuda.current_stream().synchronize()  # written again by the library's own stream one step from now)
            cur = state["rows"]
            state["pending"] = dist.gather(cur if backend == "nccl" else cur.cpu(), gathered, dst=0, async_op=True)
            state["i"] += 1
            state["rows"] = rows_bufs[state["i"] % 2]

    def sync():
        if state["pending"] is not None:
            state["pending"].wait()
            state["pending"] = None
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(w.warmup):
        step()
    for e in engines:
        e.set_profiling(True)
    ktimes = {}
    sync()
    t0 = time.perf_counter()
    for _ in range(w.steps):
        step()
        for e in engines:
            per = {}
            for name, ms in e.kernel_times():  # HIP events recorded on each engine's stream; one entry per launch
                per[name] = per.get(name, 0.0) + ms
            for name, ms in per.items():
                ktimes.setdefault(name, []).append(ms)
    sync()
    dt = time.perf_counter() - t0
    for e in engines:
        e.set_profiling(False)
    lib.assert_one_runtime()  # (torch's device memory went through the library all along: one HIP runtime, or fail loudly)
    if multi:
        t = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    total_reads = R * world * w.steps
    value = total_reads / dt
    out = None
    if rank == 0:
        # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of THIS workload shape, where one was taken (else null)
        traffic = None
        shape = None
        if w.primary == "llr" and w.adc_step == 0 and NS == 1 and os.environ.get("ADP_GROUPS", "1") == "1" and not w.no_start_peak:
            if w.lens == "full" and w.max_obs_trace == 200000:
                shape = "int16" if getattr(w, "int16", False) else "f32"
            elif w.lens == "pareto" and w.max_obs_trace == 200000 and not getattr(w, "int16", False):
                shape = "pareto"
            elif w.lens == "full" and w.max_obs_trace == 16000 and not getattr(w, "int16", False):
                shape = "default_window"
        tnames = {"f32": ("r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json"),
                  "int16": ("r05_traffic_int16.json", "r04_traffic_int16.json", "r03_traffic_int16.json", "r02_traffic_int16.json"),
                  "pareto": ("r05_traffic_pareto.json",), "default_window": ("r05_traffic_default_window.json",)}.get(shape, ())
        for tname in tnames:
            tfile = os.path.join(ROOT, "profiles", tname)
            if os.path.exists(tfile):
                with open(tfile) as fh:
                    traffic = json.load(fh)
                traffic["_file"] = "profiles/" + tname
                break
        rows = np.zeros(R, dtype=lib.ROW_DTYPE)
        eng.d2h(rows, rows_bufs[(state["i"] - 1) % 2 if multi else 0].data_ptr())  # (the last step's rows)
        n_ok = int(rows["success"].sum())
        import hashlib

        gather_ok = None
        if multi:  # what the collective delivered for this rank's block against the block itself
            gather_ok = bool(np.array_equal(gathered[0].cpu().numpy().reshape(-1), rows.view(np.uint8).reshape(-1)))
        hrows = rows.copy()
        hrows["open_pores_more"] = 0  # (arena offsets: handed out by an atomic counter, any order)
        rows_digest = hashlib.sha256(hrows.tobytes()).hexdigest()  # (last step, this rank: equal between a plain and a grouped run)
        del hrows
        kavg = {k: float(np.mean(v)) for k, v in ktimes.items()}
        dom = max(kavg, key=kavg.get)
        # 4 bytes per PRELOADED sample, each read once; a launch covers the Rs reads of one engine
        mean_samples = float(np.minimum(lens_host, m).mean())
        bps = 2.0 if raw_t is not None else 4.0  # bytes per preloaded sample as it lies in HBM
        b_alg = bps * mean_samples * Rs
        achieved = b_alg / (kavg[dom] * 1e-3) / 1e9
        step_s = dt / w.steps
        out = {
            "metric": "reads/sec (adapter+polyA detect), RNA004 200k-sample reads",
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": w.steps, "warmup": w.warmup,
            "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 samples/statistics, f64 cumulative sums + LLR trace" if w.primary == "llr" else "f32 (conv net on the float32 matrix cores, statistics)",
            "data": "synthetic (device-generated, adapted_amd/synth.py twin)",
            "config": {"workload": ("BASELINE configs[1]: RNA004 LLR + start_peak + validate" if w.primary == "llr" else
                                    "BASELINE configs[2]: RNA004 CNN head (hand-written float32 MFMA conv stack) + predict + validate") +
                                   ", max_obs_trace=%d (m=%d), minibatch=%d, %d reads/step/GPU resident in HBM"
                                   % (w.max_obs_trace, m, mb, R),
                       "reads_per_step_per_gpu": R, "minibatch": mb, "m": m, "pass_rate": n_ok / R, "streams_per_gpu": NS,
                       "grouping": {k: os.environ.get(k) for k in ("ADP_GROUPS", "ADP_LANES", "ADP_STAGGER") if os.environ.get(k)} or "one group (phases in turn on one stream)"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": _traffic_of(traffic, dom, Rs),
                         "traffic_source": (traffic["_file"] + " (rocprofv3 --pmc, FETCH_SIZE doubled per MI355X_MICROARCH.md)") if traffic else None,
                         "kernel_ms": kavg[dom], "algorithmic_bytes_per_launch": b_alg,
                         "whole_path_frac": (bps * mean_samples * R / step_s) / 1e9 / HBM_PEAK_GBS},
            "kernel_ms": {k: round(v, 4) for k, v in sorted(kavg.items(), key=lambda kv: -kv[1])},
            "kernel_ms_sum": sum(kavg.values()),
            "rows_sha256": rows_digest,
        }
        if multi:
            out["gathered_block_equals_local_rows"] = gather_ok
        if w.primary == "cnn":
            # configs[2] is a dense contraction: the float32 matrix peak bounds the conv stack
            Lc = (m - spc.core.min_obs_adapter + spc.core.downscale_factor - 1) // spc.core.downscale_factor
            L1 = (Lc - 1) // 3 + 1
            f_alg = CNN_FLOP_PER_POS * L1 * Rs
            conv_ms = sum(v for k, v in kavg.items() if k.startswith("k_cnn_conv"))
            split = os.environ.get("ADP_CNN_CONV", "split") != "f32"
            f_eq = f_alg / (conv_ms * 1e-3) / 1e12  # float32-equivalent (algorithmic) TFLOP/s of the whole conv stack
            if split:
                # the default stack: the two 64 -> 64 layers run THREE float16 MFMAs per block of products (hi hi, hi lo, lo hi of
                # split float32 operands, cnn_conv_split.h), the first / last layer stay float32 vector code -- priced against the
                # float16 matrix peak for what the matrix cores execute, with the float32-equivalent rate beside it
                f64l = 2.0 * (2 * 64 * 64 * 7) * L1 * Rs
                row_passes = 4.0  # split activation rows the two kernels move through HBM: in and out of each
                if os.environ.get("ADP_CNN_FOLD", "1") != "0":
                    # layer 3 rides in layer 2's kernel as one more GEMM on the matrix cores: 16 (of the instruction's 32) rows x 64 channels per
                    # position, on tiles that overlap by 2 of their 64 NT positions -- counted as executed (the padded rows, not the overlap)
                    f64l += 2.0 * (32 * 64) * L1 * Rs
                    row_passes -= 1.0  # layer 2's rows never leave the chip
                if os.environ.get("ADP_CNN_FUSE_IN", "1") != "0":
                    # layer 0 rides in layer 1's kernel: 64 channels x 16 k-values (the 7 taps padded to the instruction's 16) per position
                    f64l += 2.0 * (64 * 16) * L1 * Rs
                    row_passes -= 1.0  # layer 0's rows are made in LDS
                ms64 = sum(v for k, v in kavg.items() if k.startswith("k_cnn_conv64"))
                rows_b = 272.0 * L1 * Rs  # one split activation row per position
                out["roofline"] = {"bound": "mfma", "dtype": "f16 x 3 (split float32 operands, float32 accumulate)",
                                   "kernel": "k_cnn_conv64 x 2 (split float16 MFMA)",
                                   "achieved": 3.0 * f64l / (ms64 * 1e-3) / 1e12, "peak": F16_MFMA_PEAK_TF, "unit": "TFLOP/s",
                                   "frac": 3.0 * f64l / (ms64 * 1e-3) / 1e12 / F16_MFMA_PEAK_TF, "traffic": None,
                                   "kernel_ms": ms64, "executed_flop_per_launch": 3.0 * f64l, "algorithmic_flop_per_launch": f_alg,
                                   "hbm_GBps_of_the_two_layers": row_passes * rows_b / (ms64 * 1e-3) / 1e9,
                                   "conv_stack_ms": conv_ms, "f32_equivalent_tflops": f_eq,
                                   # the step's float32-equivalent FLOP/s over the rate this stack can reach at best: three float16 MFMAs per
                                   # block of products = a third of the float16 matrix peak (833 TF); never above 1
                                   "whole_path_frac": CNN_FLOP_PER_POS * L1 * R / step_s / 1e12 / (F16_MFMA_PEAK_TF / 3.0),
                                   # the same rates over the FLOAT32 matrix pipe's peak (157.3 TF, the bound SURVEY 8(d) names): speed-ups, not
                                   # fractions -- the stack left that pipe
                                   "conv_stack_speed_vs_f32_matrix_pipe": f_eq / F32_MFMA_PEAK_TF,
                                   "whole_path_speed_vs_f32_matrix_pipe": CNN_FLOP_PER_POS * L1 * R / step_s / 1e12 / F32_MFMA_PEAK_TF,
                                   "whole_path_hbm_frac": (4.0 * mean_samples * R / step_s) / 1e9 / HBM_PEAK_GBS,
                                   "slowest_kernel": dom, "slowest_kernel_ms": kavg[dom]}
                out["dtype"] = "f32 (conv net: split float16 MFMA at float32 accuracy; statistics f32)"
                out["config"]["workload"] = out["config"]["workload"].replace("hand-written float32 MFMA conv stack", "hand-written conv stack, split float16 MFMA at float32 accuracy")
            else:
                out["roofline"] = {"bound": "mfma", "dtype": "f32", "kernel": "conv stack (k_cnn_conv_in + k_cnn_conv64 x 2 + k_cnn_conv_out)",
                                   "achieved": f_eq, "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                                   "frac": f_eq / F32_MFMA_PEAK_TF, "traffic": None,
                                   "kernel_ms": conv_ms, "algorithmic_flop_per_launch": f_alg,
                                   "whole_path_frac": CNN_FLOP_PER_POS * L1 * R / step_s / 1e12 / F32_MFMA_PEAK_TF,
                                   "whole_path_hbm_frac": (4.0 * mean_samples * R / step_s) / 1e9 / HBM_PEAK_GBS,
                                   "slowest_kernel": dom, "slowest_kernel_ms": kavg[dom]}
        if raw_t is not None:
            out["config"]["workload"] = out["config"]["workload"].replace("RNA004 LLR", "RNA004 LLR over RAW int16 ADC samples + per-read calibration (2 B per sample in HBM; extension of the float32 boundary)")
            out["config"]["rows_of_first_minibatch_equal_float32_path"] = i16_check
            out["dtype"] = "int16 samples calibrated to f32 in registers, f32 statistics, f64 cumulative sums + LLR trace"
        if w.lens != "full":
            out["config"]["lens"] = "%s: mean %.0f samples of m = %d preloaded (%.0f %% of the matrix is NaN padding)" % (
                w.lens, mean_samples, m, 100.0 * (1.0 - mean_samples / m))
        if w.adc_step > 0 or w.lens != "full":
            c = eng.debug_counters(24)  # k_partition_stats tallies over the large segments (cumulative over all steps)
            if w.adc_step > 0:
                out["config"]["adc_step_pa"] = w.adc_step
            out["partition_paths"] = {"large_segments": int(c[0]), "mad_proven_in_bracket": int(c[1]), "median_generic_select": int(c[2]),
                                      "mad_not_predicted": int(c[3]), "mad_bracket_overflow": int(c[4]),
                                      "n1_fused_minibatches": int(c[5]), "n1_fused_fallbacks": int(c[6]) + int(c[7]),
                                      "n1_heavy_keys": int(c[22]), "n1_heavy_samples": int(c[23])}
        if world == 1 and not multi and w.cpu_sample > 0 and w.primary == "llr" and w.lens == "full" and w.adc_step == 0 and raw_t is None:
            n_s = min(w.cpu_sample, R)
            n_all = min(w.cpu_sample_all, R // max(cpu_procs, 1)) if cpu_procs > 1 else 0
            out["cpu_baseline"] = cpu_baseline(eng, spc, sig_t.data_ptr(), n_s, m, rows, lens_host, cpu_pool, cpu_procs if n_all > 0 else 0,
                                               n_all, w.max_obs_trace)
        else:
            out["cpu_baseline"] = None
    for e in engines:
        e.close()
    del sig_t, rows_t, rows_bufs, len_t, raw_t, cal_t
    state.clear()
    torch.cuda.empty_cache()
    return out


def launch_command(n_ranks: int, port: int, argv):
    """the launcher's own form of the driver's command: torchrun as a MODULE of this interpreter, one rank per GPU, 127.0.0.1"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks ourselves, as a CHILD process (this
    process never touches the GPU), relay rank 0's JSON line and exit with the child's code."""
    import socket
    import subprocess

    import torch

    have = torch.cuda.device_count()  # (counting devices does not initialise the GPU)
    if have < args.gpus and os.environ.get("ADP_BENCH_BACKEND", "nccl") == "nccl":  # (a gloo rehearsal shares devices)
        sys.stderr.write("bench.py: --gpus %d but only %d device(s) visible\n" % (args.gpus, have))
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = launch_command(args.gpus, port, argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in proc.stdout:
        if ln.startswith('{"metric"'):
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line is not None:
        print(line)
    elif rc == 0:
        rc = 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=None,
                    help="reads per step per GPU (whole minibatches); default 96 000 (LLR: 96 000 x 806 KB = 77 GB of signal + ~55 GB of "
                         "workspace in the 288 GB HBM) or 4 000 (CNN at the 200 k window) / 32 000 (CNN, shorter windows)")
    ap.add_argument("--minibatch", type=int, default=1000)
    ap.add_argument("--max_obs_trace", type=int, default=200000)
    ap.add_argument("--cpu-sample", type=int, default=1000, help="reads timed on the CPU oracle, one thread (rank 0, N=1)")
    ap.add_argument("--cpu-sample-all", type=int, default=500, help="reads per worker process of the all-cores CPU baseline")
    ap.add_argument("--cpu-procs", type=int, default=None, help="worker processes of the all-cores CPU baseline (default: every CPU the affinity "
                                                                "mask and the cgroup quota grant; 16 when neither restricts a big host)")
    ap.add_argument("--lens", choices=["full", "pareto"], default="full",
                    help="read lengths: full (the headline workload: every read fills the window) or pareto (BASELINE configs[4]: "
                         "Pareto(1.2) clipped to [10k, 1M] samples -- most reads much shorter than the window, NaN padded; a probe)")
    ap.add_argument("--ragged", action="store_true", help="with --host-pipeline: reads packed back to back in the staging slots, "
                                                          "the padded matrix laid out on the device (adp_expand_ragged)")
    ap.add_argument("--adc-step", type=float, default=0.0,
                    help="(robustness probe, not the headline workload) round the synthetic samples to multiples of this many pA, "
                         "like calibrated int16 ADC data (~0.18 pA): exercises the tie handling of the exact selections")
    ap.add_argument("--no-start-peak", action="store_true")
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--primary", choices=["llr", "cnn"], default="llr",
                    help="llr: BASELINE configs[1] (default, the headline); cnn: configs[2] (hand-written conv head)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary workloads (configs[2] at the 200 k and the default window, configs[4]'s Pareto lengths, int16 input, two streams) that a "
                         "default 1-GPU run attaches to its JSON line (and the three PCIe-inclusive host-pipeline runs)")
    ap.add_argument("--with-grouped", action="store_true", help="add the grouped-execution timeline experiment (ADP_GROUPS=9 over 3 lanes) to the secondaries")
    ap.add_argument("--streams", type=int, default=1,
                    help="engines (HIP streams) per GPU; each owns reads/streams whole minibatches and runs in its own host thread")
    ap.add_argument("--host-pipeline", type=int, default=0, metavar="N",
                    help="instead of the resident benchmark: stream N minibatches from HOST memory through adapted_amd.pipeline "
                         "(pinned staging, H2D overlapped with detect) and print the PCIe-inclusive rate -- never the headline value")
    ap.add_argument("--group", type=int, default=4, help="with --host-pipeline: minibatches per staging slot / detect call")
    ap.add_argument("--fill-threads", type=int, default=1, help="with --host-pipeline: host threads copying a slot's reads (the stand-in reader)")
    ap.add_argument("--int16", action="store_true", help="the resident workload as RAW int16 ADC samples + per-read calibration (every kernel reads them "
                                                         "natively: adp_detect_llr_i16); with --host-pipeline: stream raw int16 samples from host memory")
    args = ap.parse_args()

    world_env = int(os.environ.get("WORLD_SIZE", "0"))
    if args.gpus > 1 and world_env == 0:  # no launcher around us: be the launcher (before anything touches the GPU)
        raise SystemExit(spawn_ranks(args, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = max(world_env, 1)
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE = %d" % (args.gpus, world))
    if args.reads is None:
        args.reads = 96000 if args.primary == "llr" else (4000 if args.max_obs_trace > 32000 else 32000)
    default_run = (args.primary == "llr" and args.lens == "full" and args.adc_step == 0 and args.host_pipeline == 0 and
                   args.max_obs_trace == 200000 and not args.no_start_peak and not args.int16)

    # the all-cores CPU baseline runs in worker processes: start them before this process initialises the GPU
    cpu_pool, cpu_procs = None, 0
    if world == 1 and args.cpu_sample > 0 and args.cpu_sample_all > 0 and default_run and os.environ.get("ADP_BENCH_FORCE_DIST", "0") != "1":
        global _CPU_WHY
        cpu_procs, _CPU_WHY = cpu_worker_count(args.cpu_procs)
        if cpu_procs > 1:
            cpu_pool = _cpu_pool_start(cpu_procs)

    import torch

    dist = None
    # ADP_BENCH_BACKEND=gloo rehearses the N > 1 code path on fewer GPUs than ranks (ranks share devices, the row gather
    # goes through host memory); the real multi-GPU run uses RCCL ("nccl"), one GPU per rank
    backend = os.environ.get("ADP_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and world > ndev:
        raise SystemExit("bench.py: %d ranks but only %d device(s) visible" % (world, ndev))
    if backend != "nccl":
        local = local % max(ndev, 1)
    torch.cuda.set_device(local)
    # ADP_BENCH_FORCE_DIST=1: the N > 1 code path on a process group of ONE (RCCL init, double-buffered row gather, barrier,
    # all-reduce of the clock) -- how the RCCL branch is exercised on a one-GPU box (tests/test_gpu_rccl.py)
    force_dist = os.environ.get("ADP_BENCH_FORCE_DIST", "0") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist

        if world_env == 0:  # (forced, no launcher: a rendezvous of our own)
            import socket

            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints its version banner on STDOUT when its first communicator comes up; the contract is ONE JSON line there:
        # file descriptor 1 points at stderr while the group and its first collective are made
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend, rank=rank, world_size=world)
            if backend == "nccl":
                t0 = torch.zeros(1, device=torch.device("cuda", local))
                dist.all_reduce(t0)
                torch.cuda.synchronize()
            else:
                dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    if args.host_pipeline > 0:
        print(json.dumps(host_pipeline_bench(args, make_spc(args.max_obs_trace), local)))
        return
    out = run_workload(args, rank, world, local, dist, backend, cpu_pool, cpu_procs)
    if cpu_pool is not None:
        cpu_pool.close()
        cpu_pool.join()
    if rank == 0 and dist is not None:
        from adapted_amd import lib as _lib

        out["dist"] = {"backend": backend, "world_size": world, "forced_group_of_one": bool(force_dist and world == 1),
                       "hip_runtimes": _lib.hip_runtimes(), "row_gather": "dist.gather of %d-byte rows, double-buffered" % _lib.ROW_DTYPE.itemsize}
    if rank == 0 and world == 1 and dist is None and default_run and not args.no_secondary:
        # the other single-GPU configurations of BASELINE.json, driver-run with the headline (each its own roofline)
        sec = {}
        detail = {"headline": {"kernel_ms": out["kernel_ms"], "ms_per_step": out["ms_per_step"]}}
        # (24 000 reads per step: the one-wave-per-chain series kernel takes the same ~11 ms for 1000 and for 24 000 reads)
        plan = [("cnn_200k", dict(primary="cnn", max_obs_trace=200000, reads=24000, steps=3, warmup=1)),
                # the same step on the exact-float32 MFMA conv stack (cnn_conv.h): the round-2 kernels, 0.75 of the float32 matrix peak
                ("cnn_200k_f32_stack", dict(primary="cnn", max_obs_trace=200000, reads=24000, steps=2, warmup=1, env={"ADP_CNN_CONV": "f32"})),
                ("cnn_default", dict(primary="cnn", max_obs_trace=16000, reads=32000, steps=4, warmup=1)),
                ("pareto", dict(lens="pareto", steps=4, warmup=1)),
                # the reference's own defaults: the preset's 16 000-sample window, 1000 reads per minibatch
                ("llr_default_window", dict(max_obs_trace=16000, steps=4, warmup=1))]
        # (the int16-resident workload left the line in round 5: its rows equal the float32 path's, its step takes the same time --
        # `--int16` still runs it)
        if args.with_grouped:
            # the headline cut into groups of minibatches software-pipelined over three internal streams (adp_detect_llr's opt-in
            # grouped execution): a TIMELINE experiment -- its kernels overlap each other, so it carries no roofline object
            plan.append(("grouped", dict(env={"ADP_GROUPS": "9", "ADP_LANES": "3", "ADP_STAGGER": "0"}, steps=4, warmup=1)))
        for name, kw in plan:
            w = argparse.Namespace(**vars(args))
            w.cpu_sample = 0
            env = kw.pop("env", {})
            for k, v in kw.items():
                setattr(w, k, v)
            saved = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            try:
                o = run_workload(w, 0, 1, local, None, backend)
            finally:
                for k, v in saved.items():
                    os.environ.pop(k, None)
                    if v is not None:
                        os.environ[k] = v
            detail[name] = {"kernel_ms": o["kernel_ms"], "ms_per_step": o["ms_per_step"], "roofline": o["roofline"], "config": o["config"]}
            rf = o["roofline"]
            sec[name] = {"value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"], "steps": o["steps"],
                         "workload": o["config"]["workload"], "pass_rate": o["config"]["pass_rate"],
                         "roofline": None if name == "grouped" else
                         {k: rf[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "whole_path_frac",
                                             "f32_equivalent_tflops", "conv_stack_speed_vs_f32_matrix_pipe", "whole_path_speed_vs_f32_matrix_pipe",
                                             "slowest_kernel", "slowest_kernel_ms") if k in rf}}
        out["secondary"] = sec
        # PCIe-inclusive (SURVEY 8(d) "report H2D-inclusive end-to-end separately"): 12 minibatches from pinned host staging through
        # adapted_amd.pipeline -- the padded float32 matrix, raw int16 samples + device calibration, and Pareto lengths packed back to back
        hp = {}
        for name, kw in (("f32_padded", dict()), ("int16_padded", dict(int16=True)), ("int16_ragged_pareto", dict(int16=True, ragged=True, lens="pareto"))):
            w = argparse.Namespace(**vars(args))
            w.host_pipeline, w.fill_threads, w.group = 12, 4, 4
            for k, v in kw.items():
                setattr(w, k, v)
            try:
                o = host_pipeline_bench(w, make_spc(args.max_obs_trace), local)
                hp[name] = {"value": o["value"], "unit": o["unit"], "h2d_GB_per_s": o["h2d_GB_per_s"], "input": o["input"], "lens": o["lens"],
                            "minibatches": o["minibatches"], "without_host_assembly": o["without_host_assembly"], "pass_rate": o["pass_rate"]}
            except Exception as e:  # (a box short of pinnable host memory: the headline stands without it)
                hp[name] = {"error": repr(e)[:200]}
        sec["host_pipeline"] = hp
        # per-kernel times of every workload of this run: a file beside the line (the line itself must survive a log tail)
        try:
            ddir = os.path.join(ROOT, "gpurun_out")
            os.makedirs(ddir, exist_ok=True)
            with open(os.path.join(ddir, "bench_kernels.json"), "w") as fh:
                json.dump(detail, fh, indent=1)
            out["kernel_ms_file"] = "gpurun_out/bench_kernels.json"
        except OSError:
            pass
        # LAST on the line, compact: every BASELINE configuration that fits one GPU as [reads/s, roofline.frac, whole_path_frac]
        # (roofline: HBM for the LLR workloads, the matrix cores for the CNN ones -- the objects above say which)
        out["summary"] = {"headline": [round(out["value"]), round(out["roofline"]["frac"], 4), round(out["roofline"]["whole_path_frac"], 4)]}
        for name, o in sec.items():
            if name == "host_pipeline":
                continue
            rf = o["roofline"] or {}
            out["summary"][name] = [round(o["value"]), round(rf["frac"], 4) if "frac" in rf else None,
                                    round(rf["whole_path_frac"], 4) if "whole_path_frac" in rf else None]
        # PCIe-inclusive entries: [reads/s, GB/s across PCIe] -- never the headline value
        out["summary"]["host_pipeline"] = {k: ([round(v["value"]), round(v["h2d_GB_per_s"], 1)] if "value" in v else None) for k, v in sec["host_pipeline"].items()}
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
