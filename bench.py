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


def make_spc(max_obs_trace: int):
    from adapted_amd.config import get_chemistry_specific_config

    spc = get_chemistry_specific_config("RNA004")
    spc.llr_boundaries.llr_detect = True
    spc.cnn_boundaries.cnn_detect = False
    spc.core.max_obs_trace = max_obs_trace
    spc.update_primary_method()
    spc.update_sig_preload_size()
    return spc


def cpu_baseline(eng, spc, dsig, n_sample: int, m: int, gpu_rows, lens=None):
    """Time the CPU oracle (a single-threaded C port of the reference path) on the first
    n_sample reads of the resident batch, as ONE minibatch, and check the GPU rows of an
    identically composed minibatch against it."""
    from adapted_amd import lib
    from oracle import oracle

    oracle.lib()
    sig = np.zeros((n_sample, m), dtype=np.float32)
    eng.d2h(sig, dsig)
    lens = np.full(n_sample, m, dtype=np.int32) if lens is None else np.ascontiguousarray(lens[:n_sample], dtype=np.int32)
    t0 = time.perf_counter()
    want = oracle.detect_llr(sig, lens, spc, with_start_peak=True)
    dt = time.perf_counter() - t0
    rows, _ = eng.detect_llr_rows(sig, lens, n_sample, n_sample, with_start_peak=True)
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
    return {"value": n_sample / dt, "unit": "reads/s", "cores": 1, "kind": "port",
            "sample": "%d reads (one minibatch) of the same synthetic workload, oracle/adapted_oracle.c, "
                      "1 thread, %.1f s; GPU rows of the same minibatch differ from it in %d fields"
                      % (n_sample, dt, mism)}


def host_pipeline_bench(args, spc, device):
    """PCIe-inclusive rate: minibatches assembled in host memory (a memcpy per minibatch out of a small pool stands in for
    the reader), pinned staging, H2D, detect, rows back.  One JSON line of its own."""
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
    print(json.dumps({"metric": "reads/sec (adapter+polyA detect), RNA004 200k-sample reads, HOST buffers (PCIe-inclusive; not the headline)",
                      "value": total / dt, "unit": "reads/s", "n_gpus": 1, "minibatches": args.host_pipeline,
                      "h2d_GB_per_s": gb / dt, "pass_rate": n_ok[0] / max(total + total2, 1),
                      "without_host_assembly": {"value": total2 / dt2, "h2d_GB_per_s": total2 * per_read * bps / 1e9 / dt2},
                      "input": ("int16 ADC + device calibration" if i16 else "float32 pA") + (", reads packed back to back (ragged)" if ragged else ", padded [N, m] matrix"),
                      "lens": "%s (mean %.0f of m = %d samples preloaded)" % (args.lens, float(take.mean()), m), "fill_threads": K,
                      "config": {"workload": "LLR + start_peak + validate from pinned host staging, m=%d, minibatch=%d, %d minibatches per call" % (m, args.minibatch, G)}}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=96000,
                    help="reads per step per GPU (whole minibatches); 96 000 x 806 KB = 77 GB of signal + ~55 GB of workspace in the 288 GB HBM")
    ap.add_argument("--minibatch", type=int, default=1000)
    ap.add_argument("--max_obs_trace", type=int, default=200000)
    ap.add_argument("--cpu-sample", type=int, default=1000, help="reads timed on the CPU oracle (rank 0, N=1)")
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
                    help="llr: BASELINE configs[1] (default); cnn: configs[2] (PyTorch-ROCm conv head, per-minibatch calls)")
    ap.add_argument("--streams", type=int, default=1,
                    help="engines (HIP streams) per GPU; each owns reads/streams whole minibatches and runs in its own host thread")
    ap.add_argument("--host-pipeline", type=int, default=0, metavar="N",
                    help="instead of the resident benchmark: stream N minibatches from HOST memory through adapted_amd.pipeline "
                         "(pinned staging, H2D overlapped with detect) and print the PCIe-inclusive rate -- never the headline value")
    ap.add_argument("--group", type=int, default=4, help="with --host-pipeline: minibatches per staging slot / detect call")
    ap.add_argument("--fill-threads", type=int, default=1, help="with --host-pipeline: host threads copying a slot's reads (the stand-in reader)")
    ap.add_argument("--int16", action="store_true", help="with --host-pipeline: stream raw int16 ADC samples and calibrate on the device")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus must equal WORLD_SIZE")
    dist = None
    # ADP_BENCH_BACKEND=gloo rehearses the N > 1 code path on fewer GPUs than ranks (ranks share devices, the row gather
    # goes through host memory); the real multi-GPU run uses RCCL ("nccl"), one GPU per rank
    backend = os.environ.get("ADP_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    if world > 1:
        import torch.distributed as dist

        torch.cuda.set_device(local)
        dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    comm_dev = dev if backend == "nccl" else torch.device("cpu")

    from adapted_amd import lib

    spc = make_spc(args.max_obs_trace)
    if args.host_pipeline > 0:
        return host_pipeline_bench(args, spc, local)
    if args.primary == "cnn":
        spc.llr_boundaries.llr_detect = False
        spc.cnn_boundaries.cnn_detect = True
        spc.update_primary_method()
        from adapted_amd.detect import cnn as cnn_mod

        cnn_model = None  # the weights named in the config, handed to the engine once (no PyTorch module involved)
    m = spc.sig_preload_size
    R, mb = args.reads, args.minibatch
    assert R % mb == 0, "--reads must be a whole number of minibatches"
    NS = max(1, args.streams)
    assert (R // mb) % NS == 0, "--reads must split into whole minibatches per stream"
    Rs = R // NS
    engines = [lib.Engine(spc, Rs, m, device=local) for _ in range(NS)]
    eng = engines[0]
    sig_t = torch.empty((R, m), dtype=torch.float32, device=dev)
    lens_host = np.full(R, m, dtype=np.int32)
    if args.lens == "pareto":
        from adapted_amd import synth as _synth

        lens_host = np.array([_synth.pareto_length(args.seed, rank * R + i) for i in range(R)], dtype=np.int32)
    len_t = torch.from_numpy(lens_host).to(dev)
    rows_t = torch.empty((R, lib.ROW_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    # rank r owns reads [r*R, (r+1)*R) of the global stream: contiguous whole minibatches
    for k, e in enumerate(engines):
        e.synth_fill(sig_t.data_ptr() + k * Rs * m * 4, len_t.data_ptr() + k * Rs * 4, Rs, seed=args.seed,
                     first_read=rank * R + k * Rs, decorate=True)
    if args.adc_step > 0:
        for s0 in range(0, R, 1000):
            sig_t[s0:s0 + 1000].div_(args.adc_step).round_().mul_(args.adc_step)
        torch.cuda.synchronize()
    gathered = None
    if world > 1 and rank == 0:
        gathered = [torch.empty_like(rows_t, device=comm_dev) for _ in range(world)]

    import threading

    def run_part(k):
        if args.primary == "cnn":  # one library call (adp_detect_cnn) per engine; find_peaks / row compaction per minibatch
            s0 = k * Rs
            rows = cnn_mod.detect_rows_device(engines[k], sig_t.data_ptr() + s0 * m * 4, len_t.data_ptr() + s0 * 4, Rs,
                                              lens_host[s0:s0 + Rs], cnn_model, spc, minibatch=mb)
            engines[k].h2d(rows_t.data_ptr() + s0 * lib.ROW_DTYPE.itemsize, rows)
            return
        engines[k].detect_llr_rows(sig_t.data_ptr() + k * Rs * m * 4, len_t.data_ptr() + k * Rs * 4, Rs, mb,
                                   with_start_peak=not args.no_start_peak, device_ptrs=True,
                                   rows_dev=rows_t.data_ptr() + k * Rs * lib.ROW_DTYPE.itemsize,
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
        if world > 1:
            dist.gather(rows_t if backend == "nccl" else rows_t.cpu(), gathered, dst=0)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    for e in engines:
        e.set_profiling(True)
    ktimes = {}
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        for e in engines:
            for name, ms in e.kernel_times():  # HIP events recorded on each engine's stream
                ktimes.setdefault(name, []).append(ms)
    sync()
    dt = time.perf_counter() - t0
    for e in engines:
        e.set_profiling(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    total_reads = R * world * args.steps
    value = total_reads / dt

    if rank == 0:
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "r01_traffic.json")  # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
        if os.path.exists(tfile):
            with open(tfile) as fh:
                traffic = json.load(fh)
        rows = np.zeros(R, dtype=lib.ROW_DTYPE)
        eng.d2h(rows, rows_t.data_ptr())
        n_ok = int(rows["success"].sum())
        kavg = {k: float(np.mean(v)) for k, v in ktimes.items()} or {"(torch conv stack + host top-k)": dt / args.steps * 1e3}
        dom = max(kavg, key=kavg.get)
        b_alg = 4.0 * m * Rs  # 4*m input bytes per read, each launch covers Rs reads
        achieved = b_alg / (kavg[dom] * 1e-3) / 1e9
        ksum = sum(kavg.values())
        out = {
            "metric": "reads/sec (adapter+polyA detect), RNA004 200k-sample reads",
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 samples/statistics, f64 cumulative sums + LLR trace", "data": "synthetic (device-generated, adapted_amd/synth.py twin)",
            "config": {"workload": ("BASELINE configs[1]: RNA004 LLR + start_peak + validate" if args.primary == "llr" else
                                    "BASELINE configs[2]: RNA004 CNN head (PyTorch-ROCm fp32) + validate") +
                                   ", max_obs_trace=%d (m=%d), minibatch=%d, %d reads/step/GPU resident in HBM"
                                   % (args.max_obs_trace, m, mb, R),
                       "reads_per_step_per_gpu": R, "minibatch": mb, "m": m, "pass_rate": n_ok / R, "streams_per_gpu": NS},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": (traffic.get(dom, {}).get("hbm_bytes") * (Rs / traffic["_reads_per_launch"])
                                     if traffic and dom in traffic else None),
                         "traffic_source": "profiles/r01_traffic.json (rocprofv3 --pmc, FETCH_SIZE doubled per MI355X_MICROARCH.md)",
                         "kernel_ms": kavg[dom], "algorithmic_bytes_per_launch": b_alg,
                         "whole_path_frac": (4.0 * m * R / (dt / args.steps)) / 1e9 / HBM_PEAK_GBS},
            "kernel_ms": {k: round(v, 4) for k, v in sorted(kavg.items(), key=lambda kv: -kv[1])},
            "kernel_ms_sum": ksum,
        }
        if args.lens != "full":
            out["config"]["lens"] = "%s: mean %.0f samples of m = %d preloaded (%.0f %% of the matrix is NaN padding)" % (
                args.lens, float(np.minimum(lens_host, m).mean()), m, 100.0 * (1.0 - float(np.minimum(lens_host, m).mean()) / m))
        if args.adc_step > 0 or args.lens != "full":
            c = eng.debug_counters(24)  # k_partition_stats tallies over the large segments (cumulative over all steps)
            if args.adc_step > 0:
                out["config"]["adc_step_pa"] = args.adc_step
            out["partition_paths"] = {"large_segments": int(c[0]), "mad_proven_in_bracket": int(c[1]), "median_generic_select": int(c[2]),
                                      "mad_not_predicted": int(c[3]), "mad_bracket_overflow": int(c[4]),
                                      "n1_fused_minibatches": int(c[5]), "n1_fused_fallbacks": int(c[6]) + int(c[7]),
                                      "n1_heavy_keys": int(c[22]), "n1_heavy_samples": int(c[23]), "n1_dbg": [int(c[19]), int(c[18]), int(c[17]), int(c[6]), int(c[7])]}
        if world == 1 and args.cpu_sample > 0 and args.primary == "llr":
            n_s = min(args.cpu_sample, R)
            out["cpu_baseline"] = cpu_baseline(eng, spc, sig_t.data_ptr(), n_s, m, rows, lens_host)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    for e in engines:
        e.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
