"""`adapted detect` / `adapted continue` on MI355X.

Keeps the reference's command line (adapted/parser.py:37-269), run-directory layout
(adapted_<ver>_<uuid8>/{command.json, config.toml, adapted.log, boundaries/detected_boundaries_<k>.csv,
failed_reads/failed_reads_<k>.csv}) and CSV columns, but replaces the process pool and manager
queues of adapted/file_proc.py:612-823 with: one process per GPU, minibatches streamed into the
GPU engine, rows written in output batches of `-b` reads.  Launch under torchrun for several
GPUs: ranks own contiguous blocks of whole minibatches and rank 0 writes after one gather.
"""
from __future__ import annotations

import argparse
import json
import logging
import os
import shutil
import sys
import time
import uuid
from typing import List, Set

import numpy as np

from . import lib, parallel
from ._version import __version__
from .config import get_chemistry_specific_config, load_nested_config_from_file
from .container_types import ReadResult
from .io_utils import input_to_filelist, yield_minibatches
from .output import save_detected_boundaries


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog="adapted", formatter_class=argparse.RawTextHelpFormatter,
                                description="ADAPTed detect on MI355X: adapter and poly(A) boundaries in raw dRNA-seq signals.")
    sub = p.add_subparsers(dest="mode", required=True)
    d = sub.add_parser("detect", help="Detect adapter and poly(A) signal boundaries and calculate statistics.")
    c = sub.add_parser("continue", help="Continue processing from a previous incomplete run.")
    c.add_argument("continue_from", type=str)
    d.add_argument("-i", "--input", type=str, nargs="+", required=True, help="input files / directories (.pod5 or .npz bundles)")
    d.add_argument("-o", "--output", type=str, default=None)
    d.add_argument("--config", type=str, default=None, help="config TOML (overrides --chemistry)")
    d.add_argument("-c", "--chemistry", type=str, choices=["RNA002", "RNA004"], default=None)
    d.add_argument("--max_obs_trace", type=int, default=None)
    d.add_argument("--read_id_csv", type=str, default=None)
    d.add_argument("--read_id_csv_colname", type=str, default="read_id")
    d.add_argument("-j", "--num_proc", type=int, default=None, help="accepted for compatibility; the GPU engine does not use a process pool")
    d.add_argument("-b", "--batch_size", type=int, default=4000, help="Number of reads per output file.")
    d.add_argument("-s", "--minibatch_size", type=int, default=1000, help="Number of reads per minibatch (normalisation unit).")
    d.add_argument("--start_peak", action="store_true", help="(extension) also fill the start_peak_* columns on the LLR path")
    d.add_argument("--device", type=int, default=None, help="GPU index (default: LOCAL_RANK or 0)")
    d.add_argument("--int16_ingest", action="store_true",
                   help="(extension) move raw int16 ADC samples + calibration to the GPU and compute pA there "
                        "(pA = scale * (float32(adc) + offset)); .pod5 inputs or .npz bundles with raw/scale/offset")
    return p


def scan_processed_reads(run_dir: str):
    done: Set[str] = set()
    mx = {"failed_reads": -1, "boundaries": -1}
    for sub, prefix in (("failed_reads", "failed_reads_"), ("boundaries", "detected_boundaries_")):
        d = os.path.join(run_dir, sub)
        if not os.path.isdir(d):
            continue
        for f in os.listdir(d):
            if f.startswith(prefix) and f.endswith(".csv"):
                mx[sub] = max(mx[sub], int(f.split("_")[-1].split(".")[0]))
                with open(os.path.join(d, f)) as fh:
                    done.update(line.split(",")[0] for line in fh.readlines()[1:])
    return done, mx["boundaries"], mx["failed_reads"]


class _Writer:
    """Accumulates pass / fail results and flushes CSV files of `batch` reads each."""

    def __init__(self, run_dir: str, batch: int, bidx_pass: int = 0, bidx_fail: int = 0):
        self.dirs = {True: os.path.join(run_dir, "boundaries"), False: os.path.join(run_dir, "failed_reads")}
        for d in self.dirs.values():
            os.makedirs(d, exist_ok=True)
        self.names = {True: "detected_boundaries", False: "failed_reads"}
        self.bidx = {True: bidx_pass, False: bidx_fail}
        self.pending = {True: [], False: []}
        self.batch = batch
        self.n = {True: 0, False: 0}

    def add(self, results: List[ReadResult]):
        for r in results:
            self.pending[bool(r.success)].append(r)
        for ok in (True, False):
            while len(self.pending[ok]) >= self.batch:
                self._flush(ok, self.pending[ok][: self.batch])
                self.pending[ok] = self.pending[ok][self.batch:]

    def _flush(self, ok: bool, items: List[ReadResult]):
        fn = os.path.join(self.dirs[ok], "%s_%d.csv" % (self.names[ok], self.bidx[ok]))
        save_detected_boundaries(items, fn, save_fail_reasons=not ok)
        self.bidx[ok] += 1
        self.n[ok] += len(items)

    def close(self):
        for ok in (True, False):
            if self.pending[ok]:
                self._flush(ok, self.pending[ok])
                self.pending[ok] = []


def _init_dist(device=None):
    """one process per GPU: the process group (RCCL = "nccl"; ADAPTED_DIST_BACKEND=gloo for CPU-side rehearsals / tests)"""
    rank, ws, local = parallel.world()
    if ws <= 1 and os.environ.get("ADAPTED_DIST_FORCE", "0") != "1":  # (forced: the several-ranks code on a group of one)
        return None
    import torch
    import torch.distributed as dist

    backend = os.environ.get("ADAPTED_DIST_BACKEND", "nccl")
    if not dist.is_initialized():
        if backend == "nccl":
            torch.cuda.set_device(local if device is None else device)
        if "MASTER_ADDR" not in os.environ:  # (forced group of one without a launcher)
            import socket

            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            os.environ["MASTER_ADDR"] = "127.0.0.1"
        dist.init_process_group(backend, rank=rank, world_size=ws)
    return dist


def run_detect(files, read_ids_incl, read_ids_excl, spc, run_dir, minibatch, batch_out, device, start_peak=False,
               bidx_pass=0, bidx_fail=0, int16_ingest=False):
    rank, ws, local = parallel.world()
    if device is None:
        device = local
        if ws > 1 and os.environ.get("ADAPTED_DIST_BACKEND", "nccl") != "nccl":
            device = local % max(lib.load().adp_device_count(), 1)  # (a rehearsal may share devices)
    dist = _init_dist(device)
    multi = dist is not None  # several ranks (or a forced group of one): rows gathered to rank 0, written in stream order
    primary = spc.primary_method
    model = None  # CNN primary: the weights named in the config go to the engine on first use (no PyTorch module needed)
    m = spc.sig_preload_size
    writer = _Writer(run_dir, batch_out, bidx_pass, bidx_fail) if rank == 0 else None
    t0 = time.time()
    my_rows, my_ids, my_ord = [], [], []
    dropped_text = {1: "MAD normalization failed: scale is 0", 2: "a read has no signal after min_obs_adapter"}
    from .io_utils import GroupSharder
    from .pipeline import HostPipeline

    # pinned staging slots filled in place by a producer thread, H2D overlapped with the detect call, CSV writing in
    # a third thread (adapted_amd/pipeline.py).  The reads travel packed back to back and the padded matrix is laid out on
    # the device (only the samples that exist cross PCIe).  Several ranks: every rank walks the stream's METADATA and
    # decodes only the groups of whole minibatches a GroupSharder assigns to it (balanced by preloaded samples).
    GROUP = 4  # minibatches per staging slot and detect call (normalisation stays per minibatch)
    pipe = HostPipeline(spc, minibatch, m, device=device, primary=primary, with_start_peak=start_peak,
                        model=model, int16_input=int16_ingest, group=GROUP, ragged=True)
    sharder = GroupSharder(ws, rank, m) if multi else None
    ordinals: List[int] = []  # stream index of the first read of every group this rank yields

    def fill(get_buffers):
        from .io_utils import yield_minibatches_packed

        for k, ids in yield_minibatches_packed(files, read_ids_incl, read_ids_excl, minibatch * GROUP, m, get_buffers,
                                               int16=int16_ingest, sharder=sharder, ordinals=ordinals):
            # ids travel with the stream position of each read: (id, ordinal) pairs survive the slicing of dropped minibatches
            tagged = np.empty((k, 2), dtype=object)
            tagged[:, 0] = ids[:k]
            tagged[:, 1] = np.arange(ordinals[-1], ordinals[-1] + k)
            yield k, tagged

    def on_rows(tagged, rows):
        if multi:
            my_rows.append(rows)
            my_ids.extend(tagged[:, 0].tolist())
            my_ord.extend(int(x) for x in tagged[:, 1])
        else:
            res = lib.rows_to_results(rows, primary, consume=True)
            writer.add([ReadResult(read_id=str(rid), success=r.success, fail_reason=r.fail_reason, detect_results=r)
                        for rid, r in zip(tagged[:, 0], res)])

    def on_dropped(tagged, status):
        logging.error("minibatch of %d reads dropped: %s", len(tagged), dropped_text.get(status, status))

    try:
        pipe.run(fill, on_rows, on_dropped)
    finally:
        pipe.close()
    if multi:
        rows = np.concatenate(my_rows) if my_rows else np.zeros(0, dtype=lib.ROW_DTYPE)
        allrows = parallel.gather_rows(rows, dst=0, always=True)
        lists = [None] * ws if rank == 0 else None
        dist.gather_object((my_ids, my_ord), lists, dst=0)
        if rank == 0:
            ids = [x for part in lists for x in part[0]]
            order = np.argsort(np.array([x for part in lists for x in part[1]], dtype=np.int64), kind="stable")
            res = lib.rows_to_results(allrows[order], primary, consume=True)  # stream order: the files read like a one-GPU run's
            writer.add([ReadResult(read_id=str(ids[i]), success=r.success, fail_reason=r.fail_reason, detect_results=r)
                        for i, r in zip(order, res)])
    if writer is not None:
        writer.close()
        tot = writer.n[True] + writer.n[False]
        logging.info("Processed %d reads in %.2f s (%.0f reads/s on %d GPU(s))", tot, time.time() - t0,
                     tot / max(time.time() - t0, 1e-9), ws)
        if tot:
            logging.info("Pass: %d (%.2f%%), fail: %d", writer.n[True], 100.0 * writer.n[True] / tot, writer.n[False])
    if multi:
        dist.barrier()
        logging.info("process group: backend %s, %d rank(s); HIP runtimes mapped: %s", dist.get_backend(), ws, ", ".join(lib.hip_runtimes()))


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.mode == "continue":
        run_dir = args.continue_from
        try:
            with open(os.path.join(run_dir, "command.json")) as fh:
                cmd = json.load(fh)
        except FileNotFoundError:
            raise SystemExit("No command.json file found in the continue_from directory.")
        shutil.copy(os.path.join(run_dir, "command.json"), os.path.join(run_dir, "command_previous.json"))
        for k, v in cmd.items():
            if not hasattr(args, k):
                setattr(args, k, v)
    else:
        args.output = args.output or os.getcwd()
        run_dir = os.path.join(args.output, "adapted_" + __version__.replace(".", "_") + "_" + str(uuid.uuid4())[:8])
        dist = _init_dist(getattr(args, "device", None))
        if dist is not None:  # one run directory for all ranks: rank 0's name
            box = [run_dir]
            dist.broadcast_object_list(box, src=0)
            run_dir = box[0]
    if not args.config and not args.chemistry:
        raise SystemExit("Either --config or --chemistry must be provided.")
    read_ids: List[str] = []
    if args.read_id_csv:
        import pandas as pd

        read_ids = pd.read_csv(args.read_id_csv)[args.read_id_csv_colname].astype(str).tolist()
    files = input_to_filelist(args.input)
    if not files:
        print("No valid input files found.\nProvided path: {}".format(args.input))
        raise SystemExit(1)
    spc = load_nested_config_from_file(args.config) if args.config else get_chemistry_specific_config(args.chemistry)
    if args.max_obs_trace:
        spc.core.max_obs_trace = args.max_obs_trace
    spc.update_primary_method()
    spc.update_sig_preload_size()
    rank = parallel.world()[0]
    if rank == 0:
        os.makedirs(run_dir, exist_ok=True)
        with open(os.path.join(run_dir, "command.json"), "w") as fh:
            json.dump(vars(args), fh, indent=2)
        spc.to_toml(os.path.join(run_dir, "config.toml"))
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(levelname)s %(message)s",
                        handlers=[logging.StreamHandler(sys.stdout)] +
                        ([logging.FileHandler(os.path.join(run_dir, "adapted.log"))] if rank == 0 else []))
    logging.info("Command: %s", " ".join(sys.argv))
    logging.info("Saving output to: %s", run_dir)
    excl: Set[str] = set()
    bp = bf = 0
    if args.mode == "continue":
        excl, mp, mf = scan_processed_reads(run_dir)
        bp, bf = mp + 1, mf + 1
        logging.info("Found %d previously processed reads.", len(excl))
    run_detect(files, set(read_ids), excl, spc, run_dir, args.minibatch_size, args.batch_size, args.device,
               start_peak=getattr(args, "start_peak", False), bidx_pass=bp, bidx_fail=bf,
               int16_ingest=getattr(args, "int16_ingest", False))
    logging.info("Done.")


if __name__ == "__main__":
    main()
