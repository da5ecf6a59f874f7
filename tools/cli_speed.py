"""End-to-end rate of the CLI (`python -m adapted_amd.main detect`) on a synthetic .npz bundle of short reads: reader,
packed staging, H2D, detect, row conversion, CSV writing.  (GPU box.)
    python tools/cli_speed.py [n_reads]
"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from adapted_amd import synth  # noqa: E402
from adapted_amd import main as cli  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    m = 201500
    tmp = tempfile.mkdtemp()
    lens = np.array([min(synth.pareto_length(5, i), 60000) for i in range(n)], dtype=np.int32)
    ids = np.array(["read_%06d" % i for i in range(n)], dtype=object)
    arrs = {"full_lengths": lens, "read_ids": ids}
    t0 = time.time()
    for i in range(n):
        arrs["signal_%d" % i] = synth.synth_read(5, i % 64, int(lens[i]))  # (64 distinct reads, cut to length: cheap to make)
    np.savez(os.path.join(tmp, "reads_0.npz"), **arrs)
    print("bundle of %d reads (mean %.0f samples) written in %.1f s" % (n, lens.mean(), time.time() - t0), flush=True)
    out = os.path.join(tmp, "out")
    t0 = time.time()
    from adapted_amd.config import get_chemistry_specific_config

    spc = get_chemistry_specific_config("RNA004")
    spc.llr_boundaries.llr_detect, spc.cnn_boundaries.cnn_detect = True, False  # the LLR primary (the preset's default is the CNN)
    spc.core.max_obs_trace = 200000
    spc.update_primary_method()
    spc.update_sig_preload_size()
    cfg = os.path.join(tmp, "cfg.toml")
    spc.to_toml(cfg)
    argv = ["detect", "-i", os.path.join(tmp, "reads_0.npz"), "-o", out, "--config", cfg]
    if len(sys.argv) > 2 and sys.argv[2] == "profile":
        import cProfile
        import pstats

        pr = cProfile.Profile()
        pr.enable()
        cli.main(argv)
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
    else:
        cli.main(argv)
    dt = time.time() - t0
    print("CLI: %d reads in %.2f s = %.0f reads/s" % (n, dt, n / dt))


if __name__ == "__main__":
    main()
