"""Build libadapted_hip.so (hipcc, gfx950) in-tree.  Used by __graft_entry__.build() and on
first import when the library is missing or older than its sources."""
from __future__ import annotations

import glob
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "lib", "libadapted_hip.so")

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         # bit-exact float32/float64 arithmetic: no FMA contraction, IEEE divide/sqrt
         "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hip"))
                  + [os.path.join(ROOT, "include", "adapted_hip.h")])


STAMP = LIB + ".sources.sha256"  # digest of the sources the library was built from (travels with it; not in git)


def _digest() -> str:
    import hashlib

    h = hashlib.sha256()
    h.update(" ".join(FLAGS).encode())
    for s in sources():
        h.update(os.path.basename(s).encode())
        with open(s, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def stale() -> bool:
    """Is the library missing or built from other sources?  By content, not by time stamps: a copied tree (the GPU box
    gets a snapshot) keeps neither their order nor their values."""
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as fh:
        return fh.read().strip() != _digest()


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not stale():
        return LIB
    import fcntl

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    # several ranks of one node may get here at once (torchrun): one builds, the others wait and find it fresh; the
    # library appears by an atomic rename, never half written
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or stale():
                tmp = "%s.%d.tmp" % (LIB, os.getpid())
                cmd = [hipcc] + FLAGS + ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", tmp,
                                         os.path.join(CSRC, "adapted_hip.hip")]
                if verbose:
                    print(" ".join(cmd))
                subprocess.check_call(cmd)
                os.replace(tmp, LIB)
                with open(STAMP + ".tmp", "w") as fh:
                    fh.write(_digest() + "\n")
                os.replace(STAMP + ".tmp", STAMP)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
