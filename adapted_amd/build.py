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


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in sources())


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
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
