"""Build-owned restatement of bottleneck 1.3.x ``move_mean`` / ``move_var`` for float32
input WITHOUT NaNs in the data (the only case on the hot path: reference
adapted/detect/mvs.py:93-106).  TEST INFRASTRUCTURE.

bottleneck is a third-party dependency of the reference (unpinned in its setup.py:35-42;
environment.yml pins nothing for it); its algorithm is the published streaming recurrence
of bottleneck/src/move_template.c: every intermediate is held in the INPUT dtype (float32
here), the first full window divides by the count, later windows multiply by a float32
reciprocal.  Pinned against the real bottleneck 1.3.2 (conda python3.9 in this container)
by tests/test_oracle_primitives.py::test_bn_shim_vs_real and by the golden vectors in
tests/golden/bn_*.npz.
"""
import numpy as np

F = np.float32


def move_mean(a, window, min_count=None):
    a = np.asarray(a)
    if a.dtype != np.float32:
        raise TypeError("bn_shim only restates the float32 path")
    n = a.size
    if window < 1 or window > n:
        raise ValueError("Moving window (=%d) must between 1 and %d, inclusive" % (window, n))
    out = np.empty(n, dtype=np.float32)
    asum = F(0)
    for i in range(window - 1):
        asum = F(asum + a[i])
        out[i] = np.nan
    i = window - 1
    asum = F(asum + a[i])
    out[i] = F(asum / F(window))
    inv = F(1.0 / window)  # double division rounded to f32
    for i in range(window, n):
        asum = F(asum + F(a[i] - a[i - window]))
        out[i] = F(asum * inv)
    return out


def move_var(a, window, min_count=None, ddof=0):
    a = np.asarray(a)
    if a.dtype != np.float32:
        raise TypeError("bn_shim only restates the float32 path")
    n = a.size
    if window < 1 or window > n:
        raise ValueError("Moving window (=%d) must between 1 and %d, inclusive" % (window, n))
    out = np.empty(n, dtype=np.float32)
    amean = F(0)
    assqdm = F(0)
    count = 0
    for i in range(window):
        ai = a[i]
        count += 1
        delta = F(ai - amean)
        amean = F(amean + F(delta / F(count)))
        assqdm = F(assqdm + F(delta * F(ai - amean)))
        if i < window - 1:
            out[i] = np.nan
        else:
            if assqdm < 0:
                assqdm = F(0)
            out[i] = F(assqdm / F(count - ddof))
    ddof_inv = F(1.0 / (count - ddof))
    count_inv = F(1.0 / count)
    for i in range(window, n):
        ai = a[i]
        aold = a[i - window]
        delta = F(ai - aold)
        aold = F(aold - amean)
        amean = F(amean + F(delta * count_inv))
        ai = F(ai - amean)
        assqdm = F(assqdm + F(F(ai + aold) * delta))
        if assqdm < 0:
            assqdm = F(0)
        out[i] = F(assqdm * ddof_inv)
    return out
