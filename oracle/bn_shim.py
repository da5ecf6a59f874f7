"""Build-owned restatement of bottleneck 1.3.x ``move_mean`` / ``move_var`` for float32
input (reference adapted/detect/mvs.py:93-106), NaN samples included: they are counted out of the
window, and a window with fewer than ``min_count`` (= window) valid samples yields NaN.  TEST INFRASTRUCTURE.

bottleneck is a third-party dependency of the reference (unpinned in its setup.py:35-42;
environment.yml pins nothing for it); its algorithm is the published streaming recurrence
of bottleneck/src/move_template.c: every intermediate is held in the INPUT dtype (float32
here), the first full window divides by the count, later windows multiply by a float32
reciprocal.  Pinned against the real bottleneck 1.3.2 (conda python3.9 in this container)
by the golden vectors in tests/golden/bn_move.npz and bn_move_nan.npz (made with the real library,
oracle/gen_golden.py gen_bottleneck) -- tests/test_oracle_primitives.py.
"""
import numpy as np

F = np.float32


def _check(a, window):
    a = np.asarray(a)
    if a.dtype != np.float32:
        raise TypeError("bn_shim only restates the float32 path")
    if window < 1 or window > a.size:
        raise ValueError("Moving window (=%d) must between 1 and %d, inclusive" % (window, a.size))
    return a


def move_mean(a, window, min_count=None):
    a = _check(a, window)
    mc = window if min_count is None else min_count
    n = a.size
    out = np.empty(n, dtype=np.float32)
    asum = F(0)
    count = 0
    for i in range(window):
        ai = a[i]
        if ai == ai:
            asum = F(asum + ai)
            count += 1
        out[i] = F(asum / F(count)) if (i >= mc - 1 and count >= mc) else np.nan
    with np.errstate(divide="ignore"):
        inv = F(np.float64(1.0) / np.float64(count))  # double division rounded to f32
        for i in range(window, n):
            ai = a[i]
            aold = a[i - window]
            if ai == ai:
                if aold == aold:
                    asum = F(asum + F(ai - aold))
                else:
                    asum = F(asum + ai)
                    count += 1
                    inv = F(np.float64(1.0) / np.float64(count))
            elif aold == aold:
                asum = F(asum - aold)
                count -= 1
                inv = F(np.float64(1.0) / np.float64(count))
            out[i] = F(asum * inv) if count >= mc else np.nan
    return out


def move_var(a, window, min_count=None, ddof=0):
    a = _check(a, window)
    mc = window if min_count is None else min_count
    n = a.size
    out = np.empty(n, dtype=np.float32)
    amean = F(0)
    assqdm = F(0)
    count = 0
    for i in range(window):
        ai = a[i]
        if ai == ai:
            count += 1
            delta = F(ai - amean)
            amean = F(amean + F(delta / F(count)))
            assqdm = F(assqdm + F(delta * F(ai - amean)))
        if i >= mc - 1 and count >= mc:
            if assqdm < 0:
                assqdm = F(0)
            out[i] = F(assqdm / F(count - ddof))
        else:
            out[i] = np.nan
    with np.errstate(divide="ignore"):
        count_inv = F(np.float64(1.0) / np.float64(count))
        ddof_inv = F(np.float64(1.0) / np.float64(count - ddof))
        for i in range(window, n):
            ai = a[i]
            aold = a[i - window]
            if ai == ai:
                if aold == aold:
                    delta = F(ai - aold)
                    aold = F(aold - amean)
                    amean = F(amean + F(delta * count_inv))
                    ai = F(ai - amean)
                    assqdm = F(assqdm + F(F(ai + aold) * delta))
                else:
                    count += 1
                    count_inv = F(np.float64(1.0) / np.float64(count))
                    ddof_inv = F(np.float64(1.0) / np.float64(count - ddof))
                    delta = F(ai - amean)
                    amean = F(amean + F(delta * count_inv))
                    assqdm = F(assqdm + F(delta * F(ai - amean)))
            elif aold == aold:
                count -= 1
                count_inv = F(np.float64(1.0) / np.float64(count))
                ddof_inv = F(np.float64(1.0) / np.float64(count - ddof))
                if count > 0:
                    delta = F(aold - amean)
                    amean = F(amean - F(delta * count_inv))
                    assqdm = F(assqdm - F(delta * F(aold - amean)))
                else:
                    amean = F(0)
                    assqdm = F(0)
            if count >= mc:
                if assqdm < 0:
                    assqdm = F(0)
                out[i] = F(assqdm * ddof_inv)
            else:
                out[i] = np.nan
    return out
