"""Cases of the `_c_llr` trace API (reference adapted/detect/_c_llr.pyx:67-236: `_gains`, `c_llr_trace_gains`, `c_llr_trace` with
stride and both early-stopping forms), shared by ``oracle/gen_trace_golden.py`` (runs the REAL reference's Cython module in the
build container) and the parity tests.  Python >= 3.8 syntax only."""
import numpy as np


def squiggle(seed, n, adapter=(400, 700), polya=(80, 260), nan_at=None):
    """a pooled, normalised read as float64: adapter (low level, wide), poly(A) (higher, narrow), RNA (events) -- the shape whose
    trace rises to the adapter end and falls again, so that the early-stopping rules fire"""
    rng = np.random.default_rng(seed)
    a = int(rng.integers(*adapter))
    p = int(rng.integers(*polya))
    x = np.empty(n, dtype=np.float64)
    x[:a] = rng.normal(-1.0, 0.9, a)[: min(a, n)]
    e = min(n, a + p)
    if e > a:
        x[a:e] = rng.normal(1.2, 0.25, e - a)
    if n > e:
        lv = np.repeat(rng.normal(0.3, 1.4, (n - e) // 3 + 1), 3)[: n - e]
        x[e:] = lv + rng.normal(0, 0.3, n - e)
    # (float32 values, like the pooled signal the reference hands over after .astype(float64))
    x = x.astype(np.float32).astype(np.float64)
    if nan_at is not None:
        x[nan_at] = np.nan
    return x


_D = dict(stride=1, adapter_early_stopping=0, adapter_early_stop_window=500, adapter_early_stop_stride=100,
          polya_early_stopping=0, polya_early_stop_window=50, polya_early_stop_stride=10)


def _c(name, seed, n, start, end, min_obs, border_trim, **kw):
    d = dict(_D)
    sig = {k: kw.pop(k) for k in ("adapter", "polya", "nan_at") if k in kw}
    d.update(kw)
    return dict(name=name, seed=seed, n=n, start=start, end=end, min_obs=min_obs, border_trim=border_trim, sig=sig, args=d)


CASES = [
    _c("plain_5_5", 1, 1990, 0, 1989, 5, 5),
    _c("plain_start300_1_1", 2, 1990, 300, 1989, 1, 1),
    _c("stride3", 3, 1500, 0, 1499, 5, 5, stride=3),
    _c("offset_head_0", 4, 300, 10, 299, 0, 1),                         # i = start: 0 * log(0) = NaN
    _c("end_before_size", 5, 800, 20, 640, 7, 9),
    _c("short_12", 6, 12, 0, 11, 5, 5),
    _c("short_11_empty_range", 6, 11, 0, 10, 5, 5),
    _c("nan_inside", 7, 900, 0, 899, 5, 5, nan_at=450),
    _c("adapter_es", 8, 3000, 0, 2999, 5, 5, adapter_early_stopping=1),
    _c("adapter_es_stride5", 9, 3000, 0, 2999, 5, 5, stride=5, adapter_early_stopping=1),
    _c("adapter_es_window_off_grid", 10, 3000, 0, 2999, 5, 5, stride=4, adapter_early_stopping=1, adapter_early_stop_window=250,
       adapter_early_stop_stride=100),
    _c("adapter_es_w1000_s500", 11, 4000, 0, 3999, 5, 5, adapter_early_stopping=1, adapter_early_stop_window=1000,
       adapter_early_stop_stride=500),
    _c("adapter_es_never", 12, 600, 0, 599, 5, 5, adapter_early_stopping=1),          # shorter than the first check
    _c("adapter_es_from_start", 13, 3200, 150, 3199, 1, 1, adapter_early_stopping=1, adapter_early_stop_window=300,
       adapter_early_stop_stride=50),
    _c("polya_es", 14, 3000, 0, 2999, 5, 5, polya_early_stopping=1),
    _c("polya_es_flags_both", 15, 3000, 0, 2999, 5, 5, adapter_early_stopping=1, polya_early_stopping=1),
    _c("polya_es_stride2", 16, 3000, 0, 2999, 5, 5, stride=2, polya_early_stopping=1),
    _c("polya_es_long_polya", 17, 4000, 0, 3999, 5, 5, polya_early_stopping=1, polya=(600, 900)),
    _c("polya_es_w_larger_than_adapter_w", 18, 2500, 0, 2499, 5, 5, polya_early_stopping=1, adapter_early_stop_window=20,
       adapter_early_stop_stride=10, polya_early_stop_window=50, polya_early_stop_stride=10),   # negative slice start: Python wraps it
    _c("polya_es_w1000", 19, 5000, 0, 4999, 5, 5, polya_early_stopping=1, adapter_early_stop_window=1000,
       adapter_early_stop_stride=500),
    _c("polya_es_w2_stride1", 20, 2000, 0, 1999, 5, 5, polya_early_stopping=1, polya_early_stop_window=2, polya_early_stop_stride=1),
    _c("adapter_es_200k_window", 21, 19900, 0, 19899, 5, 5, adapter_early_stopping=1, adapter=(3000, 5000), polya=(300, 1500)),
    _c("polya_es_200k_window", 22, 19900, 0, 19899, 5, 5, polya_early_stopping=1, adapter=(3000, 5000), polya=(300, 1500)),
]

# the reference's asserts (early-stop strides have to be multiples of the stride)
ASSERT_CASES = [
    _c("assert_adapter", 30, 1200, 0, 1199, 5, 5, stride=3, adapter_early_stopping=1),
    _c("assert_polya", 31, 1200, 0, 1199, 5, 5, stride=4, polya_early_stopping=1, adapter_early_stop_stride=100, polya_early_stop_stride=10),
]


def signal_of(case):
    return squiggle(case["seed"], case["n"], **case["sig"])
