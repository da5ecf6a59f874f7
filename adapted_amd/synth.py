"""Synthetic dRNA squiggles: host (numpy) twin of the device generator in csrc/synth.hip.

The generator is counter based and uses integer hashing plus two correctly rounded f32
operations per sample (one multiply, one add, never fused), so the host and the device
produce BIT-IDENTICAL float32 samples for a given (seed, read index, sample index).  Any
read of a device-generated benchmark batch can therefore be regenerated on the host and
pushed through the CPU oracle.

Shape of a read: adapter N(80, 7^2) for U[2500,4500) samples,
poly(A) N(108, 2.5^2) for U[400,2500) samples, then RNA events: levels N(95, 14^2) held
12 samples with N(0, 3^2) noise on top.  "Normal" deviates are Irwin-Hall sums of eight
hash bytes (integer arithmetic), scaled in f32.  Optional decorations (chosen per read by
hash bits): an RNA004-style start peak inside the first 1500 samples and a short open-pore
blip (>= 200 pA) inside the adapter.
"""
from __future__ import annotations

import numpy as np

_M32 = np.uint64(0xFFFFFFFF)

# stream ids
_S_PARAMS, _S_NOISE, _S_LEVEL = 1, 2, 3

# Irwin-Hall(8 bytes): mean 1020, variance 8*(256^2-1)/12
IH_MEAN = 1020
IH_SD = float(np.sqrt(8.0 * (256.0**2 - 1.0) / 12.0))

EVENT_LEN = 12

FLAG_START_PEAK = 1
FLAG_OPEN_PORE = 2


def _lowbias32(x: np.ndarray) -> np.ndarray:
    """32-bit integer mixer (uint64 carrier, masked)."""
    x = x & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & _M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & _M32
    x ^= x >> np.uint64(16)
    return x


def _base(seed: int, read: int, stream: int) -> np.uint64:
    h = _lowbias32(np.uint64((seed * 0x27D4EB2F + read) & 0xFFFFFFFF))
    return _lowbias32((h + np.uint64(stream * 0x9E3779B1 & 0xFFFFFFFF)) & _M32)


def _hash(base: np.uint64, ctr: np.ndarray, salt: int) -> np.ndarray:
    x = (base + ctr.astype(np.uint64) * np.uint64(0x9E3779B1)
         + np.uint64(salt * 0x85EBCA77 & 0xFFFFFFFF)) & _M32
    return _lowbias32(x)


def _bytesum(h: np.ndarray) -> np.ndarray:
    return ((h & np.uint64(0xFF)) + ((h >> np.uint64(8)) & np.uint64(0xFF))
            + ((h >> np.uint64(16)) & np.uint64(0xFF)) + ((h >> np.uint64(24)) & np.uint64(0xFF)))


def _z(base: np.uint64, ctr: np.ndarray) -> np.ndarray:
    """Approximately N(0, IH_SD^2) integer deviate as exact float32."""
    s = _bytesum(_hash(base, ctr, 0)) + _bytesum(_hash(base, ctr, 1))
    return (s.astype(np.int64) - IH_MEAN).astype(np.float32)


def read_params(seed: int, read: int, decorate: bool = True):
    """Per-read structure (integers only)."""
    pb = _base(seed, read, _S_PARAMS)
    h = _hash(pb, np.arange(4, dtype=np.uint64), 0)
    adapter_len = 2500 + int(h[0] % np.uint64(2000))
    polya_len = 400 + int(h[1] % np.uint64(2100))
    flags = 0
    sp_start = op_start = 0
    if decorate:
        if int(h[2] & np.uint64(3)) == 0:  # 25 %: start peak
            flags |= FLAG_START_PEAK
            sp_start = 200 + int((h[2] >> np.uint64(8)) % np.uint64(900))
        if int(h[3] % np.uint64(32)) == 0:  # ~3 %: open-pore blip inside the adapter
            flags |= FLAG_OPEN_PORE
            op_start = 100 + int((h[3] >> np.uint64(8)) % np.uint64(adapter_len - 200))
    return adapter_len, polya_len, flags, sp_start, op_start


SP_LEN = 300   # start-peak length (samples)
OP_LEN = 30    # open-pore blip length (samples)

_F = np.float32


def synth_read(seed: int, read: int, m: int, full_len: int | None = None,
               decorate: bool = True) -> np.ndarray:
    """float32[m] raw pA samples of read ``read``; NaN from ``full_len`` on (B0 padding)."""
    a_len, p_len, flags, sp_start, op_start = read_params(seed, read, decorate)
    i = np.arange(m, dtype=np.uint64)
    nb = _base(seed, read, _S_NOISE)
    lb = _base(seed, read, _S_LEVEL)
    z = _z(nb, i)
    rna0 = a_len + p_len
    ev = np.maximum(np.arange(m, dtype=np.int64) - rna0, 0) // EVENT_LEN
    zl = _z(lb, ev.astype(np.uint64))
    level = _F(95.0) + zl * _F(14.0 / IH_SD)          # f32 mul, f32 add
    mean = np.where(i < np.uint64(a_len), _F(80.0),
                    np.where(i < np.uint64(rna0), _F(108.0), level)).astype(np.float32)
    sd = np.where(i < np.uint64(a_len), _F(7.0 / IH_SD),
                  np.where(i < np.uint64(rna0), _F(2.5 / IH_SD), _F(3.0 / IH_SD))).astype(np.float32)
    if flags & FLAG_START_PEAK:
        sel = (i >= np.uint64(sp_start)) & (i < np.uint64(sp_start + SP_LEN))
        mean = np.where(sel, _F(150.0), mean).astype(np.float32)
    if flags & FLAG_OPEN_PORE:
        sel = (i >= np.uint64(op_start)) & (i < np.uint64(op_start + OP_LEN))
        mean = np.where(sel, _F(230.0), mean).astype(np.float32)
    x = (mean + z * sd).astype(np.float32)            # f32 mul, f32 add
    if full_len is not None and full_len < m:
        x[full_len:] = np.nan
    return x


def pareto_length(seed: int, read: int, lo: int = 10_000, hi: int = 1_000_000,
                  alpha: float = 1.2) -> int:
    """Integer read length ~ Pareto(alpha) clipped to [lo, hi] (host only; lengths are
    handed to the device generator as an int32 array)."""
    h = int(_hash(_base(seed, read, 7), np.arange(1, dtype=np.uint64), 0)[0])
    u = (h + 0.5) / 4294967296.0
    return int(min(hi, max(lo, lo * (1.0 - u) ** (-1.0 / alpha))))


def synth_batch(seed: int, first_read: int, n: int, m: int, full_lens=None,
                decorate: bool = True):
    """(signals f32[n, m] NaN padded, full_lens i32[n]) -- the reference's minibatch layout
    (reference adapted/file_proc.py:143-190)."""
    sig = np.empty((n, m), dtype=np.float32)
    if full_lens is None:
        full_lens = np.full(n, m, dtype=np.int32)
    full_lens = np.asarray(full_lens, dtype=np.int32)
    for k in range(n):
        sig[k] = synth_read(seed, first_read + k, m, int(full_lens[k]), decorate)
    return sig, full_lens
