"""Host-side streaming pipeline for `adapted detect`.

The reference moves minibatches between processes through pickling Manager queues
(adapted/file_proc.py:612-823); at MI355X speeds the detect step takes ~1.5 ms per 1000 reads, so the
host side is all that is left: assembling the float32 [N, m] minibatch and getting it across PCIe
(806 KB per read at the 200 k window -> about 78 k reads/s per GPU at 63 GB/s).  This module keeps that
path busy and nothing else in the way:

  producer thread   fills pinned (page-locked) staging slots in place -- no intermediate copy -- from any
                    iterator of (signals, lengths, ids) writers (io_utils.yield_minibatches(buffers=...));
  transfer/compute  (caller's thread) per slot: one asynchronous H2D copy on a copy stream, the detect call
                    on device pointers, the 544-byte result rows back; the H2D of slot k+1 is started
                    BEFORE the detect call of slot k, so copies and kernels overlap;
  consumer          `on_rows(ids, rows)` (CSV writer) runs in a third thread.

Pinned memory, device buffers and the copy stream come from the HIP library itself (adp_host_alloc,
adp_dev_alloc, adp_memcpy_h2d_async on the handle's copy stream): the LLR / start-peak path needs no PyTorch.
"""
from __future__ import annotations

import queue
import threading
from typing import Callable, Iterable, Optional, Tuple

import numpy as np

from . import lib


class HostPipeline:
    def __init__(self, spc, minibatch: int, m: int, device: int = 0, n_slots: int = 3, primary: str = "llr",
                 with_start_peak: bool = False, model=None, int16_input: bool = False, group: int = 1, ragged: bool = False):
        """int16_input: the staging slots hold raw ADC samples (int16) plus per-read (scale, offset); they are calibrated
        to float32 pA on the device (adp_calibrate_i16), so only 2 bytes per sample cross PCIe.  get_buffers() then hands
        out (raw, lengths, scale, offset) instead of (signals, lengths)."""
        """group: minibatches per staging slot and per detect call (a call over several minibatches fills the GPU better
        than one over 1000 reads; normalisation stays per minibatch)."""
        """ragged: the staging slots hold the reads packed back to back (flat array + offsets int64 [N + 1]); only the samples
        that exist cross PCIe and the NaN-padded [N, m] minibatch is laid out on the device (adp_expand_ragged).  With
        heavy-tailed read lengths most of the padded matrix is padding.  get_buffers() hands out (flat, lengths, offsets)
        -- plus (scale, offset) with int16_input."""
        self.spc, self.mb, self.m, self.device = spc, int(minibatch), int(m), int(device)
        self.N = self.mb * max(1, int(group))  # reads per slot
        self.primary, self.with_start_peak, self.model, self.i16 = primary, with_start_peak, model, bool(int16_input)
        self.ragged = bool(ragged)
        self.eng = lib.Engine(spc, self.N, self.m, device=self.device)
        self.slots = []
        # int16 input + LLR primary: the kernels read the raw samples themselves (adp_detect_llr_i16) -- no float32 matrix is made
        self.native_i16 = self.i16 and primary == "llr" and self.m % 4 == 0
        # the float32 minibatch made on the device (calibrated and / or laid out from packed reads); one: detect is serial
        # (native int16 + packed reads: the raw int16 matrix instead)
        self.dsig16 = None
        if self.native_i16:
            self.dsig16 = self.eng.dev_alloc(self.N * self.m * 2) if self.ragged else None
        elif self.i16 or self.ragged:
            self.dsig16 = self.eng.dev_alloc(self.N * self.m * 4)
        # staging slots are made when first handed out (page-locking a 3.2 GB slot takes about a second: a short run pins only
        # the slots it uses, and the second and third are pinned by the producer thread while the first is being processed)
        self.slots = [None] * min(16, max(2, n_slots))
        self.free: "queue.Queue[int]" = queue.Queue()
        for i in range(len(self.slots)):
            self.free.put(i)

    def _slot(self, j: int):
        if self.slots[j] is None:
            sig = self.eng.host_alloc((self.N * self.m,) if self.ragged else (self.N, self.m), np.int16 if self.i16 else np.float32)
            lens = self.eng.host_alloc((self.N,), np.int32)
            slot = {"sig": sig, "lens": lens, "ds": self.eng.dev_alloc(self.N * self.m * (2 if self.i16 else 4)),
                    "dl": self.eng.dev_alloc(self.N * 4)}
            if self.ragged:
                slot["offs"] = self.eng.host_alloc((self.N + 1,), np.int64)
                slot["do"] = self.eng.dev_alloc((self.N + 1) * 8)
            if self.i16:
                slot["cal"] = self.eng.host_alloc((2, self.N), np.float32)  # scale, offset
                slot["dcal"] = self.eng.dev_alloc(2 * self.N * 4)
            self.slots[j] = slot
        return self.slots[j]

    def close(self):
        for s in self.slots:
            if s is None:
                continue
            self.eng.host_free(s["sig"])
            self.eng.host_free(s["lens"])
            self.eng.dev_free(s["ds"])
            self.eng.dev_free(s["dl"])
            if "cal" in s:
                self.eng.host_free(s["cal"])
                self.eng.dev_free(s["dcal"])
            if "offs" in s:
                self.eng.host_free(s["offs"])
                self.eng.dev_free(s["do"])
        if self.dsig16:
            self.eng.dev_free(self.dsig16)
            self.dsig16 = None
        self.slots = []
        self.eng.close()

    # -- stages -------------------------------------------------------------------------------
    def _start_h2d(self, j: int, n: int):
        s = self.slots[j]
        if self.ragged:  # only the samples that exist
            self.eng.h2d_async(s["ds"], s["sig"], int(s["offs"][n]) * (2 if self.i16 else 4))
            self.eng.h2d_async(s["do"], s["offs"], (n + 1) * 8)
        else:
            self.eng.h2d_async(s["ds"], s["sig"], n * self.m * (2 if self.i16 else 4))
        self.eng.h2d_async(s["dl"], s["lens"], n * 4)
        if self.i16:
            self.eng.h2d_async(s["dcal"], s["cal"])
        self.eng.copy_mark(j)

    def _detect(self, j: int, n: int):
        """-> (rows, per-minibatch status or None)"""
        s = self.slots[j]
        self.eng.copy_wait(j)  # this slot's copies only: the next slot's may still be in flight
        dsig, dlen = s["ds"], s["dl"]
        if self.native_i16:
            if self.ragged:
                self.eng.expand_ragged_i16(dsig, s["do"], dlen, n, self.dsig16)
                dsig = self.dsig16
            return self.eng.detect_llr_rows_i16(dsig, dlen, s["dcal"], s["dcal"] + self.N * 4, n, self.mb, with_start_peak=self.with_start_peak)
        if self.ragged:  # packed reads (-> calibrated) -> float32 [n, m], NaN beyond each read
            if self.i16:
                self.eng.expand_ragged(dsig, True, s["do"], dlen, n, self.dsig16, s["dcal"], s["dcal"] + self.N * 4)
            else:
                self.eng.expand_ragged(dsig, False, s["do"], dlen, n, self.dsig16)
            dsig = self.dsig16
        elif self.i16:  # raw ADC -> float32 pA, NaN beyond the read, on the engine's stream ahead of the detect kernels
            self.eng.calibrate_i16(dsig, dlen, s["dcal"], s["dcal"] + self.N * 4, n, self.dsig16)
            dsig = self.dsig16
        if self.primary == "llr":
            # (the staging slots are NaN padded by the reader / the on-device calibration: the passes may stop at each read's end)
            rows, mbs = self.eng.detect_llr_rows(dsig, dlen, n, self.mb, with_start_peak=self.with_start_peak, device_ptrs=True,
                                                 tails_nan=True)
            return rows, mbs
        if self.primary == "start_peak":
            return self.eng.detect_start_peak_rows(dsig, dlen, n, self.mb, device_ptrs=True), None
        from .detect import cnn as _cnn

        # (the reference runs find_peaks and its row compaction per minibatch: adapted/detect/cnn.py:136-160)
        return _cnn.detect_rows_device(self.eng, dsig, dlen, n, s["lens"][:n], self.model, self.spc, minibatch=self.mb), None

    # -- driver -------------------------------------------------------------------------------
    def run(self, fill: Callable[[Callable[[], Tuple[np.ndarray, np.ndarray]]], Iterable[Tuple[int, object]]],
            on_rows: Callable[[object, np.ndarray], None], on_dropped: Optional[Callable[[object, int], None]] = None) -> int:
        """fill(get_buffers) -> iterator of (n, ids): every item announces that the buffers handed out by the LAST
        get_buffers() call now hold n reads (up to group * minibatch; ids: a sequence of n ids, sliced per minibatch when
        one is dropped -- or any tag when group == 1).  Returns the number of reads processed."""
        filled: "queue.Queue" = queue.Queue(maxsize=len(self.slots))
        done: "queue.Queue" = queue.Queue(maxsize=4 * len(self.slots))
        err = []
        cur = {"j": None}

        def get_buffers():
            j = self.free.get()
            cur["j"] = j
            sl = self._slot(j)
            head = (sl["sig"], sl["lens"], sl["offs"]) if self.ragged else (sl["sig"], sl["lens"])
            if self.i16:
                return head + (sl["cal"][0], sl["cal"][1])
            return head

        stop = threading.Event()

        def put_filled(item):
            while not stop.is_set():
                try:
                    filled.put(item, timeout=0.2)
                    return
                except queue.Full:
                    continue

        def producer():
            gen = fill(get_buffers)
            try:
                for n, ids in gen:
                    if stop.is_set():
                        break
                    put_filled((cur["j"], int(n), ids))
            except BaseException as e:  # noqa: BLE001 -- handed to the caller's thread
                err.append(e)
            finally:
                if hasattr(gen, "close"):
                    try:
                        gen.close()  # (lets the reader shut its copy pool down)
                    except BaseException as e:  # noqa: BLE001
                        err.append(e)
                put_filled(None)

        def consumer():
            while True:
                item = done.get()
                if item is None:
                    return
                try:
                    on_rows(*item)
                except BaseException as e:  # noqa: BLE001
                    err.append(e)

        tp = threading.Thread(target=producer, daemon=True)
        tc = threading.Thread(target=consumer, daemon=True)
        tp.start()
        tc.start()
        total = 0
        pending = None  # slot whose H2D is in flight
        try:
            while True:
                item = filled.get()
                if item is not None:
                    self._start_h2d(item[0], item[1])
                if pending is not None:
                    j, n, ids = pending
                    rows, mbs = self._detect(j, n)
                    if mbs is None or (mbs == lib.MB_OK).all():
                        done.put((ids, rows))
                        total += n
                    else:  # some minibatch of the group was dropped (the reference logs it and goes on)
                        for q, st in enumerate(mbs):
                            a, b = q * self.mb, min(n, (q + 1) * self.mb)
                            sub = ids[a:b] if hasattr(ids, "__getitem__") and not isinstance(ids, tuple) else ids
                            if st == lib.MB_OK:
                                done.put((sub, rows[a:b]))
                                total += b - a
                            elif on_dropped:
                                on_dropped(sub, int(st))
                    self.free.put(j)
                pending = item
                if item is None:
                    break
                if err:
                    break
        finally:
            # an error (or an exception out of _detect) may leave the producer mid-slot: it has to be gone before the caller
            # frees the pinned slots it writes into (close()), so stop it, unblock it and wait for it
            stop.set()
            while True:
                try:
                    filled.get_nowait()
                except queue.Empty:
                    break
            for j in range(len(self.slots)):
                self.free.put(j)
            tp.join(timeout=120)
            producer_gone = not tp.is_alive()
            done.put(None)
            tc.join()
            try:
                self.eng.copy_wait(-1)  # no H2D copy may still be reading a pinned slot when the caller frees them
            except Exception as e:  # noqa: BLE001
                err.append(e)
            if not producer_gone:
                # still inside a read / decode that does not return: its slots must outlive it -- close() will not free them
                self.slots = [None] * len(self.slots)
                err.insert(0, RuntimeError("the reader thread did not stop within 120 s; its pinned staging slots are left allocated"))
            # the free list back to one token per slot for the next run()
            while True:
                try:
                    self.free.get_nowait()
                except queue.Empty:
                    break
            for j in range(len(self.slots)):
                self.free.put(j)
        if err:
            raise err[0]
        return total
