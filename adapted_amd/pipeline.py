"""Host-side streaming pipeline for `adapted detect`.

The reference moves minibatches between processes through pickling Manager queues
(adapted/file_proc.py:612-823); at MI355X speeds the detect step takes ~1.5 ms per 1000 reads, so the
host side is all that is left: assembling the float32 [N, m] minibatch and getting it across PCIe
(806 KB per read at the 200 k window -> about 78 k reads/s per GPU at 63 GB/s).  This module keeps that
path busy and nothing else in the way:

  producer thread   fills pinned (page-locked) staging slots in place -- no intermediate copy -- from any
                    iterator of (signals, lengths, ids) writers (io_utils.yield_minibatches(buffers=...));
  transfer/compute  (caller's thread) per slot: one asynchronous H2D copy on a copy stream, the detect call
                    on device pointers, the 528-byte result rows back; the H2D of slot k+1 is started
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
                 with_start_peak: bool = False, model=None):
        self.spc, self.N, self.m, self.device = spc, int(minibatch), int(m), int(device)
        self.primary, self.with_start_peak, self.model = primary, with_start_peak, model
        self.eng = lib.Engine(spc, self.N, self.m, device=self.device)
        self.slots = []
        for _ in range(min(16, max(2, n_slots))):
            sig = self.eng.host_alloc((self.N, self.m), np.float32)
            lens = self.eng.host_alloc((self.N,), np.int32)
            self.slots.append({"sig": sig, "lens": lens, "ds": self.eng.dev_alloc(self.N * self.m * 4),
                               "dl": self.eng.dev_alloc(self.N * 4)})
        self.free: "queue.Queue[int]" = queue.Queue()
        for i in range(len(self.slots)):
            self.free.put(i)

    def close(self):
        for s in self.slots:
            self.eng.host_free(s["sig"])
            self.eng.host_free(s["lens"])
            self.eng.dev_free(s["ds"])
            self.eng.dev_free(s["dl"])
        self.slots = []
        self.eng.close()

    # -- stages -------------------------------------------------------------------------------
    def _start_h2d(self, j: int, n: int):
        s = self.slots[j]
        self.eng.h2d_async(s["ds"], s["sig"], n * self.m * 4)
        self.eng.h2d_async(s["dl"], s["lens"], n * 4)
        self.eng.copy_mark(j)

    def _detect(self, j: int, n: int) -> np.ndarray:
        s = self.slots[j]
        self.eng.copy_wait(j)  # this slot's copies only: the next slot's may still be in flight
        dsig, dlen = s["ds"], s["dl"]
        if self.primary == "llr":
            rows, mbs = self.eng.detect_llr_rows(dsig, dlen, n, n, with_start_peak=self.with_start_peak, device_ptrs=True)
            if mbs[0] != lib.MB_OK:
                raise lib.MinibatchDropped(int(mbs[0]))
            return rows
        if self.primary == "start_peak":
            return self.eng.detect_start_peak_rows(dsig, dlen, n, n, device_ptrs=True)
        from .detect import cnn as _cnn

        return _cnn.detect_rows_device(self.eng, dsig, dlen, n, s["lens"][:n], self.model, self.spc)

    # -- driver -------------------------------------------------------------------------------
    def run(self, fill: Callable[[Callable[[], Tuple[np.ndarray, np.ndarray]]], Iterable[Tuple[int, object]]],
            on_rows: Callable[[object, np.ndarray], None], on_dropped: Optional[Callable[[object, int], None]] = None) -> int:
        """fill(get_buffers) -> iterator of (n, ids): every item announces that the buffers handed out by the LAST
        get_buffers() call now hold n reads.  Returns the number of reads processed."""
        filled: "queue.Queue" = queue.Queue(maxsize=len(self.slots))
        done: "queue.Queue" = queue.Queue(maxsize=4 * len(self.slots))
        err = []
        cur = {"j": None}

        def get_buffers():
            j = self.free.get()
            cur["j"] = j
            return self.slots[j]["sig"], self.slots[j]["lens"]

        def producer():
            try:
                for n, ids in fill(get_buffers):
                    filled.put((cur["j"], int(n), ids))
            except BaseException as e:  # noqa: BLE001 -- handed to the caller's thread
                err.append(e)
            finally:
                filled.put(None)

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
                    try:
                        rows = self._detect(j, n)
                        done.put((ids, rows))
                        total += n
                    except lib.MinibatchDropped as e:
                        if on_dropped:
                            on_dropped(ids, e.status)
                    self.free.put(j)
                pending = item
                if item is None:
                    break
                if err:
                    break
        finally:
            done.put(None)
            tc.join()
        if err:
            raise err[0]
        return total
