"""Host -> HBM ingest of uint8 exemplar chunks, overlapped with compute.

SURVEY.md section 8(f) rank 1: once the GPU side is fast the reference's
"whole dataset as fp32 in host RAM, synchronous pageable `.to(device)` per
batch" (`src/milan/decoders.py:860-861`) becomes the bottleneck.  Here a chunk
stays uint8 (3.76 MB per neuron instead of 15 MB), is staged by a worker thread
into one of two pinned buffers (the numpy copy out of the page cache releases
the GIL) and travels on a side HIP stream while the previous chunk computes.

Plumbing only (torch streams / events / pinned memory); no arithmetic.
"""
import queue
import threading
from typing import Callable, Iterator, Optional, Tuple

import torch

Chunk = Tuple[torch.Tensor, Optional[torch.Tensor]]


class ChunkPrefetcher:
    """Iterate device-resident (images, masks) chunks with one-chunk lookahead.

    `fetch(i)` returns CPU uint8 (or float) tensors for chunk i (masks may be
    None); it runs in a worker thread.  The yielded tensors live in two
    rotating device buffers: consume chunk i before asking for chunk i+2.
    """

    def __init__(self, fetch: Callable[..., Chunk], n_chunks: int,
                 device: torch.device, depth: int = 2,
                 alloc: Optional[Callable[[], Chunk]] = None):
        """`alloc()` (optional) returns one slot's pinned (images, masks)
        staging buffers, sized for the largest chunk; `fetch(i, out)` is then
        called with them and fills them in place (one pass over the bytes
        instead of fetch + copy)."""
        self.fetch, self.n, self.device, self.depth = fetch, n_chunks, device, depth
        self.alloc = alloc
        self.copy_stream = torch.cuda.Stream(device=device)
        self._q: 'queue.Queue' = queue.Queue()
        self._free = threading.Semaphore(depth)  # pinned slots not in flight
        self._pinned = [None] * depth
        self._worker = threading.Thread(target=self._produce, daemon=True)
        self._error: Optional[BaseException] = None

    def _pin_like(self, slot: int, t: Optional[torch.Tensor], which: int):
        if t is None:
            return None
        if t.is_pinned():
            return t  # already DMA-able: no staging copy
        bufs = self._pinned[slot] or [None, None]
        buf = bufs[which]
        if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
            buf = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        buf.copy_(t)  # page cache / pageable -> pinned, GIL released in C++
        bufs[which] = buf
        self._pinned[slot] = bufs
        return buf

    def _produce(self):
        try:
            for i in range(self.n):
                self._free.acquire()  # slot's previous H2D has completed
                slot = i % self.depth
                if self.alloc is not None:
                    if self._pinned[slot] is None:
                        self._pinned[slot] = list(self.alloc())
                    images, masks = self.fetch(i, tuple(self._pinned[slot]))
                else:
                    images, masks = self.fetch(i)
                self._q.put((i, self._pin_like(slot, images, 0),
                             self._pin_like(slot, masks, 1)))
        except BaseException as error:  # surfaced in the consumer
            self._error = error
            self._q.put(None)

    def __iter__(self) -> Iterator[Chunk]:
        self._worker.start()
        main = torch.cuda.current_stream(self.device)
        done_events = [None] * self.depth  # compute finished reading slot
        dev = [None] * self.depth

        def enqueue(item):
            """H2D of one staged chunk on the copy stream -> (slot, images, masks, ready)."""
            if item is None:
                raise self._error
            i, images, masks = item
            slot = i % self.depth
            with torch.cuda.stream(self.copy_stream):
                if done_events[slot] is not None:
                    self.copy_stream.wait_event(done_events[slot])
                d_im = images.to(self.device, non_blocking=True)
                d_mk = None if masks is None else masks.to(self.device,
                                                           non_blocking=True)
                ready = torch.cuda.Event()
                ready.record(self.copy_stream)
            dev[slot] = (d_im, d_mk)  # keep alive until reused
            d_im.record_stream(main)
            if d_mk is not None:
                d_mk.record_stream(main)
            return slot, d_im, d_mk, ready

        ahead = None
        for k in range(self.n):
            cur = ahead if ahead is not None else enqueue(self._q.get())
            ahead = None
            # One-chunk lookahead on the COPY queue too: if chunk k+1 is already staged, its H2D
            # is requested before chunk k is handed out.  A consumer that synchronises inside
            # its call (round 5's encoder did: a 4-byte read-back; round 6's does not, but a
            # guarded call still reads the status word) and then queues a result D2H would
            # otherwise submit that D2H -- blocked until chunk k is computed -- ahead of the
            # next H2D, which then ran AFTER chunk k instead of under it (measured: 34 ms per
            # 640-neuron chunk, tools/debug_host_loop.py).  Never waits for a slow fetch.
            if k + 1 < self.n:
                try:
                    ahead = enqueue(self._q.get(timeout=0.002))
                except queue.Empty:
                    ahead = None
            slot, d_im, d_mk, ready = cur
            main.wait_event(ready)
            yield d_im, d_mk
            ev = torch.cuda.Event()
            ev.record(main)
            done_events[slot] = ev
            # the pinned buffer of this slot may be refilled once its H2D copy
            # is done (long done by now: the chunk has been computed on)
            ready.synchronize()
            self._free.release()
