"""Batch construction on a side stream (the device half of the reference's loader → trainer hand-over, learning/main.py:186-192).

With `--loader_device 1 --batch_device 1` everything a batch needs on the GPU is produced by kernels and asynchronous copies
issued from the collate (device loader, `GraphConvInfo.set_batch_device`, clouds / labels H2D).  Issued on the training stream
they queue BEHIND the previous step and delay the next one; `SideStreamBatches` runs the loader iterator under a second HIP
stream instead, so this device work overlaps the training step that is still executing, and hands every batch over with one
event (the training stream waits for it; the host never does).

The construction runs ONE batch ahead: batch j+1 is built (host work + side-stream launches) right after batch j has been
handed over and before the consumer enqueues step j, so its chain of small dependent launches has the whole of step j to finish.

Memory discipline (no `record_stream` bookkeeping needed): tensors of a batch are allocated in the side stream's pool and are
dropped by the consumer at the earliest when it receives the next batch.  When batch j is built the consumer still holds batch
j-2 (it receives j-1 only after this build), so the newest blocks that can have been freed -- and that this build may get again --
belong to batch j-3, last read by step j-3.  Before building batch j the side stream therefore waits for the event recorded on
the training stream at the PREVIOUS build (when steps <= j-3 had been enqueued), and overlaps with steps j-2 and j-1."""
from __future__ import annotations

import torch


_side_streams = {}      # device index -> the ONE side stream of that device


def side_stream() -> 'torch.cuda.Stream':
    """The batch-construction stream of the current device.  One long-lived stream: the caching allocator keeps a pool per
    stream, so a fresh stream per epoch would start every epoch with cold (hipMalloc-ed) buffers."""
    dev = torch.cuda.current_device()
    if dev not in _side_streams:
        _side_streams[dev] = torch.cuda.Stream(device=dev)
    return _side_streams[dev]


class SideStreamBatches:
    """for batch in SideStreamBatches(loader): ...  -- `loader` is any iterable whose `__next__` enqueues the batch's device work
    on the CURRENT stream (a `DataLoader` with `num_workers == 0` and a device collate, or a generator)."""

    def __init__(self, loader, stream: 'torch.cuda.Stream | None' = None):
        self.loader = loader
        self.side = stream

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        if not torch.cuda.is_available():
            raise RuntimeError('SideStreamBatches needs a GPU: the superpoint_graph_amd batch construction has no CPU path')
        main = torch.cuda.current_stream()
        side = self.side if self.side is not None else side_stream()
        it = iter(self.loader)
        fence_prev = [None]                   # recorded on `main` at the previous build

        def build():
            """-> (batch, event) or None.  Called BEFORE the consumer enqueues the step of the batch handed out last, so the
            construction chain (a dozen dependent small launches, ~0.2 ms of latency) starts next to the step still in flight and
            has a whole step of slack -- independent of how far the host runs ahead of the GPU."""
            fence = torch.cuda.Event()
            fence.record(main)                # everything the consumer has enqueued so far (steps <= j-2 when batch j is built)
            # ... of which steps <= j-3 = everything up to the PREVIOUS build are the last readers of whatever batch j's buffers may
            # reuse (see the module docstring); the first build of an iteration waits for everything enqueued so far (the side
            # stream and its pool outlive the iteration: the previous epoch's last steps may still be reading what they freed)
            side.wait_event(fence_prev[0] if fence_prev[0] is not None else fence)
            fence_prev[0] = fence
            with torch.cuda.stream(side):
                try:
                    batch = next(it)
                except StopIteration:
                    return None
                built = torch.cuda.Event()
                built.record(side)
            return batch, built

        pending = build()
        while pending is not None:
            batch, built = pending
            pending = build()                 # batch j+1 is under construction before step j is enqueued
            main.wait_event(built)
            yield batch
