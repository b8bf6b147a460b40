"""Batch construction on a side stream (the device half of the reference's loader → trainer hand-over, learning/main.py:186-192).

With `--loader_device 1 --batch_device 1` everything a batch needs on the GPU is produced by kernels and asynchronous copies
issued from the collate (device loader, `GraphConvInfo.set_batch_device`, clouds / labels H2D).  Issued on the training stream
they queue BEHIND the previous step and delay the next one; `SideStreamBatches` runs the loader iterator under a second HIP
stream instead, so this device work overlaps the training step that is still executing, and hands every batch over with one
event (the training stream waits for it; the host never does).

Memory discipline (no `record_stream` bookkeeping needed): tensors of batch j are allocated in the side stream's pool and are
dropped by the consumer at the earliest when it asks for batch j+1, i.e. after step j has been enqueued.  A later build may get
the same blocks again; before building batch j the side stream therefore waits for the event that was recorded on the training
stream when batch j-1 was requested (all of step j-2 and everything before it) -- the last possible reader of anything that can
have been freed -- while step j-1 is the one it overlaps with."""
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
        fence_prev = None                     # recorded on `main` at the previous request
        while True:
            fence = torch.cuda.Event()
            fence.record(main)                # everything the consumer has enqueued so far (steps <= j-1)
            # steps <= j-2 are complete before batch j's buffers may be (re)written; the first batch of an iteration waits for
            # everything enqueued so far (the side stream and its pool outlive the iteration: the previous epoch's last steps
            # may still be reading what they freed)
            side.wait_event(fence_prev if fence_prev is not None else fence)
            with torch.cuda.stream(side):
                try:
                    batch = next(it)
                except StopIteration:
                    return
                built = torch.cuda.Event()
                built.record(side)
            main.wait_event(built)
            fence_prev = fence
            yield batch
