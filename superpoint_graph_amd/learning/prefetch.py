"""Batch construction on a side stream (the device half of the reference's loader → trainer hand-over, learning/main.py:186-192).

With `--loader_device 1 --batch_device 1` everything a batch needs on the GPU is produced by kernels and asynchronous copies
issued from the collate (device loader, `GraphConvInfo.set_batch_device`, clouds / labels H2D).  Issued on the training stream
they queue BEHIND the previous step and delay the next one; `SideStreamBatches` runs the loader iterator under a second HIP
stream instead, so this device work overlaps the training step that is still executing, and hands every batch over with one
event (the training stream waits for it; the host never does).

The construction runs ONE batch ahead: batch j+1 is built (host work + side-stream launches) right after batch j has been
handed over and before the consumer enqueues step j, so its launches have the whole of step j to finish.

Cross-stream dependencies are NOT free on this stack (an event record is a marker packet that drains the queue's pipeline, a
wait is a barrier packet: measured ~25 us each, tools/trainer_window_attrib.py), so the loop uses as few as it can:
* training stream <- side stream: the batch was built a whole step ago; `built.query()` on the host is normally already true
  and then nothing is enqueued at all (completion observed by the host orders everything enqueued afterwards);
* side stream <- training stream (memory safety: buffers of a batch live in the side stream's allocator pool; once the
  consumer drops them a later build may get the same blocks): this object KEEPS a reference to every batch it has handed out,
  so none of their blocks can be freed, and only every `fence_every` builds records ONE event on the training stream (all
  steps enqueued so far = the steps of every batch handed out), makes the side stream wait for it and only then lets go of
  those batches.  Whenever their blocks are freed afterwards, every later side-stream operation is already ordered behind
  their last reader.  Cost: `fence_every` + 2 batches stay allocated.
"""
from __future__ import annotations

import collections
import time

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

    def __init__(self, loader, stream: 'torch.cuda.Stream | None' = None, fence_every: int = 4, host_wait: bool = True):
        self.loader = loader
        self.side = stream
        self.fence_every = max(1, int(fence_every))
        # host_wait: a batch that is not built yet when its step is about to be enqueued is waited for on the HOST instead of with a
        # cross-stream dependency on the training stream.  Round 6: with the batch's small vectors in one staging copy the host
        # enqueues a fresh-batch step in 0.44 ms (0.50 before) -- far ahead of the GPU -- so `built.query()` was more often false and
        # the loop paid the ~25 us pipeline drain of a stream wait on MORE steps than the slower host had (window 1.255 -> 1.30 ms on
        # one box).  The build was enqueued a whole step earlier and runs next to the step in flight, so the host wait is short, and
        # the training stream never sees a barrier packet.
        self.host_wait = bool(host_wait)
        self.host_wait_seconds = 0.0          # time the host spent WAITING for a build (not working): bench.py takes it off its enqueue figure

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        if not torch.cuda.is_available():
            raise RuntimeError('SideStreamBatches needs a GPU: the superpoint_graph_amd batch construction has no CPU path')
        main = torch.cuda.current_stream()
        side = self.side if self.side is not None else side_stream()
        it = iter(self.loader)
        held = collections.deque()            # batches built and not yet released by this object
        nbuilt = [0]
        # the side stream and its pool outlive the iteration: the previous epoch's last steps may still be reading what they freed
        start = torch.cuda.Event()
        start.record(main)
        side.wait_event(start)

        def build():
            """-> (batch, event) or None.  Called BEFORE the consumer enqueues the step of the batch handed out last, so the
            construction (launches + copies, ~0.2 ms of latency) starts next to the step still in flight and has a whole step of
            slack -- independent of how far the host runs ahead of the GPU."""
            if nbuilt[0] and nbuilt[0] % self.fence_every == 0:
                # main has enqueued the steps of every batch handed out so far = everything in `held` but the newest (built, not
                # yet handed out): order the side stream behind them, THEN let go of those batches
                fence = torch.cuda.Event()
                fence.record(main)
                side.wait_event(fence)
                while len(held) > 1:
                    held.popleft()
            with torch.cuda.stream(side):
                try:
                    batch = next(it)
                except StopIteration:
                    return None
                built = torch.cuda.Event()
                built.record(side)
            held.append(batch)
            nbuilt[0] += 1
            return batch, built

        pending = build()
        while pending is not None:
            batch, built = pending
            pending = build()                 # batch j+1 is under construction before step j is enqueued
            if not built.query():             # built a whole step ago: normally complete already -- then the training stream needs
                if self.host_wait:            # no cross-stream dependency at all (one costs ~25 us of pipeline drain)
                    t_w = time.perf_counter()
                    built.synchronize()
                    self.host_wait_seconds += time.perf_counter() - t_w
                else:
                    main.wait_event(built)
            yield batch
