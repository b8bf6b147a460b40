"""The two torchnet meters the reference's training loop uses (learning/main.py:183-184: `tnt.meter.AverageValueMeter`,
`tnt.meter.ClassErrorMeter(accuracy=True)`), with the same arithmetic, so logged numbers agree; no torchnet dependency.
Values may be added as 0-d CUDA tensors: they are fetched lazily (one transfer when the value is read), so the training
loop does not synchronise with the GPU every iteration."""
import math

import numpy as np
import torch


class AverageValueMeter:
    def __init__(self):
        self.reset()

    def reset(self):
        self._pending, self.n = [], 0
        self.sum, self.var, self.mean, self.std = 0.0, 0.0, np.nan, np.nan
        self.mean_old, self.m_s = 0.0, 0.0

    def add(self, value, n=1):
        self._pending.append((value, n))

    def _flush(self):
        if not self._pending:
            return
        dev = [v for v, _ in self._pending if torch.is_tensor(v)]
        fetched = iter(torch.stack([d.detach().float().reshape(()) for d in dev]).double().cpu().tolist()) if dev else iter(())
        for v, n in self._pending:
            value = next(fetched) if torch.is_tensor(v) else float(v)
            self.sum += value * n
            self.n += n
            if self.n == 1:                      # torchnet's running mean / variance (Welford)
                self.mean, self.std = 0.0 + value, np.inf
                self.mean_old, self.m_s = self.mean, 0.0
            else:
                self.mean = self.mean_old + (value - n * self.mean_old) / float(self.n)
                self.m_s += (value - self.mean_old) * (value - self.mean)
                self.mean_old = self.mean
                self.std = math.sqrt(self.m_s / (self.n - 1.0))
        self._pending = []

    def value(self):
        self._flush()
        return self.mean, self.std


class ClassErrorMeter:
    """top-1 accuracy (accuracy=True) or error in percent; `add(output [N, C], target [N])` like torchnet, or
    `add_counts(correct, counted)` when the counting happened on the device (spg_eval_accumulate)."""

    def __init__(self, topk=[1], accuracy=False):
        assert list(topk) == [1], 'only top-1 is used by the training loop'
        self.accuracy = accuracy
        self.reset()

    def reset(self):
        self.correct, self.n = 0, 0

    def add(self, output, target):
        output = output.cpu().numpy() if torch.is_tensor(output) else np.asarray(output)
        target = target.cpu().numpy() if torch.is_tensor(target) else np.asarray(target)
        if output.ndim == 1:
            output = output[None]
        pred = np.argmax(output, 1)
        self.correct += int((pred == target.reshape(-1)).sum())
        self.n += int(output.shape[0])

    def add_counts(self, correct, counted):
        self.correct += int(correct)
        self.n += int(counted)

    def value(self, k=-1):
        wrong = self.n - self.correct              # torchnet keeps the number of errors
        v = (1.0 - float(wrong) / self.n) * 100.0 if self.accuracy else float(wrong) / self.n * 100.0
        return v if k != -1 else [v]
