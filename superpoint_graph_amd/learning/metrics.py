"""Evaluation metrics with the reference's interface (learning/metrics.py:8-89 `ConfusionMatrix`), accumulated on the
device: `count_predicted_batch_device` is the GPU form of `count_predicted_batch(tvec, np.argmax(o, 1))` as the
evaluation loops call it (learning/main.py:246-263, eval_final :267-311), including the mean over the test-time samples
and the removal of superpoints without ground truth (`filter_valid`, :447-452).  Counts are exact integers, so every
derived score equals the reference's float64 arithmetic bit for bit."""
import numpy as np
import torch


class ConfusionMatrix:
    def __init__(self, number_of_labels=2, device=None):
        self.number_of_labels = number_of_labels
        self._dev = None if device is None else torch.zeros(number_of_labels, number_of_labels, dtype=torch.int64, device=device)
        self._counters = None if device is None else torch.zeros(2, dtype=torch.int64, device=device)
        self._host = np.zeros(shape=(number_of_labels, number_of_labels))

    # ---- device accumulation ----
    def count_predicted_batch_device(self, label_vec, logits, label_mode):
        """label_vec i64 [N, C] (points per class), logits f32 [N, C] or [S, N, C] (S test-time samples), label_mode i64 [N]
        (-100 = no ground truth), all on the GPU.  Returns the predictions i64 [N] (argmax of the mean logits)."""
        from .. import ops
        if self._dev is None:
            self._dev = torch.zeros(self.number_of_labels, self.number_of_labels, dtype=torch.int64, device=logits.device)
            self._counters = torch.zeros(2, dtype=torch.int64, device=logits.device)
        return ops.eval_accumulate(logits, label_mode, label_vec, self._dev, self._counters)

    def allreduce(self, group=None):
        """Data-parallel evaluation: every rank counted its own scenes; the counts are exact integers, so the sum over the ranks
        equals the single-process matrix bit for bit (reference learning/metrics.py:16-18 adds the same counts sequentially)."""
        import torch.distributed as dist
        if not (dist.is_initialized() and dist.get_world_size(group) > 1):
            return
        if self._dev is None:        # a rank without a single evaluated batch still takes part in the collective
            dev = torch.device('cuda', torch.cuda.current_device())
            self._dev = torch.zeros(self.number_of_labels, self.number_of_labels, dtype=torch.int64, device=dev)
            self._counters = torch.zeros(2, dtype=torch.int64, device=dev)
        buf = torch.cat([self._dev.reshape(-1), self._counters, torch.from_numpy(self._host.astype(np.int64).reshape(-1)).to(self._dev.device)])
        if dist.get_backend(group) == 'gloo':
            host = buf.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            buf = host.to(self._dev.device)
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        n = self.number_of_labels * self.number_of_labels
        self._dev.copy_(buf[:n].view_as(self._dev))
        self._counters.copy_(buf[n:n + 2])
        self._host = buf[n + 2:].cpu().numpy().astype(np.float64).reshape(self._host.shape)

    @property
    def confusion_matrix(self):
        """float64 [C, C] like the reference's attribute (host counts + device counts)."""
        if self._dev is None:
            return self._host
        return self._host + self._dev.cpu().numpy().astype(np.float64)

    def accuracy_counts(self):
        """(correct, counted) of the top-1 accuracy meter (tnt ClassErrorMeter(accuracy=True) = 100 * correct / counted)."""
        if self._counters is None:
            return 0, 0
        c = self._counters.cpu().numpy()
        return int(c[0]), int(c[1])

    # ---- the reference's host interface ----
    def count_predicted(self, ground_truth, predicted, number_of_added_elements=1):
        self._host[ground_truth][predicted] += number_of_added_elements

    def count_predicted_batch(self, ground_truth_vec, predicted):
        np.add.at(self._host.T, np.asarray(predicted), np.asarray(ground_truth_vec, dtype=np.float64))

    def count_predicted_batch_hard(self, ground_truth_vec, predicted):
        np.add.at(self._host, (np.asarray(ground_truth_vec), np.asarray(predicted)), 1)

    def get_count(self, ground_truth, predicted):
        return self.confusion_matrix[ground_truth][predicted]

    def get_confusion_matrix(self):
        return self.confusion_matrix

    def get_intersection_union_per_class(self):
        m = self.confusion_matrix
        diag = np.diag(m)
        divisor = m.sum(1) + m.sum(0) - diag          # diagonal + row errors + column errors
        divisor = np.where(diag == 0, 1.0, divisor)
        return [float(d) / v for d, v in zip(diag, divisor)]

    def get_overall_accuracy(self):
        m = self.confusion_matrix
        total = m.sum()
        return float(np.trace(m)) / (total if total != 0 else 1)

    def get_average_intersection_union(self):
        m = self.confusion_matrix
        class_seen = ((m.sum(1) + m.sum(0)) != 0).sum()
        return sum(self.get_intersection_union_per_class()) / class_seen

    def get_mean_class_accuracy(self):
        m = self.confusion_matrix
        return float(sum(m[i][i] / max(1, np.sum(m[i, :])) for i in range(self.number_of_labels))) / self.number_of_labels

    def count_gt(self, ground_truth):
        return self.confusion_matrix[ground_truth, :].sum()
