"""Running meters for evaluation results -- same meter names and update rule as the reference's
utils/metrics.py (AccumulateMeter :4-42, MetricResult :46-89): count-weighted running means ordered
loss, h1, h3, h10, h50, mrr, mr."""
from __future__ import annotations

from collections import OrderedDict


class AccumulateMeter:
    def __init__(self, greater_is_better=True, print_precision=4):
        self.greater_is_better = greater_is_better
        self.print_precision = print_precision
        self.reset()

    def reset(self):
        self.avg, self.val, self.count = 0.0, 0.0, 0

    def update(self, val, n=1):
        if n <= 0:
            return
        self.val = val
        self.avg = (self.avg * self.count + val * n) / (self.count + n)
        self.count += n

    def __add__(self, other):
        if other.count > 0:
            self.update(other.avg, other.count)
        return self

    def avg_better_than(self, other):
        return self.avg > other.avg if self.greater_is_better else self.avg < other.avg

    def __repr__(self):
        return f"{self.avg:.{self.print_precision}f}"


class MetricResult(OrderedDict):
    NAMES = ("loss", "h1", "h3", "h10", "h50", "mrr", "mr")

    def __init__(self):
        super().__init__()
        self["loss"] = AccumulateMeter(greater_is_better=False, print_precision=7)
        for k in self.NAMES[1:]:
            self[k] = AccumulateMeter()

    @property
    def averages(self):
        return "  ".join(f"{k}: {v}" for k, v in self.items())

    @property
    def averages_dict(self):
        return {k: v.avg for k, v in self.items()}

    def __add__(self, other):
        if other is not None:
            for k in self:
                self[k] += other[k]
        return self

    def reset(self):
        for v in self.values():
            v.reset()
