"""Concatenation of the per-degradation training sets (reference basicsr/data/concat_dataset.py:41-93): dataset ``k`` is
repeated ``enlarge_ratios[k]`` times, and every sample carries ``dataset_idx = k`` -- the label of the degradation-classification
loss (DCPTModel.feed_data)."""
from __future__ import annotations

from bisect import bisect_right

from torch.utils.data import Dataset


class ConcatDataset(Dataset):
    def __init__(self, datasets, enlarge_ratios):
        super().__init__()
        self.datasets = list(datasets)
        self.enlarge_ratios = list(enlarge_ratios)
        assert len(self.datasets) > 0, "datasets should not be an empty iterable"
        assert len(self.datasets) == len(self.enlarge_ratios), "one enlarge ratio per dataset"
        self.datasets_length = [len(d) for d in self.datasets]
        self.cumulative_sizes, total = [], 0
        for n, ratio in zip(self.datasets_length, self.enlarge_ratios):
            total += n * ratio
            self.cumulative_sizes.append(total)

    def __len__(self):
        return self.cumulative_sizes[-1]

    def __getitem__(self, idx):
        if idx < 0:
            if -idx > len(self):
                raise ValueError("absolute value of index should not exceed dataset length")
            idx += len(self)
        k = bisect_right(self.cumulative_sizes, idx)
        offset = idx - (self.cumulative_sizes[k - 1] if k > 0 else 0)
        data = self.datasets[k][offset % self.datasets_length[k]]
        data["dataset_idx"] = k
        return data
