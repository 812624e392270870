"""Rank-strided sampler over an ``ratio``-times enlarged index space (reference basicsr/data/data_sampler.py:7-52): one
process per GPU draws ``ceil(len * ratio / world)`` indices per epoch from a permutation seeded by the epoch number."""
from __future__ import annotations

import math

import torch
from torch.utils.data.sampler import Sampler


class EnlargedSampler(Sampler):
    def __init__(self, dataset, num_replicas, rank, ratio=1):
        self.dataset, self.num_replicas, self.rank = dataset, num_replicas, rank
        self.epoch = 0
        self.num_samples = math.ceil(len(dataset) * ratio / num_replicas)
        self.total_size = self.num_samples * num_replicas

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.epoch)
        n = len(self.dataset)
        mine = torch.randperm(self.total_size, generator=g)[self.rank:self.total_size:self.num_replicas]
        return iter((mine % n).tolist())

    def __len__(self):
        return self.num_samples

    def set_epoch(self, epoch):
        self.epoch = epoch
