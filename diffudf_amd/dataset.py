# coding: utf-8
"""Batch source for the training loop, with the reference's yield contract (reference
src/dataset.py:134-185): an iterable whose `__iter__` yields `batchesPerEpoch` tuples

    coords (1,N,3), normals (1,N,3) [zeros off-surface], sdf (1,N,1) [zeros on-surface]     fp32, CPU

with N = int(B*p0) + 2*(int(B*p1)//2) points ordered [on-surface | far | near] (reference :162-163, :27-28).

This is also the SHARDING SEAM of the multi-GPU path: `rank`/`world` select this rank's equal slice of each
of the three strata, so every rank keeps the on/far/near mix and the union over ranks is the global batch.

`SyntheticPointCloud` draws the batch from `diffudf_amd.synth` (uniform coordinates, the same three strata).
The mesh-backed sampler (OBJ + point-to-triangle distance on the GPU) is the next row of the scope table
(SURVEY.md §8(f) rank 1); the reference's own one needs open3d, which this image does not have.
"""
import numpy as np
import torch

from . import synth


def batch_size_of(batchSize, samplingPercentiles):
    n_on = int(batchSize * samplingPercentiles[0])
    n_off = int(batchSize * samplingPercentiles[1])
    return n_on + 2 * (n_off // 2)


class SyntheticPointCloud:
    def __init__(self, batchSize=30000, samplingPercentiles=(0.333, 0.666), batchesPerEpoch=1, seed=123,
                 rank=0, world=1):
        self.batchSize = batchSize
        self.samplingPercentiles = list(samplingPercentiles)
        self.batchesPerEpoch = batchesPerEpoch
        self.seed, self.rank, self.world = seed, rank, world
        self.n_global = batch_size_of(batchSize, samplingPercentiles)
        self._step = 0

    def local_indices(self):
        return synth.stratified_shard(self.n_global, self.rank, self.world)

    def __iter__(self):
        for _ in range(self.batchesPerEpoch):
            idx = self.local_indices()
            cuts = np.flatnonzero(np.diff(idx) != 1) + 1
            xs, ns, ss = [], [], []
            for part in np.split(idx, cuts):
                x, n, s = synth.training_batch(self.n_global, seed=self.seed, step=self._step, lo=int(part[0]),
                                               hi=int(part[-1]) + 1)
                xs.append(x); ns.append(n); ss.append(s)
            self._step += 1
            yield (torch.from_numpy(np.concatenate(xs))[None], torch.from_numpy(np.concatenate(ns))[None],
                   torch.from_numpy(np.concatenate(ss))[None])


class PointCloud:
    """Reference constructor signature (reference src/dataset.py:135-155).  Mesh-backed sampling is not built
    yet: constructing it says so instead of failing later on a missing open3d."""

    def __init__(self, meshPath, batchSize, samplingPercentiles, batchesPerEpoch, device=None, onlyPCloud=False):
        raise NotImplementedError(
            "PointCloud(mesh): the GPU surface sampler / point-to-triangle distance kernel is the next scope row "
            "(SURVEY.md §8(f) rank 1); use SyntheticPointCloud or set \"dataset\": \"synthetic\" in the config")
