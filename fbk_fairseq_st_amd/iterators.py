"""Batch iterator of the S2T data path.

Same batches, in the same order, as the reference's `FairseqTask.get_batch_iterator` + `EpochBatchIterator`
(fairseq/tasks/fairseq_task.py:107-199, fairseq/data/iterators.py:170-340): indices in dataset order, utterances beyond
`max_positions` dropped, frame-budget batches (native batcher), batches shuffled with numpy seeded by `seed + epoch`
(identical on every rank), dealt round-robin to the shards and padded with empty batches, collated per batch.

MI355X-first: collation runs in a background thread that pins the host buffers, so the GPU never waits for it and the H2D
copy of batch t+1 can overlap the kernels of batch t (`Trainer.prepare(..., non_blocking=True)`).
"""
import math
import queue
import threading

import numpy as np
import torch

from .indexed import batch_by_size


def filter_by_size(indices, dataset, max_positions, raise_exception=False):
    """fairseq/data/data_utils.py:163-197 (component-wise comparison of dataset.size(idx) with max_positions)"""
    def ok(idx):
        sz = dataset.size(int(idx))
        if isinstance(max_positions, (int, float)):
            return (max(sz) if isinstance(sz, tuple) else sz) <= max_positions
        if not isinstance(sz, tuple):
            sz = (sz,)
        return all(a is None or b is None or a <= b for a, b in zip(sz, max_positions))
    keep = np.array([ok(i) for i in indices], dtype=bool)
    ignored = indices[~keep]
    if len(ignored) > 0 and raise_exception:
        raise Exception("Size of sample #{} is invalid (={}) since max_positions={}, skip this example with "
                        "--skip-invalid-size-inputs-valid-test".format(ignored[0], dataset.size(int(ignored[0])), max_positions))
    return indices[keep]


def _pin(x):
    if torch.is_tensor(x):
        return x.pin_memory() if torch.cuda.is_available() else x
    if isinstance(x, dict):
        return {k: _pin(v) for k, v in x.items()}
    return x


class EpochBatchIterator:
    def __init__(self, dataset, collate_fn, batch_sampler, seed=1, num_shards=1, shard_id=0, epoch=1, prefetch=2, pin_memory=True):
        self.dataset, self.collate_fn = dataset, collate_fn
        self.frozen_batches = tuple(tuple(b) for b in batch_sampler)
        self.seed, self.num_shards, self.shard_id = seed, num_shards, shard_id
        self.epoch = max(epoch, 1) - 1                   # incremented by next_epoch_itr, as in the reference
        self.prefetch, self.pin_memory = prefetch, pin_memory
        self._count = 0
        self._len = int(math.ceil(len(self.frozen_batches) / float(num_shards)))

    def __len__(self):
        return self._len

    def batches_for_epoch(self, epoch, shuffle=True):
        batches = list(self.frozen_batches)
        if shuffle:
            state = np.random.get_state()                # data_utils.numpy_seed: seed, shuffle, restore
            np.random.seed(self.seed + epoch)
            np.random.shuffle(batches)
            np.random.set_state(state)
        mine = batches[self.shard_id::self.num_shards]
        return mine + [()] * (self._len - len(mine))     # ShardedIterator(fill_value=[])

    def next_epoch_itr(self, shuffle=True):
        self.epoch += 1
        self._count = 0
        return self._iterate(self.batches_for_epoch(self.epoch, shuffle))

    def end_of_epoch(self):
        return self._count >= self._len

    def _make(self, batch):
        out = self.collate_fn([self.dataset[i] for i in batch]) if len(batch) else {}
        return _pin(out) if self.pin_memory else out

    def _iterate(self, batches):
        if self.prefetch <= 0:
            for b in batches:
                self._count += 1
                yield self._make(b)
            return
        q = queue.Queue(maxsize=self.prefetch)

        def work():
            try:
                for b in batches:
                    q.put(self._make(b))
                q.put(StopIteration)
            except BaseException as e:                   # surface loader errors in the consumer
                q.put(e)

        threading.Thread(target=work, daemon=True).start()
        while True:
            item = q.get()
            if item is StopIteration:
                return
            if isinstance(item, BaseException):
                raise item
            self._count += 1
            yield item


def get_batch_iterator(dataset, max_tokens=None, max_sentences=None, max_positions=None, ignore_invalid_inputs=False,
                       required_batch_size_multiple=1, seed=1, num_shards=1, shard_id=0, epoch=1, prefetch=2, pin_memory=True,
                       bucket_by_length=False):
    """bucket_by_length (build-defined, off by default: the reference batches filterbanks in dataset order, fbank_dataset.py:78-81):
    utterances are ordered by frame count (stable) before the frame-budget batcher runs, so a batch holds utterances of similar
    length and the zero padding the kernels would chew through shrinks; batches are still shuffled
    per epoch."""
    indices = dataset.ordered_indices()
    if max_positions is not None:
        indices = filter_by_size(indices, dataset, max_positions, raise_exception=not ignore_invalid_inputs)
    lens = getattr(dataset, "frame_lengths", None)
    if bucket_by_length:
        key = np.asarray(lens)[indices] if lens is not None else np.array([dataset.num_tokens(int(i)) for i in indices])
        indices = indices[np.argsort(key, kind="mergesort")]
    sampler = batch_by_size(indices, lens if lens is not None else dataset.num_tokens, max_tokens=max_tokens,
                            max_sentences=max_sentences, required_batch_size_multiple=required_batch_size_multiple)
    return EpochBatchIterator(dataset, dataset.collater, sampler, seed=seed, num_shards=num_shards, shard_id=shard_id, epoch=epoch,
                              prefetch=prefetch, pin_memory=pin_memory)
