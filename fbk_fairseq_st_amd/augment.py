"""Input augmentation on the device: TimeStretch and SpecAugment of the reference
(examples/speech_recognition/modules/time_stretch.py, modules/specaugment.py), applied by the task's train_step in that order
(tasks/speech_recognition.py:254-258).

The reference draws its random numbers from Python's `random` and numpy's global generator and applies each mask with a slice write per
utterance.  Here the host draws the SAME numbers in the SAME order into small integer tables (so a run seeded like the reference's
produces the identical batch) and one kernel launch (`s2t_augment`) applies row gather + masks to the whole batch.
"""
import random

import numpy as np
import torch

from . import lib as L


def _spec_tables(B, tau, v, F_par, T_par, nF, nT, rate):
    """mask tables of SpecAugment.forward + specaugment() (specaugment.py:55-112); (f0, width) / (t0, width), width 0 = no mask"""
    fm = np.zeros((B, max(nF, 1), 2), np.int32); tm = np.zeros((B, max(nT, 1), 2), np.int32)
    for b in range(B):
        if random.random() < rate:
            for i in range(nF):
                f = int(np.random.uniform(low=0.0, high=F_par))
                fm[b, i] = (random.randint(0, v - f), f)
            for i in range(nT):
                t = int(np.random.uniform(low=1.0, high=min(T_par, tau)))
                tm[b, i] = (random.randint(0, tau - t), t)
    return fm, tm


def _stretch_ids(time_len, w, low, high):
    """time_stretch_seq (time_stretch.py:42-57): the source row of every output row"""
    ids = []
    if time_len < 10 and low < 1.0:
        low = 1.0
    for i in range(int(round(time_len / w))):
        s = random.uniform(low, high) * min(w, time_len - w * i)
        e = min(time_len, w * (i + 1))
        r = torch.round(torch.linspace(w * i, e - 1, int(s))).long()      # torch's own linspace/round: bit-identical indices
        ids.append(r)
    return torch.cat(ids) if ids else torch.zeros(0, dtype=torch.long)


def _apply(x, row_map, fm, tm, To):
    """x [B,T,F] f32 on the GPU -> [B,To,F]"""
    L.require_cuda(x)
    B, T, F = x.shape
    x = x.contiguous()
    out = torch.empty((B, To, F), dtype=torch.float32, device=x.device)
    dev = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(x.device)
    rm, f, t = dev(row_map), dev(fm), dev(tm)
    nF = 0 if fm is None else fm.shape[1]
    nT = 0 if tm is None else tm.shape[1]
    L.check(L.load().s2t_augment(L.ptr(x), L.ptr(out), L.ptr(rm), L.ptr(f), L.ptr(t), B, T, To, F, nF, nT, L.stream()), "s2t_augment")
    return out


class SpecAugment:
    def __init__(self, frequency_masking_pars, time_masking_pars, frequency_masking_num, time_masking_num, rate=1.0):
        self.F, self.T, self.nF, self.nT, self.rate = frequency_masking_pars, time_masking_pars, frequency_masking_num, time_masking_num, rate

    def tables(self, B, tau, v):
        return _spec_tables(B, tau, v, self.F, self.T, self.nF, self.nT, self.rate)

    def __call__(self, batch):
        x = batch["net_input"]["src_tokens"]
        B, tau, v = x.shape
        fm, tm = self.tables(B, tau, v)
        batch["net_input"]["src_tokens"] = _apply(x, None, fm if self.nF else None, tm if self.nT else None, tau)
        return batch


class TimeStretch:
    def __init__(self, rate, w, low, high):
        if w < 1:
            raise ValueError("w must be greater than 1")
        self.rate, self.w, self.low, self.high = rate, w, low, high

    def row_map(self, lengths):
        ids = []
        for length in lengths:
            length = int(length)
            ids.append(_stretch_ids(length, self.w, self.low, self.high) if random.random() < self.rate else torch.arange(length))
        new_len = [int(i.numel()) for i in ids]
        rm = np.full((len(ids), max(new_len)), -1, np.int32)
        for b, i in enumerate(ids):
            rm[b, :i.numel()] = i.numpy()
        return rm, new_len

    def __call__(self, batch):
        ni = batch["net_input"]
        rm, new_len = self.row_map(ni["src_lengths"].tolist())
        out = dict(batch)
        out["net_input"] = dict(ni)
        out["net_input"]["src_tokens"] = _apply(ni["src_tokens"], rm, None, None, rm.shape[1])
        out["net_input"]["src_lengths"] = torch.tensor(new_len, dtype=torch.long, device=ni["src_lengths"].device)
        if "nframes" in out:
            out["nframes"] = int(sum(new_len))
        return out
