"""Data-parallel side of the S2T training step: one process per GPU, RCCL over xGMI through
torch.distributed (backend "nccl" is RCCL on ROCm; "gloo" for the CPU tests).

Replaces on this path:
  fairseq/distributed_utils.py:79-129 (distributed_init + warm-up all-reduce), :254-302 (all_reduce_dict)
  fairseq/legacy_distributed_data_parallel.py:96-170 (flat-buffer gradient all-reduce AFTER backward, no overlap)
  fairseq/trainer.py:722-774 (_fast_stat_sync_sum, _check_grad_norms)

MI355X-first differences: gradients already live in one flat f32 arena, so a bucket is a slice, not a
copy; buckets are all-reduced asynchronously as the hand-written backward finishes layers (last layer
first), overlapping RCCL with the remaining backward GEMM / convolution kernels; bucket size defaults to
64 MiB because xGMI ring collectives are per-link bound (7 x ~153 GB/s point-to-point links, no switch):
few large transfers amortise the ~10 us launch+sync cost per collective better than DDP's 25 MiB.
"""
import os

import torch
import torch.distributed as dist


def is_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_initialized() else 1


def get_rank():
    return dist.get_rank() if is_initialized() else 0


def distributed_init(backend=None, device=None):
    """env:// rendezvous (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT, as torch.distributed.run sets them),
    then the 1-element warm-up all-reduce of distributed_utils.py:98-103."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0
    if not is_initialized():
        if backend is None:
            backend = "nccl" if (device is not None and torch.device(device).type == "cuda") else "gloo"
        dist.init_process_group(backend=backend, init_method="env://")
    t = torch.zeros(1, device=device if device is not None else "cpu")
    dist.all_reduce(t)
    return dist.get_rank()


def bucket_plan(n, bucket_elems):
    """Static bucket boundaries of a flat gradient buffer of `n` elements: contiguous [start, end) ranges cut from the TAIL
    (backward finishes the arena last-parameter-first), every one `bucket_elems` long except the head remainder.  A pure
    function of the arena size and the bucket size, so every rank issues the same collectives of the same sizes in the same
    order whatever its backward reported and whenever (the reference keeps its ranks aligned the same way: one fixed flat
    buffer walk, legacy_distributed_data_parallel.py:120-170)."""
    plan, end = [], n
    while end > 0:
        start = max(0, end - bucket_elems)
        plan.append((start, end))
        end = start
    return plan


class BucketedGradReducer:
    """Sum-all-reduce of the flat gradient buffer in STATIC contiguous buckets (bucket_plan), launched from backward.

    The engine reports finished parameter groups through notify(start, end) (element ranges of the arena, arriving in
    descending address order because the arena is laid out in forward order).  A bucket is all-reduced asynchronously as
    soon as everything at or above its start is final; finish() launches the buckets that are left, in plan order, and
    waits for all handles.  notify() only decides WHEN a bucket goes out, never its boundaries or its position in the
    sequence: a rank whose backward reported nothing (or ran on a dummy batch) still issues the identical collectives."""

    def __init__(self, flat_grad, bucket_bytes=64 << 20, group=None):
        self.flat = flat_grad
        self.n = flat_grad.numel()
        self.bucket_elems = max(1, bucket_bytes // flat_grad.element_size())
        self.plan = bucket_plan(self.n, self.bucket_elems)
        self.group = group
        self.record_events = False     # bench.py: a device event per bucket launch (where on the compute stream's timeline it left)
        # bench.py (one GPU): dict(stream, scratch, workgroups, ranks, bus_gbps) -> at every bucket launch a stand-in for the collective
        # (s2t_comm_standin: that many workgroups resident for the all-reduce's duration, moving its bytes through HBM) runs on a side
        # stream behind the bucket's gradients, and finish() waits for it as it would for the RCCL handles
        self.standin = None
        self.reset()

    def reset(self):
        self.next = 0              # index into the plan: buckets [0, next) have been handed to RCCL
        self.ready_low = self.n    # everything in [ready_low, n) is final
        self.handles = []
        self.launched = []
        self.early_elems = 0       # elements whose bucket left from inside backward (notify), not from finish()
        self.events = []           # record_events: (start, end, early, event recorded on the current stream at launch)
        self.finish_event = None

    def _launch_next(self, early=False):
        start, end = self.plan[self.next]
        self.next += 1
        self.launched.append((start, end))
        if early:
            self.early_elems += end - start
        if self.record_events:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.events.append((start, end, early, ev))
        if self.standin is not None:
            from . import lib as L
            sd = self.standin
            side = sd["stream"]
            side.wait_stream(torch.cuda.current_stream())
            src = self.flat[start:end].data_ptr()
            skip = (-src) % 16                                    # bucket bounds are element counts from the tail: 16-byte align the copy
            nbytes = (end - start) * self.flat.element_size() - skip
            nbytes = min(nbytes, sd["scratch"].numel() * sd["scratch"].element_size()) // 16 * 16
            L.check(L.load().s2t_comm_standin(src + skip, sd["scratch"].data_ptr(), nbytes, int(sd["workgroups"]),
                                              int(sd["ranks"]), float(sd["bus_gbps"]), side.cuda_stream), "s2t_comm_standin")
            self._standin_used = True
        if get_world_size() > 1:
            self.handles.append(dist.all_reduce(self.flat[start:end], group=self.group, async_op=True))

    def notify(self, start, end):
        """Gradients of arena elements [start, end) are final (no kernel of this update writes them again)."""
        if end >= self.ready_low and start < self.ready_low:
            self.ready_low = start
        while self.next < len(self.plan) and self.plan[self.next][0] >= self.ready_low:
            self._launch_next(early=True)

    def finish(self):
        if self.record_events:
            self.finish_event = torch.cuda.Event(enable_timing=True)
            self.finish_event.record()
        while self.next < len(self.plan):
            self._launch_next()
        self.ready_low = 0
        for h in self.handles:
            h.wait()
        self.handles = []
        if self.standin is not None and getattr(self, "_standin_used", False):
            torch.cuda.current_stream().wait_stream(self.standin["stream"])      # clip / Adam need the reduced gradients
            self._standin_used = False


def project_exposed_allreduce(launches, finish_ms, n_ranks, bus_gbps, elem_bytes=4):
    """PROJECTION, not a measurement: given when each bucket became launchable on one rank's timeline (launches = [(elements,
    ms since the first launch)], finish_ms = when backward ended on the same clock), how long after backward would the last bucket
    of an N-rank ring all-reduce finish if the buckets run back to back on one RCCL stream at `bus_gbps` GB/s of bus bandwidth
    (an all-reduce of S bytes takes 2 (N - 1) / N * S / bus)?  -> (exposed ms, total all-reduce ms)"""
    t_free, total = 0.0, 0.0
    for elems, t_ms in launches:
        dur = 2.0 * (n_ranks - 1) / n_ranks * elems * elem_bytes / (bus_gbps * 1e9) * 1e3
        t_free = max(t_free, t_ms) + dur
        total += dur
    return max(0.0, t_free - finish_ms), total


def all_reduce_stats(values, device=None):
    """Sum a dict of python / tensor scalars over the ranks with ONE f64 all-reduce
    (trainer.py:749-753 + distributed_utils.all_reduce_dict use one per origin device)."""
    keys = sorted(values)
    if get_world_size() == 1:
        return {k: float(values[k]) for k in keys}
    buf = torch.tensor([float(values[k]) for k in keys], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(buf)
    out = buf.tolist()
    return dict(zip(keys, out))


def all_reduce_tensor(t, group=None):
    """in-place sum of a device tensor over the ranks, stream-ordered, nothing read back (the update's sample size: trainer.train_step)"""
    if get_world_size() > 1:
        dist.all_reduce(t, group=group)
    return t


def check_grad_norms(grad_norm, device=None):
    """trainer.py:764-774: every rank must see the same gradient norm after the all-reduce."""
    world = get_world_size()
    if world == 1:
        return True
    buf = torch.zeros(world, dtype=torch.float64, device=device if device is not None else "cpu")
    buf[get_rank()] = float(grad_norm)
    dist.all_reduce(buf)
    ok = bool(((buf - buf[0]).abs() <= 1e-6 * buf[0].abs().clamp(min=1.0)).all())
    if not ok:
        raise FloatingPointError("Fatal error: gradients are inconsistent between workers: %s" % buf.tolist())
    return ok
