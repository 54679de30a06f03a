"""Order of operations of the data-parallel wrapper's gradient reduction (dist_shard.DistributedFHVAE) with fakes -- no GPU, no
process group: the combination the N > 1 default takes (persistent recurrences + deferred weight gradients + the early
all-reduce) cannot run on a one-GPU box, so its control flow is pinned here (VERDICT r03 #4d)."""
import torch

import dist_shard
import hip_binding as hb


class _Handle:
    def __init__(self, log):
        self.log = log

    def wait(self):
        self.log.append("wait")


class _Sh:
    world = 8

    def __init__(self, log, sync=False):
        self.log, self.sync = log, sync

    def all_reduce_(self, t, op=None, async_op=False):
        self.log.append(("all_reduce", t.data_ptr(), t.numel(), bool(async_op)))
        return None if self.sync else (_Handle(self.log) if async_op else None)


class _Opt:
    def __init__(self, log, flat):
        self.log = log
        self.g_arena = type("A", (), {"flat": flat})()

    def flat_grad(self):
        self.log.append("flat_grad")
        return self.g_arena.flat


def _runner(log, sync=False):
    r = object.__new__(dist_shard.DistributedFHVAE)
    flat = torch.zeros(30)
    r.sh, r.opt_nets = _Sh(log, sync), _Opt(log, flat)
    r._buckets = [(0, 10), (10, 20), (20, 30)]     # decoder, z1 encoder, z2 encoder: the order their backward completes
    r._views = [flat[0:10], flat[10:20], flat[20:30]]
    r._bucket_of_ptr = {v.data_ptr(): g for g, v in enumerate(r._views)}
    r._pending, r.overlap = None, True
    return r, flat


def test_early_all_reduce_order(monkeypatch):
    log = []
    monkeypatch.setattr(hb, "flush_param_grads", lambda: log.append("flush"))
    monkeypatch.setattr(hb, "flush_param_grads_except_last", lambda: log.append("flush_except_last"))
    r, flat = _runner(log)
    r._on_lstm_rec_done([r._views[0]])       # decoder's recurrence enqueued: nothing yet
    assert log == []
    r._on_lstm_rec_done([r._views[1]])       # z1 encoder's: the queued weight gradients (decoder + z1) go out as one launch
    assert log == ["flush"]
    r._on_lstm_rec_done([r._views[2]])       # the LAST recurrence is enqueued: buckets 0-1 are complete -> ONE async all-reduce
    assert log[1:] == ["flush_except_last", ("all_reduce", flat.data_ptr(), 20, True)]
    r._on_lstm_rec_done([r._views[2]])       # fired again (a repeated backward): no second early collective
    assert len(log) == 3
    r._reduce_gradients()                    # after backward: flush the last net's weight gradients, reduce ITS bucket, then wait
    assert log[3:] == ["flat_grad", ("all_reduce", flat[20:].data_ptr(), 10, False), "wait"]
    assert r._pending is None


def test_synchronous_transport_and_no_overlap(monkeypatch):
    log = []
    monkeypatch.setattr(hb, "flush_param_grads", lambda: log.append("flush"))
    monkeypatch.setattr(hb, "flush_param_grads_except_last", lambda: log.append("flush_except_last"))
    r, flat = _runner(log, sync=True)        # staged gloo: the "async" call returns no handle
    for g in (0, 1, 2):
        r._on_lstm_rec_done([r._views[g]])
    assert r._pending is False
    r._reduce_gradients()
    assert log[-2:] == ["flat_grad", ("all_reduce", flat[20:].data_ptr(), 10, False)] and r._pending is None
    log.clear()
    r.overlap = False                        # one rank / FC model: one collective over all buckets after the backward
    for g in (0, 1, 2):
        r._on_lstm_rec_done([r._views[g]])
    r._reduce_gradients()
    assert log == ["flat_grad", ("all_reduce", flat.data_ptr(), 30, False)]


def test_stale_pending_handle_is_dropped_before_the_next_step(monkeypatch):
    log = []
    monkeypatch.setattr(hb, "flush_param_grads", lambda: log.append("flush"))
    monkeypatch.setattr(hb, "flush_param_grads_except_last", lambda: log.append("flush_except_last"))
    r, flat = _runner(log)
    for g in (0, 1, 2):
        r._on_lstm_rec_done([r._views[g]])
    assert r._pending is not None            # ... and the backward raised here: _reduce_gradients never ran
    r._drop_stale_pending()                  # what train_step does first
    assert log[-1] == "wait" and r._pending is None
    log.clear()
    for g in (0, 1, 2):                      # the next step issues its own early all-reduce again
        r._on_lstm_rec_done([r._views[g]])
    assert ("all_reduce", flat.data_ptr(), 20, True) in log
