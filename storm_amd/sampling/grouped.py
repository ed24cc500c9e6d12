"""Several micro-batches of one stream through ONE grouped score evaluation per sampler step.

BASELINE.json configs[4]: utterances of 2 - 10 s are micro-batched by padded frame count (the reference pads every file to its own
multiple of 64 frames, util/other.py:102-109, and processes one file per call, enhancement.py:66-72), which leaves 2 - 3 rows per
micro-batch - launches whose deep levels are a handful of pixel tiles.  Concurrent queues are closed on this platform
(profiles/r06_concurrent_repro.txt), so the micro-batches share LAUNCHES instead: every micro-batch runs its own unmodified sampler
(PC or ODE, any predictor / corrector, its own noise stream and step control) on a host thread of its own, and the threads meet
where they evaluate the score network - the ScoreBatcher collects the pending evaluations of all micro-batches that are still
running and issues them as one storm_ncsnpp_forward_group call (csrc/ncsnpp_graph.hip), on the one stream every thread launches on.
A micro-batch that finishes early (ODE: fewer accepted steps) simply stops taking part.

Nothing of a sampler changes, so a row's result is what its own micro-batch's call gives, up to the kernels the grouped evaluation
selects by the GROUP's tile count (16-bit: agreement to the rounding of the activations; fp32: bit for bit)."""
import threading

import torch

_tls = threading.local()


def current_batcher():
    return getattr(_tls, "batcher", None)


class ScoreBatcher:
    """Rendezvous of `n` participant threads around net.forward_parts_group."""

    def __init__(self, net, n):
        self.net, self.active = net, n
        self.cond = threading.Condition()
        self.pending = {}          # participant id -> (ins, t)
        self.results = {}          # participant id -> output tensor
        self.error = None
        self.calls = 0             # grouped evaluations issued
        self.rows = 0              # rows evaluated

    def _flush(self):
        """(holding the lock) every running participant is waiting: one grouped evaluation for all of them"""
        ids = sorted(self.pending)
        mine, _tls.batcher = getattr(_tls, "batcher", None), None       # (the evaluation itself must not come back here)
        try:
            outs = self.net.forward_parts_group([self.pending[i][0] for i in ids], [self.pending[i][1] for i in ids])
        except BaseException as e:  # noqa: BLE001 - every waiter must be released with the error
            self.error = e
            outs = [None] * len(ids)
        finally:
            _tls.batcher = mine
        self.calls += 1
        self.rows += sum(self.pending[i][0][0].shape[0] for i in ids)
        for i, o in zip(ids, outs):
            self.results[i] = o
        self.pending.clear()
        self.cond.notify_all()

    def submit(self, pid, ins, t):
        with self.cond:
            if self.error is not None:
                raise RuntimeError("grouped evaluation failed in another micro-batch") from self.error
            self.pending[pid] = (ins, t)
            if len(self.pending) >= self.active:
                self._flush()
            else:
                while pid not in self.results and self.error is None:
                    self.cond.wait()
            if self.error is not None:
                raise RuntimeError("grouped evaluation failed") from self.error
            return self.results.pop(pid)

    def leave(self):
        """a participant is done (or failed): the others no longer wait for it"""
        with self.cond:
            self.active -= 1
            if self.pending and len(self.pending) >= self.active:
                self._flush()

    def fail(self, e):
        with self.cond:
            if self.error is None:
                self.error = e
            self.cond.notify_all()


def run_grouped(net, fns, device=None, width=None):
    """Run the callables `fns` (one per micro-batch; each runs a whole sampler that evaluates `net`) on threads whose evaluations of
    `net` are grouped.  Returns (results in order, batcher).  With one callable, or on a net without grouped evaluation, runs them
    one after the other.
    width: at most this many micro-batches in flight (None = all of them).  A worker that finishes its micro-batch takes the next one
    from the list, so the group stays full until the list runs dry - a long stream holds the scratch of `width` problems instead of
    all of them, and samplers whose micro-batches need different numbers of evaluations (ODE) do not thin the group out before the
    list's end.  Every micro-batch still computes what its own call computes (module docstring), whatever it is grouped with."""
    if len(fns) < 2 or not hasattr(net, "forward_parts_group"):
        return [f() for f in fns], None
    n_workers = len(fns) if width is None else max(1, min(int(width), len(fns)))
    if n_workers < 2:
        return [f() for f in fns], None
    batcher = ScoreBatcher(net, n_workers)
    results, errors = [None] * len(fns), [None] * len(fns)
    grad = torch.is_grad_enabled()
    todo = iter(range(len(fns)))
    todo_lock = threading.Lock()

    def worker(w):
        _tls.batcher, _tls.pid = batcher, w
        k = None
        try:
            if device is not None and torch.device(device).type == "cuda":
                torch.cuda.set_device(device)
            with torch.set_grad_enabled(grad):
                while batcher.error is None:
                    with todo_lock:
                        k = next(todo, None)
                    if k is None:
                        break
                    results[k] = fns[k]()
        except BaseException as e:  # noqa: BLE001
            errors[k if k is not None else 0] = e
            batcher.fail(e)
        finally:
            _tls.batcher = None
            batcher.leave()

    threads = [threading.Thread(target=worker, args=(w,), daemon=True) for w in range(n_workers)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    for e in errors:
        if e is not None:
            raise e
    return results, batcher


def grouped_forward_parts(net, ins, time_cond):
    """called by NCSNpp.forward_parts: route the evaluation through the calling thread's batcher when it serves this net"""
    b = current_batcher()
    if b is None or b.net is not net:
        return None
    return b.submit(_tls.pid, ins, time_cond)
