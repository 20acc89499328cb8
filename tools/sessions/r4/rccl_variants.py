#!/usr/bin/env python
"""bench.py with the reducer's exchange patched for an A/B of where the piecewise step's +3 ms come from (1 GPU, 1-rank RCCL group):
VARIANT=events_only : ready / done events and cross-stream waits as usual, no dist.all_reduce call
VARIANT=per_bucket  : one event pair per bucket (the round's first implementation) instead of one per segment
VARIANT=same_stream : the all-reduces issued on the compute stream, no side stream
VARIANT=default     : nothing patched"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch
from flamingo_mini_amd import data_parallel as dp

v = os.environ.get("VARIANT", "default")
R = dp.GradientAllReducer
if v == "events_only":
    R._mean_in_place = lambda self, t: None
elif v == "per_bucket":
    def per_bucket(self, buckets):
        if self.active:
            for flat, owners in buckets:
                self._early.update(id(p) for p, _, _ in owners)
                self._reduce_async(flat, list(owners))
    R.reduce_buckets = per_bucket
elif v == "same_stream":
    def same_stream(self, buckets):
        if self.active:
            for flat, owners in buckets:
                self._early.update(id(p) for p, _, _ in owners)
                self._mean_in_place(flat)
                self.pending.append((flat, dp._StreamWork(None), list(owners)))
    R.reduce_buckets = same_stream
elif v == "ready_only":          # an event record on the compute stream after every segment, nothing else
    def ready_only(self, buckets):
        e = torch.cuda.Event(); e.record()
    R.reduce_buckets = ready_only
elif v == "ready_wait":          # + the side stream waits for it
    def ready_wait(self, buckets):
        e = torch.cuda.Event(); e.record()
        self.stream.wait_event(e)
    R.reduce_buckets = ready_wait
elif v == "ready_wait_done":     # + a done event on the side stream, finish() waits for every one of them; no all_reduce
    R._mean_in_place = lambda self, t: None
elif v == "last_done":           # all of it incl. the all_reduce calls, but finish() waits for the LAST done event only
    orig = R.finish
    def finish(self):
        works = [w for _, w, _ in self.pending if w.event is not None]
        for w in works[:-1]:
            w.event = None
        orig(self)
    R.finish = finish
elif v in ("lag1", "every3", "lag1_wait_only"):
    from flamingo_mini_amd import graphs as G
    def call(self, batch=None):
        if self.optimizer is not None and hasattr(self.optimizer, "sync_device_hyperparams"):
            self.optimizer.sync_device_hyperparams()
        self.graphs[0].replay()
        red, todo, acc = self.reducer, None, []
        for i, (g, buckets) in enumerate(zip(self.graphs[1:], self.segment_buckets)):
            g.replay()
            if v == "every3":
                acc += buckets
                if (i % 3 == 2 or i == len(self.segment_buckets) - 1) and acc:
                    red.reduce_buckets(acc); acc = []
                continue
            e = torch.cuda.Event(); e.record()
            if todo is not None:
                issue(red, *todo)
            todo = (e, buckets)
        if todo is not None:
            issue(red, *todo)
        red.finish()
        if self._opt_graph is not None:
            self._opt_graph.replay()
        return self.loss
    def issue(red, e, buckets):        # the side stream waits for an event recorded BEFORE the graph launch that was issued in between
        if not buckets:
            return
        for _, owners in buckets:
            red._early.update(id(p) for p, _, _ in owners)
        with torch.cuda.stream(red.stream):
            red.stream.wait_event(e)
            if v == "lag1":
                for flat, _ in buckets:
                    red._mean_in_place(flat)
                done = torch.cuda.Event(); done.record()
                for i, (flat, owners) in enumerate(buckets):
                    red.pending.append((flat, dp._StreamWork(done if i == 0 else None), list(owners)))
    G.PiecewiseGraphedTrainStep.__call__ = call
if v in ("host_paced", "host_paced_all"):
    # every sub-graph is enqueued first; then the HOST waits for segment k's event and only then issues its collectives - no barrier packet
    # sits in the side queue while the compute queue dispatches thousands of small kernels.  host_paced: the last segment stays stream-ordered
    # (its collectives + finish + the optimizer graph are enqueued without waiting for the GPU)
    from flamingo_mini_amd import graphs as G
    def call(self, batch=None):
        if self.optimizer is not None and hasattr(self.optimizer, "sync_device_hyperparams"):
            self.optimizer.sync_device_hyperparams()
        self.graphs[0].replay()
        red, evs = self.reducer, []
        for g, buckets in zip(self.graphs[1:], self.segment_buckets):
            g.replay()
            e = torch.cuda.Event(); e.record(); evs.append(e)
        with_b = [i for i, b in enumerate(self.segment_buckets) if b]
        for i in with_b:
            buckets = self.segment_buckets[i]
            if i != with_b[-1] or v == "host_paced_all":
                evs[i].synchronize()
                for _, owners in buckets:
                    red._early.update(id(p) for p, _, _ in owners)
                with torch.cuda.stream(red.stream):
                    if os.environ.get("COALESCE"):
                        import torch.distributed as dist
                        with dist.distributed_c10d._coalescing_manager(group=red.group):
                            for flat, _ in buckets:
                                dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=red.group)
                    else:
                        for flat, _ in buckets:
                            red._mean_in_place(flat)
                    done = torch.cuda.Event(); done.record()
                for j, (flat, owners) in enumerate(buckets):
                    red.pending.append((flat, dp._StreamWork(done if j == 0 else None), list(owners)))
            else:
                red.reduce_buckets(buckets)
        red.finish()
        if self._opt_graph is not None:
            self._opt_graph.replay()
        return self.loss
    G.PiecewiseGraphedTrainStep.__call__ = call
if os.environ.get("NOOP_REDUCE"):
    R._mean_in_place = lambda self, t: None
if os.environ.get("SIDE_PRIORITY"):          # the reducer's side stream on a high-priority hardware queue
    init = R.__init__
    def init2(self, *a, **k):
        init(self, *a, **k)
        if self.cuda:
            self.stream = torch.cuda.Stream(priority=int(os.environ["SIDE_PRIORITY"]))
    R.__init__ = init2
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
