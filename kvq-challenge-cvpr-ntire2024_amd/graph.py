"""hipGraph replay of a whole per-video forward.

The per-video forwards of the harness are launch-bound when the batch is one video: KSVQE enqueues ~360 kernels for ~4 ms
of GPU work and the Python / ctypes enqueue costs ~6 ms, so the GPU idles (tools/ksvqe_probe.py: 168 -> 296 samples/s at
one sample per forward with replay).  ``LaneGraphs`` records one forward per (HIP stream "lane", input signature) into a
hipGraph — static input buffers, the forward's temporaries in the graph's private pool — and replays it for every later
video of that signature; lanes replay concurrently, like the eager streams of ``Trainer._score_all``.

What makes a forward capturable here: no host synchronisation inside it (host-cached scalars, device-side index logic),
per-stream plans / workspaces / weight images created by the two eager warm-up runs on the lane's stream, and every kernel
of ``libkvq_hip.so`` launched on the caller's stream handle.  Nothing else changes: the same kernels run, in the same order,
on the same data — the scores are bit-identical to the eager path (tests/test_gpu_harness.py).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Tuple

import torch


def _is_lazy(v) -> bool:
    return hasattr(v, "materialise")


def _slot_ok(v) -> bool:
    """a lazily sampled batch (kernels.FragmentSource) the fused embedding read can take through a pointer table"""
    return _is_lazy(v) and hasattr(v, "pointer_table") and v.c_struct() is not None


def _signature(inputs: Dict[str, torch.Tensor]) -> Tuple:
    """what a recording is valid for: tensor shapes / dtypes; for a lazily sampled batch also everything the embedding launch bakes
    into its parameters — sampler geometry, source frame shape / stride / dtype and the normalisation constants"""
    def lazy_part(v):
        v0 = v.videos[0]
        return (v.geometry, tuple(v0.shape), v0.stride(0), v0.dtype, None if v.mean is None else tuple(v.mean),
                None if v.std is None else tuple(v.std))
    return tuple((k, tuple(v.shape), v.dtype) + (lazy_part(v) if _is_lazy(v) else ()) for k, v in sorted(inputs.items()))


def _has_lazy(inputs) -> bool:
    return any(_is_lazy(v) for v in inputs.values())


class LaneGraphs:
    def __init__(self, fn: Callable[[Dict[str, torch.Tensor]], torch.Tensor], lanes: List[torch.cuda.Stream], max_signatures: int = 4,
                 warmup: int = 2):
        """fn(inputs) -> device tensor, inputs a dict of device tensors; ``lanes``: side streams (capture cannot run on the
        default stream).  At most ``max_signatures`` input signatures are recorded per lane; further ones run eagerly."""
        self.fn, self.lanes, self.max_signatures, self.warmup = fn, lanes, max_signatures, warmup
        self._graphs: List[Dict[Tuple, Tuple]] = [dict() for _ in lanes]
        self._slot_fails = False            # a lazy batch could not be recorded through a FragmentSlot: later ones are sampled first
        self.replays = self.eager_runs = 0

    def _record(self, lane: int, sig: Tuple, inputs: Dict[str, torch.Tensor]):
        st = self.lanes[lane]
        with torch.cuda.stream(st):
            from .kernels import FragmentSlot
            static = {k: (FragmentSlot(v) if _is_lazy(v) else v.clone()) for k, v in inputs.items()}
            for _ in range(self.warmup):                       # plans, workspaces, weight images, tap tables: created eagerly
                self.fn(dict(static))
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        try:
            # thread_local: the prefetch thread (datasets/prefetch.py) keeps allocating and copying while this thread records
            with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
                out = self.fn(dict(static))
        except Exception as e:  # noqa: BLE001
            # a forward that cannot be recorded (a host synchronisation or an allocation of the library inside it) still runs:
            # the same kernels, launched eagerly — say so once per signature instead of failing the job
            import warnings
            warnings.warn(f"hipGraph capture failed for {sig[0][0]}... ({type(e).__name__}: {str(e).splitlines()[0][:160]}); "
                          f"this input signature runs with eager launches", RuntimeWarning)
            torch.cuda.synchronize()
            self._graphs[lane][sig] = None
            return
        self._graphs[lane][sig] = (g, static, out)

    def _materialised(self, lane: int, inputs):
        """the lazy views of ``inputs`` as their fp32 tensors (on the lane's stream): a signature that no longer depends on the source
        resolution, stride or dtype"""
        with torch.cuda.stream(self.lanes[lane]):
            return {k: (v.materialise() if _is_lazy(v) else v) for k, v in inputs.items()}

    def run(self, lane: int, inputs: Dict[str, torch.Tensor]) -> torch.Tensor:
        """Enqueue fn(inputs) on lane's stream.  The returned tensor is the graph's static output: consume it on the same
        stream before the lane's next ``run`` (stream order makes that safe without host synchronisation)."""
        if any(_is_lazy(v) and not _slot_ok(v) for v in inputs.values()):
            # a recorded forward reads fixed addresses.  A lazily sampled view (kernels.FragmentSource: per-video frame / draw
            # pointers) stays lazy — the recorded forward reads them from a device table (kernels.FragmentSlot) that is rewritten in
            # front of each replay; one the fused read cannot take (fp32 frames, > 16 clips) becomes its fp32 tensor first
            with torch.cuda.stream(self.lanes[lane]):
                inputs = {k: (v.materialise() if _is_lazy(v) and not _slot_ok(v) else v) for k, v in inputs.items()}
        graphs = self._graphs[lane]
        if self._slot_fails and _has_lazy(inputs):
            inputs = self._materialised(lane, inputs)
        sig = _signature(inputs)
        lazy_sigs = sum(1 for s_ in graphs if any(len(part) > 3 for part in s_))
        if sig not in graphs and _has_lazy(inputs) and lazy_sigs >= self.max_signatures:
            # the budget of per-resolution recordings is spent (a dataset of many source resolutions): the least recently replayed
            # one makes room — its graph, static slot and the frames the slot kept alive are released
            victim = next(s_ for s_ in graphs if any(len(part) > 3 for part in s_))
            del graphs[victim]
        if sig not in graphs and (_has_lazy(inputs) or len(graphs) - lazy_sigs < self.max_signatures):
            self._record(lane, sig, inputs)
        rec = graphs.get(sig)
        if rec is None and _has_lazy(inputs):
            # a lazy batch whose forward cannot be recorded through the slot (a model that materialises it: the fused read off, an
            # embedding width without it, the upsample fallback): sample first into a STATIC fp32 tensor and record the forward on
            # that — one recording whatever the source resolution, instead of eager launches on the replay configuration
            self._slot_fails = True
            inputs = self._materialised(lane, inputs)
            sig = _signature(inputs)
            if sig not in graphs and len(graphs) - lazy_sigs < self.max_signatures:
                self._record(lane, sig, inputs)
            rec = graphs.get(sig)
        if rec is None:                                        # beyond max_signatures, or a signature whose capture failed
            self.eager_runs += 1
            with torch.cuda.stream(self.lanes[lane]):
                return self.fn(inputs)
        graphs[sig] = graphs.pop(sig)                          # most recently used last
        g, static, out = rec
        with torch.cuda.stream(self.lanes[lane]):
            for k, v in inputs.items():
                if _is_lazy(v):
                    static[k].load(v)
                else:
                    static[k].copy_(v, non_blocking=True)
            g.replay()
            for k, v in inputs.items():
                if _is_lazy(v):
                    static[k].release(self.lanes[lane])        # the frames live as long as the replay that reads them, not longer
        self.replays += 1
        return out
