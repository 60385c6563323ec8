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
    return tuple((k, tuple(v.shape), v.dtype) + ((v.geometry, tuple(v.videos[0].shape), v.videos[0].stride(0)) if _is_lazy(v) else ())
                 for k, v in sorted(inputs.items()))


class LaneGraphs:
    def __init__(self, fn: Callable[[Dict[str, torch.Tensor]], torch.Tensor], lanes: List[torch.cuda.Stream], max_signatures: int = 4,
                 warmup: int = 2):
        """fn(inputs) -> device tensor, inputs a dict of device tensors; ``lanes``: side streams (capture cannot run on the
        default stream).  At most ``max_signatures`` input signatures are recorded per lane; further ones run eagerly."""
        self.fn, self.lanes, self.max_signatures, self.warmup = fn, lanes, max_signatures, warmup
        self._graphs: List[Dict[Tuple, Tuple]] = [dict() for _ in lanes]
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

    def run(self, lane: int, inputs: Dict[str, torch.Tensor]) -> torch.Tensor:
        """Enqueue fn(inputs) on lane's stream.  The returned tensor is the graph's static output: consume it on the same
        stream before the lane's next ``run`` (stream order makes that safe without host synchronisation)."""
        if any(_is_lazy(v) and not _slot_ok(v) for v in inputs.values()):
            # a recorded forward reads fixed addresses.  A lazily sampled view (kernels.FragmentSource: per-video frame / draw
            # pointers) stays lazy — the recorded forward reads them from a device table (kernels.FragmentSlot) that is rewritten in
            # front of each replay; one the fused read cannot take (fp32 frames, > 16 clips) becomes its fp32 tensor first
            with torch.cuda.stream(self.lanes[lane]):
                inputs = {k: (v.materialise() if _is_lazy(v) and not _slot_ok(v) else v) for k, v in inputs.items()}
        sig = _signature(inputs)
        if sig not in self._graphs[lane] and len(self._graphs[lane]) < self.max_signatures:
            self._record(lane, sig, inputs)
        rec = self._graphs[lane].get(sig)
        if rec is None:                                        # beyond max_signatures, or a signature whose capture failed
            self.eager_runs += 1
            with torch.cuda.stream(self.lanes[lane]):
                return self.fn(inputs)
        g, static, out = rec
        with torch.cuda.stream(self.lanes[lane]):
            for k, v in inputs.items():
                if _is_lazy(v):
                    static[k].load(v)
                else:
                    static[k].copy_(v, non_blocking=True)
            g.replay()
        self.replays += 1
        return out
