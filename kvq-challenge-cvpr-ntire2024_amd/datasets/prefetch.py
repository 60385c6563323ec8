"""Input pipeline of the inference loop: the reference hands videos to the model through a DataLoader
(worker processes decode and sample on the CPU, fp32 fragments cross PCIe; ``trainer.py:256-283``).  Here a dataset item is
built ON the device — the frames cross PCIe once, as uint8, and the sampler / resize / normalise kernels of
``fusion_datasets.py`` run in HBM — and ``DevicePrefetcher`` builds the items ``depth`` videos ahead of the model: a host
thread (file read / decode releases the GIL) that enqueues the item's copies and kernels on its own HIP stream and hands
each item over with an event, so neither the host-side decode nor the H2D copy sits between two forwards."""
from __future__ import annotations

import queue
import threading

import torch


class DevicePrefetcher:
    def __init__(self, dataset, indices, device, depth: int = 2):
        self.dataset, self.indices, self.device = dataset, list(indices), torch.device(device)
        self.queue: "queue.Queue" = queue.Queue(maxsize=max(1, depth))
        self.stream = torch.cuda.Stream(device=self.device)
        self._stop = False
        self.thread = threading.Thread(target=self._work, name="kvq-prefetch", daemon=True)
        self.thread.start()

    def _work(self):
        try:
            torch.cuda.set_device(self.device)
            for i in self.indices:
                if self._stop:
                    break
                with torch.cuda.stream(self.stream):
                    item = self.dataset[i]
                    ready = torch.cuda.Event()
                    ready.record(self.stream)
                if not self._put((i, item, ready)):
                    return
        except BaseException as e:  # noqa: BLE001  (re-raised in the consumer)
            self._put(e)
            return
        self._put(None)

    def _put(self, x) -> bool:
        """queue.put that gives up once close() was called: a worker blocked on a full queue must not outlive its consumer
        (it would keep the item's device tensors and the pinned staging buffers alive)."""
        while not self._stop:
            try:
                self.queue.put(x, timeout=0.05)
                return True
            except queue.Full:
                continue
        return False

    def __iter__(self):
        """Yields (index, item, ready): make the consuming stream ``wait_event(ready)`` before it touches the item, and call
        ``hand_over(item, stream)`` so that the caching allocator keeps the item's memory until that stream is done with it."""
        while True:
            x = self.queue.get()
            if x is None:
                return
            if isinstance(x, BaseException):
                raise x
            yield x

    @staticmethod
    def hand_over(item, stream):
        for v in item.values():
            if (torch.is_tensor(v) and v.is_cuda) or hasattr(v, "split_clips"):      # tensors and kernels.FragmentSource
                v.record_stream(stream)

    def close(self):
        """Stop the worker and JOIN it: the queue is drained until the thread has exited (a worker blocked in put wakes up,
        sees ``_stop`` and returns), so no item, event or staging buffer survives an early exit of the consumer."""
        self._stop = True
        while self.thread.is_alive():
            try:
                while True:
                    self.queue.get_nowait()
            except queue.Empty:
                pass
            self.thread.join(timeout=0.05)
        try:
            while True:
                self.queue.get_nowait()
        except queue.Empty:
            pass
