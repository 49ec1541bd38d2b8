"""One-batch-ahead sample preparation on the training GPU itself.

The reference prepares samples on DataLoader worker PROCESSES (efg/data/build.py `build_dataloader`: num_workers
x NumPy augmentation + CPU voxelization, then collate + upload).  With the loader-side chain running as HIP kernels
on the training GPU (gpu_pipeline.py, gt_database.py) worker processes are the wrong shape: the clouds and the object
database live in this process's HBM.  What the chain still needs is to stay OFF the training stream: its range filter
sizes the cloud with one count read-back, and on the main stream that read-back waits for every kernel of the
previous step still queued there (measured: 37.8 -> 46.2 ms/step inline, 38.9 through this loader;
scripts/ubench/pipeline_step.py, profiles/r02k_pipeline_step.txt).

`DeviceLoader` runs `produce(index) -> (sample, info)` on ONE background thread under its own HIP stream, `depth`
batches ahead; each sample leaves with its batch's `ready_event` recorded on that stream, which is what VoxelDETR /
CenterPoint / TrajectoryFormer wait on before their first kernel (operators/voxelize.py `wait_for_points`), so no stream
ever waits for the host.
One producer thread (not a pool) keeps the reference's NumPy random stream in order: sample i draws before sample
i+1, as with num_workers=0.  The model's own NumPy draws, if any, must use their own generator or hold
`NUMPY_GLOBAL_RNG_LOCK` (below).
"""
import queue
import threading

import torch


# The reference's processors (and their mirrors here: gpu_pipeline.py, gt_database.py, tracking/aug.py,
# TrajectoryFormer.prepare) draw from NumPy's GLOBAL generator, some of them rewinding it with get_state / set_state --
# kept, because the goldens pin exactly those draws.  The producer thread holds this lock while it makes a batch;
# anything else in the process that uses the global generator while a DeviceLoader is alive (none of this package's
# model code does: CDN noise is a torch.Generator, the synthetic scenes use default_rng) must take it too, or use its
# own np.random.Generator.
NUMPY_GLOBAL_RNG_LOCK = threading.RLock()


class _Failure:
    def __init__(self, exc):
        self.exc = exc


class DeviceLoader:
    def __init__(self, produce, batch_size, length, device=None, depth=2, collate=None):
        """produce(i) for i in range(length * batch_size), batched in order; device: the GPU the chain runs on
        (None: plain host producer, no streams -- used by the CPU tests); collate(batch) -> batch: per-batch work on
        the loader's thread and stream (TrajectoryFormer.prepare: NMS, linking, hypotheses, point crop, targets)."""
        self.produce, self.batch_size, self.length, self.depth = produce, batch_size, length, depth
        self.collate = collate
        self.device = torch.device(device) if device is not None else None
        self._queue = queue.Queue(maxsize=depth)
        self._stop = threading.Event()
        self._stream = None
        if self.device is not None:
            self._stream = torch.cuda.Stream(device=self.device)
            # whatever the producer reads (raw clouds, the object database) was uploaded on the caller's stream
            self._stream.wait_stream(torch.cuda.current_stream(self.device))
        self._thread = threading.Thread(target=self._work, name="efg-device-loader", daemon=True)
        self._served = 0
        self._thread.start()

    def _put(self, item):
        while not self._stop.is_set():
            try:
                self._queue.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def _make(self, b):
        batch = []
        with NUMPY_GLOBAL_RNG_LOCK:
            for i in range(b * self.batch_size, (b + 1) * self.batch_size):
                if self._stop.is_set():
                    return None
                batch.append(self.produce(i))
            return self.collate(batch) if self.collate is not None else batch

    def _work(self):
        try:
            if self.device is not None:
                torch.cuda.set_device(self.device)   # the current device is per thread
            for b in range(self.length):
                if self._stream is None:
                    batch = self._make(b)
                else:
                    with torch.cuda.stream(self._stream):
                        batch = self._make(b)
                        if batch is not None:
                            event = torch.cuda.Event()
                            event.record(self._stream)   # ONE event per batch: everything above is complete when it fires
                            for sample, _ in batch:
                                (sample[0] if isinstance(sample, (list, tuple)) else sample)["ready_event"] = event
                if batch is None or not self._put(batch):
                    return
        except BaseException as exc:  # noqa: BLE001 -- handed to the consumer, which re-raises it
            self._put(_Failure(exc))

    def __len__(self):
        return self.length

    def __iter__(self):
        return self

    def __next__(self):
        if self._served == self.length:
            raise StopIteration
        while True:   # never block for good: the producer may be gone (close(), or it died without a word)
            try:
                item = self._queue.get(timeout=0.2)
                break
            except queue.Empty:
                if self._stop.is_set():
                    raise StopIteration from None
                if not self._thread.is_alive() and self._queue.empty():
                    served, self._served = self._served, self.length
                    raise RuntimeError("DeviceLoader: the producer thread exited after %d of %d batches without "
                                       "reporting an error" % (served, self.length)) from None
        if isinstance(item, _Failure):
            self._served = self.length
            raise item.exc
        self._served += 1
        return item

    def close(self):
        self._stop.set()
        while True:   # unblock a producer waiting on a full queue
            try:
                self._queue.get_nowait()
            except queue.Empty:
                break
        self._thread.join(timeout=30)
        if self._stream is not None and not self._thread.is_alive():
            self._stream.synchronize()   # nothing of the loader is in flight once close() has returned

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
