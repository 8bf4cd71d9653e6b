"""Batch loader with host threads for datasets whose __getitem__ voxelizes on the GPU (hrem.HREMEventFlow, mvsec.MvsecEventFlow).

torch.utils.data.DataLoader workers are processes: they cannot share this process's device context, and the reference's worker
pool (train_EEMFlow_HREM.py:103-104, `--num_workers`) exists to hide exactly the part that stays on the host here - reading and
inflating the event and flow files.  Those release the GIL, so threads do: `ThreadedBatchLoader` maps the flag onto a thread pool
that builds samples ahead (each on its thread's current stream) and hands finished batches to the training loop in the
sampler's order, a HIP event ordering every batch behind the launches that produced it.  Same iteration protocol as the
DataLoader it replaces: iterable of dicts of stacked tensors, `len()`, `.dataset`, `.sampler`, drop_last.
"""
import collections
import concurrent.futures

import torch


class ThreadedBatchLoader:
    def __init__(self, dataset, batch_size, shuffle=False, sampler=None, threads=4, drop_last=True, ahead=2, seed=0):
        self.dataset, self.batch_size, self.sampler = dataset, int(batch_size), sampler
        self.shuffle, self.drop_last, self.threads, self.ahead = shuffle, drop_last, max(1, int(threads)), max(1, int(ahead))
        self._epoch, self._seed = 0, seed
        self._pool = None                                        # created once, reused by every epoch (close() / __del__ shut it down)

    def _executor(self):
        if self._pool is None:
            self._pool = concurrent.futures.ThreadPoolExecutor(max_workers=self.threads, thread_name_prefix="eemflow-loader")
        return self._pool

    def close(self):
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None

    def __del__(self):
        try:
            self.close()
        except Exception:                                        # noqa: BLE001 - interpreter shutdown
            pass

    def _indices(self):
        if self.sampler is not None:
            return list(iter(self.sampler))
        n = len(self.dataset)
        if not self.shuffle:
            return list(range(n))
        g = torch.Generator()
        g.manual_seed(self._seed + self._epoch)
        return torch.randperm(n, generator=g).tolist()

    def __len__(self):
        n = len(self.sampler) if self.sampler is not None else len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    @staticmethod
    def _cuda_device_of(sample):
        """Device of the first CUDA tensor in a sample dict / sequence (None: a host-side sample)."""
        vals = sample.values() if isinstance(sample, dict) else (sample if isinstance(sample, (list, tuple)) else (sample,))
        for v in vals:
            if torch.is_tensor(v) and v.is_cuda:
                return v.device
        return None

    def _sample(self, idx):
        # the current device is per THREAD: a worker starts on device 0 whatever the rank's GPU is, so the sample is built (and its
        # hand-over event recorded) under the dataset's device
        dev = getattr(self.dataset, "device", None)
        if torch.cuda.is_available() and dev is not None and torch.device(dev).type == "cuda":
            with torch.cuda.device(dev):
                sample = self.dataset[idx]
                ready = torch.cuda.Event()
                ready.record(torch.cuda.current_stream(dev))
            return sample, ready, torch.device(dev)
        sample = self.dataset[idx]
        # a wrapped dataset (Subset, ConcatDataset, a user class) has no `.device`: if it built CUDA tensors on this worker thread, the
        # consumer still has to wait for this thread's stream on THAT device
        sdev = self._cuda_device_of(sample) if torch.cuda.is_available() else None
        if sdev is not None:
            with torch.cuda.device(sdev):
                ready = torch.cuda.Event()
                ready.record(torch.cuda.current_stream(sdev))
            return sample, ready, sdev
        return sample, None, None

    def __iter__(self):
        idx = self._indices()
        self._epoch += 1
        batches = [idx[i:i + self.batch_size] for i in range(0, len(idx), self.batch_size)]
        if self.drop_last and batches and len(batches[-1]) < self.batch_size:
            batches.pop()
        pool = self._executor()
        dev = getattr(self.dataset, "device", None)
        queue = collections.deque()
        todo = iter(batches)

        def submit():
            b = next(todo, None)
            if b is not None:
                queue.append([pool.submit(self._sample, i) for i in b])
        for _ in range(self.ahead):
            submit()
        try:
            while queue:
                futures = queue.popleft()
                submit()
                samples = []
                for f in futures:
                    sample, ready, sdev = f.result()
                    if ready is not None:
                        torch.cuda.current_stream(sdev).wait_event(ready)
                    samples.append(sample)
                yield {k: (torch.stack([s[k] for s in samples]) if torch.is_tensor(samples[0][k]) else [s[k] for s in samples])
                       for k in samples[0]}
        finally:                                                 # an abandoned epoch (break in the training loop): do not leave work queued
            for futures in queue:
                for f in futures:
                    f.cancel()
