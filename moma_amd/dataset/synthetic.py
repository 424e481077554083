"""Synthetic loader: yields `(images fp32 [B,3,S,S], labels int64 [B])` like the reference's
DatasetSerial.__getitem__ batches (dataset/dataset.py:27-44).  The real loaders need author-local folders /
network downloads (out of scope); benchmarks and smoke runs use this one.  Batches are
pre-generated on the target device so the loader never sits in the timed region."""
import torch


class SyntheticLoader:
    def __init__(self, n_batches, batch_size, image_size, n_cls, seed=12345, device="cpu", distinct=2,
                 last_batch=None):
        g = torch.Generator().manual_seed(seed)
        self.n_batches = n_batches
        self.last_batch = last_batch            # optional smaller final batch (drop_last=False behaviour)
        self.images = [torch.randn(batch_size, 3, image_size, image_size, generator=g).to(device)
                       for _ in range(distinct)]
        self.labels = [torch.randint(0, n_cls, (batch_size,), generator=g).to(device) for _ in range(distinct)]

    def __len__(self):
        return self.n_batches

    def __iter__(self):
        for i in range(self.n_batches):
            x, y = self.images[i % len(self.images)], self.labels[i % len(self.labels)]
            if self.last_batch and i == self.n_batches - 1:
                x, y = x[:self.last_batch], y[:self.last_batch]
            yield x, y
