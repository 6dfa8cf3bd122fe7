"""Batch feeders for the HIP path.

The reference feeds the model through torch's DataLoader(num_workers=0) over a per-sample `__getitem__`
(`process_data.py:18`, `base_dataset.py:105-120`): a Python call and a dict per SAMPLE, then a collate — a few
hundred thousand samples per second at best, two orders of magnitude below what the HIP path consumes.
The encoded columns of a BaseDataset / MultiTaskDataset are plain tensors (`data_dict`, labels), so a batch is
just a slice of every column:

  DeviceBatchLoader   the columns are moved to the device ONCE; a batch is a dict of device slices (shuffle = one
                      permutation + one gather per column per epoch).  For data that fits HBM (288 GB per MI355X).
  PinnedBatchLoader   the columns stay in PINNED host memory, laid out batch-major ([n_batches, n_cols, batch]) so
                      that a batch is ONE contiguous block per dtype; a copy stream moves batch i+1 (two
                      hipMemcpyAsync) into one of `depth` device buffers while the compute stream works on batch i,
                      with events both ways (the consumer waits for the copy; a buffer is reused only after the
                      step that read it).  For data larger than HBM: ~17 MB per 65536-sample Criteo batch,
                      ~0.3 ms of PCIe Gen5 time hidden behind a ~2 ms train step.

Both yield exactly the dicts torch's DataLoader would have produced — same keys, dtypes, batch boundaries, and,
with `shuffle=True` and no explicit generator, THE SAME ORDER under the same global seed: like DataLoader +
RandomSampler they draw one int64 "base seed" from the global CPU RNG per `iter()` and (when shuffling) one more
that seeds the epoch's `torch.randperm` (torch/utils/data/dataloader.py, sampler.py), so a run fed by these
loaders consumes the global RNG exactly like the reference's run and is comparable batch by batch.
RankTrainer / BenchmarkTrainer accept them wherever they accept a DataLoader (`.dataset`, `.batch_size`,
`len()`, iteration).
"""
from typing import Dict, Iterator, List, Optional

import torch


def _columns(dataset) -> Dict[str, torch.Tensor]:
    cols = {k: dataset.data_dict[k] for k in dataset.feature_name}
    cols.update(dataset._label_columns())  # 'label' / 'task{i}_label': what __getitem__ adds to the features
    return cols


def _epoch_permutation(n: int, shuffle: bool, generator: Optional[torch.Generator]) -> Optional[torch.Tensor]:
    """The index order torch's DataLoader(shuffle=...) would use for this epoch, consuming the global CPU RNG the
    same way (base seed per iter(); RandomSampler's own seed + randperm when shuffling)."""
    if generator is None:
        torch.empty((), dtype=torch.int64).random_()  # DataLoader._get_base_seed()
    if not shuffle:
        return None
    if generator is None:
        seed = int(torch.empty((), dtype=torch.int64).random_().item())  # RandomSampler.__iter__
        generator = torch.Generator()
        generator.manual_seed(seed)
    return torch.randperm(n, generator=generator)


class DeviceBatchLoader:
    def __init__(self, dataset, batch_size: int, shuffle: bool = False, device: Optional[torch.device] = None,
                 drop_last: bool = False, generator: Optional[torch.Generator] = None):
        self.dataset = dataset
        self.batch_size = int(batch_size)
        self.shuffle, self.drop_last, self.generator = shuffle, drop_last, generator
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self.columns: Dict[str, torch.Tensor] = {k: v.to(self.device) for k, v in _columns(dataset).items()}
        self.n = len(dataset)

    def __len__(self) -> int:
        return self.n // self.batch_size if self.drop_last else (self.n + self.batch_size - 1) // self.batch_size

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        cols = self.columns
        perm = _epoch_permutation(self.n, self.shuffle, self.generator)
        if perm is not None:
            perm = perm.to(self.device)
            cols = {k: v[perm] for k, v in cols.items()}
        for i in range(len(self)):
            lo = i * self.batch_size
            yield {k: v[lo:lo + self.batch_size] for k, v in cols.items()}


class PinnedBatchLoader:
    """Host-resident columns, pinned; double-buffered asynchronous host->device feed (see the module docstring)."""

    def __init__(self, dataset, batch_size: int, shuffle: bool = False, device: Optional[torch.device] = None,
                 drop_last: bool = False, generator: Optional[torch.Generator] = None, depth: int = 2):
        self.dataset = dataset
        self.batch_size = int(batch_size)
        self.shuffle, self.drop_last, self.generator = shuffle, drop_last, generator
        self.device = torch.device(device) if device is not None else torch.device("cuda")
        if self.device.type != "cuda":
            raise ValueError("PinnedBatchLoader feeds a HIP device; use DeviceBatchLoader / DataLoader on the CPU")
        self.depth = max(2, int(depth))
        # batches the consumer holds at once: 1 = done with batch i when it asks for i+1; 2 = a loop that looks one batch
        # ahead (model_pipeline._one_ahead asks for i+1 BEFORE it runs step i).  Read at the start of each epoch.
        self.hold = 1
        cols = _columns(dataset)
        self.n = len(dataset)
        # one block per dtype: ids (int64) and everything else (float32), column order = the batch dict's key order
        self._keys = list(cols.keys())
        self._groups: List[tuple] = []
        for dt in (torch.int64, torch.float32):
            ks = [k for k in self._keys if cols[k].dtype == dt]
            if ks:
                self._groups.append((dt, ks, torch.stack([cols[k].reshape(-1) for k in ks])))  # [ncols, n]
        other = [k for k in self._keys if cols[k].dtype not in (torch.int64, torch.float32)]
        if other:
            raise TypeError(f"columns {other}: only int64 ids and float32 values are fed")
        self._host: List[torch.Tensor] = []   # per group: [n_batches, ncols, batch] pinned, allocated ONCE, refilled per epoch
        lab = getattr(dataset, "_label_columns", None)
        self._label_keys = set(lab().keys()) if callable(lab) else {k for k in self._keys if k == "label" or k.endswith("_label")}
        self._copy_stream = torch.cuda.Stream(device=self.device)
        self.bytes_per_batch = sum(len(ks) * self.batch_size * (8 if dt == torch.int64 else 4)
                                   for dt, ks, _ in self._groups)

    def __len__(self) -> int:
        return self.n // self.batch_size if self.drop_last else (self.n + self.batch_size - 1) // self.batch_size

    def _stage_epoch(self, perm: Optional[torch.Tensor]):
        """Batch-major pinned layout for this epoch: host[g][i] = the [ncols, batch] block of batch i (the tail of the
        last, partial batch is padding that is never handed out)."""
        nb, bs = len(self), self.batch_size
        first = not self._host
        if not first:
            # the pinned blocks are refilled IN PLACE: every host->device copy of the previous epoch that was still queued
            # (the host runs ahead of the device) must have read them before they are overwritten
            self._copy_stream.synchronize()
        for g, (dt, ks, mat) in enumerate(self._groups):
            if first:  # the pinned batch-major block of this group: one hipHostMalloc for the loader's lifetime
                self._host.append(torch.zeros((nb, len(ks), bs), dtype=dt).pin_memory())
            src = mat if perm is None else torch.index_select(mat, 1, perm)
            full = min(src.shape[1], nb * bs) // bs  # whole batches; the tail batch (if any) is copied separately
            if full:
                self._host[g][:full].copy_(src[:, :full * bs].reshape(len(ks), full, bs).permute(1, 0, 2))
            rest = min(src.shape[1], nb * bs) - full * bs
            if rest:
                self._host[g][full, :, :rest].copy_(src[:, full * bs:full * bs + rest])
            if perm is None and not self.shuffle:
                self._groups[g] = (dt, ks, None)  # never re-staged: the stacked copy is not needed any more

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        perm = _epoch_permutation(self.n, self.shuffle, self.generator)
        if perm is not None or not self._host:
            self._stage_epoch(perm)
        nb, bs = len(self), self.batch_size
        hold = max(1, int(self.hold))
        depth = max(self.depth, hold + 1)
        dev_bufs = [[torch.empty((len(ks), bs), dtype=dt, device=self.device) for dt, ks, _ in self._groups]
                    for _ in range(depth)]
        copied = [torch.cuda.Event() for _ in range(depth)]   # copy of the buffer finished
        released = [None] * depth                             # consumer done with the buffer

        def launch(i):
            slot = i % depth
            with torch.cuda.stream(self._copy_stream):
                if released[slot] is not None:
                    self._copy_stream.wait_event(released[slot])   # the step that read this buffer has been enqueued
                for g, buf in enumerate(dev_bufs[slot]):
                    buf.copy_(self._host[g][i], non_blocking=True)
                copied[slot].record(self._copy_stream)

        for i in range(min(depth - hold, nb)):
            launch(i)
        for i in range(nb):
            if i + depth - hold < nb:
                launch(i + depth - hold)   # keep depth-hold batches in flight ahead of the consumer
            slot = i % depth
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(copied[slot])
            rows = bs if (i + 1 < nb or self.n % bs == 0 or self.drop_last) else self.n - i * bs
            batch = {}
            for (dt, ks, _), buf in zip(self._groups, dev_bufs[slot]):
                for j, k in enumerate(ks):
                    # feature columns are views of the recycled buffer (consumed within the step); label columns
                    # are kept by the training loop for the epoch's metrics, so they get their own (tiny) tensor
                    batch[k] = buf[j, :rows].clone() if k in self._label_keys else buf[j, :rows]
            yield {k: batch[k] for k in self._keys}
            # asked for the next batch: everything the consumer enqueued for batch i - (hold - 1) is ahead of this event
            # on its stream
            if i - (hold - 1) >= 0:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
                released[(i - (hold - 1)) % depth] = ev
