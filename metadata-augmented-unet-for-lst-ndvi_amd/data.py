"""Input pipeline for the MI355X path: mirrors ``src/dataset.py`` of the reference.

The reference ships every tile as 23 fp32 planes (``FuturePredictionDataset.__getitem__``, src/dataset.py:53-72;
``collate_fn``, :91-112) although 18 of them are the one-hot expansion of two 9-class land-cover maps
(``np.vstack([dw_t1, rgb, ndvi, temp, dw_t2])``, src/data/processing_10m/process.py:176-181; ``one_hot_encode`` =
``np.eye(9)[img]``, normalization.py:96-100).  At the step rates of the HIP path that host copy (92 B/pixel) is the
bottleneck, so here

* a sample is kept *compact*: the two class maps as uint8 + the 5 continuous planes as fp32 (22 B/pixel, 4.2x less);
* batches are collated into pinned host memory and copied on a side stream while the previous step computes
  (``DeviceLoader``);
* the one-hot channels, the NHWC-ld layout, the bf16 cast and ``RandomFlip`` (src/dataset.py:134-141) are ONE device
  kernel, ``mau_pack_tile_onehot`` (include/mau_hip.h); the network takes its result directly
  (``UrbanPredictor.forward(maps=Act)``).

The values the network sees are bit-identical to the reference's dense path followed by ``ToNHWC``
(tests/test_gpu_ops.py::test_pack_tile_onehot_matches_dense_path).
"""
from __future__ import annotations

import os
import random
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch.nn.utils.rnn import pad_sequence
from torch.utils.data import DataLoader, Dataset

from .functional import Act, _stream, dtype_code, pad8
from ._lib import call

NUM_CLASSES = 9          # Dynamic World classes (normalization.py:96)
N_CONT = 5               # rgb (3) + ndvi + lst


# --------------------------------------------------------------------------- #
# compact <-> dense sample (host, numpy)
# --------------------------------------------------------------------------- #
def compact_input(inp: np.ndarray, num_classes: int = NUM_CLASSES) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(2*nc + k, H, W) dense input of the reference -> (cls_a uint8 (H,W), cls_b uint8 (H,W), cont fp32 (k,H,W)).

    Raises ``ValueError`` unless the first and last ``num_classes`` planes are exact one-hot encodings -- the compact
    form must reproduce the dense one bit for bit."""
    inp = np.asarray(inp)
    if inp.ndim != 3 or inp.shape[0] < 2 * num_classes:
        raise ValueError(f"compact_input: expected (>= {2 * num_classes}, H, W), got {inp.shape}")
    a, b = inp[:num_classes], inp[inp.shape[0] - num_classes:]
    cont = np.ascontiguousarray(inp[num_classes:inp.shape[0] - num_classes], dtype=np.float32)
    if cont.shape[0] > 8:
        raise ValueError("compact_input: at most 8 continuous planes")
    out = []
    for name, oh in (("first", a), ("last", b)):
        if not (np.all((oh == 0) | (oh == 1)) and np.all(oh.sum(axis=0) == 1)):
            raise ValueError(f"compact_input: the {name} {num_classes} planes are not a one-hot class map")
        out.append(np.argmax(oh, axis=0).astype(np.uint8))
    return out[0], out[1], cont


def expand_input(cls_a: np.ndarray, cls_b: np.ndarray, cont: np.ndarray, num_classes: int = NUM_CLASSES) -> np.ndarray:
    """Host inverse of :func:`compact_input` (the reference's stacking order); used by tests and the dense fallback."""
    eye = np.eye(num_classes, dtype=np.float32)
    return np.vstack([eye[cls_a.astype(int)].transpose(2, 0, 1), cont.astype(np.float32),
                      eye[cls_b.astype(int)].transpose(2, 0, 1)])


# --------------------------------------------------------------------------- #
# transforms
# --------------------------------------------------------------------------- #
class RandomFlip:
    """Horizontal flip with probability 1/2 -- src/dataset.py:134-141, same random stream (``random.seed(seed)`` at
    construction, one ``random.random()`` per sample).  ``__call__(x, y)`` is the reference's host transform;
    ``draw()`` only takes the decision and leaves the mirroring to the device kernel."""

    def __init__(self, seed: int = 42):
        random.seed(seed)

    def draw(self) -> bool:
        return random.random() < 0.5

    def __call__(self, x, y):
        if self.draw():
            x = np.flip(x, axis=2).copy()
            y = np.flip(y, axis=2).copy()
        return x, y


# --------------------------------------------------------------------------- #
# dataset
# --------------------------------------------------------------------------- #
class FuturePredictionDataset(Dataset):
    """``.npz`` tiles written by the reference's ``process_and_save_subset`` (keys input / target / metadata /
    temperature_serie), file-name date parsing as src/dataset.py:44-51.

    ``compact=True`` (default) yields a dict with the class maps as uint8 and a flip flag; ``compact=False`` yields
    the reference's 6-tuple ``(input, metadata, temp_series, t1_date, t2_date, target)`` with the transform applied on
    the host exactly like the reference."""

    def __init__(self, split: str, transform=None, processed_dir: Optional[str] = None, compact: bool = True,
                 num_classes: int = NUM_CLASSES):
        if processed_dir is None:
            processed_dir = os.environ.get("PROCESSED_IMAGE_DATASET", os.path.join("data", "processed"))
        self.processed_dir, self.split, self.transform = processed_dir, split, transform
        self.compact, self.num_classes = compact, num_classes
        self.data_dir = os.path.join(processed_dir, split)
        if not os.path.isdir(self.data_dir):
            raise FileNotFoundError(f"Directory for split '{split}' not found at: {self.data_dir}")
        self.file_list = sorted(os.path.join(self.data_dir, f) for f in os.listdir(self.data_dir) if f.endswith(".npz"))

    def __len__(self):
        return len(self.file_list)

    @staticmethod
    def _dates(filepath: str):
        parts = os.path.basename(filepath).split("_")
        return (int(parts[-5]), int(parts[-4])), (int(parts[-2]), int(parts[-1].split(".")[0]))

    def __getitem__(self, idx):
        filepath = self.file_list[idx]
        (t1y, t1m), (t2y, t2m) = self._dates(filepath)
        data = np.load(filepath)
        inp, tgt = data["input"], data["target"]
        meta = torch.from_numpy(data["metadata"]).float()
        ts = torch.from_numpy(data["temperature_serie"]).float()
        t1 = torch.tensor([t1y, t1m]).float()
        t2 = torch.tensor([t2y, t2m]).float()
        if not self.compact:
            if self.transform:
                inp, tgt = self.transform(inp, tgt)
            return torch.from_numpy(inp).float(), meta, ts, t1, t2, torch.from_numpy(tgt).float()
        flip = False
        if self.transform is not None:
            if not hasattr(self.transform, "draw"):
                raise TypeError("compact samples need a transform with draw() (mau_amd.data.RandomFlip); use compact=False otherwise")
            flip = self.transform.draw()
        a, b, cont = compact_input(inp, self.num_classes)
        return {"cls_a": torch.from_numpy(a), "cls_b": torch.from_numpy(b), "cont": torch.from_numpy(cont), "flip": flip,
                "metadata": meta, "temp_series": ts, "t1_date": t1, "t2_date": t2,
                "target": torch.from_numpy(np.ascontiguousarray(tgt, dtype=np.float32))}

    def get_metadata_from_idx(self, idx: int) -> dict:
        parts = os.path.basename(self.file_list[idx]).split("_")
        return {"city": " ".join(parts[:-8]), "lat": float(parts[-7]), "lon": float(parts[-6])}


@dataclass
class CompactBatch:
    """A collated batch in pinned host memory (or already on the device after ``.to``)."""
    cls_a: torch.Tensor            # (B,H,W) uint8
    cls_b: torch.Tensor            # (B,H,W) uint8
    cont: torch.Tensor             # (B,k,H,W) fp32
    flip: torch.Tensor             # (B,) uint8
    metadatas: torch.Tensor
    temp_series: torch.Tensor      # padded like the reference's collate_fn
    temp_series_lengths: torch.Tensor
    t1_dates: torch.Tensor
    t2_dates: torch.Tensor
    targets: torch.Tensor          # (B,2,H,W) fp32, NOT yet flipped
    num_classes: int = NUM_CLASSES

    _DEVICE_FIELDS = ("cls_a", "cls_b", "cont", "flip", "metadatas", "temp_series", "t1_dates", "t2_dates", "targets")

    def pin(self) -> "CompactBatch":
        if torch.cuda.is_available():
            for f in self._DEVICE_FIELDS:
                setattr(self, f, getattr(self, f).pin_memory())
        return self

    def to(self, device, non_blocking: bool = True) -> "CompactBatch":
        kw = {f: getattr(self, f).to(device, non_blocking=non_blocking) for f in self._DEVICE_FIELDS}
        return CompactBatch(temp_series_lengths=self.temp_series_lengths, num_classes=self.num_classes, **kw)

    def host_bytes(self) -> int:
        return sum(getattr(self, f).numel() * getattr(self, f).element_size() for f in self._DEVICE_FIELDS)


def collate_fn(batch: Sequence[dict]) -> CompactBatch:
    """Compact counterpart of the reference's ``collate_fn`` (src/dataset.py:91-112): stacks on the HOST; the device
    copy is the loader's job (the reference copies inside collate_fn, which serialises it with the step)."""
    batch = [b for b in batch if b is not None]
    if not batch:
        raise ValueError("collate_fn: empty batch")
    ts = [b["temp_series"] for b in batch]
    return CompactBatch(
        cls_a=torch.stack([b["cls_a"] for b in batch]), cls_b=torch.stack([b["cls_b"] for b in batch]),
        cont=torch.stack([b["cont"] for b in batch]), flip=torch.tensor([1 if b["flip"] else 0 for b in batch], dtype=torch.uint8),
        metadatas=torch.stack([b["metadata"] for b in batch]).float(),
        temp_series=pad_sequence(ts, batch_first=True, padding_value=0.0).float(),
        temp_series_lengths=torch.tensor([len(t) for t in ts]),
        t1_dates=torch.stack([b["t1_date"] for b in batch]).float(), t2_dates=torch.stack([b["t2_date"] for b in batch]).float(),
        targets=torch.stack([b["target"] for b in batch]).float())


# --------------------------------------------------------------------------- #
# device side
# --------------------------------------------------------------------------- #
def pack_tiles(cls_a: torch.Tensor, cls_b: torch.Tensor, cont: torch.Tensor, flip: Optional[torch.Tensor],
               dtype: torch.dtype, num_classes: int = NUM_CLASSES) -> Act:
    """Device tensors (B,H,W) uint8 x2 + (B,k,H,W) fp32 [+ (B,) uint8 flip flags] -> the network's input ``Act``."""
    if not cls_a.is_cuda:
        raise RuntimeError("pack_tiles: the MI355X path takes device tensors (no CPU fallback)")
    if cls_a.dtype != torch.uint8 or cls_b.dtype != torch.uint8 or cls_a.shape != cls_b.shape or cls_a.dim() != 3:
        raise ValueError("pack_tiles: class maps must be (B,H,W) uint8 of equal shape")
    B, H, W = cls_a.shape
    k = cont.shape[1] if cont is not None and cont.numel() else 0
    if k and (cont.dtype != torch.float32 or cont.shape != (B, k, H, W)):
        raise ValueError("pack_tiles: cont must be (B,k,H,W) fp32")
    if flip is not None and (flip.dtype != torch.uint8 or flip.shape != (B,)):
        raise ValueError("pack_tiles: flip must be (B,) uint8")
    C = 2 * num_classes + k
    out = torch.empty((B, H, W, pad8(C)), dtype=dtype, device=cls_a.device)
    call("mau_pack_tile_onehot", cls_a.contiguous().data_ptr(), cls_b.contiguous().data_ptr(),
         cont.contiguous().data_ptr() if k else None, flip.contiguous().data_ptr() if flip is not None else None,
         out.data_ptr(), out.shape[-1], dtype_code(dtype), B, H, W, num_classes, k, _stream())
    return Act(out, C)


def flip_targets(targets: torch.Tensor, flip: torch.Tensor) -> torch.Tensor:
    """The target half of RandomFlip on the device: (B,C,H,W) fp32, mirrored along W where flip[b] != 0."""
    if not targets.is_cuda:
        raise RuntimeError("flip_targets: device tensors only")
    t = targets.contiguous().float()
    B, C, H, W = t.shape
    out = torch.empty_like(t)
    call("mau_flip_rows", t.data_ptr(), out.data_ptr(), flip.contiguous().data_ptr(), B, C, H, W, _stream())
    return out


def to_network_inputs(batch: CompactBatch, dtype: torch.dtype):
    """A device-resident :class:`CompactBatch` -> the 7-tuple of the reference's ``collate_fn``
    ``(inputs, metadatas, temp_series_padded, temp_series_lengths, t1_dates, t2_dates, targets)`` where ``inputs`` is
    the packed ``Act`` the network consumes directly."""
    inputs = pack_tiles(batch.cls_a, batch.cls_b, batch.cont, batch.flip, dtype, batch.num_classes)
    targets = flip_targets(batch.targets, batch.flip)
    return inputs, batch.metadatas, batch.temp_series, batch.temp_series_lengths, batch.t1_dates, batch.t2_dates, targets


class DeviceLoader:
    """Iterates a DataLoader of :class:`CompactBatch` and hands out device batches one step ahead: the pinned-host ->
    HBM copy and the pack kernel of batch i+1 run on a side stream while the caller trains on batch i."""

    def __init__(self, loader, device, dtype: torch.dtype = torch.bfloat16):
        self.loader, self.device, self.dtype = loader, torch.device(device), dtype
        self.stream = torch.cuda.Stream(device=self.device)
        self._epoch = 0

    def __len__(self):
        return len(self.loader)

    @property
    def sampler(self):
        return getattr(self.loader, "sampler", None)

    def set_epoch(self, epoch: int):
        """Data parallel: reseed the DistributedSampler's permutation (the reference's single-process DataLoader reshuffles
        every epoch by itself).  ``__iter__`` advances the epoch on its own when this is never called."""
        self._epoch = int(epoch)
        s = self.sampler
        if hasattr(s, "set_epoch"):
            s.set_epoch(self._epoch)

    def _stage(self, host: CompactBatch):
        with torch.cuda.stream(self.stream):
            dev = host.pin().to(self.device, non_blocking=True)
            out = to_network_inputs(dev, self.dtype)
        return out, host                                   # keep the pinned buffers alive until the copy is consumed

    def __iter__(self):
        self.set_epoch(self._epoch)          # a new permutation per pass over the data (no-op without a distributed sampler)
        self._epoch += 1
        it = iter(self.loader)
        nxt = None
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        while nxt is not None:
            cur, _keep = nxt
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
            for t in cur:                                   # allocator: these tensors are now used on the consumer's stream
                tt = t.t if isinstance(t, Act) else t
                if isinstance(tt, torch.Tensor) and tt.is_cuda:
                    tt.record_stream(torch.cuda.current_stream(self.device))
            try:
                nxt = self._stage(next(it))
            except StopIteration:
                nxt = None
            yield cur


def create_dataloader(split: str, batch_size: int, shuffle: bool, dataset_type: str = "future", transform=None,
                      num_workers: int = 0, processed_dir: Optional[str] = None, device=None,
                      dtype: torch.dtype = torch.bfloat16):
    """Signature of the reference's ``create_dataloader`` (src/dataset.py:114-131) + where the data lives; with a
    ``device`` the result is a :class:`DeviceLoader` yielding the reference's 7-tuple with packed inputs."""
    assert dataset_type == "future", "Only 'future' dataset_type is supported in create_dataloader."
    ds = FuturePredictionDataset(split=split, transform=transform, processed_dir=processed_dir, compact=True)
    sampler, drop_last = None, False
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if shuffle:
            # TRAINING loader, data parallel: every rank must run the SAME number of steps (one BatchNorm / gradient collective
            # per layer and step, a missing partner deadlocks) on equally sized local batches -> disjoint equal shards, ragged
            # tail dropped (SyncBN itself tolerates unequal local batch sizes: the pixel count travels with the statistics);
            # DeviceLoader.set_epoch / __iter__ reseed the permutation every epoch
            from torch.utils.data.distributed import DistributedSampler
            sampler = DistributedSampler(ds, shuffle=True, drop_last=True)
            shuffle, drop_last = False, True
        else:
            # EVALUATION loader: eval-mode forwards issue no collective, so ranks may run different numbers of steps -- every
            # sample is visited exactly once (rank r takes samples r, r + world, ...; nothing dropped, nothing repeated)
            sampler = list(range(dist.get_rank(), len(ds), dist.get_world_size()))
    dl = DataLoader(ds, batch_size=batch_size, shuffle=shuffle, sampler=sampler, drop_last=drop_last, num_workers=num_workers,
                    collate_fn=collate_fn)
    return DeviceLoader(dl, device, dtype) if device is not None else dl
