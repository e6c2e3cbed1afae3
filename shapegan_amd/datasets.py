"""Input pipeline for the voxel / SDF-cloud training scripts.

Reference: datasets.py:7-40 (`VoxelDataset`: one `.npy [R,R,R] fp32` file per shape, clamp to +-0.1 and rescale to
[-1,1] per item on the CPU, fed through `DataLoader(shuffle=True, num_workers=8)`), train_sdf_autodecoder.py:20-28,55-69
(`data/sdf_points.to [M*200000,3]`, `data/sdf_values.to [M*200000]`, sign-balanced index batches).  File formats are
kept byte-compatible.

MI355X-first data flow: a 32^3 chair set is 0.8 GB and a 64^3 one 6.3 GB — a fraction of one GPU's 288 GB — so the
dataset is read ONCE into HBM (`VoxelDataset.resident`), clamped / rescaled there by one HIP kernel pass, and every
batch is a device-side row gather of shuffled indices: no per-step host work, no H2D copy, no worker processes.
`VoxelStream` covers sets larger than the budget: reader thread -> pinned double buffer -> async H2D on a side stream
-> the same HIP kernel.  Index order of both equals `DataLoader(dataset, shuffle=True, batch_size)` under the same
torch global seed (bit-exact index work, tested on CPU against the real DataLoader).
"""
import glob as _glob
import os
import threading

import numpy as np
import torch

from . import ops


class VoxelDataset(object):
    """datasets.py:7-40: same constructor, `__len__`, `__getitem__` (host path, one item: exactly the reference's
    numpy -> clamp_ -> /= sequence), `glob`, `from_split`.  `resident()` / `stream()` are the native batch paths."""

    def __init__(self, files, clamp=0.1, rescale_sdf=True):
        self.files = files
        self.clamp = clamp
        self.rescale_sdf = rescale_sdf

    def __len__(self):
        return len(self.files)

    def __getitem__(self, index):
        item = torch.from_numpy(np.load(self.files[index]))
        if self.clamp is None:
            return item
        item.clamp_(-self.clamp, self.clamp)
        return item.div_(self.clamp) if self.rescale_sdf else item

    @staticmethod
    def glob(pattern):
        """datasets.py:25-32: recursive glob, sorted; an empty match raises."""
        found = sorted(_glob.glob(pattern, recursive=True))
        if not found:
            raise Exception('No files found for glob pattern {:s}.'.format(pattern))
        return VoxelDataset(found)

    @staticmethod
    def from_split(pattern, split_file_name):
        """datasets.py:34-40: one id per line of the split file, formatted into `pattern`; missing files are skipped."""
        with open(split_file_name, 'r') as fh:
            candidates = [pattern.format(line.strip()) for line in fh.readlines()]
        return VoxelDataset([name for name in candidates if os.path.exists(name)])

    def _divisor(self):
        return float(self.clamp) if (self.clamp is not None and self.rescale_sdf) else 0.0

    def resident(self, device="cuda", max_bytes=64 << 30):
        """Loads every file into one [N,R,R,R] HBM tensor and applies clamp / rescale there (one kernel pass)."""
        first = np.load(self.files[0])
        need = first.nbytes * len(self.files)
        if need > max_bytes:
            raise RuntimeError("dataset needs %.1f GB, above max_bytes; use VoxelDataset.stream()" % (need / 2 ** 30))
        host = torch.empty((len(self.files),) + tuple(first.shape), dtype=torch.float32).pin_memory()
        for i, name in enumerate(self.files):
            host[i] = torch.from_numpy(np.load(name))
        data = host.to(device, non_blocking=True)
        if self.clamp is not None:
            ops.voxel_prepare(data, self.clamp, self._divisor())
        return ResidentVoxels(data)

    def stream(self, batch_size, device="cuda", shuffle=True, drop_last=False, into=None):
        return VoxelStream(self, batch_size, device, shuffle, drop_last, into)


def loader_index_order(n, shuffle=True):
    """The item order `DataLoader(dataset, shuffle=shuffle)` visits under the current torch global RNG state: the
    iterator draws its base seed from the global generator first, then RandomSampler seeds a private generator from it
    and takes one randperm.  Consumes the same global draws, so whatever the scripts draw next is unchanged too."""
    torch.empty((), dtype=torch.int64).random_()                       # _BaseDataLoaderIter._base_seed
    if not shuffle:
        return torch.arange(n)
    seed = int(torch.empty((), dtype=torch.int64).random_().item())   # RandomSampler.__iter__
    generator = torch.Generator()
    generator.manual_seed(seed)
    return torch.randperm(n, generator=generator)


class ResidentVoxels(object):
    """A preprocessed voxel set living in HBM.  `batches()` is one epoch of DataLoader-ordered batches, each produced
    by a device-side row gather (sg_gather_rows)."""

    def __init__(self, data):
        self.data = data
        self.rows = data.reshape(data.shape[0], -1)

    def __len__(self):
        return self.data.shape[0]

    def batches(self, batch_size, shuffle=True, drop_last=False, into=None):
        """into: destination tensors (e.g. `WGANTrainer.real_slots(batch_size)`), used round-robin: batch k is gathered straight
        into into[k % len(into)] and THAT tensor is yielded (reshaped views of it keep its storage), so a trainer whose critic
        batch owns the slot finds its real half in place — no device copy between the loader and the step (train_wgan.py:56-66
        reads the loader's batch where `.to(device)` put it).  A short last batch is yielded as a fresh tensor."""
        order = loader_index_order(len(self), shuffle).to(self.data.device)
        n = len(self)
        stop = n - n % batch_size if drop_last else n
        for k, start in enumerate(range(0, stop, batch_size)):
            idx = order[start:min(start + batch_size, n)]
            dst = into[k % len(into)] if into and idx.numel() == batch_size else None
            with torch.no_grad():
                batch = ops.gather_rows(self.rows, idx, out=dst)
            yield batch if dst is not None else batch.reshape((idx.numel(),) + tuple(self.data.shape[1:]))


class VoxelStream(object):
    """Streaming variant for sets that should not live in HBM: a reader thread fills pinned buffers one batch ahead,
    the copy runs on a side stream, clamp / rescale on the device; the consumer waits on an event only."""

    def __init__(self, dataset, batch_size, device="cuda", shuffle=True, drop_last=False, into=None):
        """into: destination tensors used round-robin (see ResidentVoxels.batches): the async H2D copy of batch k lands in
        into[k % len(into)], clamp / rescale run in place there.  Give at least as many slots as batches are alive at once (a
        WGANTrainer unit: five)."""
        self.dataset, self.batch_size, self.device = dataset, batch_size, torch.device(device)
        self.shuffle, self.drop_last, self.into = shuffle, drop_last, into
        shape = tuple(np.load(dataset.files[0]).shape)
        self.pinned = [torch.empty((batch_size,) + shape, dtype=torch.float32).pin_memory() for _ in range(2)]
        self.copy_stream = torch.cuda.Stream(device=self.device)

    def _fill(self, slot, names):
        buf = self.pinned[slot]
        for i, name in enumerate(names):
            buf[i] = torch.from_numpy(np.load(name))

    def __iter__(self):
        ds = self.dataset
        order = loader_index_order(len(ds), self.shuffle).tolist()
        n = len(order)
        stop = n - n % self.batch_size if self.drop_last else n
        chunks = [[ds.files[j] for j in order[s:min(s + self.batch_size, n)]] for s in range(0, stop, self.batch_size)]
        if not chunks:
            return
        reader = threading.Thread(target=self._fill, args=(0, chunks[0]))
        reader.start()
        free = [None, None]   # event after which pinned[slot] may be overwritten
        for k, names in enumerate(chunks):
            slot = k & 1
            reader.join()
            if k + 1 < len(chunks):
                if free[slot ^ 1] is not None:
                    free[slot ^ 1].synchronize()
                reader = threading.Thread(target=self._fill, args=(slot ^ 1, chunks[k + 1]))
                reader.start()
            dst = self.into[k % len(self.into)] if self.into and len(names) == self.batch_size else None
            if dst is not None:      # the slot's previous batch may still be read by kernels on the consumer's stream
                self.copy_stream.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.copy_stream):
                if dst is not None:
                    batch = dst
                    batch.view(self.pinned[slot].shape).copy_(self.pinned[slot], non_blocking=True)
                else:
                    batch = self.pinned[slot][:len(names)].to(self.device, non_blocking=True)
                if ds.clamp is not None:
                    ops.voxel_prepare(batch, ds.clamp, ds._divisor())
                done = torch.cuda.Event()
                done.record(self.copy_stream)
            free[slot] = done
            torch.cuda.current_stream(self.device).wait_event(done)
            batch.record_stream(torch.cuda.current_stream(self.device))
            yield batch


def load_sdf_clouds(directory="data", device="cuda"):
    """train_sdf_autodecoder.py:20-28: the combined point clouds written by prepare_data.py:102-122
    (`sdf_points.to [M*200000,3]`, `sdf_values.to [M*200000]`, torch.save format); returns (points, sdf, signs) with
    `signs = sdf > 0` taken BEFORE the trainer clamps, as the reference does (:25-27)."""
    points = torch.load(os.path.join(directory, "sdf_points.to")).to(device)
    sdf = torch.load(os.path.join(directory, "sdf_values.to")).to(device)
    signs = sdf.cpu().numpy() > 0
    return points, sdf, signs


def create_batches(signs, batch_size, rng=np.random):
    """train_sdf_autodecoder.py:55-69: one epoch of sign-balanced index batches.  Only the LARGER sign class is shuffled
    (ties: the positive one) and cut to the smaller one's size; negatives then positives are concatenated, shuffled and
    cut into batches, the last batch keeping the remainder.  The numpy RNG is called in the reference's order, so the
    batches are bit-identical under the same np.random.seed."""
    pos, neg = np.nonzero(signs)[0], np.nonzero(~signs)[0]
    if neg.shape[0] > pos.shape[0]:
        rng.shuffle(neg)
        neg = neg[:pos.shape[0]]
    else:
        rng.shuffle(pos)
        pos = pos[:neg.shape[0]]
    epoch = np.concatenate((neg, pos))
    rng.shuffle(epoch)
    full = epoch.shape[0] // batch_size
    for b in range(full - 1):
        yield epoch[b * batch_size:(b + 1) * batch_size]
    yield epoch[(full - 1) * batch_size:]


class PointDataset(object):
    """datasets.py:53-92 (train_point_gan.py:28-29): per shape `<root>/uniform/<name>.npy` and `<root>/surface/<name>.npy`
    ([M,4] xyz+sdf rows); an item is the SAME random subset of `num_points` rows of both (one np.random.choice draw per
    item, with replacement, from the global numpy RNG — bit-exact index work under np.random.seed)."""

    def __init__(self, root, filenames, num_points=1024, transform=None):
        self.root = os.path.expanduser(os.path.join(os.path.normpath(root)))
        self.filenames = filenames
        self.num_points = num_points
        assert 0 < self.num_points <= 64 ** 3
        self.transform = transform

    def __len__(self):
        return len(self.filenames)

    def __getitem__(self, idx):
        name = self.filenames[idx]
        clouds = [torch.from_numpy(np.load(os.path.join(self.root, kind, '{}.npy'.format(name))))
                  for kind in ('uniform', 'surface')]
        rows = np.random.choice(clouds[0].size(0), self.num_points)
        item = (clouds[0][rows], clouds[1][rows])
        return item if self.transform is None else self.transform(item)

    @staticmethod
    def from_split(root, split, num_points=1024, transform=None):
        with open(os.path.join(root, '{}.txt'.format(split)), 'r') as fh:
            names = fh.read().split('\n')
        if names and names[-1] == '':
            names = names[:-1]
        return PointDataset(root, names, num_points, transform)
