"""Token-id datasets: the reference's pre-tokenized dataset directories, and the format this build
trains from - flat int32 shards that are memory-mapped, sliced per rank without reading what other ranks own, and staged to
the GPU through pinned buffers ahead of the step that consumes them.

Reference: DataManagers/DataUtils.py:12-38 (a directory of `*.pkl` files, each a serialized list of equal-length id lists,
plus one `*.json` {file name: number of samples}) and DataManagers/PickleBasedDataloader.py:19-185.  What its loader does,
with `doShuffle=False`, is a CYCLIC STREAM: files are appended to a sample stream in directory order, whole batches are cut
from the front, an epoch is `int(numSamples / batchSize)` batches and the next epoch continues where the last one stopped
(the stream is never reset); with data parallelism every file is cut into `world` contiguous chunks (sizes differ by at most
one, the first `n % world` chunks are the larger ones) and a rank streams only its chunk of every file (:71-102).
`TokenStreamLoader` follows that batch sequence.  With `doShuffle=True` the reference shuffles file order, the pending stream
and each loader chunk from numpy's global RNG; here: a seeded permutation of the file order per cycle and of the samples of
each file as it enters the stream - reproducible, the same distribution of batches, not the same order.

Shard format (`convertPickleDataset`): `<name>.ids` = little-endian int32 [samples][seq_len], row-major, no header;
`dataset.json` = {"format": "branchgan-ids-v1", "seq_len": S, "files": {"<name>.ids": samples}}.  50k-token vocabularies fit
int32; ids are widened to int64 on the GPU (what the embedding / sampling kernels take).

Reading `*.pkl`: a dataset file is nothing but nested lists of integers, which the pickle format encodes without naming a
single class or function.  `_read_id_lists` therefore decodes with an unpickler that REFUSES every global lookup: a file that
tries to import or call anything is rejected with an error and nothing from it runs.  Plain `pickle.load` is never used.
"""
import json
import os

import numpy as np
import torch

FORMAT = "branchgan-ids-v1"


def getAllMatchingDatasetFiles(mainPath, dataFileExtension='.pkl', metaDataFileExtension='.json'):
    """DataUtils.py:12-28: every data file under mainPath (os.walk order) and the parsed meta file."""
    dataFiles, metaFile = [], None
    for root, _, files in os.walk(mainPath):
        for name in files:
            if name.endswith(dataFileExtension):
                dataFiles.append(os.path.join(root, name))
            elif name.endswith(metaDataFileExtension):
                metaFile = os.path.join(root, name)
    if metaFile is None:
        raise ValueError('No meta file found in {}'.format(mainPath))
    with open(metaFile, 'r') as fp:
        return dataFiles, json.load(fp)


def _read_id_lists(path):
    """A `*.pkl` dataset file -> its nested lists of ints.  Data only: any global (class, function, module attribute) the
    stream asks for is refused, so no code from the file can execute."""
    import pickle

    class _DataOnly(pickle.Unpickler):
        def find_class(self, module, name):
            raise pickle.UnpicklingError(f"{path}: not a plain list of token ids (the file refers to {module}.{name}); refusing to load it")

        def persistent_load(self, pid):
            raise pickle.UnpicklingError(f"{path}: persistent ids are not part of a token-id file")

    with open(path, 'rb') as fp:
        data = _DataOnly(fp).load()
    if not isinstance(data, list) or not all(isinstance(row, list) for row in data):
        raise ValueError(f"{path}: expected a list of id lists")
    return data


def convertPickleDataset(srcPath, dstPath):
    """One-time conversion of a reference dataset directory into int32 shards.  -> the meta dict written."""
    files, meta = getAllMatchingDatasetFiles(srcPath)
    os.makedirs(dstPath, exist_ok=True)
    out, seq_len = {}, None
    for f in files:
        key = os.path.basename(f)
        if key not in meta:
            continue                                    # the reference warns and skips files without meta data
        rows = np.asarray(_read_id_lists(f), dtype=np.int64)
        if rows.ndim != 2 or (seq_len is not None and rows.shape[1] != seq_len):
            raise ValueError(f"{f}: expected equal-length id lists (seq_len {seq_len}), got shape {rows.shape}")
        if rows.size and (rows.min() < 0 or rows.max() > np.iinfo(np.int32).max):
            raise ValueError(f"{f}: token ids outside int32")
        seq_len = rows.shape[1]
        rows = rows[:meta[key]]
        name = os.path.splitext(key)[0] + ".ids"
        rows.astype('<i4').tofile(os.path.join(dstPath, name))
        out[name] = int(rows.shape[0])
    info = {"format": FORMAT, "seq_len": int(seq_len or 0), "files": out}
    with open(os.path.join(dstPath, "dataset.json"), "w") as fp:
        json.dump(info, fp, indent=1)
    return info


class _ListFileSource:
    def __init__(self, path, count):
        self.path, self.count, self._rows = path, count, None

    def rows(self, start, end):
        if self._rows is None or self._rows[0] != (start, end):
            self._rows = ((start, end), np.asarray(_read_id_lists(self.path)[start:end], dtype=np.int64))    # one file in memory at a time
        return self._rows[1]


class _ShardSource:
    def __init__(self, path, count, seq_len):
        self.path, self.count, self.seq_len = path, count, seq_len

    def rows(self, start, end):
        mm = np.memmap(self.path, dtype='<i4', mode='r', shape=(self.count, self.seq_len))
        return mm[start:end]                             # pages of other ranks' chunks are never touched


def _rank_chunk(n, world, rank):
    """PickleBasedDataloader.py:75-92."""
    base, rest = n // world, n % world
    start = rank * base + min(rank, rest)
    return start, start + base + (1 if rank < rest else 0)


class TokenStreamLoader:
    """Batches of token ids, (batchSize, seq_len) int64, as a cyclic stream over the (rank-local part of the) files.

    `device`: batches are staged through pinned host buffers and copied on a side stream `prefetch` batches ahead;
    `__next__` hands out a device tensor whose copy the current stream has been ordered after (no host synchronisation).
    device=None: CPU tensors."""

    def __init__(self, sources, batchSize, doShuffle=True, maxNumSamples=None, ddpWorldSize=None, ddpRank=None, seed=0, device=None,
                 prefetch=2):
        self.batchSize, self.doShuffle = int(batchSize), doShuffle
        spans, total = [], 0
        for src in sources:                              # :50-69: per-file sample counts, capped by maxNumSamples over the file list
            n = src.count
            if maxNumSamples is not None:
                n = min(n, maxNumSamples - total)
            spans.append((src, 0, n))
            total += n
            if maxNumSamples is not None and total >= maxNumSamples:
                break
        if ddpWorldSize is not None and ddpRank is not None:
            local = []
            for src, _, n in spans:
                s, e = _rank_chunk(n, ddpWorldSize, ddpRank)
                if e > s:
                    local.append((src, s, e))
            spans = local
        self.spans = spans
        self.numSamples = sum(e - s for _, s, e in spans)
        self.numBatches = int(self.numSamples / self.batchSize)
        if maxNumSamples is not None and self.batchSize > maxNumSamples:
            self.batchSize = maxNumSamples              # :43-45 (after numBatches was computed, as there)
        if self.batchSize < 1 or self.numSamples < self.batchSize:
            raise ValueError(f"TokenStreamLoader: {self.numSamples} samples cannot fill a batch of {self.batchSize}")
        self.rng = np.random.RandomState(seed)
        self._order, self._next_file, self._pending = list(range(len(spans))), 0, []
        self._device = torch.device(device) if device is not None else None
        self._stream, self._depth, self._ring, self._slot, self._queue = None, 0, [], 0, []
        if self._device is not None and self._device.type == "cuda":
            self._stream = torch.cuda.Stream(device=self._device)
            self._depth = max(1, int(prefetch))
        self.batchCounter = 0

    # -- the host-side stream -----------------------------------------------------------------------------------------------
    def _load_next_file(self):
        if self._next_file == 0 and self.doShuffle:
            self.rng.shuffle(self._order)
        src, s, e = self.spans[self._order[self._next_file]]
        self._next_file = (self._next_file + 1) % len(self.spans)
        rows = src.rows(s, e)
        if self.doShuffle:
            rows = rows[self.rng.permutation(rows.shape[0])]
        self._pending.append(rows)
        return rows.shape[0]

    def _host_batch(self):
        have = sum(r.shape[0] for r in self._pending)
        while have < self.batchSize:
            have += self._load_next_file()
        parts, need = [], self.batchSize
        while need:
            head = self._pending[0]
            take = min(need, head.shape[0])
            parts.append(head[:take])
            need -= take
            if take == head.shape[0]:
                self._pending.pop(0)
            else:
                self._pending[0] = head[take:]
        return parts

    # -- staging to the GPU -------------------------------------------------------------------------------------------------
    def _stage(self):
        parts = self._host_batch()
        if self._stream is None:
            out = torch.from_numpy(np.concatenate(parts).astype(np.int64))
            return out if self._device is None else out.to(self._device)
        S = parts[0].shape[1]
        n_buf = self._depth + 1                          # one more than can be in flight: a buffer is rewritten only after its copy
        i = self._slot % n_buf
        if i == len(self._ring):
            self._ring.append([torch.empty((self.batchSize, S), dtype=torch.int32).pin_memory(), None])
        slot = self._ring[i]
        self._slot += 1
        if slot[1] is not None:
            slot[1].synchronize()                        # the copy that last read this buffer (finished long ago in steady state)
        view, at = slot[0].numpy(), 0
        for p in parts:
            view[at:at + p.shape[0]] = p
            at += p.shape[0]
        with torch.cuda.stream(self._stream):
            dev = slot[0].to(self._device, non_blocking=True).long()
            done = torch.cuda.Event()
            done.record(self._stream)
        slot[1] = done
        return dev, done

    def __iter__(self):
        self.batchCounter = 0
        return self

    def __len__(self):
        return self.numBatches

    def __next__(self):
        self.batchCounter += 1
        if self.batchCounter > self.numBatches:
            raise StopIteration
        if self._stream is None:
            return self._stage()
        ahead = min(self._depth, self.numBatches - self.batchCounter + 1)         # never consume the stream past the epoch's end
        while len(self._queue) < ahead:
            self._queue.append(self._stage())
        dev, done = self._queue.pop(0)
        cur = torch.cuda.current_stream(self._device)
        cur.wait_event(done)
        dev.record_stream(cur)
        return dev


def loadPreTokenizedDataloader(batchSize, datasetPath, doShuffle=True, maxNumSamples=None, ddpWorldSize=None, ddpRank=None, verbose=True,
                               seed=0, device=None):
    """DataUtils.py:31-38.  A directory with a `dataset.json` of this build's shard format is memory-mapped; anything else is
    read as the reference's `*.pkl` + `*.json` layout."""
    shard_meta = os.path.join(datasetPath, "dataset.json")
    if os.path.exists(shard_meta):
        with open(shard_meta) as fp:
            info = json.load(fp)
        if info.get("format") != FORMAT:
            raise ValueError(f"{shard_meta}: unknown format {info.get('format')!r}")
        sources = [_ShardSource(os.path.join(datasetPath, name), n, info["seq_len"]) for name, n in info["files"].items()]
    else:
        files, meta = getAllMatchingDatasetFiles(datasetPath)
        sources = [_ListFileSource(f, meta[os.path.basename(f)]) for f in files if os.path.basename(f) in meta]
    return TokenStreamLoader(sources, batchSize, doShuffle=doShuffle, maxNumSamples=maxNumSamples, ddpWorldSize=ddpWorldSize, ddpRank=ddpRank,
                             seed=seed, device=device)
