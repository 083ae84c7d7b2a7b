"""Token-id datasets on the CPU: `branchgan.data.TokenStreamLoader` yields, batch for batch over two
epochs, what the reference's PickleBasedDataloader yielded with doShuffle=False for the same files, batch sizes and
(world, rank) splits - tests/golden/g8_dataloader.json, captured from the reference by tests/golden/make_goldens.py, which
also holds the small synthetic dataset so that the files can be rebuilt here.  Plus: the int32 shard conversion gives the
same stream, shuffling is a seeded permutation of the same samples, and a `*.pkl` that names any class or function is
refused without running anything."""
import json
import os
import pickle

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def g8():
    with open(os.path.join(GOLDEN, "g8_dataloader.json")) as f:
        return json.load(f)


@pytest.fixture()
def dataset(tmp_path, g8):
    src = tmp_path / "src"
    src.mkdir()
    for name, rows in g8["data"].items():
        with open(src / name, "wb") as fp:
            pickle.dump(rows, fp)
    with open(src / "meta.json", "w") as fp:
        json.dump(g8["files"], fp)
    return src


def _sources(kind, dataset, g8, tmp_path):
    from branchgan import data
    if kind == "pkl":
        return [data._ListFileSource(str(dataset / name), g8["files"][name]) for name in g8["file_order"]]
    info = data.convertPickleDataset(str(dataset), str(tmp_path / "shards"))
    assert info["format"] == data.FORMAT and info["seq_len"] == 5
    return [data._ShardSource(str(tmp_path / "shards" / (os.path.splitext(name)[0] + ".ids")), g8["files"][name], 5) for name in g8["file_order"]]


@pytest.mark.parametrize("kind", ["pkl", "shards"])
def test_stream_matches_reference_loader(kind, dataset, g8, tmp_path):
    from branchgan import data
    sources = _sources(kind, dataset, g8, tmp_path)
    for case in g8["cases"]:
        dl = data.TokenStreamLoader(sources, case["batch"], doShuffle=False, maxNumSamples=case["maxNumSamples"], ddpWorldSize=case["world"],
                                    ddpRank=case["rank"])
        assert (len(dl), dl.numSamples) == (case["numBatches"], case["numSamples"]), case
        if case["numBatches"] and not any(case["epochs"]):
            # Waived quirk: the reference sizes a loader chunk from the LAST file it loaded alone (PickleBasedDataloader.py:147-150);
            # when a rank's share of that file is smaller than the batch the chunk is empty and its epoch ends at once.  This
            # build keeps cutting batches from the stream: the rank's chunks of the files, in file order, cyclically.
            local = []
            for name in g8["file_order"]:
                s, e = data._rank_chunk(g8["files"][name], case["world"], case["rank"])
                local += g8["data"][name][s:e]
            flat = [r for _ in range(2) for b in dl for r in b.tolist()]
            assert len(flat) == 2 * case["numBatches"] * case["batch"]
            assert flat == [local[i % len(local)] for i in range(len(flat))]
            continue
        for want in case["epochs"]:                        # the second epoch continues the stream where the first stopped
            got = [b for b in dl]
            assert all(b.dtype == torch.int64 for b in got)
            assert [b.tolist() for b in got] == want, (kind, case["batch"], case["world"], case["rank"])


def test_directory_loader_and_shuffle(dataset, g8, tmp_path):
    from branchgan import data
    every = sorted(tuple(r) for rows in g8["data"].values() for r in rows)
    dl = data.loadPreTokenizedDataloader(5, str(dataset), doShuffle=False)
    assert len(dl) == 6 and sorted(tuple(r) for b in dl for r in b.tolist()) == every      # 30 samples, all of them, once
    data.convertPickleDataset(str(dataset), str(tmp_path / "sh"))
    a = [b.tolist() for b in data.loadPreTokenizedDataloader(5, str(tmp_path / "sh"), doShuffle=True, seed=4)]
    b = [b.tolist() for b in data.loadPreTokenizedDataloader(5, str(tmp_path / "sh"), doShuffle=True, seed=4)]
    c = [b.tolist() for b in data.loadPreTokenizedDataloader(5, str(tmp_path / "sh"), doShuffle=True, seed=5)]
    assert a == b and a != c
    assert sorted(tuple(r) for batch in a for r in batch) == every
    parts = [sorted(tuple(r) for batch in data.loadPreTokenizedDataloader(1, str(tmp_path / "sh"), doShuffle=False, ddpWorldSize=4, ddpRank=r)
                    for r in batch.tolist()) for r in range(4)]
    assert sorted(sum(parts, [])) == every and [len(p) for p in parts] == [9, 8, 7, 6]     # 10|7|13 -> (3,2,4) (3,2,3) (2,2,3) (2,1,3)
    with pytest.raises(ValueError, match="No meta file"):
        data.getAllMatchingDatasetFiles(str(tmp_path / "sh"), metaDataFileExtension=".yaml")
    with pytest.raises(ValueError, match="cannot fill"):
        data.loadPreTokenizedDataloader(64, str(tmp_path / "sh"))


class _Marker:
    ran = False

    def __reduce__(self):
        return (_mark, ())


def _mark():
    _Marker.ran = True
    return [[1, 2, 3]]


def test_dataset_file_that_names_code_is_refused(tmp_path):
    """A token-id file is lists of ints; anything that asks the loader to import or call something is rejected unexecuted."""
    from branchgan import data
    bad = tmp_path / "bad.pkl"
    with open(bad, "wb") as fp:
        pickle.dump(_Marker(), fp)
    with pytest.raises(pickle.UnpicklingError, match="refusing to load"):
        data._read_id_lists(str(bad))
    assert _Marker.ran is False
    arr = tmp_path / "array.pkl"
    with open(arr, "wb") as fp:
        pickle.dump(np.arange(6).reshape(2, 3), fp)        # a numpy array also names a class: convert such data to lists first
    with pytest.raises(pickle.UnpicklingError):
        data._read_id_lists(str(arr))
    flat = tmp_path / "flat.pkl"
    with open(flat, "wb") as fp:
        pickle.dump([1, 2, 3], fp)
    with pytest.raises(ValueError, match="list of id lists"):
        data._read_id_lists(str(flat))
