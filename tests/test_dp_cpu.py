"""Data-parallel path on the CPU with gloo, world_size 2: the flat gradient arena + bucketed all-reduce gives every rank the
same averaged gradients as a single process on the full batch, parameters stay in sync after optimizer steps, and parameters
without a gradient are left out of the arena (the reference needs find_unused_parameters=True for them)."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Tiny(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(6, 5)
        self.b = torch.nn.Linear(5, 1)
        self.unused = torch.nn.Linear(5, 3)     # like the generator's valueHead: never gets a gradient

    def forward(self, x):
        return self.b(torch.tanh(self.a(x))).squeeze(-1)


def _worker(rank, world, port, tmp):
    sys.path.insert(0, os.path.join(ROOT, "branch-gan_amd"))
    from branchgan import training
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = training.init_data_parallel("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    model = Tiny()
    data, target = torch.randn(8, 6), torch.randn(8)
    b = training.worker_batch_size(8, world, rank)
    lo = sum(training.worker_batch_size(8, world, i) for i in range(rank))
    opt = training.OptimizerWrapper(model, training.make_optimizer(model.parameters(), "sgd", 0.1), gradAccum=2,
                                    arena=training.GradArena(model, bucket_bytes=64))   # tiny buckets: several collectives
    for step in range(4):
        loss = ((model(data[lo:lo + b]) - target[lo:lo + b]) ** 2).mean()
        opt.performUpdateStep({"loss": loss}, step)
    pad8 = lambda n: (n + 7) & ~7                      # every parameter starts on a 32-byte boundary of the arena
    assert model.unused.weight.grad is None and opt.arena.arena.numel() == pad8(6 * 5) + pad8(5) + pad8(5) + pad8(1)
    assert opt.arena.offsets == [0, 32, 40, 48]
    torch.save({k: v.clone() for k, v in model.state_dict().items()}, os.path.join(tmp, f"rank{rank}.pt"))
    dist.destroy_process_group()


def _sharded_worker(rank, world, port, tmp):
    """BRANCHGAN_DP_SHARDED's collective half on the CPU: reduce-scatter of the padded arena = this rank's slice of the all-reduce."""
    sys.path.insert(0, os.path.join(ROOT, "branch-gan_amd"))
    from branchgan import training
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    training.init_data_parallel("gloo")
    torch.manual_seed(0)
    model = Tiny()
    data, target = torch.randn(8, 6), torch.randn(8)
    ((model(data[rank * 4:rank * 4 + 4]) - target[rank * 4:rank * 4 + 4]) ** 2).mean().backward()
    arena = training.GradArena(model, sharded=True)
    arena._build()
    assert arena.arena.numel() % (8 * world) == 0 and arena.sharded_active()
    want = arena.arena.clone()
    dist.all_reduce(want)
    arena.reduce_async(average=False)
    arena.wait()
    lo, hi = arena.shard_range()
    assert (lo, hi) == (rank * arena.arena.numel() // world, (rank + 1) * arena.arena.numel() // world)
    assert torch.equal(arena.shard, want[lo:hi]) and bool(arena.shard.any() or rank == 1)
    full = torch.zeros_like(want)
    training.all_gather_slices(full, arena.shard)
    assert torch.equal(full, want)
    try:
        training.OptimizerWrapper(model, training.make_optimizer(model.parameters(), "sgd", 0.1), arena=arena)
        raise AssertionError("a whole-parameter optimizer must be refused under sharding")
    except ValueError as e:
        assert "BRANCHGAN_DP_SHARDED" in str(e)
    dist.destroy_process_group()


def test_sharded_reduce_scatter_and_all_gather_on_gloo(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_sharded_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)


def test_two_rank_gloo_matches_single_process(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    sd0 = torch.load(tmp_path / "rank0.pt", weights_only=True)
    sd1 = torch.load(tmp_path / "rank1.pt", weights_only=True)
    for k in sd0:
        assert torch.equal(sd0[k], sd1[k]), k
    # single process, full batch, same schedule (backward every micro-step, step when step % 2 == 0: quirk 11)
    torch.manual_seed(0)
    model = Tiny()
    data, target = torch.randn(8, 6), torch.randn(8)
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    for step in range(4):
        ((model(data) - target) ** 2).mean().backward()
        if step % 2 == 0:
            opt.step()
            opt.zero_grad()
    for k, v in model.state_dict().items():
        torch.testing.assert_close(sd0[k], v, atol=1e-6, rtol=1e-5)
