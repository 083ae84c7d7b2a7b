"""The data-parallel train step with two ranks sharing ONE GPU (gloo: RCCL refuses two ranks on one device), on the real GPU
path: flat arenas, all-reduce of the SUM with 1/world folded into the Adam kernel, three HIP graphs per step with the
discriminator's optimizer step deferred behind the next generator forward (Training/SpawnedExperiment.py:47-82,165-191 is the
reference's per-GPU worker).

Checks: both ranks hold bit-identical parameters, and three implementations of the same two-rank step agree - eager with
torch's fused Adam and a pre-averaged all-reduce (the reference's arithmetic: DDP averages, Adam steps), eager with the arena
Adam kernel, and the graphed schedule.  Same gradients up to rounding; Adam's m / sqrt(v) turns the rounding noise of a
mathematically-zero gradient (the key bias of softmax attention) into steps of a few 1e-5, so parameters are compared at
rtol 1e-4 / atol 1e-4 after 5 Adam steps with lr 1e-3 (a wrong step would be off by ~5e-3).  (A single process on the full batch is not the
yardstick here: a different batch partition changes every summation order, and Adam amplifies that noise wherever a
gradient is close to zero; tests/test_dp_cpu.py makes that comparison with SGD.)"""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SPARSE = [2, 9, 17, 25, 33, 41, 50, 60]


def _run(rank, world, port, tmp, variant, batch=4):
    graphed = variant.endswith("graphed")
    sgd = variant.startswith("sgd")
    sys.path.insert(0, os.path.join(ROOT, "branch-gan_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      BRANCHGAN_DIST_BACKEND="gloo", BRANCHGAN_ARENA_ADAM="0" if variant == "torch-adam" else "1",
                      BRANCHGAN_DP_SHARDED="1" if variant.startswith("sharded") else "0")
    from branchgan import configs, gan, sampling, training
    from branchgan.graphs import GraphedTrainStep
    training.init_data_parallel()
    torch.cuda.set_device(0)
    torch.manual_seed(0)
    model = gan.GannSetup("gpt2-tiny", "gpt2-tiny", mleLossWeight=1, rlLossWeight=0 if variant == "sgd-mle" else 1, discValueHeadLossWeight=0.25,
                          device="cuda").cuda()
    rules = [sampling.SamplingRule(1, sampling.SamplingTechniques.TOP_K, {}) for _ in range(3)]
    conf = configs.SamplingConfig(rules, includeAllMultiStepSamplesForDisc=True, numSampleSequences=8, seed=77)
    conf.sparseIndex = torch.tensor(SPARSE, device="cuda")
    rl = configs.ReinforcementLearningConfig(labelMode=configs.ReinforcementLabelMode.ALTERATIONS_AND_INPUT_UNWEIGHTED,
                                             sectionWeightA=1, sectionWeightB=1, sectionWeightC=1.5, teacherForcing=True)
    if sgd:
        g_opt = training.OptimizerWrapper(model.generator, training.make_optimizer(model.generator.parameters(), 'sgd', 0.05))
        d_opt = training.OptimizerWrapper(model.discriminator, training.make_optimizer(model.discriminator.parameters(), 'sgd', 0.05))
    else:
        g_opt = training.OptimizerWrapper(model.generator, training.make_optimizer(model.generator.parameters(), 'adam', 1e-3, betas=(0.5, 0.999)))
        d_opt = training.OptimizerWrapper(model.discriminator, training.make_optimizer(model.discriminator.parameters(), 'adam', 1e-3))
    V = configs.LMConfig.from_any("gpt2-tiny").vocab_size
    data = torch.randint(0, V, (3, batch, 64), generator=torch.Generator().manual_seed(5)).cuda()
    lo = sum(training.worker_batch_size(batch, world, i) for i in range(rank))          # SpawnedExperiment.py:63-68
    mine = data[:, lo:lo + training.worker_batch_size(batch, world, rank)]
    if graphed:
        step = GraphedTrainStep(model, conf, g_opt, d_opt, mine[0], 9999, rl, None, warmup=3, restore_after_warmup=False)     # 3 eager steps on batch 0, kept
        assert step._defer_d == (world > 1)
        if variant == "sharded-regraphed":
            # a SECOND graphed step built mid-training with the default restore_after_warmup=True (what efficentTrainingLoop does at
            # every new tree depth): its snapshot needs whole Adam moments, which a sharded step leaves sliced
            step.step(mine[1], 3)
            step.check_nan()
            step = GraphedTrainStep(model, conf, g_opt, d_opt, mine[0], 9999, rl, None, warmup=2)
            step.step(mine[2], 4)
        else:
            for i in range(2):
                step.step(mine[1 + i], 3 + i)
        step.check_nan()
    elif sgd:                                          # ONE step: after it the sampled branch tokens may differ between the partitions
        training.train_step(model, mine[0], conf, g_opt, d_opt, 0, 9999, rl)
    else:
        for i in range(3):
            training.train_step(model, mine[0], conf, g_opt, d_opt, i, 9999, rl)
        for i in range(2):
            training.train_step(model, mine[1 + i], conf, g_opt, d_opt, 3 + i, 9999, rl)
    if variant != "torch-adam" and not sgd:
        assert g_opt.optimizer.grad_scale == d_opt.optimizer.grad_scale == 1.0 / world      # the mean is folded into the kernel
    torch.cuda.synchronize()
    if variant.startswith("sharded") and world > 1:
        lo_s, hi_s = g_opt.arena.shard_range()
        assert (hi_s - lo_s) * world == g_opt.arena.arena.numel() and g_opt.arena.shard is not None
        with pytest.raises(RuntimeError, match="gather_state"):
            g_opt.optimizer.state_dict()
        g_opt.optimizer.gather_state()
        d_opt.optimizer.gather_state()
        sd = g_opt.optimizer.state_dict()["state"]
        torch.save({f"m{k}": v["exp_avg"].detach().cpu() for k, v in sd.items()}, os.path.join(tmp, f"{variant}_moments_r{rank}.pt"))
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, os.path.join(tmp, f"{variant}_w{world}_b{batch}_r{rank}.pt"))
    if world > 1:
        torch.distributed.destroy_process_group()


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _two_ranks(tmp, variant, batch=4):
    mp.spawn(_run, args=(2, _port(), tmp, variant, batch), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp, f"{variant}_w2_b{batch}_r0.pt"), weights_only=True)
    r1 = torch.load(os.path.join(tmp, f"{variant}_w2_b{batch}_r1.pt"), weights_only=True)
    for k in r0:
        assert torch.equal(r0[k], r1[k]), (variant, k)
    return r0


def test_two_ranks_on_one_gpu():
    """+ BRANCHGAN_DP_SHARDED=1 (reduce-scatter, each rank steps its slice of the arena, all-gather of the parameters), eager
    and graphed: the same parameters as the all-reduce forms, and whole moments on every rank after gather_state()."""
    tmp = __import__("tempfile").mkdtemp()
    got = {}
    for variant in ("torch-adam", "arena-adam", "graphed", "sharded", "sharded-graphed", "sharded-regraphed"):
        got[variant] = _two_ranks(tmp, variant)
    for k, v in got["sharded-regraphed"].items():        # the warm-up steps of the second build were put back: nothing but steps 3, 4
        torch.testing.assert_close(v, got["sharded-graphed"][k], rtol=1e-5, atol=1e-6, msg=lambda m: f"regraphed {k}: {m}")
    m0 = torch.load(os.path.join(tmp, "sharded_moments_r0.pt"), weights_only=True)
    m1 = torch.load(os.path.join(tmp, "sharded_moments_r1.pt"), weights_only=True)
    assert len(m0) > 10 and all(torch.equal(m0[k], m1[k]) for k in m0) and any(bool(v.any()) for v in m0.values())
    for k, v in got["sharded"].items():                  # the element-wise kernel on a slice = the kernel on the whole arena
        assert torch.equal(v, got["arena-adam"][k]), k
    ref = got["torch-adam"]
    moved = 0.0
    for variant in ("arena-adam", "graphed", "sharded", "sharded-graphed", "sharded-regraphed"):
        for k, v in got[variant].items():
            torch.testing.assert_close(v, ref[k], rtol=1e-4, atol=1e-4, msg=lambda m: f"{variant} {k}: {m}")
    torch.manual_seed(0)
    sys.path.insert(0, os.path.join(ROOT, "branch-gan_amd"))
    from branchgan import gan
    init = gan.GannSetup("gpt2-tiny", "gpt2-tiny", mleLossWeight=1, rlLossWeight=1, discValueHeadLossWeight=0.25).state_dict()
    for k, v in ref.items():
        moved = max(moved, float((v - init[k]).abs().max()))
    assert moved > 1e-3          # the steps did train something


@pytest.mark.parametrize("variant", ["sgd-mle", "sgd"])
def test_two_rank_sgd_step_matches_one_process_on_the_full_batch(variant):
    """The real Branch-GAN step (gpt2-tiny G + D, depth 3, top-k) under SGD: two ranks with half the batch each against ONE process
    on the whole batch, after one step with lr 0.05 (a second step would already sample other branch tokens wherever a top-k choice
    is a near-tie).  The discriminator's losses and the generator's MLE loss are means over a fixed number of entries per row, so
    the mean of the two ranks' gradients IS the full-batch gradient: parameters equal to fp32 summation order.  The REINFORCE term
    is a mean over the entries its advantage mask keeps (ReinforceLoss.py:21-52) - a count that differs from rank to rank - so
    with it each rank normalises by its own count, exactly as the reference's DDP workers do (SpawnedExperiment.py:165-191), and
    the generator is NOT the full-batch one: checked with rlLossWeight 0 (whole model) and 1 (discriminator)."""
    tmp = __import__("tempfile").mkdtemp()
    two = _two_ranks(tmp, variant)
    mp.spawn(_run, args=(1, _port(), tmp, variant), nprocs=1, join=True)
    one = torch.load(os.path.join(tmp, f"{variant}_w1_b4_r0.pt"), weights_only=True)
    torch.manual_seed(0)
    sys.path.insert(0, os.path.join(ROOT, "branch-gan_amd"))
    from branchgan import gan
    init = gan.GannSetup("gpt2-tiny", "gpt2-tiny", mleLossWeight=1, rlLossWeight=1, discValueHeadLossWeight=0.25).state_dict()
    for part in ("generator", "discriminator"):
        assert max(float((one[k] - init[k]).abs().max()) for k in one if k.startswith(part)) > 1e-3      # the step trained both
    worst = {"generator": 0.0, "discriminator": 0.0}
    for k, v in two.items():
        part = k.split(".")[0]
        worst[part] = max(worst[part], float((v - one[k]).abs().max()))
        if part == "discriminator" or variant == "sgd-mle":
            torch.testing.assert_close(v, one[k], rtol=1e-4, atol=2e-5, msg=lambda m: f"{k}: {m}")
    print(f"{variant}: two ranks vs one process, worst |difference| {worst}")
    if variant == "sgd":
        assert 1e-5 < worst["generator"] < 5e-2          # per-rank normalisation of the masked REINFORCE mean: different, not wrong


def test_uneven_split_through_the_graphed_step():
    """B % W != 0 (SpawnedExperiment.py:63-68: the first ranks take one row more): rank 0 captures its graphs at 3 rows, rank 1 at
    2; the gradients are averaged with equal weight per RANK as DDP does.  Graphed = eager, rank 0 = rank 1."""
    tmp = __import__("tempfile").mkdtemp()
    eager = _two_ranks(tmp, "arena-adam", batch=5)
    graphed = _two_ranks(tmp, "graphed", batch=5)
    for k, v in graphed.items():
        torch.testing.assert_close(v, eager[k], rtol=1e-4, atol=1e-4, msg=lambda m: f"{k}: {m}")


@pytest.mark.parametrize("sharded", ["0", "1"])
def test_rccl_one_rank_rehearsal_of_the_multi_rank_schedule(sharded):
    """BRANCHGAN_REHEARSE_RCCL=1: process group on the nccl (= RCCL) backend, bucketed all-reduces on the side stream, host
    syncs after graph replays and the deferred discriminator step - with the ONE rank a one-GPU box allows.  Guards against
    the RCCL-side hang a 2-rank gloo rehearsal cannot show; the bench line must come out with finite losses.  sharded = 1:
    BRANCHGAN_DP_SHARDED's reduce_scatter_tensor / all_gather_into_tensor on RCCL, with world 1."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BRANCHGAN_REHEARSE_RCCL="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541",
               HSA_ENABLE_IPC_MODE_LEGACY="0", BRANCHGAN_DP_SHARDED=sharded)
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "c1", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=280)
    assert run.returncode == 0, run.stderr[-2000:]
    line = json.loads(run.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
    assert all(abs(v) < 1e4 for v in line["config"]["final_losses"].values())
