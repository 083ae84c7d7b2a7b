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


def _run(rank, world, port, tmp, variant):
    graphed = variant == "graphed"
    sys.path.insert(0, os.path.join(ROOT, "branch-gan_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      BRANCHGAN_DIST_BACKEND="gloo", BRANCHGAN_ARENA_ADAM="0" if variant == "torch-adam" else "1")
    from branchgan import configs, gan, sampling, training
    from branchgan.graphs import GraphedTrainStep
    training.init_data_parallel()
    torch.cuda.set_device(0)
    torch.manual_seed(0)
    model = gan.GannSetup("gpt2-tiny", "gpt2-tiny", mleLossWeight=1, rlLossWeight=1, discValueHeadLossWeight=0.25, device="cuda").cuda()
    rules = [sampling.SamplingRule(1, sampling.SamplingTechniques.TOP_K, {}) for _ in range(3)]
    conf = configs.SamplingConfig(rules, includeAllMultiStepSamplesForDisc=True, numSampleSequences=8, seed=77)
    conf.sparseIndex = torch.tensor(SPARSE, device="cuda")
    rl = configs.ReinforcementLearningConfig(labelMode=configs.ReinforcementLabelMode.ALTERATIONS_AND_INPUT_UNWEIGHTED,
                                             sectionWeightA=1, sectionWeightB=1, sectionWeightC=1.5, teacherForcing=True)
    g_opt = training.OptimizerWrapper(model.generator, training.make_optimizer(model.generator.parameters(), 'adam', 1e-3, betas=(0.5, 0.999)))
    d_opt = training.OptimizerWrapper(model.discriminator, training.make_optimizer(model.discriminator.parameters(), 'adam', 1e-3))
    V = configs.LMConfig.from_any("gpt2-tiny").vocab_size
    data = torch.randint(0, V, (3, 4, 64), generator=torch.Generator().manual_seed(5)).cuda()
    b = 4 // world
    mine = data[:, rank * b:(rank + 1) * b]
    if graphed:
        step = GraphedTrainStep(model, conf, g_opt, d_opt, mine[0], 9999, rl, None, warmup=3, restore_after_warmup=False)     # 3 eager steps on batch 0, kept
        assert step._defer_d == (world > 1)
        for i in range(2):
            step.step(mine[1 + i], 3 + i)
        step.check_nan()
    else:
        for i in range(3):
            training.train_step(model, mine[0], conf, g_opt, d_opt, i, 9999, rl)
        for i in range(2):
            training.train_step(model, mine[1 + i], conf, g_opt, d_opt, 3 + i, 9999, rl)
    if variant != "torch-adam":
        assert g_opt.optimizer.grad_scale == d_opt.optimizer.grad_scale == 1.0 / world      # the mean is folded into the kernel
    torch.cuda.synchronize()
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, os.path.join(tmp, f"{variant}_r{rank}.pt"))
    if world > 1:
        torch.distributed.destroy_process_group()


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_on_one_gpu():
    tmp = __import__("tempfile").mkdtemp()
    got = {}
    for variant in ("torch-adam", "arena-adam", "graphed"):
        mp.spawn(_run, args=(2, _port(), tmp, variant), nprocs=2, join=True)
        r0 = torch.load(os.path.join(tmp, f"{variant}_r0.pt"), weights_only=True)
        r1 = torch.load(os.path.join(tmp, f"{variant}_r1.pt"), weights_only=True)
        for k in r0:
            assert torch.equal(r0[k], r1[k]), (variant, k)
        got[variant] = r0
    ref = got["torch-adam"]
    moved = 0.0
    for variant in ("arena-adam", "graphed"):
        for k, v in got[variant].items():
            torch.testing.assert_close(v, ref[k], rtol=1e-4, atol=1e-4, msg=lambda m: f"{variant} {k}: {m}")
    torch.manual_seed(0)
    sys.path.insert(0, os.path.join(ROOT, "branch-gan_amd"))
    from branchgan import gan
    init = gan.GannSetup("gpt2-tiny", "gpt2-tiny", mleLossWeight=1, rlLossWeight=1, discValueHeadLossWeight=0.25).state_dict()
    for k, v in ref.items():
        moved = max(moved, float((v - init[k]).abs().max()))
    assert moved > 1e-3          # the steps did train something


def test_rccl_one_rank_rehearsal_of_the_multi_rank_schedule():
    """BRANCHGAN_REHEARSE_RCCL=1: process group on the nccl (= RCCL) backend, bucketed all-reduces on the side stream, host
    syncs after graph replays and the deferred discriminator step - with the ONE rank a one-GPU box allows.  Guards against
    the RCCL-side hang a 2-rank gloo rehearsal cannot show; the bench line must come out with finite losses."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BRANCHGAN_REHEARSE_RCCL="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "c1", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=280)
    assert run.returncode == 0, run.stderr[-2000:]
    line = json.loads(run.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
    assert all(abs(v) < 1e4 for v in line["config"]["final_losses"].values())
