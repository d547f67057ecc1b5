"""GPU: the data-parallel (SyncBatchNorm) path of refign_amd/bn.py + csrc/bn.hip.
  * the split-phase kernels through the C ABI on two halves of a batch with the buffers ADDED between the passes (what
    an all-reduce does) == nn.BatchNorm2d(train) on the whole batch;
  * the same with two real processes sharing the one GPU of the box, exchanging over gloo (RCCL refuses two ranks on
    one device: "Duplicate GPU detected");
  * a 1-rank RCCL group: the student passes WITH the statistics exchanges inside are captured into hipGraphs and replay
    to the eager trajectory (what N > 1 runs with RFN_DDP_MODE=direct / direct3; exchanges forced on for the 1-rank group)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _reference(full, gfull, weight, bias, relu):
    ref = torch.nn.BatchNorm2d(full.shape[1]).to(full.device)
    with torch.no_grad():
        ref.weight.copy_(weight)
        ref.bias.copy_(bias)
    xr = full.float().requires_grad_(True)
    yr = ref(xr)
    yr = torch.relu(yr) if relu else yr
    yr.backward(gfull.float())
    return ref, xr, yr


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("relu", [True, False])
def test_split_phase_kernels_with_added_buffers_equal_full_batch(dev, dtype, relu):
    from refign_amd import bn as bnk
    torch.manual_seed(3)
    C, parts = 64, [(0, 3), (3, 4)]                  # unequal replica batches: the row counts travel in the buffer
    full = (torch.randn(4, 13, 11, C, device=dev) * 2 + 0.7).to(dtype)
    gfull = torch.randn(4, 13, 11, C, device=dev).to(dtype)
    weight = torch.linspace(0.5, 1.5, C, device=dev)
    bias = torch.linspace(-0.4, 0.4, C, device=dev)
    mods = [torch.nn.BatchNorm2d(C).to(dev) for _ in parts]
    xs = [full[a:b].contiguous() for a, b in parts]
    gs = [gfull[a:b].contiguous() for a, b in parts]
    sums = [torch.empty(2 * C + 1, device=dev, dtype=torch.float64) for _ in parts]
    for x, s in zip(xs, sums):
        bnk._stats_fwd(x, s)
    tot = sums[0] + sums[1]
    assert float(tot[2 * C]) == 4 * 13 * 11
    ys = [torch.empty_like(x) for x in xs]
    for x, y, m in zip(xs, ys, mods):
        bnk._apply_fwd(x, weight, bias, y, tot, m, relu)
    bs = [torch.empty(2, C, device=dev) for _ in parts]
    for x, g, b in zip(xs, gs, bs):
        bnk._stats_bwd(x, g, tot, weight, bias, b, 1e-5, relu)
    btot = bs[0] + bs[1]
    gxs = [torch.empty_like(x) for x in xs]
    for x, g, gx in zip(xs, gs, gxs):
        bnk._apply_bwd(x, g, tot, btot, weight, bias, gx, 1e-5, relu)
    ref, xr, yr = _reference(full.permute(0, 3, 1, 2), gfull.permute(0, 3, 1, 2), weight, bias, relu)
    y = torch.cat(ys).permute(0, 3, 1, 2).float()
    gx = torch.cat(gxs).permute(0, 3, 1, 2).float()
    e = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert float((y - yr).abs().max()) <= 4 * e * float(yr.abs().max()) + 1e-3
    assert float((gx - xr.grad).abs().max()) <= 0.02 * float(xr.grad.abs().max()) + 1e-4
    assert float((btot[1] - ref.weight.grad).abs().max()) <= 0.02 * float(ref.weight.grad.abs().max()) + 1e-3
    assert float((btot[0] - ref.bias.grad).abs().max()) <= 0.02 * float(ref.bias.grad.abs().max()) + 1e-3
    for m in mods:                                   # every replica tracks the statistics of the WHOLE batch
        assert torch.allclose(m.running_mean, ref.running_mean, atol=1e-4)
        assert torch.allclose(m.running_var, ref.running_var, rtol=1e-4, atol=1e-5)


def test_statistics_of_a_far_off_centre_channel(dev):
    """|mean| >> std over a million rows: the one-pass variance E[x^2] - mean^2 would lose every digit in fp32
    (mean^2 = 1e6, var = 1/16: 7 decimal digits are gone before the subtraction).  The statistics buffer is fp64
    (csrc/bn.hip header): batch and running statistics agree with an fp64 two-pass reference."""
    from refign_amd import bn as bnk
    torch.manual_seed(5)
    C, T = 16, 1 << 20
    x = (torch.randn(T, C, device=dev) * 0.25 + torch.linspace(-1000.0, 1000.0, C, device=dev)).to(torch.float16)
    x = x.view(4, 512, 512, C)
    xd = x.double().view(-1, C)
    mean, var = xd.mean(0), xd.var(0, unbiased=False)
    sums = torch.empty(2 * C + 1, device=dev, dtype=torch.float64)
    bnk._stats_fwd(x, sums)
    got_mean = sums[:C] / T
    got_var = sums[C:2 * C] / T - got_mean * got_mean
    assert float(sums[2 * C]) == T
    assert float((got_mean - mean).abs().max()) < 1e-6
    assert float(((got_var - var) / var).abs().max()) < 1e-6
    mod = torch.nn.BatchNorm2d(C).to(dev)
    y = torch.empty_like(x)
    bnk._apply_fwd(x, mod.weight.detach(), mod.bias.detach(), y, sums, mod, False)
    want = ((xd - mean) * torch.rsqrt(var + mod.eps)).view_as(x)
    assert float((y.double() - want).abs().max()) < 2.0 ** -9 * float(want.abs().max())
    assert torch.allclose(mod.running_var.double(), 0.9 + 0.1 * var * T / (T - 1), rtol=1e-5)
    again = torch.empty_like(sums)
    bnk._stats_fwd(x, again)
    assert torch.equal((again[:2 * C] / T).float(), (sums[:2 * C] / T).float())     # atomic order does not show in fp32


def _two_rank_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    from refign_amd import bn as bnk
    torch.manual_seed(11)
    C, per = 256, 2
    full = (torch.randn(world * per, C, 9, 14, device=dev) * 1.5 + 0.3).to(torch.bfloat16)
    gfull = torch.randn(world * per, C, 9, 14, device=dev).to(torch.bfloat16)
    weight = torch.linspace(0.5, 1.5, C, device=dev)
    bias = torch.linspace(-0.4, 0.4, C, device=dev)
    mod = torch.nn.SyncBatchNorm(C).to(dev).train()
    with torch.no_grad():
        mod.weight.copy_(weight)
        mod.bias.copy_(bias)
    assert bnk.usable(full, mod, torch.bfloat16) and bnk.sync_group(mod) is not None
    x = full[rank * per:(rank + 1) * per].contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = bnk.bn_act_train(x, mod, True, torch.bfloat16)
    y.backward(gfull[rank * per:(rank + 1) * per])
    ref, xr, yr = _reference(full, gfull, weight, bias, True)
    gw, gb = mod.weight.grad.clone(), mod.bias.grad.clone()
    dist.all_reduce(gw)
    dist.all_reduce(gb)
    sl = slice(rank * per, (rank + 1) * per)
    res = {"y": float((y.float() - yr[sl]).abs().max() / yr.abs().max()),
           "gx": float((x.grad.float() - xr.grad[sl]).abs().max() / xr.grad.abs().max()),
           "gw": float((gw - ref.weight.grad).abs().max() / ref.weight.grad.abs().max()),
           "gb": float((gb - ref.bias.grad).abs().max() / ref.bias.grad.abs().max()),
           "rm": float((mod.running_mean - ref.running_mean).abs().max()),
           "rv": float((mod.running_var - ref.running_var).abs().max())}
    torch.save(res, f"{out}/r{rank}.pt")
    dist.destroy_process_group()


def test_two_processes_on_one_gpu_exchange_statistics(dev, tmp_path):
    port, out = _free_port(), str(tmp_path)
    mp.spawn(_two_rank_worker, args=(2, port, out), nprocs=2, join=True)
    for rank in range(2):
        r = torch.load(f"{out}/r{rank}.pt")
        assert r["y"] < 0.02 and r["gx"] < 0.02 and r["gw"] < 0.02 and r["gb"] < 0.02, r
        assert r["rm"] < 1e-4 and r["rv"] < 1e-4, r


def _rccl_step_worker(rank, world, port, out):
    # a 1-rank RCCL group doing everything one rank of N does (RFN_DDP_REHEARSAL): the exchanges are really issued
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RFN_DDP_REHEARSAL="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import random
    import test_step_gpu as T
    from refign_amd import bn as bnk
    from refign_amd.trainer import Trainer
    dev = torch.device("cuda:0")
    calls = {"n": 0, "groups": set()}
    real_all_reduce = dist.all_reduce

    def counting_all_reduce(t, *a, **k):
        if "group" in k:                          # the statistics exchanges (the gradient buckets go without a group)
            calls["n"] += 1
            calls["groups"].add(id(k["group"]))
        return real_all_reduce(t, *a, **k)
    traj = {}
    # RFN_DDP_MODE (refign_amd/bn.py: ddp_mode): "torch" = the N > 1 default, every exchange through torch.distributed, eager
    # student passes; "direct" = exchanges as RCCL calls of our own inside the graphed passes, passes in stream order (here with
    # the gradient all-reduce of the finished ranges INSIDE the captured mixed pass on a communicator / stream of its own);
    # "direct3" = + the mixed pass next to the source pass on a third communicator (here with the two gradient buffers reduced
    # separately)
    for mode in ("torch-eager", "torch", "direct3", "direct"):
        # "torch" = the default as it runs since round 6: eager passes around the statistics exchanges, their backbones replayed
        # from graphs (graphs.GraphedSegment); "torch-eager": RFN_GRAPH_SEGMENTS=0, everything eager (the round-5 default)
        os.environ["RFN_GRAPH_SEGMENTS"] = "0" if mode == "torch-eager" else "1"
        mode = "torch" if mode == "torch-eager" else mode
        key = "torch-eager" if os.environ["RFN_GRAPH_SEGMENTS"] == "0" else mode
        os.environ["RFN_DDP_MODE"] = mode
        os.environ["RFN_DDP_DIRECT_REDUCE"] = "1" if mode != "torch" else "0"   # gradient reduce on our own comm
        os.environ["RFN_DDP_TWO_BUFFER"] = "1" if mode == "direct3" else "0"
        model = T.build(True, dev)
        trainer = Trainer(model, sync_batchnorm=True, fused_optimizer=False)
        n_sync = sum(isinstance(m, torch.nn.SyncBatchNorm) for m in model.modules())
        assert bnk.data_parallel() and trainer.data_parallel
        random.seed(5); np.random.seed(5); torch.manual_seed(5)
        rows = []
        bnk.dist.all_reduce = counting_all_reduce
        calls["n"], calls["groups"] = 0, set()
        try:
            for it in range(5):
                batch = T.make_batch(2, 128, 128, 64, dev)
                batch["image_src"] = batch["image_src"] + 0.1 * it
                with torch.autocast("cuda", dtype=torch.bfloat16):   # the BatchNorm kernels are the 16-bit path
                    trainer.step(batch, it)
                rows.append([float(model.logged[k]) for k in ("train_loss_src", "train_loss_featdist_src",
                                                              "train_loss_uda_trg")])
        finally:
            bnk.dist.all_reduce = real_all_reduce
        captured = all(g.captured() for n, g in model._graphs.items() if n in ("source_pass", "mixed_pass")) \
            if mode != "torch" else None
        bn = torch.cat([b.flatten().double() for n, b in model.head.named_buffers() if "running" in n]).cpu()
        if key == "torch":
            captured = all(model._graphs[n].captured() for n in ("source_backbone", "mixed_backbone"))
        traj[key] = (np.array(rows), float(sum(p.double().abs().sum() for p in model.live_parameters())), captured, n_sync,
                      calls["n"], len(calls["groups"]), model.__dict__.get("_mixed_concurrent_steps", 0), bn,
                      getattr(trainer.grads, "overlapped_fraction", 0.0),
                      trainer.grads._comm is not None)
    torch.save(traj, f"{out}/traj.pt")
    torch.cuda.synchronize()
    from refign_amd import rccl
    n_live = len(rccl._LIVE)
    rccl.destroy_all()
    assert n_live >= 5 and not rccl._LIVE
    dist.destroy_process_group()


def test_student_graphs_with_rccl_statistics_exchange_inside(dev, tmp_path):
    """What one rank of N > 1 runs, on a 1-rank RCCL group (RFN_DDP_REHEARSAL=1): 5 eager steps with every exchange through
    torch.distributed (RFN_DDP_MODE=torch, RFN_GRAPH_SEGMENTS=0) against the N > 1 default (the same exchanges, backbones replayed
    from graph segments) and against 5 steps with the student passes captured into hipGraphs
    WITH the SyncBatchNorm exchanges inside as RCCL calls of our own (direct: passes in stream order; direct3: a communicator
    per pass, mixed pass next to the source pass)."""
    port, out = _free_port(), str(tmp_path)
    mp.spawn(_rccl_step_worker, args=(1, port, out), nprocs=1, join=True)
    traj = torch.load(f"{out}/traj.pt", weights_only=False)
    e = traj["torch-eager"]
    # round 6, the N > 1 default: passes eager around the exchanges (the same torch.distributed calls, the same two groups), the
    # collective-free backbones of both passes replayed from graphs -- same trajectory
    t = traj["torch"]
    assert t[2], "the backbone segments of the student passes were not captured"
    assert t[4] == e[4] and t[5] == 2, (t[4:6], e[4:6])
    # ... and once both segments replay, the mixed pass runs on its own stream next to the source pass (its gradients in the second
    # flat buffer; torch.distributed keeps the two streams' collectives in the host's issue order, the same on every rank)
    assert t[6] >= 2, "the mixed pass did not run next to the source pass under torch-mode data parallelism"
    np.testing.assert_allclose(t[0], e[0], rtol=3e-2)
    assert abs(t[1] - e[1]) < 1e-4 * e[1]
    assert float((t[7] - e[7]).abs().max()) < 5e-2 * float(e[7].abs().max())
    assert e[3] > 0, "no SyncBatchNorm module in the model: nothing was exchanged"
    # eager: every exchange goes through dist.all_reduce each step, over two communicators (student, teacher)
    assert e[4] > 5 * 4 and e[5] == 2, e[4:6]
    assert e[6] == 0, "all-eager passes stay in stream order"
    # exchanges as direct RCCL calls (student passes AND teacher): nothing goes through torch's process group any more,
    # both passes captured, and the mixed pass runs next to the source pass once both replay
    # (bf16 passes with atomics in the weight-gradient kernels: the trajectories agree to rounding, not to the bit)
    d = traj["direct3"]
    assert d[2], "student passes were not captured with direct RCCL exchanges"
    assert d[5] == 0 and d[4] == 0, d[4:6]
    assert d[6] >= 2, "the mixed pass did not run next to the source pass"
    assert d[9] and d[8] > 0.85, f"only {d[8]:.2f} of the two gradient buffers was reduced before the tail"
    np.testing.assert_allclose(d[0], e[0], rtol=3e-2)
    assert abs(d[1] - e[1]) < 1e-4 * e[1]
    assert float((d[7] - e[7]).abs().max()) < 5e-2 * float(e[7].abs().max())
    # two communicators: captured, stream order, and most of the gradient buffer reduced from inside the replayed mixed pass
    f = traj["direct"]
    assert f[2] and f[6] == 0 and f[9], (f[2], f[6], f[9])
    assert f[8] > 0.8, f"only {f[8]:.2f} of the gradient buffer was reduced inside the captured backward pass"
    np.testing.assert_allclose(f[0], e[0], rtol=3e-2)
    assert abs(f[1] - e[1]) < 1e-4 * e[1]
    assert float((f[7] - e[7]).abs().max()) < 5e-2 * float(e[7].abs().max())


def _two_rank_segment_worker(rank, world, port, out):
    # two REAL ranks sharing the box's one GPU, exchanging over gloo (RCCL refuses two ranks on one device): the N > 1 default
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RFN_DDP_MODE="torch")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import random
    import test_step_gpu as T
    from refign_amd.trainer import Trainer
    dev = torch.device("cuda:0")
    res = {}
    for seg in ("1", "0"):
        os.environ["RFN_GRAPH_SEGMENTS"] = seg
        random.seed(5); np.random.seed(5); torch.manual_seed(5)
        model = T.build(True, dev)
        trainer = Trainer(model, sync_batchnorm=True, fused_optimizer=False)
        assert trainer.data_parallel and trainer.ddp_mode == "torch" and trainer.guard is not None
        random.seed(7 + rank); np.random.seed(7 + rank); torch.manual_seed(7 + rank)       # each rank its own pairs and draws
        rows = []
        for it in range(5):
            batch = T.make_batch(2, 128, 128, 64, dev)
            batch["image_src"] = batch["image_src"] + 0.1 * it + 0.05 * rank
            batch["image_trg"] = batch["image_trg"] - 0.03 * rank
            with torch.autocast("cuda", dtype=torch.bfloat16):
                trainer.step(batch, it)
            rows.append([float(model.logged[k]) for k in ("train_loss_src", "train_loss_featdist_src", "train_loss_uda_trg")])
        res[seg] = (np.array(rows), float(sum(p.double().abs().sum() for p in model.live_parameters())),
                    all(model._graphs[n].captured() for n in ("source_backbone", "mixed_backbone")),
                    model.__dict__.get("_mixed_concurrent_steps", 0))
        trainer.close()
    torch.save(res, f"{out}/seg{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


def test_backbone_segments_equal_eager_passes_on_two_ranks(dev, tmp_path):
    """Round 6, the N > 1 default with TWO real ranks (one GPU, gloo): 5 steps with the student backbones replayed from graph segments
    and the mixed pass next to the source pass (graphs.GraphedSegment; capture on the third step) against 5 all-eager steps
    (RFN_GRAPH_SEGMENTS=0), per rank: the same losses, the same parameters -- and the two ranks, fed different pairs, end with the SAME
    parameters (one mean gradient, SyncBatchNorm statistics of the global batch)."""
    port, out = _free_port(), str(tmp_path)
    mp.spawn(_two_rank_segment_worker, args=(2, port, out), nprocs=2, join=True)
    r = [torch.load(f"{out}/seg{k}.pt", weights_only=False) for k in range(2)]
    for k in range(2):
        seg, eag = r[k]["1"], r[k]["0"]
        assert seg[2] and not eag[2], "segments not captured / captured with RFN_GRAPH_SEGMENTS=0"
        assert seg[3] >= 2 and eag[3] == 0
        np.testing.assert_allclose(seg[0], eag[0], rtol=3e-2)
        assert abs(seg[1] - eag[1]) < 1e-4 * eag[1]
    assert abs(r[0]["1"][1] - r[1]["1"][1]) < 1e-6 * r[0]["1"][1], "the replicas drifted apart"
    assert float(np.abs(r[0]["1"][0] - r[1]["1"][0]).max()) > 1e-4, "the ranks saw the same data: the test would not see a missing exchange"
