"""CPU, world_size 2 over gloo: the data-parallel plumbing of refign_amd/trainer.py -- every p.grad is a view into one
flat buffer, one bucketed all-reduce per step gives the mean over ranks, parameters are broadcast from rank 0, and a
2-rank step equals the 1-rank step on the concatenated batch for a batch-mean loss."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Tiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.backbone = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8), nn.ReLU())
        self.head = nn.Conv2d(8, 4, 1)

    def forward(self, x):
        return self.head(self.backbone(x))


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from refign_amd.trainer import FlatGradBuffer
    torch.manual_seed(100 + rank)                 # different init per rank on purpose
    model = Tiny()
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src=0)
    model = nn.SyncBatchNorm.convert_sync_batchnorm(model) if False else model   # gloo SyncBN is GPU-only; BN local
    model.eval()                                   # eval BN => loss is a pure per-sample mean
    grads = FlatGradBuffer(model.parameters())
    g = torch.Generator().manual_seed(7)
    x = torch.randn(4, 3, 8, 8, generator=g)
    y = torch.randn(4, 4, 8, 8, generator=g)
    shard = slice(rank * 2, rank * 2 + 2)
    for _ in range(3):                              # three backward passes accumulate into the flat buffer
        ((model(x[shard]) - y[shard]) ** 2).mean().backward()
    assert all(p.grad.data_ptr() >= grads.flat.data_ptr() for p in grads.params)   # still views
    grads.all_reduce_mean(bucket_mb=0.0001)         # tiny buckets => many async all-reduces
    if rank == 0:
        views = torch.cat([p.grad.flatten() for p in grads.params])     # the buffer minus its alignment padding
        assert grads.flat.numel() % grads.ALIGN == 0 and grads.flat.numel() >= views.numel()
        assert torch.equal(grads.flat.double().sum(), views.double().sum())             # padding stays zero
        assert all((p.grad.data_ptr() - grads.flat.data_ptr()) % (4 * grads.ALIGN) == 0 for p in grads.params)
        torch.save({"flat": views.clone(), "w": model.head.weight.detach().clone()}, out)
    dist.destroy_process_group()


def test_flat_gradient_allreduce_equals_full_batch(tmp_path):
    port, out = _free_port(), str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    torch.manual_seed(100)
    model = Tiny().eval()
    assert torch.equal(model.head.weight, got["w"])          # rank-0 init was broadcast
    g = torch.Generator().manual_seed(7)
    x = torch.randn(4, 3, 8, 8, generator=g)
    y = torch.randn(4, 4, 8, 8, generator=g)
    for _ in range(3):
        ((model(x) - y) ** 2).mean().backward()
    want = torch.cat([p.grad.flatten() for p in model.parameters()])
    assert torch.allclose(got["flat"], want, rtol=1e-5, atol=1e-6)


def test_lr_schedule_matches_reference_formula():
    from refign_amd.trainer import LinearWarmupPolynomialLR
    p = nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=6e-5)
    sch = LinearWarmupPolynomialLR(opt, max_steps=40000, warmup_iters=1500, warmup_ratio=1e-6, power=1.0)
    lrs = []
    for _ in range(3000):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
    assert abs(lrs[0] - 6e-5 * 1e-6) < 1e-15                       # helpers/lr_scheduler.py:47-50 at t=0
    assert abs(lrs[750] - 6e-5 * (1 - 0.5 * (1 - 1e-6))) < 1e-12
    assert abs(lrs[1500] - 6e-5) < 1e-12
    assert abs(lrs[2500] - 6e-5 * (1 - 1000 / 38500)) < 1e-12       # :54-56, power 1


# ----------------------------------------------------------------------------------------------------------------
# whole Trainer.step on 2 ranks (gloo, CPU): the UDA model with a tiny MiT-b0, align/refine served by the CPU oracle
# (test infrastructure) -- exercises parameter broadcast, the three backward passes into the flat buffer, the single
# all-reduce, optimiser + scheduler + EMA on every rank, and checks the replicas stay bit-identical.
# ----------------------------------------------------------------------------------------------------------------
def _step_worker(rank, world, port, out):
    import os
    import random
    import sys
    import numpy as np
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "oracle"))
    import cpu_align
    from refign_amd.align import VGG, UAWarpCHead
    from refign_amd.seg import DAFormerHead, MixVisionTransformer, PixelWeightedCrossEntropyLoss
    from refign_amd.trainer import Trainer
    from refign_amd.uda import DomainAdaptationSegmentationModel
    dims = [32, 64, 160, 256]
    torch.manual_seed(500 + rank)                     # different init per rank: the trainer must broadcast rank 0's
    model = DomainAdaptationSegmentationModel(
        {"class_path": "torch.optim.AdamW", "init_args": {"lr": 1e-3, "weight_decay": 0.01}},
        {"class_path": "helpers.lr_scheduler.LinearWarmupPolynomialLR",
         "init_args": {"warmup_iters": 2, "warmup_ratio": 0.1, "power": 1.0, "max_steps": 10}},
        backbone=MixVisionTransformer("mit_b0", drop_path_rate=0.0),
        head=DAFormerHead(dims, [0, 1, 2, 3], 19, 'multiple_select', dropout_ratio=0.0),
        loss=PixelWeightedCrossEntropyLoss(),
        alignment_backbone=VGG('vgg16', out_indices=[2, 3, 4]),
        alignment_head=UAWarpCHead(in_index=[0, 1], input_transform='multiple_select', estimate_uncertainty=True),
        backbone_lr_factor=0.1, use_refign=True, adapt_to_ref=False, enable_fdist=True, color_jitter_p=1.0,
        blur=False).train()
    _, corr_fn = cpu_align._corr_fn_default()
    model.align = lambda lr, ir, it: cpu_align.align(model.alignment_backbone, model.alignment_head, lr, ir, it, corr_fn)
    model.refine = lambda lt, lr, m, c: cpu_align.refine(lt, lr, m, c, gamma=model.gamma)
    trainer = Trainer(model, sync_batchnorm=False, bucket_mb=1, fused_optimizer=False)
    g = torch.Generator().manual_seed(1000 + rank)    # each rank its own shard of the global batch
    H, W = 64, 64
    lbl = torch.randint(0, 19, (1, 2, 2), generator=g).repeat_interleave(32, 1).repeat_interleave(32, 2)
    batch = {"image_src": torch.randn(1, 3, H, W, generator=g), "semantic_src": lbl,
             "image_trg": torch.randn(1, 3, H, W, generator=g), "image_ref": torch.randn(1, 3, H, W, generator=g)}
    random.seed(3); np.random.seed(3)
    for _ in range(2):
        trainer.step(batch)
    live = torch.cat([p.detach().flatten() for p in model.live_parameters()])
    ema = torch.cat([p.detach().flatten() for p in model.ema_parameters()])
    gathered = [torch.zeros_like(live) for _ in range(world)]
    dist.all_gather(gathered, live)
    gathered_ema = [torch.zeros_like(ema) for _ in range(world)]
    dist.all_gather(gathered_ema, ema)
    overlapped = trainer.grads.overlapped_elements / trainer.grads.flat.numel()
    if rank == 0:
        torch.save({"overlapped": overlapped, "live": live.clone(), "same_live": bool(torch.equal(gathered[0], gathered[1])),
                    "same_ema": bool(torch.equal(gathered_ema[0], gathered_ema[1])),
                    "finite": bool(torch.isfinite(live).all()), "step": model.global_step,
                    "lr": trainer.optimizer.param_groups[0]["lr"]}, out)
    dist.destroy_process_group()


def test_two_rank_training_step_keeps_replicas_identical(tmp_path):
    port, out = _free_port(), str(tmp_path / "step.pt")
    mp.spawn(_step_worker, args=(2, port, out), nprocs=2, join=True)
    r = torch.load(out)
    assert r["same_live"] and r["same_ema"] and r["finite"] and r["step"] == 2
    # the ranges of the decode head and MiT stages 4..2 went on the wire from inside the last backward pass
    assert r["overlapped"] > 0.8, r["overlapped"]
    # ... and releasing them early changes nothing: the same run with the whole buffer reduced after the backward pass
    os.environ["RFN_DDP_OVERLAP"] = "0"
    try:
        port2, out2 = _free_port(), str(tmp_path / "step_noovl.pt")
        mp.spawn(_step_worker, args=(2, port2, out2), nprocs=2, join=True)
    finally:
        del os.environ["RFN_DDP_OVERLAP"]
    r2 = torch.load(out2)
    assert r2["overlapped"] == 0.0 and torch.equal(r["live"], r2["live"])
    assert abs(r["lr"] - 1e-3) < 1e-9       # warm-up of 2 iterations finished (LinearWarmupPolynomialLR)


# ---- SyncBatchNorm semantics of refign_amd/bn.py (statistics exchange between the kernel passes) ------------------------
def _syncbn_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import bn_standin
    from refign_amd import bn as bnk
    bn_standin.install(bnk)                         # CPU restatements of the four kernel passes (tests/bn_standin.py)
    torch.manual_seed(7)
    C, per = 16, 3
    full = torch.randn(world * per, C, 5, 6, dtype=torch.float64) * 2 + 0.5
    gfull = torch.randn(world * per, C, 5, 6, dtype=torch.float64)
    results = {}
    for relu in (False, True):
        for teacher in (False, True):               # teacher: train-mode statistics under no_grad
            ref = nn.BatchNorm2d(C).double().train()
            with torch.no_grad():
                ref.weight.copy_(torch.linspace(0.5, 1.5, C))
                ref.bias.copy_(torch.linspace(-0.3, 0.3, C))
            mod = nn.SyncBatchNorm(C).double().train()
            mod.load_state_dict(ref.state_dict())
            group = dist.new_group() if teacher else None
            mod.process_group = group
            assert bnk.sync_group(mod) is not None
            x = full[rank * per:(rank + 1) * per].clone().requires_grad_(not teacher)
            xr = full.clone().requires_grad_(not teacher)
            with torch.set_grad_enabled(not teacher):
                y = bnk._BNActTrain.apply(x.permute(0, 2, 3, 1).contiguous(), mod.weight, mod.bias, mod, relu,
                                          bnk.sync_group(mod)).permute(0, 3, 1, 2)
                yr = ref(xr)
                yr = torch.relu(yr) if relu else yr
            err = {"y": (y - yr[rank * per:(rank + 1) * per]).abs().max().item(),
                   "rm": (mod.running_mean - ref.running_mean).abs().max().item(),
                   "rv": (mod.running_var - ref.running_var).abs().max().item(),
                   "nbt": int(mod.num_batches_tracked)}
            if not teacher:
                y.backward(gfull[rank * per:(rank + 1) * per])
                yr.backward(gfull)
                err["gx"] = (x.grad - xr.grad[rank * per:(rank + 1) * per]).abs().max().item()
                # affine gradients are LOCAL sums; their sum over ranks is the full-batch gradient
                gw, gb = mod.weight.grad.clone(), mod.bias.grad.clone()
                dist.all_reduce(gw)
                dist.all_reduce(gb)
                err["gw"] = (gw - ref.weight.grad).abs().max().item()
                err["gb"] = (gb - ref.bias.grad).abs().max().item()
            results[(relu, teacher)] = err
    torch.save(results, f"{out}/syncbn_{rank}.pt")
    dist.destroy_process_group()


def test_sync_batchnorm_statistics_exchange_equals_full_batch(tmp_path):
    """2 ranks over gloo, each with its part of a batch: the SyncBatchNorm path of refign_amd/bn.py (one all-reduce of
    (sum, sum^2, rows) between the statistics and the apply pass, one of the backward sums; kernel passes replaced by
    the torch restatements of tests/bn_standin.py) == nn.BatchNorm2d on the whole batch: outputs, running statistics
    (unbiased variance over ALL rows), input gradients, and affine gradients summed over the ranks -- for the student
    (autograd) and for the EMA teacher's train-mode BatchNorms under no_grad on a process group of their own."""
    port, out = _free_port(), str(tmp_path)
    mp.spawn(_syncbn_worker, args=(2, port, out), nprocs=2, join=True)
    for rank in range(2):
        res = torch.load(f"{out}/syncbn_{rank}.pt")
        for key, err in res.items():
            assert err.pop("nbt") == 1
            for k, v in err.items():
                assert v < 2e-5, (rank, key, k, v)          # the exchanged statistics are fp32, as in the kernels


class _StandInComm:
    """What refign_amd/rccl.DirectComm is to bn.py -- `all_reduce_(t)`: in-place sum over the ranks on the caller's
    stream -- over a gloo group, counting its calls."""

    def __init__(self, group):
        self.group, self.calls = group, 0

    def all_reduce_(self, t):
        self.calls += 1
        dist.all_reduce(t, group=self.group)
        return t


def _direct_routing_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import bn_standin
    from refign_amd import bn as bnk
    bn_standin.install(bnk)
    torch.manual_seed(11)
    C, per = 8, 2
    full = torch.randn(world * per, C, 4, 5, dtype=torch.float64) + 0.25
    gfull = torch.randn(world * per, C, 4, 5, dtype=torch.float64)
    main, mixed, teacher_c = (_StandInComm(dist.new_group()) for _ in range(3))
    teacher_group = dist.new_group()

    def run(mod, grad):
        ref = nn.BatchNorm2d(C).double().train()
        x = full[rank * per:(rank + 1) * per].clone().requires_grad_(grad)
        xr = full.clone().requires_grad_(grad)
        with torch.set_grad_enabled(grad):
            y = bnk._BNActTrain.apply(x.permute(0, 2, 3, 1).contiguous(), mod.weight, mod.bias, mod, True,
                                      bnk.sync_group(mod)).permute(0, 3, 1, 2)
            yr = torch.relu(ref(xr))
        err = (y - yr[rank * per:(rank + 1) * per]).abs().max().item()
        if grad:
            y.backward(gfull[rank * per:(rank + 1) * per])
            yr.backward(gfull)
            err = max(err, (x.grad - xr.grad[rank * per:(rank + 1) * per]).abs().max().item())
        return err

    student = nn.SyncBatchNorm(C).double().train()
    teacher = nn.SyncBatchNorm(C).double().train()
    teacher.process_group = teacher_group
    res = {}
    # 1. no direct communicators: everything through torch.distributed
    bnk._DIRECT["default"] = None
    res["torch"] = (run(student, True), main.calls + mixed.calls + teacher_c.calls)
    # 2. the pass on the main stream: forward + backward exchange through the default communicator
    bnk._DIRECT["default"] = main
    res["main"] = (run(student, True), main.calls, mixed.calls)
    # 3. a pass captured / run inside direct_comm(): its own communicator, the default one untouched
    with bnk.direct_comm(mixed):
        res["mixed"] = (run(student, True), main.calls, mixed.calls)
    res["after"] = (run(student, False), main.calls, mixed.calls)
    # 4. the teacher: a module with a group of its own stays on torch.distributed unless it carries a communicator,
    #    and then it uses that one whatever the context says
    res["teacher_torch"] = (run(teacher, False), main.calls, mixed.calls, teacher_c.calls)
    teacher._rfn_direct = teacher_c
    with bnk.direct_comm(mixed):
        res["teacher_direct"] = (run(teacher, False), main.calls, mixed.calls, teacher_c.calls)
    bnk._DIRECT["default"] = None
    torch.save(res, f"{out}/routing_{rank}.pt")
    dist.destroy_process_group()


def test_statistics_exchange_routing_over_direct_communicators(tmp_path):
    """Which communicator a SyncBatchNorm exchange goes through (refign_amd/bn.py: _exchange_comm, direct_comm) -- 2 ranks
    over gloo with stand-ins for rccl.DirectComm: the student's exchanges use the default communicator (forward AND
    backward), a pass inside direct_comm() its own, the teacher's modules theirs regardless of the context, and without
    any of them torch.distributed; every variant == nn.BatchNorm2d on the whole batch."""
    port, out = _free_port(), str(tmp_path)
    mp.spawn(_direct_routing_worker, args=(2, port, out), nprocs=2, join=True)
    for rank in range(2):
        r = torch.load(f"{out}/routing_{rank}.pt")
        for k, v in r.items():
            assert v[0] < 2e-5, (rank, k, v)
        assert r["torch"][1] == 0
        assert r["main"][1:] == (2, 0)                       # forward + backward
        assert r["mixed"][1:] == (2, 2)
        assert r["after"][1:] == (3, 2)                      # no-grad forward: one exchange, default communicator again
        assert r["teacher_torch"][1:] == (3, 2, 0)
        assert r["teacher_direct"][1:] == (3, 2, 1)


def _graph_reduce_worker(rank, world, port, out):
    """The bookkeeping of a gradient reduce that lives INSIDE the captured last backward pass (FlatGradBuffer.use_direct /
    on_ready(capturing) / end_capture / replayed), with a stand-in communicator over gloo: a range reduced "by the graph" must
    not be reduced again by the tail, a range the graph does not hold must be, step after step."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from refign_amd.trainer import FlatGradBuffer
    ps = [nn.Parameter(torch.zeros(n)) for n in (1000, 70001, 333, 4099)]
    buf = FlatGradBuffer(ps, [("head", ps[:1]), ("stage3", ps[1:2]), ("stage1", ps[2:])], bucket_mb=0.01)
    comm = _StandInComm(None)
    buf.use_direct(comm, None)
    res = []

    def fill(step):
        buf.zero()
        for i, p in enumerate(ps):
            p.grad.fill_(float((rank + 1) * (i + 1) + step))

    # "capture": the callback fires for two of the three groups while the pass is being recorded
    fill(0)
    buf.on_ready("head", capturing=True)
    buf.on_ready("stage3", capturing=True)
    ranges = buf.end_capture()
    assert len(ranges) == 2 and buf._released == [] and buf.captured_ranges == ranges
    n_capture = comm.calls
    buf.replayed(ranges)
    buf.all_reduce_mean()
    res.append([float(p.grad.mean()) for p in ps])
    n_tail = comm.calls - n_capture
    # "replays": the graph's reduces are simulated by calling them again; the tail must only add the third group
    for step in (1, 2):
        fill(step)
        for _, a, b in ranges:
            comm.all_reduce_(buf.flat[a:b])
        buf.replayed(ranges)
        before = comm.calls
        buf.all_reduce_mean()
        assert comm.calls - before == n_tail
        res.append([float(p.grad.mean()) for p in ps])
        assert buf.overlapped_elements == sum(b - a for _, a, b in ranges)
    # a step whose graph did NOT replay (eager fallback): everything through the tail
    fill(3)
    buf.all_reduce_mean()
    res.append([float(p.grad.mean()) for p in ps])
    if rank == 0:
        torch.save(res, out)
    dist.destroy_process_group()


def test_gradient_reduce_inside_a_captured_pass_is_not_repeated_by_the_tail(tmp_path):
    port, out = _free_port(), str(tmp_path / "r.pt")
    mp.spawn(_graph_reduce_worker, args=(2, port, out), nprocs=2, join=True)
    res = torch.load(out)
    for step, row in enumerate(res):
        for i, v in enumerate(row):
            want = (1.5 * (i + 1)) + step                  # mean over ranks of (rank + 1)(i + 1) + step
            assert abs(v - want) < 1e-5, (step, i, v, want)


def _two_buffer_worker(rank, world, port, out):
    """Two buffers reduced separately (the mixed pass next to the source pass under data parallelism): the first whole and
    early, the second range by range from "inside its graph", the rest at the tail, then added -- equals the mean of the sum."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from refign_amd.trainer import FlatGradBuffer
    ps = [nn.Parameter(torch.zeros(n)) for n in (1000, 70001, 333)]
    buf = FlatGradBuffer(ps, [("head", ps[:1]), ("stage3", ps[1:2]), ("stage1", ps[2:])], bucket_mb=0.01)
    c1, c2 = _StandInComm(None), _StandInComm(None)
    buf.use_direct(c1, None, c2)
    rows = []
    for step in range(3):
        buf.zero()
        for i, p in enumerate(ps):
            p.grad.fill_(float((rank + 1) * (i + 1)))                  # "source pass"
        assert buf.reduce_first_now()
        with buf.into_second():
            for i, p in enumerate(ps):
                p.grad.fill_(float(10 * (rank + 1) + step))            # "mixed pass"
            if step == 0:                                              # the capture step: recorded, handed back at replay
                buf.on_ready("head", capturing=True)
                buf.on_ready("stage3", capturing=True)
                ranges = buf.end_capture()
                assert all(r[0] == 1 for r in ranges) and len(ranges) == 2
            else:
                for _, a, b in ranges:
                    c2.all_reduce_(buf.flat2[a:b])
        buf.replayed(ranges)
        n1, n2 = c1.calls, c2.calls
        buf.all_reduce_mean()
        assert c1.calls == n1 and c2.calls > n2                        # first buffer done; only the second one's remainder
        rows.append([float(p.grad.mean()) for p in ps])
    if rank == 0:
        torch.save(rows, out)
    dist.destroy_process_group()


def test_two_gradient_buffers_reduced_separately_and_added(tmp_path):
    port, out = _free_port(), str(tmp_path / "r.pt")
    mp.spawn(_two_buffer_worker, args=(2, port, out), nprocs=2, join=True)
    for step, row in enumerate(torch.load(out)):
        for i, v in enumerate(row):
            want = 1.5 * (i + 1) + 15 + step
            assert abs(v - want) < 1e-4, (step, i, v, want)
