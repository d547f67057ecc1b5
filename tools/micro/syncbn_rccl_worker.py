#!/usr/bin/env python3
"""tools/micro/syncbn_rccl_worker.py -- the FIRST thing to run when more than one GPU is visible (tools/ddp_first_contact.sh):
one rank per GPU under torchrun, RCCL backend.  Each rank runs the split-phase BatchNorm kernels (refign_amd/bn.py) on ITS slice
of a batch, (1) with the statistics exchanged through torch's process group, (2) through a communicator of our own
(refign_amd/rccl.py: the default of the N > 1 step), (3) the same call captured into a hipGraph and replayed, and compares
outputs, input / affine gradients and running statistics with nn.BatchNorm2d on the WHOLE batch; then the flat gradient buffer's
bucketed all-reduce (trainer.FlatGradBuffer) against a plain sum.  Prints one line per check on rank 0; exit code 0 = all passed."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    from refign_amd import bn as bnk, rccl
    from refign_amd.trainer import FlatGradBuffer
    ok = True

    def say(name, good, detail=""):
        nonlocal ok
        flag = torch.tensor([1.0 if good else 0.0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = ok and bool(flag.item())
        if rank == 0:
            print(f"{'PASS' if flag.item() else 'FAIL'}  {name}  {detail}", flush=True)

    torch.manual_seed(11)
    C, per = 256, 2
    full = (torch.randn(world * per, C, 9, 14, device=dev) * 1.5 + 0.3).to(torch.bfloat16)
    gfull = torch.randn(world * per, C, 9, 14, device=dev).to(torch.bfloat16)
    weight, bias = torch.linspace(0.5, 1.5, C, device=dev), torch.linspace(-0.4, 0.4, C, device=dev)
    ref = torch.nn.BatchNorm2d(C).to(dev)
    with torch.no_grad():
        ref.weight.copy_(weight)
        ref.bias.copy_(bias)
    xr = full.float().requires_grad_(True)
    yr = torch.relu(ref(xr))
    yr.backward(gfull.float())
    sl = slice(rank * per, (rank + 1) * per)

    def one(mode):
        mod = torch.nn.SyncBatchNorm(C).to(dev).train()
        with torch.no_grad():
            mod.weight.copy_(weight)
            mod.bias.copy_(bias)
        bnk._DIRECT["default"] = comm if mode != "torch" else None
        x = full[sl].contiguous(memory_format=torch.channels_last).requires_grad_(True)
        if mode == "graph":
            # forward only, captured: statistics pass, exchange (a kernel node), apply pass
            xs = x.detach()
            s = torch.cuda.Stream(dev)
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s), torch.no_grad():
                bnk.bn_act_train(xs, mod, True, torch.bfloat16)         # warm-up
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g), torch.no_grad():
                y = bnk.bn_act_train(xs, mod, True, torch.bfloat16)
            g.replay()
            torch.cuda.synchronize()
            e = float((y.float() - yr[sl]).abs().max() / yr.abs().max())
            return e < 0.02, f"y {e:.2e}"
        y = bnk.bn_act_train(x, mod, True, torch.bfloat16)
        y.backward(gfull[sl])
        gw, gb = mod.weight.grad.clone(), mod.bias.grad.clone()
        dist.all_reduce(gw)
        dist.all_reduce(gb)
        errs = {"y": float((y.float() - yr[sl]).abs().max() / yr.abs().max()),
                "gx": float((x.grad.float() - xr.grad[sl]).abs().max() / xr.grad.abs().max()),
                "gw": float((gw - ref.weight.grad).abs().max() / ref.weight.grad.abs().max()),
                "gb": float((gb - ref.bias.grad).abs().max() / ref.bias.grad.abs().max())}
        run = {"rm": float((mod.running_mean - ref.running_mean).abs().max()),
               "rv": float((mod.running_var - ref.running_var).abs().max())}
        return all(v < 0.02 for v in errs.values()) and all(v < 1e-4 for v in run.values()), \
            " ".join(f"{k} {v:.1e}" for k, v in {**errs, **run}.items())

    comm = None
    say("SyncBatchNorm kernels, statistics over torch.distributed (RCCL backend), %d ranks" % world, *one("torch"))
    try:
        comm = rccl.DirectComm(dev)
        say("direct RCCL communicator set-up (ncclCommInitRank over a broadcast id)", True)
    except Exception as e:                                             # noqa: BLE001 -- reported, then the run goes on without
        say("direct RCCL communicator set-up", False, f"{type(e).__name__}: {e}")
    if comm is not None:
        say("SyncBatchNorm kernels, statistics over the direct communicator", *one("direct"))
        say("the same exchange captured into a hipGraph and replayed", *one("graph"))
    bnk._DIRECT["default"] = None

    # the gradient buffer: ranges released early + the rest, bucketed, mean over ranks
    ps = [torch.nn.Parameter(torch.zeros(n, device=dev)) for n in (1000, 70001, 333, 2 ** 20 + 5)]
    buf = FlatGradBuffer(ps, [("a", ps[:2]), ("b", ps[2:])], bucket_mb=1)
    for i, p in enumerate(ps):
        p.grad.fill_(float(rank + 1) * (i + 1))
    buf.on_ready("a")
    buf.all_reduce_mean()
    torch.cuda.synchronize()
    want = sum(range(1, world + 1)) / world
    good = all(float((p.grad - want * (i + 1)).abs().max()) < 1e-5 for i, p in enumerate(ps))
    say("flat gradient buffer: released range + remainder, 1 MB buckets, mean over ranks", good)

    torch.cuda.synchronize()
    rccl.destroy_all()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
