"""AdamW step of the whole model as ONE kernel launch (csrc/reduce.hip: rfn_multi_adamw_f32).

The reference instantiates `torch.optim.AdamW` from its YAML (`optimizer:` section) and Lightning steps it; here the
torch optimizer object stays the owner of everything a checkpoint holds -- param_groups, `state[p]['exp_avg' |
'exp_avg_sq' | 'step']`, the LR scheduler writes `group['lr']` as always -- and only the arithmetic of `step()` moves:
torch's fused implementation needs ~4 ms of host time per step to regroup 1 090 tensors into 33 multi-tensor launches
(measured, profiles/r02_step_phases_events.txt); a chunk table built once makes it one launch.  torch performs the FIRST
step itself (it creates the state exactly as it would), and any configuration outside plain AdamW (amsgrad, maximize,
non-fp32 / non-CUDA parameters, gradients that are not where the table expects them) stays on torch's step."""
import math

import numpy as np
import torch

from . import _lib
from ._tensor import current_stream, on_device, ptr
from .params import refresh


class MultiTensorAdamW:
    def __init__(self, optimizer):
        self.opt = optimizer
        self._table = None
        self._sig = None
        self._steps = None
        self._t = 0
        self._synced_t = 0
        self._params = None
        self.launches = 0                                  # diagnostics / tests
        # state['step'] of the 1 000 parameters is brought up to date when somebody looks (checkpoint, torch's own step)
        optimizer.register_state_dict_pre_hook(lambda opt: self._sync_steps())
        # load_state_dict REPLACES every state tensor (mid-run resume / rollback): the chunk table would keep pointing at
        # the freed moment buffers and the kernel's own step counter would ignore the loaded one -- drop everything, the
        # next step() rebuilds from the loaded state
        optimizer.register_load_state_dict_post_hook(lambda opt: self._reset())

    def _reset(self):
        self._table = self._sig = self._steps = self._params = None
        self._t = self._synced_t = 0

    def _sync_steps(self):
        if self._steps is not None and self._synced_t != self._t:
            with torch.no_grad():
                torch._foreach_add_(self._steps, float(self._t - self._synced_t))
            self._synced_t = self._t

    def _plain(self):
        if type(self.opt) is not torch.optim.AdamW or len(self.opt.param_groups) > 8:
            return False
        for g in self.opt.param_groups:
            if g.get("amsgrad") or g.get("maximize") or g.get("differentiable") or torch.is_tensor(g["lr"]):
                return False
            for p in g["params"]:
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad is not None
                        and p.grad.dtype == torch.float32 and p.grad.is_contiguous()):
                    return False
        return True

    def _signature(self):
        ps = self._params
        if ps is None or len(ps) != sum(len(g["params"]) for g in self.opt.param_groups):
            ps = self._params = [p for g in self.opt.param_groups for p in g["params"]]
        g0, g1 = ps[0].grad, ps[-1].grad
        # ... and where the first and the last parameter's first moments live: replaced state is seen even without the hook
        m0, m1 = (self.opt.state.get(p, {}).get("exp_avg") for p in (ps[0], ps[-1]))
        return (len(ps), ps[0].data_ptr(), ps[-1].data_ptr(), None if g0 is None else g0.data_ptr(),
                None if g1 is None else g1.data_ptr(), None if m0 is None else m0.data_ptr(),
                None if m1 is None else m1.data_ptr(),
                tuple((g.get("amsgrad"), g.get("maximize"), torch.is_tensor(g["lr"])) for g in self.opt.param_groups))

    def _build(self):
        lib = _lib.load_library()
        chunk = lib.rfn_multi_cast_chunk_elems()
        rows, steps, t = [], [], None
        for gi, g in enumerate(self.opt.param_groups):
            for p in g["params"]:
                st = self.opt.state.get(p)
                if not st or "exp_avg" not in st or not torch.is_tensor(st.get("step")):
                    return False                            # state not created yet (or a foreign layout): torch steps
                m, v = st["exp_avg"], st["exp_avg_sq"]
                if not (m.dtype == torch.float32 and v.dtype == torch.float32 and m.is_contiguous() and v.is_contiguous()):
                    return False
                steps.append(st["step"])
                n, pp, gp, mp, vp = p.numel(), p.data_ptr(), p.grad.data_ptr(), m.data_ptr(), v.data_ptr()
                rows += [(pp + 4 * o, gp + 4 * o, mp + 4 * o, vp + 4 * o, min(chunk, n - o) | (gi << 56))
                         for o in range(0, n, chunk)]
        ts = {float(s) for s in steps[:1] + steps[-1:]}
        if len(ts) != 1:
            return False
        self._t = self._synced_t = int(ts.pop())
        dev = self.opt.param_groups[0]["params"][0].device
        self._table = (torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(dev), len(rows), dev)
        self._steps = steps
        self._sig = self._signature()
        return True

    def step(self):
        sig = self._signature()
        if self._table is None or sig != self._sig:         # first call, or parameters / gradients / flags changed
            self._sync_steps()
            self._table = None
            if not (self._plain() and self._build()):
                return self.opt.step()                      # torch's own step (it also creates the state on step 1)
        self._t += 1
        t = self._t
        args = []
        for g in self.opt.param_groups:
            b1, b2 = g["betas"]
            args += [float(g["lr"]), b1, b2, float(g["eps"]), float(g["weight_decay"]), 1.0 - b1 ** t,
                     math.sqrt(1.0 - b2 ** t), 1.0 - b1, 1.0 - b2]
        host = np.asarray(args, dtype=np.float32)
        table, n, dev = self._table
        with torch.no_grad():
            with on_device(dev):
                rc = _lib.load_library().rfn_multi_adamw_f32(ptr(table), n, host.ctypes.data, len(self.opt.param_groups),
                                                             current_stream(dev))
            _lib.check(rc, "multi_adamw_f32")
        self.launches += 1
        self.opt._opt_called = True                        # what Optimizer.step's wrapper tells the LR scheduler
        refresh((p for g in self.opt.param_groups for p in g["params"]), plan_key=("optimizer", id(self.opt)))
