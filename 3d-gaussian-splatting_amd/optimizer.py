"""``Adam``: host side of ``gsr_adam_step``, with the call surface the reference trainer uses of
``torch.optim.Adam`` - construction from ``model.parameters()`` with ``lr`` and ``betas``
(``Trainer.py:24``), ``zero_grad(set_to_none=True)`` (``Trainer.py:53``), ``step()``
(``Trainer.py:80``) and ``state_dict()`` (``GaussianModel.py:330``, in PyTorch's layout so a
checkpoint written by either loads into the other).  All parameter tensors are updated by ONE
kernel launch.  There is no CPU path.
"""
import ctypes as C
from typing import Iterable, Optional, Sequence

import torch

from . import _lib
from .rasterizer import _call, _stream


class Adam:
    def __init__(self, params: Iterable[torch.Tensor], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        self.params = list(params)
        if not self.params:
            raise ValueError("optimizer got an empty parameter list")
        if len(self.params) > _lib.ADAM_MAX_TENSORS:
            raise ValueError(f"at most {_lib.ADAM_MAX_TENSORS} parameter tensors, got {len(self.params)}")
        for p in self.params:
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("Adam needs contiguous float32 parameters on the GPU (there is no CPU path)")
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.exp_avg = [torch.zeros_like(p, requires_grad=False) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p, requires_grad=False) for p in self.params]
        self.steps = [0] * len(self.params)

    # torch.optim.Optimizer exposes the hyper-parameters this way
    @property
    def param_groups(self):
        return [{"params": self.params, "lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": 0,
                 "amsgrad": False}]

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            if p.grad is None:
                continue
            if set_to_none:
                p.grad = None
            else:
                p.grad.detach_().zero_()

    @torch.no_grad()
    def step(self, grads: Optional[Sequence[Optional[torch.Tensor]]] = None):
        """One update of every parameter that has a gradient (``p.grad``, or ``grads[i]`` when given -
        e.g. the views of the flat gradient bucket of a view minibatch).  A parameter without a
        gradient is skipped and keeps its step count, as in PyTorch."""
        if grads is None:
            grads = [p.grad for p in self.params]
        if len(grads) != len(self.params):
            raise ValueError(f"{len(grads)} gradients for {len(self.params)} parameters")
        idx = [i for i, g in enumerate(grads) if g is not None]
        if not idx:
            return
        k = len(idx)
        keep = []     # gradient tensors made contiguous must outlive the launch
        vp = C.c_void_p * k
        p_arr, g_arr, m_arr, v_arr = vp(), vp(), vp(), vp()
        numel, steps = (C.c_uint64 * k)(), (C.c_uint32 * k)()
        for j, i in enumerate(idx):
            p, g = self.params[i], grads[i]
            if g.shape != p.shape or g.dtype != torch.float32 or g.device != p.device:
                raise ValueError(f"gradient {i}: shape {tuple(g.shape)} / {g.dtype} does not match the parameter")
            g = g if g.is_contiguous() else g.contiguous()
            keep.append(g)
            self.steps[i] += 1
            p_arr[j], g_arr[j] = p.data_ptr(), g.data_ptr()
            m_arr[j], v_arr[j] = self.exp_avg[i].data_ptr(), self.exp_avg_sq[i].data_ptr()
            numel[j], steps[j] = p.numel(), self.steps[i]
        lib = _lib.load()
        _call("gsr_adam_step", lib.gsr_adam_step, k, p_arr, g_arr, m_arr, v_arr, numel, steps, self.lr, self.betas[0],
              self.betas[1], self.eps, _stream(self.params[0].device))

    def resize(self, params: Iterable[torch.Tensor]):
        """Adopt parameter tensors that grew along dimension 0 (``GaussianModel.expand_capacity``,
        ``GaussianModel.py:116-140``): the moment buffers are padded with zeros and the step counts
        kept.  (The reference leaves its optimizer holding the tensors from before the growth,
        which then never receive a gradient again; a caller that wants to go on training calls this.)"""
        params = list(params)
        if len(params) != len(self.params):
            raise ValueError("resize needs the same number of parameter tensors")
        for i, p in enumerate(params):
            old = self.exp_avg[i].shape
            if tuple(p.shape[1:]) != tuple(old[1:]) or p.shape[0] < old[0]:
                raise ValueError(f"parameter {i}: {tuple(old)} cannot grow into {tuple(p.shape)}")
            if p.shape[0] > old[0]:
                for buf in (self.exp_avg, self.exp_avg_sq):
                    grown = torch.zeros_like(p, requires_grad=False)
                    grown[:old[0]] = buf[i]
                    buf[i] = grown
        self.params = params

    def state_dict(self):
        state = {i: {"step": torch.tensor(float(self.steps[i])), "exp_avg": self.exp_avg[i], "exp_avg_sq": self.exp_avg_sq[i]}
                 for i in range(len(self.params)) if self.steps[i] > 0}
        group = {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                 "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        group = sd["param_groups"][0]
        self.lr, self.betas, self.eps = float(group["lr"]), tuple(float(b) for b in group["betas"]), float(group["eps"])
        for i, p in enumerate(self.params):
            st = sd["state"].get(i)
            if st is None:
                self.steps[i] = 0
                self.exp_avg[i].zero_()
                self.exp_avg_sq[i].zero_()
                continue
            self.steps[i] = int(st["step"])
            self.exp_avg[i] = st["exp_avg"].to(device=p.device, dtype=torch.float32).contiguous().clone()
            self.exp_avg_sq[i] = st["exp_avg_sq"].to(device=p.device, dtype=torch.float32).contiguous().clone()
