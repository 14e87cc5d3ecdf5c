"""AdamW + gradient clipping of the training loop in seven launches (csrc/optimizer.hip).

``torch.optim.AdamW(model.parameters(), lr, weight_decay)`` followed every iteration by
``clip_grad_norm_(model.parameters(), 10)`` is what the reference's training script runs
(train_ScanNet_DDP_WarmUP.py:237-241, :421).  `FusedAdamW` keeps torch's interface and state layout
(``state[p] = {'step', 'exp_avg', 'exp_avg_sq'}``, one parameter group per hyper-parameter set) and torch's fused
arithmetic, and does norm + clip + update over all tensors of a group in ``2 * ceil(n / 72) + 1`` launches.  The
learning rate, step counter and clip coefficient live in a device record, so a captured HIP graph of the iteration
replays with their current values.
"""
import ctypes

import torch

import pcf_cuda
from pcf_fused import _call, _guard, _lib, _sig, _stream

_P, _I, _LL, _F, _D = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_float, ctypes.c_double
_PP = ctypes.POINTER(ctypes.c_void_p)
_max_tensors = _sig('pcf_hip_adamw_max_tensors', [])
_chunk = _sig('pcf_hip_adamw_chunk', [])
_list = _sig('pcf_hip_adamw_list', [_I, _I, _PP, _PP, _PP, _PP, ctypes.POINTER(_LL), _P, _P, _I, _D, _D, _D, _D, _P])
_finish = _sig('pcf_hip_adamw_finish', [_P, _I, _P, _F, _D, _D, _P])


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW for float32 parameters on one HIP device, with clip_grad_norm_ folded in:
    ``step(max_grad_norm=10)`` = ``clip_grad_norm_(params, 10); step()`` with ONE norm over all parameter groups.
    `last_grad_norm` is the (device) total norm of the gradients before clipping.

    Step counter: one float32 device counter per parameter group (``state[p]['step']`` of every parameter is a view of
    it).  Two consequences, both different from torch's per-parameter counters: a parameter that receives its first gradient
    later than the rest of its group takes the group's bias correction, not its own; and the counter is exact up to
    2**24 steps (16.7 M -- the reference's schedules run 300 epochs x ~600 iterations)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        if lr < 0 or eps < 0 or weight_decay < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1:
            raise ValueError('FusedAdamW: invalid hyper-parameter')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.last_grad_norm = None
        self._recs = {}

    def _record(self, gi, group, dev):
        """device float[8] of a group: lr, step, norm, coef, bias corrections; lr refreshed when the host value moved."""
        rec = self._recs.get(gi)
        if rec is None:
            t = torch.zeros(8, dtype=torch.float32, device=dev)
            steps = [float(self.state[p]['step']) for p in group['params'] if p in self.state and 'step' in self.state[p]]
            t[1] = max(steps) if steps else 0.0                     # resumed from a state_dict
            t[0] = float(group['lr'])
            rec = self._recs[gi] = [t, float(group['lr'])]
        elif rec[1] != float(group['lr']):
            rec[0][0:1].fill_(float(group['lr']))
            rec[1] = float(group['lr'])
        return rec[0]

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._recs = {}                      # step counters are re-read from the loaded state

    def sync_hyperparameters(self):
        """Push host-side learning-rate changes (a scheduler's) into the device records: call before replaying a captured
        iteration; `step()` does it by itself."""
        for gi, group in enumerate(self.param_groups):
            if gi in self._recs:
                self._record(gi, group, self._recs[gi][0].device)

    @torch.no_grad()
    def step(self, closure=None, max_grad_norm=None):
        """One AdamW update of every group; with `max_grad_norm` the gradients of ALL groups are first clipped by ONE global
        2-norm, as ``clip_grad_norm_(model.parameters(), max_norm)`` does (train_ScanNet_DDP_WarmUP.py:421): the squared
        norms of every group go into one partial buffer, every group's record then gets the same total norm and
        coefficient, and the updates run per group with the group's own hyper-parameters."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        nmax, chunk = _max_tensors(), _chunk()
        work, dev = [], None
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group['params'] if p.grad is not None]
            if not ps:
                continue
            dev = ps[0].device if dev is None else dev
            for p in ps:
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_cuda or p.device != dev \
                        or not p.is_contiguous() or p.grad.is_sparse:
                    raise RuntimeError('FusedAdamW: float32 contiguous parameters on one HIP device only')
                if not p.grad.is_contiguous():
                    p.grad = p.grad.contiguous()
                st = self.state[p]
                if not st:
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            rec = self._record(gi, group, dev)
            for p in ps:
                self.state[p]['step'] = rec[1]                       # one shared device counter (a view of the record)
            lists = [ps[i:i + nmax] for i in range(0, len(ps), nmax)]
            nparts = [sum((p.numel() + chunk - 1) // chunk for p in l) for l in lists]
            tables = []
            for l in lists:
                n = len(l)
                arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
                tables.append((n, arr(l), arr([p.grad for p in l]), arr([self.state[p]['exp_avg'] for p in l]),
                               arr([self.state[p]['exp_avg_sq'] for p in l]), (_LL * n)(*[p.numel() for p in l])))
            work.append((group, rec, tables, nparts))
        if not work:
            return loss
        total = sum(sum(nparts) for _, _, _, nparts in work)
        partials = torch.empty(total, dtype=torch.float32, device=dev)
        stream = _stream(dev)
        with _guard(dev):
            off = 0
            for group, rec, tables, nparts in work:          # squared norms of every group into ONE list of partials
                b1, b2 = group['betas']
                for (n, pp, gg, mm, vv, cc), k in zip(tables, nparts):
                    _call(_list, 0, n, pp, gg, mm, vv, cc, rec.data_ptr(), partials.data_ptr(), off, b1, b2, group['eps'],
                          group['weight_decay'], stream)
                    off += k
            for group, rec, tables, nparts in work:          # the same global norm and coefficient into every group's record
                b1, b2 = group['betas']
                _call(_finish, partials.data_ptr(), off, rec.data_ptr(), float(max_grad_norm or 0.0), b1, b2, stream)
            for group, rec, tables, nparts in work:
                b1, b2 = group['betas']
                for n, pp, gg, mm, vv, cc in tables:
                    _call(_list, 1, n, pp, gg, mm, vv, cc, rec.data_ptr(), None, 0, b1, b2, group['eps'], group['weight_decay'],
                          stream)
        self.last_grad_norm = work[0][1][2]
        return loss
