"""Optimizer wrapper with the reference's surface (src/optim.py:4-54): `Optimizer(parameters, optimizer, lr,
eps, lr_scheduler, tf_start, tf_end, tf_step, ...)`, `.pre_step(step) -> tf_rate`, `.step()`, `.opt`,
`.get_opt_state_dict()/.load_opt_state_dict()`.

`.opt` is HipAdadelta: torch.optim.Adadelta arithmetic as ONE fused kernel over the model's flat
parameter/gradient buffers, with the global-norm clip and the NaN guard of BaseSolver.backward
(src/solver.py:96-103) folded in on the device (no host synchronisation per step).
"""
import numpy as np
import torch

from src import hipabi as H


def _flat_of(params):
    p0 = params[0]
    if not hasattr(p0, '_asr_flat'):
        raise RuntimeError('the fused optimizers need parameters in flat storage (src.asr.ASR / src.lm.RNNLM)')
    return p0._asr_flat


class HipAdadelta(torch.optim.Optimizer):
    def __init__(self, params, lr=1.0, rho=0.9, eps=1e-6, weight_decay=0):
        defaults = dict(lr=lr, rho=rho, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        allp = [p for g in self.param_groups for p in g['params']]
        self.flat_param, self.flat_grad, self._offsets = _flat_of(allp)
        n = self.flat_param.numel()
        self.square_avg = torch.zeros(n, dtype=torch.float32, device=self.flat_param.device)
        self.acc_delta = torch.zeros(n, dtype=torch.float32, device=self.flat_param.device)
        self.normsq = torch.zeros(1, dtype=torch.float64, device=self.flat_param.device)
        self._steps = 0
        for p in allp:   # per-parameter views so that state_dict() has torch.optim.Adadelta's layout
            o, k = self._offsets[id(p)], p.numel()
            self.state[p] = {'step': 0, 'square_avg': self.square_avg[o:o + k].view(p.shape),
                             'acc_delta': self.acc_delta[o:o + k].view(p.shape)}

    def zero_grad(self, set_to_none=False):
        self.flat_grad.zero_()

    def grad_norm(self, grad_mul=1.0):
        """Global L2 norm of the (scaled) flat gradient as a device scalar (no sync).  Parameters with
        requires_grad=False (ASR.fix_ctc_layer) take no part: the kernels accumulate into the flat buffer regardless,
        so their ranges are cleared first - they neither enter the clip norm nor move (torch skips them likewise)."""
        for g in self.param_groups:
            for p in g['params']:
                if not p.requires_grad and p.grad is not None:
                    p.grad.zero_()
        H.call('asr_sumsq', H.ptr(self.flat_grad), self.flat_grad.numel(), H.ptr(self.normsq), H.stream_ptr())
        return self.normsq

    @torch.no_grad()
    def step(self, clip=0.0, grad_mul=1.0, use_norm=False):
        g = self.param_groups[0]
        H.call('asr_adadelta_step', H.ptr(self.flat_param), H.ptr(self.flat_grad), H.ptr(self.square_avg),
               H.ptr(self.acc_delta), self.flat_param.numel(), float(g['lr']), float(g['rho']), float(g['eps']),
               float(g['weight_decay']), float(clip), H.ptr(self.normsq) if use_norm else None, float(grad_mul),
               H.ptr(H.collect_status()), H.stream_ptr())
        self._steps += 1

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        # re-point the loaded per-parameter tensors into the flat state buffers
        for g in self.param_groups:
            for p in g['params']:
                o, k = self._offsets[id(p)], p.numel()
                st = self.state[p]
                self.square_avg[o:o + k].copy_(st['square_avg'].reshape(-1))
                self.acc_delta[o:o + k].copy_(st['acc_delta'].reshape(-1))
                st['square_avg'] = self.square_avg[o:o + k].view(p.shape)
                st['acc_delta'] = self.acc_delta[o:o + k].view(p.shape)


class HipAdam(torch.optim.Optimizer):
    """torch.optim.Adam arithmetic (L2 weight decay, optional amsgrad) as one fused kernel over the model's flat buffers
    (asr_adam_step), with the same clip / NaN guard / status refusal as HipAdadelta.  The reference trains its RNN-LM with it
    (bin/train_lm.py:38, config lm_example.yaml: Adam, lr 1e-4)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad)
        super().__init__(params, defaults)
        allp = [p for g in self.param_groups for p in g['params']]
        self.flat_param, self.flat_grad, self._offsets = _flat_of(allp)
        n, dev = self.flat_param.numel(), self.flat_param.device
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.max_exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev) if amsgrad else None
        self.normsq = torch.zeros(1, dtype=torch.float64, device=dev)
        # ONE device-side count of APPLIED updates (asr_adam_step advances it under the kernel's own guard: a refused update -
        # abort status, NaN norm - does not move the bias corrections, as in torch); read back only where the host syncs anyway
        self.steps_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        for p in allp:   # per-parameter views: state_dict() has torch.optim.Adam's layout
            o, k = self._offsets[id(p)], p.numel()
            self.state[p] = {'step': torch.tensor(0.0), 'exp_avg': self.exp_avg[o:o + k].view(p.shape),
                             'exp_avg_sq': self.exp_avg_sq[o:o + k].view(p.shape)}
            if amsgrad:
                self.state[p]['max_exp_avg_sq'] = self.max_exp_avg_sq[o:o + k].view(p.shape)

    def zero_grad(self, set_to_none=False):
        self.flat_grad.zero_()

    def grad_norm(self, grad_mul=1.0):
        """As HipAdadelta.grad_norm: the ranges of requires_grad=False parameters are cleared first."""
        for g in self.param_groups:
            for p in g['params']:
                if not p.requires_grad and p.grad is not None:
                    p.grad.zero_()
        H.call('asr_sumsq', H.ptr(self.flat_grad), self.flat_grad.numel(), H.ptr(self.normsq), H.stream_ptr())
        return self.normsq

    @property
    def _steps(self):
        return int(self.steps_dev.item())

    @torch.no_grad()
    def step(self, clip=0.0, grad_mul=1.0, use_norm=False):
        g = self.param_groups[0]
        H.call('asr_adam_step', H.ptr(self.flat_param), H.ptr(self.flat_grad), H.ptr(self.exp_avg), H.ptr(self.exp_avg_sq),
               H.ptr(self.max_exp_avg_sq), self.flat_param.numel(), float(g['lr']), float(g['betas'][0]), float(g['betas'][1]),
               float(g['eps']), float(g['weight_decay']), 0, float(clip), H.ptr(self.normsq) if use_norm else None,
               float(grad_mul), H.ptr(H.collect_status()), H.ptr(self.steps_dev), H.stream_ptr())

    def state_dict(self):
        n = float(self._steps)                  # synchronises: checkpoints only
        for st in self.state.values():
            st['step'] = torch.tensor(n)
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        steps = 0
        for g in self.param_groups:
            for p in g['params']:
                o, k = self._offsets[id(p)], p.numel()
                st = self.state[p]
                steps = max(steps, int(float(st.get('step', 0))))
                for name, flat in (('exp_avg', self.exp_avg), ('exp_avg_sq', self.exp_avg_sq), ('max_exp_avg_sq', self.max_exp_avg_sq)):
                    if flat is not None and name in st:
                        flat[o:o + k].copy_(st[name].reshape(-1))
                        st[name] = flat[o:o + k].view(p.shape)
        self.steps_dev.fill_(steps)


class Optimizer():
    def __init__(self, parameters, optimizer, lr, eps, lr_scheduler=None, tf_start=1, tf_end=1, tf_step=1,
                 tf_step_start=0, weight_decay=0, amsgrad=False, **kwargs):
        self.tf_type = tf_end != 1
        self.tf_rate = lambda step: max(
            tf_end, tf_start - (tf_start - tf_end) * (step - tf_step_start) / tf_step if step >= tf_step_start else 1)
        self.opt_type = optimizer
        self.init_lr = lr
        self.sch_type = lr_scheduler
        if optimizer not in ('Adadelta', 'Adam'):
            raise NotImplementedError('HIP path implements the Adadelta and Adam steps (got %s)' % optimizer)
        cls = HipAdadelta if optimizer == 'Adadelta' else HipAdam
        if lr_scheduler == 'warmup':
            warmup_step = 4000.0
            init_lr = lr
            self.lr_scheduler = lambda step: init_lr * warmup_step ** 0.5 * \
                np.minimum((step + 1) * warmup_step ** -1.5, (step + 1) ** -0.5)
            self.opt = cls(parameters, lr=1.0)
        else:
            self.lr_scheduler = None
            if optimizer == 'Adam':
                self.opt = HipAdam(parameters, lr=lr, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad)
            else:
                self.opt = HipAdadelta(parameters, lr=lr, eps=eps, weight_decay=weight_decay)

    def get_opt_state_dict(self):
        return self.opt.state_dict()

    def load_opt_state_dict(self, state_dict):
        self.opt.load_state_dict(state_dict)

    def pre_step(self, step):
        self.opt.zero_grad()
        return self.tf_rate(step)

    def get_lr(self, step):
        return self.lr_scheduler(step) if self.lr_scheduler is not None else self.init_lr

    def step(self, **kw):
        self.opt.step(**kw)

    def create_msg(self):
        return ['Optim.spec.| Algo. = {}\t| Lr = {}\t (schedule = {})| Scheduled sampling = {}'
                .format(self.opt_type, self.init_lr, self.sch_type, self.tf_type)]
