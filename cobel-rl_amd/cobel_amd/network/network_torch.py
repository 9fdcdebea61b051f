"""PyTorch network adapter — ``cobel.network.TorchNetwork`` (network/network_torch.py:31-451) for
PyTorch-ROCm, plus the stacked form the vectorised DQN uses.

Single network: same constructor and methods as the reference (NumPy in, NumPy out; loss with
``reduction='none'`` then ``.mean()``; Adam / MSE defaults; ``get_weights`` = state_dict values;
``get_layer_activity`` returns ``(units, batch)``).  Two deliberate differences: ``clone()`` keeps
the device (the reference silently drops it, network_torch.py:212-217) and ``*_on_device``
methods take / return tensors without the host round trip of ``predict_on_batch``.

``replicate(n)`` returns a ``StackedTorchNetwork``: n independent copies of the same architecture
whose parameters are stacked along a leading instance axis and evaluated with
``torch.func.vmap(functional_call)`` — one batched GEMM per layer for all instances, one fused
optimizer step over the stacked parameters (element-wise optimizers act per instance exactly as
n separate optimizers would).
"""
from __future__ import annotations

import copy
import ctypes as C

import numpy as np
import torch
from torch import nn, optim
from torch.func import functional_call, stack_module_state, vmap

from .network import Network

_LOSSES = {
    'kl_divergence': nn.KLDivLoss, 'cosine_similarity': nn.CosineEmbeddingLoss,
    'poisson': nn.PoissonNLLLoss, 'gaussian': nn.GaussianNLLLoss,
    'binary_crossentropy': nn.BCELoss, 'hinge': nn.HingeEmbeddingLoss,
    'margin_ranking': nn.MarginRankingLoss, 'multi_label_margin_ranking': nn.MultiLabelMarginLoss,
    'mean_absolute_error': nn.L1Loss, 'mae': nn.L1Loss, 'mean_squared_error': nn.MSELoss,
    'mse': nn.MSELoss, 'huber_loss': nn.HuberLoss, 'huber': nn.HuberLoss,
    'categorical_crossentropy': nn.CrossEntropyLoss, 'crossentropy': nn.CrossEntropyLoss,
    'connectionist_temporal_classification': nn.CTCLoss, 'ctc': nn.CTCLoss,
    'negative_log_likelihood': nn.NLLLoss, 'nll': nn.NLLLoss,
}
_OPTIMIZERS = {name.lower(): getattr(optim, name) for name in
               ('Adadelta', 'Adagrad', 'Adam', 'AdamW', 'Adamax', 'SparseAdam', 'ASGD', 'LBFGS',
                'NAdam', 'RAdam', 'RMSprop', 'Rprop', 'SGD')}


def _make_optimizer(spec, params, kwargs):
    if isinstance(spec, optim.Optimizer):
        return spec
    cls = _OPTIMIZERS.get(str(spec).lower(), optim.Adam)     # Adam by default
    return cls(params, **(kwargs or {}))


def _make_loss(spec, kwargs):
    if isinstance(spec, nn.modules.loss._Loss):
        crit = type(spec)(**kwargs) if kwargs else spec
    else:
        crit = _LOSSES.get(str(spec).lower() if spec is not None else 'mse', nn.MSELoss)(
            **(kwargs or {}))
    crit.reduction = 'none'    # per-sample losses so that samples can be weighted
    return crit


class TorchNetwork(Network):
    def __init__(self, model: nn.Module, optimizer=None, loss=None, optimizer_params=None,
                 loss_params=None, activations=None, device: str = 'cpu') -> None:
        self.model = model
        self.set_device(device)
        self.set_optimizer(optimizer, optimizer_params)
        self.set_loss(loss, loss_params)
        self.activations = {} if activations is None else activations

    # -- device entry points ------------------------------------------------------------------
    def predict_on_device(self, batch: torch.Tensor) -> torch.Tensor:
        with torch.inference_mode():
            return self.model(batch)

    def train_on_device(self, batch: torch.Tensor, targets: torch.Tensor, sample_weights=None):
        self.optimizer.zero_grad()
        loss = self.criterion(self.model(batch), targets)
        if sample_weights is not None:
            loss = loss * sample_weights
        loss.mean().backward()
        self.optimizer.step()

    # -- reference surface (NumPy) ------------------------------------------------------------
    def _tensor(self, a):
        return torch.as_tensor(np.asarray(a), device=self.device)

    def predict_on_batch(self, batch):
        return self.predict_on_device(self._tensor(batch)).detach().cpu().numpy()

    def train_on_batch(self, batch, targets, sample_weights=None) -> None:
        targets = np.asarray(targets)
        if targets.ndim == 1:
            targets = targets.reshape(targets.shape[0], 1)
        w = None
        if sample_weights is not None:
            w = np.asarray(sample_weights)
            w = self._tensor(w if w.ndim == 2 else w.reshape(w.size, 1))
        self.train_on_device(self._tensor(batch), self._tensor(targets), w)

    def get_weights(self):
        return [v.detach().cpu().numpy().copy() for v in self.model.state_dict().values()]

    def set_weights(self, weights) -> None:
        state = self.model.state_dict()
        for key, w in zip(state, weights):
            state[key] = torch.as_tensor(np.asarray(w), device=self.device)
        self.model.load_state_dict(state)

    def clone(self):
        net = copy.deepcopy(self.model)
        twin = type(self)(net, type(self.optimizer)(params=net.parameters()),
                          copy.deepcopy(self.criterion),
                          activations=copy.deepcopy(self.activations), device=str(self.device))
        twin.criterion.load_state_dict(self.criterion.state_dict())
        twin.optimizer.load_state_dict(self.optimizer.state_dict())
        return twin

    def set_optimizer(self, optimizer, parameters=None) -> None:
        self.optimizer = _make_optimizer(optimizer, self.model.parameters(), parameters)

    def set_loss(self, loss, parameters=None) -> None:
        self.criterion = _make_loss(loss, parameters)

    def get_layer_activity(self, batch, layer):
        children = list(self.model.named_children())
        if isinstance(layer, str):
            key = layer
            index = next((i for i, (n, _) in enumerate(children) if n == layer), 10**10)
        else:
            key, index = children[int(layer)][0], int(layer)
        seen = {}
        handle = self.model.get_submodule(key).register_forward_hook(
            lambda mod, inp, out: seen.__setitem__(key, out.detach()))
        self.predict_on_batch(batch)
        handle.remove()
        act = seen[key]
        if isinstance(self.activations, list):
            assert len(self.activations) >= index, 'Index not in activation list!'
            fn = self.activations[index]
        else:
            assert isinstance(self.activations, dict), \
                'String layerkeys not compatible with activation lists!'
            assert key in self.activations, 'Key not in activation dict!'
            fn = self.activations[key]
        if fn is not None:
            act = fn(act)
        return act.cpu().numpy().T

    def set_trainable(self, layers, trainable) -> None:
        names = [k.split('.')[0] for k in self.model.state_dict() if '.weight' in k]
        for pos, layer in enumerate(layers):
            if isinstance(layer, int):
                assert layer < len(names)
                name = names[layer]
                flag = trainable[pos] if isinstance(trainable, list) else trainable
            else:
                assert layer in names
                name = layer
                flag = trainable[layer] if isinstance(trainable, dict) else trainable
            self.model.get_parameter(name + '.weight').requires_grad = bool(flag)
            self.model.get_parameter(name + '.bias').requires_grad = bool(flag)

    def set_device(self, device='cpu') -> None:
        self.device = torch.device(device)
        self.model.to(self.device)

    # -- vectorised form ------------------------------------------------------------------------
    def replicate(self, n: int) -> 'StackedTorchNetwork':
        return StackedTorchNetwork(self, n)


class FlexibleTorchNetwork(TorchNetwork):
    """``cobel.network.FlexibleTorchNetwork`` (network/network_torch.py:454-1050) for the case the
    accelerated path meets: ONE array-valued input and one output head, where the reference class
    behaves exactly like ``TorchNetwork`` (``prepare_batch`` just wraps the array).  Multi-input /
    multi-head models (dict or list batches, per-head losses and ``loss_weights``) belong to the
    simulator-backed interfaces and are outside this path."""

    def __init__(self, model: nn.Module, optimizer=None, loss=None, optimizer_params=None,
                 loss_params=None, loss_weights=None, activations=None,
                 device: str = 'cpu') -> None:
        if isinstance(loss, (list, dict)) or loss_weights is not None:
            raise NotImplementedError('multi-head losses are outside the accelerated path')
        super().__init__(model, optimizer, loss, optimizer_params, loss_params, activations, device)

    def clone(self):
        net = copy.deepcopy(self.model)
        twin = type(self)(net, type(self.optimizer)(params=net.parameters()),
                          copy.deepcopy(self.criterion),
                          activations=copy.deepcopy(self.activations), device=str(self.device))
        twin.criterion.load_state_dict(self.criterion.state_dict())
        twin.optimizer.load_state_dict(self.optimizer.state_dict())
        return twin


class StackedTorchNetwork:
    """n independent copies of one architecture with parameters stacked on axis 0."""

    def __init__(self, proto: TorchNetwork, n: int) -> None:
        self.n, self.device = n, proto.device
        self.base = copy.deepcopy(proto.model).to('meta')   # structure only
        params, buffers = stack_module_state([proto.model])
        self.params = {k: v.detach().expand(n, *v.shape[1:]).clone().requires_grad_(True)
                       for k, v in params.items()}
        self.buffers = {k: v.detach().expand(n, *v.shape[1:]).clone() for k, v in buffers.items()}
        self.criterion = copy.deepcopy(proto.criterion)
        opt = proto.optimizer
        self.optimizer = type(opt)(list(self.params.values()), **{
            k: v for k, v in opt.defaults.items() if k not in ('foreach', 'fused', 'capturable',
                                                               'differentiable', 'maximize')})

        def one(p, b, x):
            return functional_call(self.base, (p, b), (x,))
        self._fwd = vmap(one)

    def forward(self, batch: torch.Tensor) -> torch.Tensor:
        """batch [n, B, ...] -> [n, B, out]"""
        return self._fwd(self.params, self.buffers, batch)

    def predict_on_device(self, batch: torch.Tensor) -> torch.Tensor:
        with torch.no_grad():
            return self.forward(batch)

    def train_on_device(self, batch: torch.Tensor, targets: torch.Tensor, active=None,
                        sample_mask=None, blend_into=None, tau: float = 0.0) -> None:
        """Per instance: mean over the per-sample losses, summed over instances so that every
        instance receives exactly the gradient it would compute alone.  ``active`` ([n] bool)
        freezes the other instances completely — parameters AND optimizer state — as if their
        own single-instance run had simply not executed this step (Adam / AdamW-free path;
        other optimizers only get their parameters restored).  ``sample_mask`` ([n, B] bool):
        instance i trains on the samples it marks only, exactly as if it had been handed that
        sub-batch (mean over its elements); instances that mark none must be outside ``active``.
        ``blend_into`` (another stack of the same shape) receives ``w += tau * (w_new - w)`` for
        the instances that stepped — the DQN target update, fused into the optimizer kernel where
        that one runs."""
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.criterion(self.forward(batch), targets)
        if sample_mask is None:
            loss.reshape(self.n, -1).mean(dim=1).sum().backward()
        else:
            per_sample = loss.reshape(self.n, loss.shape[1], -1)
            width = per_sample.shape[2]
            kept = per_sample * sample_mask[..., None].to(per_sample.dtype)
            count = sample_mask.sum(dim=1).clamp(min=1).to(per_sample.dtype)
            (kept.sum(dim=(1, 2)) / (count * width)).sum().backward()
        if self._fused_adam_ok():
            self._adam_fused(active, blend_into, tau)
            return
        if blend_into is not None:
            self._train_step_torch(active)
            blend_into.blend_from(self, tau, active)
            return
        self._train_step_torch(active)

    def _train_step_torch(self, active) -> None:
        if active is None and getattr(self, '_diverged', False):
            # per-instance step counts exist: torch's optimizer (one shared count) no longer fits
            active = torch.ones(self.n, dtype=torch.bool, device=self.device)
        if active is None:
            self.optimizer.step()
        elif type(self.optimizer) is optim.Adam and not self.optimizer.defaults.get('amsgrad'):
            self._adam_masked(active)
        else:
            keep = [p.detach().clone() for p in self.params.values()]
            self.optimizer.step()
            with torch.no_grad():
                for p, old in zip(self.params.values(), keep):
                    mask = active.view(self.n, *([1] * (p.dim() - 1)))
                    p.copy_(torch.where(mask, p, old))

    # -- fused Adam (libcobel_hip: cobel_adam_step) ---------------------------------------------
    fused_adam = True      # False keeps torch's own optimizer step (tests compare the two)

    def _fused_adam_ok(self) -> bool:
        opt = self.optimizer
        if not self.fused_adam or type(opt) is not optim.Adam or self.device.type != 'cuda':
            return False
        g = opt.param_groups[0]
        return (len(opt.param_groups) == 1 and not g.get('amsgrad') and not g.get('maximize')
                and not torch.is_tensor(g['lr'])
                and all(p.dtype in (torch.float32, torch.float64) for p in self.params.values()))

    @torch.no_grad()
    def _adam_fused(self, active, blend_into=None, tau: float = 0.0) -> None:
        """torch.optim.Adam's update for the active instances in one kernel per parameter tensor;
        per-instance step counts as in ``_adam_masked`` (which it replaces on the GPU)."""
        from .. import _lib
        self._diverged = True
        opt = self.optimizer
        group = opt.param_groups[0]
        lr, (b1, b2), eps, wd = group['lr'], group['betas'], group['eps'], group['weight_decay']
        steps = getattr(self, '_steps', None)
        if steps is None:    # seeded from whatever the optimizer has counted so far
            seen = [float(st['steps'].max()) if 'steps' in st else float(st.get('step', 0.0))
                    for st in opt.state.values()]
            steps = self._steps = torch.full((self.n,), max(seen, default=0.0),
                                             dtype=torch.float64, device=self.device)
            for p in self.params.values():
                st = opt.state[p]
                if 'steps' in st:
                    steps.copy_(st['steps'])
                    break
        if active is None:
            steps += 1.0
            mask = None
        else:
            steps += active.to(torch.float64)
            mask = active.to(torch.uint8)
        stream = _lib.current_stream(self.device)
        for name, p in self.params.items():
            if p.grad is None:
                continue
            st = opt.state[p]
            if 'exp_avg' not in st:
                st['exp_avg'] = torch.zeros_like(p)
                st['exp_avg_sq'] = torch.zeros_like(p)
                st['step'] = torch.zeros((), dtype=torch.float32, device=p.device)
            st['steps'] = steps          # one shared count per instance for all tensors
            grad = p.grad.contiguous()
            tgt = None if blend_into is None else blend_into.params[name]
            _lib.check(_lib.lib().cobel_adam_step(
                _lib.ptr(p), _lib.ptr(grad), _lib.ptr(st['exp_avg']), _lib.ptr(st['exp_avg_sq']),
                _lib.ptr(steps), _lib.ptr(mask), self.n, p.numel() // self.n,
                1 if p.dtype == torch.float64 else 0, float(lr), float(b1), float(b2), float(eps),
                float(wd), _lib.ptr(tgt), float(tau), stream))
        if blend_into is not None and self.buffers:
            for k, v in blend_into.buffers.items():
                if v.is_floating_point():
                    w = tau if active is None else \
                        active.view(self.n, *([1] * (v.dim() - 1))).to(v.dtype) * tau
                    v.lerp_(self.buffers[k], w)

    # -- fused DQN replay step (libcobel_hip: cobel_dqn_replay) -----------------------------------
    fused_mlp = True       # False keeps the PyTorch forward / backward (tests compare the two)

    def _mlp3_names(self):
        """Names of the three Linear layers if the model computes Linear - ReLU - Linear - ReLU -
        Linear on a (flattened) input, else None (cached).  Recognised by behaviour, not by class:
        the reference's demo and test models are custom ``Module``s that call
        ``torch.nn.functional.relu`` in ``forward`` (demo/topology/demo_dqn.py:36-60), others are
        ``nn.Sequential``s — so the model must own exactly three Linear layers (all trainable,
        nothing else with parameters or buffers) whose shapes chain, and instance 0 must agree
        with the explicit formula on random probes."""
        if hasattr(self, '_mlp3'):
            return self._mlp3
        self._mlp3 = None
        lin = [(k, m) for k, m in self.base.named_modules() if type(m) is nn.Linear]
        want = [k + s for k, _ in lin for s in ('.weight', '.bias')]
        if len(lin) != 3 or self.buffers or sorted(want) != sorted(self.params) or \
                not all(p.requires_grad for p in self.params.values()):
            return None
        names = [k for k, _ in lin]
        w = [self.params[k + '.weight'] for k in names]
        if w[1].shape[2] != w[0].shape[1] or w[2].shape[2] != w[1].shape[1]:
            return None
        # structure: besides the three Linear layers only modules that cannot change the function
        # (a saturating activation module such as ReLU6 / Hardtanh is refused outright)
        leaves = [m for m in self.base.modules() if not list(m.children())]
        if not all(type(m) in (nn.Linear, nn.ReLU, nn.Flatten, nn.Identity) for m in leaves
                   if m is not self.base or type(m) is nn.Linear):
            return None
        try:
            with torch.no_grad():
                gen = torch.Generator(device='cpu').manual_seed(1234)
                ok = True
                # behaviour: at the weights as they are, and with inputs and weights scaled up so
                # that hidden pre-activations reach ~1e5 — a clamp, ReLU6 or Hardtanh written
                # inside a custom forward() equals ReLU on small activations only
                for x_scale, w_scale in ((1.0, 1.0), (1e3, 30.0)):
                    p0 = {k: v[0] * w_scale for k, v in self.params.items()}
                    x = ((torch.rand((16, w[0].shape[2]), generator=gen, dtype=torch.float64) * 4
                          - 2) * x_scale).to(device=self.device, dtype=w[0].dtype)
                    got = functional_call(self.base, (p0, {}), (x,))
                    h = x
                    for k in names[:2]:
                        h = torch.relu(torch.nn.functional.linear(h, p0[k + '.weight'],
                                                                  p0[k + '.bias']))
                    ref = torch.nn.functional.linear(h, p0[names[2] + '.weight'],
                                                     p0[names[2] + '.bias'])
                    tol = 1e-11 if w[0].dtype == torch.float64 else 1e-5
                    hidden_used = bool((h == 0).any()) and bool((h > 0).any())  # both sides of
                    ok = ok and got.shape == ref.shape and hidden_used and \
                        torch.allclose(got, ref, rtol=tol, atol=tol * float(ref.abs().max() + 1))
                if ok:                                                          # the ReLUs seen
                    self._mlp3 = names
        except Exception:      # a forward that does not take one [B, D] batch: not this shape
            self._mlp3 = None
        return self._mlp3

    def dqn_replay_fused(self, target: 'StackedTorchNetwork', states, actions, rewards, next_states,
                         nonterminal, gamma: float, ddqn: bool, tau: float, active=None) -> bool:
        """The whole replay step of ``DQN.replay`` (targets, MSE backward, Adam step, target blend
        with ``tau``; 0 = no blend) in one kernel per call.  Returns False — and does nothing — if
        the network, loss, optimizer or batch is not of the shape ``cobel_dqn_replay`` covers; the
        caller then takes the PyTorch path."""
        from .. import _lib
        names = self._mlp3_names() if self.fused_mlp else None
        if names is None or not self._fused_adam_ok() or type(self.criterion) is not nn.MSELoss \
                or getattr(self.criterion, 'reduction', '') != 'none' \
                or target._mlp3_names() != names or states.dim() != 3:
            return False
        w = [self.params[n + '.weight'] for n in names]
        dtype = w[0].dtype
        dims = (w[0].shape[2], w[0].shape[1], w[1].shape[1], w[2].shape[1])
        if w[1].shape[2] != dims[1] or w[2].shape[2] != dims[2] or states.dtype != dtype or \
                _lib.lib().cobel_dqn_replay_query(dims[0], dims[1], dims[2], dims[3],
                                                  states.shape[1], int(dtype == torch.float64),
                                                  None) != _lib.OK:
            return False
        self._diverged = True
        opt = self.optimizer
        group = opt.param_groups[0]
        steps = getattr(self, '_steps', None)
        if steps is None:    # seeded from whatever the optimizer has counted so far
            seen = [float(st['steps'].max()) if 'steps' in st else float(st.get('step', 0.0))
                    for st in opt.state.values()]
            steps = self._steps = torch.full((self.n,), max(seen, default=0.0),
                                             dtype=torch.float64, device=self.device)
            for p in self.params.values():
                if 'steps' in opt.state[p]:
                    steps.copy_(opt.state[p]['steps'])
                    break
        mask = None
        if active is None:
            steps += 1.0
        else:
            steps += active.to(torch.float64)
            mask = active.to(torch.uint8)
        run = _lib.DQNReplay()
        keep = [mask, steps]
        for k, n in enumerate(names):
            for kind, dst_p, dst_t, dst_m, dst_v in (('.weight', run.w, run.w_target, run.m_w, run.v_w),
                                                     ('.bias', run.b, run.b_target, run.m_b, run.v_b)):
                p = self.params[n + kind]
                st = opt.state[p]
                if 'exp_avg' not in st:
                    st['exp_avg'] = torch.zeros_like(p)
                    st['exp_avg_sq'] = torch.zeros_like(p)
                    st['step'] = torch.zeros((), dtype=torch.float32, device=p.device)
                st['steps'] = steps
                dst_p[k], dst_t[k] = _lib.ptr(p), _lib.ptr(target.params[n + kind])
                dst_m[k], dst_v[k] = _lib.ptr(st['exp_avg']), _lib.ptr(st['exp_avg_sq'])
        tensors = [states.contiguous(), next_states.contiguous(),
                   actions.to(torch.int64).contiguous(), rewards.to(dtype).contiguous(),
                   nonterminal.to(dtype).contiguous()]
        keep += tensors
        run.steps, run.active = _lib.ptr(steps), _lib.ptr(mask)
        run.states, run.next_states, run.actions = (_lib.ptr(t) for t in tensors[:3])
        run.rewards, run.nonterminal = _lib.ptr(tensors[3]), _lib.ptr(tensors[4])
        run.n, run.batch = self.n, states.shape[1]
        run.n_inputs, run.n_hidden1, run.n_hidden2, run.n_actions = dims
        run.is_float64, run.ddqn = int(dtype == torch.float64), int(bool(ddqn))
        run.gamma, run.lr = float(gamma), float(group['lr'])
        run.beta1, run.beta2 = (float(b) for b in group['betas'])
        run.eps, run.weight_decay, run.tau = float(group['eps']), float(group['weight_decay']), float(tau)
        _lib.check(_lib.lib().cobel_dqn_replay(C.byref(run), _lib.current_stream(self.device)))
        return True

    def make_capturable(self) -> None:
        """Prepare the optimizer for HIP-graph capture: its step counters move to the device
        (torch's ``capturable`` mode), so that a replayed graph advances them."""
        opt = self.optimizer
        for group in opt.param_groups:
            if 'capturable' in group:
                group['capturable'] = True
        for st in opt.state.values():
            if 'step' in st and torch.is_tensor(st['step']):
                # float64: in capturable mode the bias corrections beta ** step are formed from this
                # tensor (torch's default float32 counter would cost ~1e-7 relative per step)
                st['step'] = st['step'].to(device=self.device, dtype=torch.float64)

    @torch.no_grad()
    def _adam_masked(self, active: torch.Tensor) -> None:
        """torch.optim.Adam's update (same operation order) applied to the active instances only;
        step counts are kept per instance so bias corrections match each instance's own history."""
        self._diverged = True
        opt = self.optimizer
        group = opt.param_groups[0]
        lr, (b1, b2), eps, wd = group['lr'], group['betas'], group['eps'], group['weight_decay']
        for p in self.params.values():
            st = opt.state[p]
            if 'exp_avg' not in st:
                st['exp_avg'] = torch.zeros_like(p)
                st['exp_avg_sq'] = torch.zeros_like(p)
                st['step'] = torch.zeros((), dtype=torch.float32, device=p.device)
            if 'steps' not in st:   # per-instance step counts (seeded from torch's scalar count)
                st['steps'] = torch.full((self.n,), float(st['step']), dtype=torch.float64,
                                         device=p.device)
            shape = (self.n,) + (1,) * (p.dim() - 1)
            mask = active.view(shape)
            g = p.grad if wd == 0 else p.grad.add(p, alpha=wd)
            st['steps'] += active.to(torch.float64)
            t = st['steps'].clamp(min=1.0).view(shape)
            m_new = st['exp_avg'].lerp(g, 1 - b1)
            v_new = (st['exp_avg_sq'] * b2).addcmul_(g, g, value=1 - b2)
            st['exp_avg'].copy_(torch.where(mask, m_new, st['exp_avg']))
            st['exp_avg_sq'].copy_(torch.where(mask, v_new, st['exp_avg_sq']))
            bc1 = 1 - b1 ** t
            bc2_sqrt = (1 - b2 ** t).sqrt()
            denom = (st['exp_avg_sq'].sqrt() / bc2_sqrt.to(p.dtype)).add_(eps)
            upd = p - (lr / bc1).to(p.dtype) * (st['exp_avg'] / denom)
            p.copy_(torch.where(mask, upd, p))
            st['step'] += 1   # keeps torch's own bookkeeping monotone (used only if all step)

    def blend_from(self, other: 'StackedTorchNetwork', tau: float, active=None) -> None:
        """w += tau * (w_other - w) for every state entry (dqn.py:366-371), in place on device;
        instances outside ``active`` keep their weights."""
        with torch.no_grad():
            for k, v in list(self.params.items()) + list(self.buffers.items()):
                if not v.is_floating_point():
                    continue
                src = other.params[k] if k in other.params else other.buffers[k]
                if active is None:
                    v.lerp_(src, tau)
                else:
                    w = active.view(self.n, *([1] * (v.dim() - 1))).to(v.dtype) * tau
                    v.lerp_(src, w)

    def copy_from(self, other: 'StackedTorchNetwork', active=None) -> None:
        with torch.no_grad():
            for k, v in list(self.params.items()) + list(self.buffers.items()):
                src = other.params[k] if k in other.params else other.buffers[k]
                if active is None:
                    v.copy_(src)
                else:
                    v.copy_(torch.where(active.view(self.n, *([1] * (v.dim() - 1))), src, v))

    def clone(self) -> 'StackedTorchNetwork':
        twin = copy.copy(self)
        twin._diverged = False
        twin.params = {k: v.detach().clone().requires_grad_(True) for k, v in self.params.items()}
        twin.buffers = {k: v.clone() for k, v in self.buffers.items()}
        twin.criterion = copy.deepcopy(self.criterion)
        twin.optimizer = type(self.optimizer)(list(twin.params.values()), **{
            k: v for k, v in self.optimizer.defaults.items()
            if k not in ('foreach', 'fused', 'capturable', 'differentiable', 'maximize')})

        def one(p, b, x):
            return functional_call(twin.base, (p, b), (x,))
        twin._fwd = vmap(one)
        return twin

    def get_weights(self, instance: int = 0):
        return [v[instance].detach().cpu().numpy().copy()
                for v in list(self.params.values()) + list(self.buffers.values())]

    def write_back(self, net: 'TorchNetwork', instance: int = 0) -> None:
        """Copy one instance's parameters and buffers into a single network's module — after a
        run the user's ``model_online`` / ``model_target`` hold what was trained (instance 0), as
        the reference's attributes do (agent/dqn.py:108-110)."""
        with torch.no_grad():
            state = dict(net.model.named_parameters())
            state.update(dict(net.model.named_buffers()))
            for k, v in list(self.params.items()) + list(self.buffers.items()):
                state[k].copy_(v[instance].to(state[k].device))

    def matches(self, net: 'TorchNetwork', instance: int = 0) -> bool:
        """Whether a single network still holds this instance's values (False after the user
        assigned new weights to it)."""
        state = dict(net.model.named_parameters())
        state.update(dict(net.model.named_buffers()))
        return all(torch.equal(v[instance], state[k].detach().to(v.device))
                   for k, v in list(self.params.items()) + list(self.buffers.items()))

    def load_from(self, net: 'TorchNetwork') -> None:
        """Every instance takes the single network's current values."""
        with torch.no_grad():
            state = dict(net.model.named_parameters())
            state.update(dict(net.model.named_buffers()))
            for k, v in list(self.params.items()) + list(self.buffers.items()):
                v.copy_(state[k].detach().to(v.device).expand_as(v))
