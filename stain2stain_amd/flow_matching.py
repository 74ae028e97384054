"""Flow-matching step logic around the network, mirroring the reference's LightningModule surface.

  reference                                                               here
  ----------------------------------------------------------------------  ---------------------------
  torchcfm.ConditionalFlowMatcher.sample_location_and_conditional_flow    ConditionalFlowMatcher
    (third-party, call site src/models/conditional_flow_matching.py:66)
  ConditionalFlowMatchingLitModule.model_step / training_step /           ConditionalFlowMatchingModule
    configure_optimizers / generate (conditional_flow_matching.py:53-170)
  NeuralODE(dopri5).trajectory in generate (:157-170, torchdyn, absent)   euler_generate (fixed step,
                                                                          BASELINE.json config 4)

``ConditionalFlowMatchingModule`` subclasses ``lightning.LightningModule`` when Lightning is
importable and ``torch.nn.Module`` otherwise (this image has no Lightning), with the same method
names, argument meaning and return values, so the reference's training loop can drive it.
"""
from __future__ import annotations

from typing import Any, Dict, Optional, Tuple

import torch

from . import ops

try:  # pragma: no cover - Lightning is not installed in the build image
    from lightning import LightningModule as _Base
except Exception:  # noqa: BLE001
    class _Base(torch.nn.Module):
        def log(self, *args, **kwargs) -> None:
            pass


class ConditionalFlowMatcher:
    """Independent-coupling conditional flow matching: xt = t x1 + (1-t) x0 + sigma eps, ut = x1 - x0."""

    def __init__(self, sigma: float = 0.0):
        self.sigma = float(sigma)

    def sample_location_and_conditional_flow(self, x0: torch.Tensor, x1: torch.Tensor,
                                             t: Optional[torch.Tensor] = None, return_noise: bool = False):
        if t is None:
            t = torch.rand(x0.shape[0], device=x0.device, dtype=torch.float32)
        eps = torch.randn_like(x0) if (self.sigma != 0.0 or return_noise) else None
        xt, ut = ops.cfm_sample(x0.contiguous().float(), x1.contiguous().float(), t.contiguous().float(),
                                self.sigma, eps)
        return (t, xt, ut, eps) if return_noise else (t, xt, ut)


class _MSE(torch.autograd.Function):
    """mean((v-u)^2) with the fused loss/gradient kernel."""

    @staticmethod
    def forward(ctx, v: torch.Tensor, u: torch.Tensor):
        loss, dv = ops.mse_loss(v.detach().contiguous(), u.detach().contiguous(), want_grad=True)
        ctx.save_for_backward(dv)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        (dv,) = ctx.saved_tensors
        return dv * gloss, None


class _eval_mode:
    """``module.eval()`` for the duration of a sampling loop, previous mode restored even when the loop raises.  Only
    the module handed in is touched: wrappers around a LightningModule are plain callables, never ``nn.Module``s that
    would register it as a child and flip its mode behind its back."""

    def __init__(self, net):
        self.net = net if isinstance(net, torch.nn.Module) else None

    def __enter__(self):
        if self.net is not None:
            self.was = self.net.training
            self.net.eval()

    def __exit__(self, *exc):
        if self.net is not None:
            self.net.train(self.was)
        return False


def _prep_source(source_img: torch.Tensor) -> torch.Tensor:
    if source_img.dim() == 3:
        source_img = source_img.unsqueeze(0)
    return source_img.detach().float().contiguous().clone()


@torch.no_grad()
def euler_integrate(f, x: torch.Tensor, num_steps: int) -> torch.Tensor:
    """x <- x + f(t_k, x)/n at t_k = k/n for a callable ``f(t[B], x)``; ``x`` is advanced in place and returned."""
    dt = 1.0 / num_steps
    for k in range(num_steps):
        t = torch.full((x.shape[0],), k * dt, device=x.device, dtype=torch.float32)
        ops.axpy_(x, f(t, x).contiguous(), dt)
    return x


@torch.no_grad()
def euler_generate(net, source_img: torch.Tensor, num_steps: int = 50, graph: bool = False) -> torch.Tensor:
    """x(0) = source, x <- x + v(t_k, x)/n at t_k = k/n, network in eval mode (its mode is restored); returns x(1).
    ``graph=True`` replays one captured Euler step ``num_steps`` times (GraphedVelocity; cached on the network until its
    parameters, the batch shape or the step size change): same kernels, same results, no per-kernel launch cost."""
    with _eval_mode(net):
        x = _prep_source(source_img)
        if not graph:
            return euler_integrate(net, x, num_steps)
        dt = 1.0 / num_steps
        g = getattr(net, "_s2s_euler_graph", None)
        if g is None or not g.matches(net, x, dt):
            if g is not None:
                g.close()                  # the superseded graph goes now, not whenever the cyclic collector finds it
            g = GraphedVelocity(net, x, dt)
            try:
                object.__setattr__(net, "_s2s_euler_graph", g)      # a plain attribute, not a registered sub-module
            except Exception:  # noqa: BLE001 -- a callable that takes no attributes: rebuild next time
                pass
        return g.solve_euler(x, num_steps)


class GraphedVelocity:
    """``v = net(t, x)`` as a replayed hipGraph (one ``hipGraphLaunch`` per evaluation instead of ~60 kernel launches).

    The reference samples ONE tile at a time (src/infer_simple_flowmatching.py:73-83); at batch 1 the eval-mode forward
    is ~60 short kernels and the eager path is bound by host launch time (~2.3 ms of launches per evaluation, round-1
    measurement).  Everything the forward touches has a fixed address inside the capture (torch's graph-private memory
    pool owns the intermediates; weights, packed operands and folded BatchNorm constants are warmed up before capture),
    so a replay recomputes the same kernels on whatever ``t`` / ``x`` currently hold.

    With ``dt`` the capture is a whole Euler step -- ``t <- t_k`` (device-side table walk), ``v``, ``x += dt * v``: a whole fixed-step solve is then ``num_steps``
    graph launches and nothing else.  The network must be in eval mode and its parameters must not change while the
    graph is alive (``GraphedVelocity.matches`` tells a cache when to rebuild)."""

    def __init__(self, net, x_like: torch.Tensor, dt: Optional[float] = None):
        if not x_like.is_cuda:
            raise RuntimeError("stain2stain_amd: GraphedVelocity needs GPU tensors (there is no CPU path)")
        from . import engine
        B = x_like.shape[0]
        self.net, self.dt = net, dt
        self.key = (tuple(x_like.shape), dt, engine.mutation_epoch[0], tuple(p._version for p in _params(net)))
        self.x = x_like.detach().float().contiguous().clone()
        self.t = torch.zeros((B,), dtype=torch.float32, device=x_like.device)
        if dt is not None:
            # node times exactly as the eager loop forms them (k * dt in double, rounded once); the captured step
            # reads table[counter] and advances the counter on the device
            n = int(round(1.0 / dt))
            self._table = torch.tensor([k * dt for k in range(n + 2)], dtype=torch.float32, device=x_like.device)
            self._counter = torch.zeros((1,), dtype=torch.int32, device=x_like.device)
        side = torch.cuda.Stream(device=x_like.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                   # warm-up: packed weights, folded BN constants, LDS attributes
            for _ in range(2):
                net(self.t, self.x)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with ops.capture_graph(self.graph):
            if dt is not None:
                ops.euler_tick_(self.t, self._table, self._counter)
            self.v = net(self.t, self.x).contiguous()
            if dt is not None:
                ops.axpy_(self.x, self.v, dt)

    def close(self) -> None:
        """Release the graph and its private pool now (the object sits in a cycle with its network through
        ``net._s2s_euler_graph``; see CFMTrainer.close)."""
        self.graph = None
        self.v = None
        if getattr(self.net, "_s2s_euler_graph", None) is self:
            try:
                object.__delattr__(self.net, "_s2s_euler_graph")
            except Exception:  # noqa: BLE001
                pass

    def matches(self, net, x: torch.Tensor, dt: Optional[float]) -> bool:
        from . import engine
        return net is self.net and self.key == (tuple(x.shape), dt, engine.mutation_epoch[0],
                                                tuple(p._version for p in _params(net)))

    def __call__(self, t: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        """v(t, x): copies the arguments into the graph's buffers, replays, returns the graph's output buffer (valid
        until the next call)."""
        self.x.copy_(x)
        self.t.copy_(t.to(torch.float32).expand(self.t.shape[0]) if t.dim() <= 1 and t.numel() == 1 else t)
        self.graph.replay()
        return self.v

    def solve_euler(self, source: torch.Tensor, num_steps: int) -> torch.Tensor:
        if self.dt is None or abs(self.dt * num_steps - 1.0) > 1e-9:
            raise ValueError("this graph was captured for a different step size")
        self.x.copy_(source)
        self._counter.zero_()
        for _ in range(num_steps):
            self.graph.replay()
        return self.x.clone()


def _params(net):
    return list(net.parameters()) if isinstance(net, torch.nn.Module) else []


# Dormand-Prince 5(4): nodes, stage matrix (row 7 = the 5th-order weights, first-same-as-last) and error weights
_DP_C = (0.0, 1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0)
_DP_A = ((),
         (1 / 5,),
         (3 / 40, 9 / 40),
         (44 / 45, -56 / 15, 32 / 9),
         (19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729),
         (9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656),
         (35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84))
_DP_E = (-71 / 57600, 0.0, 71 / 16695, -71 / 1920, 17253 / 339200, -22 / 525, 1 / 40)


@torch.no_grad()
def dopri5_generate(net, source_img: torch.Tensor, atol: float = 1e-4, rtol: float = 1e-4, max_steps: int = 1000,
                    return_stats: bool = False, graph: bool = False):
    """``dopri5_integrate`` on a network put in eval mode for the solve (mode restored, also when it raises).
    ``graph=True``: every stage evaluation is a replay of one captured forward (GraphedVelocity)."""
    with _eval_mode(net):
        x = _prep_source(source_img)
        if not graph:
            return dopri5_integrate(net, x, atol, rtol, max_steps, return_stats)
        gv = GraphedVelocity(net, x, None)
        try:
            return dopri5_integrate(lambda t, y: gv(t, y).clone(), x, atol, rtol, max_steps, return_stats)
        finally:
            gv.close()


@torch.no_grad()
def dopri5_integrate(net, y: torch.Tensor, atol: float = 1e-4, rtol: float = 1e-4, max_steps: int = 1000,
                     return_stats: bool = False):
    """Adaptive sampler for a callable ``net(t[B], x)``: x(1) of dx/dt = v(t, x), x(0) = source, eval-mode network, by the Dormand-Prince 5(4) pair.

    The reference integrates with torchdyn's ``NeuralODE(solver="dopri5", atol=1e-4, rtol=1e-4)``
    (conditional_flow_matching.py:157-170); torchdyn is absent, so the step-size control here is the textbook one
    (the same as scipy's ``RK45``: scaled RMS error, factor 0.9 err^(-1/5) clamped to [0.2, 10], no growth right
    after a rejection, Hairer's initial-step heuristic) -- parity with torchdyn's controller is unpinned, the
    solution agrees with any dopri5 to the tolerances.  Stage combinations and the error norm run in HIP
    (``s2s_axpy``, ``s2s_ode_error_norm``); one host read of the error norm per attempted step."""
    B = y.shape[0]

    def f(t, x):
        return net(torch.full((B,), float(t), device=x.device, dtype=torch.float32), x).contiguous()

    def rms_scaled(e, a, b):
        return float(ops.ode_error_norm(e, a, b, atol, rtol))

    t, t1 = 0.0, 1.0
    k1 = f(t, y)
    # initial step (Hairer, Norsett, Wanner II.4; scipy's select_initial_step)
    d0 = rms_scaled(y, y, y)
    d1 = rms_scaled(k1, y, y)
    h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
    ya = y.clone()
    ops.axpy_(ya, k1, h0)
    fa = f(t + h0, ya)
    ops.axpy_(fa, k1, -1.0)
    d2 = rms_scaled(fa, y, y) / h0
    h1 = max(1e-6, h0 * 1e-3) if (d1 <= 1e-15 and d2 <= 1e-15) else (0.01 / max(d1, d2)) ** (1.0 / 5.0)
    h = min(100 * h0, h1, t1 - t)
    n_acc = n_rej = 0
    rejected = False
    for _ in range(max_steps):
        if t >= t1:
            break
        h = min(h, t1 - t)
        ks = [k1]
        for i in range(1, 7):
            yi = y.clone()
            for j, a in enumerate(_DP_A[i]):
                if a != 0.0:
                    ops.axpy_(yi, ks[j], h * a)
            if i == 6:
                y_new = yi
            ks.append(f(t + _DP_C[i] * h, yi))
        err = torch.zeros_like(y)
        for j, e in enumerate(_DP_E):
            if e != 0.0:
                ops.axpy_(err, ks[j], h * e)
        en = rms_scaled(err, y, y_new)
        if en < 1.0:
            factor = 10.0 if en == 0.0 else min(10.0, 0.9 * en ** -0.2)
            if rejected:
                factor = min(1.0, factor)
            t, y, k1 = t + h, y_new, ks[6]
            h *= factor
            rejected = False
            n_acc += 1
        else:
            h *= max(0.2, 0.9 * en ** -0.2)
            rejected = True
            n_rej += 1
    else:
        raise RuntimeError("dopri5_generate: max_steps reached before t = 1")
    return (y, {"accepted": n_acc, "rejected": n_rej}) if return_stats else y


class SolverConfig:
    """Stand-in for the reference's ``solver:`` entry (a partial of torchdyn's ``NeuralODE``,
    configs/model/conditional_flow_matching.yaml:32-38): the attributes ``generate`` reads from it."""

    def __init__(self, solver: str = "dopri5", sensitivity: str = "adjoint", atol: float = 1e-4, rtol: float = 1e-4):
        self.solver, self.sensitivity, self.atol, self.rtol = solver, sensitivity, atol, rtol


def _solver_setting(solver, name: str, default):
    """``solver.<name>`` as the reference reads it (conditional_flow_matching.py:158-163: attribute if present, else
    the default); the keyword of a ``functools.partial`` -- what Hydra's ``_partial_: true`` builds -- is honoured too."""
    if hasattr(solver, name):
        return getattr(solver, name)
    kw = getattr(solver, "keywords", None)
    if isinstance(kw, dict) and name in kw:
        return kw[name]
    return default


def _solve(module, f, source_img: torch.Tensor, num_steps: int, method: Optional[str], atol: Optional[float],
           rtol: Optional[float], graph: bool = False) -> torch.Tensor:
    """The ODE solve shared by every ``generate``: like the reference it refuses to run without a solver, puts the
    module in eval mode and LEAVES it there (``self.eval()``, conditional_flow_matching.py:147-150), integrates with
    the solver's method / tolerances (dopri5, 1e-4 when it names none) and returns the end point x(1).  ``method``
    overrides the solver's choice: "euler" is the ``num_steps`` fixed-step solve of BASELINE.json configs[3]."""
    if module.solver is None:
        raise ValueError("Solver is not initialized. Cannot perform inference.")
    module.eval()
    method = method or _solver_setting(module.solver, "solver", "dopri5")
    atol = _solver_setting(module.solver, "atol", 1e-4) if atol is None else atol
    rtol = _solver_setting(module.solver, "rtol", 1e-4) if rtol is None else rtol
    x = _prep_source(source_img)
    if method == "dopri5":
        if graph:           # every stage evaluation = one replay of the captured forward
            gv = GraphedVelocity(f, x, None)
            return dopri5_integrate(lambda t, y: gv(t, y).clone(), x, atol, rtol)
        return dopri5_integrate(f, x, atol, rtol)
    if method == "euler":
        if graph:
            return GraphedVelocity(f, x, 1.0 / num_steps).solve_euler(x, num_steps)
        return euler_integrate(f, x, num_steps)
    raise ValueError(f"solver method must be 'dopri5' or 'euler', got {method!r}")


class ConditionalFlowMatchingModule(_Base):
    """Same surface as the reference's ConditionalFlowMatchingLitModule (logging hooks left out)."""

    def __init__(self, net: torch.nn.Module, flow_matcher: Optional[ConditionalFlowMatcher] = None, solver=None,
                 optimizer=None, scheduler=None, compile: bool = False, log_images: bool = False,
                 n_images_log: int = 5):
        super().__init__()
        self.net = net
        self.flow_matcher = flow_matcher or ConditionalFlowMatcher(0.0)
        self.solver, self.optimizer, self.scheduler = solver, optimizer, scheduler
        self.log_images, self.n_images_log = log_images, n_images_log

    def forward(self, t: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        return self.net(t, x)

    def model_step(self, batch: Tuple[torch.Tensor, ...]) -> torch.Tensor:
        x0, x1 = batch[:2]
        t, xt, ut = self.flow_matcher.sample_location_and_conditional_flow(x0, x1)
        vt = self.forward(t, xt)
        return _MSE.apply(vt, ut)

    def training_step(self, batch, batch_idx: int) -> torch.Tensor:
        loss = self.model_step(batch)
        self.log("train/loss", loss, on_step=True, on_epoch=True, prog_bar=True, sync_dist=True)
        return loss

    def validation_step(self, batch, batch_idx: int) -> None:
        self.log("val/loss", self.model_step(batch), on_step=False, on_epoch=True, prog_bar=True, sync_dist=True)

    def test_step(self, batch, batch_idx: int) -> None:
        self.log("test/loss", self.model_step(batch), on_step=False, on_epoch=True, prog_bar=True, sync_dist=True)

    def configure_optimizers(self) -> Dict[str, Any]:
        optimizer = self.optimizer(params=self.parameters())
        if self.scheduler is not None:
            return {"optimizer": optimizer,
                    "lr_scheduler": {"scheduler": self.scheduler(optimizer=optimizer), "monitor": "val/loss",
                                     "interval": "epoch", "frequency": 1}}
        return {"optimizer": optimizer}

    @torch.no_grad()
    def generate(self, source_img: torch.Tensor, num_steps: int = 100, method: Optional[str] = None,
                 atol: Optional[float] = None, rtol: Optional[float] = None, graph: bool = False) -> torch.Tensor:
        """x(1) of dx/dt = net(t, x), x(0) = source (conditional_flow_matching.py:133-170): the solver's method --
        adaptive dopri5 at atol = rtol = 1e-4 unless ``self.solver`` says otherwise; ``num_steps`` only defined the
        reference's output grid and plays no role for the end point -- or, with ``method="euler"``, ``num_steps``
        fixed Euler steps.  Raises without a solver and leaves the module in eval mode, as the reference does.
        ``graph=True`` replays the network evaluation as a captured hipGraph (the reference samples one tile at a time,
        infer_simple_flowmatching.py:73-83, where the eager path is launch-bound)."""
        return _solve(self, self.net, source_img, num_steps, method, atol, rtol, graph)


class _WeightedMSE(torch.autograd.Function):
    """sum(w (v-u)^2) / (sum(w) + 1e-8), w = 1 + lam * mask, with the fused loss/gradient kernels."""

    @staticmethod
    def forward(ctx, v: torch.Tensor, u: torch.Tensor, mask: torch.Tensor, lam: float):
        loss, dv = ops.weighted_mse(v.detach().float(), u.detach().float(), mask.detach(), lam, want_grad=True)
        ctx.save_for_backward(dv)
        return loss[0]

    @staticmethod
    def backward(ctx, gloss):
        (dv,) = ctx.saved_tensors
        return dv * gloss, None, None, None


class ROIWeightedFlowMatchingModule(ConditionalFlowMatchingModule):
    """conditional_flow_matching_masked.py:60-91 -- batches are (source, target, mask); pixels inside the mask
    count ``1 + roi_lambda`` times in the flow-matching MSE (``roi_lambda`` attribute, default 10 as :78)."""

    roi_lambda: float = 10.0

    def model_step(self, batch) -> torch.Tensor:
        x0, x1, mask = batch
        t, xt, ut = self.flow_matcher.sample_location_and_conditional_flow(x0, x1)
        return _WeightedMSE.apply(self.forward(t, xt), ut, mask, float(getattr(self, "roi_lambda", 10.0)))


class ROICharbonnierFlowMatchingModule(ConditionalFlowMatchingModule):
    """conditional_flow_matching_ROI_loss.py:64-97 -- flow-matching MSE plus ``lambda_roi`` times the Charbonnier
    distance between xt and x1 inside the mask (a data-only term: it shifts the logged loss, not the gradients)."""

    lambda_roi: float = 1.0

    def model_step(self, batch) -> torch.Tensor:
        x0, x1, mask = batch
        t, xt, ut = self.flow_matcher.sample_location_and_conditional_flow(x0, x1)
        loss_fm = _MSE.apply(self.forward(t, xt), ut)
        roi = ops.charbonnier_roi(xt, x1.float(), mask, 1e-3, 1e-8)[0]
        return loss_fm + float(getattr(self, "lambda_roi", 1.0)) * roi


class MaskConditionedFlowMatchingModule(ConditionalFlowMatchingModule):
    """conditional_flow_matching_conditional_mask.py:54-82 (``mask_toggle=False``) and
    conditional_flow_matching_conditional_toggle_mask.py:54-103 (``mask_toggle=True``: the training step zeroes the
    mask with probability 1/2, drawn with ``torch.rand(1).item()`` like the reference).  The mask is the fourth
    input channel of ``net`` (``in_channels: 4``)."""

    def __init__(self, *args, mask_toggle: bool = False, **kwargs):
        super().__init__(*args, **kwargs)
        self.mask_toggle = mask_toggle

    def forward(self, t: torch.Tensor, x: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        return self.net(t, torch.cat([x, mask.to(x.dtype)], dim=1))

    def model_step(self, batch, use_mask_toggle: bool = False) -> torch.Tensor:
        x0, x1, mask = batch
        if use_mask_toggle and torch.rand(1).item() < 0.5:
            mask = torch.zeros_like(mask)
        t, xt, ut = self.flow_matcher.sample_location_and_conditional_flow(x0, x1)
        return _MSE.apply(self.forward(t, xt, mask), ut)

    def training_step(self, batch, batch_idx: int) -> torch.Tensor:
        loss = self.model_step(batch, use_mask_toggle=self.mask_toggle)
        self.log("train/loss", loss, on_step=True, on_epoch=True, prog_bar=True, sync_dist=True)
        return loss

    @torch.no_grad()
    def generate(self, source_img: torch.Tensor, mask: torch.Tensor, num_steps: int = 100,
                 method: Optional[str] = None, atol: Optional[float] = None, rtol: Optional[float] = None) -> torch.Tensor:
        """conditional_flow_matching_conditional_mask.py:143-199: the mask rides along as a constant fourth channel."""
        if source_img.dim() == 3:
            source_img, mask = source_img.unsqueeze(0), mask.unsqueeze(0)
        return _solve(self, lambda t, x: self.forward(t, x, mask), source_img, num_steps, method, atol, rtol)


class ClassConditionalFlowMatchingModule(ConditionalFlowMatchingModule):
    """class_conditional_flow_matching.py:39-71,130-190 -- batches are (source, target, target_label); the label
    conditions the network call ``net(t, x, y=y)``; ``generate(source, target_class, num_steps)``."""

    def forward(self, t: torch.Tensor, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        return self.net(t, x, y=y)

    def model_step(self, batch) -> torch.Tensor:
        x0, x1, label = batch
        t, xt, ut = self.flow_matcher.sample_location_and_conditional_flow(x0, x1)
        return _MSE.apply(self.forward(t, xt, label.long()), ut)

    @torch.no_grad()
    def generate(self, source_img: torch.Tensor, target_class, num_steps: int = 100, method: Optional[str] = None,
                 atol: Optional[float] = None, rtol: Optional[float] = None) -> torch.Tensor:
        """class_conditional_flow_matching.py:130-190: ``target_class`` an int or a [B] tensor."""
        if source_img.dim() == 3:
            source_img = source_img.unsqueeze(0)
        B, dev = source_img.shape[0], source_img.device
        y = (torch.tensor([target_class] * B, device=dev) if isinstance(target_class, int)
             else target_class.to(dev))

        def f(t, x):
            return self.net(t, x, y=(y.expand(x.shape[0]) if y.dim() == 0 else y[:x.shape[0]]))

        return _solve(self, f, source_img, num_steps, method, atol, rtol)


class _SegLoss(torch.autograd.Function):
    """dw * Dice(sigmoid(z), g) + (1-dw) * BCEWithLogits(z, g) with the fused two-pass kernel."""

    @staticmethod
    def forward(ctx, z: torch.Tensor, g: torch.Tensor, smooth: float, dice_weight: float):
        out, dz = ops.seg_loss(z.detach().float(), g.detach(), smooth, dice_weight, want_grad=True)
        ctx.save_for_backward(dz)
        ctx.mark_non_differentiable(out[1], out[2])
        return out[0], out[1], out[2]

    @staticmethod
    def backward(ctx, gseg, gdice, gbce):
        (dz,) = ctx.saved_tensors
        return dz * gseg, None, None, None


class _SegLossMulticlass(torch.autograd.Function):
    """dw * softmax-Dice + (1-dw) * CrossEntropy on class-index targets."""

    @staticmethod
    def forward(ctx, z: torch.Tensor, target: torch.Tensor, ignore_index: int, smooth: float, dice_weight: float):
        out, dz = ops.seg_loss_multiclass(z.detach().float(), target, ignore_index, smooth, dice_weight,
                                          want_grad=True)
        ctx.save_for_backward(dz)
        ctx.mark_non_differentiable(out[1], out[2])
        return out[0], out[1], out[2]

    @staticmethod
    def backward(ctx, gseg, gdice, gce):
        (dz,) = ctx.saved_tensors
        return dz * gseg, None, None, None, None


class MultiTaskFlowMatchingModule(_Base):
    """Same surface as the reference's MultiTaskFlowMatchingLitModule
    (src/models/conditional_flow_matching_multitask.py:57-257,391-417): shared encoder, flow head A, mask head B,
    L = L_FM + seg_loss_weight * (dice_weight * Dice + (1 - dice_weight) * BCE).  Logging hooks left out."""

    def __init__(self, encoder, flow_decoder, seg_decoder, flow_matcher: Optional[ConditionalFlowMatcher] = None,
                 solver=None, optimizer=None, scheduler=None, compile: bool = False, log_images: bool = False,
                 seg_loss_weight: float = 1.0, dice_weight: float = 0.5, n_images_log: int = 5,
                 time_emb_dim: int = 256, num_classes: Optional[int] = None, ignore_index: int = -100):
        """``num_classes`` None: binary mask head, Dice + BCE (conditional_flow_matching_multitask.py).  An integer
        selects the multiclass form, softmax Dice + CrossEntropy on class-index masks
        (conditional_flow_matching_multitask_multiclassloss.py:92-159)."""
        super().__init__()
        self.num_classes, self.ignore_index = num_classes, ignore_index
        from .components import TimeEmbedding
        self.encoder, self.flow_decoder, self.seg_decoder = encoder, flow_decoder, seg_decoder
        self.time_embedding = TimeEmbedding(time_emb_dim)
        self.flow_matcher = flow_matcher or ConditionalFlowMatcher(0.0)
        self.solver, self.optimizer, self.scheduler = solver, optimizer, scheduler
        self.seg_loss_weight, self.dice_weight = seg_loss_weight, dice_weight
        self.dice_smooth = 1.0

    def forward_flow(self, t: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        bottleneck, skips = self.encoder(x)
        return self.flow_decoder(bottleneck, skips, self.time_embedding(t))

    def forward_segmentation(self, x: torch.Tensor) -> torch.Tensor:
        bottleneck, skips = self.encoder(x)
        return self.seg_decoder(bottleneck, skips)

    def compute_segmentation_loss(self, pred_mask: torch.Tensor, target_mask: torch.Tensor):
        if self.num_classes is not None:
            if target_mask.dim() == 4 and target_mask.shape[1] == 1:
                target_mask = target_mask.squeeze(1)
            seg, dice, ce = _SegLossMulticlass.apply(pred_mask, target_mask.long(), self.ignore_index,
                                                     self.dice_smooth, self.dice_weight)
            return seg, {"dice": dice, "ce": ce, "seg_total": seg}
        seg, dice, bce = _SegLoss.apply(pred_mask, target_mask.float(), self.dice_smooth, self.dice_weight)
        return seg, {"dice": dice, "bce": bce, "seg_total": seg}

    def model_step(self, batch):
        source_img, target_img, gt_mask = batch
        t, xt, ut = self.flow_matcher.sample_location_and_conditional_flow(source_img, target_img)
        flow_loss = _MSE.apply(self.forward_flow(t, xt), ut)
        seg_loss, d = self.compute_segmentation_loss(self.forward_segmentation(source_img), gt_mask)
        total = flow_loss + self.seg_loss_weight * seg_loss
        second = "ce" if self.num_classes is not None else "bce"
        return total, {"total": total, "flow": flow_loss, "seg": seg_loss, "seg_dice": d["dice"],
                       "seg_" + second: d[second]}

    def training_step(self, batch, batch_idx: int) -> torch.Tensor:
        loss, d = self.model_step(batch)
        for k, v in d.items():
            self.log(f"train/{k}_loss", v, on_step=True, on_epoch=True, prog_bar=(k == "total"), sync_dist=True)
        return loss

    def configure_optimizers(self) -> Dict[str, Any]:
        params = (list(self.encoder.parameters()) + list(self.flow_decoder.parameters())
                  + list(self.seg_decoder.parameters()))
        optimizer = self.optimizer(params=params)
        if self.scheduler is not None:
            return {"optimizer": optimizer,
                    "lr_scheduler": {"scheduler": self.scheduler(optimizer=optimizer), "monitor": "val/loss",
                                     "interval": "epoch", "frequency": 1}}
        return {"optimizer": optimizer}

    @torch.no_grad()
    def generate(self, source_img: torch.Tensor, num_steps: int = 100, method: Optional[str] = None,
                 atol: Optional[float] = None, rtol: Optional[float] = None):
        """(generated image, mask probabilities / class map) like the reference's generate (:419-484): the module goes
        to eval mode and stays there, the mask head runs once on the source, the flow is integrated with the solver's
        method (dopri5 at 1e-4 by default; ``method="euler"`` = ``num_steps`` fixed steps)."""
        if self.solver is None:
            raise ValueError("Solver is not initialized. Cannot perform inference.")
        self.eval()
        if source_img.dim() == 3:
            source_img = source_img.unsqueeze(0)
        logits = self.forward_segmentation(source_img)
        if self.num_classes is not None:       # class map, as the multiclass reference returns (:536-537)
            pred_mask = torch.argmax(torch.softmax(logits, dim=1), dim=1, keepdim=True)
        else:
            pred_mask = torch.sigmoid(logits)
        return _solve(self, self.forward_flow, source_img, num_steps, method, atol, rtol), pred_mask
