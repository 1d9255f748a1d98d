"""The PGGAN / WGAN-GP training-step driver on MI355X.

Restates the reference's hot loop (/root/reference/train.py:350-394: n_critic x [D loss + gradient penalty, backward,
Adam], then G loss, backward, Adam) as an engine object instead of module-level script code:

  * every parameter of a net is re-homed into ONE flat fp32 buffer, and every gradient into another, so that
    zeroing gradients is one memset, the data-parallel exchange is one RCCL all-reduce per net per step, and
    Adam is one fused kernel launch (`ngan_adam_step`) instead of ~45 ATen foreach chains;
  * hyper-parameters, step counts and the fade-in alpha live in device memory and latents can be drawn on the
    GPU, so a whole iteration can be captured into a HIP graph and replayed with no host work;
  * the critic's parameter gradients that the reference computes and throws away in the G step
    (train.py:384) are not computed.

The epoch-level semantics (LR schedule train.py:232-265, per-epoch alpha advance and growth train.py:318-333) are
provided by `lr_schedule` / `PGGANTrainer.start_epoch`; data loading, plotting and checkpoints are out of scope.
"""
import math
import os

import numpy as np
import torch
import torch.distributed as dist

from . import _C, ops
from .loss_functions import D_W_loss, D_grad_pen_loss, G_W_loss
from .utils import sample_latent_vec, sample_latent_vec_device

ADAM_CHUNK = 4096  # elements per work item of ngan_adam_step (must match csrc/adam.hip)
SEG_ALIGN = 64     # parameters start on 256-byte boundaries inside the flat buffers


class FlatParams:
    """All parameters of a net as views into one flat buffer (plus flat grad / Adam state buffers)."""

    def __init__(self, net: torch.nn.Module):
        self.params = list(net.parameters())
        self.names = [getattr(p, "_ngan_name", n) for n, p in net.named_parameters()]   # stable across growth stages
        assert self.params, "network has no parameters"
        dev = self.params[0].device  # CPU is allowed for host-logic tests; the fused Adam kernel itself needs the GPU
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + SEG_ALIGN - 1) // SEG_ALIGN * SEG_ALIGN
        self.offsets, self.total = offs, total
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(total, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(total, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(total, device=dev, dtype=torch.float32)
        for p, off in zip(self.params, offs):
            n = p.numel()
            self.flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + n].view(p.shape)
            p.grad = self.grad[off:off + n].view(p.shape)
        self.index = {id(p): i for i, p in enumerate(self.params)}
        # device-side tables for ngan_adam_step
        self.seg_off = torch.tensor(offs, dtype=torch.int64, device=dev)
        self.seg_len = torch.tensor([p.numel() for p in self.params], dtype=torch.int64, device=dev)
        self.seg_active = torch.zeros(len(self.params), dtype=torch.int32, device=dev)
        self.seg_step = torch.zeros(len(self.params), dtype=torch.float32, device=dev)
        cseg, coff = [], []
        for i, p in enumerate(self.params):
            for o in range(0, p.numel(), ADAM_CHUNK):
                cseg.append(i)
                coff.append(o)
        self.chunk_seg = torch.tensor(cseg, dtype=torch.int32, device=dev)
        self.chunk_off = torch.tensor(coff, dtype=torch.int64, device=dev)

    def set_active(self, active_params):
        """Mark which tensors receive gradients at the current stage (torch's Adam skips .grad=None tensors)."""
        flags = torch.zeros(len(self.params), dtype=torch.int32)
        for p in active_params:
            flags[self.index[id(p)]] = 1
        self.seg_active.copy_(flags, non_blocking=True)
        self.active_host = flags.numpy().copy()

    def zero_grad(self):
        self.grad.zero_()

    def ensure_grad_views(self):
        """Re-attach .grad views if something (e.g. Module.zero_grad) detached them."""
        for p, off in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                p.grad = self.grad[off:off + p.numel()].view(p.shape)


class FusedAdam:
    """optim.Adam(params, lr, betas=(beta1, 0.999)) semantics of train.py:224-225 in one kernel launch."""

    def __init__(self, flat: FlatParams, lr=1e-4, betas=(0.5, 0.999), eps=1e-8):
        self.flat = flat
        # {lr, beta1, beta2, eps, grad_scale, 1 - beta1, 1 - beta2, ln beta1, ln beta2}: include/ngan.h, ngan_adam_step
        self.hyper_host = [float(lr), float(betas[0]), float(betas[1]), float(eps), 1.0, 1.0 - float(betas[0]), 1.0 - float(betas[1]),
                           *(math.log(float(b)) if float(b) > 0 else float("-inf") for b in betas)]     # (beta = 0: 1 - 0^t = 1)
        self.hyper = torch.tensor(self.hyper_host, dtype=torch.float32, device=flat.flat.device)
        self.param_groups = [{"lr": float(lr)}]  # same handle the reference's update_lr() writes to (train.py:253-265)

    def set_lr(self, lr):
        self.param_groups[0]["lr"] = float(lr)
        self._push()

    def set_grad_scale(self, s):
        self.hyper_host[4] = float(s)
        self._push()

    def _push(self):
        self.hyper_host[0] = float(self.param_groups[0]["lr"])
        self.hyper.copy_(torch.tensor(self.hyper_host, dtype=torch.float32), non_blocking=True)

    def step(self, stem_factors=None):
        """stem_factors: (z, gc, s2, c, scale) of the generator stem (tensor 0 of the flat buffer) when its gradient was NOT stored
        (PGGANTrainer.fused_stem): its chunks are left out of the flat launch and `ngan_linear_wgrad_adam` forms the gradient from
        the factors and applies the same update in its epilogue."""
        f = self.flat
        if self.hyper_host[0] != self.param_groups[0]["lr"]:
            self._push()
        n0 = 0
        if stem_factors is not None:
            assert f.active_host[0] == 1, "the stem is active at every stage"
            n0 = (f.params[0].numel() + ADAM_CHUNK - 1) // ADAM_CHUNK
        n_chunks = int(f.chunk_seg.numel()) - n0
        _C.call("ngan_adam_step", f.flat, f.grad, f.exp_avg, f.exp_avg_sq, f.seg_off, f.seg_len, f.seg_active, f.seg_step,
                len(f.params), f.chunk_seg[n0:], f.chunk_off[n0:], n_chunks, self.hyper, self.hyper.numel())      # also advances every active step count
        if stem_factors is not None:
            z, gc, s2, c, scale = stem_factors
            n = f.params[0].numel()
            _C.call(ops._k("ngan_linear_wgrad_adam", gc), z, gc, f.flat[:n], f.exp_avg[:n], f.exp_avg_sq[:n], f.seg_step[:1], self.hyper, self.hyper.numel(),
                    z.shape[0], z.shape[1], s2, c, float(scale))
        self.repack()

    def repack(self):
        """this net's packed conv weights are stale after a step: re-pack them (and only them -- the other net's copies are still
        valid) with a single launch"""
        ops.refresh_packed(owner=id(self), params=self.flat.params)


def exchange_gradients(flat: FlatParams, world: int, group=None, force: bool = False):
    """Data-parallel gradient exchange: ONE all-reduce(SUM) over the net's flat gradient buffer (RCCL over xGMI on the
    GPU; gloo in the CPU tests).  The 1/world factor is not applied here -- it is folded into the fused Adam kernel
    (`FusedAdam.set_grad_scale`), so the averaged gradient never makes an extra pass through HBM.  No op couples
    samples (PixelNorm is per pixel, every loss is a batch mean), so N ranks x batch b with this exchange equals
    one rank x batch N*b up to fp32 summation order."""
    if world > 1 or force:   # force: exercise the collective on a one-rank group (bench.py --force-dist)
        dist.all_reduce(flat.grad, op=dist.ReduceOp.SUM, group=group)


class StemGradExchange:
    """Data-parallel exchange for the generator stem (Linear_normalized, 16.8 M of G's 17.1 M parameters at the default
    widths).  Its gradient is scale * sum_b gc[b] (x) z[b], a rank-B outer product: instead of all-reducing the 67 MB
    result, every rank all-gathers the factors (gc: B x C*S floats, z: B x K) and forms the FULL-batch gradient locally
    with the same kernel (`ngan_linear_wgrad` over world*B samples).  The result equals the sum over ranks of the per-rank
    gradients, i.e. what the all-reduce would have produced, so Adam's 1/world scaling applies unchanged."""

    def __init__(self, weight, world, group=None, wgrad_fn=None):
        self.weight, self.world, self.group = weight, world, group
        self.captured = None     # kept after finish(): under graph replay the same (static) tensors are refilled every step
        self.factors = None      # (z, gc, s2, c, scale) over ALL ranks' samples after finish(): what FusedAdam.step(stem_factors=) takes
        self._gathered = {}      # gather buffers per factor shape: captured Adam launches read them, so they must not move
        self.wgrad_fn = wgrad_fn or (lambda zs, gs, out, n, k, s2, c, scale:
                                     _C.call(ops._k("ngan_linear_wgrad", gs), zs, gs, out, n, k, s2, c, float(scale)))

    def sink(self, z, gc, weight, s2, c, scale):
        assert weight is self.weight
        self.captured = (z, gc, s2, c, scale)

    def finish(self, run_collectives=None, materialize=True, also=None):
        """all-gather the factors and (materialize) write the full-batch gradient into weight.grad (call after backward).
        `run_collectives(fn)` executes the collectives (the step driver passes its communication-stream runner,
        PGGANTrainer._on_comm_stream).  With materialize=False the gradient is never formed: `self.factors` goes to the fused Adam.
        `also`: a further collective of the caller (the all-reduce of the generator's other gradients) issued in the SAME
        communication-stream section: one hand-over between the streams per exchange instead of two."""
        run = run_collectives or (lambda fn: fn())
        if self.captured is None:
            if also is not None:
                run(also)
            return
        z, gc, s2, c, scale = self.captured
        b, k = z.shape
        if self.world > 1:
            key = (self.world * b, k) + tuple(gc.shape[1:])
            if key not in self._gathered:
                self._gathered[key] = (torch.empty((self.world * b, k), device=z.device, dtype=z.dtype),
                                       torch.empty((self.world * b,) + tuple(gc.shape[1:]), device=gc.device, dtype=gc.dtype))
            zs, gs = self._gathered[key]
            zc, gcc = z.contiguous(), gc.contiguous()

            def gather():
                dist.all_gather_into_tensor(zs, zc, group=self.group)
                dist.all_gather_into_tensor(gs, gcc, group=self.group)
                if also is not None:
                    also()
            run(gather)
        else:
            if also is not None:
                run(also)
            zs, gs = z.contiguous(), gc.contiguous()
        self.factors = (zs, gs, s2, c, scale)
        if materialize:
            self.wgrad_fn(zs, gs, self.weight.grad, zs.shape[0], k, s2, c, scale)


def active_parameters(net):
    """Parameters that take part in `forward` at the net's current stage (everything else has .grad None in torch)."""
    mods = [net.layers]
    fading = net.alpha_value() < 1
    if hasattr(net, "ToIm"):
        mods.append(net.ToIm)
        if fading:
            mods += [net.conv_block_list[0], net.ToIm_list[0]]
    else:
        mods.append(net.FromIm)
        if fading:
            mods += [net.conv_block_list[-1], net.FromIm_list[-1]]
    out = []
    for m in mods:
        out += list(m.parameters())
    return out


def lr_schedule(epoch, base_lr, transit_sch, n_epochs, total_decay=1 / 100):
    """Learning rate at `epoch` (reference update_lr, train.py:238-265): reset to base at every phase boundary,
    exponential decay by `total_decay` over the first half of each phase, then held.  Returns None where the
    reference leaves the optimiser's current value untouched (second half of a phase)."""
    bounds = [0] + list(transit_sch) + [n_epochs]
    if epoch in bounds:
        return base_lr
    phase = sum(epoch > t for t in transit_sch)
    phase_len = bounds[phase + 1] - bounds[phase]
    since = epoch - bounds[phase]
    if since <= phase_len / 2:
        gamma = np.exp(np.log(total_decay) / (phase_len / 2))
        return base_lr * (gamma ** since)
    return None


class PGGANTrainer:
    """One object per process (= per GPU).  `train_iteration(real)` is train.py:356-385 with sim_loss off."""

    def __init__(self, generator, discriminator, learning_rate=1e-4, beta1=0.5, grad_pen_lambda=10.0, drift_epsilon=0.001,
                 n_critic=1, alpha_step=1e-4, process_group=None, device_latents=False, fused_stem=None):
        self.G, self.D = generator, discriminator
        self.device = next(generator.parameters()).device
        self.n_critic = n_critic
        self.alpha_step = alpha_step
        self.device_latents = device_latents
        self.flat_g, self.flat_d = FlatParams(generator), FlatParams(discriminator)
        self.opt_g = FusedAdam(self.flat_g, learning_rate, (beta1, 0.999))
        self.opt_d = FusedAdam(self.flat_d, learning_rate, (beta1, 0.999))
        self.d_loss = D_W_loss(generator, discriminator, drift_epsilon=drift_epsilon, check_nan=False)
        self.gp_loss = D_grad_pen_loss(generator, discriminator, Lambda=grad_pen_lambda)
        self.g_loss = G_W_loss(generator, discriminator, check_nan=False)
        self.group = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.stem = None
        self._stem_grad_skipped = self._stem_sink_active = self._stem_for_exchange = False
        if self.world > 1:
            self.opt_g.set_grad_scale(1.0 / self.world)
            self.opt_d.set_grad_scale(1.0 / self.world)
            self.enable_stem_exchange()
        # The generator stem's weight gradient is not stored when the stem qualifies (GPU, latent_dim a multiple of 16, at most 512):
        # g_step hands its factors to the Adam launch (FusedAdam.step).  g_compute on its own still stores it.  fused_stem=False: never.
        self.fused_stem = False
        if fused_stem or (fused_stem is None and self.device.type == "cuda" and ops._diag_env("NGAN_FUSED_STEM_ADAM", "1") != "0"):
            self.enable_fused_stem()
        self.force_exchange = False
        self.last_z_g = None
        self._one = torch.ones((), device=self.device)
        # collectives of the RCCL backend run on a stream of their own (see _on_comm_stream); created on first need
        self._comm_stream = None
        self.comm_timing = None     # bench.py: a list that receives (tag, start event, end event) of every gradient exchange
        if self.device.type == "cuda" and dist.is_available() and dist.is_initialized() and dist.get_backend(process_group) == "nccl":
            self._comm_stream = torch.cuda.Stream(device=self.device)
        self._graphs = {}          # input shape -> captured graphs of this stage (capture / replay)
        self._graph = self._entry = None
        self.refresh_stage()
        ops.bump_weight_epoch()

    def enable_stem_exchange(self, for_exchange=True):
        """Exchange the stem's gradient as gathered factors; the all-reduce then skips its segment of the flat buffer."""
        self._stem_for_exchange = self._stem_for_exchange or for_exchange
        if self.stem is not None:
            return
        first = self.G.layers[0]
        if hasattr(first, "weight") and first.weight.dim() == 2 and self.flat_g.index[id(first.weight)] == 0:
            self.stem = StemGradExchange(first.weight, self.world, self.group)
            self._stem_elems = (first.weight.numel() + SEG_ALIGN - 1) // SEG_ALIGN * SEG_ALIGN

    def enable_fused_stem(self):
        if self.stem is None:
            self.enable_stem_exchange(for_exchange=False)
        k = self.stem.weight.shape[1] if self.stem is not None else 0
        self.fused_stem = self.stem is not None and k % 16 == 0 and 0 < k <= 512
        return self.fused_stem

    # ---- stage bookkeeping -------------------------------------------------------------------------------
    def refresh_stage(self):
        """Call after any growth event (increase_resolution / a transition completing)."""
        self.flat_g.set_active(active_parameters(self.G))
        self.flat_d.set_active(active_parameters(self.D))
        self._graphs = {}          # the captured graphs belong to the previous stage's module structure
        self._graph = self._entry = None

    def start_epoch(self, epoch, transit_sch=()):
        """Per-epoch alpha advance and growth (train.py:318-333).  Returns True if the structure changed."""
        changed = False
        ga, da = self.G.alpha_value(), self.D.alpha_value()
        if ga < 1 and da < 1:
            self.G.advance_transition(self.alpha_step)
            self.D.advance_transition(self.alpha_step)
            changed = self.G.alpha_value() >= 1
        elif ga < 1:
            raise Exception('The networks are not synchronized. Gen_alpha={:.3f}, Disc_alpha={:.3f}'.format(ga, da))
        if epoch in transit_sch:
            self.G.increase_resolution()
            self.D.increase_resolution()
            changed = True
        if changed:
            self.refresh_stage()
        return changed

    # ---- the two half-steps ---------------------------------------------------------------------------------
    def _latent(self, batch, z):
        if z is not None:
            return z
        if self.device_latents:
            return sample_latent_vec_device((batch, self.G.latent_dim), self.device)
        return sample_latent_vec((batch, self.G.latent_dim), device=self.device)

    def _on_comm_stream(self, fn, tag="exchange"):
        """Run the collectives of `fn` on this trainer's communication stream, ordered after the work already queued on the current
        stream and before whatever the current stream does next.

        Why a stream of their own (the abort on record: gpurun_out/fd2.log of round 1, tools/capture_event_probe.py reproduces it):
        a synchronous c10d collective records its completion event on the stream it was issued on, and the process group's
        watchdog thread polls that event with hipEventQuery until a poll finds it complete (~100 ms period).  HIP refuses
        hipEventQuery on an event whose last-recorded stream is part of an ACTIVE capture (hipErrorCapturedEvent) and the watchdog
        turns that into std::terminate.  Streams of the training code do become part of captures: a captured backward pass forks
        the stream on which a parameter's AccumulateGrad node was created into the capture.  So a collective issued on such a stream
        shortly before capture() (warm-up iterations, the exchanges between captured segments, the last replayed step before a growth
        event's re-capture) killed the process whenever the watchdog had not polled it yet.  Nothing but collectives ever runs on the
        communication stream -- no autograd node is created on it, no capture is begun on it, no captured stream waits on it -- so it
        can never be part of a capture and the watchdog may poll its events at any time.  (CPU tensors / the gloo rehearsal have no
        such event and run in line.)"""
        timed = self.comm_timing is not None and self.device.type == "cuda"
        if self._comm_stream is None:
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                self.comm_timing.append((tag, e0, e1))
                return None
            return fn()
        cur = torch.cuda.current_stream()
        self._comm_stream.wait_stream(cur)
        with torch.cuda.stream(self._comm_stream):
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            fn()
            if timed:
                e1.record()
                self.comm_timing.append((tag, e0, e1))
        cur.wait_stream(self._comm_stream)

    def _exchange(self, flat):
        if flat is self.flat_g and self._stem_sink_active:
            # the stem occupies the head of G's flat buffer: gather its factors, all-reduce only the tail
            tail = None
            if self.world > 1 or self.force_exchange:
                tail = lambda: dist.all_reduce(flat.grad[self._stem_elems:], op=dist.ReduceOp.SUM, group=self.group)   # noqa: E731
            self.stem.finish(lambda fn: self._on_comm_stream(fn, "generator"), materialize=not self._stem_grad_skipped, also=tail)
            return
        if self.world > 1 or self.force_exchange:
            self._on_comm_stream(lambda: exchange_gradients(flat, self.world, self.group, force=self.force_exchange),
                                 "critic" if flat is self.flat_d else "generator")

    def d_compute(self, real, z_d=None, z_gp=None, eps=None):
        """D half-step up to (and including) the backward pass: gradients end up in flat_d.grad."""
        b = real.size(0)
        self.flat_d.ensure_grad_views()
        self.flat_d.zero_grad()  # Discriminator_net.zero_grad(), train.py:357
        # the two detached generator passes of the D step (loss_functions.py:26, 167) run as one batch-2b pass; with the penalty
        # switched off (Lambda = 0, the reference CLI's default) the reference draws no second latent batch, and neither does this
        with_gp = self.gp_loss.Lambda > 0
        with torch.no_grad():
            if z_d is None and z_gp is None and self.device_latents:
                zz = self._latent((2 if with_gp else 1) * b, None)      # both latent batches in one draw: no concatenation
            else:
                zs = [self._latent(b, z_d)] + ([self._latent(b, z_gp)] if with_gp else [])
                zz = torch.cat(zs, dim=0) if with_gp else zs[0]
            fakes = self.G(zz)
        loss, s_real, s_fake = self.d_loss(real, fake_images=fakes[:b])  # train.py:358
        gp = self.gp_loss(real, x_tilde=fakes[b:] if with_gp else None, epsilon=eps)  # train.py:361
        # train.py:362, 365: D_loss += gp; D_loss.backward().  The two terms share no graph node (separate critic passes), so the sum's
        # backward is the two backwards; they run one after the other so that every critic parameter receives its contributions in
        # a fixed order (penalty terms, then the W-loss term): autograd's node order inside ONE run over both graphs is not
        # reproducible from iteration to iteration (ops.flush_wgrad), and the step driver is meant to be bit-reproducible.
        with ops.deferred_wgrad():   # weight-gradient slabs of the whole pass are reduced by one launch at the end
            if gp.requires_grad:
                gp.backward(gradient=self._one)     # (an explicit root gradient: autograd otherwise fills a ones tensor per call)
            loss.backward(gradient=self._one)
        loss = loss.detach() + gp.detach()
        return {"D_loss": loss.detach(), "score_real": s_real.detach(), "score_fake": s_fake.detach(), "D_grad_pen": gp.detach()}

    def d_step(self, real, z_d=None, z_gp=None, eps=None):
        stats = self.d_compute(real, z_d, z_gp, eps)
        self._exchange(self.flat_d)
        self.opt_d.step()  # train.py:366
        return stats

    def g_compute(self, real, z=None, skip_stem_grad=False):
        """skip_stem_grad (g_step with fused_stem): the stem's gradient is neither zeroed nor stored -- its factors wait in self.stem"""
        b = real.size(0)
        self.flat_g.ensure_grad_views()
        self._stem_grad_skipped = bool(skip_stem_grad and self.fused_stem)
        if self._stem_grad_skipped:
            self.flat_g.grad[self._stem_elems:].zero_()
        else:
            self.flat_g.zero_grad()  # Generator_net.zero_grad(), train.py:375
        d_params = self.flat_d.params
        for p in d_params:  # the reference also back-propagates into the critic's weights here and discards the result
            p.requires_grad_(False)
        try:
            loss, self.last_z_g = self.g_loss(real, z=self._latent(b, z))  # train.py:376
            # the stem hands over factors instead of a gradient when they are exchanged (data parallel) or go straight to Adam
            self._stem_sink_active = self.stem is not None and (self._stem_for_exchange or self._stem_grad_skipped)
            ops.linear_grad_sink = self.stem.sink if self._stem_sink_active else None
            try:
                with ops.deferred_wgrad():
                    loss.backward(gradient=self._one)  # train.py:384
            finally:
                ops.linear_grad_sink = None
        finally:
            for p in d_params:
                p.requires_grad_(True)
        return {"G_loss": loss.detach()}

    def materialize_stem_grad(self):
        """After a g_step that skipped the stem's gradient (fused_stem), write it into the stem weight's .grad from that step's
        factors -- for inspection (gradient norms, the parity tests); the update itself never needs it."""
        if self._stem_grad_skipped and self.stem is not None and self.stem.factors is not None:
            zs, gs, s2, c, scale = self.stem.factors
            self.stem.wgrad_fn(zs, gs, self.stem.weight.grad, zs.shape[0], zs.shape[1], s2, c, scale)

    def g_adam(self):
        if self._stem_grad_skipped:
            # the stem's .grad was neither zeroed nor written by this step's g_compute: without factors the flat Adam launch would
            # apply whatever an older step left there
            assert self.stem is not None and self.stem.factors is not None, \
                "g_compute skipped the stem's gradient but no factors arrived: call _exchange(flat_g) between g_compute and g_adam"
        self.opt_g.step(self.stem.factors if self._stem_grad_skipped else None)  # train.py:385

    @property
    def stem_grad_is_current(self):
        """False after a step that handed the stem's factors to Adam instead of storing its gradient (g_step with fused_stem): the
        stem weight's .grad then still holds an OLDER step's values -- 16.8 M of the generator's 17.1 M gradient elements.  Gradient
        norms, clipping or logging helpers must call materialize_stem_grad() first (or check this flag)."""
        return not self._stem_grad_skipped

    def g_step(self, real, z=None):
        stats = self.g_compute(real, z, skip_stem_grad=True)
        self._exchange(self.flat_g)
        self.g_adam()
        return stats

    def train_iteration(self, real, z_d=None, z_gp=None, eps=None, z_g=None):
        stats = {}
        for _ in range(self.n_critic):  # train.py:356
            stats.update(self.d_step(real, z_d, z_gp, eps))
        if self.n_critic == 0:          # adaptive critic schedule chose no critic step: losses for monitoring only (train.py:369-372)
            stats.update(self.d_compute(real, z_d, z_gp, eps))
        stats.update(self.g_step(real, z_g))
        return stats

    # ---- optimiser state for checkpoints (optional extra key; the reference saves none) ---------------------------
    def optimizer_state(self):
        out = {}
        for tag, flat, opt in (("G", self.flat_g, self.opt_g), ("D", self.flat_d, self.opt_d)):
            out[tag] = {"names": list(flat.names), "lr": opt.param_groups[0]["lr"],
                        "step": flat.seg_step.detach().cpu().clone(),
                        "exp_avg": {n: flat.exp_avg[o:o + p.numel()].detach().cpu().clone().view(p.shape)
                                    for n, p, o in zip(flat.names, flat.params, flat.offsets)},
                        "exp_avg_sq": {n: flat.exp_avg_sq[o:o + p.numel()].detach().cpu().clone().view(p.shape)
                                       for n, p, o in zip(flat.names, flat.params, flat.offsets)}}
        return out

    def load_optimizer_state(self, state):
        for tag, flat, opt in (("G", self.flat_g, self.opt_g), ("D", self.flat_d, self.opt_d)):
            st = state[tag]
            saved_step = dict(zip(st["names"], st["step"].tolist()))
            steps = flat.seg_step.detach().cpu().clone()
            for i, (n, p, o) in enumerate(zip(flat.names, flat.params, flat.offsets)):
                if n in st["exp_avg"] and tuple(st["exp_avg"][n].shape) == tuple(p.shape):
                    flat.exp_avg[o:o + p.numel()].copy_(st["exp_avg"][n].reshape(-1))
                    flat.exp_avg_sq[o:o + p.numel()].copy_(st["exp_avg_sq"][n].reshape(-1))
                    steps[i] = saved_step.get(n, 0.0)
            flat.seg_step.copy_(steps)
            opt.set_lr(st["lr"])

    # ---- HIP-graph capture of a whole iteration ---------------------------------------------------------------
    def _training_state(self):
        """everything a training iteration changes on the device (parameters, Adam moments and step counts, the device RNG)"""
        bufs = []
        for flat in (self.flat_g, self.flat_d):
            bufs += [flat.flat, flat.exp_avg, flat.exp_avg_sq, flat.seg_step]
        return bufs

    def capture(self, real_example, warmup=1, draws=None):
        """Capture `train_iteration` for this batch shape into HIP graphs (latents and epsilon drawn on the GPU inside
        the graph).  Afterwards `replay(real)` copies `real` into the static input of the graphs captured for its shape and
        launches them.  One GPU: one graph for the whole iteration.  Data parallel: three graphs -- [D forward/backward],
        [D Adam, G forward/backward], [G Adam] -- with the two gradient exchanges issued eagerly between them (on the
        communication stream, `_on_comm_stream`), so no collective is ever captured.

        capture() trains nothing: the warm-up iterations (they register and allocate the packed weight copies, workspaces and the
        allocator's blocks outside the capture) run on a snapshot -- parameters, Adam moments, per-tensor step counts and the device
        RNG state are restored afterwards -- and a stream capture itself executes no kernel.  The reference makes exactly one update
        per batch (train.py:350-385); so does capture() + replay().  Graphs are kept per input shape until the next growth event
        (`refresh_stage`), so a ragged last batch costs one extra capture per stage, not two per epoch.

        draws: optional dict of STATIC device tensors {"z_d", "z_gp", "eps", "z_g"} used instead of drawing inside the graph; the
        caller refills them before each replay (how the parity tests replay an eager trajectory exactly)."""
        if draws is None and not self.device_latents:
            raise RuntimeError("graph capture needs device_latents=True or static draws (CPU-drawn latents cannot be replayed)")
        dr = draws or {}
        d_args = (dr.get("z_d"), dr.get("z_gp"), dr.get("eps"))
        z_g = dr.get("z_g")
        segmented = self.world > 1 or self.force_exchange
        if segmented and self.n_critic != 1:
            raise RuntimeError("segmented (data-parallel) capture supports n_critic = 1")
        static_real = real_example.clone()
        n_packed_before = ops.registry_size()
        state = self._training_state()
        saved = [t.clone() for t in state]
        rng = torch.cuda.get_rng_state(self.device)
        lr_g, lr_d = self.opt_g.param_groups[0]["lr"], self.opt_d.param_groups[0]["lr"]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self.train_iteration(static_real, *d_args, z_g)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for t, v in zip(state, saved):
            t.copy_(v)
        del saved
        torch.cuda.set_rng_state(rng, self.device)
        assert (lr_g, lr_d) == (self.opt_g.param_groups[0]["lr"], self.opt_d.param_groups[0]["lr"])
        ops.bump_weight_epoch()   # the warm-up registered every packed weight (persistent buffers, allocated outside capture):
        self.opt_d.repack()       # rebuild both re-pack tables now (from the restored parameters), so that the captured Adam
        self.opt_g.repack()       # steps find them complete
        ops.refresh_packed()      # (copies of tensors that belong to neither optimiser)
        torch.cuda.synchronize()
        if ops.registry_size() != n_packed_before and self._graphs:
            # This shape registered packed-weight copies the earlier shapes' graphs know nothing about (the pack format of a layer
            # can depend on the batch: ngan_conv3x3_algorithm).  The re-pack launches captured in those graphs run from tables that
            # do not list the new copies, so a replay of them would leave the new copies stale for this shape's next replay:
            # forget the earlier graphs; they are re-captured on next sight with complete tables (the registry no longer grows then).
            self._graphs.clear()
            self._graph = self._entry = None
        # With an RCCL process group alive its watchdog thread polls collective events (hipEventQuery) at any time.  HIP's GLOBAL
        # capture mode refuses such a call from ANY thread while a capture is active (hipErrorStreamCaptureUnsupported -> the
        # watchdog terminates the process); thread_local polices the capturing thread only.  The second rule (an event whose stream
        # is part of the capture) is what _on_comm_stream takes care of.  Both reproduced case by case: tools/capture_event_probe.py.
        mode = ops._diag_env("NGAN_CAPTURE_MODE", "") or ("thread_local" if self._comm_stream is not None else "global")
        # No cyclic garbage collection while a capture is active: a collection that happens to start inside the captured region runs
        # the finalizers of whatever cyclic garbage earlier code left behind on the capturing thread.  The one that must not run there
        # is torch.cuda.CUDAGraph's: destroying an EARLIER captured graph (hipGraphExecDestroy / hipGraphDestroy and the release of its
        # private pool) while a global-mode capture is active fails with hipErrorStreamCaptureUnsupported, which a destructor can only
        # turn into std::terminate -- the `Fatal Python error: Aborted` under "Garbage-collecting" of gpurun_out/r03_a_tests.log (third
        # capture of a process: the first two trainers' graphs were cyclic garbage by then).  Events, tensors of a released graph pool
        # and tensors recorded on a second stream are harmless (tools/gc_capture_probe.py, one case per child process:
        # profiles/r04_gc_capture_probe.txt).  torch.cuda.graph() no longer collects on entry by itself.  Collect now, then keep the
        # collector off until the capture ends.  This concerns GLOBAL-mode captures (one GPU, no process group); with an RCCL group
        # the captures run in thread_local mode, where a graph destroyed on ANOTHER thread is legal -- the collector could still pick the
        # capturing thread, so it is kept off in both modes (gc.disable() is process-wide, for the few milliseconds of a capture).
        # Regression: tests/test_gpu_train.py::test_capture_survives_garbage_left_by_earlier_trainers.
        import gc
        gc.collect()
        gc_was_enabled = gc.isenabled()
        gc.disable()
        try:
            graphs, stats = self._capture_segments(segmented, mode, static_real, d_args, z_g)
        finally:
            if gc_was_enabled:
                gc.enable()
        # Two things the captured launches reference besides the static input: (a) the stem's factor tensors (`StemGradExchange.sink`
        # ran during THIS capture; another shape's capture overwrites stem.captured, and the eager `finish()` between replayed
        # segments must read this graph's factors, not the most recent capture's); (b) the device re-pack tables whose pointers
        # are baked into the captured Adam segments (ops drops its own references when a later capture registers new copies).
        # ... and (c) the two flags `_exchange` / `g_adam` read between replayed segments, as THIS capture's g_compute left them (an eager
        # g_compute in between -- a parity test, a gradient inspection -- may have set them differently)
        entry = (graphs, static_real, stats, self.stem.captured if self.stem is not None else None, ops.table_tensors(),
                 (self._stem_grad_skipped, self._stem_sink_active))
        self._graph, self._entry = graphs, entry
        self._graphs[tuple(real_example.shape)] = entry
        return graphs

    def _capture_segments(self, segmented, mode, static_real, d_args, z_g):
        if not segmented:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode=mode):
                stats = self.train_iteration(static_real, *d_args, z_g)
            graphs = [graph]
        else:
            ga, gb, gc = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(ga, capture_error_mode=mode):
                stats = self.d_compute(static_real, *d_args)
            self._exchange(self.flat_d)                  # eager, on the communication stream; its result is discarded below
            with torch.cuda.graph(gb, pool=ga.pool(), capture_error_mode=mode):
                self.opt_d.step()
                stats.update(self.g_compute(static_real, z_g, skip_stem_grad=True))
            self._exchange(self.flat_g)
            with torch.cuda.graph(gc, pool=ga.pool(), capture_error_mode=mode):
                self.g_adam()
            graphs = [ga, gb, gc]
            torch.cuda.synchronize()
            self.flat_d.grad.zero_()                     # the exchanges above summed never-computed gradients: leave nothing behind
            self.flat_g.grad.zero_()
        return graphs, stats

    def has_graph(self, shape):
        return tuple(shape) in self._graphs

    def replay(self, real=None):
        """One training iteration from the captured graphs.  `real` selects the graphs by its shape (capture() them first);
        without an argument the most recently captured graphs run on their static input as it stands."""
        if real is not None:
            entry = self._graphs.get(tuple(real.shape))
            if entry is None:
                raise RuntimeError(f"no graphs captured for input shape {tuple(real.shape)}: call capture() first (and again after "
                                   f"every growth event)")
            static_real = entry[1]
            static_real.copy_(real, non_blocking=True)
        else:
            if self._graph is None:
                raise RuntimeError("call capture() first (and again after every growth event)")
            entry = self._entry
        graphs, _, stats, stem_factors, _, stem_flags = entry
        if self.stem is not None:
            self.stem.captured = stem_factors     # the factors THESE graphs fill (see capture())
        self._stem_grad_skipped, self._stem_sink_active = stem_flags
        if len(graphs) == 1:
            graphs[0].replay()
        else:
            graphs[0].replay()
            self._exchange(self.flat_d)
            graphs[1].replay()
            self._exchange(self.flat_g)
            graphs[2].replay()
        return stats

    def step(self, real, use_graph=True):
        """train on one batch: graph replay when possible (capturing on first sight of a shape), else eager"""
        if use_graph and self.device_latents and self.n_critic == 1:
            if not self.has_graph(real.shape):
                self.capture(real)
            return self.replay(real)
        return self.train_iteration(real)


# =====================================================================================================================
# Epoch-level driver and command line: the reference's train.py as functions instead of module-level code.
# Out of scope and therefore absent: the PNG dataset with PIL/skimage augmentations (data/NeuronDataset.py), score plots,
# gradient-norm histograms, interactive prompts.  Images come from a tensor file or are synthetic.
# =====================================================================================================================
class TensorImageDataset(torch.utils.data.Dataset):
    """Images (N, C, R, R) in [-1, 1] held on the GPU; `set_image_size` serves them at the current stage's resolution by
    2x2 averaging (the role of NeuronDataset.set_image_size + Resize, data/NeuronDataset.py:112-126)."""

    def __init__(self, images: torch.Tensor):
        assert images.dim() == 4 and images.shape[-1] == images.shape[-2]
        self.full = images
        self.image_size_max = images.shape[-1]
        self.image_size = self.image_size_max
        self._cache = {self.image_size_max: images}

    @classmethod
    def synthetic(cls, n_images, image_size, n_colors=1, device="cpu", seed=123):
        g = torch.Generator().manual_seed(seed)
        return cls((torch.rand(n_images, n_colors, image_size, image_size, generator=g) * 2 - 1).to(device))

    def set_image_size(self, size):
        assert self.image_size_max % size == 0
        if size not in self._cache:
            x = self.full
            while x.shape[-1] > size:
                x = torch.nn.functional.avg_pool2d(x, 2)
            self._cache[size] = x
        self.image_size = size

    def __len__(self):
        return self.full.shape[0]

    def __getitem__(self, i):
        return self._cache[self.image_size][i]


def pggan_train(trainer, dataset, cfg, checkpoint=None, epoch_init=1, epoch_final=None, use_graph=True, log=print,
                samples_dir=None, on_epoch=None):
    """The reference's epoch loop (train.py:298-451) over a PGGANTrainer.
    Per epoch: advance alpha / grow (318-333), one pass over the dataset in batches of cfg.batch_size (350-394), sample-weighted
    epoch means of the monitors (387-398), a status line every 10 epochs (401-422), LR schedule (424-426), loss series (429-432),
    checkpoint + sample grid every cfg.checkpointing_period epochs (435-443).  The monitors stay on the GPU and are read back one
    epoch late through pinned memory, so the host never stalls the launch stream; a NaN loss raises ValueError like the
    reference's loss modules do (loss_functions.py:35-41, 70-72).
    on_epoch(epoch, trainer): optional observer, called once per epoch after the alpha / growth update, i.e. with the structure and
    the learning rate the epoch trains with (what the reference's status line prints, train.py:401-422)."""
    import time
    from .utils import Calculate_D_steps, similarity_loss
    adapt_period = 100                                                # Disc_adapt_update_period, train.py:190
    sim_lambda = float(getattr(cfg, 'sim_loss_lambda', 0.0))          # train.py:300
    sim_decay = float(getattr(cfg, 'sim_loss_lambda_decay_rate', 0.0))
    adapt_critic = bool(getattr(cfg, 'adapt_critic', False))
    G, D = trainer.G, trainer.D
    dev = trainer.device
    epoch_final = epoch_final if epoch_final is not None else cfg.N_epochs + 1
    n_images = len(dataset)
    names = ["score_real", "score_fake", "D_loss", "G_loss", "D_grad_pen"]
    pinned = torch.zeros(2, len(names), pin_memory=True)
    events = [None, None]
    pending = [None, None]
    series = {n: [] for n in names}
    dataset.set_image_size(G.image_size)
    start_time = time.time()

    def consume(slot):
        if events[slot] is None:
            return
        events[slot].synchronize()
        ep = pending[slot]
        vals = pinned[slot].tolist()
        events[slot] = None
        if any(math.isnan(v) for v in vals):
            raise ValueError(f"NaN loss at epoch {ep}: " + ", ".join(f"{n}={v}" for n, v in zip(names, vals)))
        for n, v in zip(names, vals):
            series[n].append(v)
        if checkpoint is not None and ep - 1 < len(checkpoint.Loss_real):
            checkpoint.Loss_real[ep - 1], checkpoint.Loss_fake[ep - 1] = vals[0], vals[1]
            checkpoint.Loss_D[ep - 1], checkpoint.Loss_G[ep - 1] = vals[2], vals[3]
        if ep % 10 == 0:
            done = ep - epoch_init
            log("Epoch:{}, time(s)/iter:{}, lr:{:.4g}, alpha:{: >5.3f}, Res:{}x{}, Loss_real (<D(x)>_x):{: >#7.4g}, "
                "Loss_fake (<D(G(z))>):{: >#7.4g}, G_loss:{: >#7.4g}, D_loss:{: >#7.4g}, D_grad_pen:{: >#7.4g}".format(
                    ep, "{:.3f}".format((time.time() - start_time) / done) if done > 0 else "----",
                    trainer.opt_g.param_groups[0]["lr"], G.alpha_value(), G.image_size, G.image_size, vals[0], vals[1], vals[3],
                    vals[2], vals[4]))

    for epoch in range(epoch_init, epoch_final):
        if trainer.start_epoch(epoch, cfg.transit_sch):
            dataset.set_image_size(G.image_size)
        if on_epoch is not None:
            on_epoch(epoch, trainer)
        # number of critic steps this epoch (train.py:336-340); the score series lags one epoch here (deferred read-back)
        if adapt_critic and len(series["score_real"]) > adapt_period:
            n_d_steps = Calculate_D_steps(series["score_real"], series["score_fake"], 0, cfg.n_critic, Period=adapt_period)
        else:
            n_d_steps = cfg.n_critic
        if sim_decay > 0 and sim_lambda > 0:                          # train.py:343-348
            sim_lambda = cfg.sim_loss_lambda * (1 - sim_decay) ** (epoch - 1) if sim_lambda > 1e-5 else 0.0
        acc = torch.zeros(len(names), device=dev)
        order = torch.randperm(n_images).tolist()                     # DataLoader(shuffle=True), train.py:153
        for i in range(0, n_images, cfg.batch_size):
            if hasattr(dataset, "batch"):                          # device dataset: one augmentation launch per batch (data.py)
                images = dataset.batch(order[i:i + cfg.batch_size])
            else:
                images = torch.stack([dataset[j] for j in order[i:i + cfg.batch_size]]).to(dev)
            b = images.size(0)
            trainer.n_critic = n_d_steps
            stats = trainer.step(images, use_graph=use_graph)        # graphs are cached per shape until the next growth event
            g_loss = stats["G_loss"]
            if sim_lambda > 0:
                # the reference adds similarity_loss(real images, latents) to the generator loss (train.py:379-381); it depends on
                # neither network, so it changes the monitored value only
                g_loss = g_loss + similarity_loss(images, trainer.last_z_g, sim_lambda)
            acc += b * torch.stack([stats["score_real"], stats["score_fake"], stats["D_loss"], g_loss,
                                    stats["D_grad_pen"].float()])
        slot = epoch & 1
        consume(slot)
        pinned[slot].copy_(acc / n_images, non_blocking=True)
        events[slot] = torch.cuda.Event()
        events[slot].record()
        pending[slot] = epoch
        consume(slot ^ 1)                                              # the previous epoch's numbers are ready by now
        lr = lr_schedule(epoch, cfg.learning_rate, cfg.transit_sch, cfg.N_epochs)
        if lr is not None:
            trainer.opt_d.set_lr(lr)
            trainer.opt_g.set_lr(lr)
        if checkpoint is not None and epoch % cfg.checkpointing_period == 0:
            consume(slot)
            checkpoint.lr = trainer.opt_g.param_groups[0]["lr"]
            checkpoint.save_state(epoch)
            if samples_dir is not None:
                from .utils import plot_gen_samples
                plot_gen_samples(G, N_images=16, seed=0, filename=os.path.join(samples_dir, "Samples_{}_{:d}.png".format(cfg.ID, epoch)))
    consume(0)
    consume(1)
    return series


def build_arg_parser():
    """The reference's flags (train.py:39-91), same names, types and help; defaults are irrelevant because -- as in the
    reference (train.py:95-104) -- only flags literally present on the command line override the configuration module."""
    import argparse
    import uuid
    p = argparse.ArgumentParser()
    p.add_argument('--configs', type=str, default='', help='Filename of configurations stored in ./configs')
    for name in ('root_dir', 'dataset_dir', 'images_dir', 'weights_dir', 'plots_dir', 'weights_init', 'dis_weights'):
        p.add_argument('--' + name, type=str, default='')
    p.add_argument('--wgan', action='store_true')
    p.add_argument('--n_critic', type=int, default=5)
    p.add_argument('--adapt_critic', action='store_true', default=False)
    p.add_argument('--unroll_steps', type=int, default=0)
    p.add_argument('--pggan', action='store_true')
    p.add_argument('--grad_pen_lambda', type=float, default=0.0)
    p.add_argument('--transit_sch', type=float, default=[50, 100, 150, 200, 250, 300, 350], nargs='*')
    p.add_argument('--transit_period', type=int, default=None)
    p.add_argument('--alpha_step', type=float, default=0.05)
    p.add_argument('--RMSprop', action='store_true', default=False)
    p.add_argument('--learning_rate', type=float, default=0.00002)
    p.add_argument('--batch_size', type=int, default=8)
    p.add_argument('--N_epochs', type=int, default=1000)
    p.add_argument('--beta1', type=float, default=0.8)
    p.add_argument('--sim_loss_lambda', type=float, default=0.0)
    p.add_argument('--sim_loss_lambda_decay_rate', type=float, default=0.0)
    p.add_argument('--drift_epsilon', type=float, default=0.001)
    p.add_argument('--ID', type=str, default=uuid.uuid4().hex[:4])
    p.add_argument('--resume', action='store_true', default=False)
    p.add_argument('--seed', type=int, default=1)
    p.add_argument('--checkpointing_period', type=int, default=100)
    p.add_argument('--translation', type=float, default=0.0)
    p.add_argument('--device', type=str, default='cuda', choices=['cpu', 'mps', 'cuda'])
    p.add_argument('--N_workers', type=int, default=2)
    p.add_argument('--pin_memory', action='store_true', default=False)
    # additions of this implementation
    p.add_argument('--images', type=str, default='', help='.pt / .npy file with the training images (N, C, R, R) in [-1, 1]; '
                                                          'synthetic uniform images when omitted')
    p.add_argument('--N_epochs_session', type=int, default=None)
    return p


def main(argv=None):
    """`python -m neuron_gan_amd.train ...` -- bootstrap of the reference's train.py:94-296, 623-625 for the PGGAN path."""
    import sys
    from .configs import config
    from . import models
    from .utils import Checkpointer
    argv = list(sys.argv[1:] if argv is None else argv)
    options = build_arg_parser().parse_args(argv)
    given = [a[2:] for a in argv if a.startswith('--') and a not in ('--configs', '--images')]   # train.py:95
    given = [g for g in given if g in config.configs_name]
    if options.configs:
        config.import_configs(options.configs, {a: getattr(options, a) for a in given}, create_dirs=True)
    else:
        config.set_configs(**{a: getattr(options, a) for a in given})
        config.validate_configs(create_dirs=True)
    if not config.pggan or config.wgan:
        raise NotImplementedError("only the PGGAN path (pggan=True, wgan=False) is implemented")
    if config.device != 'cuda':
        raise RuntimeError("the HIP path needs device='cuda' (there is no CPU fallback)")
    config.print_configs()
    torch.manual_seed(config.seed)
    device = torch.device('cuda:0')
    n_up = len(config.N_gen_features) - 1
    if options.images:
        data = torch.load(options.images) if options.images.endswith('.pt') else torch.from_numpy(np.load(options.images))
        data = data.float()
        if data.dim() == 4 and data.shape[1] == 1:
            # single-colour images: the device dataset with the reference's augmentation chain (data/NeuronDataset.py:112-126,
            # antialiased Resize to the stage resolution); it takes [0, 1] images and renormalises to [-1, 1] itself
            from .data import NeuronDataset
            dataset = NeuronDataset((data + 1.0) * 0.5, augmentations=True, im_translation=float(getattr(config, 'translation', 0.0)),
                                    device=device, seed=config.seed)
        else:
            dataset = TensorImageDataset(data.to(device))
    else:
        dataset = TensorImageDataset.synthetic(16, config.image_size, config.N_colors, device=device)
    size_init = dataset.image_size_max // (2 ** n_up)                                       # train.py:162-165
    G = models.Generator_PG(config.N_gen_features, image_size_init=size_init).to(device)    # train.py:172-175
    D = models.Discriminator_PG(config.N_dis_features, image_size_init=size_init).to(device)
    filename = os.path.join(config.weights_dir, 'GenDisc_{}.pth'.format(config.ID))         # train.py:196-197
    # the trainer exists before the checkpoint is read, so that `--resume` also restores the optimiser state this implementation
    # adds to its checkpoints (Adam moments and per-tensor step counts; the reference saves none, utils.py:160-169)
    trainer = PGGANTrainer(G, D, learning_rate=config.learning_rate, beta1=config.beta1, grad_pen_lambda=config.grad_pen_lambda,
                           drift_epsilon=config.drift_epsilon, n_critic=config.n_critic, alpha_step=config.alpha_step,
                           device_latents=True)
    checkpoint = Checkpointer(G, D, config.learning_rate, filename, N_epochs=config.N_epochs, device=device, extra_checkpoint_period=1e3,
                              trainer=trainer)
    if config.resume and os.path.exists(filename):
        checkpoint.load_state()
    elif config.weights_init:
        checkpoint.load_state(os.path.join(config.weights_dir, config.weights_init))
    assert G.image_size == D.image_size, 'The generator and discriminator are at different resolution'   # train.py:215-216
    epoch_init = checkpoint.epoch + 1
    lr0 = lr_schedule(epoch_init - 1, config.learning_rate, config.transit_sch, config.N_epochs)       # train.py:288-289
    if lr0 is not None:
        trainer.opt_d.set_lr(lr0)
        trainer.opt_g.set_lr(lr0)
    epoch_final = epoch_init + config.N_epochs_session if config.N_epochs_session else config.N_epochs + 1
    return pggan_train(trainer, dataset, config, checkpoint=checkpoint, epoch_init=epoch_init, epoch_final=epoch_final,
                       samples_dir=config.samples_sub_dir)


if __name__ == '__main__':
    main()
