"""Differentiable operators of the PGGAN / WGAN-GP hot path, each backed by hand-written gfx950 kernels
(include/ngan.h) and closed under differentiation, so that the gradient-penalty double-backward through the
critic (loss_functions.py:175 in the reference) runs entirely on those kernels.

All tensors here are contiguous, channels-last: (B, H, W, C), fp32 -- or, in the "bf16" mode (`set_conv_precision("bf16")`, precision
code 5 of include/ngan.h), bf16 for every ACTIVATION tensor (layer outputs and the gradients w.r.t. them; images, norms, scalars,
parameters and parameter gradients stay fp32).  An operator picks its entry point by the dtype of the activation tensors it is
handed (`_k`), so the two storage modes share every autograd.Function below.  `models.py` converts at the module edges.

Closure under differentiation (what the backward of each operator is built from):
    ConvLReLUPN   -> LReLUPNBwd, ConvDgrad, ConvWgrad, ChannelSum
    Conv          -> ConvDgrad, ConvWgrad, ChannelSum
    ConvDgrad     -> Conv, ConvWgrad                 ConvWgrad -> ConvDgrad, Conv
    LReLUPNBwd    -> lrelu_pixelnorm_bwdbwd (the only operator with a non-zero Hessian)
    FromImage / FromImageDx / FromImageDw, FinalDot / FinalDotDx / FinalDotDw: bilinear triples, closed
    Lerp <-> FadeBwd, Up2 <-> Up2Adjoint, Pool2 <-> Pool2Adjoint: linear pairs
"""
import os
import struct
import weakref

import numpy as np
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _C

RES_NONE, RES_POOL2, RES_UP2 = 0, 1, 2
EPI_NONE, EPI_LRELU_PN, EPI_PN_BWD, EPI_TO_IMAGE = 0, 1, 2, 3      # epilogues of ngan_conv3x3_fwd_ex (include/ngan.h)

# ---------------------------------------------------------------------------------------------------------
# First-order passes.  D(real), D(fake), D(G(z)) and G(z) are differentiated once (train.py:365, 384); only D(x_hat) of the
# gradient penalty needs the double-backward closure below.  Inside `first_order_only()` the model planner (models._exec) links
# each LeakyReLU->PixelNorm producer to the ONE conv that consumes its output (`PNLink`), and a plain (not create_graph) backward
# then lets the consumer's input-gradient kernel apply the producer's LeakyReLU->PixelNorm backward in its epilogue: the gradient
# w.r.t. the producer's output is never written and re-read, and the producer's own PixelNorm-backward launch disappears.
# ---------------------------------------------------------------------------------------------------------
def _diag_env(name, default):
    """Measurement A/B switches are honoured only with NGAN_DIAG=1 in the environment (tools/ab_env.sh sets it): a stray variable must
    not change which kernels a training run or a parity test exercises.  (The documented user knobs are NGAN_CONV_PRECISION and
    NGAN_LIB_PATH; the kernel library itself reads no environment variable.)"""
    return os.environ.get(name, default) if os.environ.get("NGAN_DIAG") == "1" else default


_first_order = 0
_first_order_allowed = _diag_env("NGAN_FIRST_ORDER_FUSION", "1") != "0"     # A/B switch for measurements and tests


def allow_first_order_fusion(flag):
    global _first_order_allowed
    _first_order_allowed = bool(flag)


class first_order_only:
    def __enter__(self):
        global _first_order
        _first_order += 1
        return self

    def __exit__(self, *exc):
        global _first_order
        _first_order -= 1
        return False


def first_order_enabled():
    return _first_order > 0 and _first_order_allowed


_inputs_only = 0


class inputs_only:
    """Inside, backward passes produce input gradients only; weight and bias gradients are skipped.

    For `torch.autograd.grad(outputs, inputs=<an activation>)`, i.e. the gradient penalty's d D(x_hat) / d x_hat
    (reference loss_functions.py:170-176).  The engine hands the built-in conv backward of the reference an output mask, so no
    weight gradient is computed there; a custom Function only sees `needs_input_grad`, which mirrors `requires_grad`, and would
    spend a full weight-gradient contraction per layer on a result the engine then drops."""

    def __enter__(self):
        global _inputs_only
        _inputs_only += 1
        return self

    def __exit__(self, *exc):
        global _inputs_only
        _inputs_only -= 1
        return False


def _param_grads_wanted():
    return _inputs_only == 0


class PNLink:
    """Hand-off between the LeakyReLU->PixelNorm that produced a tensor (y, rn) and the single conv consuming it.  The consumer's
    backward sets `fused` after it has applied the producer's LeakyReLU->PixelNorm backward to the gradient it returns.
    `y` / `rn` are DETACHED aliases of the producer's outputs: the producer's autograd node holds this object, and a reference to
    its own outputs (which point back to the node) would be a cycle through C++ that neither refcounting nor gc can free --
    every eager iteration would leak its activations (tools/leak_check.py)."""
    __slots__ = ("y", "rn", "slope", "fused")

    def __init__(self):
        self.y = self.rn = None
        self.slope = 0.0
        self.fused = False
# arithmetic of the 3x3 convolutions: "f32" = exact fp32 MFMA everywhere; "bf16x3" = few-channel layers on large images
# use the split-bf16 kernels (3 bf16 MFMAs per product group, fp32 accumulate, ~1e-5 relative error), the rest stays fp32
# "bf16" = bf16 activation STORAGE + one bf16 MFMA per product group, fp32 accumulate / statistics / master weights: BASELINE.json's C2
# configuration, an addition the reference does not have (train.py:136-144), with its own stated tolerance
PRECISIONS = {"f32": 0, "bf16x3": 1, "bf16": 5}
_conv_precision = PRECISIONS[os.environ.get("NGAN_CONV_PRECISION", "f32")]


def set_conv_precision(name):
    global _conv_precision
    _conv_precision = PRECISIONS[name]
    bump_weight_epoch()


def get_conv_precision():
    return [k for k, v in PRECISIONS.items() if v == _conv_precision][0]


def act_dtype():
    """storage type of activation tensors in the current mode (what the image -> feature operators allocate)"""
    return torch.bfloat16 if _conv_precision == 5 else torch.float32


def _k(name, *acts):
    """the entry point for these activation tensors: `ngan_bf16_<op>` (include/ngan.h, last section) when they are bf16"""
    for t in acts:
        if t is not None and t.dtype == torch.bfloat16:
            return "ngan_bf16_" + name[5:]
    return name

PIXELNORM_EPS = 1e-8  # models.py:105 of the reference

# ---------------------------------------------------------------------------------------------------------
# packed weights.  A conv weight is used in MFMA-fragment order, pre-scaled, possibly split into bf16 hi/lo, and in a
# forward and a flipped (dgrad) orientation.  Packed copies of PARAMETERS are persistent: one buffer per
# (parameter, orientation, precision, scale), registered on first use; after an optimiser step (which updates the
# parameters through raw pointers) `refresh_packed(owner, params)` re-packs that optimiser's parameters (`bump_weight_epoch()` marks
# every copy stale instead: precision switches, graph capture)
# with one table-driven launch.  Packed copies of other tensors (the "weights" of a double-backward) are one-shot.
# ---------------------------------------------------------------------------------------------------------
_weight_epoch = 0
_registry = {}          # key -> dict(ref, packed, cout, cin, mode, prec, scale, epoch)
_table = None           # (device table tensor, n_entries, total_elements, registry size it was built for)


def bump_weight_epoch():
    """Every packed copy is stale from now on (call after changing parameters through raw pointers)."""
    global _weight_epoch
    _weight_epoch += 1


def clear_packed():
    """Forget every packed buffer (used around HIP-graph capture so that no buffer of a private pool is kept)."""
    global _table
    _registry.clear()
    _table = None
    bump_weight_epoch()


def registry_size():
    """number of persistent packed-weight copies registered so far (the step driver compares it around a capture's warm-up)"""
    return len(_registry)


def table_tensors():
    """the device re-pack tables currently cached (a captured graph that contains their launches must keep them alive)"""
    return [c[0] for c in _table.values()] if isinstance(_table, dict) else []


def refresh_packed(owner=None, params=None):
    """Re-pack the registered packed copies with ONE launch; returns the number of entries refreshed.  With `params` (an iterable
    of parameters) and `owner` (any hashable tag for that set, e.g. id of the optimiser) only the copies of those parameters are
    re-packed -- after the critic's Adam step the generator's packed weights are still valid, and vice versa."""
    global _table
    if _table is None or not isinstance(_table, dict):
        _table = {}
    ids = None if params is None else {id(p) for p in params}
    live = [e for e in _registry.values() if e["ref"]() is not None and (ids is None or id(e["ref"]()) in ids)]
    if not live:
        return 0
    ptrs = [e["ref"]().data_ptr() for e in live]
    cached = _table.get(owner)
    if cached is None or cached[3] != len(live) or cached[4] != ptrs:
        rec, first = b"", 0
        for e in live:
            w = e["ref"]()
            rec += struct.pack("<QQiiiifiq", w.data_ptr(), e["packed"].data_ptr(), e["cout"], e["cin"], e["mode"], e["prec"],
                               e["scale"], 0, first)
            first += _C.lib().ngan_conv3x3_pack_elements(e["cout"], e["cin"], e["mode"], e["prec"])
        dev = live[0]["packed"].device
        table = torch.from_numpy(np.frombuffer(rec, dtype=np.uint8).copy()).to(dev)
        cached = _table[owner] = (table, len(live), first, len(live), ptrs)
    _C.call("ngan_conv3x3_pack_many", cached[0], cached[1], cached[2])
    for e in live:
        e["epoch"] = _weight_epoch
        e["version"] = e["ref"]()._version
    return len(live)


def _packed(weight, mode, scale, precision=0):
    cout, cin = weight.shape[0], weight.shape[1]
    n_packed = _C.conv3x3_packed_floats(cout, cin, precision)
    if n_packed <= 0:
        raise RuntimeError(f"conv3x3: unsupported channel counts Cin={cin}, Cout={cout} (must be positive multiples of 16)")
    persistent = isinstance(weight, torch.nn.Parameter) and weight.is_contiguous() and not torch.cuda.is_current_stream_capturing()
    key = (id(weight), mode, precision, float(scale))
    e = _registry.get(key)
    if e is not None and e["ref"]() is weight and e["data_ptr"] == weight.data_ptr():
        if e["epoch"] == _weight_epoch and e["version"] == weight._version:
            return e["packed"]
        packed = e["packed"]               # stale: re-pack in place (refresh_packed() normally did this already)
    else:
        e = None
        packed = torch.empty(n_packed, device=weight.device, dtype=torch.float32)
    w = weight.detach()
    if not w.is_contiguous():
        w = w.contiguous()
    _C.call("ngan_conv3x3_pack_weights", w, packed, cout, cin, mode, float(scale), precision)
    if e is not None:
        e["epoch"], e["version"] = _weight_epoch, weight._version
    elif persistent:
        global _table
        _registry[key] = dict(ref=weakref.ref(weight), packed=packed, cout=cout, cin=cin, mode=mode, prec=precision,
                              scale=float(scale), epoch=_weight_epoch, version=weight._version, data_ptr=weight.data_ptr())
        _table = None
    return packed


def _c(t):
    """contiguous view of a tensor (autograd may hand us expanded / non-contiguous grads); fp32, or bf16 activation storage"""
    if t is None:
        return None
    if t.dtype != torch.float32 and t.dtype != torch.bfloat16:
        raise RuntimeError(f"the HIP path stores tensors as fp32 (or bf16 activations in the bf16 mode), got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _conv_in_shape(out_shape_bhw, resample):
    b, h, w = out_shape_bhw
    if resample == RES_POOL2:
        return b, 2 * h, 2 * w
    if resample == RES_UP2:
        return b, h // 2, w // 2
    return b, h, w


def _conv_out_hw(x, resample):
    b, h, w, _ = x.shape
    if resample == RES_POOL2:
        return b, h // 2, w // 2
    if resample == RES_UP2:
        return b, 2 * h, 2 * w
    return b, h, w


def _pool_first(resample):
    """Exact-fp32 mode: a 2x2-average-pooled conv input is pooled by one streaming pass and the convolution then runs on the
    quarter-size tensor with plain input (the Winograd / tile / mid kernels at 0.6 - 0.9 of the fp32 MFMA peak) instead of pooling
    inside the generic kernel's staging (four dependent loads per staged element: 0.3 - 0.35).  The pooled copy also serves the
    weight gradient of the same layer.  (nn.AvgPool2d in front of a block's first conv, reference models.py:252-254.)"""
    return resample == RES_POOL2 and _conv_precision == 0 and _pool_first_allowed


_pool_first_allowed = _diag_env("NGAN_POOL_FIRST", "1") != "0"     # A/B switch for measurements and tests
_pool_out_allowed = _diag_env("NGAN_POOL_OUT", "1") != "0"         # A/B switch: pooled side output of the producing conv


def _pooled_side(x):
    """The 2x2-averaged copy a producing conv kernel wrote next to its output x (`_run_conv(..., pool_out=True)`), or None.  The copy
    travels as an attribute of the tensor OBJECT, so an op that returns a new object (contiguous(), detach(), a view) simply loses
    it -- a pooling pass runs instead, same bits.  What must never happen is a STALE copy: it is used only if x has not been written
    since the producer stored it (tensor version counter) and its shape is exactly the pooled shape of x."""
    side = getattr(x, "_ngan_pooled", None)
    if side is None:
        return None
    yp, version = side
    b, h2, w2, c = x.shape
    if version != x._version or tuple(yp.shape) != (b, h2 // 2, w2 // 2, c) or yp.dtype != x.dtype or yp.device != x.device:
        return None
    return yp


def _pooled(x):
    b, h2, w2, c = x.shape
    return _resample("ngan_pool2_fwd", x, (b, h2 // 2, w2 // 2, c), b, h2 // 2, w2 // 2, c)


def _bf16_prec(b, h, w, k, n, resample):
    """precision code of a conv on bf16 activations: 5, or an error -- there is no fallback from bf16 storage to an fp32-storage kernel"""
    if _C.conv3x3_algorithm(b, h, w, k, n, resample, 5) != 5:
        raise RuntimeError(f"conv3x3 on bf16 activations takes 16 / 32 / 64 / 128 channels per call, got K={k}, N={n}")
    return 5


def _n_chunks(n):
    """Output-channel counts one kernel launch takes are 16, 32, 64 and 128 (include/ngan.h); any other multiple of 16 -- the
    reference's wide presets have 256-channel blocks, configs/config.py:87-98 -- is computed in chunks of those sizes, largest
    first: [(first channel, count), ...]"""
    out, c0 = [], 0
    while n - c0 >= 128:
        out.append((c0, 128))
        c0 += 128
    for size in (64, 32, 16):
        if n - c0 >= size:
            out.append((c0, size))
            c0 += size
    if c0 != n:
        raise RuntimeError(f"conv3x3: channel count {n} is not a multiple of 16")
    return out


def _run_conv(x, weight, bias, resample, scale, epilogue, slope, keep_pooled=None, pool_out=False):
    """y (and rnorm) = epilogue(conv3x3(resample(x), scale*W) + bias); keep_pooled: a list that receives the pooled input copy
    when one was made (`_pool_first`); pool_out: the consumer of y is an avg-pooled conv -- where the kernel can, it also writes
    the 2x2 average of y, which travels with y as `y._ngan_pooled` (same bits as the pooling pass it replaces)"""
    bf = x.dtype == torch.bfloat16
    if _pool_first(resample) and not bf:
        side = _pooled_side(x)
        x = side if side is not None else _pooled(x)
        resample = RES_NONE
        if keep_pooled is not None:
            keep_pooled.append(x)
    b, h, w = _conv_out_hw(x, resample)
    cout, cin = weight.shape[0], weight.shape[1]
    if x.shape[3] != cin:
        raise RuntimeError(f"conv3x3: input has {x.shape[3]} channels, weight expects {cin}")
    y = torch.empty((b, h, w, cout), device=x.device, dtype=x.dtype)
    rn = torch.empty((b, h, w), device=x.device, dtype=torch.float32) if epilogue else None
    chunks = _n_chunks(cout)
    if len(chunks) > 1:
        # wide layer: one launch per output-channel chunk (each reads the whole input), then LeakyReLU -> PixelNorm over all the
        # channels as a launch of its own (in place).  A compatibility path for the wide presets' small images, not a fast one.
        for c0, n in chunks:
            yc, _ = _run_conv(x, weight[c0:c0 + n], bias[c0:c0 + n] if bias is not None else None, resample, scale, 0, 0.0)
            y[..., c0:c0 + n].copy_(yc)
        if epilogue:
            _C.call(_k("ngan_lrelu_pixelnorm_fwd", y), y, None, y, rn, b * h * w, cout, float(slope), PIXELNORM_EPS)
        return y, rn
    if bf:      # bf16 activation storage: one kernel family, resampling (pool / bilinear) while the tile is staged
        _C.call("ngan_bf16_conv3x3_fwd", x, _packed(weight, 0, scale, _bf16_prec(b, h, w, cin, cout, resample)), bias, y, rn, None, None, None,
                b, h, w, cin, cout, resample, epilogue, 0, float(slope), PIXELNORM_EPS)
        return y, rn
    prec = _C.conv3x3_algorithm(b, h, w, cin, cout, resample, _conv_precision)
    packed = _packed(weight, 0, scale, prec)
    if pool_out and epilogue == EPI_LRELU_PN and _pool_out_allowed and _C.conv3x3_pooled_output(b, h, w, cin, cout, resample, prec):
        yp = torch.empty((b, h // 2, w // 2, cout), device=x.device, dtype=torch.float32)
        _C.call("ngan_conv3x3_fwd_ex", x, packed, bias, y, rn, None, None, yp, b, h, w, cin, cout, resample, epilogue, 0, float(slope),
                PIXELNORM_EPS, prec, 0)
        y._ngan_pooled = (yp, y._version)       # valid for exactly this tensor object in exactly this state: _pooled_side
        return y, rn
    _C.call("ngan_conv3x3_fwd", x, packed, bias, y, rn, b, h, w, cin, cout, resample, epilogue, 0, float(slope), PIXELNORM_EPS, prec,
            _C.CONV_SKIP_BORDER if prec == 3 else 0)
    if prec == 3:      # bilinear x2 folded into the weights: the border ring is a launch of its own (NGAN_CONV_SKIP_BORDER, include/ngan.h)
        _C.call("ngan_conv3x3_up2_border", x, packed, bias, y, rn, b, h, w, cin, cout, epilogue, float(slope), PIXELNORM_EPS)
    return y, rn


def _run_dgrad(g, weight, resample, scale, link=None):
    """gx = resample^T(conv3x3_transposed(g, scale*W)); with `link`: followed by the backward of the LeakyReLU->PixelNorm that
    produced the conv's input (link.y, link.rn), fused into the kernel's epilogue where one exists"""
    b, h, w, cout = g.shape
    cin = weight.shape[1]
    if cout != weight.shape[0]:
        raise RuntimeError(f"conv3x3 dgrad: gradient has {cout} channels, weight has {weight.shape[0]} outputs")
    ay, arn, slope = (link.y, link.rn, float(link.slope)) if link is not None else (None, None, 0.0)
    chunks = _n_chunks(cin)
    if len(chunks) > 1:
        # wide layer: the input gradient in chunks of its channels (the kernel's N), unfused; the producer's PixelNorm backward as
        # a launch of its own over all the channels
        oh, ow = (2 * h, 2 * w) if resample == RES_POOL2 else (h, w)
        full = torch.empty((b, oh, ow, cin), device=g.device, dtype=g.dtype)
        for c0, n in chunks:
            full[..., c0:c0 + n].copy_(_run_dgrad(g, weight[:, c0:c0 + n], RES_POOL2 if resample == RES_POOL2 else RES_NONE, scale))
        if resample == RES_UP2:
            gx = torch.empty((b, h // 2, w // 2, cin), device=g.device, dtype=g.dtype)
            if link is not None:
                _C.call(_k("ngan_up2_adjoint_pnbwd", full), full, ay, arn, gx, b, h // 2, w // 2, cin, slope)
            else:
                _C.call(_k("ngan_up2_adjoint", full), full, gx, b, h // 2, w // 2, cin)
            return gx
        if link is not None:
            if tuple(ay.shape) != tuple(full.shape):
                raise RuntimeError(f"PixelNorm hand-off: producer output {tuple(ay.shape)} is not the conv input {tuple(full.shape)}")
            _C.call(_k("ngan_lrelu_pixelnorm_bwd", full), full, None, ay, arn, full, b * oh * ow, cin, slope)
        return full
    epi = EPI_PN_BWD if link is not None else EPI_NONE
    if g.dtype == torch.bfloat16:
        packed = _packed(weight, 1, scale, _bf16_prec(b, h, w, cout, cin, 0))
        if link is not None and ay.dtype != g.dtype:
            raise RuntimeError("PixelNorm hand-off: producer output and gradient differ in storage type")
        if resample == RES_UP2:
            gfull = torch.empty((b, h, w, cin), device=g.device, dtype=g.dtype)
            _C.call("ngan_bf16_conv3x3_fwd", g, packed, None, gfull, None, None, None, None, b, h, w, cout, cin, 0, EPI_NONE, 0, 0.0, 0.0)
            gx = torch.empty((b, h // 2, w // 2, cin), device=g.device, dtype=g.dtype)
            if link is not None:
                if tuple(ay.shape) != tuple(gx.shape):
                    raise RuntimeError(f"PixelNorm hand-off: producer output {tuple(ay.shape)} is not the conv input {tuple(gx.shape)}")
                _C.call("ngan_bf16_up2_adjoint_pnbwd", gfull, ay, arn, gx, b, h // 2, w // 2, cin, slope)
            else:
                _C.call("ngan_bf16_up2_adjoint", gfull, gx, b, h // 2, w // 2, cin)
            return gx
        pool = resample == RES_POOL2
        gx = torch.empty((b, 2 * h, 2 * w, cin) if pool else (b, h, w, cin), device=g.device, dtype=g.dtype)
        if link is not None and tuple(ay.shape) != tuple(gx.shape):
            raise RuntimeError(f"PixelNorm hand-off: producer output {tuple(ay.shape)} is not the conv input {tuple(gx.shape)}")
        _C.call("ngan_bf16_conv3x3_fwd", g, packed, None, gx, None, ay, arn, None, b, h, w, cout, cin, 0, epi, 1 if pool else 0, slope, 0.0)
        return gx
    prec = _C.conv3x3_algorithm(b, h, w, cout, cin, 0, _conv_precision)
    packed = _packed(weight, 1, scale, prec)
    if resample == RES_POOL2:
        gx = torch.empty((b, 2 * h, 2 * w, cin), device=g.device, dtype=torch.float32)
        if link is not None and tuple(ay.shape) != tuple(gx.shape):
            raise RuntimeError(f"PixelNorm hand-off: producer output {tuple(ay.shape)} is not the conv input {tuple(gx.shape)}")
        _C.call("ngan_conv3x3_fwd_ex", g, packed, None, gx, None, ay, arn, None, b, h, w, cout, cin, 0, epi, 1, slope, 0.0, prec, 0)
        return gx
    if resample == RES_UP2:
        gfull = torch.empty((b, h, w, cin), device=g.device, dtype=torch.float32)
        _C.call("ngan_conv3x3_fwd_ex", g, packed, None, gfull, None, None, None, None, b, h, w, cout, cin, 0, EPI_NONE, 0, 0.0, 0.0, prec, 0)
        gx = torch.empty((b, h // 2, w // 2, cin), device=g.device, dtype=torch.float32)
        if link is not None:
            if tuple(ay.shape) != tuple(gx.shape):
                raise RuntimeError(f"PixelNorm hand-off: producer output {tuple(ay.shape)} is not the conv input {tuple(gx.shape)}")
            _C.call("ngan_up2_adjoint_pnbwd", gfull, ay, arn, gx, b, h // 2, w // 2, cin, slope)
        else:
            _C.call("ngan_up2_adjoint", gfull, gx, b, h // 2, w // 2, cin)
        return gx
    gx = torch.empty((b, h, w, cin), device=g.device, dtype=torch.float32)
    if link is not None and tuple(ay.shape) != tuple(gx.shape):
        raise RuntimeError(f"PixelNorm hand-off: producer output {tuple(ay.shape)} is not the conv input {tuple(gx.shape)}")
    _C.call("ngan_conv3x3_fwd_ex", g, packed, None, gx, None, ay, arn, None, b, h, w, cout, cin, 0, epi, 0, slope, 0.0, prec, 0)
    return gx


# ---- deferred slab reduction: inside `deferred_wgrad()` the in-place weight gradients only write their slabs; leaving the
# block reduces all of them with one launch (contributions to the same gradient merged, fixed order)
_defer_depth = 0
_pending = {}       # gradient data_ptr -> dict(gw, plan, cin, sources=[(workspace, nparts, scale)])


class deferred_wgrad:
    def __enter__(self):
        global _defer_depth
        _defer_depth += 1
        return self

    def __exit__(self, *exc):
        global _defer_depth
        _defer_depth -= 1
        if _defer_depth == 0:
            flush_wgrad()
        return False


def flush_wgrad():
    """Reduce every pending slab set.  The contributions to one gradient are summed in a CANONICAL order (by the role of the node
    that produced them, then by arrival), not in the order autograd happened to run the nodes: that order is not reproducible for a
    double-backward graph (autograd numbers nodes per thread, and the nodes a create_graph pass creates on the engine's thread are
    numbered independently of the forward nodes created on the caller's thread, so their relative priority drifts from one
    iteration to the next) and a floating-point sum depends on it in the last bit."""
    if not _pending:
        return 0
    rec, n = b"", 0
    for e in _pending.values():
        src = sorted(e["sources"], key=lambda s: s[3])        # stable: equal roles keep their arrival order
        for first in range(0, len(src), 4):            # at most 4 slab sets per record; further records accumulate
            part = src[first:first + 4]
            ptrs = [s[0].data_ptr() for s in part] + [0] * (4 - len(part))
            nparts = [s[1] for s in part] + [0] * (4 - len(part))
            scales = [s[2] for s in part] + [0.0] * (4 - len(part))
            _, nslices, n_ci, co_s, ci_s = e["plan"]
            rec += struct.pack("<4QQ4i8i4f", *ptrs, e["gw"].data_ptr(), *nparts, len(part), nslices, n_ci, co_s, ci_s, e["cin"], 1, 0,
                               *scales)
            n += 1
    _C.wgrad_reduce_many(rec, n)
    _pending.clear()
    return n


def _run_wgrad(x, g, resample, scale, accumulate_into=None, role=0, pooled=None):
    """role: 0 = weight gradient of a forward conv node, 1 = of an input-gradient node (ConvDgrad.backward); see flush_wgrad.
    pooled: the 2x2-averaged copy of x the forward pass made, if it kept one (`_pool_first`)"""
    bf = g.dtype == torch.bfloat16
    if x.dtype != g.dtype:
        raise RuntimeError(f"conv3x3 wgrad: input is {x.dtype}, output gradient is {g.dtype}")
    if _pool_first(resample) and not bf:
        x = pooled if pooled is not None else _pooled(x)
        resample = RES_NONE
    b, h, w, cout = g.shape
    cin = x.shape[3]
    gw = accumulate_into if accumulate_into is not None else torch.empty((cout, cin, 3, 3), device=g.device, dtype=torch.float32)
    ws = torch.empty(_C.wgrad_workspace_bytes(b, h, w, cin, cout) // 4, device=g.device, dtype=torch.float32)
    prec = 5 if bf else _conv_precision
    if bf and prec == 5 and (cin % 16 or cout % 16):
        raise RuntimeError(f"conv3x3 wgrad on bf16 activations: Cin={cin}, Cout={cout} must be multiples of 16")

    def launch(accumulate):
        if bf:      # bf16 x and g, fp32 slabs and gradient (include/ngan.h: ngan_bf16_conv3x3_wgrad)
            _C.call("ngan_bf16_conv3x3_wgrad", x, g, gw, ws, b, h, w, cin, cout, resample, float(scale), accumulate)
        else:
            _C.call("ngan_conv3x3_wgrad", x, g, gw, ws, b, h, w, cin, cout, resample, float(scale), accumulate, prec)
    if accumulate_into is not None and _defer_depth > 0:
        launch(2)
        plan = _C.wgrad_plan(b, h, w, cin, cout, prec)
        e = _pending.setdefault(gw.data_ptr(), dict(gw=gw, plan=plan, cin=cin, sources=[]))
        e["sources"].append((ws, plan[0], float(scale), role))
        return gw
    launch(1 if accumulate_into is not None else 0)
    return gw


def _accumulates_in_place(weight):
    """True when a weight gradient can be added straight into weight.grad by the kernel: a plain (not create_graph)
    backward into a leaf parameter whose .grad buffer already exists (the step driver's flat gradient views).
    Saves one tiny elementwise add launch per weight per backward (autograd's AccumulateGrad)."""
    return (not torch.is_grad_enabled()) and weight.is_leaf and weight.grad is not None and weight.grad.is_contiguous() \
        and weight.grad.dtype == torch.float32


# Small parameter gradients (biases, FromImage / ToImage / head weights) are added into the parameter's existing .grad buffer by the
# kernel that computes them -- like the conv weight gradients -- instead of being handed to autograd, whose AccumulateGrad adds each
# with an elementwise launch of its own (13 such launches per iteration in the round-3 kernel trace)
_small_grads_in_place = _diag_env("NGAN_SMALL_GRADS_IN_PLACE", "1") != "0"


def _channel_sum(g, into=None):
    c = g.shape[-1]
    npix = g.numel() // c
    out = into if into is not None else torch.empty(c, device=g.device, dtype=torch.float32)
    ws = torch.empty(1024 * c, device=g.device, dtype=torch.float32)
    if into is not None and (c // 4) & (c // 4 - 1) == 0 and c <= 256:
        _C.call(_k("ngan_channel_sum_acc", g), g, out, ws, npix, c, 1.0, 1)
        return out
    if into is not None:
        tmp = torch.empty(c, device=g.device, dtype=torch.float32)
        _C.call(_k("ngan_channel_sum", g), g, tmp, ws, npix, c, 1.0)
        into.add_(tmp)
        return into
    _C.call(_k("ngan_channel_sum", g), g, out, ws, npix, c, 1.0)
    return out


# ---------------------------------------------------------------------------------------------------------
# 3x3 convolution family
# ---------------------------------------------------------------------------------------------------------
def _conv_backward_tail(ctx, x, weight, gc, resample, scale, in_link, has_bias, bias_index=2, pooled=None):
    """input / weight / bias gradients of a conv given the gradient gc w.r.t. its pre-activation (shared by the fused forms)"""
    gx = None
    if ctx.needs_input_grad[0]:
        if in_link is not None and not torch.is_grad_enabled():
            gx = _run_dgrad(_c(gc), weight, resample, scale, link=in_link)      # returns d/d(pre-activation of the PRODUCER)
            in_link.fused = True
        else:
            if in_link is not None:
                in_link.fused = False        # (a create_graph pass after a plain one on a retained graph)
            gx = ConvDgrad.apply(gc, weight, resample, scale)
    gw = None
    if ctx.needs_input_grad[1] and _param_grads_wanted():
        if _accumulates_in_place(weight):
            _run_wgrad(x, _c(gc), resample, scale, accumulate_into=weight.grad, pooled=pooled)   # weight.grad += ..., returns None to autograd
        else:
            gw = ConvWgrad.apply(x, gc, resample, scale)
    gb = None
    if has_bias and ctx.needs_input_grad[bias_index] and _param_grads_wanted():
        bias = getattr(ctx, "bias_param", None)
        if bias is not None and _accumulates_in_place(bias) and _small_grads_in_place:
            _channel_sum(_c(gc), into=bias.grad)          # bias.grad += ...: no gradient tensor for autograd to add with an elementwise launch
        else:
            gb = ChannelSum.apply(gc)
    return gx, gw, gb


class ConvLReLUPN(Function):
    """(y, rnorm) = PixelNorm(LeakyReLU(conv3x3(resample(x), scale*W) + bias)): one fused kernel forward.
    Optional `in_link` / `out_link` (PNLink): see `first_order_only`."""

    @staticmethod
    def forward(ctx, x, weight, bias, resample, scale, slope, in_link=None, out_link=None, pool_out=False):
        ctx.set_materialize_grads(False)
        x = _c(x)
        kept = []
        y, rn = _run_conv(x, weight, bias, resample, scale, 1, slope, keep_pooled=kept, pool_out=pool_out)
        ctx.save_for_backward(x, weight, y, rn)
        ctx.pooled = kept[0] if kept else None        # (an intermediate of this node, not an input: held outside saved_tensors)
        ctx.has_bias = bias is not None
        ctx.bias_param = bias if isinstance(bias, torch.nn.Parameter) else None
        ctx.cfg = (resample, scale, slope)
        ctx.n_in = 6 + (3 if pool_out else (in_link is not None or out_link is not None) * 2)
        if in_link is not None and (in_link.y.data_ptr() != x.data_ptr() or in_link.y.shape != x.shape):
            raise RuntimeError("PixelNorm hand-off: the conv's input is not the linked producer's output")
        ctx.in_link, ctx.out_link = in_link, out_link
        if out_link is not None:
            out_link.y, out_link.rn, out_link.slope, out_link.fused = y.detach(), rn.detach(), slope, False   # aliases without grad_fn
        ctx.stash = GradStash()
        return y, rn

    @staticmethod
    def backward(ctx, gy, gr):
        x, weight, y, rn = ctx.saved_tensors
        resample, scale, slope = ctx.cfg
        pad = (None,) * (ctx.n_in - 3)
        extra, ctx.stash.pending = ctx.stash.pending, None      # contribution left by this layer's LReLUPNBwd node (GradStash)
        if gy is None and gr is None and extra is None:
            return (None, None, None) + pad
        if ctx.out_link is not None and ctx.out_link.fused:
            if gr is not None or extra is not None:
                raise RuntimeError("PixelNorm hand-off used together with a second-order gradient")
            gc = gy       # the consumer's input-gradient kernel already applied this layer's LeakyReLU->PixelNorm backward
        elif torch.is_grad_enabled():
            # create_graph pass: a differentiable node; its backward will leave its gradient w.r.t. y in the stash
            if gy is None:
                gy = torch.zeros_like(y)
            gc = LReLUPNBwd.apply(gy, gr, y, rn, slope, ctx.stash, None)
        else:
            if gy is None:
                gy, extra = (extra, None) if extra is not None else (torch.zeros_like(y), None)
            gc = LReLUPNBwd.apply(gy, gr, y, rn, slope, None, extra) if extra is not None else LReLUPNBwd.apply(gy, gr, y, rn, slope)
        gx, gw, gb = _conv_backward_tail(ctx, x, weight, gc, resample, scale, ctx.in_link, ctx.has_bias, pooled=ctx.pooled)
        return (gx, gw, gb) + pad


class ConvLReLUPNToImage(Function):
    """t = tanh(conv1x1(PixelNorm(LeakyReLU(conv3x3(x, scale*W) + bias)), w_img)): the generator's last block conv and ToImage
    (models.py:141-146, 262-264) as ONE kernel; the C-channel activation is written only when a backward pass will need it.
    First order only (it is a generator operator); one colour channel; shapes: `to_image_fusable`."""

    @staticmethod
    def forward(ctx, x, weight, bias, w_img, resample, scale, slope, in_link=None, keep=True):
        # keep: will a backward pass follow?  (the caller passes torch.is_grad_enabled(): inside forward() grad mode is always
        # off and needs_input_grad mirrors requires_grad even under no_grad)
        x = _c(x)
        b, h, w = _conv_out_hw(x, resample)
        cout, cin = weight.shape[0], weight.shape[1]
        keep = bool(keep) and any(ctx.needs_input_grad)
        y = torch.empty((b, h, w, cout), device=x.device, dtype=x.dtype) if keep else None
        rn = torch.empty((b, h, w), device=x.device, dtype=torch.float32) if keep else None
        t = torch.empty((b, h, w, 1), device=x.device, dtype=torch.float32)
        if x.dtype == torch.bfloat16:
            _C.call("ngan_bf16_conv3x3_fwd", x, _packed(weight, 0, scale, _bf16_prec(b, h, w, cin, cout, resample)), bias, y, rn,
                    w_img.detach().reshape(-1), None, t, b, h, w, cin, cout, resample, EPI_TO_IMAGE, 0, float(slope), PIXELNORM_EPS)
        else:
            prec = _C.conv3x3_algorithm(b, h, w, cin, cout, resample, _conv_precision)
            _C.call("ngan_conv3x3_fwd_ex", x, _packed(weight, 0, scale, prec), bias, y, rn, w_img.detach().reshape(-1), None, t,
                    b, h, w, cin, cout, resample, EPI_TO_IMAGE, 0, float(slope), PIXELNORM_EPS, prec, 0)
        if keep:
            ctx.save_for_backward(x, weight, y, rn, t, w_img)
        ctx.has_bias = bias is not None
        ctx.bias_param = bias if isinstance(bias, torch.nn.Parameter) else None
        ctx.cfg = (resample, scale, slope)
        if in_link is not None and (in_link.y.data_ptr() != x.data_ptr() or in_link.y.shape != x.shape):
            raise RuntimeError("PixelNorm hand-off: the conv's input is not the linked producer's output")
        ctx.in_link = in_link
        return t

    @staticmethod
    @once_differentiable
    def backward(ctx, gt):
        x, weight, y, rn, t, w_img = ctx.saved_tensors
        resample, scale, slope = ctx.cfg
        c = y.shape[-1]
        npix = y.numel() // c
        ws = torch.empty(1024 * c, device=y.device, dtype=torch.float32)
        gc = torch.empty_like(y)      # ToImage backward and the LeakyReLU->PixelNorm backward in one pass over y
        if (_small_grads_in_place and ctx.needs_input_grad[3] and _accumulates_in_place(w_img) and (c // 4) & (c // 4 - 1) == 0
                and c <= 256):
            _C.call(_k("ngan_to_image_bwd_pnbwd_acc", y), _c(gt), t, y, rn, w_img.detach().reshape(1, c), gc, w_img.grad, ws, npix, c, 1, float(slope), 1)
            gw_img = None             # w_img.grad += ... inside the reduction
        else:
            gw_img = torch.empty_like(w_img)
            _C.call(_k("ngan_to_image_bwd_pnbwd", y), _c(gt), t, y, rn, w_img.detach().reshape(1, c), gc, gw_img, ws, npix, c, 1, float(slope))
            if not ctx.needs_input_grad[3]:
                gw_img = None         # a frozen colour weight: computed by the fused pass, handed to nobody
        gx, gw, gb = _conv_backward_tail(ctx, x, weight, gc, resample, scale, ctx.in_link, ctx.has_bias)
        return gx, gw, gb, gw_img, None, None, None, None, None


def to_image_fusable(x, weight, w_img, resample):
    """can ConvLReLUPNToImage take this layer?  (persistent-kernel shapes, one colour, no resampling)"""
    if w_img.shape[0] != 1 or resample != RES_NONE:
        return False
    b, h, w = _conv_out_hw(x, resample)
    cout, cin = weight.shape[0], weight.shape[1]
    prec = 5 if x.dtype == torch.bfloat16 else _C.conv3x3_algorithm(b, h, w, cin, cout, resample, _conv_precision)
    return _C.conv3x3_epilogue_fused(b, h, w, cin, cout, resample, EPI_TO_IMAGE, 0, prec)


class Conv(Function):
    """c = conv3x3(resample(x), scale*W) + bias, no activation."""

    @staticmethod
    def forward(ctx, x, weight, bias, resample, scale):
        x = _c(x)
        y, _ = _run_conv(x, weight, bias, resample, scale, 0, 0.0)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.cfg = (resample, scale)
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        resample, scale = ctx.cfg
        gx = ConvDgrad.apply(g, weight, resample, scale) if ctx.needs_input_grad[0] else None
        gw = ConvWgrad.apply(x, g, resample, scale) if (ctx.needs_input_grad[1] and _param_grads_wanted()) else None
        gb = ChannelSum.apply(g) if (ctx.has_bias and ctx.needs_input_grad[2] and _param_grads_wanted()) else None
        return gx, gw, gb, None, None


class ConvDgrad(Function):
    """gx = resample^T(conv3x3^T(g, scale*W)); linear in g and in W."""

    @staticmethod
    def forward(ctx, g, weight, resample, scale):
        g = _c(g)
        ctx.save_for_backward(g, weight)
        ctx.cfg = (resample, scale)
        return _run_dgrad(g, weight, resample, scale)

    @staticmethod
    def backward(ctx, h):
        g, weight = ctx.saved_tensors
        resample, scale = ctx.cfg
        kept = []
        if not ctx.needs_input_grad[0]:
            gg = None
        elif torch.is_grad_enabled():
            gg = Conv.apply(h, weight, None, resample, scale)
        else:       # the usual case (the penalty's second pass): no node to record, and the pooled copy of h serves the weight gradient too
            gg = _run_conv(_c(h), weight, None, resample, scale, 0, 0.0, keep_pooled=kept)[0]
        gw = None
        if ctx.needs_input_grad[1]:
            if _accumulates_in_place(weight):
                _run_wgrad(_c(h), g, resample, scale, accumulate_into=weight.grad, role=1, pooled=kept[0] if kept else None)
            else:
                gw = ConvWgrad.apply(h, g, resample, scale)
        return gg, gw, None, None


class ConvWgrad(Function):
    """gW (OIHW) = scale * sum_pixels g (x) resample(x); linear in x and in g."""

    @staticmethod
    def forward(ctx, x, g, resample, scale):
        x, g = _c(x), _c(g)
        ctx.save_for_backward(x, g)
        ctx.cfg = (resample, scale)
        return _run_wgrad(x, g, resample, scale)

    @staticmethod
    def backward(ctx, hw):
        x, g = ctx.saved_tensors
        resample, scale = ctx.cfg
        hw = _c(hw)
        gx = ConvDgrad.apply(g, hw, resample, scale) if ctx.needs_input_grad[0] else None
        gg = Conv.apply(x, hw, None, resample, scale) if ctx.needs_input_grad[1] else None
        return gx, gg, None, None


class ChannelSum(Function):
    """out[c] = sum over pixels of g[..., c] (bias gradient)."""

    @staticmethod
    def forward(ctx, g):
        g = _c(g)
        ctx.shape = g.shape
        return _channel_sum(g)

    @staticmethod
    def backward(ctx, h):
        return h.expand(ctx.shape)


# ---------------------------------------------------------------------------------------------------------
# LeakyReLU -> PixelNorm
# ---------------------------------------------------------------------------------------------------------
class LReLUPN(Function):
    """(y, rnorm) = PixelNorm(LeakyReLU(c + bias)) as a standalone kernel."""

    @staticmethod
    def forward(ctx, c, bias, slope):
        ctx.set_materialize_grads(False)
        c = _c(c)
        ch = c.shape[-1]
        y = torch.empty_like(c)
        rn = torch.empty(c.shape[:-1], device=c.device, dtype=torch.float32)
        _C.call(_k("ngan_lrelu_pixelnorm_fwd", c), c, bias, y, rn, c.numel() // ch, ch, float(slope), PIXELNORM_EPS)
        ctx.save_for_backward(y, rn)
        ctx.slope = slope
        ctx.has_bias = bias is not None
        return y, rn

    @staticmethod
    def backward(ctx, gy, gr):
        y, rn = ctx.saved_tensors
        if gy is None and gr is None:
            return None, None, None
        if gy is None:
            gy = torch.zeros_like(y)
        gc = LReLUPNBwd.apply(gy, gr, y, rn, ctx.slope)
        gb = ChannelSum.apply(gc) if (ctx.has_bias and ctx.needs_input_grad[1] and _param_grads_wanted()) else None
        return gc, gb, None


class GradStash:
    """Side channel for the gradient penalty's backward pass.  The gradient w.r.t. a layer output y has two contributions there:
    one from the next conv and one from the backward of this layer's own LReLUPNBwd node (y is an input of that node).  Autograd
    would add them with an elementwise kernel before calling the layer's backward.  Instead LReLUPNBwd.backward leaves its
    contribution here and returns None for y, and the layer's backward (which autograd runs afterwards: the node also feeds the
    layer's norm output, so it is a graph predecessor) hands both tensors to ONE PixelNorm-backward launch that sums them."""
    __slots__ = ("pending",)

    def __init__(self):
        self.pending = None


class LReLUPNBwd(Function):
    """gc = m * ((gy - y*mean_c(gy*y))/r + gr*y/C): first-order backward of LeakyReLU->PixelNorm."""

    @staticmethod
    def forward(ctx, gy, gr, y, rn, slope, stash=None, gy2=None):
        gy, gr = _c(gy), _c(gr)
        ch = y.shape[-1]
        gc = torch.empty_like(y)
        if gy.dtype != y.dtype:
            gy = gy.to(y.dtype)
        _C.call(_k("ngan_lrelu_pixelnorm_bwd2", y), gy, _c(gy2), gr, y, rn, gc, y.numel() // ch, ch, float(slope))
        if gy2 is None:            # (gy2 only comes from a no-grad pass, which records no graph: see ConvLReLUPN.backward)
            ctx.save_for_backward(gy, y, rn)
        ctx.had_gr = gr is not None
        ctx.slope = slope
        ctx.stash = stash
        ctx.n_in = 5 + (stash is not None or gy2 is not None) * 2
        return gc

    @staticmethod
    @once_differentiable
    def backward(ctx, h):
        if ctx.had_gr:
            raise NotImplementedError("third-order derivative through LeakyReLU->PixelNorm is not on the WGAN-GP path")
        gy, y, rn = ctx.saved_tensors
        h = _c(h)
        ch = y.shape[-1]
        ggy = torch.empty_like(y)
        gy_out = torch.empty_like(y)
        gr_out = torch.empty_like(rn)
        _C.call(_k("ngan_lrelu_pixelnorm_bwdbwd", y), h, gy, y, rn, ggy, gy_out, gr_out, y.numel() // ch, ch, float(ctx.slope))
        pad = (None,) * (ctx.n_in - 5)
        if ctx.stash is not None and ctx.stash.pending is None:
            ctx.stash.pending = gy_out          # picked up by the producing layer's backward (GradStash)
            return (ggy, None, None, gr_out, None) + pad
        return (ggy, None, gy_out, gr_out, None) + pad


# ---------------------------------------------------------------------------------------------------------
# FromImage (1x1 conv from colour space, + bias), optional avg-pool on load for the fade-in branch
# ---------------------------------------------------------------------------------------------------------
def _from_image_out_hw(x, pool):
    b, h, w, _ = x.shape
    return (b, h // 2, w // 2) if pool else (b, h, w)


class FromImage(Function):
    @staticmethod
    def forward(ctx, x, w, bias, pool):
        x = _c(x)
        b, h, wd = _from_image_out_hw(x, pool)
        c, ncol = w.shape[0], w.shape[1]
        y = torch.empty((b, h, wd, c), device=x.device, dtype=act_dtype())
        _C.call(_k("ngan_from_image_fwd", y), x, w.detach().reshape(c, ncol), bias, y, b, h, wd, ncol, c, int(pool))
        ctx.save_for_backward(x, w)
        ctx.pool = pool
        ctx.has_bias = bias is not None
        ctx.bias_param = bias if isinstance(bias, torch.nn.Parameter) else None
        return y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gx = FromImageDx.apply(g, w, ctx.pool) if ctx.needs_input_grad[0] else None
        gw = gb = None
        if (ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])) and _param_grads_wanted():
            bias = ctx.bias_param
            c, ncol = w.shape[0], w.shape[1]
            want_b = ctx.has_bias and ctx.needs_input_grad[2]
            if (_small_grads_in_place and ctx.needs_input_grad[1] and _accumulates_in_place(w) and (c // 4) & (c // 4 - 1) == 0 and c <= 256
                    and (not want_b or (bias is not None and _accumulates_in_place(bias)))):
                # (a frozen bias next to a trainable weight: its sum is not formed at all -- gb pointer NULL)
                gg = _c(g)
                b, h, wd, _ = gg.shape
                ws = torch.empty(1024 * c * (ncol + 1), device=gg.device, dtype=torch.float32)
                _C.call(_k("ngan_from_image_dw_acc", gg), x, gg, w.grad, bias.grad if want_b else None, ws, b, h, wd, ncol, c, int(ctx.pool), 3)
            else:
                gw, gb = FromImageDw.apply(x, g, ctx.pool, tuple(w.shape))
                if not ctx.needs_input_grad[1]:
                    gw = None
                if not want_b:
                    gb = None
        return gx, gw, gb, None


_first_block_allowed = _diag_env("NGAN_FIRST_BLOCK", "1") != "0"      # A/B switch for measurements


def first_block_fusable(x, w_from, w_conv):
    """can FirstBlock take FromImage -> conv3x3 -> LeakyReLU -> PixelNorm?  (one colour channel, 16 or 32 conv outputs; fp32 storage:
    the folded kernels write fp32 activations, the bf16 mode runs the two layers as they are)"""
    return (_first_block_allowed and _conv_precision != 5 and x.shape[-1] == 1 and w_from.shape[1] == 1 and w_conv.shape[0] in (16, 32)
            and w_conv.shape[1] <= 64 and x.shape[0] * x.shape[1] < 65536)


class FirstBlock(Function):
    """(y, rnorm) = PixelNorm(LeakyReLU(conv3x3(FromImage(pool?(x)), scale*W) + b)) for a ONE-colour image, first order only:
    FromImage's output is affine in the pixel value, so the pair collapses to a 3x3 conv over one channel whose tables are built
    in the kernel (csrc/first_block.hip); the C-channel tensor between the two layers never exists, and the weight gradients of
    both layers come from two 9 x N tables of pixel sums."""

    @staticmethod
    def forward(ctx, x, w_from, b_from, weight, bias, pool, scale, slope, out_link=None):
        ctx.set_materialize_grads(False)     # (no zero tensor for the norm output's absent gradient: a fill launch per backward)
        x = _c(x)
        b, h, wd = _from_image_out_hw(x, pool)
        n, c = weight.shape[0], weight.shape[1]
        p = _resample("ngan_pool2_fwd", x, (b, h, wd, 1), b, h, wd, 1) if pool else x
        y = torch.empty((b, h, wd, n), device=x.device, dtype=torch.float32)
        rn = torch.empty((b, h, wd), device=x.device, dtype=torch.float32)
        tables = torch.empty(2 * 9 * n, device=x.device, dtype=torch.float32)
        _C.call("ngan_first_block_fwd", p, weight.detach(), w_from.detach().reshape(c), b_from, bias, y, rn, tables, b, h, wd, c, n,
                float(scale), float(slope), PIXELNORM_EPS)
        ctx.save_for_backward(p, w_from, b_from, weight, y, rn, tables)
        ctx.bias_param = bias if isinstance(bias, torch.nn.Parameter) else None
        ctx.cfg = (pool, scale, slope, bias is not None)
        ctx.out_link = out_link
        if out_link is not None:
            out_link.y, out_link.rn, out_link.slope, out_link.fused = y.detach(), rn.detach(), slope, False
        ctx.mark_non_differentiable(rn)
        return y, rn

    @staticmethod
    @once_differentiable
    def backward(ctx, gy, gr):
        p, w_from, b_from, weight, y, rn, tables = ctx.saved_tensors
        pool, scale, slope, has_bias = ctx.cfg
        if gy is None:
            return (None,) * 9
        b, h, wd, n = y.shape
        c = weight.shape[1]
        if ctx.out_link is not None and ctx.out_link.fused:
            gc = _c(gy)         # the consumer's input-gradient kernel already applied this layer's LeakyReLU->PixelNorm backward
        else:
            gc = torch.empty_like(y)
            _C.call(_k("ngan_lrelu_pixelnorm_bwd", y), _c(gy), None, y, rn, gc, y.numel() // n, n, float(slope))
        dev = y.device
        in_place = False
        gw = gwf = gbf = gb = None
        if any(ctx.needs_input_grad[1:5]) and _param_grads_wanted():     # (not in the generator step: the critic is frozen there)
            ws = torch.empty(_C.lib().ngan_first_block_workspace_floats(b, h, n), device=dev, dtype=torch.float32)
            in_place = ctx.needs_input_grad[3] and _accumulates_in_place(weight)
            gw = weight.grad if in_place else torch.empty_like(weight)
            # the three small gradients likewise (accumulate bit mask of ngan_first_block_bwd: 1 conv weight, 2 / 4 FromImage weight / bias, 8 conv bias)
            bias = ctx.bias_param
            ip_wf = _small_grads_in_place and ctx.needs_input_grad[1] and _accumulates_in_place(w_from)
            ip_bf = _small_grads_in_place and b_from is not None and ctx.needs_input_grad[2] and _accumulates_in_place(b_from)
            ip_b = _small_grads_in_place and has_bias and bias is not None and ctx.needs_input_grad[4] and _accumulates_in_place(bias)
            gwf = w_from.grad.view(w_from.shape) if ip_wf else torch.empty_like(w_from)
            gbf = (b_from.grad if ip_bf else torch.empty_like(b_from)) if b_from is not None else None
            gb = (bias.grad if ip_b else torch.empty(n, device=dev, dtype=torch.float32)) if has_bias else None
            mask = (1 if in_place else 0) | (2 if ip_wf else 0) | (4 if ip_bf else 0) | (8 if ip_b else 0)
            _C.call("ngan_first_block_bwd", p, gc, weight.detach(), w_from.detach().reshape(c), b_from, gw, gwf, gbf, gb, ws,
                    b, h, wd, c, n, float(scale), mask)
            if ip_wf:
                gwf = None
            if ip_bf:
                gbf = None
            if ip_b:
                gb = None
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty((b, 2 * h, 2 * wd, 1) if pool else (b, h, wd, 1), device=dev, dtype=torch.float32)
            _C.call("ngan_first_block_dx", gc, tables, gx, b, h, wd, n, int(pool))
        return gx, gwf, gbf, (None if in_place else gw), gb, None, None, None, None


class FromImageDx(Function):
    @staticmethod
    def forward(ctx, g, w, pool):
        g = _c(g)
        b, h, wd, c = g.shape
        ncol = w.shape[1]
        gx = torch.empty((b, 2 * h, 2 * wd, ncol) if pool else (b, h, wd, ncol), device=g.device, dtype=torch.float32)
        _C.call(_k("ngan_from_image_dx", g), g, w.detach().reshape(c, ncol), gx, b, h, wd, ncol, c, int(pool))
        ctx.save_for_backward(g, w)
        ctx.pool = pool
        return gx

    @staticmethod
    def backward(ctx, h):
        g, w = ctx.saved_tensors
        gg = FromImage.apply(h, w, None, ctx.pool) if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            c, ncol = w.shape[0], w.shape[1]
            if _small_grads_in_place and _accumulates_in_place(w) and (c // 4) & (c // 4 - 1) == 0 and c <= 256:
                hh = _c(h)
                b, hgt, wd, _ = g.shape
                ws = torch.empty(1024 * c * (ncol + 1), device=g.device, dtype=torch.float32)
                _C.call(_k("ngan_from_image_dw_acc", g), hh, g, w.grad, None, ws, b, hgt, wd, ncol, c, int(ctx.pool), 1)
            else:
                gw = FromImageDw.apply(h, g, ctx.pool, tuple(w.shape))[0]
        return gg, gw, None


class FromImageDw(Function):
    @staticmethod
    def forward(ctx, x, g, pool, w_shape):
        x, g = _c(x), _c(g)
        b, h, wd, c = g.shape
        ncol = x.shape[3]
        gw = torch.empty(w_shape, device=g.device, dtype=torch.float32)
        gb = torch.empty(c, device=g.device, dtype=torch.float32)
        ws = torch.empty(1024 * c * (ncol + 1), device=g.device, dtype=torch.float32)
        _C.call(_k("ngan_from_image_dw", g), x, g, gw, gb, ws, b, h, wd, ncol, c, int(pool))
        ctx.save_for_backward(x, g)
        ctx.pool = pool
        return gw, gb

    @staticmethod
    def backward(ctx, hw, hb):
        x, g = ctx.saved_tensors
        gx = gg = None
        if hw is not None:
            hw = _c(hw)
            if ctx.needs_input_grad[0]:
                gx = FromImageDx.apply(g, hw, ctx.pool)
            if ctx.needs_input_grad[1]:
                gg = FromImage.apply(x, hw, _c(hb), ctx.pool)
        elif hb is not None and ctx.needs_input_grad[1]:
            gg = hb.expand(g.shape)
        return gx, gg, None, None


# ---------------------------------------------------------------------------------------------------------
# ToImage (1x1 conv to colour space + tanh): generator only, first order
# ---------------------------------------------------------------------------------------------------------
class ToImage(Function):
    @staticmethod
    def forward(ctx, x, w):
        x = _c(x)
        ncol, c = w.shape[0], w.shape[1]
        t = torch.empty(x.shape[:-1] + (ncol,), device=x.device, dtype=torch.float32)
        _C.call(_k("ngan_to_image_fwd", x), x, w.detach().reshape(ncol, c), t, x.numel() // c, c, ncol)
        ctx.save_for_backward(x, w, t)
        return t

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, w, t = ctx.saved_tensors
        g = _c(g)
        ncol, c = w.shape[0], w.shape[1]
        gx = torch.empty_like(x)
        gw = torch.empty_like(w)
        ws = torch.empty(1024 * c * ncol, device=x.device, dtype=torch.float32)
        _C.call(_k("ngan_to_image_bwd", x), g, t, x, w.detach().reshape(ncol, c), gx, gw, ws, x.numel() // c, c, ncol)
        return gx, gw


# ---------------------------------------------------------------------------------------------------------
# resampling and fade-in (linear pairs)
# ---------------------------------------------------------------------------------------------------------
def _resample(name, x, out_shape, b, h, w, c):
    out = torch.empty(out_shape, device=x.device, dtype=x.dtype)
    _C.call(_k(name, x), x, out, b, h, w, c)
    return out


class Up2(Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        b, h, w, c = x.shape
        return _resample("ngan_up2_fwd", x, (b, 2 * h, 2 * w, c), b, h, w, c)

    @staticmethod
    def backward(ctx, g):
        return Up2Adjoint.apply(g)


class Up2Adjoint(Function):
    @staticmethod
    def forward(ctx, g):
        g = _c(g)
        b, h2, w2, c = g.shape
        return _resample("ngan_up2_adjoint", g, (b, h2 // 2, w2 // 2, c), b, h2 // 2, w2 // 2, c)

    @staticmethod
    def backward(ctx, h):
        return Up2.apply(h)


class Pool2(Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        b, h2, w2, c = x.shape
        return _resample("ngan_pool2_fwd", x, (b, h2 // 2, w2 // 2, c), b, h2 // 2, w2 // 2, c)

    @staticmethod
    def backward(ctx, g):
        return Pool2Adjoint.apply(g)


class Pool2Adjoint(Function):
    @staticmethod
    def forward(ctx, g):
        g = _c(g)
        b, h, w, c = g.shape
        return _resample("ngan_pool2_adjoint", g, (b, 2 * h, 2 * w, c), b, h, w, c)

    @staticmethod
    def backward(ctx, h):
        return Pool2.apply(h)


class Lerp(Function):
    """out = a + alpha*(b - a); alpha is a 1-element device tensor (no gradient)."""

    @staticmethod
    def forward(ctx, a, b, alpha):
        a, b = _c(a), _c(b)
        out = torch.empty_like(a)
        if a.dtype != b.dtype:
            raise RuntimeError(f"lerp: operands differ in storage type ({a.dtype}, {b.dtype})")
        _C.call(_k("ngan_lerp", a), a, b, alpha, out, a.numel())
        ctx.save_for_backward(alpha)
        return out

    @staticmethod
    def backward(ctx, g):
        (alpha,) = ctx.saved_tensors
        ga, gb = FadeBwd.apply(g, alpha)
        return ga, gb, None


class FadeBwd(Function):
    @staticmethod
    def forward(ctx, g, alpha):
        g = _c(g)
        ga, gb = torch.empty_like(g), torch.empty_like(g)
        _C.call(_k("ngan_fade_bwd", g), g, alpha, ga, gb, g.numel())
        ctx.save_for_backward(alpha)
        return ga, gb

    @staticmethod
    def backward(ctx, ha, hb):
        (alpha,) = ctx.saved_tensors
        if ha is None and hb is None:
            return None, None
        if ha is None:
            ha = torch.zeros_like(hb)
        if hb is None:
            hb = torch.zeros_like(ha)
        return Lerp.apply(ha, hb, alpha), None


# ---------------------------------------------------------------------------------------------------------
# generator stem and critic head
# ---------------------------------------------------------------------------------------------------------
# Data-parallel hook for the generator stem: when set, LinearLReLUPN.backward hands (z, gc, weight, scale) to the sink instead
# of forming the (C*S, K) weight gradient itself -- the step driver all-gathers the rank-B factors and forms the full-batch
# gradient locally (train.StemGradExchange), which moves ~2 MB per rank instead of all-reducing 67 MB.
linear_grad_sink = None


class LinearLReLUPN(Function):
    """y (B,S,S,C) = PixelNorm(LeakyReLU(Unflatten(Linear(scale*z, W)))): generator stem, first order."""

    @staticmethod
    def forward(ctx, z, weight, size, scale, slope, out_link=None):
        ctx.set_materialize_grads(False)
        z = _c(z)
        b, k = z.shape
        s2 = size * size
        c = weight.shape[0] // s2
        y = torch.empty((b, size, size, c), device=z.device, dtype=act_dtype())
        rn = torch.empty((b, size, size), device=z.device, dtype=torch.float32)
        _C.call(_k("ngan_linear_lrelu_pn_fwd", y), z, weight.detach(), y, rn, b, k, s2, c, float(scale), float(slope), PIXELNORM_EPS)
        ctx.save_for_backward(z, weight, y, rn)
        ctx.cfg = (s2, c, scale, slope)
        ctx.mark_non_differentiable(rn)
        ctx.n_in = 5 + (out_link is not None)
        ctx.out_link = out_link
        if out_link is not None:
            out_link.y, out_link.rn, out_link.slope, out_link.fused = y.detach(), rn.detach(), slope, False   # aliases without grad_fn
        return y, rn

    @staticmethod
    @once_differentiable
    def backward(ctx, gy, _gr):
        z, weight, y, rn = ctx.saved_tensors
        s2, c, scale, slope = ctx.cfg
        if gy is None:
            return (None,) * ctx.n_in
        b, k = z.shape
        gy = _c(gy)
        if ctx.out_link is not None and ctx.out_link.fused:
            gc = gy       # the consuming conv's input-gradient kernel already applied the LeakyReLU->PixelNorm backward
        else:
            gc = torch.empty_like(y)
            _C.call(_k("ngan_lrelu_pixelnorm_bwd", y), gy, None, y, rn, gc, b * s2, c, float(slope))
        gz = gw = None
        if ctx.needs_input_grad[1]:
            if linear_grad_sink is not None:
                linear_grad_sink(z, gc, weight, s2, c, scale)      # gradient is formed later from the gathered factors
            elif _accumulates_in_place(weight) and k % 16 == 0 and k <= 512:
                # weight.grad += ... inside the kernel: no 67 MB temporary and no separate 200 MB accumulate pass
                _C.call(_k("ngan_linear_wgrad_acc", gc), z, gc, weight.grad, b, k, s2, c, float(scale), 1)
            else:
                gw = torch.empty_like(weight)
                _C.call(_k("ngan_linear_wgrad", gc), z, gc, gw, b, k, s2, c, float(scale))
        if ctx.needs_input_grad[0]:
            gz = torch.empty_like(z)
            _C.call(_k("ngan_linear_dgrad", gc), gc, weight.detach(), gz, b, k, s2, c, float(scale))
        return (gz, gw, None, None, None) + (None,) * (ctx.n_in - 5)


class FinalDot(Function):
    """out (B,1) = scale * <y[b], W> + bias: the critic's full-extent valid conv."""

    @staticmethod
    def forward(ctx, y, weight, bias, scale):
        y = _c(y)
        b = y.shape[0]
        c = y.shape[3]
        s2 = y.shape[1] * y.shape[2]
        out = torch.empty((b, 1), device=y.device, dtype=torch.float32)
        _C.call(_k("ngan_final_dot_fwd", y), y, weight.detach(), bias, out, b, s2, c, float(scale))
        ctx.save_for_backward(y, weight)
        ctx.scale = scale
        ctx.has_bias = bias is not None
        ctx.bias_param = bias if isinstance(bias, torch.nn.Parameter) else None
        return out

    @staticmethod
    def backward(ctx, go):
        y, weight = ctx.saved_tensors
        gy = FinalDotDx.apply(go, weight, ctx.scale, tuple(y.shape), y.dtype) if ctx.needs_input_grad[0] else None
        gw = gb = None
        if (ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])) and _param_grads_wanted():
            bias = ctx.bias_param
            want_b = ctx.has_bias and ctx.needs_input_grad[2]
            if (_small_grads_in_place and ctx.needs_input_grad[1] and _accumulates_in_place(weight)
                    and (not want_b or (bias is not None and _accumulates_in_place(bias)))):
                b, hh, ww, c = y.shape
                _C.call(_k("ngan_final_dot_dw_acc", y), y, _c(go), weight.grad, bias.grad if want_b else None, b, hh * ww, c, float(ctx.scale), 3)
            else:
                gw, gb = FinalDotDw.apply(y, go, ctx.scale, tuple(weight.shape))
                if not ctx.needs_input_grad[1]:
                    gw = None
                if not want_b:
                    gb = None
        return gy, gw, gb, None


class FinalDotDx(Function):
    @staticmethod
    def forward(ctx, go, weight, scale, y_shape, y_dtype=torch.float32):
        go = _c(go)
        b, h, w, c = y_shape
        gy = torch.empty(y_shape, device=go.device, dtype=y_dtype)
        _C.call(_k("ngan_final_dot_dx", gy), go, weight.detach(), gy, b, h * w, c, float(scale))
        ctx.save_for_backward(go, weight)
        ctx.scale = scale
        return gy

    @staticmethod
    def backward(ctx, h):
        go, weight = ctx.saved_tensors
        ggo = FinalDot.apply(h, weight, None, ctx.scale) if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            if _small_grads_in_place and _accumulates_in_place(weight):
                hh = _c(h)
                b, hgt, wd, c = hh.shape
                _C.call(_k("ngan_final_dot_dw_acc", hh), hh, go, weight.grad, None, b, hgt * wd, c, float(ctx.scale), 1)
            else:
                gw = FinalDotDw.apply(h, go, ctx.scale, tuple(weight.shape))[0]
        return ggo, gw, None, None, None


class FinalDotDw(Function):
    @staticmethod
    def forward(ctx, y, go, scale, w_shape):
        y, go = _c(y), _c(go)
        b, h, w, c = y.shape
        gw = torch.empty(w_shape, device=y.device, dtype=torch.float32)
        gb = torch.empty(1, device=y.device, dtype=torch.float32)
        _C.call(_k("ngan_final_dot_dw", y), y, go, gw, gb, b, h * w, c, float(scale))
        ctx.save_for_backward(y, go)
        ctx.scale = scale
        return gw, gb

    @staticmethod
    def backward(ctx, hw, hb):
        y, go = ctx.saved_tensors
        gy = ggo = None
        if hw is not None:
            hw = _c(hw)
            if ctx.needs_input_grad[0]:
                gy = FinalDotDx.apply(go, hw, ctx.scale, tuple(y.shape), y.dtype)
            if ctx.needs_input_grad[1]:
                ggo = FinalDot.apply(y, hw, None, ctx.scale)
        if hb is not None and ctx.needs_input_grad[1]:
            e = hb.expand(go.shape)
            ggo = e if ggo is None else ggo + e
        return gy, ggo, None, None


# ---------------------------------------------------------------------------------------------------------
# gradient-penalty pieces
# ---------------------------------------------------------------------------------------------------------
def xhat(real, fake, eps):
    """x_hat[b] = eps[b]*real[b] + (1-eps[b])*fake[b]  (loss_functions.py:171); inputs carry no gradient."""
    real, fake = _c(real), _c(fake)
    b = real.shape[0]
    out = torch.empty_like(real)
    _C.call("ngan_xhat", real, fake, _c(eps.reshape(b)), out, b, real.numel() // b)
    return out


class SampleL2Norm(Function):
    """norms[b] = ||g[b]||_2 over all non-batch dims (loss_functions.py:176)."""

    @staticmethod
    def forward(ctx, g):
        g = _c(g)
        b = g.shape[0]
        norms = torch.empty(b, device=g.device, dtype=torch.float32)
        ws = torch.empty(64 * b, device=g.device, dtype=torch.float32)
        _C.call("ngan_sample_l2norm", g, norms, ws, b, g.numel() // b)
        ctx.save_for_backward(g, norms)
        return norms

    @staticmethod
    @once_differentiable
    def backward(ctx, hn):
        g, norms = ctx.saved_tensors
        b = g.shape[0]
        coef = (hn / norms).contiguous()
        out = torch.empty_like(g)
        _C.call("ngan_scale_rows", g, coef, out, b, g.numel() // b)
        return out


# ---------------------------------------------------------------------------------------------------------
# scalar heads of the losses (first order: the loss values are differentiated once, train.py:365, 384)
# ---------------------------------------------------------------------------------------------------------
class WLossHead(Function):
    """(loss, mean real score, mean fake score) of scores = [n_real real | n_fake fake]:
    loss = -mean(real) + mean(fake) + drift * mean(real^2)  (loss_functions.py:21-45); n_fake = 0: -mean(scores), the generator
    loss (loss_functions.py:67).  One launch forward, one backward, instead of ~8 ATen launches each way."""

    @staticmethod
    def forward(ctx, scores, n_real, drift):
        ctx.set_materialize_grads(False)          # (the two means usually carry no gradient: no zero-filled stand-ins)
        scores = _c(scores)
        n = scores.numel()
        # three separate 0-dim tensors, not views of one buffer: the reference's loop modifies the loss in place (`D_loss_val += gp`)
        loss, s_real, s_fake = (torch.empty((), device=scores.device, dtype=torch.float32) for _ in range(3))
        _C.call("ngan_wloss_head", scores, int(n_real), int(n - n_real), float(drift), loss, s_real, s_fake)
        ctx.save_for_backward(scores)
        ctx.cfg = (int(n_real), int(n - n_real), float(drift))
        return loss, s_real, s_fake

    @staticmethod
    @once_differentiable
    def backward(ctx, gl, gr, gf):
        (scores,) = ctx.saved_tensors
        n_real, n_fake, drift = ctx.cfg
        if gl is None and gr is None and gf is None:
            return None, None, None
        gs = torch.empty_like(scores)
        _C.call("ngan_wloss_head_bwd", scores, n_real, n_fake, drift, _c(gl), _c(gr), _c(gf), gs)
        return gs, None, None


class GradPenaltyHead(Function):
    """(Lambda * mean_b((||g_b||_2 - 1)^2), norms) of the critic's input gradient g (loss_functions.py:176): SampleL2Norm and the
    penalty arithmetic as one operator; its backward scales g's rows by 2 Lambda (n_b - 1) / (B n_b)."""

    @staticmethod
    def forward(ctx, g, lam):
        ctx.set_materialize_grads(False)
        g = _c(g)
        b = g.shape[0]
        norms = torch.empty(b, device=g.device, dtype=torch.float32)
        ws = torch.empty(64 * b, device=g.device, dtype=torch.float32)
        _C.call("ngan_sample_l2norm", g, norms, ws, b, g.numel() // b)
        out = torch.empty((), device=g.device, dtype=torch.float32)
        _C.call("ngan_gp_head", norms, b, float(lam), out)
        ctx.save_for_backward(g, norms)
        ctx.lam = float(lam)
        ctx.mark_non_differentiable(norms)
        return out, norms

    @staticmethod
    @once_differentiable
    def backward(ctx, gp, _gn):
        g, norms = ctx.saved_tensors
        if gp is None:
            return None, None
        b = g.shape[0]
        coef = torch.empty(b, device=g.device, dtype=torch.float32)
        _C.call("ngan_gp_coef", norms, b, ctx.lam, _c(gp), coef)
        out = torch.empty_like(g)
        _C.call("ngan_scale_rows", g, coef, out, b, g.numel() // b)
        return out, None


def latent_normalize_(z, clamp=5.0):
    """in place: rows of normal draws -> clamp(-5, 5) -> unit L2 norm (reference utils.py:77-78); one launch"""
    _C.call("ngan_latent_normalize", z, z.shape[0], z.shape[1], float(clamp))
    return z
