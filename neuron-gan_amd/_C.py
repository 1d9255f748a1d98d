"""ctypes binding of libngan_hip.so (C ABI: include/ngan.h).

There is no CPU fallback: if the shared library is missing, or a tensor is not a contiguous CUDA
tensor, the call raises.  (Tensors are fp32, or -- activations in the "bf16" mode -- bf16: ops.py picks the entry point by dtype.)  PyTorch is used only for device memory and the current HIP stream.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NGAN_LIB_PATH") or os.path.join(_HERE, "libngan_hip.so")      # (override: A/B runs of kernel variants)

_P, _I, _L, _F, _Z = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_size_t
CONV_SKIP_BORDER = 1     # NGAN_CONV_SKIP_BORDER (include/ngan.h): per-call flag of ngan_conv3x3_fwd / _fwd_ex

# name -> argument types (the trailing void* stream included), mirroring include/ngan.h
SIGNATURES = {
    "ngan_conv3x3_pack_weights": [_P, _P, _I, _I, _I, _F, _I, _P],
    "ngan_conv3x3_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _F, _I, _I, _P],
    "ngan_conv3x3_fwd_ex": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _F, _I, _I, _P],
    "ngan_conv3x3_wgrad": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _I, _I, _P],
    "ngan_lrelu_pixelnorm_fwd": [_P, _P, _P, _P, _L, _I, _F, _F, _P],
    "ngan_lrelu_pixelnorm_bwd": [_P, _P, _P, _P, _P, _L, _I, _F, _P],
    "ngan_lrelu_pixelnorm_bwd2": [_P, _P, _P, _P, _P, _P, _L, _I, _F, _P],
    "ngan_lrelu_pixelnorm_bwdbwd": [_P, _P, _P, _P, _P, _P, _P, _L, _I, _F, _P],
    "ngan_channel_sum": [_P, _P, _P, _L, _I, _F, _P],
    "ngan_channel_sum_acc": [_P, _P, _P, _L, _I, _F, _I, _P],
    "ngan_from_image_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "ngan_from_image_dx": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "ngan_from_image_dw": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "ngan_from_image_dw_acc": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "ngan_to_image_fwd": [_P, _P, _P, _L, _I, _I, _P],
    "ngan_to_image_bwd": [_P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P],
    "ngan_to_image_bwd_pnbwd": [_P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _F, _P],
    "ngan_to_image_bwd_pnbwd_acc": [_P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _F, _I, _P],
    "ngan_up2_fwd": [_P, _P, _I, _I, _I, _I, _P],
    "ngan_up2_adjoint": [_P, _P, _I, _I, _I, _I, _P],
    "ngan_up2_adjoint_pnbwd": [_P, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    "ngan_pool2_fwd": [_P, _P, _I, _I, _I, _I, _P],
    "ngan_pool2_adjoint": [_P, _P, _I, _I, _I, _I, _P],
    "ngan_lerp": [_P, _P, _P, _P, _L, _P],
    "ngan_axpby": [_P, _P, _F, _F, _P, _L, _P],
    "ngan_fade_bwd": [_P, _P, _P, _P, _L, _P],
    "ngan_xhat": [_P, _P, _P, _P, _I, _L, _P],
    "ngan_sample_l2norm": [_P, _P, _P, _I, _L, _P],
    "ngan_scale_rows": [_P, _P, _P, _I, _L, _P],
    "ngan_gp_head": [_P, _I, _F, _P, _P],
    "ngan_gp_coef": [_P, _I, _F, _P, _P, _P],
    "ngan_wloss_head": [_P, _I, _I, _F, _P, _P, _P, _P],
    "ngan_wloss_head_bwd": [_P, _I, _I, _F, _P, _P, _P, _P, _P],
    "ngan_latent_normalize": [_P, _I, _I, _F, _P],
    "ngan_linear_lrelu_pn_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _F, _P],
    "ngan_linear_wgrad": [_P, _P, _P, _I, _I, _I, _I, _F, _P],
    "ngan_linear_wgrad_acc": [_P, _P, _P, _I, _I, _I, _I, _F, _I, _P],
    "ngan_linear_wgrad_adam": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P],
    "ngan_linear_dgrad": [_P, _P, _P, _I, _I, _I, _I, _F, _P],
    "ngan_final_dot_fwd": [_P, _P, _P, _P, _I, _I, _I, _F, _P],
    "ngan_final_dot_dx": [_P, _P, _P, _I, _I, _I, _F, _P],
    "ngan_final_dot_dw": [_P, _P, _P, _P, _I, _I, _I, _F, _P],
    "ngan_final_dot_dw_acc": [_P, _P, _P, _P, _I, _I, _I, _F, _I, _P],
    "ngan_conv3x3_pack_many": [_P, _I, _L, _P],
    "ngan_conv3x3_wgrad_reduce_many": [_P, _I, _P],
    "ngan_adam_step": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _I, _P, _I, _P],
    "ngan_augment_batch": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "ngan_conv3x3_up2_border": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _P],
    "ngan_first_block_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _F, _P],
    "ngan_first_block_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P],
    "ngan_first_block_dx": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
}
# "bf16 activation storage" section of include/ngan.h: ngan_bf16_<op> has the argument list of ngan_<op> (the activation pointers are
# bf16 tensors); the two convolution entry points carry no precision / flags arguments
for _op in ("lrelu_pixelnorm_fwd", "lrelu_pixelnorm_bwd", "lrelu_pixelnorm_bwd2", "lrelu_pixelnorm_bwdbwd", "channel_sum", "channel_sum_acc",
            "from_image_fwd", "from_image_dx", "from_image_dw", "from_image_dw_acc", "to_image_fwd", "to_image_bwd", "to_image_bwd_pnbwd",
            "to_image_bwd_pnbwd_acc", "up2_fwd", "up2_adjoint", "up2_adjoint_pnbwd", "pool2_fwd", "pool2_adjoint", "lerp", "fade_bwd",
            "linear_lrelu_pn_fwd", "linear_wgrad", "linear_wgrad_acc", "linear_wgrad_adam", "linear_dgrad", "final_dot_fwd", "final_dot_dx",
            "final_dot_dw", "final_dot_dw_acc"):
    SIGNATURES["ngan_bf16_" + _op] = SIGNATURES["ngan_" + _op]
SIGNATURES["ngan_bf16_conv3x3_fwd"] = [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _F, _P]
SIGNATURES["ngan_bf16_conv3x3_wgrad"] = [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _I, _P]

NON_STATUS = {
    "ngan_version": ([], ctypes.c_char_p),
    "ngan_last_error": ([], ctypes.c_char_p),
    "ngan_conv3x3_wgrad_workspace_bytes": ([_I, _I, _I, _I, _I], _Z),
    "ngan_augment_workspace_bytes": ([_I, _I], _Z),
    "ngan_first_block_workspace_floats": ([_I, _I, _I], _Z),
    "ngan_first_block_table_floats": ([_I], _Z),
    "ngan_conv3x3_wgrad_kernel_name": ([_I, _I, _I, _I, _I, _I, _I, ctypes.c_char_p, _I], _I),
    "ngan_conv3x3_kernel_name": ([_I, _I, _I, _I, _I, _I, _I, _I, _I, ctypes.c_char_p, _I], _I),
    "ngan_conv3x3_algorithm": ([_I, _I, _I, _I, _I, _I, _I], _I),
    "ngan_conv3x3_epilogue_fused": ([_I, _I, _I, _I, _I, _I, _I, _I, _I], _I),
    "ngan_conv3x3_pooled_output": ([_I, _I, _I, _I, _I, _I, _I], _I),
    "ngan_conv3x3_packed_floats": ([_I, _I, _I], _L),
    "ngan_conv3x3_pack_elements": ([_I, _I, _I, _I], _L),
    "ngan_conv3x3_wgrad_plan": ([_I, _I, _I, _I, _I, _I, ctypes.POINTER(ctypes.c_int)], _I),
}

_lib = None


def lib():
    """Load libngan_hip.so once.  Raises RuntimeError (never falls back) if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                               f"or `make -C neuron-gan_amd/csrc`; there is no CPU fallback")
        handle = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
        for name, (argtypes, restype) in NON_STATUS.items():
            fn = getattr(handle, name)
            fn.argtypes = argtypes
            fn.restype = restype
        _lib = handle
    return _lib


def exported_symbols():
    return list(SIGNATURES) + list(NON_STATUS)


def version() -> str:
    return lib().ngan_version().decode()


def _ptr(t, name):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor or None, got {type(t)}")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: tensor is on {t.device}; the HIP path needs CUDA/HIP device memory (no CPU fallback)")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: tensor must be contiguous, got strides {t.stride()} for shape {tuple(t.shape)}")
    return t.data_ptr()


_probe = None


def set_probe(probe):
    """Install (or remove, with None) a launch probe: an object with `wants(name, args) -> bool` and
    `add(name, args, start_event, end_event)`.  bench.py uses it to time the dominant kernel with HIP events on
    the launch stream; the product path runs with no probe."""
    global _probe
    _probe = probe


def call(name, *args):
    """Invoke a status-returning entry point on PyTorch's current stream; tensors are passed by data pointer."""
    fn = getattr(lib(), name)
    conv = []
    for i, a in enumerate(args):
        conv.append(_ptr(a, f"{name} arg {i}") if (a is None or isinstance(a, torch.Tensor)) else a)
    stream = torch.cuda.current_stream().cuda_stream
    if _probe is not None and _probe.wants(name, args):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        status = fn(*conv, stream)
        e1.record()
        _probe.add(name, args, e0, e1)
    else:
        status = fn(*conv, stream)
    if status != 0:
        msg = lib().ngan_last_error().decode()
        raise RuntimeError(f"{name} failed with status {status}: {msg}")


def wgrad_workspace_bytes(B, H, W, Cin, Cout) -> int:
    return int(lib().ngan_conv3x3_wgrad_workspace_bytes(B, H, W, Cin, Cout))


def conv3x3_kernel_name(B, H, W, K, N, resample, epilogue, out_mode, precision=0) -> str:
    buf = ctypes.create_string_buffer(128)
    if lib().ngan_conv3x3_kernel_name(B, H, W, K, N, resample, epilogue, out_mode, precision, buf, 128) != 0:
        raise RuntimeError(lib().ngan_last_error().decode())
    return buf.value.decode()


def conv3x3_wgrad_kernel_name(B, H, W, Cin, Cout, resample, precision=0) -> str:
    buf = ctypes.create_string_buffer(128)
    if lib().ngan_conv3x3_wgrad_kernel_name(B, H, W, Cin, Cout, resample, precision, buf, 128) != 0:
        raise RuntimeError(lib().ngan_last_error().decode())
    return buf.value.decode()


def conv3x3_algorithm(B, H, W, K, N, resample, precision) -> int:
    return int(lib().ngan_conv3x3_algorithm(B, H, W, K, N, resample, precision))


def conv3x3_epilogue_fused(B, H, W, K, N, resample, epilogue, out_mode, precision) -> bool:
    return bool(lib().ngan_conv3x3_epilogue_fused(B, H, W, K, N, resample, epilogue, out_mode, precision))


def conv3x3_pooled_output(B, H, W, K, N, resample, precision) -> bool:
    return bool(lib().ngan_conv3x3_pooled_output(B, H, W, K, N, resample, precision))


def conv3x3_packed_floats(cout, cin, precision) -> int:
    return int(lib().ngan_conv3x3_packed_floats(cout, cin, precision))


def wgrad_plan(B, H, W, Cin, Cout, precision=0):
    out = (ctypes.c_int * 5)()
    if lib().ngan_conv3x3_wgrad_plan(B, H, W, Cin, Cout, precision, out) != 0:
        raise RuntimeError(lib().ngan_last_error().decode())
    return tuple(out)


def wgrad_reduce_many(raw_entries: bytes, n: int):
    """raw_entries: n packed 104-byte records (see include/ngan.h)"""
    buf = ctypes.create_string_buffer(raw_entries, len(raw_entries))
    stream = torch.cuda.current_stream().cuda_stream
    status = lib().ngan_conv3x3_wgrad_reduce_many(ctypes.cast(buf, ctypes.c_void_p), n, stream)
    if status != 0:
        raise RuntimeError(f"ngan_conv3x3_wgrad_reduce_many failed with status {status}: {lib().ngan_last_error().decode()}")
