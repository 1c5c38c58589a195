"""ctypes binding of libpcr_hip.so (include/pcr.h).  There is NO fallback: if the library is
missing or a call fails, a RuntimeError is raised."""
import ctypes
import os

import torch  # must be imported before the library so that libamdhip64.so.7 resolves to torch's copy

_HERE = os.path.dirname(os.path.abspath(__file__))
# (PCR_LIB_TAG: a diagnostic build made by pcr_amd/build.py under the same variable, e.g. libpcr_hip_tune.so)
_TAG = os.environ.get("PCR_LIB_TAG", "")
SO_PATH = os.path.join(_HERE, "lib", "libpcr_hip%s.so" % ("_" + _TAG if _TAG else ""))
_lib = None

ABI_VERSION = 17


class PcrError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise PcrError(
                "libpcr_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `python point-cloud-reid_amd/pcr_amd/build.py`; there is no CPU fallback." % SO_PATH)
        lib = ctypes.CDLL(SO_PATH)
        lib.pcr_status_string.restype = ctypes.c_char_p
        lib.pcr_packed_weight_floats.restype = ctypes.c_long
        lib.pcr_packed_weight_bf16_floats.restype = ctypes.c_long
        lib.pcr_attn_kv_floats.restype = ctypes.c_long
        lib.pcr_sa_tile_ws_ints.restype = ctypes.c_long
        lib.pcr_sa_claim_ws_ints.restype = ctypes.c_long
        lib.pcr_ball_query_rows_floats.restype = ctypes.c_long
        if lib.pcr_abi_version() != ABI_VERSION:
            raise PcrError("libpcr_hip.so ABI %d != binding %d: rebuild" % (lib.pcr_abi_version(), ABI_VERSION))
        if os.environ.get("PCR_STREAM_MIN_BLOCKS"):      # launch policy of the train-dense kernels (include/pcr.h)
            lib.pcr_set_stream_min_blocks(int(os.environ["PCR_STREAM_MIN_BLOCKS"]))
        _lib = lib
    return _lib


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())


def check(status, what):
    if status != 0:
        raise PcrError("%s failed: %s" % (what, load().pcr_status_string(status).decode()))


def require_cuda(*tensors):
    """device tensors only (no CPU fallback), all on the CURRENT device: the launch goes to the current device's
    stream (mmdet3d.ops switches to the tensor's device first, as the reference's KNN wrapper does)"""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise PcrError("pcr_amd ops run only on an MI355X device tensor (got %s); there is no CPU "
                           "fallback in the product path" % t.device)
        if t.device.index != torch.cuda.current_device():
            raise PcrError("tensor on %s but the current device is cuda:%d: call torch.cuda.set_device / "
                           "torch.cuda.device(...) first" % (t.device, torch.cuda.current_device()))


def require_f32(*tensors):
    for t in tensors:
        if t is not None and t.dtype != torch.float32:
            raise PcrError("expected a float32 tensor, got %s (the kernels read raw binary32)" % t.dtype)


def require_i32(*tensors):
    """the reference's wrappers read indices through data_ptr<int>() and raise on int64; so do we"""
    for t in tensors:
        if t is not None and t.dtype != torch.int32:
            raise PcrError("expected an int32 index tensor, got %s" % t.dtype)


def require_default_eps(*norms):
    """GroupNorm / LayerNorm kernels use eps = 1e-5 (every norm layer of the ReID configs): anything else is refused
    rather than silently computed with the wrong constant"""
    for n in norms:
        eps = getattr(n, "eps", 1e-5)
        if abs(eps - 1e-5) > 1e-12:
            raise PcrError("%s with eps=%g: the HIP kernels implement eps = 1e-5 only" % (type(n).__name__, eps))
