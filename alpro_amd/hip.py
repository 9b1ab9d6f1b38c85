"""ctypes binding of libalpro_hip.so (include/alpro_hip.h) + thin tensor-level wrappers.

There is NO CPU / eager fallback: if the library is missing or a kernel reports an error the
call raises.  torch is used for device memory and streams only.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ALPRO_HIP_LIB") or os.path.join(_HERE, "lib", "libalpro_hip.so")  # ALPRO_HIP_LIB: tools/ load the ablation build

F32, BF16, F16 = 0, 1, 2
ACT_NONE, ACT_GELU, ACT_RELU, ACT_GELU_BWD, ACT_GELU_SAVE_GRAD, ACT_MUL_SAVED = 0, 1, 2, 3, 4, 5
MAP_IDENTITY, MAP_SKIP_CLS, MAP_FRAME_TOKENS, MAP_PATCH_EMBED = 0, 1, 2, 3
ADD_IDENTITY, ADD_PRE_SPATIAL, ADD_PRE_MLP, ADD_PRE_TEMPORAL = 0, 1, 2, 3
EMIT_NONE, EMIT_ROWS, EMIT_FRAME, EMIT_SKIP_CLS = 0, 1, 2, 3

_TORCH_DTYPE = {F32: torch.float32, BF16: torch.bfloat16, F16: torch.float16}
_CODE = {v: k for k, v in _TORCH_DTYPE.items()}

EXPORTS = ["alpro_hip_last_error", "alpro_hip_abi_version", "alpro_hip_set_option", "alpro_hip_set_stream_option", "alpro_gemm", "alpro_gemm_c2_tiled_rows", "alpro_layernorm_fwd",
           "alpro_attn_temporal_fwd", "alpro_attn_fwd", "alpro_patchify", "alpro_cls_mean_residual",
           "alpro_vit_final_pool", "alpro_bert_embed_fwd", "alpro_cast_from_f32", "alpro_attn_bwd", "alpro_attn_temporal_bwd",
           "alpro_layernorm_bwd", "alpro_transpose", "alpro_transpose_batch", "alpro_gelu_bwd", "alpro_cls_mean_bwd", "alpro_scatter_add_rows", "alpro_gather_cast", "alpro_sumsq", "alpro_adamw_step", "alpro_gemm_tn_acc", "alpro_gemm_tn_acc_ws", "alpro_gemm_tn_workspace_bytes", "alpro_gemm_tn_ranges", "alpro_colsum_acc", "alpro_softmax_xent", "alpro_vtc_loss_fwd", "alpro_vtc_loss_bwd", "alpro_prepare_clips", "alpro_loss_scale_update", "alpro_add_layernorm_fwd", "alpro_layernorm_bwd_emit", "alpro_gemm_batch", "alpro_tproj_small", "alpro_attn_cls_fwd", "alpro_gemm_rows_f32", "alpro_gather_seq_fwd", "alpro_gather_seq_bwd", "alpro_scatter_add_rows_ordered",
           "alpro_hip_sched_workspace_bytes", "alpro_hip_set_sched_workspace", "alpro_hip_release_stream", "alpro_gemm_qkv_tattn", "alpro_add_layernorm_pre_mlp2", "alpro_adamw_step_lp"]


class GemmDesc(ctypes.Structure):
    _fields_ = [("A", ctypes.c_void_p), ("W", ctypes.c_void_p), ("C", ctypes.c_void_p),
                ("lda", ctypes.c_int64), ("ldw", ctypes.c_int64), ("ldc", ctypes.c_int64),
                ("M", ctypes.c_int), ("N", ctypes.c_int), ("K", ctypes.c_int),
                ("dtype", ctypes.c_int), ("c_dtype", ctypes.c_int), ("alpha", ctypes.c_float),
                ("bias", ctypes.c_void_p), ("act", ctypes.c_int),
                ("row_scale", ctypes.c_void_p), ("row_scale_group", ctypes.c_int),
                ("residual", ctypes.c_void_p), ("ldr", ctypes.c_int64),
                ("map_mode", ctypes.c_int), ("map_p0", ctypes.c_int), ("map_p1", ctypes.c_int),
                ("side", ctypes.c_void_p), ("ld_side", ctypes.c_int64),
                ("C2", ctypes.c_void_p), ("ldc2", ctypes.c_int64),
                ("drop_p", ctypes.c_float), ("drop_seed", ctypes.c_uint32), ("bias2", ctypes.c_void_p), ("m_off", ctypes.c_int64),
                ("c2_tiled", ctypes.c_int32), ("reserved0", ctypes.c_int32)]


class TprojJob(ctypes.Structure):
    _fields_ = [("wfc", ctypes.c_void_p), ("bp", ctypes.c_void_p), ("b1", ctypes.c_void_p), ("db1", ctypes.c_void_p), ("g_fc", ctypes.c_void_p),
                ("g_bp", ctypes.c_void_p)]   # 48 bytes


class TransposeJob(ctypes.Structure):
    _fields_ = [("in_", ctypes.c_void_p), ("out", ctypes.c_void_p), ("ld_in", ctypes.c_int64), ("ld_out", ctypes.c_int64),
                ("R", ctypes.c_int32), ("C", ctypes.c_int32), ("Rpad", ctypes.c_int32), ("tile0", ctypes.c_int32)]   # 48 bytes


ABI_VERSION = 20
_lib = None


def load():
    """Load libalpro_hip.so (built by `python -m alpro_amd.build`); raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    hq = os.environ.get("GPU_MAX_HW_QUEUES")
    if hq and hq.isdigit() and int(hq) > 4 and any(os.environ.get(k, "1") != "0" for k in ("ALPRO_WGRAD_STREAM", "ALPRO_TEXT_STREAM", "ALPRO_PROMPTER_STREAM")):
        import warnings
        warnings.warn("alpro_amd: GPU_MAX_HW_QUEUES=%s -- with more than 4 hardware queues the training step's side streams (ALPRO_WGRAD_STREAM / "
                      "ALPRO_TEXT_STREAM / ALPRO_PROMPTER_STREAM) measured 11 %% SLOWER than with the runtime's default of 4 (profiles/r6_hw_queues.txt): "
                      "a fifth queue shares a dispatch pipe with the launch stream's" % hq)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libalpro_hip.so not found at %s -- run `python -m alpro_amd.build` "
                           "(there is no CPU fallback for the ALPRO hot path)" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib.alpro_hip_last_error.restype = ctypes.c_char_p
    for name in EXPORTS:
        getattr(lib, name)  # AttributeError if the ABI is incomplete
    vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
    lib.alpro_gemm.argtypes = [ctypes.POINTER(GemmDesc), vp]
    lib.alpro_gemm_batch.argtypes = [ctypes.POINTER(GemmDesc), vp, ctypes.c_int, vp]
    lib.alpro_tproj_small.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
    lib.alpro_layernorm_fwd.argtypes = [vp, i64, vp, vp, f32, vp, i32, i64, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    lib.alpro_add_layernorm_fwd.argtypes = [vp, vp, i32, vp, i32, vp, vp, vp, f32, vp, vp, i64, i32, i32, i32, vp]
    lib.alpro_attn_temporal_fwd.argtypes = [vp, vp, i32, i64, i32, i32, f32, vp, vp]
    u32 = ctypes.c_uint32
    lib.alpro_attn_bwd.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp, f32, u32, vp]
    lib.alpro_attn_temporal_bwd.argtypes = [vp, vp, vp, vp, vp, i32, i64, i32, i32, f32, vp]
    lib.alpro_layernorm_bwd.argtypes = [vp, i32, i64, vp, vp, i64, vp, f32, vp, i64, i32, vp, vp, i32, i32, i32, i32, i32, f32, u32, vp, ctypes.c_size_t, vp]
    lib.alpro_layernorm_bwd_emit.argtypes = [vp, i32, i64, vp, vp, i64, vp, f32, vp, i64, i32, vp, vp, i32, i32, i32, i32, i32, f32, u32,
                                             vp, i32, i32, i32, i32, vp, i32, f32, u32, vp, i32, vp, ctypes.c_size_t, vp]
    lib.alpro_transpose.argtypes = [vp, i32, i64, vp, i32, i64, i32, i32, i32, vp, vp]
    lib.alpro_gelu_bwd.argtypes = [vp, vp, vp, i32, i64, vp]
    lib.alpro_sumsq.argtypes = [vp, i64, vp, vp, ctypes.c_size_t, vp]
    lib.alpro_softmax_xent.argtypes = [vp, i64, vp, i32, vp, vp, i32, i64, vp, i32, i32, i32, vp]
    lib.alpro_gemm_tn_acc.argtypes = [vp, i64, vp, i64, vp, i64, i32, i32, i32, i32, vp, vp]
    lib.alpro_gemm_tn_acc_ws.argtypes = [vp, i64, vp, i64, vp, i64, i32, i32, i32, i32, vp, vp, ctypes.c_size_t, vp]
    lib.alpro_gemm_tn_workspace_bytes.argtypes = [i32, i32, i32]
    lib.alpro_gemm_tn_workspace_bytes.restype = ctypes.c_size_t
    lib.alpro_gemm_tn_ranges.argtypes = [i32, i32, i32, i32]
    lib.alpro_colsum_acc.argtypes = [vp, i64, vp, i32, i32, i32, vp]
    lib.alpro_transpose_batch.argtypes = [vp, i32, i32, i32, vp]
    lib.alpro_adamw_step.argtypes = [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, f32, vp, f32, f32, vp, i32, i32, i32, vp]
    lib.alpro_adamw_step_lp.argtypes = [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, f32, vp, f32, f32, vp, i32, i32, i32, vp, i32, vp]
    lib.alpro_loss_scale_update.argtypes = [vp, vp, f32, f32, i32, f32, f32, vp]
    lib.alpro_gather_cast.argtypes = [vp, i64, vp, i32, i32, i32, i32, i32, i32, vp, i32, f32, f32, u32, vp, vp, vp, ctypes.c_size_t, vp]
    lib.alpro_cls_mean_bwd.argtypes = [vp, i64, vp, i32, i32, i32, vp]
    lib.alpro_scatter_add_rows.argtypes = [vp, vp, vp, i32, i32, i32, i64, vp]
    lib.alpro_attn_fwd.argtypes = [vp, vp, i32, i32, i32, i32, f32, vp, vp, f32, u32, vp, i32, vp, vp]
    lib.alpro_attn_cls_fwd.argtypes = [vp, i32, vp, vp, vp, i32, i32, i32, i32, f32, f32, u32, vp]
    lib.alpro_gemm_rows_f32.argtypes = [vp, i64, vp, i64, vp, i64, i32, i32, i32, vp, i32, vp, vp, i64, vp, vp, f32, vp]
    lib.alpro_gemm_c2_tiled_rows.argtypes = [i64, i64, i64, i32]
    lib.alpro_gemm_c2_tiled_rows.restype = i64
    lib.alpro_patchify.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp]
    lib.alpro_cls_mean_residual.argtypes = [vp, i64, vp, vp, i64, i32, i32, i32, vp]
    lib.alpro_vit_final_pool.argtypes = [vp, vp, vp, f32, vp, vp, i32, i32, i32, i32, i32, vp]
    lib.alpro_add_layernorm_pre_mlp2.argtypes = [vp, vp, vp, vp, i32, vp, vp, vp, f32, vp, i64, i32, i32, i32, vp]
    lib.alpro_bert_embed_fwd.argtypes = [vp, vp, vp, vp, vp, vp, f32, vp, vp, i32, vp, vp, i32, i32, i32, f32, u32, vp]
    lib.alpro_cast_from_f32.argtypes = [vp, vp, i32, i64, vp]
    lib.alpro_hip_set_option.argtypes = [ctypes.c_char_p, i32]
    lib.alpro_hip_set_stream_option.argtypes = [vp, ctypes.c_char_p, i32]
    lib.alpro_hip_sched_workspace_bytes.restype = ctypes.c_size_t
    lib.alpro_hip_set_sched_workspace.argtypes = [vp, vp, ctypes.c_size_t]
    lib.alpro_hip_release_stream.argtypes = [vp]
    lib.alpro_gemm_qkv_tattn.argtypes = [vp, i64, vp, i64, vp, vp, i64, i32, i32, i32, i32, i32, f32, vp, i64, vp, vp]
    lib.alpro_scatter_add_rows_ordered.argtypes = [vp, vp, vp, i32, i32, i64, i64, vp, vp]
    lib.alpro_gather_seq_fwd.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    lib.alpro_gather_seq_bwd.argtypes = [vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.alpro_prepare_clips.argtypes = [vp, i32, vp, f32, ctypes.POINTER(f32), ctypes.POINTER(f32), vp, vp, vp, i32, i32, i32, i32, vp]
    lib.alpro_vtc_loss_fwd.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp]
    lib.alpro_vtc_loss_bwd.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    if lib.alpro_hip_abi_version() != ABI_VERSION:
        raise RuntimeError("libalpro_hip.so ABI version mismatch")
    _lib = lib
    return lib


_OPTION_DEFAULTS = {"gemm_tile": 0, "gemm_grid": 0, "gemm_tune": 1, "tn_splits": 0, "tn_kind": 2, "gemm_tail": 1, "attn_bwd": 1, "attn_order": 0, "gemm_kind": 1, "cu_budget": 0, "ln_grid": 0, "gemm_sched": 1, "gemm_epi": 1}
_option_values = {}


def set_option(name, value):
    """Measurement knob of the library (alpro_hip_set_option): 'attn_bwd' (16-bit attention backward with 5-8 key tiles: 0 two-phase,
    1 = default, best per shape, 2 key-owned; 3 / 4, the persistent key-owned variants, only in the --ablations build), 'gemm_tile', 'gemm_grid',
    'gemm_tune', 'gemm_kind' (identity-map 16-bit shapes: 0 = round-3 persistent kernel, 1 = 8-phase two-group kernel, the default), 'tn_splits'
    (token ranges of the weight-gradient GEMM), 'tn_kind' (0, 2 = two-group schedule; 1 = no wgrad epilogue, --ablations build only), 'cu_budget'
    (CUs the persistent grids are sized for; 0 = all), 'ln_grid' (cap on the LayerNorm backward's workgroup count; 0 = the default plan),
    'gemm_sched' (8-phase GEMM: 1 = tiles from per-XCD ticket counters, the default; 0 = the static round-robin walk; results are bitwise equal)."""
    _check(load().alpro_hip_set_option(name.encode(), int(value)), "alpro_hip_set_option")
    _option_values[name] = int(value)


def set_stream_option(name, value, stream=None):
    """alpro_hip_set_stream_option: the knob `name` for launches on ONE stream (default: torch's current stream; pass a raw handle to address a
    stream from another thread); value < 0 removes the override.  The library consults per-stream values for 'cu_budget' -- the situation a
    launch runs in (a collective's kernels holding CUs next to THIS stream's work), not a property of the process.  Returns the stream handle."""
    h = stream if stream is not None else (_stream() if torch.cuda.is_available() else ctypes.c_void_p(0))   # (CPU-only boxes: gloo tests of the exchange logic)
    _check(load().alpro_hip_set_stream_option(h, name.encode(), int(value)), "alpro_hip_set_stream_option")
    return h


def _stream_handle(stream):
    if stream is None:
        return _stream()
    return ctypes.c_void_p(stream.cuda_stream) if isinstance(stream, torch.cuda.Stream) else stream


_sched_ws = {}   # (device index, raw stream handle) -> the tensor that backs the stream's tile-scheduler blocks (kept alive here)


def set_sched_workspace(stream=None, tensor=None):
    """alpro_hip_set_sched_workspace: give the persistent GEMM's tile scheduler on `stream` (a torch.cuda.Stream, a raw handle, default: the
    current stream) its two counter blocks from CALLER memory -- with that the library allocates nothing, ever (otherwise the first persistent
    launch on a device makes one 135 KB hipMalloc for all its streams; include/alpro_hip.h).  `tensor`: any CUDA tensor of at least
    sched_workspace_bytes() bytes; default: a fresh one, kept alive here until release_stream().  The library clears it on the stream."""
    lib = load()
    h = _stream_handle(stream)
    need = int(lib.alpro_hip_sched_workspace_bytes())
    if tensor is None:
        tensor = torch.empty(need, dtype=torch.uint8, device="cuda")
    if not tensor.is_cuda or tensor.numel() * tensor.element_size() < need or not tensor.is_contiguous():
        raise ValueError("set_sched_workspace: a contiguous CUDA tensor of >= %d bytes" % need)
    _check(lib.alpro_hip_set_sched_workspace(h, ctypes.c_void_p(tensor.data_ptr()), tensor.numel() * tensor.element_size()), "alpro_hip_set_sched_workspace")
    _sched_ws[(tensor.device.index, h.value)] = tensor
    return tensor


def release_stream(stream=None):
    """alpro_hip_release_stream: forget everything the library keeps for `stream` on the current device -- its tile-scheduler slot (64 per
    process) and its option overrides.  Call it for a stream that is idle and about to be destroyed: the runtime may hand the same handle
    value to a later stream, which must not inherit the slot."""
    h = _stream_handle(stream)
    _check(load().alpro_hip_release_stream(h), "alpro_hip_release_stream")
    _sched_ws.pop((torch.cuda.current_device() if torch.cuda.is_available() else 0, h.value), None)


def get_option(name):
    """The value last set through set_option (the library's default otherwise)."""
    return _option_values.get(name, _OPTION_DEFAULTS[name])


class option:
    """`with hip.option('gemm_tile', 256): ...` -- set a knob for a block and put back what was set before (the value of the last
    set_option call, else the built-in default; an initial value taken from the environment is not visible here)."""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.prev = _option_values.get(self.name, _OPTION_DEFAULTS[self.name])
        set_option(self.name, self.value)

    def __exit__(self, *a):
        set_option(self.name, self.prev)


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, load().alpro_hip_last_error().decode()))


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _dev(t, dtype=None):
    if not t.is_cuda:
        raise RuntimeError("alpro_amd ops need device tensors (got %s): the hot path has no CPU fallback" % t.device)
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError("expected %s, got %s" % (dtype, t.dtype))
    if not (t.is_contiguous() or (t.dim() == 2 and t.stride(1) == 1)):
        raise RuntimeError("expected a contiguous (or row-strided 2-D) tensor")
    return t


def dtype_code(torch_dtype):
    return _CODE[torch_dtype]


def torch_dtype(code):
    return _TORCH_DTYPE[code]


# ------------------------------------------------------------------------------------------------
def gemm(a, w, out=None, bias=None, act=ACT_NONE, out_dtype=None, alpha=1.0, row_scale=None, row_scale_group=1,
         residual=None, map_mode=MAP_IDENTITY, map_p0=0, map_p1=0, side=None, out_rows=None, pre_act=None, drop_p=0.0, drop_seed=0, bias2=None,
         c2_tiled=False, _desc_only=False):
    """out[map(m)] = residual[map(m)] + row_scale * act(alpha * a @ w.T + bias) [+ bias2]   (see alpro_gemm).
    act=ACT_GELU_BWD: out = (alpha * a @ w.T + bias) * gelu'(pre_act) (pre_act read only).
    act=ACT_GELU_SAVE_GRAD: out = gelu(..), pre_act (required) RECEIVES gelu'(..); act=ACT_MUL_SAVED: out = (..) * pre_act (read only).
    c2_tiled: pre_act is a buffer of gemm_c2_tiled_rows(M, N, K, dtype) > 0 rows in the library's tile layout (written by GELU_SAVE_GRAD, read
    by MUL_SAVED of the same (M, N); nothing else may interpret it)."""
    lib = load()
    _dev(a); _dev(w, a.dtype)
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K
    out_dtype = out_dtype or a.dtype
    if out is None:
        out = torch.empty((out_rows if out_rows is not None else M, N), dtype=out_dtype, device=a.device)
    _dev(out, out_dtype)
    d = GemmDesc()
    d.A, d.W, d.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
    d.lda, d.ldw, d.ldc = a.stride(0), w.stride(0), out.shape[-1]
    d.M, d.N, d.K = M, N, K
    d.dtype, d.c_dtype, d.alpha = _CODE[a.dtype], _CODE[out_dtype], alpha
    d.bias = _dev(bias, torch.float32).data_ptr() if bias is not None else None
    d.act = act
    d.row_scale = _dev(row_scale, torch.float32).data_ptr() if row_scale is not None else None
    d.row_scale_group = row_scale_group
    d.residual = _dev(residual, torch.float32).data_ptr() if residual is not None else None
    d.ldr = residual.shape[-1] if residual is not None else 0
    d.map_mode, d.map_p0, d.map_p1 = map_mode, map_p0, map_p1
    d.side = _dev(side, torch.float32).data_ptr() if side is not None else None
    d.ld_side = side.shape[-1] if side is not None else 0
    d.C2 = _dev(pre_act, a.dtype).data_ptr() if pre_act is not None else None
    d.ldc2 = pre_act.shape[-1] if pre_act is not None else 0
    d.c2_tiled = 1 if c2_tiled else 0
    if c2_tiled and not (pre_act is not None and pre_act.is_contiguous() and pre_act.numel() >= gemm_c2_tiled_rows(M, N, K, a.dtype) * N > 0):
        raise RuntimeError("gemm: c2_tiled needs a contiguous pre_act buffer of gemm_c2_tiled_rows(M, N, K, dtype) x N elements (0 rows = not offered for this shape)")
    d.drop_p, d.drop_seed = drop_p, drop_seed
    d.bias2 = _dev(bias2, torch.float32).data_ptr() if bias2 is not None else None
    if _desc_only:
        return d, out
    _check(lib.alpro_gemm(ctypes.byref(d), _stream()), "alpro_gemm")
    return out


def gemm_c2_tiled_rows(M, N, K, dtype):
    """Rows of the (rows, N) buffer a tile-layout pre_act needs for this GEMM shape, or 0 when the shape does not run on the kernel that has
    the tile layout under the current options (alpro_gemm_c2_tiled_rows): then use the row layout."""
    return int(load().alpro_gemm_c2_tiled_rows(M, N, K, _CODE[dtype])) if dtype in (torch.float16, torch.bfloat16) else 0


class GemmBatch:
    """A fixed list of independent plain GEMMs (alpro_gemm_batch): descriptors are built once from the tensors given to add() -- whose storage
    must stay where it is (parameters, flat gradient views, persistent operand buffers) -- and launched together as often as wanted."""

    def __init__(self):
        self._descs, self._keep = [], []
        self._host = self._dev = None

    def add(self, a, w, out, bias=None, alpha=1.0, residual=None, out_dtype=None):
        d, _ = gemm(a, w, out=out, bias=bias, alpha=alpha, residual=residual, out_dtype=out_dtype or out.dtype, _desc_only=True)
        self._descs.append(d)
        self._keep.append((a, w, out, bias, residual))
        self._host = None

    def signature(self):
        return tuple((d.A, d.W, d.C, d.residual) for d in self._descs)

    def launch(self):
        if self._host is None:
            self._host = (GemmDesc * len(self._descs))(*self._descs)
            self._dev = torch.frombuffer(bytearray(bytes(self._host)), dtype=torch.uint8).to(self._keep[0][0].device)
        _check(load().alpro_gemm_batch(self._host, _ptr(self._dev), len(self._descs), _stream()), "alpro_gemm_batch")


def tproj_jobs(jobs, device):
    """[dict(wfc=, bp=, b1=, db1=, g_fc=, g_bp=)] (fp32 device tensors or None) -> device job table for tproj_small."""
    arr = (TprojJob * len(jobs))()
    for j, d in zip(arr, jobs):
        for k in ("wfc", "bp", "b1", "db1", "g_fc", "g_bp"):
            t = d.get(k)
            if t is not None:
                _dev(t, torch.float32)
                assert t.is_contiguous()
            setattr(j, k, t.data_ptr() if t is not None else None)
    return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device), len(jobs)


def tproj_small(table, njobs, D, mode):
    _check(load().alpro_tproj_small(_ptr(table), njobs, D, mode, _stream()), "alpro_tproj_small")


def layernorm(x, gamma, beta, eps, out_dtype, rows=None, out32=False, stats=False, map_mode=MAP_IDENTITY, map_p0=0, map_p1=0):
    """x: fp32 (..., 768) token tensor; returns y (rows, 768) [, y32] [, mean, rstd]."""
    lib = load()
    _dev(x, torch.float32)
    D = x.shape[-1]
    rows = rows if rows is not None else x.numel() // D
    y = torch.empty((rows, D), dtype=out_dtype, device=x.device)
    y32 = torch.empty((rows, D), dtype=torch.float32, device=x.device) if out32 else None
    mean = torch.empty(rows, dtype=torch.float32, device=x.device) if stats else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if stats else None
    _check(lib.alpro_layernorm_fwd(_ptr(x), D, _ptr(_dev(gamma, torch.float32)), _ptr(_dev(beta, torch.float32)), eps, _ptr(y),
                                   _CODE[out_dtype], D, _ptr(y32), _ptr(mean), _ptr(rstd), rows, D, map_mode, map_p0, map_p1,
                                   _stream()), "alpro_layernorm_fwd")
    res = (y,)
    if out32:
        res += (y32,)
    if stats:
        res += (mean, rstd)
    return res if len(res) > 1 else y


def add_layernorm(x_in, delta, gamma, beta, eps, mode=ADD_IDENTITY, x_out=None, want_x=True, delta_bias=None, T=0, N=0, out32=False):
    """x' = x_in + gathered delta (+ delta_bias); returns (y [, y32], x') with y = LayerNorm(x') in delta's dtype -- see alpro_add_layernorm_fwd.
    x_in fp32 token tensor (..., 768); delta (rows, 768) in the operand dtype.  x_out: tensor to receive x' (may be x_in itself), or None
    with want_x=True to allocate it, or want_x=False when nobody needs x'."""
    lib = load()
    _dev(x_in, torch.float32); _dev(delta)
    D = x_in.shape[-1]
    tokens = x_in.numel() // D
    dt = delta.dtype
    if mode == ADD_IDENTITY:
        rows = yrows = want = tokens
    else:
        if T <= 0 or N <= 0 or tokens % (1 + N * T) != 0:
            raise RuntimeError("add_layernorm: %d token rows are not a whole number of clips of 1 + %d x %d tokens" % (tokens, N, T))
        B = tokens // (1 + N * T)
        rows, yrows, want = {ADD_PRE_SPATIAL: (B * T * (N + 1), B * T * (N + 1), B * N * T), ADD_PRE_MLP: (tokens, tokens, B * T * (N + 1)),
                             ADD_PRE_TEMPORAL: (tokens, B * N * T, tokens)}[mode]
    if delta.shape[0] != want or delta.shape[-1] != D:
        raise RuntimeError("add_layernorm: delta has %s rows, mode %d over %d token rows needs (%d, %d)" % (tuple(delta.shape), mode, tokens, want, D))
    if x_out is None and want_x:
        x_out = torch.empty_like(x_in)
    if x_out is not None:
        _dev(x_out, torch.float32)
    y = torch.empty((yrows, D), dtype=dt, device=x_in.device)
    y32 = torch.empty((yrows, D), dtype=torch.float32, device=x_in.device) if (out32 and dt != torch.float32) else None
    _check(lib.alpro_add_layernorm_fwd(_ptr(x_in), _ptr(delta), _CODE[dt], _ptr(_dev(delta_bias, torch.float32)) if delta_bias is not None else None, mode,
                                       _ptr(x_out), _ptr(_dev(gamma, torch.float32)), _ptr(_dev(beta, torch.float32)), eps, _ptr(y), _ptr(y32), rows, D, T, N,
                                       _stream()), "alpro_add_layernorm_fwd")
    if out32:
        return y, (y32 if y32 is not None else y), x_out
    return y, x_out


def add_layernorm_pre_mlp2(x_in, delta_t, delta_t_bias, delta_s, gamma, beta, eps, T, N, x_out=None):
    """ADD_PRE_MLP with the temporal branch's deferred add in front (alpro_add_layernorm_pre_mlp2): x' = x_in + delta_t (x[:, 1:] order) + delta_t_bias +
    delta_s (frame order; CLS rows: the frame mean of delta_s only) -> x_out (default: x_in itself), returns LayerNorm(x') in the deltas' dtype."""
    lib = load()
    _dev(x_in, torch.float32); _dev(delta_t); _dev(delta_s)
    D = x_in.shape[-1]
    tokens = x_in.numel() // D
    if T <= 0 or N <= 0 or tokens % (1 + N * T) != 0:
        raise RuntimeError("add_layernorm_pre_mlp2: %d token rows are not a whole number of clips of 1 + %d x %d tokens" % (tokens, N, T))
    B = tokens // (1 + N * T)
    if tuple(delta_t.shape) != (B * N * T, D) or tuple(delta_s.shape) != (B * T * (N + 1), D) or delta_t.dtype != delta_s.dtype:
        raise RuntimeError("add_layernorm_pre_mlp2: deltas %s / %s do not fit %d clips" % (tuple(delta_t.shape), tuple(delta_s.shape), B))
    x_out = x_in if x_out is None else _dev(x_out, torch.float32)
    y = torch.empty((tokens, D), dtype=delta_s.dtype, device=x_in.device)
    _check(lib.alpro_add_layernorm_pre_mlp2(_ptr(x_in), _ptr(delta_t), _ptr(_dev(delta_t_bias, torch.float32)) if delta_t_bias is not None else None, _ptr(delta_s),
                                            _CODE[delta_s.dtype], _ptr(x_out), _ptr(_dev(gamma, torch.float32)), _ptr(_dev(beta, torch.float32)), eps, _ptr(y),
                                            tokens, D, T, N, _stream()), "alpro_add_layernorm_pre_mlp2")
    return y


def attn_temporal(qkv, T, H, scale, want_lse=False):
    lib = load()
    _dev(qkv)
    rows = qkv.shape[0]
    out = torch.empty((rows, H * 64), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(((rows + 31) // 32, H, 32), dtype=torch.float32, device=qkv.device) if want_lse else None
    _check(lib.alpro_attn_temporal_fwd(_ptr(qkv), _ptr(out), _CODE[qkv.dtype], rows, T, H, scale, _ptr(lse), _stream()), "alpro_attn_temporal_fwd")
    return (out, lse) if want_lse else out


def attn_temporal_bwd(qkv, out, dout, lse, T, H, scale):
    lib = load()
    _dev(qkv); _dev(out, qkv.dtype); _dev(dout, qkv.dtype); _dev(lse, torch.float32)
    dqkv = torch.empty_like(qkv)
    _check(lib.alpro_attn_temporal_bwd(_ptr(qkv), _ptr(out), _ptr(dout), _ptr(lse), _ptr(dqkv), _CODE[qkv.dtype], qkv.shape[0], T, H, scale,
                                       _stream()), "alpro_attn_temporal_bwd")
    return dqkv


def attn_bwd(qkv, out, dout, lse, batch, L, H, scale, key_bias=None, drop_p=0.0, drop_seed=0):
    lib = load()
    _dev(qkv); _dev(out, qkv.dtype); _dev(dout, qkv.dtype); _dev(lse, torch.float32)
    dqkv = torch.empty_like(qkv)
    kb = _dev(key_bias, torch.float32) if key_bias is not None else None
    _check(lib.alpro_attn_bwd(_ptr(qkv), _ptr(out), _ptr(dout), _ptr(lse), _ptr(dqkv), _CODE[qkv.dtype], batch, L, H, scale, _ptr(kb), drop_p, drop_seed, _stream()),
           "alpro_attn_bwd")
    return dqkv


def layernorm_bwd(dy, x, gamma, eps, dx, dgamma, dbeta, rows=None, dy2=None, accumulate=True, map_mode=MAP_IDENTITY, map_p0=0, map_p1=0,
                  drop_p=0.0, drop_seed=0, emit=None):
    """dx[map(m)] (+)= dLN; dgamma/dbeta (fp32, pre-initialised) are accumulated.  x, dx: fp32 (..., 768).
    emit: None, or a dict(mode=EMIT_*, rows=<output rows>, dtype=<operand dtype, default dy's>, T=, N=, scale=<fp32 row scales>, group=, drop_p=, drop_seed=, colsum_pre=,
    extra_cls=) -- the finished gradient rows ALSO leave as the dy-dtype operand rows of the next GEMMs (alpro_layernorm_bwd_emit, what a
    following gather_cast would build); then returns (dx, emitted (rows, 768) tensor)."""
    lib = load()
    _dev(dy); _dev(x, torch.float32); _dev(dx, torch.float32); _dev(dgamma, torch.float32); _dev(dbeta, torch.float32)
    D = x.shape[-1]
    rows = rows if rows is not None else dy.numel() // D
    common = (_ptr(dy), _CODE[dy.dtype], D, _ptr(_dev(dy2, torch.float32)) if dy2 is not None else None, _ptr(x), D,
              _ptr(_dev(gamma, torch.float32)), eps, _ptr(dx), D, 1 if accumulate else 0, _ptr(dgamma), _ptr(dbeta), rows, D,
              map_mode, map_p0, map_p1, drop_p, drop_seed)
    ws, wsb = _reduce_ws(dy.device)
    if emit is None:
        _check(lib.alpro_layernorm_bwd(*common, ws, wsb, _stream()), "alpro_layernorm_bwd")
        return dx
    out = torch.empty((emit["rows"], D), dtype=emit.get("dtype", dy.dtype), device=dy.device)
    sc, cp = emit.get("scale"), emit.get("colsum_pre")
    _check(lib.alpro_layernorm_bwd_emit(*common, _ptr(out), _CODE[out.dtype], emit["mode"], emit.get("T", 0), emit.get("N", 0),
                                        _ptr(_dev(sc, torch.float32)) if sc is not None else None, emit.get("group", 1), emit.get("drop_p", 0.0),
                                        emit.get("drop_seed", 0), _ptr(_dev(cp, torch.float32)) if cp is not None else None, emit.get("extra_cls", 0),
                                        ws, wsb, _stream()), "alpro_layernorm_bwd_emit")
    return dx, out


def transpose(x, out_dtype=None, pad_to=64, colsum=None):
    """(R, C) -> (C, Rpad) with the R dimension zero-padded to a multiple of `pad_to`; optional fp32 column sums (+=)."""
    lib = load()
    _dev(x)
    R, C = x.shape
    out_dtype = out_dtype or x.dtype
    Rpad = (R + pad_to - 1) // pad_to * pad_to
    out = torch.empty((C, Rpad), dtype=out_dtype, device=x.device)
    if colsum is not None and _DETERMINISTIC[0]:   # the fused column sums are one fp32 atomic per 64-row tile (order varies); fp32 exact mode only
        _dev(colsum, torch.float32)[:C].add_(x.sum(0, dtype=torch.float32))
        colsum = None
    _check(lib.alpro_transpose(_ptr(x), _CODE[x.dtype], x.stride(0), _ptr(out), _CODE[out_dtype], Rpad, R, C, Rpad,
                               _ptr(_dev(colsum, torch.float32)) if colsum is not None else None, _stream()), "alpro_transpose")
    return out


def gather_cast(src, dtype, rows=None, map_mode=MAP_IDENTITY, map_p0=0, map_p1=0, row_scale=None, row_scale_group=1, cls_scale=1.0,
                drop_p=0.0, drop_seed=0, colsum=None, colsum_pre=None):
    """fp32 (..., 768) token-gradient rows -> (rows, 768) GEMM operand in `dtype` (see alpro_gather_cast); colsum (768,) fp32 += out.sum(0)."""
    lib = load()
    _dev(src, torch.float32)
    D = src.shape[-1]
    rows = rows if rows is not None else src.numel() // D
    out = torch.empty((rows, D), dtype=dtype, device=src.device)
    ws, wsb = _reduce_ws(src.device) if (colsum is not None or colsum_pre is not None) else (None, 0)
    _check(lib.alpro_gather_cast(_ptr(src), D, _ptr(out), _CODE[dtype], rows, D, map_mode, map_p0, map_p1,
                                 _ptr(_dev(row_scale, torch.float32)) if row_scale is not None else None, row_scale_group, cls_scale, drop_p, drop_seed,
                                 _ptr(_dev(colsum, torch.float32)) if colsum is not None else None,
                                 _ptr(_dev(colsum_pre, torch.float32)) if colsum_pre is not None else None, ws, wsb, _stream()),
           "alpro_gather_cast")
    return out


def gelu_bwd(dh, u):
    lib = load()
    _dev(dh); _dev(u, dh.dtype)
    du = torch.empty_like(dh)
    _check(lib.alpro_gelu_bwd(_ptr(dh), _ptr(u), _ptr(du), _CODE[dh.dtype], dh.numel(), _stream()), "alpro_gelu_bwd")
    return du


def cls_mean_bwd(dx_out, B, T):
    lib = load()
    _dev(dx_out, torch.float32)
    D = dx_out.shape[-1]
    dside = torch.empty((B * T, D), dtype=torch.float32, device=dx_out.device)
    _check(lib.alpro_cls_mean_bwd(_ptr(dx_out), dx_out.stride(0), _ptr(dside), B, T, D, _stream()), "alpro_cls_mean_bwd")
    return dside


_SCATTER_KEYS = {}   # device -> 8192-entry key workspace of alpro_scatter_add_rows_ordered (used in stream order)


def scatter_add_rows(src, idx, dst, idx_mod=0, skip_idx=-1):
    """dst[idx[i]] += src[i] (idx None: row i % idx_mod); rows whose index is skip_idx contribute nothing (nn.Embedding's padding_idx).
    Reproducible mode (the default, see set_deterministic): the position form runs the one-writer-per-row kernel; the indexed form goes
    through a sort (torch's index_put_ with accumulate: duplicates are added in a fixed order) after the skipped rows have been zeroed --
    the atomic kernel's order of additions to a duplicated row (every [CLS], [SEP], [MASK] ...) varies run to run."""
    lib = load()
    _dev(src, torch.float32); _dev(dst, torch.float32)
    if idx is not None and _DETERMINISTIC[0]:
        idx = _dev(idx, torch.int64).view(-1)
        if src.shape[0] <= 8192 and dst.shape[0] < (1 << 19) and src.is_contiguous() and dst.is_contiguous():
            # round 5: the library's own ordered scatter (keys sorted in LDS by one workgroup, one writer per destination row, ascending source order)
            ws = _SCATTER_KEYS.get(src.device)
            if ws is None:
                ws = _SCATTER_KEYS[src.device] = torch.empty(8192, dtype=torch.int32, device=src.device)
            _check(lib.alpro_scatter_add_rows_ordered(_ptr(src), _ptr(idx), _ptr(dst), src.shape[0], src.shape[1], dst.shape[0], int(skip_idx), _ptr(ws), _stream()),
                   "alpro_scatter_add_rows_ordered")
            return dst
        rows = src if skip_idx < 0 else src * (idx != skip_idx).unsqueeze(1).to(src.dtype)
        dst.index_put_((idx,), rows, accumulate=True)
        return dst
    _check(lib.alpro_scatter_add_rows(_ptr(src), _ptr(_dev(idx, torch.int64)) if idx is not None else None, _ptr(dst), src.shape[0], idx_mod,
                                      src.shape[1], int(skip_idx), _stream()), "alpro_scatter_add_rows")
    return dst


def attn(qkv, batch, L, H, scale, key_bias=None, want_lse=False, drop_p=0.0, drop_seed=0, cls_q=None, cls_group=1, cls_out=None):
    """-> out [, lse] [, cls_out].  cls_q (batch / cls_group, 3*H*64) fp32: also evaluate the CLS query (row 0 of every sequence) in fp32 from
    these unrounded q rows against the K / V staged in LDS -> cls_out (batch, H*64) fp32 (precise CLS rows, 16-bit dtypes only)."""
    lib = load()
    _dev(qkv)
    assert qkv.shape[0] == batch * L
    out = torch.empty((batch * L, H * 64), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((batch, H, L), dtype=torch.float32, device=qkv.device) if want_lse else None
    kb = _dev(key_bias, torch.float32) if key_bias is not None else None
    if cls_q is None:
        cls_out = None
    else:
        _dev(cls_q, torch.float32)
        if cls_q.shape != (batch // cls_group, 3 * H * 64) or batch % cls_group or not cls_q.is_contiguous():
            raise RuntimeError("attn: cls_q %s does not match batch=%d H=%d cls_group=%d" % (tuple(cls_q.shape), batch, H, cls_group))
        if cls_out is None:
            cls_out = torch.empty((batch, H * 64), dtype=torch.float32, device=qkv.device)
        assert cls_out.shape == (batch, H * 64) and cls_out.is_contiguous()
        _dev(cls_out, torch.float32)
    _check(lib.alpro_attn_fwd(_ptr(qkv), _ptr(out), _CODE[qkv.dtype], batch, L, H, scale, _ptr(kb), _ptr(lse), drop_p, drop_seed,
                              _ptr(cls_q), cls_group, _ptr(cls_out), _stream()), "alpro_attn_fwd")
    res = (out,) + ((lse,) if want_lse else ()) + ((cls_out,) if cls_q is not None else ())
    return res if len(res) > 1 else out


def qkv_tattn_ok(a, T):
    """Does alpro_gemm_qkv_tattn take these rows?  16-bit operands, whole 32-token wave blocks, a frame count that divides 16."""
    return a.dtype in (torch.float16, torch.bfloat16) and a.shape[0] % 32 == 0 and T in (1, 2, 4, 8, 16) and a.shape[1] % 128 == 0 and a.stride(1) == 1 \
        and (a.stride(0) * 2) % 128 == 0


def gemm_qkv_tattn(a, w, bias, T, H, scale, out=None, want_qkv=False):
    """The temporal half's qkv Linear + frame attention in one launch (alpro_gemm_qkv_tattn): a (M, K) 16-bit rows in x[:, 1:] order, w (3*H*64, K)
    = Attention.qkv.weight in the operand dtype, bias (3*H*64) fp32 -> (M, H*64) attention output, heads merged.  want_qkv (training): also
    returns q | k | v (M, 3*H*64) as alpro_gemm would have stored them and the log-sum-exp rows alpro_attn_temporal_bwd needs:
    -> (out, qkv, lse)."""
    lib = load()
    _dev(a); _dev(w)
    M, K = a.shape
    assert w.shape == (3 * H * 64, K) and w.dtype == a.dtype and w.stride(1) == 1 and a.stride(1) == 1
    b = _dev(bias, torch.float32) if bias is not None else None
    if out is None:
        out = torch.empty((M, H * 64), dtype=a.dtype, device=a.device)
    assert out.shape == (M, H * 64) and out.stride(1) == 1 and out.dtype == a.dtype
    qkv = torch.empty((M, 3 * H * 64), dtype=a.dtype, device=a.device) if want_qkv else None
    lse = torch.empty(((M + 31) // 32, H, 32), dtype=torch.float32, device=a.device) if want_qkv else None
    _check(lib.alpro_gemm_qkv_tattn(_ptr(a), a.stride(0), _ptr(w), w.stride(0), _ptr(b), _ptr(out), out.stride(0), _CODE[a.dtype], M, H, T, K, float(scale),
                                    _ptr(qkv), 3 * H * 64, _ptr(lse), _stream()), "alpro_gemm_qkv_tattn")
    return (out, qkv, lse) if want_qkv else out


def attn_cls(qkv, qkv_cls, batch, L, H, scale, group=1, key_bias=None, drop_p=0.0, drop_seed=0):
    """fp32 attention output of the CLS query (row 0) of every (sequence, head): qkv (batch*L, 3*H*64) 16-bit as written by the qkv GEMM,
    qkv_cls (batch / group, 3*H*64) fp32 = the CLS tokens' unrounded q | k | v -> (batch, H*64) fp32 (alpro_attn_cls_fwd)."""
    lib = load()
    _dev(qkv); _dev(qkv_cls, torch.float32)
    if qkv.shape != (batch * L, 3 * H * 64) or qkv_cls.shape != (batch // group, 3 * H * 64) or batch % group:
        raise RuntimeError("attn_cls: qkv %s / qkv_cls %s do not match batch=%d L=%d H=%d group=%d" % (tuple(qkv.shape), tuple(qkv_cls.shape), batch, L, H, group))
    if not (qkv.is_contiguous() and qkv_cls.is_contiguous()):
        raise RuntimeError("attn_cls: contiguous tensors only")
    out = torch.empty((batch, H * 64), dtype=torch.float32, device=qkv.device)
    kb = _dev(key_bias, torch.float32) if key_bias is not None else None
    _check(lib.alpro_attn_cls_fwd(_ptr(qkv), _CODE[qkv.dtype], _ptr(qkv_cls), _ptr(kb), _ptr(out), batch, L, H, group, scale, drop_p, drop_seed, _stream()),
           "alpro_attn_cls_fwd")
    return out


def gemm_rows(a, w, bias=None, act=ACT_NONE, row_scale=None, residual=None, ln=None, out=None):
    """fp32 Linear on a handful of rows (alpro_gemm_rows_f32): out = residual + row_scale * act(LN(a) @ w.T + bias); a (M, K), w (N, K) fp32
    (row-strided views allowed), ln = (gamma, beta, eps) fuses the LayerNorm of a's rows."""
    lib = load()
    _dev(a, torch.float32); _dev(w, torch.float32)
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K and a.stride(1) == 1 and w.stride(1) == 1
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    _dev(out, torch.float32)
    g = b = None
    eps = 0.0
    if ln is not None:
        g, b, eps = _dev(ln[0], torch.float32), _dev(ln[1], torch.float32), float(ln[2])
    if residual is not None:
        _dev(residual, torch.float32)
        assert residual.shape == (M, N) and residual.stride(1) == 1
    _check(lib.alpro_gemm_rows_f32(_ptr(a), a.stride(0), _ptr(w), w.stride(0), _ptr(out), out.stride(0), M, N, K,
                                   _ptr(_dev(bias, torch.float32)) if bias is not None else None, act,
                                   _ptr(_dev(row_scale, torch.float32)) if row_scale is not None else None,
                                   _ptr(residual), residual.stride(0) if residual is not None else 0, _ptr(g), _ptr(b), eps, _stream()), "alpro_gemm_rows_f32")
    return out


def patchify(img, dtype):
    """img (BT, C, H, W) fp32 -> (BT * (H/16) * (W/16), C*256) im2col rows."""
    lib = load()
    _dev(img, torch.float32)
    BT, C, Hh, Ww = img.shape
    out = torch.empty((BT * (Hh // 16) * (Ww // 16), C * 256), dtype=dtype, device=img.device)
    _check(lib.alpro_patchify(_ptr(img), _ptr(out), _CODE[dtype], BT, C, Hh, Ww, _stream()), "alpro_patchify")
    return out


def gather_seq(text, video, ti, vi, dtype=None):
    """The fusion encoder's input (alpro_gather_seq_fwd): text (Pt, Lt, D) and video (Pv, Lv, D) fp32 pools, ti / vi (S,) int64 ->
    (out32 (S*(Lt+Lv), D) fp32, out_t in `dtype` or None)."""
    lib = load()
    _dev(text, torch.float32); _dev(video, torch.float32); _dev(ti, torch.int64); _dev(vi, torch.int64)
    assert text.is_contiguous() and video.is_contiguous() and ti.is_contiguous() and vi.is_contiguous() and ti.numel() == vi.numel()
    S, Lt, Lv, D = ti.numel(), text.shape[1], video.shape[1], text.shape[2]
    out32 = torch.empty((S * (Lt + Lv), D), dtype=torch.float32, device=text.device)
    out_t = torch.empty((S * (Lt + Lv), D), dtype=dtype, device=text.device) if dtype not in (None, torch.float32) else None
    _check(lib.alpro_gather_seq_fwd(_ptr(text), _ptr(video), _ptr(ti), _ptr(vi), _ptr(out32), _ptr(out_t), _CODE[dtype] if out_t is not None else F32, S, Lt, Lv, D, _stream()),
           "alpro_gather_seq_fwd")
    return out32, out_t


def gather_seq_bwd(d32, d_t, ti, vi, Pt, Pv, Lt, Lv):
    """-> (dtext (Pt, Lt, D), dvideo (Pv, Lv, D)) fp32: per pool row the sum over the sequences that used it (alpro_gather_seq_bwd)."""
    lib = load()
    _dev(d32, torch.float32)
    D = d32.shape[-1]
    assert d32.is_contiguous() and (d_t is None or (d_t.is_contiguous() and d_t.shape == d32.shape))
    dtext = torch.empty((Pt, Lt, D), dtype=torch.float32, device=d32.device)
    dvideo = torch.empty((Pv, Lv, D), dtype=torch.float32, device=d32.device)
    _check(lib.alpro_gather_seq_bwd(_ptr(d32), _ptr(d_t), _CODE[d_t.dtype] if d_t is not None else F32, _ptr(ti), _ptr(vi), _ptr(dtext), _ptr(dvideo), ti.numel(), Pt, Pv, Lt, Lv, D,
                                    _stream()), "alpro_gather_seq_bwd")
    return dtext, dvideo


def cls_mean_residual(x_in, side, x_out, B, T):
    lib = load()
    D = x_in.shape[-1]
    _check(lib.alpro_cls_mean_residual(_ptr(_dev(x_in, torch.float32)), x_in.stride(0), _ptr(_dev(side, torch.float32)), _ptr(x_out), x_out.stride(0),
                                       B, T, D, _stream()), "alpro_cls_mean_residual")
    return x_out


def vit_final_pool(x, gamma, beta, eps, B, T, N, dtype):
    lib = load()
    _dev(x, torch.float32)
    D = x.shape[-1]
    out32 = torch.empty((B, N + 1, D), dtype=torch.float32, device=x.device)
    out_t = torch.empty((B, N + 1, D), dtype=dtype, device=x.device) if dtype != torch.float32 else None
    _check(lib.alpro_vit_final_pool(_ptr(x), _ptr(gamma), _ptr(beta), eps, _ptr(out32), _ptr(out_t), _CODE[dtype], B, T, N, D, _stream()),
           "alpro_vit_final_pool")
    return out32, (out_t if out_t is not None else out32)


def bert_embed(ids, word, pos, type_emb, gamma, beta, eps, dtype, stats=False, drop_p=0.0, drop_seed=0):
    lib = load()
    _dev(ids, torch.int64)
    B, L = ids.shape
    D = word.shape[1]
    rows = B * L
    y32 = torch.empty((rows, D), dtype=torch.float32, device=ids.device)
    y_t = torch.empty((rows, D), dtype=dtype, device=ids.device) if dtype != torch.float32 else None
    mean = torch.empty(rows, dtype=torch.float32, device=ids.device) if stats else None
    rstd = torch.empty(rows, dtype=torch.float32, device=ids.device) if stats else None
    _check(lib.alpro_bert_embed_fwd(_ptr(ids), _ptr(_dev(word, torch.float32)), _ptr(_dev(pos, torch.float32)), _ptr(_dev(type_emb, torch.float32)),
                                    _ptr(gamma), _ptr(beta), eps, _ptr(y32), _ptr(y_t), _CODE[dtype], _ptr(mean), _ptr(rstd), rows, L, D,
                                    drop_p, drop_seed, _stream()), "alpro_bert_embed_fwd")
    return y32, (y_t if y_t is not None else y32)


def cast(src, dtype, out=None):
    """fp32 -> dtype copy on device (parameters are kept fp32; 16-bit operand copies are refreshed per step)."""
    lib = load()
    _dev(src, torch.float32)
    if dtype == torch.float32:
        return src
    dst = out if out is not None else torch.empty(src.shape, dtype=dtype, device=src.device)
    assert dst.dtype == dtype and dst.numel() == src.numel() and dst.is_contiguous()
    _check(lib.alpro_cast_from_f32(_ptr(src), _ptr(dst), _CODE[dtype], src.numel(), _stream()), "alpro_cast_from_f32")
    return dst


def sumsq(x, out):
    """out (1,) fp32 += sum(x^2)."""
    lib = load()
    _dev(x, torch.float32); _dev(out, torch.float32)
    ws, wsb = _reduce_ws(x.device)
    _check(lib.alpro_sumsq(_ptr(x), x.numel(), _ptr(out), ws, min(wsb, 1 << 16), _stream()), "alpro_sumsq")
    return out


def adamw_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step_size, gnorm_sq=None, max_norm=0.0, grad_scale=1.0, dyn_state=None,
               grads_scaled=True, correct_bias=True, zero_grad=False, lp=None):
    """dyn_state: (4,) fp32 device tensor {loss scale, growth tracker, applied steps, skipped steps} -- see alpro_adamw_step.
    lp: optional 16-bit mirror of p (same numel), refreshed by the same pass (alpro_adamw_step_lp)."""
    lib = load()
    for t in (p, g, m, v):
        _dev(t, torch.float32)
    if dyn_state is not None:
        _dev(dyn_state, torch.float32)
        assert dyn_state.numel() >= 4 and gnorm_sq is not None
    if lp is not None:
        _dev(lp)
        if lp.dtype not in (torch.float16, torch.bfloat16) or lp.numel() != p.numel() or not lp.is_contiguous():
            raise RuntimeError("adamw_step: lp must be a contiguous 16-bit tensor with p's number of elements")
    _check(lib.alpro_adamw_step_lp(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), lr, beta1, beta2, eps, weight_decay, step_size,
                                   _ptr(gnorm_sq), max_norm, grad_scale, _ptr(dyn_state), int(bool(grads_scaled)), int(bool(correct_bias)), int(bool(zero_grad)),
                                   _ptr(lp), _CODE[lp.dtype] if lp is not None else 0, _stream()),
           "alpro_adamw_step")


def loss_scale_update(dyn_state, gnorm_sq, growth=2.0, backoff=0.5, window=2000, min_scale=1.0, max_scale=2.0 ** 24):
    _dev(dyn_state, torch.float32); _dev(gnorm_sq, torch.float32)
    _check(load().alpro_loss_scale_update(_ptr(dyn_state), _ptr(gnorm_sq), growth, backoff, int(window), min_scale, max_scale, _stream()),
           "alpro_loss_scale_update")


_TN_WORKSPACE = {}  # (device index, raw stream handle) -> grow-only byte buffer for the partial tiles of the weight-gradient GEMM


def _tn_workspace(device, nbytes):
    """The workspace of the CURRENT stream: its users run in stream order, so one buffer per (device, stream) is enough -- and a second
    stream (the weight-gradient side stream of alpro_amd.modeling.train.wgrad) must not share the launch stream's."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _TN_WORKSPACE.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _TN_WORKSPACE[key] = torch.empty(max(nbytes, 64 << 20), dtype=torch.uint8, device=device)
    return ws


_DETERMINISTIC_WGRAD = [os.environ.get("ALPRO_ATOMIC_WGRAD", "0") != "1"]
_DETERMINISTIC = [os.environ.get("ALPRO_DETERMINISTIC", "1") != "0"]


def set_deterministic_wgrad(on=True):
    """True (the default since round 3): every weight gradient takes the workspace path -- parameter gradients, bias gradients included,
    are bit-reproducible run to run; costs ~0.3 % of the training step.  False (opt-in, also ALPRO_ATOMIC_WGRAD=1): the faster of the
    two plans per shape, fp32 atomics above 65536 tokens -- see gemm_tn_acc."""
    _DETERMINISTIC_WGRAD[0] = bool(on)


def set_deterministic(on=True):
    """True (the default since round 4; ALPRO_DETERMINISTIC=0 turns it off): every remaining reduction of the training step is summed in a
    fixed order -- the LayerNorm backward's dgamma / dbeta / bias column sums and the gather-cast's (per-workgroup partials in the reduction
    workspace + a fixed-order second kernel; the CLS row's T frame terms under the FRAME_TOKENS scatter go the same way), the squared gradient
    norm, the embedding-table scatters; the VTC loss / temperature gradient are single-writer in the kernels themselves.  Two runs of the same step are then bitwise equal
    (tests/test_dist_gpu.py::test_two_runs_of_the_same_step_are_bitwise_equal).  False: the kernels' fp32-atomic forms (no workspace)."""
    _DETERMINISTIC[0] = bool(on)


def deterministic():
    return _DETERMINISTIC[0]


_REDUCE_WS_BYTES = 8192 * 3 * 768 * 4 + 4096 * 768 * 4   # up to 8192 workgroups' column sums (the default plan uses 2048) + the CLS frame terms of up to 4096 frames (B * T)


def _reduce_ws(device):
    """(pointer, bytes) of the reduction workspace handed to alpro_layernorm_bwd / _gather_cast / _sumsq: the head of the per-device grow-only
    buffer the weight-gradient GEMM also uses (all of them run on one stream, each finishes with its own reduce kernel) -- or (None, 0)
    when reproducibility is switched off."""
    if not _DETERMINISTIC[0]:
        return None, 0
    ws = _tn_workspace(device, _REDUCE_WS_BYTES)
    return _ptr(ws), _REDUCE_WS_BYTES


def gemm_tn_acc(a, b, c, colsum=None, atomic=None):
    """c (N, K) fp32 += a(M, N)^T @ b(M, K) with 16-bit a, b in their natural row-major layout; colsum (N,) fp32 += a.sum(0) (the
    bias gradient) from the same pass.
    atomic=False: alpro_gemm_tn_acc_ws -- the token ranges' partial tiles go through a workspace (one grow-only buffer per device,
    used in stream order) and are added in a fixed order: bit-reproducible.  atomic=True: alpro_gemm_tn_acc, partials combined by
    fp32 atomics (no workspace; run-to-run differences in the last bits, tests/test_hip_bwd_ops.py pins 2e-6 of scale).
    atomic=None: the workspace (deterministic) plan, unless set_deterministic_wgrad(False) opted into the measured faster one per shape
    (round-2 measurement) -- the workspace up to 65536 tokens or <= 9 tiles (the workgroups finish together there and the atomics
    queue up: -10..-25 %), atomics above (the ranges finish spread out and the atomics hide under the stragglers' MFMAs; the extra
    reduce launch costs +3..7 % of the kernel)."""
    lib = load()
    _dev(a); _dev(b, a.dtype); _dev(c, torch.float32)
    assert a.shape[0] == b.shape[0] and tuple(c.shape) == (a.shape[1], b.shape[1])
    if colsum is not None:
        _dev(colsum, torch.float32)
        assert colsum.numel() >= a.shape[1]
    M, N, K = a.shape[0], a.shape[1], b.shape[1]
    if atomic is None:
        atomic = not _DETERMINISTIC_WGRAD[0] and M > 65536 and ((N + 255) // 256) * ((K + 255) // 256) > 9
    if atomic:
        _check(lib.alpro_gemm_tn_acc(_ptr(a), a.stride(0), _ptr(b), b.stride(0), _ptr(c), c.stride(0), _CODE[a.dtype], M, N, K, _ptr(colsum), _stream()),
               "alpro_gemm_tn_acc")
        return c
    # always hand over the buffer, even when this shape needs none (0 bytes = the workspace plan keeps ONE token range): a NULL workspace
    # would select the atomic plan, which may split where the workspace plan does not
    ws = _tn_workspace(a.device, lib.alpro_gemm_tn_workspace_bytes(M, N, K))
    _check(lib.alpro_gemm_tn_acc_ws(_ptr(a), a.stride(0), _ptr(b), b.stride(0), _ptr(c), c.stride(0), _CODE[a.dtype], M, N, K, _ptr(colsum),
                                    _ptr(ws), ws.numel(), _stream()), "alpro_gemm_tn_acc_ws")
    return c


def transpose_jobs(pairs):
    """[(src (R, C) fp32 contiguous-row device tensor, out (C, Rpad) tensor)] -> (device job table, njobs, total_tiles) for
    transpose_batch; build once, launch every step."""
    jobs = (TransposeJob * len(pairs))()
    t0 = 0
    for j, (src, out) in zip(jobs, pairs):
        _dev(src, torch.float32); _dev(out)
        assert src.dim() == 2 and out.dim() == 2 and src.stride(1) == 1 and out.stride(1) == 1 and out.shape[0] == src.shape[1] and out.shape[1] >= src.shape[0]
        j.in_, j.out, j.ld_in, j.ld_out = src.data_ptr(), out.data_ptr(), src.stride(0), out.stride(0)
        j.R, j.C, j.Rpad, j.tile0 = src.shape[0], src.shape[1], out.shape[1], t0
        t0 += ((src.shape[1] + 63) // 64) * ((out.shape[1] + 63) // 64)
    table = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(pairs[0][0].device)
    return table, len(pairs), t0


def transpose_batch(table, njobs, total_tiles, out_dtype):
    _check(load().alpro_transpose_batch(_ptr(table), njobs, total_tiles, _CODE[out_dtype], _stream()), "alpro_transpose_batch")


def colsum_acc(a, out):
    lib = load()
    _dev(a); _dev(out, torch.float32)
    _check(lib.alpro_colsum_acc(_ptr(a), a.stride(0), _ptr(out), _CODE[a.dtype], a.shape[0], a.shape[1], _stream()), "alpro_colsum_acc")
    return out


def softmax_xent(logits, labels, grad_dtype=None, grad_scale=None, ignore_index=-100, pad_to=64):
    """logits (M, V) fp32, labels (M,) int64 -> loss_rows (M,) [, dlogits (M, Vpad) in grad_dtype scaled by *grad_scale]."""
    lib = load()
    _dev(logits, torch.float32); _dev(labels, torch.int64)
    M, V = logits.shape
    loss_rows = torch.empty(M, dtype=torch.float32, device=logits.device)
    dl, Vp = None, (V + pad_to - 1) // pad_to * pad_to
    if grad_dtype is not None:
        dl = torch.empty((M, Vp), dtype=grad_dtype, device=logits.device)
        _dev(grad_scale, torch.float32)
    _check(lib.alpro_softmax_xent(_ptr(logits), logits.stride(0), _ptr(labels), ignore_index, _ptr(loss_rows), _ptr(dl),
                                  _CODE[grad_dtype] if grad_dtype is not None else 0, Vp, _ptr(grad_scale), M, V, Vp, _stream()),
           "alpro_softmax_xent")
    return (loss_rows, dl) if dl is not None else loss_rows


def vtc_loss_fwd(v, t, gv, gt, temp, col0):
    """alpro_vtc_loss_fwd: returns (loss (), sim_v2t (B, G), sim_t2v (B, G), lse (2B,))."""
    lib = load()
    for x in (v, t, gv, gt):
        _dev(x, torch.float32)
        assert x.is_contiguous()
    _dev(temp, torch.float32)
    B, E = v.shape
    G = gv.shape[0]
    sim_v2t = torch.empty((B, G), dtype=torch.float32, device=v.device)
    sim_t2v = torch.empty_like(sim_v2t)
    lse = torch.empty(2 * B, dtype=torch.float32, device=v.device)
    loss = torch.empty((), dtype=torch.float32, device=v.device)
    _check(lib.alpro_vtc_loss_fwd(_ptr(v), _ptr(t), _ptr(gv), _ptr(gt), _ptr(temp), B, G, E, int(col0), _ptr(sim_v2t), _ptr(sim_t2v), _ptr(lse),
                                  _ptr(loss), _stream()), "alpro_vtc_loss_fwd")
    return loss, sim_v2t, sim_t2v, lse


def vtc_loss_bwd(v, t, gv, gt, temp, col0, sim_v2t, sim_t2v, lse, dloss, want_dtemp=True):
    """alpro_vtc_loss_bwd: returns (dv, dt, dgv, dgt, dtemp or None)."""
    lib = load()
    B, E = v.shape
    G = gv.shape[0]
    dev = v.device
    ds = torch.empty((2, B, G), dtype=torch.float32, device=dev)
    dv, dt = torch.empty_like(v), torch.empty_like(t)
    dgv, dgt = torch.empty_like(gv), torch.empty_like(gt)
    dtemp = torch.empty((), dtype=torch.float32, device=dev) if want_dtemp else None
    dloss = dloss.reshape(()).to(torch.float32).contiguous()
    _check(lib.alpro_vtc_loss_bwd(_ptr(v), _ptr(t), _ptr(gv), _ptr(gt), _ptr(temp), B, G, E, int(col0), _ptr(sim_v2t), _ptr(sim_t2v), _ptr(lse),
                                  _ptr(dloss), _ptr(ds[0]), _ptr(ds[1]), _ptr(dv), _ptr(dt), _ptr(dgv), _ptr(dgt), _ptr(dtemp), _stream()),
           "alpro_vtc_loss_bwd")
    return dv, dt, dgv, dgt, dtemp


def prepare_clips(raw, mean, std, scale, boxes=None, want_crop=True):
    """alpro_prepare_clips: raw (B, T, 3, H, W) uint8 / fp32 device tensor -> (visual, crop, context) fp32 (crop / context None
    without boxes).  boxes: (B, 4) int32 device tensor {top, left, h, w}."""
    lib = load()
    if not raw.is_cuda or not raw.is_contiguous() or raw.dtype not in (torch.uint8, torch.float32):
        raise RuntimeError("prepare_clips needs a contiguous uint8 / fp32 device tensor (no CPU fallback)")
    B, T, C, H, W = raw.shape
    assert C == 3
    vis = torch.empty(raw.shape, dtype=torch.float32, device=raw.device)
    crop = ctx = None
    if boxes is not None and want_crop:
        _dev(boxes, torch.int32)
        crop, ctx = torch.empty_like(vis), torch.empty_like(vis)
    m3, s3 = (ctypes.c_float * 3)(*[float(x) for x in mean]), (ctypes.c_float * 3)(*[float(x) for x in std])
    _check(lib.alpro_prepare_clips(_ptr(raw), int(raw.dtype == torch.uint8), _ptr(boxes) if crop is not None else None, float(scale), m3, s3,
                                   _ptr(vis), _ptr(crop), _ptr(ctx), B, T, H, W, _stream()), "alpro_prepare_clips")
    return vis, crop, ctx
