"""Runtime knobs of the MI355X path (not part of the reference's JSON surface; all optional)."""
import os

import torch

_DTYPES = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "f16": torch.float16, "fp16": torch.float16,
           "float16": torch.float16, "f32": torch.float32, "fp32": torch.float32, "float32": torch.float32}

# Storage dtype of GEMM / attention operands.  bf16 is the throughput mode BASELINE.json quotes;
# float32 is the exact mode (fp32 MFMA) that reproduces the reference's fp32 arithmetic.
_compute_dtype = _DTYPES[os.environ.get("ALPRO_COMPUTE_DTYPE", "bf16").lower()]


def compute_dtype():
    return _compute_dtype


def set_compute_dtype(dt):
    global _compute_dtype
    _compute_dtype = _DTYPES[dt.lower()] if isinstance(dt, str) else dt
    return _compute_dtype


class use_compute_dtype:
    """Context manager: `with use_compute_dtype(torch.float32): ...`"""

    def __init__(self, dt):
        self.dt = dt

    def __enter__(self):
        self.prev = compute_dtype()
        set_compute_dtype(self.dt)

    def __exit__(self, *a):
        set_compute_dtype(self.prev)


# ---- precise CLS rows (round 4).  The VTC features are projections of one row per sequence (the CLS token); in the 16-bit modes its own
# chain of roundings (LayerNorm output -> q/k/v -> attention output -> projection -> MLP, once per block) carries most of the VTC-logit
# error, and re-evaluating exactly those rows in fp32 costs < 1e-3 of the FLOPs (csrc/cls_precise.hip).
# ALPRO_CLS_PRECISE = auto (default: on with fp16 operands -- the mode that then meets "VTC logits within 1e-3" on every reference fixture;
# off with bf16, whose 8-bit mantissa stays far from the bar either way) | 1 | 0.
_cls_precise = [os.environ.get("ALPRO_CLS_PRECISE", "auto").lower()]
_cls_off = [0]


def cls_precise(dt=None):
    dt = dt if dt is not None else _compute_dtype
    if dt == torch.float32 or _cls_off[0] > 0:
        return False
    if _cls_precise[0] == "auto":
        return dt == torch.float16
    return _cls_precise[0] in ("1", "true", "on")


# The chain's launches are skinny (32 ... 512 rows against whole weight matrices: latency, not throughput).  On a second HIP stream, ordered
# against the main path by events (modeling/timesformer/vit.py::_ClsSide), they run beside the block's big launches instead of between them
# (round 5; it rests on the dynamic tile scheduler: a persistent GEMM that finds a CU taken must not wait a round for it).  Measured
# (profiles/r5_cls_stream_ab.txt): B = 32 encoder forward 18.48 -> 18.01 ms, training step unchanged (161.6 ms either way).
# ALPRO_CLS_STREAM = infer (default: the inference forward only) | 1 (training forward too) | 0.
_cls_stream = [os.environ.get("ALPRO_CLS_STREAM", "infer").lower()]


def cls_stream(training=False):
    v = _cls_stream[0]
    if v in ("1", "true", "on"):
        return True
    return v == "infer" and not training


def set_cls_stream(v):
    _cls_stream[0] = v.lower() if isinstance(v, str) else ("1" if v else "0")


# Round 6: the temporal half's qkv Linear and its T-frame attention as ONE launch (alpro_gemm_qkv_tattn: q | k | v are consumed out of the GEMM's
# accumulators; csrc/gemm_tattn.hip).  ALPRO_FUSE_TATTN = infer (default: every no-grad forward -- the retrieval / inference models and, inside a
# training step, the frozen prompter's visual pass) | 1 (the training forward too: the kernel then also writes q | k | v and the log-sum-exp rows
# for the backward) | 0 (the two launches everywhere).  Measured (MI355X, B = 64, fp16, one box): inference form 365 us against 308 + 115 us for
# the two launches inside a step; training form 441 us against the same 423 us -- its 256 x 192 tiles run the K loop at ~975 TF/s where the
# 8-phase qkv GEMM reaches ~1150, so with q | k | v still to be written the fusion does not pay in training and stays off there.
_fuse_tattn = [os.environ.get("ALPRO_FUSE_TATTN", "infer").lower()]


def fuse_temporal_attention(training=False):
    v = _fuse_tattn[0]
    if v in ("1", "true", "on"):
        return True
    return v == "infer" and not training


def set_fuse_temporal_attention(v):
    _fuse_tattn[0] = v.lower() if isinstance(v, str) else ("1" if v else "0")


# Round 6: the no-grad encoder forward as two half batches on two HIP streams (modeling/timesformer/vit.py::run_blocks).  Every big launch of a
# block is a persistent kernel whose last round of tiles leaves most CUs idle (the N = 768 projections at B = 32: 591 tiles = 2.3 rounds on 256
# CUs) and the next launch of the same stream depends on it; the other half's launches do not, and the dynamic tile scheduler lets their
# workgroups start on the CUs the first kernel's workgroups have left.  ALPRO_SPLIT_STREAMS = 0 | 1 | auto (default; on for even B >= 16).
_split_streams = [os.environ.get("ALPRO_SPLIT_STREAMS", "auto").lower()]


def split_streams(B):
    v = _split_streams[0]
    if v in ("0", "false", "off") or B % 2 != 0:
        return False
    if v in ("1", "true", "on"):
        return B >= 2
    return B >= 16


def split_lockstep():
    """ALPRO_SPLIT_LOCKSTEP=1: the two streams meet at every block boundary (measurement aid: a block's launches are then bracketed on the launch
    stream); default: they meet once, behind the last block."""
    return os.environ.get("ALPRO_SPLIT_LOCKSTEP", "0") == "1"


def set_split_streams(v):
    _split_streams[0] = v.lower() if isinstance(v, str) else ("1" if v else "0")


# Round 6 (third session): weight gradients on a side stream.  Inside an anchored backward (alpro_amd.modeling.train.Anchor) every 16-bit weight-gradient
# GEMM (alpro_gemm_tn_acc_ws: dW += dY^T X, a leaf of the backward graph -- nothing downstream reads it before the optimizer / the gradient exchange)
# is launched on a second HIP stream behind an event of the launch stream; the launch stream goes on with the data gradient and waits for the side
# stream once, where the backward returns (and before a gradient range is reported final to the exchange).  What it buys is what the two-stream
# forward buys: the weight-gradient kernel's workgroups take the CUs a persistent GEMM's last, partly filled round of tiles leaves idle, and vice
# versa.  Sums are unchanged bit for bit (each kernel's own order; accumulations into one .grad stay ordered on the side stream).
# Measured (profiles/r6_wgrad_side_stream_ab.txt): B = 64 pretrain step -0.3 ... -1.0 % on one box, +0.08 GB peak memory.  ALPRO_WGRAD_STREAM = 0 | 1 (default 1).
_wgrad_stream = [os.environ.get("ALPRO_WGRAD_STREAM", "1") != "0"]
_wgrad_active = [0]      # > 0 while an anchored backward runs
_WGRAD_SIDE = {}         # (device index, launch stream handle) -> [side stream, work pending]


def _several_ranks():
    """True in a job with more than one rank.  The weight-gradient and prompter side streams are single-GPU schedules unless forced
    (ALPRO_WGRAD_STREAM=force / ALPRO_PROMPTER_STREAM=force): with the collective library's stream the job would have more than four streams in play,
    i.e. streams sharing a hardware queue -- and a weight-gradient kernel queued in front of an all-reduce (or behind one) on a shared queue is exactly
    the coupling the overlapped exchange must not have.  Nothing of this could be measured on more than one GPU; with the text
    side stream alone a rank has launch + collective + text + the prompter's second half-batch stream = four, each on a queue of its own."""
    try:
        import torch.distributed as td
        return td.is_available() and td.is_initialized() and td.get_world_size() > 1
    except Exception:  # noqa: BLE001
        return False


def _new_side_stream(device):
    """A side stream; ALPRO_SIDE_PRIORITY = default | high | low picks its HIP priority (measurement switch: high costs 14 %, there is no class below normal -- profiles/r6_hw_queues.txt, section 4)."""
    import torch
    pr = os.environ.get("ALPRO_SIDE_PRIORITY", "default")
    if pr == "default":
        return torch.cuda.Stream(device)
    lo, hi = torch.cuda.Stream.priority_range()   # (lowest, highest): numerically lower = higher priority
    return torch.cuda.Stream(device, priority=hi if pr == "high" else lo)


def set_wgrad_stream(v):
    _wgrad_stream[0] = bool(v)


def wgrad_stream_enabled():
    return _wgrad_stream[0]


def wgrad_scope(enter):
    _wgrad_active[0] += 1 if enter else -1


def wgrad_side_stream(device):
    """The side stream of the current launch stream, or None when weight gradients stay on the launch stream."""
    if not (_wgrad_stream[0] and _wgrad_active[0] > 0 and device.type == "cuda") or (_several_ranks() and os.environ.get("ALPRO_WGRAD_STREAM") != "force"):
        return None
    import torch
    cur = torch.cuda.current_stream(device)
    if os.environ.get("ALPRO_WGRAD_STREAM_NESTED", "0") != "1" and any(s.cuda_stream == cur.cuda_stream for s in _TEXT_SIDE.values()):
        # a backward that already runs on the text side stream keeps its weight gradients there: at most four streams are active at a time (launch, text,
        # weight gradients in backward; launch, text, 2 x prompter in forward) -- the number of hardware queues that do not share a dispatch pipe
        # (profiles/r6_hw_queues.txt; nested form 148.75 ms, this one 148.5)
        return None
    key = (device.index if device.index is not None else torch.cuda.current_device(), cur.cuda_stream)
    ent = _WGRAD_SIDE.get(key)
    if ent is None:
        ent = _WGRAD_SIDE[key] = [_new_side_stream(device), False]
    ent[1] = True
    return ent[0]


def side_streams_of_current(device):
    """Raw handles of the side streams that carry weight-gradient GEMMs for the current launch stream (created if the switch is on): per-stream
    launch options of the launch stream (FlatAdamW's cu_budget while all-reduces are in flight) have to be mirrored onto them -- the weight-gradient
    kernel plans one workgroup per available CU, and it runs on the side stream now."""
    if not (_wgrad_stream[0] and device.type == "cuda") or (_several_ranks() and os.environ.get("ALPRO_WGRAD_STREAM") != "force"):
        return []
    import torch
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)
    ent = _WGRAD_SIDE.get(key)
    if ent is None:
        ent = _WGRAD_SIDE[key] = [_new_side_stream(device), False]
    return [ent[0].cuda_stream]


def join_wgrad():
    """The current stream of every device waits for the weight gradients that were launched from it."""
    if not _WGRAD_SIDE:
        return
    import torch
    for (idx, handle), ent in _WGRAD_SIDE.items():
        if ent[1]:
            cur = torch.cuda.current_stream(idx)
            if cur.cuda_stream == handle:
                cur.wait_stream(ent[0])
                ent[1] = False


# Round 6 (third session): the pretraining forward's 2B-caption text-encoder pass (M = 2B x 40 rows: 60 ... 240 tiles per Linear, a quarter to all of the
# CUs for a few microseconds each) on a side stream beside the visual encoder's forward, and -- autograd runs a node's backward on the stream of its
# forward -- its backward beside the visual encoder's backward.  The launch stream waits for the side stream where the text rows are first needed
# (the VTC features); the side stream's parameter gradients are handed to the caller's stream by an end-of-backward callback, and any gradient range
# reported final to the exchange (N > 1) waits for the side stream first.  Measured (profiles/r6_text_side_stream_ab.txt): B = 64 pretrain step
# 151.7 -> 149.4 ms (-1.5 %, A/B/A/B on one box), outputs and gradients bit for bit, +0.14 GB.  ALPRO_TEXT_STREAM = 0 | 1 (default 1).
_text_stream = [os.environ.get("ALPRO_TEXT_STREAM", "1") != "0"]
_TEXT_SIDE = {}          # (device index, launch stream handle) -> side stream


def set_text_stream(v):
    _text_stream[0] = bool(v)


def text_stream_enabled():
    return _text_stream[0]


def text_side_stream(device):
    if not (_text_stream[0] and device.type == "cuda"):
        return None
    import torch
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)
    s = _TEXT_SIDE.get(key)
    if s is None:
        s = _TEXT_SIDE[key] = _new_side_stream(device)
    return s


def is_text_side_stream(stream):
    return any(s.cuda_stream == stream.cuda_stream for s in _TEXT_SIDE.values())


def join_text_streams():
    """The current stream waits for every text side stream that was forked from it (gradient ranges about to go on the wire)."""
    if not _TEXT_SIDE:
        return
    import torch
    for (idx, handle), s in _TEXT_SIDE.items():
        cur = torch.cuda.current_stream(idx)
        if cur.cuda_stream == handle:
            cur.wait_stream(s)


# Third session: the frozen prompter's pass (get_pseudo_labels: a no-grad B-clip encoder forward, itself two half batches on two streams)
# forked BEHIND the trained visual encoder's forward, so that it runs beside the VTC / negative sampling / fusion / head launches of the launch stream
# instead of after them.  Forked at the START of the forward it ran beside the visual encoder's forward and cost +0.2 ... +0.7 % (profiles/r6_split_streams_ab.txt:
# big beside big); forked behind it: 150.47 -> 150.09 ms (-0.25 %, A/B/A/B on one box, run-to-run spread 0.03 ms; profiles/r6_prompter_side_stream_ab.txt),
# labels bit for bit.  ALPRO_PROMPTER_STREAM = 0 | 1 (default 1).
_prompter_stream = [os.environ.get("ALPRO_PROMPTER_STREAM", "1") != "0"]
_PROMPTER_SIDE = {}


def set_prompter_stream(v):
    _prompter_stream[0] = bool(v)


def prompter_stream_enabled():
    return _prompter_stream[0]


def prompter_side_stream(device):
    if not (_prompter_stream[0] and device.type == "cuda") or (_several_ranks() and os.environ.get("ALPRO_PROMPTER_STREAM") != "force"):
        return None
    import torch
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)
    s = _PROMPTER_SIDE.get(key)
    if s is None:
        s = _PROMPTER_SIDE[key] = _new_side_stream(device)
    return s


# Round 6: the no-grad Block.forward keeps the block input until both attention halves are done -- the add + norm1 kernel reads it and writes only the
# normalised rows, the add + norm2 kernel adds the temporal AND the spatial branch (alpro_add_layernorm_pre_mlp2; bit for bit the same sums).
# ALPRO_DEFER_TEMPORAL_ADD=0: the round-3 form (x + temporal branch written by the first kernel), for A/B.
_defer_tadd = [os.environ.get("ALPRO_DEFER_TEMPORAL_ADD", "1") != "0"]


def defer_temporal_add():
    return _defer_tadd[0]


def set_defer_temporal_add(v):
    _defer_tadd[0] = bool(v)


def set_cls_precise(v):
    _cls_precise[0] = v.lower() if isinstance(v, str) else ("1" if v else "0")


class use_cls_precise:
    """`with use_cls_precise(False): ...` (tests, bench.py's mode table)."""

    def __init__(self, v):
        self.v = v

    def __enter__(self):
        self.prev = _cls_precise[0]
        set_cls_precise(self.v)

    def __exit__(self, *a):
        _cls_precise[0] = self.prev


class cls_precise_off:
    """Passes whose CLS row feeds no VTC logit (the frozen teacher's pseudo-label pass) skip the side path."""

    def __enter__(self):
        _cls_off[0] += 1

    def __exit__(self, *a):
        _cls_off[0] -= 1


# ---- fp16 operands: loss scaling (alpro_amd.amp).  The hand-written backward passes refuse to run on fp16 gradient operands unless the
# loss was scaled (an unscaled fp16 backward silently flushes most activation gradients to zero); the LM head writes its logit gradient
# at FORWARD time and therefore needs to know the scale that the coming backward will use ("armed" scaler).
_scaling = [None]   # the LossScaler of the scale_loss context we are inside, else None
_armed = [None]     # weakref to the most recently attached LossScaler


def set_armed_loss_scaler(scaler):
    import weakref
    _armed[0] = weakref.ref(scaler) if scaler is not None else None


def armed_loss_scale(device):
    """(1,) device tensor holding the loss scale the next backward will carry, or None (not fp16 / no scaler attached)."""
    if _compute_dtype != torch.float16 or _armed[0] is None:
        return None
    sc = _armed[0]()
    return None if sc is None else sc.to(device).scale


class loss_scaling:
    """`with loss_scaling(scaler): scaled_loss.backward()` -- marks the backward as scaled."""

    def __init__(self, scaler):
        self.scaler = scaler

    def __enter__(self):
        self.prev = _scaling[0]
        _scaling[0] = self.scaler
        return self.scaler

    def __exit__(self, *a):
        _scaling[0] = self.prev


def loss_scaling_active():
    return _scaling[0] is not None


def check_backward_precision(dt):
    """Called by every hand-written backward: fp16 gradient operands without loss scaling are refused loudly."""
    if dt == torch.float16 and _scaling[0] is None and os.environ.get("ALPRO_ALLOW_UNSCALED_FP16_BACKWARD", "0") != "1":
        raise RuntimeError("fp16 operands: the backward pass needs a scaled loss (activation gradients underflow fp16's range otherwise). Use "
                           "`optimizer.backward(loss)` (alpro_amd.optim.FlatAdamW) or `with amp.scale_loss(loss, optimizer) as s: s.backward()` "
                           "(alpro_amd.amp / the apex.amp facade), or pick bf16 / fp32 operands (ALPRO_COMPUTE_DTYPE).")


# ---- dropout seeds: the HIP kernels draw masks from hash(seed, element index); every dropout site of every forward
# gets a fresh 31-bit seed from this counter-based stream (deterministic given seed_dropout()).
#
# Seeding: unless seed_dropout() was called explicitly, the stream is keyed on first use by torch.initial_seed() (what the
# reference's set_random_seed -> torch.manual_seed controls, src/utils/misc.py:20-24) plus the data-parallel rank, so a different run
# seed gives different masks and the ranks of one job do not drop the same element indices.
_drop_state = [None, 0]


def seed_dropout(seed):
    _drop_state[0] = int(seed) & 0x7FFFFFFF
    _drop_state[1] = 0


def _auto_seed():
    import torch.distributed as td
    rank = td.get_rank() if (td.is_available() and td.is_initialized()) else 0
    seed_dropout((torch.initial_seed() * 0x9E3779B1 + rank * 0x85EBCA77 + 0x1234567) & 0x7FFFFFFF)


def next_dropout_seed():
    if _drop_state[0] is None:
        _auto_seed()
    _drop_state[1] += 1
    x = (_drop_state[0] * 0x9E3779B1 + _drop_state[1] * 0x85EBCA77) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x2C1B3C6D) & 0xFFFFFFFF
    x ^= x >> 12
    return (x & 0x7FFFFFFF) | 1  # never 0 (0 means "dropout off" in the C ABI)
