"""Thin Python layer over the C-ABI: tensor -> pointer plumbing and the torch.autograd.Function wrappers that make
the HIP kernels differentiable building blocks of the reference's modules.

Precision modes (set with `set_compute_dtype`):
  * torch.float32  -- "parity mode": exact-f32 MFMA GEMMs, f32 activations; checked against the CPU oracle to 1e-4.
  * torch.bfloat16 -- "throughput mode": bf16 MFMA GEMMs with f32 accumulation, bf16 GEMM operands, f32 residual
                      stream / LayerNorm statistics / softmax / loss / optimizer state.
Every tensor handed to a kernel is allocated by PyTorch; all arithmetic happens in libevtpretrain.so.
"""
import ctypes as C
import math
import os

import numpy as np

import torch

from . import _lib
from ._lib import (ACT_DGELU, ACT_DRELU, ACT_GELU, ACT_NONE, ACT_RELU, EVP_BF16, EVP_F32, GemmDesc, call, dt, ptr,
                   stream_ptr)

_compute_dtype = torch.float32


def set_compute_dtype(dtype):
    """torch.float32 (parity mode) or torch.bfloat16 (throughput mode)."""
    global _compute_dtype
    if isinstance(dtype, str):
        dtype = {"fp32": torch.float32, "f32": torch.float32, "float32": torch.float32,
                 "bf16": torch.bfloat16, "bfloat16": torch.bfloat16}[dtype]
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("compute dtype must be float32 or bfloat16")
    _compute_dtype = dtype


def get_compute_dtype():
    return _compute_dtype


def _code(dtype):
    return EVP_BF16 if dtype == torch.bfloat16 else EVP_F32


def _chk(t, dtype=None):
    if not t.is_cuda:
        raise _lib.EvpError("tensor is not in device memory; eventpretrain_amd has no CPU path")
    if not t.is_contiguous():
        raise _lib.EvpError("kernel operand must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise _lib.EvpError(f"expected {dtype}, got {t.dtype}")
    return t


# ----------------------------------------------------------------------------------------------------- raw calls
def gemm(a, b, out, *, M, N, K, trans_a=False, trans_b=False, lda=None, ldb=None, ldc=None, bias=None, act=ACT_NONE,
         aux=None, residual=None, alpha=1.0, accumulate=False, batch=(1, 1), stride_a=(0, 0), stride_b=(0, 0),
         stride_c=(0, 0), a_off=0, b_off=0, c_off=0, tile=0, splitk=0):
    """out = epilogue(alpha * A . B^T); offsets/strides in ELEMENTS. See include/evtpretrain.h (evp_gemm)."""
    if not (a.is_cuda and b.is_cuda and out.is_cuda):
        raise _lib.EvpError("gemm operands must be in device memory; eventpretrain_amd has no CPU path")
    if a.dtype != b.dtype:
        raise _lib.EvpError("gemm: A and B must share a dtype")
    d = GemmDesc()
    d.dtype = dt(a)
    d.transA, d.transB = int(trans_a), int(trans_b)
    d.M, d.N, d.K = int(M), int(N), int(K)
    es = a.element_size()
    d.A = a.data_ptr() + a_off * es
    d.lda = int(lda if lda is not None else (M if trans_a else K))
    d.strideA0, d.strideA1 = int(stride_a[0]), int(stride_a[1])
    d.B = b.data_ptr() + b_off * es
    d.ldb = int(ldb if ldb is not None else (N if trans_b else K))
    d.strideB0, d.strideB1 = int(stride_b[0]), int(stride_b[1])
    d.C = out.data_ptr() + c_off * out.element_size()
    d.c_dtype = dt(out)
    d.ldc = int(ldc if ldc is not None else N)
    d.strideC0, d.strideC1 = int(stride_c[0]), int(stride_c[1])
    d.batch0, d.batch1 = int(batch[0]), int(batch[1])
    d.alpha = float(alpha)
    d.bias = ptr(bias)
    d.act = int(act)
    d.aux = ptr(aux)
    d.ldaux = d.ldc
    d.residual = ptr(residual)
    d.ldres = d.ldc
    d.accumulate = int(accumulate)
    d.tile = int(tile)
    d.splitk = int(splitk)
    call("evp_gemm", C.byref(d), stream_ptr())
    return out


def cast(x, dtype):
    if x.dtype == dtype:
        return x
    out = torch.empty_like(x, dtype=dtype)
    call("evp_cast", ptr(_chk(x)), dt(x), ptr(out), dt(out), x.numel(), stream_ptr())
    return out


def lp_weight(w):
    """Weight in the compute dtype. In bf16 mode the shadow copy is cached on the Parameter and kept current by
    FusedAdamW (which writes it in the same kernel that updates the f32 master weight)."""
    if _compute_dtype == torch.float32:
        return w.detach()
    sh = getattr(w, "_evp_lp", None)
    if sh is None or getattr(w, "_evp_lp_version", -1) != w._version or sh.device != w.device:
        sh = torch.empty(w.shape, dtype=torch.bfloat16, device=w.device)
        call("evp_cast", ptr(_chk(w.detach())), EVP_F32, ptr(sh), EVP_BF16, w.numel(), stream_ptr())
        w._evp_lp = sh
        w._evp_lp_version = w._version
    return sh


def refresh_lp_shadows(params):
    """Re-cast the f32 master weights into their EXISTING bf16 shadow buffers (same addresses: captured HIP graphs keep
    reading them). Needed after weights were changed outside FusedAdamW, e.g. by load_state_dict."""
    for w in params:
        sh = getattr(w, "_evp_lp", None)
        if sh is not None and sh.device == w.device:
            call("evp_cast", ptr(_chk(w.detach())), EVP_F32, ptr(sh), EVP_BF16, w.numel(), stream_ptr())
            w._evp_lp_version = w._version


def colsum(x2d, out=None):
    M, N = x2d.shape
    out = torch.empty(N, dtype=torch.float32, device=x2d.device) if out is None else out
    nb = call("evp_colsum_nblk", M)
    ws = torch.empty(nb * N, dtype=torch.float32, device=x2d.device)
    call("evp_colsum", ptr(_chk(x2d)), dt(x2d), M, N, N, ptr(out), ptr(ws), stream_ptr())
    return out


def layernorm_fwd(x, gamma, beta, eps, out_dtype, x2=None, x3=None):
    M, D = x.shape
    y = torch.empty(M, D, dtype=out_dtype, device=x.device)
    mean = torch.empty(M, dtype=torch.float32, device=x.device)
    rstd = torch.empty(M, dtype=torch.float32, device=x.device)
    call("evp_layernorm_fwd", ptr(_chk(x, torch.float32)), ptr(x2), ptr(x3), ptr(gamma), ptr(beta), M, D, float(eps),
         ptr(y), dt(y), ptr(mean), ptr(rstd), stream_ptr())
    return y, mean, rstd


# Side information that travels with a residual-stream gradient from the LayerNorm backward that produced it to the
# transformer block that consumes it next (autograd hands the tensor over, not these): its bf16 copy (the next GEMM's
# operand, written by the same kernel) and the per-block column sums of it (the proj / fc2 bias gradient). Keyed by the
# data pointer and validated by (a) a weak reference to the producing tensor -- a dead producer means the address may have
# been reused -- and (b) its version counter, which views share: where a tensor has several consumers autograd SUMS the
# incoming gradients and may do so in place into the first arrival, which keeps the pointer and changes the values. A handful
# of entries at most.
_grad_side = {}
_use_grad_side = os.environ.get("EVP_GRAD_SIDE", "1") != "0"


def set_grad_side(flag):
    """A/B switch: bf16 copy + bias column sums of the residual-stream gradient handed from LayerNorm backward to the next
    block (default) or a cast kernel and a full column sum per use."""
    global _use_grad_side
    _use_grad_side = bool(flag)


def _side_put(dx, lp, colsum_part):
    import weakref
    for k in [k for k, v in _grad_side.items() if v[0]() is None]:
        del _grad_side[k]
    while len(_grad_side) >= 8:
        _grad_side.pop(next(iter(_grad_side)))
    _grad_side[dx.data_ptr()] = (weakref.ref(dx), dx.numel(), lp, colsum_part, dx._version)


def _side_take(g):
    ent = _grad_side.pop(g.data_ptr(), None)
    if ent is None or ent[0]() is None or ent[1] != g.numel() or not g.is_contiguous() or g._version != ent[4]:
        return None, None
    return ent[2], ent[3]


def layernorm_bwd(dy, x, gamma, mean, rstd, gres=None, x2=None, x3=None, want_lp=False, params=None, side=False):
    """Returns (dx f32, dx_lp bf16|None, dgamma, dbeta). With `params=(weight, bias)` leaf Parameters that may take
    deferred gradients, the per-block dgamma/dbeta partials are queued for the step's grouped column-sum launch
    instead of being reduced by a kernel of their own, and (dx, dx_lp, None, None) is returned. `side` (deferred mode
    only): the kernel also leaves per-block column sums of dx, and (bf16 copy, those partials) are registered for whoever
    receives dx as its incoming gradient (_side_take)."""
    M, D = x.shape
    dx = torch.empty(M, D, dtype=torch.float32, device=x.device)
    dx_lp = torch.empty(M, D, dtype=torch.bfloat16, device=x.device) if want_lp else None
    nb = call("evp_layernorm_bwd_nblk", M)
    defer = (params is not None and D % 8 == 0 and _deferred.can_defer(params[0]) and _deferred.can_defer(params[1]))
    # (the three-partial-row kernel keeps 4 rows x 3 D floats in LDS: D <= 3412; wider rows take the plain kernel below and the
    # consumer of dx casts / column-sums it itself)
    if defer and side and _use_grad_side and 4 * 3 * D * 4 <= 160 * 1024:
        ws = torch.empty(nb, 3 * D, dtype=torch.float32, device=x.device)
        call("evp_layernorm_bwd_cs", ptr(_chk(dy)), dt(dy), ptr(x), ptr(x2), ptr(x3), ptr(gamma), ptr(mean), ptr(rstd),
             ptr(gres), M, D, ptr(dx), ptr(dx_lp), ptr(ws), stream_ptr())
        _deferred.colsum(params[0], ws[:, :D])
        _deferred.colsum(params[1], ws[:, D:2 * D])
        _side_put(dx, dx_lp, ws[:, 2 * D:])
        return dx, dx_lp, None, None
    ws = torch.empty(nb, 2 * D, dtype=torch.float32, device=x.device)
    dgamma = dbeta = None
    if not defer:
        dgamma = torch.empty(D, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(D, dtype=torch.float32, device=x.device)
    call("evp_layernorm_bwd", ptr(_chk(dy)), dt(dy), ptr(x), ptr(x2), ptr(x3), ptr(gamma), ptr(mean), ptr(rstd),
         ptr(gres), M, D, ptr(dx), ptr(dx_lp), ptr(dgamma), ptr(dbeta), ptr(ws), stream_ptr())
    if defer:
        _deferred.colsum(params[0], ws[:, :D])
        _deferred.colsum(params[1], ws[:, D:])
    return dx, dx_lp, dgamma, dbeta


_use_fused_attention = True
_use_window_mfma = True


def set_window_mfma(flag):
    """A/B switch: Swin window attention on the MFMA kernels (default, bf16 mode) or on the f32 LDS kernels."""
    global _use_window_mfma
    _use_window_mfma = bool(flag)



def set_fused_attention(flag):
    """A/B switch: the fused per-head kernels (default) or the batched-GEMM + softmax formulation."""
    global _use_fused_attention
    _use_fused_attention = bool(flag)


def fused_attention_ok(dtype, N, dh):
    return _use_fused_attention and bool(call("evp_attention_fused_supported", _code(dtype), int(N), int(dh)))


def attention_fused_fwd(qkv, B, N, heads, dh, want_probs=False):
    """-> (out [B*N, h*dh] bf16, lse [B,h,N] f32, probs [B,h,N,ldp] bf16 | None)."""
    dev = qkv.device
    out = torch.empty(B * N, heads * dh, dtype=qkv.dtype, device=dev)
    lse = torch.empty(B, heads, N, dtype=torch.float32, device=dev)
    ldp = (N + 7) // 8 * 8
    probs = torch.empty(B, heads, N, ldp, dtype=qkv.dtype, device=dev) if want_probs else None
    call("evp_attention_fused_fwd", ptr(_chk(qkv)), B, N, heads, dh, float(dh) ** -0.5, ptr(out), ptr(lse), ptr(probs), ldp,
         stream_ptr())
    return out, lse, probs


def attention_fused_bwd(qkv, out, dout, lse, B, N, heads, dh):
    dqkv = torch.empty_like(qkv)
    call("evp_attention_fused_bwd", ptr(qkv), ptr(out), ptr(_chk(dout)), ptr(lse), B, N, heads, dh, float(dh) ** -0.5, ptr(dqkv),
         stream_ptr())
    return dqkv


def attention_fwd(qkv, B, N, heads, dh):
    """qkv [B*N, 3*h*dh] (compute dtype) -> (probs [B,h,N,ldp], out [B*N, h*dh])."""
    ldp = (N + 7) // 8 * 8
    dev = qkv.device
    scores = torch.empty(B * heads * N * ldp, dtype=torch.float32, device=dev)
    probs = torch.empty(B, heads, N, ldp, dtype=qkv.dtype, device=dev)
    out = torch.empty(B * N, heads * dh, dtype=qkv.dtype, device=dev)
    call("evp_attention_fwd", ptr(_chk(qkv)), dt(qkv), B, N, heads, dh, float(dh) ** -0.5, ptr(scores), ptr(probs), ldp,
         ptr(out), stream_ptr())
    return probs, out


def attention_bwd(qkv, probs, dout, B, N, heads, dh):
    ldp = probs.shape[-1]
    dev = qkv.device
    dp = torch.empty(B * heads * N * ldp, dtype=torch.float32, device=dev)
    ds = dp if qkv.dtype == torch.float32 else torch.empty(B * heads * N * ldp, dtype=qkv.dtype, device=dev)
    dqkv = torch.empty_like(qkv)
    call("evp_attention_bwd", ptr(qkv), ptr(probs), ptr(_chk(dout)), dt(qkv), B, N, heads, dh, float(dh) ** -0.5, ldp,
         ptr(dp), ptr(ds), ptr(dqkv), stream_ptr())
    return dqkv


def attention_dropout_fwd(qkv, B, N, heads, dh, rd):
    """Attention core with dropout on the probabilities (vit_block.py:134-140 with attn_drop > 0): the batched-GEMM formulation, because
    the probabilities have to exist -- scores GEMM (f32), row softmax, evp_dropout_fwd on the probabilities (mask = rd.masks["attn"] when
    given, else drawn from rd's stream), dropped-probabilities x V. -> (P softmax, dropped P, keep mask uint8, out). The fused kernels
    stay what every attn_drop_rate = 0 run (all of the reference's scripts) uses."""
    T = qkv.dtype
    C_ = heads * dh
    tok = 3 * C_
    ldp = (N + 7) // 8 * 8
    dev = qkv.device
    scores = torch.empty(B * heads * N * ldp, dtype=torch.float32, device=dev)
    gemm(qkv, qkv, scores, M=N, N=N, K=dh, lda=tok, ldb=tok, ldc=ldp, batch=(B, heads), stride_a=(N * tok, dh), stride_b=(N * tok, dh),
         stride_c=(heads * N * ldp, N * ldp), b_off=C_, alpha=float(dh) ** -0.5)
    probs = torch.empty(B, heads, N, ldp, dtype=T, device=dev)
    call("evp_softmax_rows", ptr(scores), ptr(probs), dt(probs), B * heads * N, N, ldp, stream_ptr())
    sub = BlockDrop(drop=rd.attn_drop, seed=rd.seed_dev if rd.seed_dev is not None else rd.seed,
                    masks=None if (rd.masks is None or "attn" not in rd.masks) else {"attn": rd.masks["attn"]})
    sub._n = rd.next_offset(probs.numel())
    dropped, mk = dropout_fwd(probs, sub, "attn")
    out = torch.empty(B * N, C_, dtype=T, device=dev)
    gemm(dropped, qkv, out, M=N, N=dh, K=N, trans_b=True, lda=ldp, ldb=tok, ldc=C_, batch=(B, heads), stride_a=(heads * N * ldp, N * ldp),
         stride_b=(N * tok, dh), stride_c=(N * C_, dh), b_off=2 * C_)
    return probs, dropped, mk, out


def attention_dropout_bwd(qkv, probs, dropped, mk, dout, B, N, heads, dh, p_drop):
    """Backward of attention_dropout_fwd: dV = dropped^T dO; dP = mask / (1 - p) * (dO V^T); dS = P * (dP - rowsum(P dP)); dQ, dK."""
    T = qkv.dtype
    C_ = heads * dh
    tok = 3 * C_
    ldp = probs.shape[-1]
    dev = qkv.device
    sP = (heads * N * ldp, N * ldp)
    dqkv = torch.empty_like(qkv)
    dp = torch.empty(B * heads * N * ldp, dtype=torch.float32, device=dev)
    gemm(dout, qkv, dp, M=N, N=N, K=dh, lda=C_, ldb=tok, ldc=ldp, batch=(B, heads), stride_a=(N * C_, dh), stride_b=(N * tok, dh), stride_c=sP,
         b_off=2 * C_)
    dpm = dropout_bwd(dp, mk, p_drop)
    ds = dpm if T == torch.float32 else torch.empty(B * heads * N * ldp, dtype=T, device=dev)
    call("evp_softmax_rows_bwd", ptr(probs), ptr(dpm), ptr(ds), dt(probs), B * heads * N, N, ldp, stream_ptr())
    gemm(dropped, dout, dqkv, M=N, N=dh, K=N, trans_a=True, trans_b=True, lda=ldp, ldb=C_, ldc=tok, batch=(B, heads), stride_a=sP,
         stride_b=(N * C_, dh), stride_c=(N * tok, dh), c_off=2 * C_)
    gemm(ds, qkv, dqkv, M=N, N=dh, K=N, trans_b=True, lda=ldp, ldb=tok, ldc=tok, batch=(B, heads), stride_a=sP, stride_b=(N * tok, dh),
         stride_c=(N * tok, dh), b_off=C_, alpha=float(dh) ** -0.5)
    gemm(ds, qkv, dqkv, M=N, N=dh, K=N, trans_a=True, trans_b=True, lda=ldp, ldb=tok, ldc=tok, batch=(B, heads), stride_a=sP,
         stride_b=(N * tok, dh), stride_c=(N * tok, dh), c_off=C_, alpha=float(dh) ** -0.5)
    return dqkv


def mask_from_noise(noise, mask_ratio):
    """vit.py:75-103 on an explicit noise tensor -> (ids_keep int64, mask f32, ids_restore int64)."""
    B, L = noise.shape
    keep = int(L * (1 - mask_ratio))
    dev = noise.device
    ids_keep = torch.empty(B, keep, dtype=torch.int64, device=dev)
    mask = torch.empty(B, L, dtype=torch.float32, device=dev)
    ids_restore = torch.empty(B, L, dtype=torch.int64, device=dev)
    call("evp_mask_from_noise", ptr(_chk(noise, torch.float32)), B, L, keep, ptr(ids_keep), ptr(mask), ptr(ids_restore),
         stream_ptr())
    return ids_keep, mask, ids_restore


def density_noise(x, patch, sign):
    B, Cc, H, W = x.shape
    out = torch.empty(B, (H // patch) * (W // patch), dtype=torch.float32, device=x.device)
    call("evp_density_noise", ptr(_chk(x, torch.float32)), B, Cc, H, W, patch, float(sign), ptr(out), stream_ptr())
    return out


def add(a, b, c=None):
    out = torch.empty_like(a)
    call("evp_add_f32", ptr(_chk(a, torch.float32)), ptr(_chk(b, torch.float32)), ptr(c), a.numel(), ptr(out), stream_ptr())
    return out


# ----------------------------------------------------------------------------------------------------- deferred grads
class _DeferredGrads:
    """Weight / bias gradients do not feed the backward chain, only the optimizer. In bf16 mode they are therefore
    queued during backward and computed at its end by ONE grouped GEMM launch (thousands of 128x128 tiles: no split-K,
    no tail) and ONE grouped column-sum launch (csrc: evp_gemm_grouped_tn_bf16, evp_colsum_grouped). The autograd
    functions return None for a queued parameter; flush() -- an autograd end-of-backward callback -- writes the
    result straight into `param.grad` (allocating it, or accumulating into an existing one)."""

    def __init__(self):
        self.enabled = True
        self.hold = False               # keep the queue at the end of backward (see build_plan)
        self.w, self.b = [], []
        self.armed = False
        self._pin = {}
        self._capture_pins = []     # pinned tables owned by captured HIP graphs (never rewritten; see _stage)
        self.flat_buffers = []      # flat f32 buffers holding the gradients written by the last flush(es)
        self._flat_consumer = None  # weakref to the data-parallel reducer: without a live consumer the list must not grow
        self.join_streams = []      # side streams that ran part of the backward (split-batch step): flush() waits for them

    def track_flats(self):
        c = self._flat_consumer
        return c is not None and c() is not None

    def arm(self):
        if not self.armed:
            try:
                torch.autograd.Variable._execution_engine.queue_callback(self.flush)
                self.armed = True
            except RuntimeError:       # not inside a backward pass: the caller flushes explicitly
                pass

    def can_defer(self, param):
        return (self.enabled and _compute_dtype == torch.bfloat16 and isinstance(param, torch.Tensor) and param.is_leaf
                and param.requires_grad and param.dtype == torch.float32 and param.is_contiguous())

    def wgrad(self, param, dy, x, n_out, k_in, rows, bias_param=None):
        self.w.append((param, dy, x, n_out, k_in, rows, bias_param))
        self.arm()

    def colsum(self, param, x2d):
        self.b.append((param, x2d))
        self.arm()

    def _stage(self, key, arr, dev):
        """numpy bytes -> pinned staging -> fresh device tensor (async H2D).
        Eager: one persistent pinned buffer per table key, guarded by an event so that it is not rewritten before the
        previous step's copy has run. Under HIP-graph capture: a NEW pinned buffer per table, owned by this object for
        the life of the process and never written again -- the graph's H2D node re-reads it at every replay, so it
        must not be shared with later eager steps or with other captures."""
        raw = torch.from_numpy(arr)
        n = raw.numel()
        if torch.cuda.is_current_stream_capturing():
            pin = torch.empty(max(n, 1), dtype=torch.uint8).pin_memory()
            pin[:n].copy_(raw)
            self._capture_pins.append(pin)
            d = torch.empty(n, dtype=torch.uint8, device=dev)
            d.copy_(pin[:n], non_blocking=True)
            return d
        ent = self._pin.get(key)
        if ent is None or ent[0].numel() < n:
            ent = [torch.empty(max(n, 1 << 16), dtype=torch.uint8).pin_memory(), None]
            self._pin[key] = ent
        pin, ev = ent
        if ev is not None:
            ev.synchronize()              # the previous step's H2D copy out of this pinned buffer must have run
        pin[:n].copy_(raw)
        d = torch.empty(n, dtype=torch.uint8, device=dev)
        d.copy_(pin[:n], non_blocking=True)
        ent[1] = torch.cuda.Event()
        ent[1].record()
        return d

    @staticmethod
    def _target(param):
        """(grad tensor to write, accumulate flag)"""
        if param.grad is None:
            param.grad = torch.empty_like(param, memory_format=torch.contiguous_format)
            return param.grad, 0
        if param.grad.dtype != torch.float32 or not param.grad.is_contiguous():
            raise _lib.EvpError("deferred gradients need contiguous float32 .grad tensors")
        return param.grad, 1

    def flush(self):
        """End-of-backward callback: launch everything that was queued (normal mode). While `hold` is set (the step
        executor capturing the forward+backward graph of the data-parallel path) the queue is kept for build_plan()."""
        self.armed = False
        if self.hold:
            return
        for s_ in self.join_streams:
            torch.cuda.current_stream().wait_stream(s_)
        if self.flat_buffers and not self.track_flats():
            self.flat_buffers = []          # leftovers of a reducer that is gone
        for st in self._build(n_chunks=1, static=False):
            st.run()

    def build_plan(self, n_chunks=4):
        """Data-parallel graph mode: turn the queue left by a captured backward into a list of re-runnable launch steps
        with STATIC device tables (the operands live in the graph's private pool, so their addresses repeat at every
        replay). Weight-gradient problems are cut into `n_chunks` launches, each writing its own flat gradient buffer, so
        the all-reduce of one chunk overlaps the next chunk's GEMMs (parallel.OverlappedPlan)."""
        self.armed = False
        return self._build(n_chunks=n_chunks, static=True)

    class _Step:
        def __init__(self, entry, pt, it, n_items, flats, zero, keep, post=()):
            self.entry, self.pt, self.it, self.n_items = entry, pt, it, n_items
            self.flats = flats          # flat gradient buffers complete once this step has run
            self.zero = zero            # buffers that accumulate with atomics: cleared before every run
            self.keep = keep            # operand tensors referenced by the tables
            self.post = list(post)      # split-K reductions: (workspace, out, n_slices, numel, accumulate)
            self.round = 0              # steps of one round touch disjoint gradients (may run concurrently); rounds are ordered

        def run(self):
            for z in self.zero:
                z.zero_()
            call(self.entry, self.pt.data_ptr(), self.it.data_ptr(), self.n_items, stream_ptr())
            for ws, out, ns, numel, acc in self.post:
                call("evp_sum_slices_f32", ws.data_ptr(), out.data_ptr(), ns, numel, acc, stream_ptr())

    def _build(self, n_chunks, static):
        import numpy as np
        w, b, self.w, self.b = self.w, self.b, [], []
        steps = []

        def table(key, arr, dev):
            if static:
                return torch.from_numpy(arr.copy()).to(dev)
            return self._stage(key, arr, dev)

        def alloc_fresh(params, dev, zeroed):
            """one flat buffer for the parameters of this launch that have no gradient yet; returns (flat | None, ids)"""
            fresh = [p_ for p_ in dict.fromkeys(params) if p_.grad is None]
            if not fresh:
                return None, set()
            sizes = [(p_.numel() + 63) // 64 * 64 for p_ in fresh]
            flat = (torch.zeros if zeroed else torch.empty)(sum(sizes), dtype=torch.float32, device=dev)
            o = 0
            for p_, n_ in zip(fresh, sizes):
                p_.grad = flat[o:o + p_.numel()].view_as(p_)
                o += n_
            if static or self.track_flats():    # only a reducer (or a plan being built) takes these; otherwise the
                self.flat_buffers.append(flat)  # parameters' .grad views are the only owners and zero_grad frees them
            return flat, {id(p_) for p_ in fresh}

        if w:
            dev = w[0][1].device

            def big_(it):       # long-K problems with >= 256-wide outputs go to the 256x256 G4 kernel (K % 32 == 0, K >= 96)
                return _use_wgrad_g4 and it[5] % 32 == 0 and it[5] >= 96 and it[3] >= 256 and it[4] >= 256
            # a bias gradient rides on its Linear's weight-gradient problem only in the 256x256 kernel; otherwise it joins
            # the grouped column sums below
            for it in w:
                if it[6] is not None and not big_(it):
                    b.append((it[6], it[1]))
            # A parameter used by several autograd nodes of one backward (rec+con: masked AND dense forward) has several
            # queued contributions. Tiles of different problems run concurrently, so contributions to the SAME gradient
            # go to successive launches (round r holds every parameter's r-th contribution; normally one round).
            rounds, count = [], {}
            for item in w:
                r = count.get(id(item[0]), 0)
                count[id(item[0])] = r + 1
                while len(rounds) <= r:
                    rounds.append([])
                rounds[r].append(item)
            pdt = np.dtype([("A", "<u8"), ("B", "<u8"), ("C", "<u8"), ("M", "<i4"), ("N", "<i4"), ("K", "<i4"), ("lda", "<i4"),
                            ("ldb", "<i4"), ("ldc", "<i4"), ("acc", "<i4"), ("cacc", "<i4"), ("colsum", "<u8")])
            for r, batch in enumerate(rounds):
                # per round: the 256x256 ring kernel (in n_chunks launches when a plan is built) and the 128x128 kernel;
                # inside a launch the longest-K tiles are listed first
                big = sorted([it for it in batch if big_(it)], key=lambda it: -it[5])
                small = sorted([it for it in batch if not big_(it)], key=lambda it: -it[5])
                groups = []
                if big:
                    k = max(1, min(n_chunks if r == 0 else 1, len(big)))
                    work = np.cumsum([float(it[3]) * it[4] * it[5] for it in big])
                    cuts = [0] + [int(np.searchsorted(work, work[-1] * (c + 1) / k, side="left")) + 1 for c in range(k - 1)] + [len(big)]
                    cuts = sorted(set(min(max(c, 0), len(big)) for c in cuts))
                    for c in range(len(cuts) - 1):
                        if cuts[c + 1] > cuts[c]:
                            part_ = big[cuts[c]:cuts[c + 1]]
                            groups.append(("w256r%dc%d" % (r, c), part_, 256, "evp_gemm_grouped_tn_g4_bf16"))
                if small:
                    groups.append(("w128r%d" % r, small, 128, "evp_gemm_grouped_tn_bf16"))
                for tag, part, T_, entry in groups:
                    # A problem with few output tiles and a very long K (ConvViT stage 1: 256x256 outputs, K = B*56*56)
                    # would keep one workgroup busy for the whole launch: cut its K into slices that run as separate
                    # problems into a workspace and are summed afterwards. The bias then takes the column-sum launch.
                    def n_slices_(it):
                        tiles = ((it[3] + T_ - 1) // T_) * ((it[4] + T_ - 1) // T_)
                        return int(min(64, it[5] // 8192)) if (tiles <= 32 and it[5] >= 32768 and (it[3] * it[4]) % 4 == 0) else 1
                    for it in part:
                        if it[6] is not None and T_ == 256 and n_slices_(it) > 1:
                            b.append((it[6], it[1]))
                    flat, fresh_ids = alloc_fresh([it[0] for it in part] +
                                                  [it[6] for it in part if it[6] is not None and T_ == 256 and n_slices_(it) == 1],
                                                  dev, zeroed=False)
                    rows_, items, post = [], [], []
                    for (param, dy, x, n_out, k_in, rows, bias_param) in part:
                        gt, acc = self._target(param)
                        if id(param) in fresh_ids:
                            acc = 0                  # first write into the freshly allocated flat slice
                        ns = n_slices_((param, dy, x, n_out, k_in, rows, bias_param))
                        tm, tn = (n_out + T_ - 1) // T_, (k_in + T_ - 1) // T_
                        if ns > 1:
                            kper = ((rows // 64 + ns - 1) // ns) * 64
                            ns = (rows + kper - 1) // kper
                            numel = n_out * k_in
                            ws = torch.empty(ns * numel, dtype=torch.float32, device=dev)
                            for sidx in range(ns):
                                k0 = sidx * kper
                                kk = min(kper, rows - k0)
                                rows_.append((dy.data_ptr() + k0 * n_out * 2, x.data_ptr() + k0 * k_in * 2, ws.data_ptr() + sidx * numel * 4,
                                              n_out, k_in, kk, n_out, k_in, k_in, 0, 0, 0, tm, tn))
                            post.append((ws, gt, ns, numel, acc))
                            continue
                        cs_ptr, cs_acc = 0, 0
                        if bias_param is not None and T_ == 256:
                            bt, cs_acc = self._target(bias_param)
                            if id(bias_param) in fresh_ids:
                                cs_acc = 0
                            cs_ptr = bt.data_ptr()
                        rows_.append((dy.data_ptr(), x.data_ptr(), gt.data_ptr(), n_out, k_in, rows, n_out, k_in, k_in, acc, cs_acc, cs_ptr, tm, tn))
                    rows_.sort(key=lambda r_: -r_[5])           # longest K first
                    probs = np.zeros(len(rows_), dtype=pdt)
                    weights = []
                    for i, r_ in enumerate(rows_):
                        probs[i] = r_[:12]
                        tm, tn = r_[12], r_[13]
                        t = np.zeros((tn, tm, 4), dtype=np.int32)
                        t[..., 0] = i
                        t[..., 1] = np.arange(tm, dtype=np.int32)[None, :]
                        t[..., 2] = np.arange(tn, dtype=np.int32)[:, None]
                        if _wgrad_xcd_order and T_ == 256:
                            t = _supertile_major(t, 2, 4)
                        items.append(t.reshape(-1, 4))
                        weights.append(np.full(tm * tn, float(r_[5]), dtype=np.float64))
                    items = np.concatenate(items, 0)
                    if _wgrad_xcd_order and T_ == 256:
                        items = _deal_to_xcds(items, np.concatenate(weights))
                    pt = table(tag + "p", probs.view(np.uint8), dev)
                    it_ = table(tag + "i", items.view(np.uint8).reshape(-1), dev)
                    steps.append(self._Step(entry, pt, it_, int(items.shape[0]), [flat] if flat is not None else [], [],
                                            [(it[1], it[2]) for it in part], post))
                    steps[-1].round = r
        if b:
            dev = b[0][1].device
            probs = np.zeros(len(b), dtype=np.dtype([("x", "<u8"), ("out", "<u8"), ("M", "<i8"), ("N", "<i4"), ("ld", "<i4"),
                                                      ("dtype", "<i4"), ("pad", "<i4")]))
            items = []
            # fresh bias gradients are slices of ONE flat buffer zeroed by one memset (they accumulate with atomics)
            flat, _ = alloc_fresh([t_[0] for t_ in b], dev, zeroed=True)
            for i, (param, x2d) in enumerate(b):
                M, N = x2d.shape
                gt, acc = self._target(param)
                probs[i] = (x2d.data_ptr(), gt.data_ptr(), M, N, x2d.stride(0), dt(x2d), 0)
                cb, rs = (N + 127) // 128, (M + 255) // 256
                t = np.zeros((rs, cb, 4), dtype=np.int32)
                t[..., 0] = i
                t[..., 1] = np.arange(cb, dtype=np.int32)[None, :]
                t[..., 2] = np.arange(rs, dtype=np.int32)[:, None]
                items.append(t.reshape(-1, 4))
            items = np.concatenate(items, 0)
            pt = table("bp", probs.view(np.uint8), dev)
            it = table("bi", items.view(np.uint8).reshape(-1), dev)
            st = self._Step("evp_colsum_grouped", pt, it, int(items.shape[0]), [flat] if flat is not None else [],
                            [flat] if (static and flat is not None) else [], [t_[1] for t_ in b])
            # in a plan the short column-sum launch goes first: its all-reduce then hides under the first GEMM chunk
            if static:
                steps.insert(0, st)
            else:
                steps.append(st)
        return steps


def _supertile_major(t, sm, sn):
    """[tn, tm, 4] tile grid -> the same tiles listed super-tile by super-tile (sn x sm tiles each, tile_m fastest inside):
    the 8 tiles of a 2 x 4 super-tile read 2 A panels and 4 B panels between them instead of 16."""
    tn, tm = t.shape[0], t.shape[1]
    out = []
    for n0 in range(0, tn, sn):
        for m0 in range(0, tm, sm):
            out.append(t[n0:n0 + sn, m0:m0 + sm].reshape(-1, 4))
    return np.concatenate(out, 0)


def _deal_to_xcds(items, work, n_xcd=8):
    """Workgroups b and b + 8 run on the same XCD (one L2 each; MI355X_MICROARCH.md, workgroup dispatch) and an XCD's CUs
    take its workgroups in launch order. Cut the item sequence (super-tile major, longest K first) into 8 CONTIGUOUS runs of
    equal estimated work (tiles x K) and interleave them, items[8 k + x] = run_x[k], so that the ~32 tiles an XCD runs at any
    moment are neighbouring super-tiles of one problem -- they advance through K in lockstep and find each other's panels in
    their L2 instead of re-reading them through the Infinity Cache (2.9x over-fetch measured in round 1). Short runs are
    padded with prob = -1 items, which the kernels skip. Placement is speed only: any assignment gives the same result."""
    n = items.shape[0]
    if n < 4 * n_xcd:
        return items
    cum = np.cumsum(work)
    cuts = [0] + [int(np.searchsorted(cum, cum[-1] * (x + 1) / n_xcd, side="left")) + 1 for x in range(n_xcd - 1)] + [n]
    cuts = [min(max(c, 0), n) for c in cuts]
    for i in range(1, len(cuts)):
        cuts[i] = max(cuts[i], cuts[i - 1])
    runs = [items[cuts[x]:cuts[x + 1]] for x in range(n_xcd)]
    longest = max(r.shape[0] for r in runs)
    out = np.full((longest, n_xcd, 4), -1, dtype=np.int32)
    for x, r in enumerate(runs):
        out[:r.shape[0], x] = r
    return out.reshape(-1, 4)


_wgrad_xcd_order = os.environ.get("EVP_WGRAD_XCD", "1") != "0"
_use_wgrad_g4 = os.environ.get("EVP_WGRAD_G4", "1") != "0"


def set_wgrad_g4(flag):
    """A/B switch: long-K weight gradients with >= 256-wide outputs on the 256x256 G4 body (one wave per SIMD, 32x32x16; default)
    or on 128x128 tiles like the rest."""
    global _use_wgrad_g4
    _use_wgrad_g4 = bool(flag)



def set_wgrad_xcd_order(flag):
    """A/B switch: XCD-aware order of the grouped 256x256 weight-gradient items (default) or the plain problem-major list."""
    global _wgrad_xcd_order
    _wgrad_xcd_order = bool(flag)


_deferred = _DeferredGrads()
def set_deferred_grads(flag):
    """A/B switch: grouped end-of-backward weight/bias gradients (default) or one GEMM / column sum per layer."""
    _deferred.enabled = bool(flag)


def flush_deferred_grads():
    if _deferred.w or _deferred.b:
        _deferred.flush()


def hold_deferred_grads(flag):
    """While set, the end-of-backward callback leaves the queued gradient work in place (for build_deferred_plan)."""
    _deferred.hold = bool(flag)


def build_deferred_plan(n_chunks=4):
    return _deferred.build_plan(n_chunks)


def track_deferred_flat_buffers(consumer):
    """Called by the data-parallel reducer with itself: while that object is alive the flat gradient buffers of each flush
    are kept for take_deferred_flat_buffers(). Without a live consumer nothing is retained (the list would otherwise pin one
    set of gradient buffers per step). Pass None to switch tracking off."""
    import weakref
    _deferred._flat_consumer = weakref.ref(consumer) if consumer is not None else None
    if consumer is None:
        _deferred.flat_buffers = []


def take_deferred_flat_buffers():
    """Flat f32 buffers that hold the deferred gradients written since the last call (their parameters' .grad are
    views into them): the data-parallel reducer all-reduces these in place."""
    out, _deferred.flat_buffers = _deferred.flat_buffers, []
    return out


def _wgrad(dy, x, n_out, k_in, rows, param=None, shape=None, bias_param=None):
    """dW[n_out,k_in] = dy^T x (f32). Returns the gradient, or None when it was queued for the grouped launch
    (then flush() delivers it into param.grad). `bias_param`: the same Linear's bias, whose gradient sum_rows(dy) is
    then produced by the same grouped kernel (or the grouped column sums) -- only pass it when this call is queued
    (see _wgrad_bias)."""
    if param is not None and dy.dtype == torch.bfloat16 and n_out % 8 == 0 and k_in % 8 == 0 and _deferred.can_defer(param):
        _deferred.wgrad(param, dy, x, n_out, k_in, rows, bias_param)
        return None
    dw = torch.empty(n_out, k_in, dtype=torch.float32, device=dy.device)
    gemm(dy, x, dw, M=n_out, N=k_in, K=rows, trans_a=True, trans_b=True, lda=n_out, ldb=k_in)
    return dw if shape is None else dw.view(shape)


def _wgrad_bias(dy, x, n_out, k_in, rows, wparam, bparam, need_w, need_b, dy_f32=None, shape=None, dy_colsum=None):
    """(dW, db) of one Linear. When both can be deferred (bf16 mode, leaf parameters) the bias gradient is attached to the
    weight-gradient problem: the 256x256 grouped kernel sums dy's columns from the A fragments it holds anyway, so dy is
    not read a second time. Otherwise falls back to the separate paths (`dy_f32`: f32 version of dy for the exact sum)."""
    # only where dy exists in bf16 alone (qkv, fc1): proj / fc2 have the f32 residual-stream gradient, whose exact column
    # sum is kept (summing its bf16 copy moved those bias gradients by ~5e-3 of their max)
    fuse = (need_w and need_b and dy_f32 is None and dy.dtype == torch.bfloat16 and n_out % 8 == 0 and k_in % 8 == 0 and
            _deferred.can_defer(wparam) and _deferred.can_defer(bparam))
    if fuse:
        _deferred.wgrad(wparam, dy, x, n_out, k_in, rows, bparam)
        return None, None
    dw = _wgrad(dy, x, n_out, k_in, rows, wparam, shape) if need_w else None
    if need_b and dy_colsum is not None and _deferred.can_defer(bparam):
        _deferred.colsum(bparam, dy_colsum)       # per-block column sums left by the LayerNorm backward: a few rows to add
        return dw, None
    db = _bgrad(dy_f32 if dy_f32 is not None else dy, bparam) if need_b else None
    return dw, db


def _bgrad(x2d, param=None):
    """db[n] = sum_m x2d[m,n] (f32); None when queued (see _wgrad)."""
    if param is not None and x2d.shape[1] % 8 == 0 and _deferred.can_defer(param):
        _deferred.colsum(param, x2d)
        return None
    return colsum(x2d)



# ----------------------------------------------------------------------------------------------------- stochastic regularisers
class BlockDrop:
    """Stochastic regularisers of ONE call of a residual block (reference vit_block.py:241,252-253 / conv_block.py:35,43-49 /
    swin_block.py:257,270-271; timm `drop_path`, nn.Dropout): `u1`, `u2` = float32 [x.shape[0]] uniform draws for the two
    DropPath applications (None = off), keep_prob = 1 - drop_path; `drop` = elementwise dropout rate of the branch outputs and the
    MLP hidden (proj_drop / Mlp.drop), its masks drawn by evp_dropout_fwd from (seed, a per-use offset). `seed` is an int (tests:
    a given stream) or a 1-element int64 DEVICE tensor (draw_drop_seed): the kernel reads it at run time, so a replayed HIP graph --
    whose kernel arguments are frozen at capture -- draws new masks on every replay. A block called with rd=None takes the fused
    fast path (residual add inside the GEMM epilogue)."""

    def __init__(self, u1=None, u2=None, keep_prob=1.0, drop=0.0, seed=0, masks=None, attn_drop=0.0):
        self.u = (u1, u2)
        self.keep_prob = float(keep_prob)
        self.drop = float(drop)
        self.attn_drop = float(attn_drop)           # dropout on the attention probabilities (vit_block.py:127,138; swin_block.py:113,152)
        self.seed_dev = seed if torch.is_tensor(seed) else None
        self.seed = 0 if torch.is_tensor(seed) else int(seed)
        self.masks = masks            # optional explicit uint8 masks {"proj": .., "hidden": .., "fc2": ..} (tests: given-mask parity)
        self._n = 0

    def next_offset(self, numel):
        o = self._n
        self._n += (numel + 3) // 4
        return o


def dropout_fwd(x, rd, key):
    """-> (x * keep / (1 - p), mask uint8) for a contiguous tensor; the mask is rd.masks[key] when given."""
    out = torch.empty_like(x)
    n = x.numel()
    if rd.masks is not None and key in rd.masks:
        mask = _chk(rd.masks[key]).view(-1)
        call("evp_dropout_apply", ptr(_chk(x)), dt(x), ptr(mask), ptr(out), n, 1.0 / (1.0 - rd.drop), stream_ptr())
        return out, mask
    mask = torch.empty(n, dtype=torch.uint8, device=x.device)
    call("evp_dropout_fwd", ptr(_chk(x)), dt(x), ptr(out), ptr(mask), n, rd.drop, rd.seed, ptr(rd.seed_dev), rd.next_offset(n), stream_ptr())
    return out, mask


def dropout_bwd(g, mask, p):
    out = torch.empty_like(g)
    call("evp_dropout_apply", ptr(_chk(g)), dt(g), ptr(mask), ptr(out), g.numel(), 1.0 / (1.0 - p), stream_ptr())
    return out


def rows_scale(x, u, keep_prob, rows_per_sample, res=None, want_out=True, want_lp=False):
    """(res + s_b * x, bf16(s_b * x)) with s_b = floor(keep_prob + u[b]) / keep_prob per sample b (evp_rows_scale_f32)."""
    M, D = x.shape
    out = torch.empty(M, D, dtype=torch.float32, device=x.device) if want_out else None
    lp = torch.empty(M, D, dtype=torch.bfloat16, device=x.device) if want_lp else None
    call("evp_rows_scale_f32", ptr(_chk(x, torch.float32)), ptr(u), float(keep_prob), ptr(res), M, D, int(rows_per_sample), ptr(out), ptr(lp),
         stream_ptr())
    return out, lp


def _branch_residual(a, w, bias, res, M, N, K, rd, which, rps, key):
    """Residual-stream output of one branch: res + drop_path(dropout(a . w^T + bias)). rd None: fused in the GEMM epilogue."""
    out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    if rd is None:
        gemm(a, w, out, M=M, N=N, K=K, bias=bias, residual=res)
        return out, None
    gemm(a, w, out, M=M, N=N, K=K, bias=bias)
    mk = None
    if rd.drop > 0:
        out, mk = dropout_fwd(out, rd, key)
    y, _ = rows_scale(out, rd.u[which], rd.keep_prob, rps, res=res)
    return y, mk


def _branch_grad(g, rd, which, rps, mk, T):
    """Gradient entering a branch whose output was added as res + drop_path(dropout(.)): (f32, compute-dtype copy)."""
    bf = T == torch.bfloat16
    gs, gs_lp = rows_scale(g, rd.u[which], rd.keep_prob, rps, want_lp=bf and mk is None)
    if mk is not None:
        gs = dropout_bwd(gs, mk, rd.drop)
        gs_lp = cast(gs, T) if bf else None
    return gs, (gs_lp if bf else gs)


# ----------------------------------------------------------------------------------------------------- autograd
class ViTBlockFn(torch.autograd.Function):
    """Pre-LN transformer block (model/sub_module/vit_block.py:246-254) on a [B,N,D] f32 residual stream."""

    @staticmethod
    def forward(ctx, x, n1w, n1b, qkvw, qkvb, pw, pb, n2w, n2b, f1w, f1b, f2w, f2b, heads, eps, want_attn, rd=None):
        B, N, D = x.shape
        M = B * N
        dh = D // heads
        T = _compute_dtype
        x2d = _chk(x.detach()).view(M, D)
        dev = x.device
        wq, wp, w1, w2 = lp_weight(qkvw), lp_weight(pw), lp_weight(f1w), lp_weight(f2w)
        ln1, mean1, rstd1 = layernorm_fwd(x2d, n1w, n1b, eps, T)
        qkv = torch.empty(M, 3 * D, dtype=T, device=dev)
        gemm(ln1, wq, qkv, M=M, N=3 * D, K=D, bias=qkvb)
        a_drop = rd is not None and rd.attn_drop > 0
        fused = fused_attention_ok(T, N, dh) and not a_drop
        ad_saved = None
        if a_drop:           # probabilities materialised, dropped, multiplied with V (the returned map is the dropped one, vit_block.py:138-143)
            stat, probs, mk_a, att = attention_dropout_fwd(qkv, B, N, heads, dh, rd)
            ad_saved = (probs, mk_a)
        elif fused:
            att, stat, probs = attention_fused_fwd(qkv, B, N, heads, dh, want_probs=want_attn)   # stat = log-sum-exp
        else:
            probs, att = attention_fwd(qkv, B, N, heads, dh)
            stat = probs
        x1, mk_p = _branch_residual(att, wp, pb, x2d, M, D, D, rd, 0, N, "proj")
        ln2, mean2, rstd2 = layernorm_fwd(x1, n2w, n2b, eps, T)
        Hd = f1w.shape[0]
        h_pre = torch.empty(M, Hd, dtype=T, device=dev)
        h_act = torch.empty(M, Hd, dtype=T, device=dev)
        gemm(ln2, w1, h_act, M=M, N=Hd, K=D, bias=f1b, act=ACT_GELU, aux=h_pre)
        mk_h = None
        if rd is not None and rd.drop > 0:
            h_act, mk_h = dropout_fwd(h_act, rd, "hidden")
        x2, mk_2 = _branch_residual(h_act, w2, f2b, x1, M, D, Hd, rd, 1, N, "fc2")
        ctx.save_for_backward(x2d, n1w, n2w, mean1, rstd1, ln1, qkv, stat, att, x1, mean2, rstd2, ln2, h_pre, h_act,
                              wq, wp, w1, w2)
        ctx.dims = (B, N, D, heads, dh, Hd)
        ctx.fused = fused
        ctx.attn_drop = ad_saved
        ctx.rd, ctx.drop_masks = rd, (mk_p, mk_h, mk_2)
        ctx.prm = (qkvw, qkvb, pw, pb, f1w, f1b, f2w, f2b)     # leaf parameters: targets of the deferred gradients
        ctx.nprm = (n1w, n1b, n2w, n2b)
        out = x2.view(B, N, D)
        if want_attn:
            attn = probs[..., :N]
            ctx.mark_non_differentiable(attn)
            return out, attn
        return out

    @staticmethod
    def backward(ctx, g2, *_):
        (x2d, n1w, n2w, mean1, rstd1, ln1, qkv, stat, att, x1, mean2, rstd2, ln2, h_pre, h_act, wq, wp, w1, w2) = \
            ctx.saved_tensors
        B, N, D, heads, dh, Hd = ctx.dims
        M = B * N
        T = qkv.dtype
        dev = g2.device
        bf = T == torch.bfloat16
        g2 = _chk(g2.contiguous(), torch.float32).view(M, D)
        rd = ctx.rd
        mk_p, mk_h, mk_2 = ctx.drop_masks
        # bf16 copy and column-sum partials of the incoming gradient, if the LayerNorm backward that made it left them
        g2_side_lp, g2_cs = _side_take(g2) if bf else (None, None)
        if rd is None:
            g2b, g2_lp = g2, (g2_side_lp if g2_side_lp is not None else cast(g2, T))
        else:                           # the MLP branch sees s_b * g (DropPath) times the fc2-output dropout mask
            (g2b, g2_lp), g2_cs = _branch_grad(g2, rd, 1, N, mk_2, T), None
        qkvw_, qkvb_, pw_, pb_, f1w_, f1b_, f2w_, f2b_ = ctx.prm
        need = ctx.needs_input_grad
        # MLP
        dw2, db2 = _wgrad_bias(g2_lp, h_act, D, Hd, M, f2w_, f2b_, need[11], need[12], dy_f32=g2b, dy_colsum=g2_cs)
        dh_pre = torch.empty(M, Hd, dtype=T, device=dev)
        gemm(g2_lp, w2, dh_pre, M=M, N=Hd, K=D, trans_b=True, ldb=Hd, act=ACT_DGELU, aux=h_pre)
        if mk_h is not None:            # hidden dropout sits between GELU and fc2: its mask commutes with the GELU' product
            dh_pre = dropout_bwd(dh_pre, mk_h, rd.drop)
        dw1, db1 = _wgrad_bias(dh_pre, ln2, Hd, D, M, f1w_, f1b_, need[9], need[10])
        dln2 = torch.empty(M, D, dtype=T, device=dev)
        gemm(dh_pre, w1, dln2, M=M, N=D, K=Hd, trans_b=True, ldb=D)
        side = bf and rd is None
        g1, g1_lp, dn2w, dn2b = layernorm_bwd(dln2, x1, n2w, mean2, rstd2, gres=g2, want_lp=side, params=ctx.nprm[2:], side=side)
        g1_side_lp, g1_cs = _side_take(g1) if side else (None, None)
        g1b = g1
        if rd is not None:
            g1b, g1_lp = _branch_grad(g1, rd, 0, N, mk_p, T)
        elif not bf:
            g1_lp = g1
        # attention
        dwp, dbp = _wgrad_bias(g1_lp, att, D, D, M, pw_, pb_, need[5], need[6], dy_f32=g1b, dy_colsum=g1_cs)
        datt = torch.empty(M, D, dtype=T, device=dev)
        gemm(g1_lp, wp, datt, M=M, N=D, K=D, trans_b=True, ldb=D)
        if ctx.attn_drop is not None:
            dqkv = attention_dropout_bwd(qkv, stat, ctx.attn_drop[0], ctx.attn_drop[1], datt, B, N, heads, dh, rd.attn_drop)
        elif ctx.fused:
            dqkv = attention_fused_bwd(qkv, att, datt, stat, B, N, heads, dh)
        else:
            dqkv = attention_bwd(qkv, stat, datt, B, N, heads, dh)
        dwq, dbq = _wgrad_bias(dqkv, ln1, 3 * D, D, M, qkvw_, qkvb_, need[3], need[4])
        dln1 = torch.empty(M, D, dtype=T, device=dev)
        gemm(dqkv, wq, dln1, M=M, N=D, K=3 * D, trans_b=True, ldb=D)
        # the block below this one takes g0 as ITS incoming gradient: leave it the bf16 copy and the column sums
        g0, _, dn1w, dn1b = layernorm_bwd(dln1, x2d, n1w, mean1, rstd1, gres=g1, want_lp=bf, params=ctx.nprm[:2], side=bf)
        return (g0.view(B, N, D), dn1w, dn1b, dwq, dbq, dwp, dbp, dn2w, dn2b, dw1, db1, dw2, db2, None, None, None, None)


class PatchEmbedFn(torch.autograd.Function):
    """Conv2d(k=s=p) + LayerNorm(eps 1e-5) + GELU (vit_block.py:60-68), + pos_embed, keeping only ids_keep tokens
    (vit.py:110-115; gathering first is equivalent because every step is per token)."""

    @staticmethod
    def forward(ctx, x, ids_keep, w, b, gamma, beta, pos, patch):
        B, Cc, H, W = x.shape
        L = (H // patch) * (W // patch)
        n_keep = L if ids_keep is None else ids_keep.shape[1]
        D = w.shape[0]
        Kc = Cc * patch * patch
        M = B * n_keep
        T = _compute_dtype
        dev = x.device
        cols = torch.empty(M, Kc, dtype=T, device=dev)
        call("evp_patchify", ptr(_chk(x.detach(), torch.float32)), ptr(ids_keep), B, Cc, H, W, patch, n_keep, ptr(cols),
             dt(cols), stream_ptr())
        wl = lp_weight(w).view(D, Kc)
        y = torch.empty(M, D, dtype=torch.float32, device=dev)
        gemm(cols, wl, y, M=M, N=D, K=Kc, bias=b)
        out = torch.empty(M, D, dtype=torch.float32, device=dev)
        mean = torch.empty(M, dtype=torch.float32, device=dev)
        rstd = torch.empty(M, dtype=torch.float32, device=dev)
        pos2d = None if pos is None else _chk(pos.detach().view(L, D), torch.float32)
        call("evp_embed_post_fwd", ptr(y), ptr(gamma), ptr(beta), ptr(pos2d), ptr(ids_keep), B, n_keep, L, D, 1e-5,
             ptr(out), ptr(mean), ptr(rstd), stream_ptr())
        ctx.save_for_backward(cols, y, gamma, beta, mean, rstd)
        ctx.wshape = tuple(w.shape)
        ctx.prm = (w, b)
        ctx.nprm = (gamma, beta)        # leaf parameters: targets of the deferred (grouped) column sums
        return out.view(B, n_keep, D)

    @staticmethod
    def backward(ctx, g):
        cols, y, gamma, beta, mean, rstd = ctx.saved_tensors
        M, D = y.shape
        Kc = cols.shape[1]
        dev = g.device
        g = _chk(g.contiguous(), torch.float32).view(M, D)
        dy = torch.empty(M, D, dtype=cols.dtype, device=dev)
        nb = call("evp_layernorm_bwd_nblk", M)
        ws = torch.empty(nb, 2 * D, dtype=torch.float32, device=dev)
        # the per-block partials of d gamma / d beta join the step's grouped column-sum launch when the parameters can take deferred
        # gradients (as layernorm_bwd does); otherwise the kernel's own finalize pass reduces them
        defer = D % 8 == 0 and _deferred.can_defer(ctx.nprm[0]) and _deferred.can_defer(ctx.nprm[1])
        dgamma = dbeta = None
        if not defer:
            dgamma = torch.empty(D, dtype=torch.float32, device=dev)
            dbeta = torch.empty(D, dtype=torch.float32, device=dev)
        call("evp_embed_post_bwd", ptr(g), ptr(y), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), M, D, ptr(dy), dt(dy),
             ptr(dgamma), ptr(dbeta), ptr(ws), stream_ptr())
        if defer:
            _deferred.colsum(ctx.nprm[0], ws[:, :D])
            _deferred.colsum(ctx.nprm[1], ws[:, D:])
        dw = _wgrad(dy, cols, D, Kc, M, ctx.prm[0], ctx.wshape) if ctx.needs_input_grad[2] else None
        db = _bgrad(dy, ctx.prm[1]) if ctx.needs_input_grad[3] else None
        return None, None, dw, db, dgamma, dbeta, None, None


class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over the last dim of f32 tokens; up to three inputs are summed first (vit.py:125-128)."""

    @staticmethod
    def forward(ctx, x, x2, x3, gamma, beta, eps):
        shp = x.shape
        D = shp[-1]
        xs = [None if t is None else _chk(t.detach().contiguous(), torch.float32).view(-1, D) for t in (x, x2, x3)]
        y, mean, rstd = layernorm_fwd(xs[0], gamma, beta, eps, torch.float32, xs[1], xs[2])
        ctx.save_for_backward(gamma, mean, rstd, *[t for t in xs if t is not None])
        ctx.prm = (gamma, beta)
        ctx.n_in = sum(t is not None for t in xs)
        ctx.has = (x2 is not None, x3 is not None)
        return y.view(shp)

    @staticmethod
    def backward(ctx, g):
        gamma, mean, rstd, *xs = ctx.saved_tensors
        D = gamma.shape[0]
        x = xs[0]
        rest = xs[1:]
        x2 = rest.pop(0) if ctx.has[0] else None
        x3 = rest.pop(0) if ctx.has[1] else None
        g2d = _chk(g.contiguous(), torch.float32).view(-1, D)
        bf = _compute_dtype == torch.bfloat16
        dx, _, dgamma, dbeta = layernorm_bwd(g2d, x, gamma, mean, rstd, x2=x2, x3=x3, params=ctx.prm, want_lp=bf, side=bf)
        dx = dx.view(g.shape)
        return dx, (dx if ctx.has[0] else None), (dx if ctx.has[1] else None), dgamma, dbeta, None


class LinearFn(torch.autograd.Function):
    """nn.Linear on f32 tokens (f32 in / f32 out; operands rounded to the compute dtype for the MFMA). An output width
    that is not a multiple of 8 (a classification head with 10 or 101 classes) runs zero-padded to the next multiple --
    the kernels' vector accesses want 16-byte-aligned rows -- and is sliced back."""

    @staticmethod
    def forward(ctx, x, w, b):
        shp = x.shape
        K = shp[-1]
        N = w.shape[0]
        Np = _pad8(N)
        x2d = _chk(x.detach().contiguous(), torch.float32).view(-1, K)
        M = x2d.shape[0]
        xl = cast(x2d, _compute_dtype)
        wl = lp_weight(w)
        bias = b
        if Np != N:
            wp = torch.zeros(Np, K, dtype=wl.dtype, device=wl.device)
            wp[:N].copy_(wl)
            wl = wp
            if b is not None:
                bias = torch.zeros(Np, dtype=torch.float32, device=wl.device)
                bias[:N].copy_(b.detach())
        y = torch.empty(M, Np, dtype=torch.float32, device=x.device)
        gemm(xl, wl, y, M=M, N=Np, K=K, bias=bias)
        ctx.save_for_backward(xl, wl)
        ctx.has_bias = b is not None
        ctx.needs_dx = x.requires_grad
        ctx.prm = (w, b)
        ctx.n_out = N
        if Np != N:
            y = y[:, :N]
        return y.reshape(*shp[:-1], N)

    @staticmethod
    def backward(ctx, g):
        xl, wl = ctx.saved_tensors
        M, K = xl.shape
        N, Np = ctx.n_out, wl.shape[0]
        g2d = _chk(g.contiguous(), torch.float32).view(M, N)
        if Np != N:
            gp = torch.zeros(M, Np, dtype=torch.float32, device=g.device)
            gp[:, :N].copy_(g2d)
            g2d = gp
        gl = cast(g2d, xl.dtype)
        dw = db = None
        if Np == N:
            dw = _wgrad(gl, xl, N, K, M, ctx.prm[0]) if ctx.needs_input_grad[1] else None
            db = _bgrad(g2d, ctx.prm[1]) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        else:
            if ctx.needs_input_grad[1]:
                dw = _wgrad(gl, xl, Np, K, M)[:N].contiguous()
            if ctx.has_bias and ctx.needs_input_grad[2]:
                db = colsum(g2d)[:N].contiguous()
        dx = None
        if ctx.needs_dx:
            dx = torch.empty(M, K, dtype=torch.float32, device=g.device)
            gemm(gl, wl, dx, M=M, N=K, K=Np, trans_b=True, ldb=K)
            dx = dx.view(*g.shape[:-1], K)
        return dx, dw, db


class UnshuffleFn(torch.autograd.Function):
    """Decoder input assembly (pr_rec_decoder.py:56-62): mask tokens, un-shuffle by ids_restore, + pos_embed."""

    @staticmethod
    def forward(ctx, emb, mask_token, pos, ids_restore):
        B, n_keep, D = emb.shape
        L = ids_restore.shape[1]
        out = torch.empty(B, L, D, dtype=torch.float32, device=emb.device)
        call("evp_unshuffle_fwd", ptr(_chk(emb.detach().contiguous(), torch.float32)), ptr(_chk(mask_token.detach())),
             ptr(_chk(pos.detach())), ptr(_chk(ids_restore)), B, n_keep, L, D, ptr(out), stream_ptr())
        ctx.save_for_backward(ids_restore)
        ctx.dims = (B, n_keep, L, D)
        ctx.mt_shape = tuple(mask_token.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        (ids_restore,) = ctx.saved_tensors
        B, n_keep, L, D = ctx.dims
        g = _chk(g.contiguous(), torch.float32)
        demb = torch.empty(B, n_keep, D, dtype=torch.float32, device=g.device)
        dmt = torch.empty(D, dtype=torch.float32, device=g.device)
        nb = (B * L + 63) // 64
        ws = torch.empty(nb * D, dtype=torch.float32, device=g.device)
        call("evp_unshuffle_bwd", ptr(g), ptr(ids_restore), B, n_keep, L, D, ptr(demb), ptr(dmt), ptr(ws), stream_ptr())
        return demb, dmt.view(ctx.mt_shape), None, None


class RecLossFn(torch.autograd.Function):
    """PrHubModel.reconstruct_loss (pr_hub_model.py:125-141) fused with frame2emb; returns a 0-dim f32 tensor."""

    @staticmethod
    def forward(ctx, pred, target, mask, patch, norm_pix):
        B, L, P = pred.shape
        _, Cc, H, W = target.shape
        dev = pred.device
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        dpred = torch.empty(B, L, P, dtype=torch.float32, device=dev)
        ws = torch.empty(2 * B * L + 4, dtype=torch.float32, device=dev)
        call("evp_rec_loss", ptr(_chk(pred.detach().contiguous(), torch.float32)),
             ptr(_chk(target.detach().contiguous(), torch.float32)), ptr(mask), B, Cc, H, W, patch, int(bool(norm_pix)),
             ptr(loss), ptr(dpred), ptr(ws), stream_ptr())
        ctx.save_for_backward(dpred)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        # upstream gradient is a device scalar (1, or 1/accum_iter): scale in place without a host sync
        call("evp_scale_f32", ptr(dpred), ptr(_chk(g.contiguous().view(1), torch.float32)), dpred.numel(), stream_ptr())
        return dpred, None, None, None, None


def vit_block(x, blk, heads, eps, want_attn=False, rd=None):
    return ViTBlockFn.apply(x, blk.norm1.weight, blk.norm1.bias, blk.attn.qkv.weight, blk.attn.qkv.bias,
                            blk.attn.proj.weight, blk.attn.proj.bias, blk.norm2.weight, blk.norm2.bias,
                            blk.mlp.fc1.weight, blk.mlp.fc1.bias, blk.mlp.fc2.weight, blk.mlp.fc2.bias,
                            heads, eps, want_attn, rd)


# ----------------------------------------------------------------------------------------------------- contrastive stage
def _pad8(n):
    return (n + 7) // 8 * 8


class LinearBNFn(torch.autograd.Function):
    """Linear(no bias) -> BatchNorm over the B*L token rows (training statistics) -> optional ReLU: one stage of the
    MoCo-v3 heads (mlp_head.py:4-24 applied as pr_hub_model.py:223-237). Hidden stages keep the compute dtype, the
    last stage (affine=False, no ReLU) returns f32 tokens. Running statistics are updated in place."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, running_mean, running_var, eps, momentum, relu, last):
        B, L, K = x.shape
        R, N = B * L, w.shape[0]
        T = _compute_dtype
        dev = x.device
        xl = cast(_chk(x.detach().contiguous()).view(R, K), T)
        wl = lp_weight(w)
        ydt = torch.float32 if last else T
        y_pre = torch.empty(R, N, dtype=ydt, device=dev)
        gemm(xl, wl, y_pre, M=R, N=N, K=K)
        y = torch.empty(R, N, dtype=ydt, device=dev)
        mean = torch.empty(N, dtype=torch.float32, device=dev)
        invstd = torch.empty(N, dtype=torch.float32, device=dev)
        nb = call("evp_batchnorm_nblk", R)
        ws = torch.empty((2 * nb + 2) * N, dtype=torch.float32, device=dev)
        call("evp_batchnorm_fwd", ptr(y_pre), dt(y_pre), R, N, ptr(gamma), ptr(beta), float(eps), float(momentum), int(relu),
             ptr(y), ptr(mean), ptr(invstd), ptr(running_mean), ptr(running_var), ptr(ws), stream_ptr())
        ctx.save_for_backward(xl, wl, y_pre, y, mean, invstd, gamma)
        ctx.cfg = (B, L, K, N, bool(relu), x.dtype, x.requires_grad)
        ctx.prm = (w,)
        return y.view(B, L, N)

    @staticmethod
    def backward(ctx, g):
        xl, wl, y_pre, y, mean, invstd, gamma = ctx.saved_tensors
        B, L, K, N, relu, xdt, need_dx = ctx.cfg
        R = B * L
        dev = g.device
        g2d = cast(_chk(g.contiguous()).view(R, N), y.dtype)
        dpre = torch.empty(R, N, dtype=y.dtype, device=dev)
        nb = call("evp_batchnorm_nblk", R)
        ws = torch.empty((2 * nb + 2) * N, dtype=torch.float32, device=dev)
        affine = gamma is not None
        dgamma = torch.empty(N, dtype=torch.float32, device=dev) if affine else None
        dbeta = torch.empty(N, dtype=torch.float32, device=dev) if affine else None
        call("evp_batchnorm_bwd", ptr(g2d), ptr(y_pre), ptr(y), dt(y), R, N, ptr(gamma), ptr(mean), ptr(invstd), int(relu),
             ptr(dpre), ptr(dgamma), ptr(dbeta), ptr(ws), stream_ptr())
        dl = cast(dpre, xl.dtype)
        dw = _wgrad(dl, xl, N, K, R, ctx.prm[0]) if ctx.needs_input_grad[1] else None
        dx = None
        if need_dx:
            dx = torch.empty(R, K, dtype=xdt if xdt == torch.float32 else xl.dtype, device=dev)
            gemm(dl, wl, dx, M=R, N=K, K=N, trans_b=True, ldb=K)
            dx = dx.view(B, L, K)
        return dx, dw, dgamma, dbeta, None, None, None, None, None, None


def linear_bn_tokens(x, lin, bn, relu):
    """One [Linear, BatchNorm2d(, ReLU)] stage of an nn.Sequential head on (B, L, C) tokens."""
    if bn is None:
        return LinearFn.apply(x, lin.weight, None)
    if not bn.training:
        raise NotImplementedError("the pre-training heads only run in training mode (batch statistics)")
    last = not relu
    out = LinearBNFn.apply(x, lin.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps,
                           0.1 if bn.momentum is None else bn.momentum, relu, last)
    if bn.num_batches_tracked is not None:
        bn.num_batches_tracked += 1
    return out


class L2NormFn(torch.autograd.Function):
    """F.normalize(x, dim=-1) on f32 tokens."""

    @staticmethod
    def forward(ctx, x):
        C_ = x.shape[-1]
        x2 = _chk(x.detach().contiguous(), torch.float32).view(-1, C_)
        y = torch.empty_like(x2)
        nrm = torch.empty(x2.shape[0], dtype=torch.float32, device=x.device)
        call("evp_l2norm_rows_fwd", ptr(x2), x2.shape[0], C_, ptr(y), ptr(nrm), stream_ptr())
        ctx.save_for_backward(y, nrm)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, g):
        y, nrm = ctx.saved_tensors
        g2 = _chk(g.contiguous(), torch.float32).view(y.shape)
        dx = torch.empty_like(y)
        call("evp_l2norm_rows_bwd", ptr(g2), ptr(y), ptr(nrm), y.shape[0], y.shape[1], ptr(dx), stream_ptr())
        return dx.view(g.shape)


class InfoNCEQueueFn(torch.autograd.Function):
    """Per-position InfoNCE of normalised q against the positive key and the (C,L,K) queue
    (pr_hub_model.py:150-160). q, k: normalised f32 (B,L,C)."""

    @staticmethod
    def forward(ctx, q, k, queue, T_):
        B, L, C_ = q.shape
        K = queue.shape[2]
        Kp = _pad8(K)
        R = B * L
        dev = q.device
        Tc = _compute_dtype
        q2, k2 = _chk(q.detach().contiguous(), torch.float32).view(R, C_), _chk(k.detach().contiguous(), torch.float32).view(R, C_)
        pos = torch.empty(R, dtype=torch.float32, device=dev)
        call("evp_rowdot_f32", ptr(q2), ptr(k2), R, C_, ptr(pos), stream_ptr())
        qw = queue.detach() if Kp == K else torch.nn.functional.pad(queue.detach(), (0, Kp - K))   # tiny test queues only
        ql, qul = cast(q2, Tc), cast(qw.contiguous(), Tc)
        if qul.data_ptr() == queue.data_ptr():
            # f32 mode with K % 8 == 0: cast()/contiguous() returned the live buffer, which _dequeue_and_enqueue overwrites
            # in place before backward reads it (the reference uses self.queue.clone().detach(), pr_hub_model.py:152)
            qul = qul.clone()
        neg = torch.empty(B, L, Kp, dtype=torch.float32, device=dev)
        # neg[b,l,:] = q[b,l,:] . queue[:,l,:]   ('blc,clk->blk'), batched over l
        gemm(ql, qul, neg, M=B, N=K, K=C_, trans_b=True, lda=L * C_, ldb=L * Kp, ldc=L * Kp, batch=(L, 1),
             stride_a=(C_, 0), stride_b=(Kp, 0), stride_c=(Kp, 0))
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        dpos = torch.empty(R, dtype=torch.float32, device=dev)
        dneg = torch.empty(B, L, Kp, dtype=torch.float32, device=dev)
        ws = torch.empty(R, dtype=torch.float32, device=dev)
        call("evp_infonce_queue", ptr(pos), ptr(neg), R, K, Kp, 1.0 / T_, ptr(loss), ptr(dpos), ptr(dneg), ptr(ws), stream_ptr())
        ctx.save_for_backward(q2, k2, qul, dpos, dneg)
        ctx.cfg = (B, L, C_, K, Kp)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        q2, k2, qul, dpos, dneg = ctx.saved_tensors
        B, L, C_, K, Kp = ctx.cfg
        R = B * L
        dev = g.device
        dq_pos = torch.empty(R, C_, dtype=torch.float32, device=dev)
        call("evp_scale_rows_f32", ptr(k2), ptr(dpos), None, R, C_, ptr(dq_pos), stream_ptr())
        dk = torch.empty(R, C_, dtype=torch.float32, device=dev)
        call("evp_scale_rows_f32", ptr(q2), ptr(dpos), None, R, C_, ptr(dk), stream_ptr())
        dq = torch.empty(R, C_, dtype=torch.float32, device=dev)
        dnl = cast(dneg, qul.dtype)
        # dq[b,l,:] = dpos*k + dneg[b,l,:] . queue[:,l,:]^T    ('blk,clk->blc')
        gemm(dnl, qul, dq, M=B, N=C_, K=K, lda=L * Kp, ldb=L * Kp, ldc=L * C_, batch=(L, 1), stride_a=(Kp, 0),
             stride_b=(Kp, 0), stride_c=(C_, 0), residual=dq_pos)
        gs = _chk(g.contiguous().view(1), torch.float32)
        call("evp_scale_f32", ptr(dq), ptr(gs), dq.numel(), stream_ptr())
        call("evp_scale_f32", ptr(dk), ptr(gs), dk.numel(), stream_ptr())
        return dq.view(B, L, C_), dk.view(B, L, C_), None, None


class InfoNCEInBatchFn(torch.autograd.Function):
    """MoCo-v3 in-batch InfoNCE (pr_hub_model.py:170-188): logits[n,l,m] = q[n,l,:].k_all[m,l,:] / T, label of row
    (n,l) = n + N*rank. k_all carries no gradient when it was all-gathered (as in the reference)."""

    @staticmethod
    def forward(ctx, q, k_all, T_, rank, k_has_grad):
        N, L, C_ = q.shape
        Mk = k_all.shape[0]
        Mp = _pad8(Mk)
        dev = q.device
        Tc = _compute_dtype
        q2 = _chk(q.detach().contiguous(), torch.float32)
        ka = _chk(k_all.detach().contiguous(), torch.float32)
        ql, kl = cast(q2, Tc), cast(ka, Tc)
        logits = torch.zeros(N, L, Mp, dtype=torch.float32, device=dev)
        gemm(ql, kl, logits, M=N, N=Mk, K=C_, lda=L * C_, ldb=L * C_, ldc=L * Mp, batch=(L, 1), stride_a=(C_, 0),
             stride_b=(C_, 0), stride_c=(Mp, 0), alpha=1.0 / T_)
        labels = (torch.arange(N, device=dev, dtype=torch.int64) + N * rank).unsqueeze(1).expand(N, L).contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        dlog = torch.empty(N, L, Mp, dtype=torch.float32, device=dev)
        ws = torch.empty(N * L, dtype=torch.float32, device=dev)
        call("evp_cross_entropy", ptr(logits), ptr(labels), N * L, Mk, Mp, ptr(loss), ptr(dlog), ptr(ws), stream_ptr())
        ctx.save_for_backward(ql, kl, dlog)
        ctx.cfg = (N, L, C_, Mk, Mp, T_, bool(k_has_grad))
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        ql, kl, dlog = ctx.saved_tensors
        N, L, C_, Mk, Mp, T_, k_has_grad = ctx.cfg
        dev = g.device
        dl = cast(dlog, ql.dtype)
        gs = _chk(g.contiguous().view(1), torch.float32)
        dq = torch.empty(N, L, C_, dtype=torch.float32, device=dev)
        gemm(dl, kl, dq, M=N, N=C_, K=Mk, trans_b=True, lda=L * Mp, ldb=L * C_, ldc=L * C_, batch=(L, 1), stride_a=(Mp, 0),
             stride_b=(C_, 0), stride_c=(C_, 0), alpha=1.0 / T_)
        call("evp_scale_f32", ptr(dq), ptr(gs), dq.numel(), stream_ptr())
        dk = None
        if k_has_grad:
            dk = torch.empty(Mk, L, C_, dtype=torch.float32, device=dev)
            gemm(dl, ql, dk, M=Mk, N=C_, K=N, trans_a=True, trans_b=True, lda=L * Mp, ldb=L * C_, ldc=L * C_, batch=(L, 1),
                 stride_a=(Mp, 0), stride_b=(C_, 0), stride_c=(C_, 0), alpha=1.0 / T_)
            call("evp_scale_f32", ptr(dk), ptr(gs), dk.numel(), stream_ptr())
        return dq, dk, None, None, None


def info_nce_queue(emb_h, clip_emb, queue, T_):
    """-> (loss, normalised keys (detached) for the enqueue)."""
    q = L2NormFn.apply(emb_h)
    k = L2NormFn.apply(clip_emb)
    return InfoNCEQueueFn.apply(q, k, queue, T_), k.detach()


def info_nce_inbatch(emb_h, clip_emb, T_, distributed=False, gather=None):
    """`gather` (engine.ForwardCollectives.gather): takes the all-gather's place under a captured data-parallel step."""
    q = L2NormFn.apply(emb_h)
    k = L2NormFn.apply(clip_emb)
    rank = 0
    if distributed:
        import torch.distributed as dist
        from .model.pretrain.pr_hub_model import concat_all_gather
        rank = dist.get_rank()
        # RCCL all-gather over xGMI; no gradient (reference pr_hub_model.py:176-182,248-259)
        k_all = concat_all_gather(k) if gather is None else gather(k.detach())
        return InfoNCEInBatchFn.apply(q, k_all, T_, rank, False)
    return InfoNCEInBatchFn.apply(q, k, T_, rank, True)


def enqueue_keys(queue, keys, ptr_):
    B, L, C_ = keys.shape
    call("evp_enqueue_keys", ptr(_chk(queue, torch.float32)), ptr(_chk(keys.contiguous(), torch.float32)), int(ptr_), B, L, C_,
         queue.shape[2], stream_ptr())


def enqueue_keys_dev(queue, keys, queue_ptr):
    """_dequeue_and_enqueue with the pointer kept on the device (queue_ptr: int64 [1] buffer): no host sync."""
    B, L, C_ = keys.shape
    call("evp_enqueue_keys_dev", ptr(_chk(queue, torch.float32)), ptr(_chk(keys.contiguous(), torch.float32)),
         ptr(_chk(queue_ptr, torch.int64)), B, L, C_, queue.shape[2], stream_ptr())


# ----------------------------------------------------------------------------------------------------- ConvViT stages
class PatchEmbedNHWCFn(torch.autograd.Function):
    """PatchEmbed (Conv2d k=s=p -> LayerNorm(eps 1e-5) -> GELU, vit_block.py:60-68) on a channels-last token map
    x (B, H*W, C) f32, optionally restricted to ids_keep and with a positional table: the conv is a GEMM on the
    gathered patch matrix (convvit.py:22-25,141,153)."""

    @staticmethod
    def forward(ctx, x, ids_keep, w, b, gamma, beta, pos, patch, H, W):
        B = x.shape[0]
        Cin = x.shape[-1]
        D = w.shape[0]
        L = (H // patch) * (W // patch)
        n_keep = L if ids_keep is None else ids_keep.shape[1]
        Kc = Cin * patch * patch
        M = B * n_keep
        T = _compute_dtype
        dev = x.device
        xin = _chk(x.detach().contiguous(), torch.float32)
        cols = torch.empty(M, Kc, dtype=T, device=dev)
        call("evp_patchify_nhwc", ptr(xin), ptr(ids_keep), B, H, W, Cin, patch, n_keep, ptr(cols), dt(cols), stream_ptr())
        wl = lp_weight(w).view(D, Kc)
        y = torch.empty(M, D, dtype=torch.float32, device=dev)
        gemm(cols, wl, y, M=M, N=D, K=Kc, bias=b)
        out = torch.empty(M, D, dtype=torch.float32, device=dev)
        mean = torch.empty(M, dtype=torch.float32, device=dev)
        rstd = torch.empty(M, dtype=torch.float32, device=dev)
        pos2d = None if pos is None else _chk(pos.detach().view(L, D), torch.float32)
        call("evp_embed_post_fwd", ptr(y), ptr(gamma), ptr(beta), ptr(pos2d), ptr(ids_keep), B, n_keep, L, D, 1e-5, ptr(out),
             ptr(mean), ptr(rstd), stream_ptr())
        ctx.save_for_backward(cols, y, gamma, beta, mean, rstd, wl, ids_keep)
        ctx.cfg = (B, H, W, Cin, D, patch, n_keep, Kc, tuple(w.shape), x.requires_grad)
        ctx.prm = (w, b)
        return out.view(B, n_keep, D)

    @staticmethod
    def backward(ctx, g):
        cols, y, gamma, beta, mean, rstd, wl, ids_keep = ctx.saved_tensors
        B, H, W, Cin, D, patch, n_keep, Kc, wshape, need_dx = ctx.cfg
        M = B * n_keep
        dev = g.device
        g = _chk(g.contiguous(), torch.float32).view(M, D)
        dy = torch.empty(M, D, dtype=cols.dtype, device=dev)
        dgamma = torch.empty(D, dtype=torch.float32, device=dev)
        dbeta = torch.empty(D, dtype=torch.float32, device=dev)
        nb = call("evp_layernorm_bwd_nblk", M)
        ws = torch.empty(2 * nb * D, dtype=torch.float32, device=dev)
        call("evp_embed_post_bwd", ptr(g), ptr(y), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), M, D, ptr(dy), dt(dy), ptr(dgamma),
             ptr(dbeta), ptr(ws), stream_ptr())
        dw = _wgrad(dy, cols, D, Kc, M, ctx.prm[0], wshape) if ctx.needs_input_grad[2] else None
        db = _bgrad(dy, ctx.prm[1]) if ctx.needs_input_grad[3] else None
        dx = None
        if need_dx:
            dcols = torch.empty(M, Kc, dtype=cols.dtype, device=dev)
            gemm(dy, wl, dcols, M=M, N=Kc, K=D, trans_b=True, ldb=Kc)
            dx = torch.empty(B, H * W, Cin, dtype=torch.float32, device=dev)
            call("evp_unpatchify_nhwc", ptr(dcols), dt(dcols), ptr(ids_keep), B, H, W, Cin, patch, n_keep, 0, ptr(dx), stream_ptr())
        return dx, None, dw, db, dgamma, dbeta, None, None, None, None


class StridedConvTokensFn(torch.autograd.Function):
    """Conv2d(k=s=p) + bias on a channels-last map, evaluated only at the tokens in ids_keep (the multi-scale fusion
    convs stage1/2_output_decode followed by the gather, convvit.py:137-140,149-151)."""

    @staticmethod
    def forward(ctx, x, ids_keep, w, b, patch, H, W):
        B, _, Cin = x.shape
        D = w.shape[0]
        L = (H // patch) * (W // patch)
        n_keep = L if ids_keep is None else ids_keep.shape[1]
        Kc = Cin * patch * patch
        M = B * n_keep
        T = _compute_dtype
        dev = x.device
        cols = torch.empty(M, Kc, dtype=T, device=dev)
        call("evp_patchify_nhwc", ptr(_chk(x.detach().contiguous(), torch.float32)), ptr(ids_keep), B, H, W, Cin, patch, n_keep,
             ptr(cols), dt(cols), stream_ptr())
        wl = lp_weight(w).view(D, Kc)
        y = torch.empty(M, D, dtype=torch.float32, device=dev)
        gemm(cols, wl, y, M=M, N=D, K=Kc, bias=b)
        ctx.save_for_backward(cols, wl, ids_keep)
        ctx.cfg = (B, H, W, Cin, D, patch, n_keep, Kc, tuple(w.shape))
        ctx.prm = (w, b)
        return y.view(B, n_keep, D)

    @staticmethod
    def backward(ctx, g):
        cols, wl, ids_keep = ctx.saved_tensors
        B, H, W, Cin, D, patch, n_keep, Kc, wshape = ctx.cfg
        M = B * n_keep
        dev = g.device
        g2 = _chk(g.contiguous(), torch.float32).view(M, D)
        gl = cast(g2, cols.dtype)
        dw = _wgrad(gl, cols, D, Kc, M, ctx.prm[0], wshape) if ctx.needs_input_grad[2] else None
        db = _bgrad(g2, ctx.prm[1]) if ctx.needs_input_grad[3] else None
        dcols = torch.empty(M, Kc, dtype=cols.dtype, device=dev)
        gemm(gl, wl, dcols, M=M, N=Kc, K=D, trans_b=True, ldb=Kc)
        dx = torch.empty(B, H * W, Cin, dtype=torch.float32, device=dev)
        call("evp_unpatchify_nhwc", ptr(dcols), dt(dcols), ptr(ids_keep), B, H, W, Cin, patch, n_keep, 0, ptr(dx), stream_ptr())
        return dx, None, dw, db, None, None, None


class AddPosGatherFn(torch.autograd.Function):
    """x[b,j,:] + pos[ids[b,j],:] (pos is a frozen table)."""

    @staticmethod
    def forward(ctx, x, pos, ids):
        B, n, D = x.shape
        L = pos.shape[-2]
        out = torch.empty(B, n, D, dtype=torch.float32, device=x.device)
        call("evp_add_rows_gather_f32", ptr(_chk(x.detach().contiguous(), torch.float32)), ptr(_chk(pos.detach().view(L, D))),
             ptr(ids), B, n, L, D, ptr(out), stream_ptr())
        return out

    @staticmethod
    def backward(ctx, g):
        return g, None, None


class ConvBlockFn(torch.autograd.Function):
    """ConvMAE-style block on a channels-last token map (conv_block.py:41-51):
    x = x + conv2(dw5x5(keep * conv1(LN(x)))); x = x + fc2(GELU(fc1(LN(x)))), 1x1 convs as GEMMs."""

    @staticmethod
    def forward(ctx, x, n1w, n1b, c1w, c1b, aw, ab, c2w, c2b, n2w, n2b, f1w, f1b, f2w, f2b, mask, mask_scale, H, W, rd=None):
        B, HW, Cc = x.shape
        M = B * HW
        T = _compute_dtype
        dev = x.device
        x2d = _chk(x.detach().contiguous(), torch.float32).view(M, Cc)
        w1, w2, wf1, wf2 = (lp_weight(c1w).view(Cc, Cc), lp_weight(c2w).view(Cc, Cc), lp_weight(f1w).view(-1, Cc),
                            lp_weight(f2w).view(Cc, -1))
        Hd = wf1.shape[0]
        ln1, mean1, rstd1 = layernorm_fwd(x2d, n1w, n1b, 1e-5, T)
        c1 = torch.empty(M, Cc, dtype=T, device=dev)
        gemm(ln1, w1, c1, M=M, N=Cc, K=Cc, bias=c1b)
        a = torch.empty(M, Cc, dtype=T, device=dev)
        awf = _chk(aw.detach().contiguous(), torch.float32)
        call("evp_dwconv5x5_fwd", ptr(c1), dt(c1), ptr(mask), int(mask_scale), ptr(awf), ptr(ab), B, H, W, Cc, ptr(a), stream_ptr())
        # (CMlp.drop sits after GELU and after fc2, conv_block.py:19-21; the conv branch has no dropout of its own)
        x1, _ = _branch_residual(a, w2, c2b, x2d, M, Cc, Cc, rd if (rd is None or rd.drop == 0) else BlockDrop(rd.u[0], rd.u[1], rd.keep_prob), 0, HW, "conv2")
        ln2, mean2, rstd2 = layernorm_fwd(x1, n2w, n2b, 1e-5, T)
        h_pre = torch.empty(M, Hd, dtype=T, device=dev)
        h_act = torch.empty(M, Hd, dtype=T, device=dev)
        gemm(ln2, wf1, h_act, M=M, N=Hd, K=Cc, bias=f1b, act=ACT_GELU, aux=h_pre)
        mk_h = None
        if rd is not None and rd.drop > 0:
            h_act, mk_h = dropout_fwd(h_act, rd, "hidden")
        x2, mk_2 = _branch_residual(h_act, wf2, f2b, x1, M, Cc, Hd, rd, 1, HW, "fc2")
        ctx.rd, ctx.drop_masks = rd, (mk_h, mk_2)
        ctx.save_for_backward(x2d, n1w, n2w, mean1, rstd1, ln1, c1, a, x1, mean2, rstd2, ln2, h_pre, h_act, w1, w2, wf1, wf2, awf, mask)
        ctx.cfg = (B, H, W, Cc, Hd, int(mask_scale), tuple(aw.shape))
        ctx.prm = (c1w, c1b, c2w, c2b, f1w, f1b, f2w, f2b)
        ctx.nprm = (n1w, n1b, n2w, n2b)
        return x2.view(B, HW, Cc)

    @staticmethod
    def backward(ctx, g2):
        (x2d, n1w, n2w, mean1, rstd1, ln1, c1, a, x1, mean2, rstd2, ln2, h_pre, h_act, w1, w2, wf1, wf2, awf, mask) = ctx.saved_tensors
        B, H, W, Cc, Hd, mask_scale, awshape = ctx.cfg
        c1w_, c1b_, c2w_, c2b_, f1w_, f1b_, f2w_, f2b_ = ctx.prm
        need = ctx.needs_input_grad
        M = B * H * W
        T = c1.dtype
        bf = T == torch.bfloat16
        dev = g2.device
        g2 = _chk(g2.contiguous(), torch.float32).view(M, Cc)
        # as in ViTBlockFn: bf16 copy / column sums left by the LayerNorm backward of the block above, bias gradients on the
        # weight-gradient launch, LayerNorm dgamma / dbeta through the grouped column sums
        g2_side_lp, g2_cs = _side_take(g2) if bf else (None, None)
        rd = ctx.rd
        mk_h, mk_2 = ctx.drop_masks
        HW = H * W
        if rd is None:
            g2b, g2_lp = g2, (g2_side_lp if g2_side_lp is not None else cast(g2, T))
        else:
            (g2b, g2_lp), g2_cs = _branch_grad(g2, rd, 1, HW, mk_2, T), None
        dwf2, db2 = _wgrad_bias(g2_lp, h_act, Cc, Hd, M, f2w_, f2b_, need[13], need[14], dy_f32=g2b, shape=tuple(f2w_.shape), dy_colsum=g2_cs)
        dh_pre = torch.empty(M, Hd, dtype=T, device=dev)
        gemm(g2_lp, wf2, dh_pre, M=M, N=Hd, K=Cc, trans_b=True, ldb=Hd, act=ACT_DGELU, aux=h_pre)
        if mk_h is not None:
            dh_pre = dropout_bwd(dh_pre, mk_h, rd.drop)
        dwf1, db1 = _wgrad_bias(dh_pre, ln2, Hd, Cc, M, f1w_, f1b_, need[11], need[12], shape=tuple(f1w_.shape))
        dln2 = torch.empty(M, Cc, dtype=T, device=dev)
        gemm(dh_pre, wf1, dln2, M=M, N=Cc, K=Hd, trans_b=True, ldb=Cc)
        side = bf and rd is None
        g1, g1_lp, dn2w, dn2b = layernorm_bwd(dln2, x1, n2w, mean2, rstd2, gres=g2, want_lp=side, params=ctx.nprm[2:], side=side)
        g1_side_lp, g1_cs = _side_take(g1) if side else (None, None)
        g1b = g1
        if rd is not None:
            g1b, g1_lp = _branch_grad(g1, rd, 0, HW, None, T)
        elif not bf:
            g1_lp = g1
        # conv branch
        dwc2, dbc2 = _wgrad_bias(g1_lp, a, Cc, Cc, M, c2w_, c2b_, need[7], need[8], dy_f32=g1b, shape=tuple(c2w_.shape), dy_colsum=g1_cs)
        da = torch.empty(M, Cc, dtype=T, device=dev)
        gemm(g1_lp, w2, da, M=M, N=Cc, K=Cc, trans_b=True, ldb=Cc)
        dc1 = torch.empty(M, Cc, dtype=T, device=dev)
        daw = torch.empty(Cc, 25, dtype=torch.float32, device=dev)
        dab = torch.empty(Cc, dtype=torch.float32, device=dev)
        ns = call("evp_dwconv5x5_bwd_nslab", B, H, W)
        ws = torch.empty(ns * 26 * Cc, dtype=torch.float32, device=dev)
        call("evp_dwconv5x5_bwd", ptr(da), ptr(c1), dt(c1), ptr(mask), mask_scale, ptr(awf), B, H, W, Cc, ptr(dc1), ptr(daw), ptr(dab),
             ptr(ws), stream_ptr())
        dwc1, dbc1 = _wgrad_bias(dc1, ln1, Cc, Cc, M, c1w_, c1b_, need[3], need[4], shape=tuple(c1w_.shape))
        dln1 = torch.empty(M, Cc, dtype=T, device=dev)
        gemm(dc1, w1, dln1, M=M, N=Cc, K=Cc, trans_b=True, ldb=Cc)
        g0, _, dn1w, dn1b = layernorm_bwd(dln1, x2d, n1w, mean1, rstd1, gres=g1, want_lp=bf, params=ctx.nprm[:2], side=bf)
        return (g0.view(B, H * W, Cc), dn1w, dn1b, dwc1, dbc1, daw.view(awshape), dab, dwc2, dbc2, dn2w, dn2b, dwf1, db1, dwf2, db2,
                None, None, None, None, None)


def conv_block(x, blk, H, W, mask=None, mask_scale=1, rd=None):
    return ConvBlockFn.apply(x, blk.norm1.weight, blk.norm1.bias, blk.conv1.weight, blk.conv1.bias, blk.attn.weight, blk.attn.bias,
                             blk.conv2.weight, blk.conv2.bias, blk.norm2.weight, blk.norm2.bias, blk.mlp.fc1.weight,
                             blk.mlp.fc1.bias, blk.mlp.fc2.weight, blk.mlp.fc2.bias, mask, mask_scale, H, W, rd)


# ----------------------------------------------------------------------------------------------------- Swin (row a15)
def gather_rows(x, idx, n_in=None):
    """out[b,s,:] = idx[s] >= 0 ? x[b,idx[s],:] : 0 for f32 tokens [B,n_in,C]; idx int32 [n_out] shared by the batch."""
    B, n, C = x.shape
    n_out = idx.shape[0]
    out = torch.empty(B, n_out, C, dtype=torch.float32, device=x.device)
    call("evp_gather_rows_f32", ptr(_chk(x, torch.float32)), ptr(_chk(idx, torch.int32)), ptr(out), B, n, n_out, C, 0, stream_ptr())
    return out


class GatherRowsFn(torch.autograd.Function):
    """GroupingModule.group / merge and PatchMerging's regrouping (swin_block.py:454-466,193-201) as one row gather.
    `idx_bwd` is the index of the adjoint gather: the real entries of `idx_fwd` form a permutation, padding slots
    (which the reference fills with a copy of token 0 and later drops) receive and send no gradient."""

    @staticmethod
    def forward(ctx, x, idx_fwd, idx_bwd):
        ctx.save_for_backward(idx_bwd)
        return gather_rows(x.detach().contiguous(), idx_fwd)

    @staticmethod
    def backward(ctx, g):
        (idx_bwd,) = ctx.saved_tensors
        return gather_rows(g.contiguous(), idx_bwd), None, None


class PatchProjFn(torch.autograd.Function):
    """Conv2d(k=s=patch) of an NCHW f32 image evaluated only at the tokens `ids_keep` (int64 [B,n_keep]), returned as
    f32 tokens [B,n_keep,D] (swin_block.py:58-62 before its norm; gathering first is equivalent, every step is per token)."""

    @staticmethod
    def forward(ctx, x, ids_keep, w, b, patch):
        B, Cc, H, W = x.shape
        L = (H // patch) * (W // patch)
        n_keep = L if ids_keep is None else ids_keep.shape[1]
        D = w.shape[0]
        Kc = Cc * patch * patch
        M = B * n_keep
        cols = torch.empty(M, Kc, dtype=_compute_dtype, device=x.device)
        call("evp_patchify", ptr(_chk(x.detach(), torch.float32)), ptr(ids_keep), B, Cc, H, W, patch, n_keep, ptr(cols),
             dt(cols), stream_ptr())
        y = torch.empty(M, D, dtype=torch.float32, device=x.device)
        gemm(cols, lp_weight(w).view(D, Kc), y, M=M, N=D, K=Kc, bias=b)
        ctx.save_for_backward(cols)
        ctx.wshape = tuple(w.shape)
        ctx.prm = (w, b)
        return y.view(B, n_keep, D)

    @staticmethod
    def backward(ctx, g):
        (cols,) = ctx.saved_tensors
        M, Kc = cols.shape
        D = ctx.wshape[0]
        g2d = _chk(g.contiguous(), torch.float32).view(M, D)
        gl = cast(g2d, cols.dtype)
        dw = _wgrad(gl, cols, D, Kc, M, ctx.prm[0], ctx.wshape) if ctx.needs_input_grad[2] else None
        db = _bgrad(g2d, ctx.prm[1]) if ctx.needs_input_grad[3] else None
        return None, None, dw, db, None


class SwinFuseConvFn(torch.autograd.Function):
    """swin.py:201-208: scatter the visible stage tokens into a zero R x R grid, Conv2d(k, stride k) down to the
    decoder grid, gather each sample's ids_keep cells -- formed directly as the [B*K, C*k*k] patch rows of the kept
    cells times weight.view(out, C*k*k)."""

    @staticmethod
    def forward(ctx, x, w, b, tokmap, coords, ids_keep, ids_restore, R, k):
        B, n, C = x.shape
        K = ids_keep.shape[1]
        Dout = w.shape[0]
        Kc = C * k * k
        M = B * K
        A = torch.empty(M, Kc, dtype=torch.float32, device=x.device)
        call("evp_swin_fuse_gather_f32", ptr(_chk(x.detach().contiguous(), torch.float32)), ptr(_chk(tokmap, torch.int32)),
             ptr(_chk(ids_keep, torch.int64)), ptr(A), B, n, K, C, R, k, stream_ptr())
        Al = cast(A, _compute_dtype)
        wl = lp_weight(w).view(Dout, Kc)
        y = torch.empty(M, Dout, dtype=torch.float32, device=x.device)
        gemm(Al, wl, y, M=M, N=Dout, K=Kc, bias=b)
        ctx.save_for_backward(Al, wl, coords, ids_restore)
        ctx.dims = (B, n, K, C, R, k, Dout, Kc)
        ctx.wshape = tuple(w.shape)
        ctx.prm = (w, b)
        return y.view(B, K, Dout)

    @staticmethod
    def backward(ctx, g):
        Al, wl, coords, ids_restore = ctx.saved_tensors
        B, n, K, C, R, k, Dout, Kc = ctx.dims
        M = B * K
        g2d = _chk(g.contiguous(), torch.float32).view(M, Dout)
        gl = cast(g2d, Al.dtype)
        dw = _wgrad(gl, Al, Dout, Kc, M, ctx.prm[0], ctx.wshape) if ctx.needs_input_grad[1] else None
        db = _bgrad(g2d, ctx.prm[1]) if ctx.needs_input_grad[2] else None
        dA = torch.empty(M, Kc, dtype=torch.float32, device=g.device)
        gemm(gl, wl, dA, M=M, N=Kc, K=Dout, trans_b=True, ldb=Kc)
        dx = torch.empty(B, n, C, dtype=torch.float32, device=g.device)
        call("evp_swin_fuse_gather_bwd_f32", ptr(dA), ptr(coords), ptr(ids_restore), ptr(dx), B, n, K, C, R, k, stream_ptr())
        return dx, dw, db, None, None, None, None, None, None


class SwinBlockFn(torch.autograd.Function):
    """SwinTransformerBlock.forward (swin_block.py:260-273) on grouped tokens [Bg, N, D] (Bg = batch * n_groups):
    pre-LN, window attention with the gathered relative-position bias and the -100 group mask (:135-158), MLP."""

    @staticmethod
    def forward(ctx, x, table, rel, n1w, n1b, qkvw, qkvb, pw, pb, n2w, n2b, f1w, f1b, f2w, f2b, heads, eps, want_attn, rd=None):
        Bg, N, D = x.shape
        nG = rel.shape[0]
        M = Bg * N
        dh = D // heads
        if dh != 32:
            raise _lib.EvpError(f"window attention kernel is built for d_h=32 (got {dh})")
        T = _compute_dtype
        dev = x.device
        x2d = _chk(x.detach().contiguous(), torch.float32).view(M, D)
        wq, wp, w1, w2 = lp_weight(qkvw), lp_weight(pw), lp_weight(f1w), lp_weight(f2w)
        ln1, mean1, rstd1 = layernorm_fwd(x2d, n1w, n1b, eps, T)
        qkv = torch.empty(M, 3 * D, dtype=T, device=dev)
        gemm(ln1, wq, qkv, M=M, N=3 * D, K=D, bias=qkvb)
        att = torch.empty(M, D, dtype=T, device=dev)
        probs = torch.empty(Bg, heads, N, N, dtype=torch.float32, device=dev) if want_attn else None
        tab = _chk(table.detach(), torch.float32)
        R = tab.shape[0]
        scale = dh ** -0.5
        # bf16 mode: the MFMA kernels of the ViT attention with the gathered bias as an additive matrix per (group, head); the
        # f32 LDS kernels stay the parity path and serve the blocks that return their probabilities
        a_drop = rd is not None and rd.attn_drop > 0
        fused = _use_window_mfma and T == torch.bfloat16 and not want_attn and N <= 128 and not a_drop
        lse = addm = addmT = None
        keep, keep_scale = None, 1.0
        if a_drop:           # swin_block.py:113,152: the LDS kernels hold the probabilities and take the keep flags; the MFMA form never does
            keep, keep_scale = _attn_keep_mask(rd, (Bg, heads, N, N), dev), 1.0 / (1.0 - rd.attn_drop)
        if fused:
            NP = call("evp_window_attention_fused_np", N)
            addm = torch.empty(nG * heads * NP * NP, dtype=torch.float32, device=dev)
            addmT = torch.empty_like(addm)
            call("evp_window_bias_build", ptr(tab), ptr(_chk(rel, torch.int32)), nG, N, heads, R, ptr(addm), ptr(addmT), stream_ptr())
            lse = torch.empty(Bg * heads * N, dtype=torch.float32, device=dev)
            call("evp_window_attention_fused_fwd", ptr(qkv), ptr(addm), Bg, nG, N, heads, scale, ptr(att), ptr(lse), stream_ptr())
        else:
            call("evp_window_attention_fwd", ptr(qkv), ptr(tab), ptr(_chk(rel, torch.int32)), ptr(att), ptr(probs), Bg, nG, N, heads, R,
                 scale, dt(qkv), ptr(keep), keep_scale, stream_ptr())
        x1, mk_p = _branch_residual(att, wp, pb, x2d, M, D, D, rd, 0, N, "proj")
        ln2, mean2, rstd2 = layernorm_fwd(x1, n2w, n2b, eps, T)
        Hd = f1w.shape[0]
        h_pre = torch.empty(M, Hd, dtype=T, device=dev)
        h_act = torch.empty(M, Hd, dtype=T, device=dev)
        gemm(ln2, w1, h_act, M=M, N=Hd, K=D, bias=f1b, act=ACT_GELU, aux=h_pre)
        mk_h = None
        if rd is not None and rd.drop > 0:
            h_act, mk_h = dropout_fwd(h_act, rd, "hidden")
        x2, mk_2 = _branch_residual(h_act, w2, f2b, x1, M, D, Hd, rd, 1, N, "fc2")
        ctx.rd, ctx.drop_masks = rd, (mk_p, mk_h, mk_2)
        ctx.save_for_backward(x2d, n1w, n2w, mean1, rstd1, ln1, qkv, att, x1, mean2, rstd2, ln2, h_pre, h_act, wq, wp, w1, w2,
                              tab, rel)
        ctx.dims = (Bg, nG, N, D, heads, dh, Hd, R, scale)
        ctx.prm = (qkvw, qkvb, pw, pb, f1w, f1b, f2w, f2b)
        ctx.nprm = (n1w, n1b, n2w, n2b)
        ctx.win = (lse, addm, addmT)          # plain tensors of this node (not inputs / outputs): kept on ctx
        ctx.keep = (keep, keep_scale)
        out = x2.view(Bg, N, D)
        if want_attn:
            ctx.mark_non_differentiable(probs)
            return out, probs
        return out

    @staticmethod
    def backward(ctx, g2, *_):
        (x2d, n1w, n2w, mean1, rstd1, ln1, qkv, att, x1, mean2, rstd2, ln2, h_pre, h_act, wq, wp, w1, w2, tab, rel) = \
            ctx.saved_tensors
        Bg, nG, N, D, heads, dh, Hd, R, scale = ctx.dims
        M = Bg * N
        T = qkv.dtype
        dev = g2.device
        bf = T == torch.bfloat16
        g2 = _chk(g2.contiguous(), torch.float32).view(M, D)
        # as in ViTBlockFn: the LayerNorm backward that produced g2 (the block above, when no regrouping gather sits in between:
        # the single-group stages) may have left its bf16 copy and column sums; bias gradients ride on the weight-gradient launch
        g2_side_lp, g2_cs = _side_take(g2) if bf else (None, None)
        rd = ctx.rd
        mk_p, mk_h, mk_2 = ctx.drop_masks
        if rd is None:
            g2b, g2_lp = g2, (g2_side_lp if g2_side_lp is not None else cast(g2, T))
        else:
            (g2b, g2_lp), g2_cs = _branch_grad(g2, rd, 1, N, mk_2, T), None
        qkvw_, qkvb_, pw_, pb_, f1w_, f1b_, f2w_, f2b_ = ctx.prm
        need = ctx.needs_input_grad
        dw2, db2 = _wgrad_bias(g2_lp, h_act, D, Hd, M, f2w_, f2b_, need[13], need[14], dy_f32=g2b, dy_colsum=g2_cs)
        dh_pre = torch.empty(M, Hd, dtype=T, device=dev)
        gemm(g2_lp, w2, dh_pre, M=M, N=Hd, K=D, trans_b=True, ldb=Hd, act=ACT_DGELU, aux=h_pre)
        if mk_h is not None:
            dh_pre = dropout_bwd(dh_pre, mk_h, rd.drop)
        dw1, db1 = _wgrad_bias(dh_pre, ln2, Hd, D, M, f1w_, f1b_, need[11], need[12])
        dln2 = torch.empty(M, D, dtype=T, device=dev)
        gemm(dh_pre, w1, dln2, M=M, N=D, K=Hd, trans_b=True, ldb=D)
        side = bf and rd is None
        g1, g1_lp, dn2w, dn2b = layernorm_bwd(dln2, x1, n2w, mean2, rstd2, gres=g2, want_lp=side, params=ctx.nprm[2:], side=side)
        g1_side_lp, g1_cs = _side_take(g1) if side else (None, None)
        g1b = g1
        if rd is not None:
            g1b, g1_lp = _branch_grad(g1, rd, 0, N, mk_p, T)
        elif not bf:
            g1_lp = g1
        dwp, dbp = _wgrad_bias(g1_lp, att, D, D, M, pw_, pb_, need[7], need[8], dy_f32=g1b, dy_colsum=g1_cs)
        datt = torch.empty(M, D, dtype=T, device=dev)
        gemm(g1_lp, wp, datt, M=M, N=D, K=D, trans_b=True, ldb=D)
        dqkv = torch.empty(M, 3 * D, dtype=T, device=dev)
        dtable = torch.empty(R, heads, dtype=torch.float32, device=dev)
        lse, addm, addmT = ctx.win
        if lse is not None:
            nchunk = call("evp_window_attention_fused_nchunk", Bg, nG, heads)
            dA = torch.empty(nchunk * addm.numel(), dtype=torch.float32, device=dev)
            call("evp_window_attention_fused_bwd", ptr(qkv), ptr(att), ptr(datt), ptr(lse), ptr(addm), ptr(addmT), Bg, nG, N, heads, scale,
                 ptr(dqkv), ptr(dA), stream_ptr())
            call("evp_window_bias_reduce", ptr(dA), ptr(rel), Bg, nG, N, heads, R, ptr(dtable), stream_ptr())
        else:
            call("evp_window_attention_bwd", ptr(qkv), ptr(tab), ptr(rel), ptr(att), ptr(datt), ptr(dqkv), ptr(dtable), Bg, nG, N, heads,
                 R, scale, dt(qkv), ptr(ctx.keep[0]), ctx.keep[1], stream_ptr())
        dwq, dbq = _wgrad_bias(dqkv, ln1, 3 * D, D, M, qkvw_, qkvb_, need[5], need[6])
        dln1 = torch.empty(M, D, dtype=T, device=dev)
        gemm(dqkv, wq, dln1, M=M, N=D, K=3 * D, trans_b=True, ldb=D)
        g0, _, dn1w, dn1b = layernorm_bwd(dln1, x2d, n1w, mean1, rstd1, gres=g1, want_lp=bf, params=ctx.nprm[:2], side=bf)
        return (g0.view(Bg, N, D), dtable, None, dn1w, dn1b, dwq, dbq, dwp, dbp, dn2w, dn2b, dw1, db1, dw2, db2, None, None, None, None)


def _attn_keep_mask(rd, shape, dev):
    """uint8 keep flags for dropout on attention probabilities of `shape`: rd.masks["attn"] when given, else drawn by evp_dropout_fwd from
    rd's Philox stream (the same draw the ViT path makes in attention_dropout_fwd: flag e <- word e of the stream at rd's next offset)."""
    if rd.masks is not None and "attn" in rd.masks:
        m = _chk(rd.masks["attn"]).view(-1)
        if m.numel() != math.prod(shape) or m.dtype != torch.uint8:
            raise _lib.EvpError(f"attention keep mask: expected uint8 with {math.prod(shape)} elements, got {m.dtype} {tuple(m.shape)}")
        return m
    sub = BlockDrop(drop=rd.attn_drop, seed=rd.seed_dev if rd.seed_dev is not None else rd.seed)
    sub._n = rd.next_offset(math.prod(shape))
    return dropout_fwd(torch.ones(shape, dtype=torch.float32, device=dev), sub, "attn")[1]


def swin_block(x, blk, rel, eps, want_attn=False, rd=None):
    a = blk.attn
    return SwinBlockFn.apply(x, a.relative_position_bias_table, rel, blk.norm1.weight, blk.norm1.bias, a.qkv.weight, a.qkv.bias,
                             a.proj.weight, a.proj.bias, blk.norm2.weight, blk.norm2.bias, blk.mlp.fc1.weight, blk.mlp.fc1.bias,
                             blk.mlp.fc2.weight, blk.mlp.fc2.bias, a.num_heads, eps, want_attn, rd)


class DropoutFn(torch.autograd.Function):
    """nn.Dropout on a float32 tensor (vit.py:114 / convvit.py:132 / swin.py:185 `pos_drop`): evp_dropout_fwd / evp_dropout_apply."""

    @staticmethod
    def forward(ctx, x, p, seed, mask=None):
        xc = _chk(x.detach().contiguous(), torch.float32)
        rd = BlockDrop(drop=p, seed=seed, masks=None if mask is None else {"x": mask})
        out, mk = dropout_fwd(xc, rd, "x")
        ctx.save_for_backward(mk)
        ctx.p = p
        return out.view(x.shape)

    @staticmethod
    def backward(ctx, g):
        (mk,) = ctx.saved_tensors
        return dropout_bwd(_chk(g.contiguous(), torch.float32), mk, ctx.p).view(g.shape), None, None, None


def draw_block_drop(module, n_samples, device):
    """BlockDrop for one training-mode call of a residual block module carrying `drop_path_rate` / `drop_rate` attributes, or None
    when both are off (the fused fast path). The draws come from torch's default generator of the device (as timm's `drop_path`
    and nn.Dropout do), so torch.manual_seed reproduces a run; the per-element masks use the Philox stream keyed by a seed
    drawn from the same device generator (draw_drop_seed)."""
    dp = float(getattr(module, "drop_path_rate", 0.0) or 0.0)
    dr = float(getattr(module, "drop_rate", 0.0) or 0.0)
    da = float(getattr(module, "attn_drop_rate", 0.0) or 0.0)
    if not module.training or (dp <= 0.0 and dr <= 0.0 and da <= 0.0):
        return None
    u1 = u2 = None
    if dp > 0.0:
        u = torch.rand(2, n_samples, device=device)
        u1, u2 = u[0], u[1]
    seed = draw_drop_seed(device) if (dr > 0.0 or da > 0.0) else 0
    return BlockDrop(u1, u2, 1.0 - dp, dr, seed, attn_drop=da)


def draw_drop_seed(device):
    """Key of one dropout stream, drawn ON the device from torch's generator (no host read-back; under HIP-graph capture torch's
    generator advances per replay, so every replay of a captured step gets fresh keys -- ADVICE r3)."""
    return torch.randint(0, 2 ** 62, (1,), device=device, dtype=torch.int64)


class AddFn(torch.autograd.Function):
    """a + b on f32 tokens (the 4-way stage sum of swin.py:239 exceeds LayerNormFn's three fused inputs)."""

    @staticmethod
    def forward(ctx, a, b):
        return add(a.detach().contiguous(), b.detach().contiguous())

    @staticmethod
    def backward(ctx, g):
        return g, g


# ----------------------------------------------------------------------------------------------------- fine-tune (cls) head
class TokenMeanFn(torch.autograd.Function):
    """emb_h.mean(dim=1) over the tokens (ft_cls_hub_model.py:136) on f32 [B,N,D]."""

    @staticmethod
    def forward(ctx, x):
        B, N, D = x.shape
        out = torch.empty(B, D, dtype=torch.float32, device=x.device)
        call("evp_token_mean_fwd", ptr(_chk(x.detach().contiguous(), torch.float32)), B, N, D, ptr(out), stream_ptr())
        ctx.dims = (B, N, D)
        return out

    @staticmethod
    def backward(ctx, g):
        B, N, D = ctx.dims
        dx = torch.empty(B, N, D, dtype=torch.float32, device=g.device)
        call("evp_token_mean_bwd", ptr(_chk(g.contiguous(), torch.float32)), B, N, D, ptr(dx), stream_ptr())
        return dx


class CrossEntropyFn(torch.autograd.Function):
    """nn.CrossEntropyLoss()(pred, label) (mean reduction; ft_cls_trainer.py:66) on f32 logits [B, n_cls]; smoothing > 0:
    timm's LabelSmoothingCrossEntropy(smoothing)(pred, label) (ft_cls_trainer.py:63-64)."""

    @staticmethod
    def forward(ctx, logits, labels, smoothing=0.0):
        B, Cn = logits.shape
        lg = _chk(logits.detach().contiguous(), torch.float32)
        loss = torch.empty(1, dtype=torch.float32, device=lg.device)
        dlog = torch.empty(B, Cn, dtype=torch.float32, device=lg.device)
        ws = torch.empty(B, dtype=torch.float32, device=lg.device)
        call("evp_cross_entropy_smooth", ptr(lg), ptr(_chk(labels.contiguous(), torch.int64)), B, Cn, Cn, float(smoothing), ptr(loss),
             ptr(dlog), ptr(ws), stream_ptr())
        ctx.save_for_backward(dlog)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        (dlog,) = ctx.saved_tensors
        out = dlog.clone()
        call("evp_scale_f32", ptr(out), ptr(_chk(g.contiguous().view(1), torch.float32)), out.numel(), stream_ptr())
        return out, None, None
