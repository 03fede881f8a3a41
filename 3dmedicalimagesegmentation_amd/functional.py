"""Host-side composition of the HIP kernels into differentiable blocks (torch.autograd.Function).

PyTorch is plumbing here: it owns device memory (caching allocator), streams and the autograd graph; every
FLOP of the hot path runs in libunetr_hip.so through the C ABI of include/unetr_hip.h.  Tensors that reach
these functions must live on a ROCm device -- there is no CPU or eager-PyTorch fallback.

Layouts: token matrices [B*L, H]; feature maps channels-last [B, D, H, W, C] (possibly row-pitched views).
"""
import ctypes
import os
import weakref

import torch

from . import _capi
from ._capi import GemmDesc, GemmBf16Desc, call, call_rc

LN_EPS = 1e-5
IN_EPS = 1e-5

_WS = {}
_WS_BYTES = int(os.environ.get("UNETR_AMD_WS_MB", "256")) << 20     # largest users: split-K slabs (M x N x splits fp32), InstanceNorm
                                                                     # partial rows, weight-gradient partials outside arena mode
_WS_SHARED = {}                                                      # device index -> stream ids that use the device's shared buffer


def share_workspace(stream):
    """Declare that work on `stream` never runs concurrently with work on the device's default stream or on another stream declared
    here -- the caller orders them with wait_stream / graph-replay order (train_step.side_stream: eager warm-up and graph capture
    run there while the launching stream waits; replays are launched on the launching stream).  Such streams use ONE scratch buffer
    per device instead of one each (256 MB per (device, stream) before: main + warm-up + capture = 768 MB)."""
    _WS_SHARED.setdefault(stream.device.index, set()).add(stream.cuda_stream)


def workspace(device):
    """Scratch for split-K slabs and reduction partials: one buffer per (device, stream) -- reuse inside a stream is
    stream-ordered and therefore safe; two streams (e.g. two models stepping concurrently) never share one -- except the streams
    declared by share_workspace, which together with the default stream share the device's buffer."""
    sid = torch.cuda.current_stream(device).cuda_stream
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if sid == 0 or sid in _WS_SHARED.get(idx, ()):
        sid = 0
    key = (device.type, idx, sid)
    ws = _WS.get(key)
    if ws is None:
        ws = torch.empty(_WS_BYTES // 4, dtype=torch.float32, device=device)
        _WS[key] = ws
    return ws


def release_workspaces():
    """drop every scratch buffer (they are re-created on demand; captured graphs that baked a pointer must be dropped first)"""
    _WS.clear()


def _stream():
    return torch.cuda.current_stream().cuda_stream


# ---- gradient sinks: parameters registered here get their gradients written straight into a slice of one flat
# arena (UNETR.use_flat_buffers), so AdamW is one launch and the data-parallel all-reduce needs no flatten copies.
# Everything mutable about that fast path -- the sink table, the deferred weight-gradient queues, the readiness
# callbacks of the data-parallel reducer -- belongs to ONE ArenaState per model (UNETR.use_flat_buffers creates it), so
# several models in one process do not share a queue.  The module-level table only maps a parameter's storage address
# to its owner; entries are validated against the live parameter (weak reference + address) on every lookup.
class ArenaState:
    def __init__(self):
        self.sinks = {}            # data_ptr -> (weakref(param), arena view)
        self.ready_cb = []         # data-parallel reducers: called with a parameter once its arena gradient is final
        self.defer = {"wgrad": [], "wgrad_b": [], "colsum": [], "reduce": [], "params": [], "armed": False, "prec": 0}
        self.unmaintained = 0      # arena shadows of THIS model re-derived since its flat optimizer last vouched for them
        self.fuse = None           # optim.AdamW.begin_fused_step: the deferred weight-gradient launch applies AdamW itself

    def reset_deferred(self):
        """Drop whatever a backward pass that raised left queued (its end-of-pass callback never ran): without this the
        queue stays armed, no later pass re-arms the flush, and AdamW steps on stale arena gradients."""
        d = self.defer
        d["wgrad"], d["wgrad_b"], d["colsum"], d["reduce"], d["params"], d["armed"] = [], [], [], [], [], False

    def clear(self):
        self.fuse = None
        for k in list(self.sinks):
            _GRAD_SINK.pop(k, None)
        self.sinks.clear()
        self.reset_deferred()


_DEFAULT_STATE = ArenaState()
_GRAD_SINK = {}                    # data_ptr -> ArenaState owning that parameter


def register_grad_sinks(params_and_views, state=None):
    state = state if state is not None else _DEFAULT_STATE
    for p, view in params_and_views:
        state.sinks[p.data_ptr()] = (weakref.ref(p), view)
        _GRAD_SINK[p.data_ptr()] = state
    return state


def clear_grad_sinks(state=None):
    """forget the arena registrations (of one model's state, or of every model when called without one)"""
    if state is not None:
        state.clear()
        return
    for st in set(_GRAD_SINK.values()):
        st.clear()
    _DEFAULT_STATE.clear()
    _GRAD_SINK.clear()


def _sink(w):
    """(state, param, view) when tensor `w` IS a registered, still-living parameter at its registered address"""
    st = _GRAD_SINK.get(w.data_ptr())
    if st is None:
        return None
    ent = st.sinks.get(w.data_ptr())
    if ent is None:
        return None
    p = ent[0]()
    if p is None or p.data_ptr() != w.data_ptr() or ent[1].shape != w.shape:
        return None
    return st, p, ent[1]


def _gout(w):
    """Destination for the gradient of parameter tensor `w`: its arena slice when registered and no gradient is
    currently attached (accumulating into an existing .grad must not alias it), else None (fresh tensor)."""
    ent = _sink(w)
    if ent is None or ent[1].grad is not None:
        return None
    return ent[2]


def _ret(w, g, deferred=False):
    """What a Function.backward returns for parameter `w`: when `g` was (or, if deferred, will be) written into
    w's arena slice, attach the slice as .grad directly (autograd would clone it: the arena keeps a second
    reference) and return None.  Readiness callbacks (data-parallel buckets) fire now, or at flush if deferred."""
    ent = _sink(w) if g is not None else None
    if ent is not None and g.data_ptr() == ent[2].data_ptr():
        st, p, view = ent
        p.grad = view
        if deferred:
            st.defer["params"].append(p)
        else:
            for cb in st.ready_cb:
                cb(p)
        return None
    return g


# ---- deferred, grouped weight gradients -------------------------------------------------------------------
# In arena mode the Linear weight/bias gradients of the ViT are not needed by anything inside backward, so they
# are queued (per ArenaState) and executed at the end of the backward pass as ONE grouped GEMM launch + ONE grouped
# column-sum launch (csrc: gemm_grouped_wgrad_kernel, colsum_grouped_kernel) instead of ~90 small latency-bound launches.
def _arm_flush(st):
    if not st.defer["armed"]:
        st.defer["armed"] = True
        torch.autograd.Variable._execution_engine.queue_callback(lambda: flush_deferred(st))


def wgrad_or_defer(dy, x, prec, w, dyb=None, xb=None):
    """dw[N,K] = dy^T x for Linear weight `w`; returns (grad_or_None_for_autograd).  dyb / xb: bf16 twins of dy / x -- the
    deferred launch then runs the bf16-storage grouped kernel (unetr_gemm_bf16_grouped_wgrad)."""
    out = _gout(w)
    twins = dyb is not None and xb is not None and dyb.shape[0] % 8 == 0 and dyb.shape[1] % 8 == 0 and xb.shape[1] % 8 == 0
    if out is None:
        if not twins:
            return linear_wgrad(dy, x, prec)
        dw = torch.empty(dyb.shape[1], xb.shape[1], dtype=torch.float32, device=dyb.device)
        _launch_deferred((), (), [(dyb, xb, dw)])
        return dw
    st = _GRAD_SINK[w.data_ptr()]
    if twins:
        st.defer["wgrad_b"].append((dyb, xb, out))
        _arm_flush(st)
        return _ret(w, out, deferred=True)
    st.defer["wgrad"].append((dy, x, out))
    st.defer["prec"] = prec
    _arm_flush(st)
    return _ret(w, out, deferred=True)


def colsum_or_defer(x, M, N, ld, b, view_shape=None):
    out = _gout(b)
    if out is None:
        if x.dtype == torch.bfloat16:          # (bf16 rows: the grouped kernel reads them; one problem, launched now)
            r = torch.empty(N, dtype=torch.float32, device=x.device)
            _launch_deferred((), [(x, r, M, N, ld)])
        else:
            r = colsum(x, M, N, ld)
        return r.view(view_shape) if view_shape is not None else r
    st = _GRAD_SINK[b.data_ptr()]
    st.defer["colsum"].append((x, out, M, N, ld))
    _arm_flush(st)
    return _ret(b, out, deferred=True)


def reduce_defer_enabled():
    """UNETR_AMD_REDUCE_DEFER=0 (A/B hook): every conv weight-gradient kernel launches its own partial-sum reduction again"""
    return os.environ.get("UNETR_AMD_REDUCE_DEFER", "1") != "0"


def _launch_reduces(rq):
    """every queued weight-gradient reduction (dst[i] = sum over rows of part[row][i]) in ONE launch"""
    arr = (_capi.ReduceProblem * len(rq))()
    for i, (part, dst, n, rows) in enumerate(rq):
        arr[i].part, arr[i].dst, arr[i].n, arr[i].rows = part.data_ptr(), dst.data_ptr(), n, rows
    call("unetr_reduce_rows_grouped", arr, len(rq), _stream())


def _launch_deferred(wq, cq, wbq=(), prec=0, fuse=None):
    if wbq and fuse is not None:
        # the optimizer rides on the launch: problems whose destination is a registered arena slice get AdamW in the epilogue
        # (their gradient is never stored); anything else keeps the plain launch below
        index = fuse["index"]
        fused = [(q, index[q[2].data_ptr()]) for q in wbq if q[2].data_ptr() in index]
        wbq = [q for q in wbq if q[2].data_ptr() not in index]
        if fused:
            arr = (_capi.GroupedProblem * len(fused))()
            idx = (ctypes.c_int * len(fused))()
            for i, ((dyb, xb, out), pi) in enumerate(fused):
                arr[i].dy, arr[i].x, arr[i].dw = dyb.data_ptr(), xb.data_ptr(), out.data_ptr()
                arr[i].M, arr[i].N, arr[i].K = dyb.shape[0], dyb.shape[1], xb.shape[1]
                idx[i] = pi
            if fuse.get("kind") == "bf16out":      # data-parallel step, bf16 communication: bf16(dW) straight into the comm buffer
                call("unetr_gemm_bf16_grouped_wgrad_bf16out", arr, len(fused), fuse["grad"], fuse["out"], fuse["total"], _stream())
            else:
                call("unetr_gemm_bf16_grouped_wgrad_adamw", arr, len(fused), ctypes.byref(fuse["arena"]), idx, _stream())
            fuse["done"].extend(pi for _, pi in fused)
    if wbq:
        arr = (_capi.GroupedProblem * len(wbq))()
        for i, (dyb, xb, out) in enumerate(wbq):
            arr[i].dy, arr[i].x, arr[i].dw = dyb.data_ptr(), xb.data_ptr(), out.data_ptr()
            arr[i].M, arr[i].N, arr[i].K = dyb.shape[0], dyb.shape[1], xb.shape[1]
        call("unetr_gemm_bf16_grouped_wgrad", arr, len(wbq), _stream())
    if wq and prec == _capi.PREC_BF16X3 and os.environ.get("UNETR_AMD_X3_WGRAD_STACK", "1") != "0":
        # bf16x3: dW = dY^T X over (hi, lo) halves is the bf16 kernel's contraction over three times the rows of the stacks
        # [dYh; dYh; dYl] / [Xh; Xl; Xh] (two streaming split launches per problem) -- the generic fp32-storage grouped kernel spends
        # 2.2 ms per step on these 76 GF, the bf16 kernel 0.2 ms per 432 rows; with an optimizer epilogue armed it rides here too
        rest, stacked, splits = [], [], []
        for dy, x, out in wq:
            M, N, K = dy.shape[0], dy.shape[1], x.shape[1]
            if (M * N) % 8 or (M * K) % 8 or (3 * M) % 8 or N % 8 or K % 8 or not dy.is_contiguous() or not x.is_contiguous():
                rest.append((dy, x, out))
                continue
            dys = torch.empty(3 * M, N, dtype=torch.bfloat16, device=dy.device)
            xs = torch.empty(3 * M, K, dtype=torch.bfloat16, device=dy.device)
            splits += [(dy, dys, M, N, 0), (x, xs, M, K, 1)]
            stacked.append((dys, xs, out))
        wq = rest
        if stacked:
            arr = (_capi.SplitProblem * len(splits))()
            for i, (src, dst, rows, cols, second) in enumerate(splits):
                arr[i].src, arr[i].dst, arr[i].rows, arr[i].cols, arr[i].second = src.data_ptr(), dst.data_ptr(), rows, cols, second
            call("unetr_split_stack_bf16_grouped", arr, len(splits), _stream())          # every stack of the pass in one launch
            _launch_deferred((), (), stacked, prec, fuse)
    if wq:
        arr = (_capi.GroupedProblem * len(wq))()
        for i, (dy, x, out) in enumerate(wq):
            arr[i].dy, arr[i].x, arr[i].dw = dy.data_ptr(), x.data_ptr(), out.data_ptr()
            arr[i].M, arr[i].N, arr[i].K = dy.shape[0], dy.shape[1], x.shape[1]
        call("unetr_gemm_grouped_wgrad", arr, len(wq), prec, _stream())
    if cq:
        arr = (_capi.ColsumProblem * len(cq))()
        for i, (x, out, M, N, ld) in enumerate(cq):
            arr[i].x, arr[i].out, arr[i].ld, arr[i].M, arr[i].N = x.data_ptr(), out.data_ptr(), ld, M, N
            arr[i].x_bf16 = int(x.dtype == torch.bfloat16)
        call("unetr_colsum_grouped", arr, len(cq), _stream())


def flush_deferred(st=None):
    """Runs at the end of the backward pass (autograd engine callback) on the backward stream."""
    st = st if st is not None else _DEFAULT_STATE
    d = st.defer
    wq, cq, wbq, params, prec, rq = d["wgrad"], d["colsum"], d["wgrad_b"], d["params"], d["prec"], d["reduce"]
    st.reset_deferred()
    if rq:
        _launch_reduces(rq)
    _launch_deferred(wq, cq, wbq, prec, st.fuse)
    for p in params:
        for cb in st.ready_cb:
            cb(p)


def act_dtype(prec):
    """storage type of the conv side's feature maps and their gradients: bf16 in bf16 precision mode (half the HBM bytes of
    the bandwidth-bound InstanceNorm / conv passes), fp32 in fp32 mode.  The C ABI infers it from `prec` (csrc/common.hpp)."""
    return torch.bfloat16 if prec == _capi.PREC_BF16 else torch.float32


def _a16(t):
    return int(t.dtype == torch.bfloat16)


def _prec_of(t):
    return _capi.PREC_BF16 if t.dtype == torch.bfloat16 else _capi.PREC_F32


def _as_act(t, prec):
    """tensor in the activation storage type of `prec` (a torch cast: only on the rare generic-fallback boundaries)"""
    dt = act_dtype(prec)
    return t if t.dtype == dt else t.to(dt)


def _require_gpu(t, act=False):
    if not t.is_cuda:
        raise RuntimeError("3dmedicalimagesegmentation_amd: the HIP backend needs tensors on a ROCm device "
                           "(got a CPU tensor); there is no CPU fallback.")
    if t.dtype != torch.float32 and not (act and t.dtype == torch.bfloat16):
        raise RuntimeError(f"3dmedicalimagesegmentation_amd: fp32 storage expected, got {t.dtype}")
    if t.device.index != torch.cuda.current_device():
        # kernels are launched on the CURRENT device's stream with raw pointers: a tensor of another GPU would fault
        raise RuntimeError(f"3dmedicalimagesegmentation_amd: tensor lives on {t.device} but the current device is "
                           f"cuda:{torch.cuda.current_device()}; call torch.cuda.set_device({t.device.index}) (one process per GPU)")


def _rows(t):
    """(tensor, ld) for a tensor usable as a row-pitched matrix [rows, C]: last dim contiguous and all
    leading dims collapsible with a single pitch.  Falls back to a contiguous copy."""
    c = t.shape[-1]
    if t.stride(-1) == 1 and t.dim() >= 2:
        ld = t.stride(-2)
        al = 16 // t.element_size()            # elements per 16 bytes
        ok = ld >= c and (ld % al == 0 or (ld == c and t.dtype == torch.float32)) and t.data_ptr() % 16 == 0
        exp = ld
        for d in range(t.dim() - 2, -1, -1):
            if t.shape[d] != 1 and t.stride(d) != exp:
                ok = False
                break
            exp *= t.shape[d]
        if ok:
            return t, ld
    t = t.contiguous()
    return t, c


# ------------------------------------------------------------------------------------------------ wrappers
def gemm(A, B, C, M, N, K, *, lda, ldb, ldc, prec, a_trans=False, b_trans=False, bias=None, res=None, ldr=0,
         res_mod=0, pre=None, aux=None, ldaux=0, act=0, accumulate=False, alpha=1.0, b_words=None):
    """b_words (bf16x3 mode): the pre-split word shadow of the weight B (weight_x3) -- read instead of B where the LDS-DMA kernel takes
    the shape (M >= 32, K % 32 == 0, N % 4 == 0, plain row pitches)"""
    d = GemmDesc()
    d.b_x3words = 0
    if (b_words is not None and prec == _capi.PREC_BF16X3 and not a_trans and M >= 32 and K % 32 == 0 and N % 4 == 0 and lda % 4 == 0
            and ldb % 4 == 0 and ldc % 4 == 0 and (not b_trans or N >= 64)):
        B = b_words
        d.b_x3words = 1
    d.M, d.N, d.K, d.batch = M, N, K, 1
    d.a_trans, d.b_trans = int(a_trans), int(b_trans)
    d.lda, d.ldb, d.ldc = lda, ldb, ldc
    d.strideA = d.strideB = d.strideC = d.strideR = 0
    d.bias = bias.data_ptr() if bias is not None else None
    d.res = res.data_ptr() if res is not None else None
    d.ldr, d.res_mod = ldr, res_mod
    d.pre = pre.data_ptr() if pre is not None else None
    d.aux = aux.data_ptr() if aux is not None else None
    d.ldaux = ldaux
    d.act, d.accumulate, d.alpha, d.prec = act, int(accumulate), alpha, prec
    ws = workspace(C.device)
    call("unetr_gemm", ctypes.byref(d), A.data_ptr(), B.data_ptr(), C.data_ptr(), ws.data_ptr(), ws.numel() * 4, _stream())


def cast_bf16(src, out=None):
    """fp32 -> bf16 (RNE) copy through the HIP cast kernel"""
    src = src.contiguous()
    if out is None:
        out = torch.empty(src.shape, dtype=torch.bfloat16, device=src.device)
    call("unetr_cast_bf16", src.data_ptr(), out.data_ptr(), src.numel(), _stream())
    return out


def gemm_bf16(A, B, M, N, K, *, b_kn=False, C=None, Cb=None, lda=None, ldb=None, bias=None, res=None, ldr=0, res_mod=0,
              pre=None, aux=None, ldaux=0, act=0, accumulate=False, alpha=1.0, ldcb=None, tc=None):
    """C / Cb [M,N] = epilogue(A[M,K] @ (B[N,K]^T | B[K,N])) with bf16-stored operands (csrc/gemm_bf16.hip)"""
    assert A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16
    d = GemmBf16Desc()
    d.M, d.N, d.K, d.b_kn = M, N, K, int(b_kn)
    d.lda = lda if lda is not None else K
    d.ldb = ldb if ldb is not None else (N if b_kn else K)
    d.ldc = d.ldcb = N
    if ldcb is not None:
        d.ldcb = ldcb
    if tc is not None:              # (D, H, W, Cout): scatter the bf16 output as a 2x2x2 transposed conv (tap-major columns)
        d.tc_d, d.tc_h, d.tc_w, d.tc_cout = tc
    d.bias = bias.data_ptr() if bias is not None else None
    d.res = res.data_ptr() if res is not None else None
    d.ldr, d.res_mod = ldr, res_mod
    d.pre = pre.data_ptr() if pre is not None else None
    d.aux = aux.data_ptr() if aux is not None else None
    d.ldaux = ldaux
    d.act, d.accumulate, d.alpha = act, int(accumulate), alpha
    ws = workspace(A.device)
    call("unetr_gemm_bf16", ctypes.byref(d), A.data_ptr(), B.data_ptr(), C.data_ptr() if C is not None else None,
         Cb.data_ptr() if Cb is not None else None, ws.data_ptr(), ws.numel() * 4, _stream())


def ln_gemm_bf16(x, gamma, beta, Wb, *, bias=None, act=0, C=None, Cb=None, pre=None, xn=None, mean=None, rstd=None):
    """C / Cb [M,N] = act(LayerNorm(x)[M,K] @ Wb[N,K]^T + bias) in one launch (csrc/encoder.hip); optionally keeps the
    normalised rows `xn` (bf16), `mean`, `rstd` and the pre-activation `pre` for backward"""
    M, K = x.shape
    N = Wb.shape[0]
    d = _capi.LnGemmDesc()
    d.x, d.ldx, d.gamma, d.beta, d.eps = x.data_ptr(), K, gamma.data_ptr(), beta.data_ptr(), LN_EPS
    d.W, d.ldw, d.bias, d.act = Wb.data_ptr(), K, _p(bias), act
    d.pre, d.ldpre, d.Cb, d.ldcb, d.C, d.ldc = _p(pre), N, _p(Cb), N, _p(C), N
    d.xn, d.mean, d.rstd = _p(xn), _p(mean), _p(rstd)
    d.M, d.N, d.K = M, N, K
    call("unetr_ln_gemm_bf16", ctypes.byref(d), _stream())


def fused_ln_enabled():
    """LayerNorm as the prologue of the consuming GEMM (unetr_ln_gemm_bf16).  Off by default: measured on MI355X at 432
    rows it LOSES to LayerNorm kernel + bf16 GEMM (13.3 vs 3.1 + 6.6 us for qkv, 17.5 vs 3.1 + 10.5 us for linear1): a CU
    pulls ~70 GB/s from L2, and the fused tile reads its 64 fp32 rows (2x the bytes of the bf16 rows the GEMM alone reads)
    once per column tile -- 36 times per row tile."""
    return os.environ.get("UNETR_AMD_FUSED_LN", "0") == "1"


def bf16_attention_enabled():
    return os.environ.get("UNETR_AMD_BF16_ATTENTION", "1") != "0"


# ---- bf16 operand storage (bf16 precision mode) ---------------------------------------------------------------
# The encoder's Linear layers read bf16-STORED operands through unetr_gemm_bf16: activations are emitted as bf16 by
# the producing kernels (LayerNorm, attention, GELU epilogue), weights come from a bf16 shadow.  A shadow is fresh when
# (a) the flat-arena AdamW maintains it (the optimizer kernel writes the bf16 copy next to the fp32 master) and torch
# has not modified the parameter since (version counter), or (b) it was cast in this "weight epoch" -- every code path
# of this package that changes weights behind torch's back (per-parameter AdamW, DDP broadcast) bumps the epoch.
# Under hipGraph capture a non-maintained shadow is always re-cast, so the cast is part of the captured step.
_SHADOW = {}
_WEIGHT_EPOCH = [0]


def mark_flat_maintained(params, state=None):
    """flat-arena AdamW has just rewritten the whole bf16 shadow arena together with the masters: every registered shadow
    of these parameters is in step again.  Only walks the table after something of THIS model had to be re-derived: the
    count lives in the model's ArenaState (a process-wide one let model A's step clear what model B still had to redo)."""
    if state is None and params:
        state = _GRAD_SINK.get(params[0].data_ptr())
    if state is not None and not state.unmaintained:
        return
    for w in params:
        ent = _SHADOW.get(id(w))
        if ent is not None and ent[4]() is w and ent[5] == w.data_ptr():
            ent[1], ent[3] = w._version, True
    if state is not None:
        state.unmaintained = 0


def bf16_storage_enabled():
    return os.environ.get("UNETR_AMD_BF16_STORAGE", "1") != "0"


def invalidate_weight_shadows():
    """Declare every derived weight copy (bf16 shadows, packed conv weights) stale.  MANDATORY after writing a parameter
    through ``.data`` in place (``p.data.copy_()``, ``p.data.mul_()``, an EMA swap, a hand-written broadcast) once the
    copies are optimizer-maintained (this package's AdamW has stepped): such a write changes neither the version counter
    nor the storage address, so nothing else can notice it.  Everything torch can see is handled automatically: in-place
    ops on the parameter, torch optimizers, ``load_state_dict``, re-pointed ``.data``, and -- for copies no optimizer of
    this package maintains -- any change at all between two forward passes (``begin_forward``)."""
    _WEIGHT_EPOCH[0] += 1
    for ent in _SHADOW.values():
        ent[1] = -1
    for ent in _SHADOW_X3.values():
        ent[1] = -1
    for ent in _PACKS.values():
        ent[1] = -1


def begin_forward(state=None):
    """Called by UNETR.forward.  (1) A derived weight copy that no optimizer of this package keeps in step is trusted for
    ONE forward/backward pass only: a new pass starts a new weight epoch, so the first use re-derives it (a ``.data``
    write between two passes is then always seen; optimizer-maintained copies stay valid, see
    invalidate_weight_shadows).  (2) Deferred weight-gradient work a failed backward left behind is dropped, and so is a fused
    epilogue (AdamW.begin_fused_step / TrainStep's bf16 communication epilogue) that was armed for a step which never finished."""
    if not torch.cuda.is_current_stream_capturing():
        _WEIGHT_EPOCH[0] += 1
    if state is not None:
        state.reset_deferred()
        state.fuse = None          # an optimizer / communication epilogue is armed AFTER the forward of the step it belongs to:
                                   # whatever a failed step left armed must not ride on the next backward


# ---- bf16x3 mode: word shadow of the weights ---------------------------------------------------------------------------------
# The bf16x3 Linear GEMMs split their fp32 operands into (hi, lo) bf16 pairs in registers; the weight operand of a step is constant,
# so with flat arenas its split form is kept as a second arena of words [hi | lo << 16] (unetr_split_words: one streaming launch
# over the parameter arena after every optimizer step, AdamW.* -> refresh_x3_shadow) and the GEMM kernel splits only the activation
# operand.  Same freshness rule as the bf16 shadow: valid while torch has not modified the parameter (version counter, address);
# ``.data`` writes behind torch's back need invalidate_weight_shadows().  Created on first use (an eager pass: never under capture).
_SHADOW_X3 = {}


def x3_words_enabled():
    """UNETR_AMD_X3_WORDS=0 (A/B hook): the bf16x3 GEMMs split the fp32 weights in registers like the activations"""
    return os.environ.get("UNETR_AMD_X3_WORDS", "1") != "0"


def _split_words(src, dst, n):
    call("unetr_split_words", src.data_ptr(), dst.data_ptr(), n, _stream())


def weight_x3(w):
    """the weight as split words (int32 view, same shape) when `w` lives in a flat arena; else None (the kernel splits the fp32 weight)"""
    if not x3_words_enabled():
        return None
    st = _GRAD_SINK.get(w.data_ptr())
    flat = getattr(st, "flat", None) if st is not None else None
    if flat is None:
        return None
    if flat.get("shadow_x3") is None:
        if torch.cuda.is_current_stream_capturing():
            return None                      # (the arena must not come out of a graph's private pool)
        arena = torch.empty(flat["total"], dtype=torch.int32, device=w.device)
        _split_words(flat["param"], arena, flat["total"])
        flat["shadow_x3"] = arena
        for p, o in zip(flat["params"], flat["offsets"]):
            _SHADOW_X3[id(p)] = [arena[o:o + p.numel()].view_as(p), p._version, weakref.ref(p), p.data_ptr(), (p.numel() + 7) // 8 * 8]
    ent = _SHADOW_X3.get(id(w))
    if ent is None or ent[2]() is not w or ent[3] != w.data_ptr() or ent[0].device != w.device:
        return None
    if ent[1] != w._version:                 # torch modified the parameter since the last derive: redo this slice (whole padded slot)
        _split_words(w, ent[0], ent[4])
        ent[1] = w._version
    return ent[0]


def refresh_x3_shadow(flat, written=False):
    """optimizer side (flat arenas): the parameter arena has just been updated by kernels of this package.  written: those kernels
    wrote the word shadow themselves (AdamW arena launches take its pointer); otherwise one derive launch over the arena -- when the
    bf16x3 GEMMs have asked for a word shadow at all"""
    arena = flat.get("shadow_x3") if flat is not None else None
    if arena is None:
        return
    if not written:
        _split_words(flat["param"], arena, flat["total"])
    for p in flat["params"]:
        ent = _SHADOW_X3.get(id(p))
        if ent is not None and ent[2]() is p and ent[3] == p.data_ptr():
            ent[1] = p._version


def register_weight_shadow(w, shadow):
    """flat-arena mode: `shadow` is the bf16 view the optimizer kernel keeps in step with `w`"""
    _SHADOW[id(w)] = [shadow, w._version, _WEIGHT_EPOCH[0], True, weakref.ref(w), w.data_ptr()]   # caller has just cast it


def weight_bf16(w):
    ent = _SHADOW.get(id(w))
    if ent is None or ent[4]() is not w or ent[0].shape != w.shape or ent[0].device != w.device:
        ent = [torch.empty(w.shape, dtype=torch.bfloat16, device=w.device), -1, -1, False, weakref.ref(w), 0]
        _SHADOW[id(w)] = ent
    # (a re-pointed parameter -- `p.data = other` -- keeps its version counter: the storage address is part of the key)
    fresh = ent[1] == w._version and ent[5] == w.data_ptr() and (
        ent[3] or (ent[2] == _WEIGHT_EPOCH[0] and not torch.cuda.is_current_stream_capturing()))
    if not fresh:
        cast_bf16(w.detach(), out=ent[0])
        # re-derived here = somebody other than this package's optimizer changed the weight (or may have): the copy is no
        # longer optimizer-maintained until that optimizer steps again (shadow_ptr_for_update / mark_flat_maintained)
        ent[1], ent[2], ent[5], ent[3] = w._version, _WEIGHT_EPOCH[0], w.data_ptr(), False
        st = _GRAD_SINK.get(w.data_ptr())
        if st is not None:
            st.unmaintained += 1
    return ent[0]


# ---- packed conv weights (MFMA operand layouts of csrc/conv3.hip) ------------------------------------------------------
# Same freshness rules as the bf16 shadows: a pack is fresh when the optimizer of this package maintains it (one grouped
# re-pack launch right after the update, refresh_conv_packs) and torch has not modified the parameter since, or it was
# packed in this weight epoch outside graph capture.  kind 0 / 1: 3x3x3 forward / data gradient, 2 / 3: 1x1x1 forward /
# transposed.  Without this the step spent ~25 five-microsecond pack launches on its critical path.
_PACKS = {}


def conv_pack_get(w, kind, prec):
    key = (id(w), kind, prec)
    ent = _PACKS.get(key)
    cout, cin = w.shape[0], w.shape[1]
    if ent is None or ent[4]() is not w or ent[0].device != w.device:
        lib = _capi.load()
        if kind == 4:          # transposed conv [in, out, 2, 2, 2] -> bf16 [in][tap][out]
            nbytes = w.numel() * 2
        else:
            nbytes = lib.unetr_conv3_packed_bytes(cin, cout, kind, prec) if kind <= 1 else (
                lib.unetr_conv3_packed_1x1_bytes(cin, cout, prec) if kind == 2 else lib.unetr_conv3_packed_1x1_bytes(cout, cin, prec))
        ent = [torch.empty(nbytes, dtype=torch.uint8, device=w.device), -1, -1, False, weakref.ref(w), 0]
        _PACKS[key] = ent
    fresh = ent[1] == w._version and ent[5] == w.data_ptr() and (
        ent[3] or (ent[2] == _WEIGHT_EPOCH[0] and not torch.cuda.is_current_stream_capturing()))
    if not fresh:
        _pack_launch([(w, ent[0], cin, cout, kind)], prec)
        ent[1], ent[2], ent[5], ent[3] = w._version, _WEIGHT_EPOCH[0], w.data_ptr(), False
    return ent[0]


def _pack_launch(items, prec):
    arr = (_capi.PackProblem * len(items))()
    for i, (w, buf, cin, cout, kind) in enumerate(items):
        arr[i].w, arr[i].out, arr[i].Cin, arr[i].Cout, arr[i].kind = w.data_ptr(), buf.data_ptr(), cin, cout, kind
    call("unetr_conv3_pack_grouped", arr, len(items), prec, _stream())


def refresh_conv_packs():
    """optimizer side: re-pack every registered conv weight in one launch per precision; the packs are optimizer-maintained
    from here on"""
    by_prec = {}
    for key, ent in list(_PACKS.items()):
        w = ent[4]()
        if w is None:
            del _PACKS[key]
            continue
        if not w.is_cuda or ent[1] != w._version or ent[5] != w.data_ptr():
            continue                                  # torch touched the parameter: conv_pack_get re-packs on demand
        by_prec.setdefault(key[2], []).append((w, ent[0], w.shape[1], w.shape[0], key[1]))
        ent[3] = True
    for prec, items in by_prec.items():
        _pack_launch(items, prec)


def shadow_ptr_for_update(w):
    """optimizer side: where the bf16 copy of `w` lives, if the GEMMs have asked for one; the caller's kernel rewrites
    it together with the fp32 master, which makes the shadow optimizer-maintained from here on"""
    ent = _SHADOW.get(id(w))
    if ent is None or ent[4]() is not w or ent[1] != w._version or ent[5] != w.data_ptr():
        return None
    ent[3] = True
    return ent[0].data_ptr()


def _bf16_path(prec, *kdims):
    return prec == _capi.PREC_BF16 and bf16_storage_enabled() and all(k % 64 == 0 for k in kdims)


def _twin(t):
    """bf16 copy of an fp32 activation / gradient: the twin its producer attached, else a cast"""
    if t.dtype == torch.bfloat16:
        return t
    tw = getattr(t, "_unetr_bf16", None)
    if tw is not None and tw[0].numel() == t.numel() and tw[1] == t._version and t.is_contiguous():   # autograd may accumulate into t in place
        return tw[0].view(t.shape)
    return cast_bf16(t)


def _attach_twin(t, tb):
    t._unetr_bf16 = (tb, t._version)


def carry_twin(src, dst):
    """dst is the same values as src under another tensor object (a detached leaf, an alias returned by a Function): the bf16
    twin and the stashed LayerNorm of src describe dst too"""
    tw = getattr(src, "_unetr_bf16", None)
    if tw is not None and tw[1] == src._version:
        dst._unetr_bf16 = (tw[0], dst._version)
    if hasattr(src, "_unetr_ln"):
        dst._unetr_ln = src._unetr_ln
    return dst


def add_cast(a, b, want_bf16=True):
    """a + b (fp32) and its bf16 twin in one launch (csrc: add_cast_bf16_kernel)"""
    a, b = a.contiguous(), b.contiguous()
    out = torch.empty_like(a)
    ok16 = want_bf16 and a.numel() % 4 == 0
    ob = torch.empty(a.shape, dtype=torch.bfloat16, device=a.device) if ok16 else None
    if a.numel() % 4 or a.dtype != torch.float32 or b.dtype != torch.float32:
        raise RuntimeError("add_cast needs fp32 tensors with a multiple of 4 elements")
    call("unetr_add_cast_bf16", a.data_ptr(), b.data_ptr(), out.data_ptr(), _p(ob), a.numel(), _stream())
    if ob is not None:
        _attach_twin(out, ob)
    return out


class TapFn(torch.autograd.Function):
    """A hidden state with TWO consumers (unetr.py:197-201: z3 / z6 / z9 feed the next transformer block and encoder2-4):
    forward hands out two aliases, backward sums the two gradients AND forms the bf16 operand of the producing block's backward
    GEMMs in one HIP launch -- autograd's own fan-in was an elementwise add kernel, and the twin-less sum then cost a cast."""

    @staticmethod
    def forward(ctx, x, want_bf16):
        ctx.want_bf16 = bool(want_bf16)
        a, b = x.view_as(x), x.view_as(x)
        carry_twin(x, a)
        carry_twin(x, b)
        return a, b

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None or gb is None:
            return (ga if gb is None else gb), None
        _require_gpu(ga)
        return add_cast(ga, gb, ctx.want_bf16), None


def linear_fwd(x, w, bias, prec, res=None, res_mod=0, act=0, pre=None):
    """y[M,N] = act(x[M,K] @ w[N,K]^T + bias) + res"""
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty(M, N, dtype=torch.float32, device=x.device)
    gemm(x, w, y, M, N, K, lda=K, ldb=K, ldc=N, prec=prec, bias=bias, res=res, ldr=N, res_mod=res_mod, act=act, pre=pre,
         b_words=weight_x3(w) if prec == _capi.PREC_BF16X3 else None)
    return y


def linear_dgrad(dy, w, prec, aux=None):
    """dx[M,K] = dy[M,N] @ w[N,K]  (optionally times gelu'(aux))"""
    M, N = dy.shape
    K = w.shape[1]
    dx = torch.empty(M, K, dtype=torch.float32, device=dy.device)
    gemm(dy, w, dx, M, K, N, lda=N, ldb=K, ldc=K, prec=prec, b_trans=True, aux=aux, ldaux=K, act=2 if aux is not None else 0,
         b_words=weight_x3(w) if prec == _capi.PREC_BF16X3 else None)
    return dx


def linear_wgrad(dy, x, prec, out=None):
    """dw[N,K] = dy[M,N]^T @ x[M,K]"""
    M, N = dy.shape
    K = x.shape[1]
    dw = out if out is not None else torch.empty(N, K, dtype=torch.float32, device=dy.device)
    gemm(dy, x, dw, N, K, M, lda=N, ldb=K, ldc=K, prec=prec, a_trans=True, b_trans=True)
    return dw


def colsum(x, M, N, ld, out=None):
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=x.device)
    ws = workspace(x.device)
    call("unetr_colsum", x.data_ptr(), ld, M, N, out.data_ptr(), 0, ws.data_ptr(), ws.numel() * 4, _stream())
    return out


def _p(t):
    return t.data_ptr() if t is not None else None


def bf16_like(t):
    return torch.empty(t.shape, dtype=torch.bfloat16, device=t.device)


def layernorm_fwd(x, w, b, bf16_out=None, want_fp32=True):
    M, H = x.shape
    y = torch.empty_like(x) if want_fp32 else None
    mean = torch.empty(M, dtype=torch.float32, device=x.device)
    rstd = torch.empty(M, dtype=torch.float32, device=x.device)
    call("unetr_layernorm_fwd", x.data_ptr(), w.data_ptr(), b.data_ptr(), _p(y), _p(bf16_out), mean.data_ptr(),
         rstd.data_ptr(), M, H, LN_EPS, _stream())
    return y, mean, rstd


def layernorm_bwd(dy, x, w, mean, rstd, dres=None, out_w=None, out_b=None, dx_bf16=None):
    M, H = x.shape
    dx = torch.empty_like(x)
    dw = out_w if out_w is not None else torch.empty(H, dtype=torch.float32, device=x.device)
    db = out_b if out_b is not None else torch.empty(H, dtype=torch.float32, device=x.device)
    ws = workspace(x.device)
    call("unetr_layernorm_bwd", dy.data_ptr(), x.data_ptr(), w.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(),
         _p(dx_bf16), dres.data_ptr() if dres is not None else None, dw.data_ptr(), db.data_ptr(), M, H, ws.data_ptr(), ws.numel() * 4, _stream())
    return dx, dw, db


def layernorm_bwd_params(dy, x, w, b, mean, rstd, dres=None, dx_bf16=None):
    """LayerNorm backward returning (dx, grad_w, grad_b) as autograd wants them.  In arena mode the dgamma/dbeta
    reduction over row blocks is not needed inside backward: the kernel leaves its partials in a private buffer and the
    two column sums join the grouped launch at the end of the pass (one launch for all 25 LayerNorms)."""
    ow, ob = _gout(w), _gout(b)
    if ow is None or ob is None:
        dx, dw, db = layernorm_bwd(dy, x, w, mean, rstd, dres=dres, out_w=ow, out_b=ob, dx_bf16=dx_bf16)
        return dx, _ret(w, dw), _ret(b, db)
    M, H = x.shape
    nblk = (M + 3) // 4
    part = torch.empty(nblk * 2 * H, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    call("unetr_layernorm_bwd", dy.data_ptr(), x.data_ptr(), w.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(),
         _p(dx_bf16), _p(dres), None, None, M, H, part.data_ptr(), part.numel() * 4, _stream())
    st = _GRAD_SINK[w.data_ptr()]
    st.defer["colsum"].append((part, ow, nblk, H, 2 * H))
    st.defer["colsum"].append((part[H:], ob, nblk, H, 2 * H))
    _arm_flush(st)
    return dx, _ret(w, ow, deferred=True), _ret(b, ob, deferred=True)


def x3_ride_ok(M, N, K):
    """shapes the bf16x3 LDS-DMA GEMM takes (csrc/gemm_bf16.hip: unetr_gemm_x3_dma) -- where the LayerNorm-riding forms can be used"""
    return M >= 32 and K % 32 == 0 and N % 4 == 0 and N >= 64 and os.environ.get("UNETR_X3_GEMM_DMA", "1") != "0"


def gemm_bf16_ln_fwd(A, B, M, N, K, C, gamma, beta, y_bf16, bias=None, res=None, ldr=0, y=None, b_words=None):
    """C[M,N] = A[M,K] @ B[N,K]^T + bias + res, and y_bf16 = LayerNorm(C) (gamma, beta) with its (mean, rstd):
    unetr_gemm_bf16_ln_fwd -- the LayerNorm of the NEXT layer rides on the split-K reduction of this GEMM.
    bf16x3 mode: A / B fp32 (B optionally as its word shadow b_words), the normalised rows go to the fp32 tensor ``y``."""
    x3 = A.dtype == torch.float32
    assert x3 or (A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16)
    d = GemmBf16Desc()
    d.x3 = (2 if b_words is not None else 1) if x3 else 0
    if b_words is not None:
        B = b_words
    d.M, d.N, d.K, d.b_kn = M, N, K, 0
    d.lda, d.ldb, d.ldc, d.ldcb = K, K, N, N
    d.bias = bias.data_ptr() if bias is not None else None
    d.res = res.data_ptr() if res is not None else None
    d.ldr, d.alpha = ldr, 1.0
    mean = torch.empty(M, dtype=torch.float32, device=C.device)
    rstd = torch.empty(M, dtype=torch.float32, device=C.device)
    ws = workspace(C.device)
    call("unetr_gemm_bf16_ln_fwd", ctypes.byref(d), A.data_ptr(), B.data_ptr(), C.data_ptr(), gamma.data_ptr(), beta.data_ptr(), LN_EPS,
         _p(y), _p(y_bf16), mean.data_ptr(), rstd.data_ptr(), ws.data_ptr(), ws.numel() * 4, _stream())
    return mean, rstd


def ln_ride_enabled():
    """norm1 of block i+1 is formed by the kernel that sums block i's last split-K GEMM (unetr_gemm_bf16_ln_fwd) instead of
    its own launch: bit-identical and one launch less per block.  On by default since round 3 (4.527 vs 4.544 ms/step in three
    interleaved rounds on one box) -- it had measured neutral while the LayerNorm kernel walked the slabs in a run-time loop, one
    memory round trip per slab and vector; with every load of the row requested up front the row-owning form wins.
    UNETR_AMD_LN_RIDE=0 restores the separate split-K reduce + LayerNorm launches.  The backward counterpart
    (unetr_gemm_bf16_ln_bwd) is always on."""
    return os.environ.get("UNETR_AMD_LN_RIDE", "1") == "1"


def _stash_ln(x, gamma, beta, xn, mean, rstd):
    """remember on the residual-stream tensor x that LayerNorm(x; gamma, beta) already exists (computed by the kernel that
    produced x): the consumer that would launch exactly that LayerNorm takes it from here (_stashed_ln)"""
    x._unetr_ln = (xn, mean, rstd, gamma.data_ptr(), gamma._version, beta.data_ptr(), beta._version, x._version)


def _stashed_ln(x, gamma, beta):
    st = getattr(x, "_unetr_ln", None)
    if st is None or st[3:] != (gamma.data_ptr(), gamma._version, beta.data_ptr(), beta._version, x._version) or st[0].shape != x.shape:
        return None
    return st[0], st[1], st[2]


def gemm_ln_bwd_params(A, Bw, M, N, K, x, w, b, mean, rstd, dres=None, dx_bf16=None, b_words=None):
    """(dx, grad_w, grad_b) of a LayerNorm whose output gradient is dy = A[M,K] @ Bw[K,N] (the data gradient of the Linear
    layer behind it, bf16-stored operands, Bw read as the [K, N] operand): unetr_gemm_bf16_ln_bwd -- when the GEMM is cut into K
    slabs the LayerNorm kernel sums them itself, so the separate split-K reduce launch disappears (bit-identical)."""
    x3 = A.dtype == torch.float32              # bf16x3 mode: fp32 operands (Bw optionally as its word shadow b_words)
    assert x3 or (A.dtype == torch.bfloat16 and Bw.dtype == torch.bfloat16)
    d = GemmBf16Desc()
    d.x3 = (2 if b_words is not None else 1) if x3 else 0
    if b_words is not None:
        Bw = b_words
    d.M, d.N, d.K, d.b_kn = M, N, K, 1
    d.lda, d.ldb, d.ldc, d.ldcb = K, N, N, N
    d.alpha = 1.0
    scratch = torch.empty(M, N, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    ws = workspace(x.device)
    ow, ob = _gout(w), _gout(b)
    nblk = (M + 3) // 4
    if ow is None or ob is None:
        dw = ow if ow is not None else torch.empty(N, dtype=torch.float32, device=x.device)
        db = ob if ob is not None else torch.empty(N, dtype=torch.float32, device=x.device)
        lnws = torch.empty(nblk * 2 * N, dtype=torch.float32, device=x.device)
        call("unetr_gemm_bf16_ln_bwd", ctypes.byref(d), A.data_ptr(), Bw.data_ptr(), scratch.data_ptr(), x.data_ptr(), w.data_ptr(),
             mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), _p(dx_bf16), _p(dres), dw.data_ptr(), db.data_ptr(),
             lnws.data_ptr(), lnws.numel() * 4, ws.data_ptr(), ws.numel() * 4, _stream())
        return dx, _ret(w, dw), _ret(b, db)
    part = torch.empty(nblk * 2 * N, dtype=torch.float32, device=x.device)
    call("unetr_gemm_bf16_ln_bwd", ctypes.byref(d), A.data_ptr(), Bw.data_ptr(), scratch.data_ptr(), x.data_ptr(), w.data_ptr(),
         mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), _p(dx_bf16), _p(dres), None, None,
         part.data_ptr(), part.numel() * 4, ws.data_ptr(), ws.numel() * 4, _stream())
    st = _GRAD_SINK[w.data_ptr()]
    st.defer["colsum"].append((part, ow, nblk, N, 2 * N))
    st.defer["colsum"].append((part[N:], ob, nblk, N, 2 * N))
    _arm_flush(st)
    return dx, _ret(w, ow, deferred=True), _ret(b, ob, deferred=True)


def attention_fwd(qkv, B, L, heads, dh, prec, out_bf16=None):
    out = torch.empty(B * L, heads * dh, dtype=torch.float32, device=qkv.device)
    lse = torch.empty(B, heads, L, dtype=torch.float32, device=qkv.device)
    call("unetr_attention_fwd", qkv.data_ptr(), out.data_ptr(), _p(out_bf16), lse.data_ptr(), B, L, heads, dh, float(dh) ** -0.5, prec, _stream())
    return out, lse


def attention_bwd(qkv, out, dout, lse, B, L, heads, dh, prec, dqkv_bf16=None):
    dqkv = torch.empty_like(qkv)
    delta = torch.empty_like(lse)
    call("unetr_attention_bwd", qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), _p(dqkv_bf16),
         delta.data_ptr(),
         B, L, heads, dh, float(dh) ** -0.5, prec, _stream())
    return dqkv


def attention_bf16_fwd(qkvb, B, L, heads, dh, out_bf16, out=None):
    lse = torch.empty(B, heads, L, dtype=torch.float32, device=qkvb.device)
    call("unetr_attention_bf16_fwd", qkvb.data_ptr(), _p(out), out_bf16.data_ptr(), lse.data_ptr(), B, L, heads, dh, float(dh) ** -0.5, _stream())
    return lse


def attention_bf16_bwd(qkvb, outb, doutb, lse, B, L, heads, dh, dqkv=None):
    dqkvb = torch.empty_like(qkvb)
    delta = torch.empty_like(lse)
    call("unetr_attention_bf16_bwd", qkvb.data_ptr(), outb.data_ptr(), doutb.data_ptr(), lse.data_ptr(), _p(dqkv), dqkvb.data_ptr(),
         delta.data_ptr(), B, L, heads, dh, float(dh) ** -0.5, _stream())
    return dqkvb


def _use_gemm_conv():
    """UNETR_AMD_CONV=gemm routes 3x3x3 convs through the generic im2col-loader GEMM family instead of the
    dedicated LDS-halo kernels (kept for cross-checking one HIP path against the other)."""
    import os
    return os.environ.get("UNETR_AMD_CONV", "halo") == "gemm"


def conv3(x, ldx, w, dims, prec, mode=0, out=None, ldo=None, accumulate=False):
    """3x3x3 conv (mode 0: w[Cout,Cin,3,3,3] applied to x with Cin channels) or its data gradient
    (mode 1: x carries Cout channels, result has Cin channels)."""
    B, D, H, W = dims
    cout_w, cin_w = w.shape[0], w.shape[1]
    cin, cout = (cin_w, cout_w) if mode == 0 else (cout_w, cin_w)
    if _use_gemm_conv() or cout % 16 != 0:
        return conv_fwd(x, ldx, conv_pack(w, mode), dims, cin, cout, 3, prec, out=out, ldo=ldo, accumulate=accumulate)
    wp = conv_pack_get(w, mode, prec)
    if out is None:
        out = torch.empty(B, D, H, W, cout, dtype=act_dtype(prec), device=x.device)
        ldo = cout
    call("unetr_conv3_fwd", x.data_ptr(), ldx, wp.data_ptr(), out.data_ptr(), ldo, int(accumulate), B, D, H, W, cin, cout, prec, _stream())
    return out


def conv3_fused(x, ldx, w, w3, dims, prec):
    """First half of a residual block in one launch: c = conv3x3x3(x, w) with its InstanceNorm statistics and, when w3 is
    given, c3 = conv1x1x1(x, w3) with its statistics (both convs read the same staged window).  Returns
    (c, stats, c3, stats3) or None when the shape has to take the unfused kernels."""
    B, D, H, W = dims
    cout, cin = w.shape[0], w.shape[1]
    level = int(os.environ.get("UNETR_AMD_CONV_FUSE", "2"))     # tuning hook: 0 unfused, 1 statistics only, 2 + 1x1x1 conv
    if _use_gemm_conv() or cout % 16 != 0 or level == 0 or (w3 is not None and level < 2):
        return None
    dev = x.device
    wp = conv_pack_get(w, 0, prec)
    adt = act_dtype(prec)
    x_f32 = int(prec == _capi.PREC_BF16 and x.dtype == torch.float32)      # the image in front of encoder1
    c = torch.empty(B, D, H, W, cout, dtype=adt, device=dev)
    st = torch.empty(B, cout, 2, dtype=torch.float32, device=dev)
    wp3 = c3 = st3 = None
    if w3 is not None:
        wp3 = conv_pack_get(w3, 2, prec)
        c3 = torch.empty(B, D, H, W, cout, dtype=adt, device=dev)
        st3 = torch.empty(B, cout, 2, dtype=torch.float32, device=dev)
    ws = workspace(dev)
    rc = call_rc("unetr_conv3_fwd_fused", x.data_ptr(), ldx, wp.data_ptr(), c.data_ptr(), cout, st.data_ptr(), _p(wp3), _p(c3), cout,
                 _p(st3), IN_EPS, B, D, H, W, cin, cout, prec, x_f32, ws.data_ptr(), ws.numel() * 4, _stream())
    if rc != 0:
        return None
    return c, st, c3, st3


CONV3_MAX_ROWS = 1024        # = UNETR_CONV3_MAX_ROWS (include/unetr_hip.h)


def in_fuse_level():
    """UNETR_AMD_IN_FUSE (tuning / A-B hook), a bit mask: 1 = the forward statistics finalize rides in the prologue of the apply
    kernel (no finalize launches in the forward of a residual block), 2 = the backward statistics of a block's first norm are
    formed in the epilogue of the data-gradient conv that produces its gradient (no reduction pass, no finalize launch)."""
    return int(os.environ.get("UNETR_AMD_IN_FUSE", "3"))


def img_branch_enabled():
    """UNETR_AMD_IMG_BRANCH=0 (A/B hook): the residual block on the image stores and re-reads its 1x1x1 branch like every other block"""
    return os.environ.get("UNETR_AMD_IMG_BRANCH", "1") != "0"


def conv3_parts(x, ldx, w, w3, dims, prec, store3=True):
    """conv3_fused without the statistics finalize: returns (c, part, c3, part3, rows) -- part* = InstanceNorm partial rows
    [B, rows, 2, Cout] that instnorm_apply_fin reduces in its prologue -- or None when the shape takes another route.
    store3=False: the 1x1x1 branch is not stored (c3 is None), only its statistics rows are formed."""
    B, D, H, W = dims
    cout, cin = w.shape[0], w.shape[1]
    if _use_gemm_conv() or cout % 16 != 0 or cout > 128 or int(os.environ.get("UNETR_AMD_CONV_FUSE", "2")) < 2:
        return None
    dev = x.device
    wp = conv_pack_get(w, 0, prec)
    adt = act_dtype(prec)
    x_f32 = int(prec == _capi.PREC_BF16 and x.dtype == torch.float32)
    c = torch.empty(B, D, H, W, cout, dtype=adt, device=dev)
    part = torch.empty(B, CONV3_MAX_ROWS, 2, cout, dtype=torch.float32, device=dev)
    wp3 = c3 = part3 = None
    if w3 is not None:
        wp3 = conv_pack_get(w3, 2, prec)
        c3 = torch.empty(B, D, H, W, cout, dtype=adt, device=dev) if store3 else None
        part3 = torch.empty(B, CONV3_MAX_ROWS, 2, cout, dtype=torch.float32, device=dev)
    rows = ctypes.c_int(0)
    rc = call_rc("unetr_conv3_fwd_parts", x.data_ptr(), ldx, wp.data_ptr(), c.data_ptr(), cout, part.data_ptr(), _p(wp3), _p(c3), cout,
                 _p(part3), ctypes.byref(rows), B, D, H, W, cin, cout, prec, x_f32, _stream())
    if rc != 0:
        return None
    return c, part, c3, part3, rows.value


def instnorm_apply_fin(x, part, rows, B, V, C, lrelu, x2=None, part_b=None, rows_b=0, out=None, ldo=None):
    """instnorm_apply with the statistics formed in the kernel's prologue from partial rows; returns (y, stats, stats_b) or None"""
    y = torch.empty_like(x) if out is None else out
    sa = torch.empty(B, C, 2, dtype=torch.float32, device=x.device)
    sb = torch.empty(B, C, 2, dtype=torch.float32, device=x.device) if x2 is not None else None
    rc = call_rc("unetr_instnorm_apply_fin", x.data_ptr(), C, part.data_ptr(), rows, _p(x2), C, _p(part_b), rows_b, sa.data_ptr(), _p(sb),
                 IN_EPS, y.data_ptr(), C if out is None else ldo, B, V, C, int(lrelu), _a16(x), _stream())
    if rc != 0:
        return None
    return y, sa, sb


IN_IMG_MAX_ROWS = 512       # = UNETR_IN_IMG_MAX_ROWS


def instnorm_apply_fin_img(x, part, rows, img, cin, w3, part_b, rows_b, B, V, C, out=None, ldo=None):
    """block end of the residual block on the image: lrelu(norm(x) + norm(conv1x1x1(img; w3))) with the 1x1x1 branch formed from the
    image in the kernel; returns (y, stats, stats_b) or None"""
    y = torch.empty_like(x) if out is None else out
    sa = torch.empty(B, C, 2, dtype=torch.float32, device=x.device)
    sb = torch.empty(B, C, 2, dtype=torch.float32, device=x.device)
    rc = call_rc("unetr_instnorm_apply_fin_img", x.data_ptr(), C, part.data_ptr(), rows, img.data_ptr(), cin, w3.data_ptr(), part_b.data_ptr(), rows_b,
                 sa.data_ptr(), sb.data_ptr(), IN_EPS, y.data_ptr(), C if out is None else ldo, B, V, C, 1, _a16(x), _stream())
    return (y, sa, sb) if rc == 0 else None


def instnorm_bwd_img(dy, lddy, x, sa, img, cin, w3, sb, B, V, C):
    """backward of the same block end: (dx of the 3x3x3 branch, dw3 partial rows [rows, C * cin], rows) or None"""
    if dy.dtype != x.dtype:
        dy = dy.to(x.dtype).contiguous(); lddy = dy.stride(-2)
    dx = torch.empty_like(x)
    part = torch.empty(IN_IMG_MAX_ROWS, C * cin, dtype=torch.float32, device=x.device)
    rows = ctypes.c_int(0)
    ws = workspace(x.device)
    rc = call_rc("unetr_instnorm_bwd_img", dy.data_ptr(), lddy, x.data_ptr(), C, sa.data_ptr(), img.data_ptr(), cin, w3.data_ptr(), sb.data_ptr(),
                 dx.data_ptr(), C, part.data_ptr(), ctypes.byref(rows), B, V, C, 1, ws.data_ptr(), ws.numel() * 4, _a16(x), _stream())
    return (dx, part, rows.value) if rc == 0 else None


def conv3_dgrad_stats(dy, w, xn, stats, dims, prec):
    """da = conv3x3x3^T(dy; w) together with the partial sums of the InstanceNorm backward of xn's norm (the norm whose
    lrelu'd output the conv read): returns (da, part, rows) or None when the shape takes the unfused route"""
    B, D, H, W = dims
    cout, cin = w.shape[0], w.shape[1]
    if _use_gemm_conv() or cin % 16 != 0 or cin > 128 or xn.stride(-2) != cin:
        return None
    wp = conv_pack_get(w, 1, prec)
    da = torch.empty(B, D, H, W, cin, dtype=act_dtype(prec), device=dy.device)
    part = torch.empty(B, CONV3_MAX_ROWS, 2, cin, dtype=torch.float32, device=dy.device)
    rows = ctypes.c_int(0)
    rc = call_rc("unetr_conv3_dgrad_stats", dy.data_ptr(), cout, wp.data_ptr(), da.data_ptr(), cin, xn.data_ptr(), cin, stats.data_ptr(),
                 part.data_ptr(), ctypes.byref(rows), B, D, H, W, cin, cout, prec, _stream())
    if rc != 0:
        return None
    return da, part, rows.value


def instnorm_bwd_apply_fin(dy, lddy, x, sa, part, rows, nsp, B, V, C, lrelu):
    """dx of y = lrelu?(norm(x)) from dy and the partial sums `part` [B, rows, nsp, C]; None when unsupported"""
    dx = torch.empty_like(x)
    rc = call_rc("unetr_instnorm_bwd_apply_fin", dy.data_ptr(), lddy, x.data_ptr(), C, sa.data_ptr(), None, 0, None, part.data_ptr(), rows, nsp,
                 dx.data_ptr(), C, None, 0, B, V, C, int(lrelu), _a16(x), _stream())
    return dx if rc == 0 else None


def out_fuse_enabled():
    """UNETR_AMD_OUT_FUSE=0 (A/B hook): decoder2's block end is stored and the out conv reads it, as every other block end"""
    return os.environ.get("UNETR_AMD_OUT_FUSE", "1") != "0"


def outconv_in_fwd(c2, p2, rows2, c3, p3, rows3, wo, bo, B, V, C):
    """logits of the out conv on lrelu(norm(c2) + norm(c3)) without storing that tensor (csrc/norm_misc.hip: outconv_in_fwd_kernel);
    returns (logits [B, Cout, V] fp32, stats of c2, stats of c3) or None when the shape takes the unfused sequence"""
    cout = wo.shape[0]
    if wo.shape[1] != C or not wo.is_contiguous():
        return None
    logits = torch.empty(B, cout, V, dtype=torch.float32, device=c2.device)
    sa = torch.empty(B, C, 2, dtype=torch.float32, device=c2.device)
    sb = torch.empty(B, C, 2, dtype=torch.float32, device=c2.device)
    rc = call_rc("unetr_outconv_in_fwd", c2.data_ptr(), C, p2.data_ptr(), rows2, c3.data_ptr(), C, p3.data_ptr(), rows3, sa.data_ptr(), sb.data_ptr(),
                 IN_EPS, wo.data_ptr(), _p(bo), logits.data_ptr(), B, V, C, cout, _a16(c2), _stream())
    return (logits, sa, sb) if rc == 0 else None


def outconv_in_bwd(dl, c2, s2, c3, s3, wo, bo, B, V, C):
    """backward of outconv_in_fwd up to the block end's normalisation: (dout in the storage type of c2, InstanceNorm backward partial
    rows [B, rows, 3, C], rows, dwo, dbo)"""
    cout = wo.shape[0]
    rows = _capi.load().unetr_outconv_in_bwd_rows(B, V, C, _a16(c2))
    if rows <= 0:
        raise RuntimeError("unetr_outconv_in_bwd: shape accepted by the forward kernel is declined by backward")
    dl = dl.contiguous()
    dout = torch.empty_like(c2)
    part = torch.empty(B, rows, 3, C, dtype=torch.float32, device=c2.device)
    gw, gb = _gout(wo), _gout(bo)
    dw = gw if gw is not None else torch.empty_like(wo)
    db = gb if gb is not None else torch.empty(cout, dtype=torch.float32, device=c2.device)
    ws = workspace(c2.device)
    call("unetr_outconv_in_bwd", dl.data_ptr(), c2.data_ptr(), C, s2.data_ptr(), c3.data_ptr(), C, s3.data_ptr(), wo.data_ptr(), dout.data_ptr(), C,
         part.data_ptr(), dw.data_ptr(), db.data_ptr(), B, V, C, cout, ws.data_ptr(), ws.numel() * 4, _a16(c2), _stream())
    return dout, part, rows, dw, db


def instnorm_bwd_apply_fin_dual(dy, lddy, x, sa, x2, sb, part, rows, B, V, C):
    """(dx, dx2) of y = lrelu(norm(x) + norm(x2)) from dy and the partial sums `part` [B, rows, 3, C]; None when unsupported"""
    dx, dx2 = torch.empty_like(x), torch.empty_like(x2)
    rc = call_rc("unetr_instnorm_bwd_apply_fin", dy.data_ptr(), lddy, x.data_ptr(), C, sa.data_ptr(), x2.data_ptr(), C, sb.data_ptr(), part.data_ptr(), rows, 3,
                 dx.data_ptr(), C, dx2.data_ptr(), C, B, V, C, 1, _a16(x), _stream())
    return (dx, dx2) if rc == 0 else None


def conv3_dgrad_fused(dc1, dc3, w1, w3, dx, dims, prec):
    """dx = conv3x3x3^T(dc1; w1) + conv1x1x1^T(dc3; w3) in one launch (the input gradient of a residual block);
    returns False when the shape has to take the two-kernel route."""
    B, D, H, W = dims
    cout, cin = w1.shape[0], w1.shape[1]
    if _use_gemm_conv() or cin % 16 != 0 or int(os.environ.get("UNETR_AMD_CONV_FUSE", "2")) < 2:
        return False
    wp = conv_pack_get(w1, 1, prec)
    w3t = conv_pack_get(w3, 3, prec)
    ws = workspace(dc1.device)
    rc = call_rc("unetr_conv3_dgrad_fused", dc1.data_ptr(), cout, wp.data_ptr(), dc3.data_ptr(), cout, w3.data_ptr(), w3t.data_ptr(), dx.data_ptr(), cin,
                 B, D, H, W, cin, cout, prec, ws.data_ptr(), ws.numel() * 4, _stream())
    return rc == 0


def conv3_wgrad(x, ldx, dy, lddy, dims, cin, cout, prec, out=None, dy3=None, out3=None, defer=None):
    """dw of the 3x3x3 conv; with dy3/out3 also the weight gradient of the 1x1x1 conv sharing the input x.
    defer = the ArenaState owning `out` (and `out3`): returns (dw, True) when the reduction of the partial sums was queued for the
    end-of-backward grouped launch; plain calls return dw."""
    want = defer is not None                 # such callers get (dw, queued)
    if _use_gemm_conv():
        if dy3 is not None:
            x32 = x if x.dtype == torch.float32 else x.float().contiguous()
            gemm(dy3.float() if dy3.dtype != torch.float32 else dy3, x32, out3, cout, cin, dims[0] * dims[1] * dims[2] * dims[3], lda=cout,
                 ldb=x32.stride(-2), ldc=cin, prec=_capi.PREC_F32, a_trans=True, b_trans=True)
        r = conv_wgrad(x, ldx, dy, lddy, dims, cin, cout, 3, prec, out=out)
        return (r, False) if want else r
    B, D, H, W = dims
    dw = out if out is not None else torch.empty(cout, cin, 3, 3, 3, dtype=torch.float32, device=x.device)
    x_f32 = int(prec == _capi.PREC_BF16 and x.dtype == torch.float32)
    if defer is not None and reduce_defer_enabled():
        # arena mode: the kernel leaves its per-workgroup partial sums in a buffer of its own and the reduction joins the ONE grouped
        # reduce launch at the end of the backward pass (functional.flush_deferred)
        rows = _capi.load().unetr_conv3_wgrad_rows(B, D, H, W, cin, cout, prec, x_f32, int(dy3 is not None))
        n, n3 = 27 * cin * cout, (cin * cout if dy3 is not None else 0)
        if rows > 0 and rows * (n + n3) * 4 <= (256 << 20):
            part = torch.empty(rows * (n + n3), dtype=torch.float32, device=x.device)
            got = ctypes.c_long(0)
            rc = call_rc("unetr_conv3_wgrad_parts", x.data_ptr(), ldx, dy.data_ptr(), lddy, _p(dy3), cout, part.data_ptr(), part.numel() * 4,
                         ctypes.byref(got), B, D, H, W, cin, cout, prec, x_f32, _stream())
            if rc == 0:
                g = got.value
                defer.defer["reduce"].append((part, dw, n, g))
                if dy3 is not None:
                    defer.defer["reduce"].append((part[g * n:], out3, n3, g))
                _arm_flush(defer)
                return dw, True
    ws = workspace(x.device)
    call("unetr_conv3_wgrad", x.data_ptr(), ldx, dy.data_ptr(), lddy, dw.data_ptr(),
         dy3.data_ptr() if dy3 is not None else None, cout, out3.data_ptr() if out3 is not None else None,
         B, D, H, W, cin, cout, prec, x_f32, ws.data_ptr(), ws.numel() * 4, _stream())
    return (dw, False) if want else dw


def conv_pack(w, mode):
    """torch Conv3d weight [Cout,Cin,k,k,k] -> GEMM operand layout (mode 0 fwd, mode 1 dgrad)."""
    cout, cin, ks = w.shape[0], w.shape[1], w.shape[2]
    wp = torch.empty(w.numel(), dtype=torch.float32, device=w.device)
    call("unetr_conv_pack_weight", w.data_ptr(), wp.data_ptr(), cin, cout, ks, mode, _stream())
    return wp


def conv_fwd(x, ldx, wpack, dims, cin, cout, ks, prec, out=None, ldo=None, accumulate=False):
    """x: rows [B*D*H*W, cin] pitch ldx -> y rows [.., cout]"""
    B, D, H, W = dims
    if prec == _capi.PREC_BF16:
        # generic im2col-loader GEMM family: fp32 storage only -- cast at the boundary (shapes the dedicated kernels decline)
        x32 = x.float().contiguous() if x.dtype != torch.float32 else x
        l32 = x32.stride(-2) if x32.dim() >= 2 else ldx
        o32 = torch.empty(B, D, H, W, cout, dtype=torch.float32, device=x.device)
        if accumulate and out is not None:
            o32.copy_(out)
        ws = workspace(x.device)
        call("unetr_conv_gemm_fwd", x32.data_ptr(), l32, wpack.data_ptr(), o32.data_ptr(), cout, int(accumulate), B, D, H, W, cin, cout, ks,
             prec, ws.data_ptr(), ws.numel() * 4, _stream())
        if out is None:
            return o32.to(act_dtype(prec))
        out.copy_(o32)
        return out
    if out is None:
        out = torch.empty(B, D, H, W, cout, dtype=torch.float32, device=x.device)
        ldo = cout
    ws = workspace(x.device)
    call("unetr_conv_gemm_fwd", x.data_ptr(), ldx, wpack.data_ptr(), out.data_ptr(), ldo, int(accumulate), B, D, H, W, cin, cout, ks,
         prec, ws.data_ptr(), ws.numel() * 4, _stream())
    return out


def conv_wgrad(x, ldx, dy, lddy, dims, cin, cout, ks, prec, out=None):
    B, D, H, W = dims
    dw = out if out is not None else torch.empty(cout, cin, ks, ks, ks, dtype=torch.float32, device=x.device)
    ws = workspace(x.device)
    if x.dtype != torch.float32:
        x = x.float().contiguous(); ldx = x.stride(-2)
    if dy.dtype != torch.float32:
        dy = dy.float().contiguous(); lddy = dy.stride(-2)
    call("unetr_conv_gemm_wgrad", x.data_ptr(), ldx, dy.data_ptr(), lddy, dw.data_ptr(), B, D, H, W, cin, cout, ks, prec,
         ws.data_ptr(), ws.numel() * 4, _stream())
    return dw


def instnorm_stats(x, ld, B, V, C):
    stats = torch.empty(B, C, 2, dtype=torch.float32, device=x.device)
    ws = workspace(x.device)
    call("unetr_instnorm_stats", x.data_ptr(), ld, B, V, C, IN_EPS, stats.data_ptr(), ws.data_ptr(), ws.numel() * 4, _a16(x), _stream())
    return stats


def instnorm_apply(x, sa, B, V, C, lrelu, x2=None, sb=None, out=None, ldo=None):
    """``out`` / ``ldo``: write into an existing buffer with row pitch ldo (the skip half of a concatenation buffer)"""
    y = torch.empty_like(x) if out is None else out
    call("unetr_instnorm_apply", x.data_ptr(), C, sa.data_ptr(), x2.data_ptr() if x2 is not None else None, C,
         sb.data_ptr() if sb is not None else None, y.data_ptr(), C if out is None else ldo, B, V, C, int(lrelu), _a16(x), _stream())
    return y


def instnorm_bwd(dy, lddy, x, sa, B, V, C, lrelu, x2=None, sb=None):
    if dy.dtype != x.dtype:
        dy = dy.to(x.dtype).contiguous(); lddy = dy.stride(-2)
    dx = torch.empty_like(x)
    dx2 = torch.empty_like(x2) if x2 is not None else None
    ws = workspace(x.device)
    call("unetr_instnorm_bwd", dy.data_ptr(), lddy, x.data_ptr(), C, sa.data_ptr(), x2.data_ptr() if x2 is not None else None, C,
         sb.data_ptr() if sb is not None else None, dx.data_ptr(), C, dx2.data_ptr() if dx2 is not None else None, C,
         B, V, C, int(lrelu), ws.data_ptr(), ws.numel() * 4, _a16(x), _stream())
    return dx, dx2


def _tconv_as_gemm(prec, M, cin, cout, ld_in):
    """the small transposed convs (768 channels at 6^3, 64 / 128 at 12^3) go through the bf16-storage GEMM: few voxels, many
    channels -- exactly where the dedicated voxel-tile kernels decline and the gather-loader GEMM family was 3-6x slower"""
    return (prec == _capi.PREC_BF16 and bf16_storage_enabled() and os.environ.get("UNETR_AMD_TCONV_GEMM", "1") != "0"
            and cin % 64 == 0 and M % 8 == 0 and ld_in == cin and M < 8192)


def tconv_fwd(x, ldx, w, dims, cin, cout, prec, out=None, ldo=None):
    B, D, H, W = dims
    adt = act_dtype(prec)
    if out is None:
        out = torch.empty(B, 2 * D, 2 * H, 2 * W, cout, dtype=adt, device=x.device)
        ldo = cout
    M = B * D * H * W
    if _tconv_as_gemm(prec, M, cin, cout, ldx):
        xb = _twin(x).view(M, cin)             # (a bf16 feature map is its own operand; tokens bring their bf16 twin)
        if out.dtype == torch.bfloat16 and cout % 4 == 0 and ldo % 4 == 0 and os.environ.get("UNETR_AMD_TCONV_SCATTER", "1") != "0":
            # tap-major weight pack [Cin][tap][Cout] (re-packed with the conv weights after every update): the GEMM's epilogue
            # writes each (voxel, tap) row of Cout channels straight to its output voxel -- no fp32 [M, 8 Cout] intermediate, no
            # pixel-shuffle launch
            wp = conv_pack_get(w, 4, prec).view(torch.bfloat16)
            gemm_bf16(xb, wp, M, 8 * cout, cin, b_kn=True, Cb=out, ldcb=ldo, tc=(D, H, W, cout))
            return out, xb
        tmp = torch.empty(M, 8 * cout, dtype=torch.float32, device=x.device)
        gemm_bf16(xb, weight_bf16(w).view(cin, 8 * cout), M, 8 * cout, cin, b_kn=True, C=tmp)
        call("unetr_pixel_shuffle2", tmp.data_ptr(), out.data_ptr(), ldo, B, D, H, W, cout, _a16(out), _stream())
        return out, xb
    ws = workspace(x.device)
    if prec == _capi.PREC_BF16 and x.dtype != adt:
        x = x.to(adt).contiguous(); ldx = cin      # (fp32 tokens in front of a shape the GEMM form does not take)
    args = (x.data_ptr(), ldx, w.data_ptr(), out.data_ptr(), ldo, B, D, H, W, cin, cout, prec, ws.data_ptr(), ws.numel() * 4, _stream())
    if call_rc("unetr_tconv2_fwd", *args) != 0:       # dedicated kernel declines the shape -> generic GEMM family (fp32 storage)
        if prec == _capi.PREC_BF16:
            x32 = x.float().contiguous()
            o32 = torch.empty(B, 2 * D, 2 * H, 2 * W, cout, dtype=torch.float32, device=x.device)
            call("unetr_tconv_fwd", x32.data_ptr(), cin, w.data_ptr(), o32.data_ptr(), cout, B, D, H, W, cin, cout, prec,
                 ws.data_ptr(), ws.numel() * 4, _stream())
            # (through the row-copy kernel, not Tensor.copy_: `out` may be a view handed out by TconvFn.forward -- the skip half of
            # a concatenation buffer -- and a torch in-place op on it inside a custom Function invalidates the view's grad_fn)
            o16 = o32.to(adt)
            call("unetr_copy_rows", out.data_ptr(), ldo, o16.data_ptr(), cout, B * 8 * D * H * W, cout, 0, _a16(out), _stream())
        else:
            call("unetr_tconv_fwd", *args)
    return out, None


def tconv_bwd(x, ldx, xb, dy, lddy, w, dims, cin, cout, prec, need_dx):
    """(dx or None, dw as autograd wants it) of the transposed conv; xb: the bf16 input the GEMM form of forward kept"""
    B, D, H, W = dims
    M = B * D * H * W
    if xb is not None:
        dyg = torch.empty(M, 8 * cout, dtype=torch.bfloat16, device=dy.device)
        call("unetr_pixel_unshuffle2_bf16", dy.data_ptr(), lddy, dyg.data_ptr(), B, D, H, W, cout, _a16(dy), _stream())
        dx = None
        if need_dx:
            dx = torch.empty(B, D, H, W, cin, dtype=x.dtype, device=dy.device)        # fp32 for tokens, bf16 for feature maps
            wb = weight_bf16(w).view(cin, 8 * cout)
            if dx.dtype == torch.bfloat16:
                gemm_bf16(dyg, wb, M, cin, 8 * cout, Cb=dx.view(M, cin))
            else:
                gemm_bf16(dyg, wb, M, cin, 8 * cout, C=dx.view(M, cin))
        # dw[Cin, Cout*8] = x^T dyg is torch's [Cin, Cout, 2, 2, 2] as it stands: joins the grouped end-of-backward launch
        dw = wgrad_or_defer(None, None, prec, w, xb, dyg)
        if dw is not None:
            dw = dw.view_as(w)
        return dx, dw
    if prec == _capi.PREC_BF16 and x.dtype != torch.bfloat16:
        x = x.to(torch.bfloat16).contiguous(); ldx = cin
    if prec == _capi.PREC_BF16 and dy.dtype != torch.bfloat16:
        dy = dy.to(torch.bfloat16).contiguous(); lddy = cout
    dx = tconv_dgrad(dy, lddy, w, dims, cin, cout, prec) if need_dx else None
    gw = _gout(w)
    st = _GRAD_SINK.get(w.data_ptr()) if gw is not None else None
    if st is not None and reduce_defer_enabled():
        # arena mode: the partial sums of the weight gradient stay in a buffer of their own; their reduction joins the grouped launch
        # at the end of the backward pass
        rows = _capi.load().unetr_tconv2_wgrad_rows(B, D, H, W, cin, cout)
        n = cin * cout * 8
        if rows > 0:
            part = torch.empty(rows * n, dtype=torch.float32, device=x.device)
            got = ctypes.c_long(0)
            if call_rc("unetr_tconv2_wgrad_parts", x.data_ptr(), ldx, dy.data_ptr(), lddy, part.data_ptr(), part.numel() * 4, ctypes.byref(got),
                       B, D, H, W, cin, cout, prec, _stream()) == 0:
                st.defer["reduce"].append((part, gw, n, got.value))
                _arm_flush(st)
                return dx, _ret(w, gw, deferred=True)
    dw = tconv_wgrad(x, ldx, dy, lddy, dims, cin, cout, prec, out=gw)
    return dx, _ret(w, dw)


def tconv_dgrad(dy, lddy, w, dims, cin, cout, prec):
    B, D, H, W = dims
    dx = torch.empty(B, D, H, W, cin, dtype=act_dtype(prec), device=dy.device)
    ws = workspace(dy.device)
    args = (dy.data_ptr(), lddy, w.data_ptr(), dx.data_ptr(), cin, 0, B, D, H, W, cin, cout, prec, ws.data_ptr(), ws.numel() * 4, _stream())
    if call_rc("unetr_tconv2_dgrad", *args) != 0:
        if prec == _capi.PREC_BF16:                   # generic GEMM family: fp32 storage
            d32 = dy.float().contiguous() if lddy == cout else dy[..., :cout].float().contiguous()
            x32 = torch.empty(B, D, H, W, cin, dtype=torch.float32, device=dy.device)
            call("unetr_tconv_dgrad", d32.data_ptr(), cout, w.data_ptr(), x32.data_ptr(), cin, 0, B, D, H, W, cin, cout, prec,
                 ws.data_ptr(), ws.numel() * 4, _stream())
            dx.copy_(x32)
        else:
            call("unetr_tconv_dgrad", *args)
    return dx


def tconv_wgrad(x, ldx, dy, lddy, dims, cin, cout, prec, out=None):
    B, D, H, W = dims
    dw = out if out is not None else torch.empty(cin, cout, 2, 2, 2, dtype=torch.float32, device=x.device)
    ws = workspace(x.device)
    args = (x.data_ptr(), ldx, dy.data_ptr(), lddy, dw.data_ptr(), B, D, H, W, cin, cout, prec, ws.data_ptr(), ws.numel() * 4, _stream())
    if call_rc("unetr_tconv2_wgrad", *args) != 0:
        if prec == _capi.PREC_BF16:
            x32 = x.float().contiguous() if ldx == cin else x[..., :cin].float().contiguous()
            d32 = dy.float().contiguous() if lddy == cout else dy[..., :cout].float().contiguous()
            call("unetr_tconv_wgrad", x32.data_ptr(), cin, d32.data_ptr(), cout, dw.data_ptr(), B, D, H, W, cin, cout, prec,
                 ws.data_ptr(), ws.numel() * 4, _stream())
        else:
            call("unetr_tconv_wgrad", *args)
    return dw


# --------------------------------------------------------------------------------- ViT encoder functions
class PatchEmbedFn(torch.autograd.Function):
    """MONAI PatchEmbeddingBlock(pos_embed="perceptron"): rearrange + Linear + position embedding."""

    @staticmethod
    def forward(ctx, x_in, w, b, pos, patch, prec):
        _require_gpu(x_in)
        x_in = x_in.contiguous()
        B, C, D, H, W = x_in.shape
        L = (D // patch) * (H // patch) * (W // patch)
        pd = C * patch ** 3
        hid = w.shape[0]
        ctx.pb = None
        if _bf16_path(prec, pd) and (B * L) % 8 == 0 and hid % 8 == 0:
            # the bf16 GEMM operand straight from the gather kernel (no fp32 patch matrix, no cast pass).  Only where backward's
            # weight gradient can take the bf16 grouped launch (wgrad_or_defer: rows and both widths multiples of 8): its fp32
            # fallback needs the fp32 patch matrix this branch does not keep (e.g. 48^3 at batch 1: 27 token rows)
            z = torch.empty(B * L, hid, dtype=torch.float32, device=x_in.device)
            ctx.pb = torch.empty(B * L, pd, dtype=torch.bfloat16, device=x_in.device)
            patches = x_in.new_empty(0)
            call("unetr_patch_gather", x_in.data_ptr(), None, ctx.pb.data_ptr(), B, C, D, H, W, patch, _stream())
            gemm_bf16(ctx.pb, weight_bf16(w), B * L, hid, pd, C=z, bias=b, res=pos, ldr=hid, res_mod=L)
        else:
            patches = torch.empty(B * L, pd, dtype=torch.float32, device=x_in.device)
            call("unetr_patch_gather", x_in.data_ptr(), patches.data_ptr(), None, B, C, D, H, W, patch, _stream())
            z = linear_fwd(patches, w, b, prec, res=pos, res_mod=L)
        ctx.save_for_backward(patches, w, b, pos)
        ctx.meta = (B, L, hid, prec)
        return z

    @staticmethod
    def backward(ctx, dz):
        patches, w, b, pos = ctx.saved_tensors
        B, L, hid, prec = ctx.meta
        dz = dz.contiguous()
        dw = wgrad_or_defer(dz, patches, prec, w, _twin(dz) if ctx.pb is not None else None, ctx.pb)
        db = colsum_or_defer(dz, B * L, hid, hid, b)
        dpos = colsum_or_defer(dz, B, L * hid, L * hid, pos, view_shape=(1, L, hid))
        return None, dw, db, dpos, None, None


def _tblock_forward(x, n1w, n1b, wqkv, wp, bp, n2w, n2b, w1, b1, w2, b2, B, L, heads, prec, train, next_ln=None, emit_twin=False):
    """kernels of one transformer block; returns (x2, tensors backward needs, bf16 twins or None).  next_ln = (gamma, beta) of
    the LayerNorm the next layer starts with: computed by the kernel that forms x2 and left on x2 (_stash_ln)"""
    hid = x.shape[1]
    dh = hid // heads
    M, mlp = x.shape[0], w1.shape[0]
    if _bf16_path(prec, hid, mlp) and M % 8 == 0:     # (the bf16 weight-gradient kernel wants whole 8-token groups)
        # bf16-stored operands: every GEMM input below is written as bf16 by its producer (fp32 copies stay for
        # the weight-gradient GEMMs and the LayerNorm / attention backward kernels)
        f32 = dict(dtype=torch.float32, device=x.device)
        fused = fused_ln_enabled() and hid <= 1024      # LayerNorm as the prologue of the GEMM that consumes it
        y1 = y2 = a = x.new_empty(0)         # the fp32 twins are not materialised: every consumer reads bf16
        b16att = bf16_attention_enabled() and dh == 64   # q/k/v stay bf16 from the GEMM epilogue to the attention kernels' LDS-DMA
        qkv = torch.empty(M, 3 * hid, dtype=torch.bfloat16 if b16att else torch.float32, device=x.device)
        if fused:
            y1b = bf16_like(x) if train else None
            m1 = torch.empty(M, **f32) if train else None
            r1 = torch.empty(M, **f32) if train else None
            ln_gemm_bf16(x, n1w, n1b, weight_bf16(wqkv), C=None if b16att else qkv, Cb=qkv if b16att else None, xn=y1b, mean=m1, rstd=r1)
        else:
            pre = _stashed_ln(x, n1w, n1b)       # norm1(x) may have been formed by the kernel that produced x
            if pre is not None:
                y1b, m1, r1 = pre
            else:
                y1b = bf16_like(x)
                _, m1, r1 = layernorm_fwd(x, n1w, n1b, bf16_out=y1b, want_fp32=False)
            gemm_bf16(y1b, weight_bf16(wqkv), M, 3 * hid, hid, C=None if b16att else qkv, Cb=qkv if b16att else None)
        attb = bf16_like(x)
        if b16att:
            att = x.new_empty(0)
            lse = attention_bf16_fwd(qkv, B, L, heads, dh, attb)
        else:
            att, lse = attention_fwd(qkv, B, L, heads, dh, prec, out_bf16=attb)
        x1 = torch.empty(M, hid, **f32)
        gemm_bf16(attb, weight_bf16(wp), M, hid, hid, C=x1, bias=bp, res=x, ldr=hid)
        u = torch.empty(M, mlp, **f32) if train else None      # pre-activation, only GELU' in backward reads it
        ab = torch.empty(M, mlp, dtype=torch.bfloat16, device=x.device)
        if fused:
            y2b = bf16_like(x) if train else None
            m2 = torch.empty(M, **f32) if train else None
            r2 = torch.empty(M, **f32) if train else None
            ln_gemm_bf16(x1, n2w, n2b, weight_bf16(w1), bias=b1, act=1, Cb=ab, pre=u, xn=y2b, mean=m2, rstd=r2)
        else:
            y2b = bf16_like(x)
            _, m2, r2 = layernorm_fwd(x1, n2w, n2b, bf16_out=y2b, want_fp32=False)
            gemm_bf16(y2b, weight_bf16(w1), M, mlp, hid, Cb=ab, bias=b1, act=1, pre=u)
        if not train:
            u = m1 = r1 = m2 = r2 = y1b = y2b = x.new_empty(0)
        x2 = torch.empty(M, hid, **f32)
        if next_ln is not None and not fused and not emit_twin:      # (a tapped block writes the bf16 twin of x2 from its own epilogue)
            xn = bf16_like(x)
            mn, rn = gemm_bf16_ln_fwd(ab, weight_bf16(w2), M, hid, mlp, x2, next_ln[0], next_ln[1], xn, bias=b2, res=x1, ldr=hid)
            _stash_ln(x2, next_ln[0], next_ln[1], xn, mn, rn)
        else:
            # emit_twin: this block's output also feeds a skip-path transposed conv that reads bf16 tokens (its GEMM form):
            # the epilogue writes the bf16 copy next to the fp32 residual stream (was a separate cast launch)
            x2b = bf16_like(x) if emit_twin else None
            gemm_bf16(ab, weight_bf16(w2), M, hid, mlp, C=x2, Cb=x2b, bias=b2, res=x1, ldr=hid)
            if x2b is not None:
                _attach_twin(x2, x2b)
        twins = (y1b, attb, y2b, ab)          # bf16 operands of the weight-gradient GEMMs
    else:
        twins = None
        x3ride = prec == _capi.PREC_BF16X3 and ln_ride_enabled() and x3_ride_ok(M, hid, mlp)
        pre = _stashed_ln(x, n1w, n1b) if x3ride else None      # (bf16x3: norm1(x) may have ridden on the GEMM that produced x)
        y1, m1, r1 = pre if pre is not None else layernorm_fwd(x, n1w, n1b)
        qkv = linear_fwd(y1, wqkv, None, prec)
        att, lse = attention_fwd(qkv, B, L, heads, dh, prec)
        x1 = linear_fwd(att, wp, bp, prec, res=x)
        y2, m2, r2 = layernorm_fwd(x1, n2w, n2b)
        u = torch.empty(M, mlp, dtype=torch.float32, device=x.device)
        a = linear_fwd(y2, w1, b1, prec, act=1, pre=u)
        if x3ride and next_ln is not None:
            # the next block's norm1 rides on the split-K sum of linear2 (as in bf16 mode): no reduce launch, no LayerNorm launch
            x2 = torch.empty(M, hid, dtype=torch.float32, device=x.device)
            xn = torch.empty_like(x2)
            mn, rn = gemm_bf16_ln_fwd(a, w2, M, hid, mlp, x2, next_ln[0], next_ln[1], None, bias=b2, res=x1, ldr=hid, y=xn, b_words=weight_x3(w2))
            _stash_ln(x2, next_ln[0], next_ln[1], xn, mn, rn)
        else:
            x2 = linear_fwd(a, w2, b2, prec, res=x1)
    return x2, (y1, m1, r1, qkv, att, lse, x1, y2, m2, r2, u, a), twins


class TransformerBlockFn(torch.autograd.Function):
    """MONAI TransformerBlock: x + attn(norm1(x)); then + mlp(norm2(.)).  ``ckpt`` = activation checkpointing
    (BASELINE.json config[3]): only the block input is kept and backward recomputes the block's forward kernels first."""

    @staticmethod
    def forward(ctx, x, n1w, n1b, wqkv, wp, bp, n2w, n2b, w1, b1, w2, b2, B, L, heads, prec, ckpt=False, next_n1w=None, next_n1b=None,
                emit_twin=False):
        """next_n1w / next_n1b: weight and bias of the LayerNorm the NEXT block starts with (not differentiated here: that block
        owns its backward) -- its forward is formed by this block's last kernel"""
        _require_gpu(x)
        xc = x.contiguous()
        if xc is not x and hasattr(x, "_unetr_ln"):
            xc._unetr_ln = x._unetr_ln
        x = xc
        train = any(ctx.needs_input_grad[:12])
        x2, acts, twins = _tblock_forward(x, n1w, n1b, wqkv, wp, bp, n2w, n2b, w1, b1, w2, b2, B, L, heads, prec,
                                          train and not ckpt, None if next_n1w is None else (next_n1w, next_n1b), bool(emit_twin))
        ctx.ckpt = bool(ckpt) and train
        if ctx.ckpt:
            acts, twins = (), None
        ctx.twins = twins
        ctx.save_for_backward(x, n1w, wqkv, wp, n2w, w1, w2, n1b, bp, n2b, b1, b2, *acts)
        ctx.meta = (B, L, heads, x.shape[1] // heads, prec)
        return x2

    @staticmethod
    def backward(ctx, dx2):
        x, n1w, wqkv, wp, n2w, w1, w2, n1b, bp, n2b, b1, b2, *acts = ctx.saved_tensors
        B, L, heads, dh, prec = ctx.meta
        twins = ctx.twins
        if ctx.ckpt:
            _, acts, twins = _tblock_forward(x, n1w, n1b, wqkv, wp, bp, n2w, n2b, w1, b1, w2, b2, B, L, heads, prec, True)
        y1, m1, r1, qkv, att, lse, x1, y2, m2, r2, u, a = acts
        M, hid = x.shape
        dx2 = dx2.contiguous()
        mlp = w1.shape[0]
        fast = _bf16_path(prec, hid, mlp) and M % 8 == 0
        f32 = dict(dtype=torch.float32, device=x.device)
        # MLP
        y1b = attb = y2b = ab = dx2b = dub = None
        if fast:
            # data gradients dX = dY . W read W [out, in] as the [K_reduce, N_out] operand (b_kn) -- no transposed copy
            if twins is not None:
                y1b, attb, y2b, ab = twins
            dx2b = _twin(dx2)
            # (du exists as bf16 only: the weight gradient of linear1 is formed from these bf16 values, and so is its bias gradient
            # -- the fp32 copy was 5.3 MB written and read back per block for the column sum alone)
            du, dub = None, torch.empty(M, mlp, dtype=torch.bfloat16, device=x.device)
            gemm_bf16(dx2b, weight_bf16(w2), M, mlp, hid, b_kn=True, Cb=dub, act=2, aux=u, ldaux=mlp)
        else:
            du = linear_dgrad(dx2, w2, prec, aux=u)
        dw2 = wgrad_or_defer(dx2, a, prec, w2, dx2b, ab)
        db2 = colsum_or_defer(dx2, M, hid, hid, b2)
        dw1 = wgrad_or_defer(du, y2, prec, w1, dub, y2b)
        db1 = colsum_or_defer(dub if fast else du, M, mlp, mlp, b1)
        if fast:
            dx1b = bf16_like(x)
            dx1, dn2w, dn2b = gemm_ln_bwd_params(dub, weight_bf16(w1), M, hid, mlp, x1, n2w, n2b, m2, r2, dres=dx2, dx_bf16=dx1b)
        else:
            dx1b = None
            if prec == _capi.PREC_BF16X3 and ln_ride_enabled() and x3_ride_ok(M, hid, mlp):
                # (bf16x3: LayerNorm backward sums the K slabs of the data-gradient GEMM itself, as in bf16 mode)
                dx1, dn2w, dn2b = gemm_ln_bwd_params(du, w1, M, hid, mlp, x1, n2w, n2b, m2, r2, dres=dx2, b_words=weight_x3(w1))
            else:
                dy2 = linear_dgrad(du, w1, prec)
                dx1, dn2w, dn2b = layernorm_bwd_params(dy2, x1, n2w, n2b, m2, r2, dres=dx2, dx_bf16=dx1b)
        # attention
        b16att = fast and qkv.dtype == torch.bfloat16
        if b16att:
            dattb = bf16_like(x)
            gemm_bf16(dx1b, weight_bf16(wp), M, hid, hid, b_kn=True, Cb=dattb)
        elif fast:
            datt = torch.empty(M, hid, **f32)
            gemm_bf16(dx1b, weight_bf16(wp), M, hid, hid, b_kn=True, C=datt)
        else:
            datt = linear_dgrad(dx1, wp, prec)
        dwp = wgrad_or_defer(dx1, att, prec, wp, dx1b, attb)
        dbp = colsum_or_defer(dx1, M, hid, hid, bp)
        if b16att:
            dqkv = None
            dqkvb = attention_bf16_bwd(qkv, attb, dattb, lse, B, L, heads, dh)
        else:
            dqkvb = torch.empty(M, 3 * hid, dtype=torch.bfloat16, device=x.device) if fast else None
            dqkv = attention_bwd(qkv, att, datt, lse, B, L, heads, dh, prec, dqkv_bf16=dqkvb)
        dwqkv = wgrad_or_defer(dqkv, y1, prec, wqkv, dqkvb, y1b)
        if fast:
            dxb = bf16_like(x)
            dx, dn1w, dn1b = gemm_ln_bwd_params(dqkvb, weight_bf16(wqkv), M, hid, 3 * hid, x, n1w, n1b, m1, r1, dres=dx1, dx_bf16=dxb)
        else:
            dxb = None
            if prec == _capi.PREC_BF16X3 and ln_ride_enabled() and x3_ride_ok(M, hid, 3 * hid):
                dx, dn1w, dn1b = gemm_ln_bwd_params(dqkv, wqkv, M, hid, 3 * hid, x, n1w, n1b, m1, r1, dres=dx1, b_words=weight_x3(wqkv))
            else:
                dy1 = linear_dgrad(dqkv, wqkv, prec)
                dx, dn1w, dn1b = layernorm_bwd_params(dy1, x, n1w, n1b, m1, r1, dres=dx1, dx_bf16=dxb)
        if fast:
            _attach_twin(dx, dxb)      # the block below picks its bf16 operand up from here (functional._twin)
        return (dx, dn1w, dn1b, dwqkv, dwp, dbp, dn2w, dn2b, dw1, db1, dw2, db2, None, None, None, None, None, None, None, None)


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, twin=False):
        _require_gpu(x)
        x = x.contiguous()
        yb = bf16_like(x) if twin else None       # (twin: decoder5's transposed conv reads bf16 tokens in its GEMM form)
        y, mean, rstd = layernorm_fwd(x, w, b, bf16_out=yb)
        if yb is not None:
            _attach_twin(y, yb)
        ctx.save_for_backward(x, w, mean, rstd, b)
        ctx.twin = bool(twin)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, rstd, b = ctx.saved_tensors
        dxb = bf16_like(x) if ctx.twin else None    # bf16 operand for the last transformer block's backward GEMMs
        dx, dw, db = layernorm_bwd_params(dy.contiguous(), x, w, b, mean, rstd, dx_bf16=dxb)
        if dxb is not None:
            _attach_twin(dx, dxb)
        return dx, dw, db, None


# ------------------------------------------------------------------------------ conv-side building blocks
def skip_half(dims, C, prec, device):
    """A fresh concatenation buffer [B,D,H,W,2C] for UnetrUpBlock's torch.cat((up, skip), dim=1), and the view of its second
    half: the producer of the skip tensor writes there directly (``to_cat``) and UpBlockFn(skip_in_cat=True) writes the
    transposed conv into the first half -- the concatenation then costs nothing (it was a 113 MB copy at 96^3)."""
    cat = torch.empty(*dims, 2 * C, dtype=act_dtype(prec), device=device)
    return cat, cat[..., C:]


def cat_of_skip(skip, C):
    """the whole concatenation buffer whose second half ``skip`` is (see skip_half); raises when it is not such a view"""
    B, D, H, W, c = skip.shape
    want = (D * H * W * 2 * C, H * W * 2 * C, W * 2 * C, 2 * C, 1)
    if c != C or tuple(skip.stride()) != want or skip.storage_offset() != C or skip.untyped_storage().nbytes() != B * want[0] * skip.element_size():
        raise RuntimeError("skip_in_cat=True needs the second half of a buffer made by functional.skip_half")
    return skip.as_strided((B, D, H, W, 2 * C), want, 0)


def _resblock_fwd(x, ldx, dims, cin, cout, w1, w2, w3, prec, to_cat=False, head=None):
    """MONAI UnetResBlock (instance norm, in != out): lrelu(IN(conv2(lrelu(IN(conv1 x)))) + IN(conv3 x)).
    head = (wo, bo) of the out conv that is the block's only consumer: returns (logits [B, Cout, V] or out, saved, fused) -- fused:
    the block end was folded into the out conv and never stored (the caller applies the out conv itself otherwise)."""
    B, D, H, W = dims
    V = D * H * W
    if in_fuse_level() & 1:
        r = _resblock_fwd_fin(x, ldx, dims, cout, w1, w2, w3, prec, to_cat, head if out_fuse_enabled() else None)
        if r is not None:
            return r if head is not None else r[:2]
    f1 = conv3_fused(x, ldx, w1, w3, dims, prec)
    if f1 is not None:
        c1, s1, c3, s3 = f1
    else:
        f1 = conv3_fused(x, ldx, w1, None, dims, prec)
        if f1 is not None:
            c1, s1 = f1[0], f1[1]
        else:
            c1 = conv3(x, ldx, w1, dims, prec)
            s1 = instnorm_stats(c1, cout, B, V, cout)
        # 1x1x1 conv on the generic GEMM family (fp32 storage: cast at the boundary in bf16 mode)
        x32 = x if x.dtype == torch.float32 else x.float()
        c3 = torch.empty(B, D, H, W, cout, dtype=torch.float32, device=x.device)
        gemm(x32, w3, c3, B * V, cout, cin, lda=x32.stride(-2), ldb=cin, ldc=cout, prec=prec)
        c3 = _as_act(c3, prec)
        s3 = instnorm_stats(c3, cout, B, V, cout)
    a1 = instnorm_apply(c1, s1, B, V, cout, True)
    f2 = conv3_fused(a1, cout, w2, None, dims, prec)
    if f2 is not None:
        c2, s2 = f2[0], f2[1]
    else:
        c2 = conv3(a1, cout, w2, dims, prec)
        s2 = instnorm_stats(c2, cout, B, V, cout)
    if to_cat:
        _, half = skip_half(dims, cout, prec, x.device)
        out = instnorm_apply(c2, s2, B, V, cout, True, x2=c3, sb=s3, out=half, ldo=2 * cout)
    else:
        out = instnorm_apply(c2, s2, B, V, cout, True, x2=c3, sb=s3)
    if head is not None:
        return out, (c1, s1, a1, c2, s2, c3, s3), False
    return out, (c1, s1, a1, c2, s2, c3, s3)


def _materialize_c3(x, w3, dims, cin, cout, prec):
    """the 1x1x1 branch of the block on the image as a stored tensor (fallback of the image form: generic GEMM family)"""
    B, D, H, W = dims
    V = D * H * W
    c3f = torch.empty(B * V, cout, dtype=torch.float32, device=x.device)
    gemm(x.reshape(B * V, cin), w3, c3f, B * V, cout, cin, lda=cin, ldb=cin, ldc=cout, prec=prec)
    return _as_act(c3f.view(B, D, H, W, cout), prec)


def _resblock_fwd_fin(x, ldx, dims, cout, w1, w2, w3, prec, to_cat, head=None):
    """the same block in FOUR launches: both convs leave their InstanceNorm sums as partial rows and the two apply kernels form
    the statistics in their prologues (no finalize launches); None when a shape declines (the caller takes the route above).
    Returns (out, saved, fused); head = (wo, bo): the block end goes into the out conv's kernel when that accepts the shape."""
    B, D, H, W = dims
    V = D * H * W
    cin = w1.shape[1]
    # the block on the image (<= 4 fp32 input channels, no input gradient): its 1x1x1 branch is never stored -- the block-end
    # kernels form it from the image (csrc/norm_misc.hip: ImgBranch)
    img = (img_branch_enabled() and x.dtype == torch.float32 and cin <= 4 and ldx == cin and not x.requires_grad and w3.is_contiguous()
           and cout % 8 == 0 and 64 % max(1, cout // 8) == 0)
    f1 = conv3_parts(x, ldx, w1, w3, dims, prec, store3=not img)
    if f1 is None and img:
        img = False
        f1 = conv3_parts(x, ldx, w1, w3, dims, prec)
    if f1 is None:
        return None
    c1, p1, c3, p3, rows1 = f1
    r1 = instnorm_apply_fin(c1, p1, rows1, B, V, cout, True)
    if r1 is None:
        return None
    a1, s1, _ = r1
    f2 = conv3_parts(a1, cout, w2, None, dims, prec)
    if f2 is None:
        return None
    c2, p2, _, _, rows2 = f2
    if head is not None and not img and c3 is not None and not to_cat:
        r = outconv_in_fwd(c2, p2, rows2, c3, p3, rows1, head[0], head[1], B, V, cout)
        if r is not None:
            return r[0].view(B, head[0].shape[0], D, H, W), (c1, s1, a1, c2, r[1], c3, r[2]), True
    half = skip_half(dims, cout, prec, x.device)[1] if to_cat else None
    if img:
        r2 = instnorm_apply_fin_img(c2, p2, rows2, x, cin, w3, p3, rows1, B, V, cout, out=half, ldo=2 * cout if to_cat else None)
        if r2 is None:                         # (a shape the image form declines: materialise the branch after all)
            c3 = _materialize_c3(x, w3, dims, cin, cout, prec)
            r2 = instnorm_apply_fin(c2, p2, rows2, B, V, cout, True, x2=c3, part_b=p3, rows_b=rows1, out=half, ldo=2 * cout if to_cat else None)
    else:
        r2 = instnorm_apply_fin(c2, p2, rows2, B, V, cout, True, x2=c3, part_b=p3, rows_b=rows1, out=half, ldo=2 * cout if to_cat else None)
    if r2 is None:
        return None
    out, s2, s3 = r2
    return out, (c1, s1, a1, c2, s2, c3, s3), False


def _resblock_bwd(dout, x, ldx, dims, cin, cout, w1, w2, w3, saved, prec, need_dx, head=None):
    """head = (wo, bo): the block end was folded into the out conv (``_resblock_fwd(..., head=)`` returned fused): ``dout`` is the
    gradient of the LOGITS; the result then ends with (dwo, dbo)"""
    B, D, H, W = dims
    V = D * H * W
    c1, s1, a1, c2, s2, c3, s3 = saved
    head_grads = ()
    dc23 = None
    if head is not None:
        dout, hpart, hrows, dwo, dbo = outconv_in_bwd(dout, c2, s2, c3, s3, head[0], head[1], B, V, cout)
        head_grads = (dwo, dbo)
        dc23 = instnorm_bwd_apply_fin_dual(dout, cout, c2, s2, c3, s3, hpart, hrows, B, V, cout)
    dout, lddo = _rows(dout)
    img3 = None
    if dc23 is not None:
        pass
    elif c3 is None:
        # the block on the image: the 1x1x1 branch was never stored -- its gradient is formed per voxel inside the backward apply and
        # leaves only as partial rows of dw3
        img3 = instnorm_bwd_img(dout, lddo, c2, s2, x, cin, w3, s3, B, V, cout)
        if img3 is None:
            c3 = _materialize_c3(x, w3, dims, cin, cout, prec)
    if dc23 is not None:
        dc2, dc3 = dc23
    elif img3 is not None:
        dc2, dc3 = img3[0], None
    else:
        dc2, dc3 = instnorm_bwd(dout, lddo, c2, s2, B, V, cout, True, x2=c3, sb=s3)
    # conv3 (1x1x1)
    g3 = _gout(w3)
    dw3 = g3 if g3 is not None else torch.empty(cout, cin, 1, 1, 1, dtype=torch.float32, device=x.device)
    # arena mode: the three weight gradients' partial-sum reductions join the grouped launch at the end of the backward pass
    g1, g2 = _gout(w1), _gout(w2)
    st = _GRAD_SINK.get(w1.data_ptr()) if (g1 is not None and g2 is not None and g3 is not None) else None
    if st is not None and (_GRAD_SINK.get(w2.data_ptr()) is not st or _GRAD_SINK.get(w3.data_ptr()) is not st):
        st = None
    q1 = q2 = False
    # conv2
    if st is not None:
        dw2, q2 = conv3_wgrad(a1, cout, dc2, cout, dims, cout, cout, prec, out=g2, defer=st)
    else:
        dw2 = conv3_wgrad(a1, cout, dc2, cout, dims, cout, cout, prec, out=g2)
    dc1 = None
    if in_fuse_level() & 2:
        # the backward sums of the first norm come out of the data-gradient conv's epilogue: no reduction pass over (da1, c1)
        f = conv3_dgrad_stats(dc2, w2, c1, s1, dims, prec)
        if f is not None:
            da1, bpart, brows = f
            dc1 = instnorm_bwd_apply_fin(da1, cout, c1, s1, bpart, brows, 2, B, V, cout, True)
    if dc1 is None:
        da1 = conv3(dc2, cout, w2, dims, prec, mode=1)
        dc1, _ = instnorm_bwd(da1, cout, c1, s1, B, V, cout, True)
    o3 = dw3 if dc3 is not None else None
    if st is not None:
        dw1, q1 = conv3_wgrad(x, ldx, dc1, cout, dims, cin, cout, prec, out=g1, dy3=dc3, out3=o3, defer=st)
    else:
        dw1 = conv3_wgrad(x, ldx, dc1, cout, dims, cin, cout, prec, out=g1, dy3=dc3, out3=o3)
    if img3 is not None:
        rq = [(img3[1], dw3, cout * cin, img3[2])]
        if st is not None and q1 and reduce_defer_enabled():
            st.defer["reduce"].append(rq[0])           # joins the grouped reduce at the end of the backward pass (armed by conv3_wgrad)
        else:
            _launch_reduces(rq)
    dx = None
    if need_dx:
        dx = torch.empty(B, D, H, W, cin, dtype=act_dtype(prec), device=x.device)
        if not conv3_dgrad_fused(dc1, dc3, w1, w3, dx, dims, prec):
            d32 = torch.empty(B, D, H, W, cin, dtype=torch.float32, device=x.device)
            gemm(dc3.float() if dc3.dtype != torch.float32 else dc3, w3, d32, B * V, cin, cout, lda=cout, ldb=cin, ldc=cin, prec=prec, b_trans=True)
            dx.copy_(d32)
            conv3(dc1, cout, w1, dims, prec, mode=1, out=dx, ldo=cin, accumulate=True)
    if head is not None:
        return dx, dw1, dw2, dw3, (q1, q2), head_grads
    return dx, dw1, dw2, dw3, (q1, q2)


class ResBlockFn(torch.autograd.Function):
    """UnetrBasicBlock(res_block=True) = one UnetResBlock (encoder1, unetr.py:90-98)."""

    @staticmethod
    def forward(ctx, x, w1, w2, w3, prec, to_cat=False):
        """to_cat: the result is the second half of a fresh concatenation buffer (functional.skip_half)"""
        _require_gpu(x, act=True)
        x, ldx = _rows(x)
        B, D, H, W, cin = x.shape
        cout = w1.shape[0]
        out, saved = _resblock_fwd(x, ldx, (B, D, H, W), cin, cout, w1, w2, w3, prec, to_cat)
        ctx.save_for_backward(x, w1, w2, w3, *saved)
        ctx.meta = (ldx, (B, D, H, W), cin, cout, prec)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w1, w2, w3, *saved = ctx.saved_tensors
        ldx, dims, cin, cout, prec = ctx.meta
        dx, dw1, dw2, dw3, (q1, q2) = _resblock_bwd(dout, x, ldx, dims, cin, cout, w1, w2, w3, saved, prec, ctx.needs_input_grad[0])
        return dx, _ret(w1, dw1, deferred=q1), _ret(w2, dw2, deferred=q2), _ret(w3, dw3, deferred=q1), None, None


class TconvFn(torch.autograd.Function):
    """2x2x2 stride-2 ConvTranspose3d, bias=False (UnetrPrUpBlock with conv_block=False, unetr.py:99-134)."""

    @staticmethod
    def forward(ctx, x, w, prec, to_cat=False):
        """to_cat: the result is the second half of a fresh concatenation buffer (functional.skip_half)"""
        _require_gpu(x, act=True)
        x, ldx = _rows(x)
        B, D, H, W, cin = x.shape
        cout = w.shape[1]
        if to_cat:
            _, y = skip_half((B, 2 * D, 2 * H, 2 * W), cout, prec, x.device)
            _, xb = tconv_fwd(x, ldx, w, (B, D, H, W), cin, cout, prec, out=y, ldo=2 * cout)
        else:
            y, xb = tconv_fwd(x, ldx, w, (B, D, H, W), cin, cout, prec)
        ctx.save_for_backward(x, w)
        ctx.xb = xb
        ctx.meta = (ldx, (B, D, H, W), cin, cout, prec)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        ldx, dims, cin, cout, prec = ctx.meta
        dy, lddy = _rows(dy)
        dx, dw = tconv_bwd(x, ldx, ctx.xb, dy, lddy, w, dims, cin, cout, prec, ctx.needs_input_grad[0])
        return dx, dw, None, None


def _outconv_fwd(x, ldx, w, b):
    B, D, H, W, cin = x.shape
    cout = w.shape[0]
    logits = torch.empty(B, cout, D, H, W, dtype=torch.float32, device=x.device)
    call("unetr_outconv_fwd", x.data_ptr(), ldx, w.data_ptr(), b.data_ptr(), logits.data_ptr(), B, D * H * W, cin, cout, _a16(x), _stream())
    return logits


def _outconv_bwd(dl, x, ldx, w, b):
    B, D, H, W, cin = x.shape
    cout = w.shape[0]
    dl = dl.contiguous()
    dx = torch.empty(B, D, H, W, cin, dtype=x.dtype, device=x.device)
    gw, gb = _gout(w), _gout(b)
    dw = gw if gw is not None else torch.empty_like(w)
    db = gb if gb is not None else torch.empty(cout, dtype=torch.float32, device=x.device)
    ws = workspace(x.device)
    call("unetr_outconv_bwd", dl.data_ptr(), x.data_ptr(), ldx, w.data_ptr(), dx.data_ptr(), cin, dw.data_ptr(), db.data_ptr(),
         B, D * H * W, cin, cout, ws.data_ptr(), ws.numel() * 4, _a16(x), _stream())
    return dx, dw, db


class UpBlockFn(torch.autograd.Function):
    """MONAI UnetrUpBlock(res_block=True): tconv(inp) -> cat(up, skip) -> UnetResBlock (unetr.py:135-174).
    The transposed conv writes straight into the first half of the concatenation buffer.
    With (wo, bo) -- the weights of the UnetOutBlock that is this block's only consumer (decoder2 -> out, unetr.py:165-175,206-207)
    -- the Function returns the LOGITS: the block end lrelu(IN(c2) + IN(c3)) is formed inside the out conv's kernels and never
    stored (forward: one tensor write + one read fewer; backward: the InstanceNorm backward reduction pass disappears)."""

    @staticmethod
    def forward(ctx, inp, skip, wt, w1, w2, w3, prec, skip_in_cat=False, wo=None, bo=None):
        """skip_in_cat: ``skip`` already is the second half of a concatenation buffer made by functional.skip_half (its
        producer ran with to_cat=True): no copy"""
        _require_gpu(inp, act=True)
        inp, ldi = _rows(inp)
        B, D, H, W, cin = inp.shape
        C = wt.shape[1]
        dims2 = (B, 2 * D, 2 * H, 2 * W)
        rows2 = B * 8 * D * H * W
        if skip_in_cat and skip.dtype == act_dtype(prec):
            cat = cat_of_skip(skip, C)
            _, ctx.xb = tconv_fwd(inp, ldi, wt, (B, D, H, W), cin, C, prec, out=cat, ldo=2 * C)
        else:
            cat = torch.empty(*dims2, 2 * C, dtype=act_dtype(prec), device=inp.device)
            skip, lds = _rows(_as_act(skip, prec))
            _, ctx.xb = tconv_fwd(inp, ldi, wt, (B, D, H, W), cin, C, prec, out=cat, ldo=2 * C)
            call("unetr_copy_rows", cat.data_ptr() + cat.element_size() * C, 2 * C, skip.data_ptr(), lds, rows2, C, 0, _a16(cat), _stream())
        ctx.meta = (ldi, (B, D, H, W), cin, C, prec)
        if wo is None:
            out, saved = _resblock_fwd(cat, 2 * C, dims2, 2 * C, C, w1, w2, w3, prec)
            ctx.head = None
            ctx.save_for_backward(inp, wt, w1, w2, w3, cat, *saved)
            return out
        res, saved, fused = _resblock_fwd(cat, 2 * C, dims2, 2 * C, C, w1, w2, w3, prec, head=(wo, bo))
        ctx.head = fused
        if fused:
            ctx.save_for_backward(inp, wt, w1, w2, w3, cat, wo, bo, *saved)
            return res
        ctx.save_for_backward(inp, wt, w1, w2, w3, cat, wo, bo, res, *saved)
        return _outconv_fwd(res, C, wo, bo)

    @staticmethod
    def backward(ctx, dout):
        ldi, dims, cin, C, prec = ctx.meta
        B, D, H, W = dims
        dims2 = (B, 2 * D, 2 * H, 2 * W)
        head_ret = (None, None)
        if ctx.head is None:
            inp, wt, w1, w2, w3, cat, *saved = ctx.saved_tensors
            dcat, dw1, dw2, dw3, (q1, q2) = _resblock_bwd(dout, cat, 2 * C, dims2, 2 * C, C, w1, w2, w3, saved, prec, True)
        elif ctx.head:
            inp, wt, w1, w2, w3, cat, wo, bo, *saved = ctx.saved_tensors
            dcat, dw1, dw2, dw3, (q1, q2), (dwo, dbo) = _resblock_bwd(dout, cat, 2 * C, dims2, 2 * C, C, w1, w2, w3, saved, prec, True, head=(wo, bo))
            head_ret = (_ret(wo, dwo), _ret(bo, dbo))
        else:
            inp, wt, w1, w2, w3, cat, wo, bo, out, *saved = ctx.saved_tensors
            dblk, dwo, dbo = _outconv_bwd(dout, out, C, wo, bo)
            dcat, dw1, dw2, dw3, (q1, q2) = _resblock_bwd(dblk, cat, 2 * C, dims2, 2 * C, C, w1, w2, w3, saved, prec, True)
            head_ret = (_ret(wo, dwo), _ret(bo, dbo))
        dinp, dwt = tconv_bwd(inp, ldi, ctx.xb, dcat, 2 * C, wt, dims, cin, C, prec, ctx.needs_input_grad[0])
        dskip = dcat[..., C:] if ctx.needs_input_grad[1] else None
        return (dinp, dskip, dwt, _ret(w1, dw1, deferred=q1), _ret(w2, dw2, deferred=q2), _ret(w3, dw3, deferred=q1), None, None) + head_ret


class OutConvFn(torch.autograd.Function):
    """MONAI UnetOutBlock: 1x1x1 conv with bias; emits NCDHW logits (unetr.py:175,207)."""

    @staticmethod
    def forward(ctx, x, w, b):
        _require_gpu(x, act=True)
        x, ldx = _rows(x)
        ctx.save_for_backward(x, w, b)
        ctx.ldx = ldx
        return _outconv_fwd(x, ldx, w, b)

    @staticmethod
    def backward(ctx, dl):
        x, w, b = ctx.saved_tensors
        dx, dw, db = _outconv_bwd(dl, x, ctx.ldx, w, b)
        return dx, _ret(w, dw), _ret(b, db)


class ToNCDHWFn(torch.autograd.Function):
    """channels-last [B,D,H,W,C] -> torch NCDHW [B,C,D,H,W] (the layout enc4 is returned in, unetr.py:208)."""

    @staticmethod
    def forward(ctx, x):
        _require_gpu(x, act=True)
        x, ldx = _rows(x)
        B, D, H, W, C = x.shape
        y = torch.empty(B, C, D, H, W, dtype=torch.float32, device=x.device)
        call("unetr_nhwc_to_nchw", x.data_ptr(), ldx, y.data_ptr(), B, C, D * H * W, 0, _a16(x), _stream())
        ctx.adt = x.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        B, C, D, H, W = dy.shape
        dx = torch.empty(B, D, H, W, C, dtype=ctx.adt, device=dy.device)
        call("unetr_nchw_to_nhwc", dy.data_ptr(), dx.data_ptr(), C, B, C, D * H * W, _a16(dx), _stream())
        return dx


def to_channels_last(x_in):
    """NCDHW image -> channels-last (no gradient: the image is a leaf input). C == 1 is a free view."""
    _require_gpu(x_in)
    B, C, D, H, W = x_in.shape
    x_in = x_in.contiguous()
    if C == 1:
        return x_in.view(B, D, H, W, 1)
    y = torch.empty(B, D, H, W, C, dtype=torch.float32, device=x_in.device)      # (the image stays fp32 in every mode)
    call("unetr_nchw_to_nhwc", x_in.data_ptr(), y.data_ptr(), C, B, C, D * H * W, 0, _stream())
    return y


class DiceCEFn(torch.autograd.Function):
    """DiceCELoss -> (loss, dice, ce) as a 3-vector.  multilabel=False: DiceCELoss(to_onehot_y=True, softmax=True)
    (unetr_segmentation_3d.py:404); multilabel=True: DiceCELoss(to_onehot_y=False, sigmoid=True) (:477-482)."""

    @staticmethod
    def forward(ctx, logits, label, smooth_nr, smooth_dr, multilabel=False, scalar=False):
        """scalar=True: returns the 0-dim loss itself (what DiceCELoss.forward hands to backward()): selecting element 0 of the
        3-vector through autograd cost a fill, a zero and a copy launch in every backward pass"""
        _require_gpu(logits)
        logits = logits.contiguous()
        label = label.contiguous().to(torch.float32)
        B, C = logits.shape[0], logits.shape[1]
        V = logits.numel() // (B * C)
        if multilabel and label.numel() != B * C * V:
            raise ValueError(f"multi-label target must be [B,C,*spatial] like the logits, got {tuple(label.shape)}")
        if not multilabel and label.numel() != B * V:
            raise ValueError(f"label must be [B,1,*spatial] class indices, got {tuple(label.shape)}")
        out = torch.empty(3, dtype=torch.float32, device=logits.device)
        coef = torch.empty(B * C * 2, dtype=torch.float32, device=logits.device)
        ws = workspace(logits.device)
        call("unetr_dicece_fwd", logits.data_ptr(), label.data_ptr(), B, C, V, int(multilabel), smooth_nr, smooth_dr, out.data_ptr(),
             coef.data_ptr(), ws.data_ptr(), ws.numel() * 4, _stream())
        ctx.save_for_backward(logits, label, coef)
        ctx.meta = (B, C, V, int(multilabel))
        ctx.scalar = bool(scalar)
        return out[0] if scalar else out

    @staticmethod
    def backward(ctx, dout):
        logits, label, coef = ctx.saved_tensors
        B, C, V, ml = ctx.meta
        # only out[0] (= dice + ce) is differentiable; out[1], out[2] are reporting copies
        dloss = dout.reshape(1) if ctx.scalar else dout[0:1]
        dloss = dloss.contiguous()
        dlogits = torch.empty_like(logits)
        call("unetr_dicece_bwd", logits.data_ptr(), label.data_ptr(), coef.data_ptr(), dloss.data_ptr(), dlogits.data_ptr(), B, C, V,
             ml, _stream())
        return dlogits, None, None, None, None, None
