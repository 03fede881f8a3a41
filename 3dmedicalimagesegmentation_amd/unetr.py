"""MI355X-native UNETR behind the reference's own interface.

Drop-in for ``/root/reference/unetr.py:21-208`` (``class UNETR(nn.Module)``): same constructor signature,
same three constructor exceptions (unetr.py:60-67), same attributes (unetr.py:69-77), same
``forward(x_in, freeze_encoder=False) -> (enc4, logits)`` (unetr.py:182-208), and the same ``state_dict``
keys / shapes as the MONAI 0.6.0 blocks the reference instantiates (unetr.py:78-175), so checkpoints written
by either side load strictly into the other (unetr_segmentation_3d.py:516-518).

The sub-modules below are *parameter holders* (torch layers used only for their shapes, default
initialisation and names); their ``forward`` is never called.  All arithmetic goes through the HIP kernels
in libunetr_hip.so via ``functional``.  Tensors must be on a ROCm device; there is no CPU fallback.

``UNETRLogits`` is the ``monai.networks.nets.UNETR`` call convention used by
unetr_segmentation_3d.py:221 / :109 (forward returns logits only) over the same parameters.
"""
import os
from typing import Tuple, Union

import torch
import torch.nn as nn

try:
    from . import _capi
    from . import functional as Fn
except ImportError:
    # imported as the top-level module `unetr` -- the reference scripts' own `from unetr import UNETR`
    # (unetr_ranking_pretraining_3d.py:34) with this directory on sys.path: load the package by its real name
    import importlib
    import sys
    _here = os.path.dirname(os.path.abspath(__file__))
    if os.path.dirname(_here) not in sys.path:
        sys.path.insert(0, os.path.dirname(_here))
    _pkg = importlib.import_module(os.path.basename(_here))
    _capi, Fn = _pkg._capi, _pkg.functional

_PRECISIONS = {"fp32": _capi.PREC_F32, "bf16": _capi.PREC_BF16, "bf16x3": _capi.PREC_BF16X3}


def default_precision() -> str:
    return os.environ.get("UNETR_AMD_PRECISION", "fp32")


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder: arithmetic runs in the HIP kernels, call UNETR.forward")


class _Rearrange(_Holder):
    """stands in for einops Rearrange at Sequential index 0 so the Linear keeps the key 'patch_embeddings.1.*'"""


def _trunc_normal_linear(m: nn.Linear):
    nn.init.trunc_normal_(m.weight, mean=0.0, std=0.02, a=-2.0, b=2.0)
    if m.bias is not None:
        nn.init.constant_(m.bias, 0)


class _PatchEmbedding(_Holder):
    def __init__(self, in_channels, img_size, patch_size, hidden_size):
        super().__init__()
        n_patches = 1
        for i, p in zip(img_size, patch_size):
            if i < p:
                raise AssertionError("patch_size should be smaller than img_size.")
            if i % p != 0:
                raise AssertionError("img_size should be divisible by patch_size for perceptron patch embedding.")
            n_patches *= i // p
        patch_dim = in_channels * patch_size[0] * patch_size[1] * patch_size[2]
        self.patch_embeddings = nn.Sequential(_Rearrange(), nn.Linear(patch_dim, hidden_size))
        self.position_embeddings = nn.Parameter(torch.zeros(1, n_patches, hidden_size))
        self.cls_token = nn.Parameter(torch.zeros(1, 1, hidden_size))  # registered by MONAI, never used
        nn.init.trunc_normal_(self.position_embeddings, mean=0.0, std=0.02, a=-2.0, b=2.0)
        _trunc_normal_linear(self.patch_embeddings[1])


class _SA(_Holder):
    def __init__(self, hidden_size):
        super().__init__()
        self.out_proj = nn.Linear(hidden_size, hidden_size)
        self.qkv = nn.Linear(hidden_size, hidden_size * 3, bias=False)


class _MLP(_Holder):
    def __init__(self, hidden_size, mlp_dim):
        super().__init__()
        self.linear1 = nn.Linear(hidden_size, mlp_dim)
        self.linear2 = nn.Linear(mlp_dim, hidden_size)


class _TransformerBlock(_Holder):
    def __init__(self, hidden_size, mlp_dim):
        super().__init__()
        self.mlp = _MLP(hidden_size, mlp_dim)
        self.norm1 = nn.LayerNorm(hidden_size)
        self.attn = _SA(hidden_size)
        self.norm2 = nn.LayerNorm(hidden_size)


class _ViT(_Holder):
    def __init__(self, in_channels, img_size, patch_size, hidden_size, mlp_dim, num_layers):
        super().__init__()
        self.patch_embedding = _PatchEmbedding(in_channels, img_size, patch_size, hidden_size)
        self.blocks = nn.ModuleList([_TransformerBlock(hidden_size, mlp_dim) for _ in range(num_layers)])
        self.norm = nn.LayerNorm(hidden_size)


class _Conv(nn.Sequential):
    """monai Convolution(conv_only=True): Sequential with one child named 'conv'."""

    def __init__(self, in_ch, out_ch, k, transposed=False, bias=False):
        super().__init__()
        if transposed:
            self.add_module("conv", nn.ConvTranspose3d(in_ch, out_ch, k, k, 0, 0, bias=bias))
        else:
            self.add_module("conv", nn.Conv3d(in_ch, out_ch, k, 1, (k - 1) // 2, bias=bias))


class _ResBlock(_Holder):
    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.conv1 = _Conv(in_ch, out_ch, 3)
        self.conv2 = _Conv(out_ch, out_ch, 3)
        self.conv3 = _Conv(in_ch, out_ch, 1)


class _BasicBlock(_Holder):
    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.layer = _ResBlock(in_ch, out_ch)


class _PrUpBlock(_Holder):
    def __init__(self, in_ch, out_ch, num_layer):
        super().__init__()
        self.transp_conv_init = _Conv(in_ch, out_ch, 2, transposed=True)
        self.blocks = nn.ModuleList([_Conv(out_ch, out_ch, 2, transposed=True) for _ in range(num_layer)])


class _UpBlock(_Holder):
    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.transp_conv = _Conv(in_ch, out_ch, 2, transposed=True)
        self.conv_block = _ResBlock(out_ch + out_ch, out_ch)


class _OutBlock(_Holder):
    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.conv = _Conv(in_ch, out_ch, 1, bias=True)


class UNETR(nn.Module):
    """
    UNETR based on: "Hatamizadeh et al.,
    UNETR: Transformers for 3D Medical Image Segmentation <https://arxiv.org/abs/2103.10504>"
    (same constructor as the reference, unetr.py:27-41)
    """

    def __init__(
        self,
        in_channels: int,
        out_channels: int,
        img_size: Tuple[int, int, int],
        feature_size: int,
        hidden_size: int,
        mlp_dim: int,
        num_heads: int,
        pos_embed: str,
        norm_name: Union[Tuple, str],
        conv_block: bool = False,
        res_block: bool = False,
        dropout_rate: float = 0.0,
    ) -> None:
        super().__init__()

        if not (0 <= dropout_rate <= 1):
            raise AssertionError("dropout_rate should be between 0 and 1.")

        if hidden_size % num_heads != 0:
            raise AssertionError("hidden size should be divisible by num_heads.")

        if pos_embed not in ["conv", "perceptron"]:
            raise KeyError(f"Position embedding layer of type {pos_embed} is not supported.")

        # the one combination both reference scripts instantiate (unetr_segmentation_3d.py:501-513,
        # unetr_ranking_pretraining_3d.py:450-462) is what the HIP path implements
        nname = norm_name[0] if isinstance(norm_name, (tuple, list)) else norm_name
        unsupported = []
        if pos_embed != "perceptron":
            unsupported.append(f"pos_embed={pos_embed!r}")
        if str(nname).lower() != "instance":
            unsupported.append(f"norm_name={norm_name!r}")
        if not res_block:
            unsupported.append("res_block=False")
        if conv_block:
            unsupported.append("conv_block=True")
        if dropout_rate != 0.0:
            unsupported.append(f"dropout_rate={dropout_rate}")
        if in_channels == feature_size:
            unsupported.append("in_channels == feature_size (identity residual)")
        if unsupported:
            raise NotImplementedError("HIP UNETR supports pos_embed='perceptron', norm_name='instance', res_block=True, "
                                      "conv_block=False, dropout_rate=0.0; got " + ", ".join(unsupported))

        self.num_layers = 12
        self.patch_size = (16, 16, 16)
        self.feat_size = (
            img_size[0] // self.patch_size[0],
            img_size[1] // self.patch_size[1],
            img_size[2] // self.patch_size[2],
        )
        self.hidden_size = hidden_size
        self.classification = False
        self.num_heads = num_heads
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.img_size = tuple(img_size)
        self.precision = default_precision()
        # config[3] of BASELINE.json ("activation checkpointing on encoder"): when True the transformer blocks keep only
        # their input for backward and recompute their forward there (functional.TransformerBlockFn)
        self.encoder_checkpointing = False

        f = feature_size
        self.vit = _ViT(in_channels, img_size, self.patch_size, hidden_size, mlp_dim, self.num_layers)
        # (MONAI leaves the transformer Linears at torch's default init; only PatchEmbeddingBlock applies trunc-normal)
        self.encoder1 = _BasicBlock(in_channels, f)
        self.encoder2 = _PrUpBlock(hidden_size, f * 2, 2)
        self.encoder3 = _PrUpBlock(hidden_size, f * 4, 1)
        self.encoder4 = _PrUpBlock(hidden_size, f * 8, 0)
        self.decoder5 = _UpBlock(hidden_size, f * 8)
        self.decoder4 = _UpBlock(f * 8, f * 4)
        self.decoder3 = _UpBlock(f * 4, f * 2)
        self.decoder2 = _UpBlock(f * 2, f)
        self.out = _OutBlock(f, out_channels)  # type: ignore

    # ------------------------------------------------------------------------------------------------
    def use_flat_buffers(self):
        """Optional fast path (call once, AFTER ``.to(device)``): move every parameter into one flat fp32 arena
        and allocate a matching gradient arena.  Parameters stay ordinary ``nn.Parameter`` views (state_dict,
        load_state_dict, optimisers keep working); backward then writes each gradient straight into its arena
        slice, so ``AdamW(..., flat=...)`` is a single kernel launch per contiguous run and the data-parallel
        all-reduce runs in place on arena slices (no flatten / unflatten copies).  Do not call ``.to()`` after."""
        params = list(self.parameters())
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("use_flat_buffers() needs the module on a ROCm device (call .to(device) first)")
        offs, n = [], 0
        for p in params:
            offs.append(n)
            n += (p.numel() + 7) // 8 * 8          # fp32 slices 32-byte, their bf16 shadow slices 16-byte aligned
        flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
        flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        views = []
        with torch.no_grad():
            for p, o in zip(params, offs):
                v = flat_p[o:o + p.numel()].view_as(p)
                v.copy_(p)
                p.data = v
                p.grad = None
                views.append((p, flat_g[o:o + p.numel()].view_as(p)))
        old = getattr(self, "_arena_state", None)
        if old is not None:
            old.clear()
        self._arena_state = Fn.register_grad_sinks(views, Fn.ArenaState())
        # bf16 shadow of the whole arena: the bf16-storage GEMMs read weights from it, AdamW(flat=...) rewrites it in
        # the optimizer kernel (functional.weight_bf16 re-casts a slice if torch modifies the parameter itself)
        shadow = Fn.cast_bf16(flat_p)
        for p, o in zip(params, offs):
            Fn.register_weight_shadow(p, shadow[o:o + p.numel()].view_as(p))
        self._flat = dict(param=flat_p, grad=flat_g, offsets=offs, params=params, total=n, shadow=shadow,
                          state=self._arena_state)
        self._arena_state.flat = self._flat        # (functional.weight_x3: the bf16x3 word shadow is an arena next to these)
        return self._flat

    # first block of each ViT backward pass after the first, in execution order of backward (see forward_staged): passes run
    # blocks 11..8 (+ vit.norm), 7..4, 3..1, and last block 0 + patch embedding -- the last pass is kept SHORT because its
    # gradients are the only ones whose all-reduce cannot hide under later backward work (41 MB fp32 instead of 126 MB)
    backward_stage_starts = (8, 4, 1)

    def _pass_of_block(self, j):
        """index (1-based; 0 is the conv side) of the backward pass that runs transformer block j"""
        return 1 + sum(1 for s in self.backward_stage_starts if s > j)

    def stage_ranges(self):
        """Arena ranges [lo, hi) of the gradients each backward stage of ``forward_staged`` completes, in completion
        order: the conv side (encoder1-4, decoder5-2, out), ViT blocks 8-11 + vit.norm, blocks 4-7, blocks 1-3, patch
        embedding + block 0.  These are the data-parallel buckets (reverse execution order)."""
        flat = getattr(self, "_flat", None)
        if flat is None:
            raise RuntimeError("stage_ranges() needs use_flat_buffers()")
        index = {id(p): i for i, p in enumerate(flat["params"])}
        offs = flat["offsets"]
        first = lambda m: offs[index[id(next(m.parameters()))]]
        conv = first(self.encoder1)
        bounds = [conv] + [first(self.vit.blocks[s]) for s in sorted(self.backward_stage_starts, reverse=True)] + [0]
        return [(conv, flat["total"])] + [(lo, hi) for hi, lo in zip(bounds[:-1], bounds[1:])]

    def _prec(self) -> int:
        try:
            return _PRECISIONS[self.precision]
        except KeyError:
            raise ValueError(f"precision must be one of {sorted(_PRECISIONS)}, got {self.precision!r}")

    def proj_feat(self, x, hidden_size, feat_size):
        """unetr.py:177-180.  Kept for API parity (returns the NCDHW tensor the reference returns); the
        forward below never materialises it because [B, L, H] already IS channels-last [B, g, g, g, H]."""
        x = x.view(x.size(0), feat_size[0], feat_size[1], feat_size[2], hidden_size)
        x = x.permute(0, 4, 1, 2, 3).contiguous()
        return x

    def _tokens_cl(self, tok, B):
        g = self.feat_size
        return Fn.carry_twin(tok, tok.view(B, g[0], g[1], g[2], self.hidden_size))

    def _res_w(self, blk):
        return blk.conv1.conv.weight, blk.conv2.conv.weight, blk.conv3.conv.weight

    def _encode(self, x_in, prec, stages=None):
        """ViT + the four skip-path encoders.  ``stages`` (a list, staged mode): the ViT is cut into three groups of four
        blocks and the conv side is fed detached copies of the ViT outputs it consumes, so that backward can run as four
        consecutive passes (forward_staged); each list entry receives the (root, leaf) pairs that pass k starts from."""
        B = x_in.shape[0]
        if tuple(x_in.shape[1:]) != (self.in_channels, *self.img_size):
            raise ValueError(f"expected input [B,{self.in_channels},{self.img_size}], got {tuple(x_in.shape)}")
        pe = self.vit.patch_embedding
        lin = pe.patch_embeddings[1]
        L = pe.position_embeddings.shape[1]
        Fn._require_gpu(x_in)

        def cut(t, k):
            if stages is None:
                return t
            leaf = Fn.carry_twin(t, t.detach().requires_grad_(True))   # (bf16 twin / stashed LayerNorm formed by t's producer)
            stages[k].append((t, leaf))
            return leaf

        ckpt = self.encoder_checkpointing and torch.is_grad_enabled()
        x = Fn.PatchEmbedFn.apply(x_in, lin.weight, lin.bias, pe.position_embeddings, self.patch_size[0], prec)
        hidden_states_out = []
        nblk = len(self.vit.blocks)
        starts = set(self.backward_stage_starts)
        b16 = Fn._bf16_path(prec, self.hidden_size, self.vit.blocks[0].mlp.linear1.weight.shape[0])
        taps = (3, 6, 9)                       # hidden states the conv side consumes (unetr.py:197-201)
        for i, blk in enumerate(self.vit.blocks):
            nxt = self.vit.blocks[i + 1].norm1 if i + 1 < nblk and Fn.ln_ride_enabled() else None   # its forward rides on this block's last kernel
            x = Fn.TransformerBlockFn.apply(
                x, blk.norm1.weight, blk.norm1.bias, blk.attn.qkv.weight, blk.attn.out_proj.weight, blk.attn.out_proj.bias,
                blk.norm2.weight, blk.norm2.bias, blk.mlp.linear1.weight, blk.mlp.linear1.bias, blk.mlp.linear2.weight,
                blk.mlp.linear2.bias, B, L, self.num_heads, prec, ckpt,
                None if nxt is None else nxt.weight.detach(), None if nxt is None else nxt.bias.detach(),
                i in taps)                      # (a tapped block's last GEMM also writes the bf16 tokens the transposed conv reads)
            k = self._pass_of_block(i) - 1      # staged mode: index of the pass that runs block i's backward
            if i in taps:
                # two consumers: the next block takes alias `xa`, the skip path alias `xb`; TapFn's backward forms the sum of
                # their gradients (+ its bf16 twin) in one launch.  Staged mode: the skip path's gradient is parked on a leaf
                # until block i's pass starts, and so is the next block's when a pass boundary lies behind block i
                xa, xb = Fn.TapFn.apply(x, b16)
                hidden_states_out.append(cut(xb, k))
                x = cut(xa, k) if (i + 1) in starts else xa
            else:
                if (i + 1) in starts:
                    x = cut(x, k)
                hidden_states_out.append(x)
        x = Fn.LayerNormFn.apply(x, self.vit.norm.weight, self.vit.norm.bias,
                                 Fn._bf16_path(prec, self.hidden_size, self.vit.blocks[0].mlp.linear1.weight.shape[0]))
        x = cut(x, 0)
        # every skip tensor is produced straight into the second half of the decoder's concatenation buffer (to_cat)
        enc1 = Fn.ResBlockFn.apply(Fn.to_channels_last(x_in), *self._res_w(self.encoder1.layer), prec, True)
        enc = []
        for tap, blk in ((3, self.encoder2), (6, self.encoder3), (9, self.encoder4)):
            t = Fn.TconvFn.apply(self._tokens_cl(hidden_states_out[tap], B), blk.transp_conv_init.conv.weight, prec, len(blk.blocks) == 0)
            for k, sub in enumerate(blk.blocks):
                t = Fn.TconvFn.apply(t, sub.conv.weight, prec, k == len(blk.blocks) - 1)
            enc.append(t)
        return x, enc1, enc[0], enc[1], enc[2]

    def _decode(self, x_in, x, enc1, enc2, enc3, enc4, prec):
        B = x_in.shape[0]
        dec4 = self._tokens_cl(x, B)
        d = self.decoder5
        dec3 = Fn.UpBlockFn.apply(dec4, enc4, d.transp_conv.conv.weight, *self._res_w(d.conv_block), prec, True)
        d = self.decoder4
        dec2 = Fn.UpBlockFn.apply(dec3, enc3, d.transp_conv.conv.weight, *self._res_w(d.conv_block), prec, True)
        d = self.decoder3
        dec1 = Fn.UpBlockFn.apply(dec2, enc2, d.transp_conv.conv.weight, *self._res_w(d.conv_block), prec, True)
        d = self.decoder2
        # (the out conv is decoder2's only consumer: its kernels form the block end themselves -- UpBlockFn with the head's weights)
        logits = Fn.UpBlockFn.apply(dec1, enc1, d.transp_conv.conv.weight, *self._res_w(d.conv_block), prec, True,
                                    self.out.conv.conv.weight, self.out.conv.conv.bias)
        return Fn.ToNCDHWFn.apply(enc4), logits

    def forward(self, x_in, freeze_encoder=False):
        """unetr.py:182-208: returns (enc4 [B,8F,2g,2g,2g], logits [B,C_out,*img_size]), both NCDHW."""
        prec = self._prec()
        Fn._require_gpu(x_in)
        Fn.begin_forward(getattr(self, "_arena_state", None))
        if freeze_encoder:
            with torch.no_grad():
                x, enc1, enc2, enc3, enc4 = self._encode(x_in, prec)
        else:
            x, enc1, enc2, enc3, enc4 = self._encode(x_in, prec)
        return self._decode(x_in, x, enc1, enc2, enc3, enc4, prec)

    # ---- staged backward: what lets the data-parallel all-reduce overlap with backward ---------------------------
    def forward_staged(self, x_in):
        """Same arithmetic as ``forward(x_in)`` (the values are bit-identical), but the autograd graph is cut so that
        backward runs as FIVE consecutive passes whose parameter gradients are complete when each pass ends:
        0: loss -> conv side (encoder1-4, decoder5-2, out), 1: vit.norm + blocks 11..8, 2: blocks 7..4,
        3: blocks 3..1, 4: block 0 + patch embedding (``backward_stage_starts``).  ``backward_staged`` drives them; between two passes the caller may start the
        all-reduce of the arena range that just became final (``stage_ranges``) on a side stream.
        Returns (enc4, logits, stages)."""
        prec = self._prec()
        Fn._require_gpu(x_in)
        Fn.begin_forward(getattr(self, "_arena_state", None))
        stages = [[] for _ in range(len(self.backward_stage_starts) + 1)]
        x, enc1, enc2, enc3, enc4 = self._encode(x_in, prec, stages)
        enc4_out, logits = self._decode(x_in, x, enc1, enc2, enc3, enc4, prec)
        return enc4_out, logits, stages

    @staticmethod
    def backward_staged(loss, stages, after_stage=None):
        """``loss.backward()`` in passes (see forward_staged); ``after_stage(k)`` runs after pass k, k = 0..len(stages)."""
        loss.backward()
        if after_stage is not None:
            after_stage(0)
        for k, st in enumerate(stages, 1):
            roots = [r for r, leaf in st if leaf.grad is not None]
            grads = [leaf.grad for r, leaf in st if leaf.grad is not None]
            if roots:
                torch.autograd.backward(roots, grads)
            if after_stage is not None:
                after_stage(k)


class UNETRLogits(UNETR):
    """``monai.networks.nets.UNETR`` call convention (unetr_segmentation_3d.py:36,221,109): forward -> logits."""

    def forward(self, x_in):  # type: ignore[override]
        return super().forward(x_in)[1]
