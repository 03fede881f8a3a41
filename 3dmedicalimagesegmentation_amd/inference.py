"""Sliding-window inference and the Dice metric behind the reference's validation loop
(/root/reference/unetr_segmentation_3d.py:103-132), with MONAI 0.6.0's call signatures:

    val_outputs = sliding_window_inference(val_inputs, (crop, crop, crop), 4, model)           # :110
    dice_metric = DiceMetric(include_background=True, reduction="mean", get_not_nans=False)    # :485
    dice_metric(y_pred=[one-hot ...], y=[one-hot ...]); dice_metric.aggregate(); dice_metric.reset()   # :118-131

The window forward is the HIP hot path (run under torch.no_grad); blending and the Dice sums are the HIP kernels of
csrc/inference.hip.  Window geometry (scan interval, dense patch starts, padding of small volumes) is host integer logic
restated from MONAI 0.6.0 (monai/inferers/utils.py, monai/data/utils.py::dense_patch_slices).  No CPU fallback.
"""
import math
from typing import Callable, Sequence, Union

import torch
import torch.nn.functional as F

from . import functional as Fn
from ._capi import call


def _scan_interval(image_size, roi_size, overlap):
    out = []
    for i, r in zip(image_size, roi_size):
        if r == i:
            out.append(int(r))
        else:
            interval = int(r * (1 - overlap))
            out.append(interval if interval > 0 else 1)
    return out


def _dense_patch_starts(image_size, patch_size, scan_interval):
    """window start corners in MONAI's order (meshgrid indexing='ij': last spatial dim fastest)"""
    starts = []
    for i, p, s in zip(image_size, patch_size, scan_interval):
        if s == 0:
            num = 1
        else:
            n = int(math.ceil(float(i) / s))
            first = next((d for d in range(n) if d * s + p >= i), None)
            num = first + 1 if first is not None else 1
        dim = []
        for idx in range(num):
            st = idx * s
            st -= max(st + p - i, 0)
            dim.append(st)
        starts.append(dim)
    return [(z, y, x) for z in starts[0] for y in starts[1] for x in starts[2]]


@torch.no_grad()
def sliding_window_inference(inputs: torch.Tensor, roi_size: Union[Sequence[int], int], sw_batch_size: int,
                             predictor: Callable[..., torch.Tensor], overlap: float = 0.25, mode: str = "constant",
                             sigma_scale: float = 0.125, padding_mode: str = "constant", cval: float = 0.0,
                             *args, **kwargs) -> torch.Tensor:
    Fn._require_gpu(inputs)
    if inputs.dim() != 5:
        raise ValueError("3-D volumes [B,C,D,H,W] expected")
    if overlap < 0 or overlap >= 1:
        raise AssertionError("overlap must be >= 0 and < 1.")
    if mode != "constant":
        # the reference never passes mode (default "constant"); MONAI's Gaussian importance map is not restated here
        raise NotImplementedError(f"blending mode {mode!r}: only mode='constant' (the reference's default) is implemented")
    B = inputs.shape[0]
    image_size_ = list(inputs.shape[2:])
    roi = [roi_size] * 3 if isinstance(roi_size, int) else list(roi_size)
    roi = [r if r and r > 0 else i for r, i in zip(roi, image_size_)]           # fall_back_tuple
    image_size = [max(i, r) for i, r in zip(image_size_, roi)]
    pad = []
    for k in range(4, 1, -1):
        diff = max(roi[k - 2] - inputs.shape[k], 0)
        half = diff // 2
        pad.extend([half, diff - half])
    if any(pad):
        inputs = F.pad(inputs, pad=pad, mode=padding_mode, value=cval)
    inputs = inputs.contiguous()
    interval = _scan_interval(image_size, roi, overlap)
    starts = _dense_patch_starts(image_size, roi, interval)
    num_win = len(starts)
    total = num_win * B
    imp = None                                  # constant importance map (all ones)
    D, H, W = image_size
    V = D * H * W
    out = count = None
    stream = Fn._stream()
    for g0 in range(0, total, sw_batch_size):
        idxs = range(g0, min(g0 + sw_batch_size, total))
        wins = []
        for idx in idxs:
            b, (z, y, x) = idx // num_win, starts[idx % num_win]
            wins.append(inputs[b:b + 1, :, z:z + roi[0], y:y + roi[1], x:x + roi[2]])
        seg = predictor(torch.cat(wins), *args, **kwargs)
        if isinstance(seg, (tuple, list)):
            seg = seg[-1]                       # the reference UNETR returns (enc4, logits)
        seg = seg.contiguous().float()
        C = seg.shape[1]
        if out is None:
            out = torch.zeros(B, C, D, H, W, dtype=torch.float32, device=inputs.device)
            count = torch.zeros(B, D, H, W, dtype=torch.float32, device=inputs.device)
        for j, idx in enumerate(idxs):
            b, (z, y, x) = idx // num_win, starts[idx % num_win]
            call("unetr_sw_accumulate", seg[j].data_ptr(), imp.data_ptr() if imp is not None else None,
                 out[b].data_ptr(), count[b].data_ptr(), C, roi[0], roi[1], roi[2], D, H, W, z, y, x, stream)
    call("unetr_sw_finalize", out.data_ptr(), count.data_ptr(), B, out.shape[1], V, stream)
    sl = [slice(None), slice(None)]
    for sp in range(3):                       # undo the padding of volumes smaller than the window
        lo = pad[(2 - sp) * 2]
        sl.append(slice(lo, lo + image_size_[sp]))
    return out[tuple(sl)]


def dice_counts(pred: torch.Tensor, y: torch.Tensor, from_logits: bool) -> torch.Tensor:
    """[B, C, 3] float64 sums (pred*y, pred, y) from the HIP kernel"""
    Fn._require_gpu(pred)
    pred = pred.contiguous()
    y = y.contiguous().float()
    B, C = pred.shape[0], pred.shape[1]
    V = pred.numel() // (B * C)
    if from_logits and y.numel() != B * V:
        raise ValueError("from_logits: y must hold one class id per voxel [B,1,*spatial]")
    if not from_logits and y.shape != pred.shape:
        raise ValueError("y_pred and y must have the same one-hot shape")
    counts = torch.empty(B, C, 3, dtype=torch.float64, device=pred.device)
    ws = Fn.workspace(pred.device)
    call("unetr_dice_counts", pred.data_ptr(), y.data_ptr(), B, C, V, int(from_logits), counts.data_ptr(), ws.data_ptr(),
         ws.numel() * 4, Fn._stream())
    return counts


class DiceMetric:
    """monai.metrics.DiceMetric (0.6.0) for the reference's two instances: reduction "mean" and "mean_batch",
    include_background=True, get_not_nans=False.  ``__call__`` takes batched tensors or lists of per-item one-hot
    tensors (what decollate_batch + AsDiscrete produce at unetr_segmentation_3d.py:111-117) and buffers per-item
    per-class Dice (NaN where the ground truth class is absent); ``from_logits`` fuses argmax + one-hot (post_pred /
    post_label at :405-406) into the counting kernel."""

    def __init__(self, include_background: bool = True, reduction: str = "mean", get_not_nans: bool = False):
        if not include_background or get_not_nans or reduction not in ("mean", "mean_batch"):
            raise NotImplementedError("DiceMetric(include_background=True, reduction='mean'|'mean_batch', get_not_nans=False)")
        self.reduction = reduction
        self._buf = []

    @staticmethod
    def _stack(v):
        return torch.stack(list(v)) if isinstance(v, (list, tuple)) else v

    def __call__(self, y_pred, y, from_logits: bool = False):
        y_pred, y = self._stack(y_pred), self._stack(y)
        c = dice_counts(y_pred, y, from_logits)
        inter, po, yo = c[..., 0], c[..., 1], c[..., 2]
        f = torch.where(yo > 0, 2.0 * inter / (yo + po), torch.full_like(yo, float("nan"))).float()
        self._buf.append(f)
        return f

    def aggregate(self):
        f = torch.cat(self._buf).clone()
        nans = torch.isnan(f)
        not_nans = (~nans).float()
        f[nans] = 0
        zero = torch.zeros(1, device=f.device)
        if self.reduction == "mean":
            nn_c = not_nans.sum(dim=1)
            f = torch.where(nn_c > 0, f.sum(dim=1) / nn_c, zero)      # channel average
            nn_b = (nn_c > 0).float().sum(dim=0)
            return torch.where(nn_b > 0, f.sum(dim=0) / nn_b, zero)   # batch average
        nn_b = not_nans.sum(dim=0)
        return torch.where(nn_b > 0, f.sum(dim=0) / nn_b, zero)       # "mean_batch": per class

    def reset(self):
        self._buf = []
