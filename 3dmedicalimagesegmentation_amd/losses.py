"""DiceCELoss behind MONAI's interface for the two configurations the reference instantiates
(unetr_segmentation_3d.py:404 ``DiceCELoss(to_onehot_y=True, softmax=True)`` for the single-channel CT tasks and
:477-482 ``DiceCELoss(to_onehot_y=False, sigmoid=True)`` for the 4-channel multi-label MR task), computed by the HIP
kernels in csrc/loss.hip (one streaming pass forward, one backward).  No CPU fallback."""
import torch
import torch.nn as nn

from . import functional as Fn


class DiceCELoss(nn.Module):
    def __init__(self, include_background: bool = True, to_onehot_y: bool = False, sigmoid: bool = False,
                 softmax: bool = False, squared_pred: bool = False, jaccard: bool = False, reduction: str = "mean",
                 smooth_nr: float = 1e-5, smooth_dr: float = 1e-5, batch: bool = False,
                 lambda_dice: float = 1.0, lambda_ce: float = 1.0) -> None:
        super().__init__()
        bad = []
        if not include_background: bad.append("include_background=False")
        if (to_onehot_y, softmax, sigmoid) not in ((True, True, False), (False, False, True)):
            bad.append(f"to_onehot_y={to_onehot_y}, softmax={softmax}, sigmoid={sigmoid}")
        if squared_pred: bad.append("squared_pred=True")
        if jaccard: bad.append("jaccard=True")
        if reduction != "mean": bad.append(f"reduction={reduction!r}")
        if batch: bad.append("batch=True")
        if lambda_dice != 1.0 or lambda_ce != 1.0: bad.append("lambda_dice/lambda_ce != 1")
        if bad:
            raise NotImplementedError("HIP DiceCELoss implements DiceCELoss(to_onehot_y=True, softmax=True) and "
                                      "DiceCELoss(to_onehot_y=False, sigmoid=True) with MONAI defaults; got " + ", ".join(bad))
        self.multilabel = bool(sigmoid)
        self.smooth_nr = float(smooth_nr)
        self.smooth_dr = float(smooth_dr)

    def terms(self, input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        """3-vector (loss, dice term, ce term); only element 0 carries gradient."""
        self._check(input, target)
        return Fn.DiceCEFn.apply(input, target, self.smooth_nr, self.smooth_dr, self.multilabel)

    def _check(self, input, target):
        if self.multilabel:
            if target.shape != input.shape:
                raise ValueError("sigmoid=True needs a multi-label target shaped like the logits [B,C,*spatial]")
        elif target.shape[1] != 1:
            raise ValueError("target must be [B,1,*spatial] class indices (to_onehot_y=True)")

    def forward(self, input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        self._check(input, target)
        return Fn.DiceCEFn.apply(input, target, self.smooth_nr, self.smooth_dr, self.multilabel, True)


class _RankingLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, slice_dimension, init_idx, temperature, kind):
        Fn._require_gpu(feat)
        feat = feat.contiguous()
        if feat.dim() != 5 or feat.shape[0] != 4:
            raise ValueError("ranking losses need a [4, C, S1, S2, S3] batch (2 volumes x 2 transforms), as "
                             "unetr_ranking_pretraining_3d.py:251-253 requires")
        _, C, S1, S2, S3 = feat.shape
        lib = Fn._capi.load()
        need = lib.unetr_ranking_workspace_floats(C, S1, S2, S3, slice_dimension)
        ws = Fn.workspace(feat.device)
        if need > ws.numel():
            raise RuntimeError("ranking loss workspace too small")
        loss = torch.empty(1, dtype=torch.float32, device=feat.device)
        W = torch.empty(C, 16, 16, dtype=torch.float32, device=feat.device)
        Fn.call("unetr_ranking_loss_fwd", feat.data_ptr(), C, S1, S2, S3, slice_dimension, init_idx, float(temperature), kind,
                loss.data_ptr(), W.data_ptr(), ws.data_ptr(), ws.numel(), Fn._stream())
        ctx.save_for_backward(feat, W)
        ctx.meta = (C, S1, S2, S3, slice_dimension, init_idx)
        return loss[0]

    @staticmethod
    def backward(ctx, dloss):
        feat, W = ctx.saved_tensors
        C, S1, S2, S3, sd, init_idx = ctx.meta
        dfeat = torch.zeros_like(feat)
        dl = dloss.reshape(1).contiguous()
        Fn.call("unetr_ranking_loss_bwd", feat.data_ptr(), C, S1, S2, S3, sd, init_idx, W.data_ptr(), dl.data_ptr(), dfeat.data_ptr(),
                Fn._stream())
        return dfeat, None, None, None, None


def ranking_loss(features, slice_dimension, init_idx, temperature, kind="ranking"):
    """Fused self-supervised pre-training loss of unetr_ranking_pretraining_3d.py.

    Equivalent to ``extract_triplets_more_partitions(f1, f2, slice_dimension)`` (lines 59-133, with the random
    ``init_idx`` passed in) followed by ``BTLoss`` (kind="ranking", lines 202-212) or ``ContrastiveLoss``
    (kind="contrastive", lines 219-231) on ``f1, f2 = torch.split(features, [2, 2])`` -- the loss value only; calling
    ``backward()`` / ``optimizer.step()`` stays with the caller as everywhere else in this package."""
    kinds = {"ranking": 0, "contrastive": 1}
    if kind not in kinds:
        raise KeyError(f"loss kind {kind!r} is not supported")
    return _RankingLossFn.apply(features, int(slice_dimension), int(init_idx), float(temperature), kinds[kind])
