"""DiceCELoss behind MONAI's interface for the configuration the reference uses on the hot path
(unetr_segmentation_3d.py:404: ``DiceCELoss(to_onehot_y=True, softmax=True)``), computed by the HIP kernels
in csrc/loss.hip (one streaming pass forward, one backward).  No CPU fallback."""
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
        if not to_onehot_y: bad.append("to_onehot_y=False")
        if sigmoid: bad.append("sigmoid=True")
        if not softmax: bad.append("softmax=False")
        if squared_pred: bad.append("squared_pred=True")
        if jaccard: bad.append("jaccard=True")
        if reduction != "mean": bad.append(f"reduction={reduction!r}")
        if batch: bad.append("batch=True")
        if lambda_dice != 1.0 or lambda_ce != 1.0: bad.append("lambda_dice/lambda_ce != 1")
        if bad:
            raise NotImplementedError("HIP DiceCELoss implements DiceCELoss(to_onehot_y=True, softmax=True) with MONAI "
                                      "defaults; got " + ", ".join(bad))
        self.smooth_nr = float(smooth_nr)
        self.smooth_dr = float(smooth_dr)

    def terms(self, input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        """3-vector (loss, dice term, ce term); only element 0 carries gradient."""
        if target.shape[1] != 1:
            raise ValueError("target must be [B,1,*spatial] class indices (to_onehot_y=True)")
        return Fn.DiceCEFn.apply(input, target, self.smooth_nr, self.smooth_dr)

    def forward(self, input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        return self.terms(input, target)[0]
