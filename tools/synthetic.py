"""Synthetic CT-like volumes: the reference pipeline yields intensities in [0, 1] after
ScaleIntensityRanged(-175, 250 -> 0, 1, clip) (unetr_segmentation_3d.py:332-339) +- 0.1 shift (:371-375); here: a sum of 8
random 3-D Gaussian blobs + N(0, 0.05) noise clipped to [-0.1, 1.1], and an integer-valued label map from thresholding
blobs (every class non-empty).  Shared by bench.py, the tests and the CPU oracle so that all sides see identical inputs."""
import torch


def synthetic_volume(batch, in_channels, size, n_classes, seed):
    g = torch.Generator().manual_seed(seed)
    ax = torch.linspace(-1, 1, size)
    zz, yy, xx = torch.meshgrid(ax, ax, ax, indexing="ij")
    img = torch.zeros(batch, in_channels, size, size, size)
    lab = torch.zeros(batch, 1, size, size, size)
    for b in range(batch):
        blobs = []
        for _ in range(8):
            c = torch.rand(3, generator=g) * 1.6 - 0.8
            s = 0.15 + 0.35 * torch.rand(1, generator=g).item()
            a = 0.3 + 0.7 * torch.rand(1, generator=g).item()
            blobs.append(a * torch.exp(-((zz - c[0]) ** 2 + (yy - c[1]) ** 2 + (xx - c[2]) ** 2) / (2 * s * s)))
        base = torch.stack(blobs).sum(0)
        base = base / base.max()
        for c in range(in_channels):
            img[b, c] = (base + 0.05 * torch.randn(base.shape, generator=g)).clamp(-0.1, 1.1)
        cls = torch.zeros_like(base)
        for k in range(1, n_classes):
            bk = blobs[(k - 1) % len(blobs)]
            cls = torch.where(bk > 0.5 * bk.max(), torch.full_like(cls, float(k)), cls)
        lab[b, 0] = cls
    return img, lab
