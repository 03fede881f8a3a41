"""CPU oracle for the UNETR training hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

PARITY UNPINNED: the arithmetic of the reference's hot path lives in the un-vendored third-party
package ``monai==0.6.0`` (pin: /root/reference/pytorch_env.yml:94) which is neither installed nor
on disk here (``import monai`` -> ModuleNotFoundError) and the reference ships no tests, fixtures
or golden vectors.  This file restates MONAI 0.6.0's published block semantics in plain
PyTorch (fp32, CPU) wired exactly as the reference wires them, and is anchored on the only
reference-derived invariants that exist: the call sites, the strict ``state_dict`` key schema,
the parameter count, the constructor exceptions and the output shapes.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module, and only as the checker / the reported CPU baseline.  The product package
(``3dmedicalimagesegmentation_amd``) never imports it and has no CPU fallback.

Reference map (file:line into /root/reference):
  UNETR.__init__            unetr.py:27-175      -> OracleUNETR.__init__
  UNETR.proj_feat           unetr.py:177-180     -> OracleUNETR.proj_feat
  UNETR.forward             unetr.py:182-208     -> OracleUNETR.forward
  ViT / blocks              unetr.py:16-18,78-89 -> OracleViT, OraclePatchEmbeddingBlock,
                                                    OracleTransformerBlock, OracleSABlock, OracleMLPBlock
  conv blocks               unetr.py:90-175      -> OracleUnetResBlock, OracleUnetrBasicBlock,
                                                    OracleUnetrPrUpBlock, OracleUnetrUpBlock, OracleUnetOutBlock
  DiceCELoss                unetr_segmentation_3d.py:404,480 -> oracle_dice_ce_loss
  train step                unetr_segmentation_3d.py:220-226,522 -> oracle_train_step
  BTLoss / ContrastiveLoss  unetr_ranking_pretraining_3d.py:59-133,202-236 -> oracle_extract_triplets,
                                                    oracle_bt_loss, oracle_contrastive_loss
"""
from itertools import permutations, product
from typing import Sequence, Tuple, Union

import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------
# ViT (monai.networks.nets.vit.ViT @0.6.0, as called at unetr.py:78-89)
# ----------------------------------------------------------------------------------------------
class _PerceptronPatches(nn.Module):
    """einops Rearrange("b c (h p1) (w p2) (d p3) -> b (h w d) (p1 p2 p3 c)") without einops."""

    def __init__(self, patch_size):
        super().__init__()
        self.p = tuple(patch_size)

    def forward(self, x):
        b, c, hh, ww, dd = x.shape
        p1, p2, p3 = self.p
        h, w, d = hh // p1, ww // p2, dd // p3
        x = x.view(b, c, h, p1, w, p2, d, p3)
        # -> b h w d p1 p2 p3 c
        x = x.permute(0, 2, 4, 6, 3, 5, 7, 1).contiguous()
        return x.view(b, h * w * d, p1 * p2 * p3 * c)


class OraclePatchEmbeddingBlock(nn.Module):
    def __init__(self, in_channels, img_size, patch_size, hidden_size, num_heads, pos_embed, dropout_rate):
        super().__init__()
        if not (0 <= dropout_rate <= 1):
            raise AssertionError("dropout_rate should be between 0 and 1.")
        if hidden_size % num_heads != 0:
            raise AssertionError("hidden size should be divisible by num_heads.")
        for m, p in zip(img_size, patch_size):
            if m < p:
                raise AssertionError("patch_size should be smaller than img_size.")
        if pos_embed not in ["conv", "perceptron"]:
            raise KeyError(f"Position embedding layer of type {pos_embed} is not supported.")
        if pos_embed == "perceptron":
            if any(i % p != 0 for i, p in zip(img_size, patch_size)):
                raise AssertionError("img_size should be divisible by patch_size for perceptron patch embedding.")
        self.n_patches = 1
        for i, p in zip(img_size, patch_size):
            self.n_patches *= i // p
        self.patch_dim = in_channels * patch_size[0] * patch_size[1] * patch_size[2]
        self.pos_embed = pos_embed
        if pos_embed == "conv":
            self.patch_embeddings = nn.Conv3d(in_channels, hidden_size, kernel_size=patch_size, stride=patch_size)
        else:
            self.patch_embeddings = nn.Sequential(_PerceptronPatches(patch_size), nn.Linear(self.patch_dim, hidden_size))
        self.position_embeddings = nn.Parameter(torch.zeros(1, self.n_patches, hidden_size))
        self.cls_token = nn.Parameter(torch.zeros(1, 1, hidden_size))  # registered, never used
        self.dropout = nn.Dropout(dropout_rate)
        nn.init.trunc_normal_(self.position_embeddings, mean=0.0, std=0.02, a=-2.0, b=2.0)
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, mean=0.0, std=0.02, a=-2.0, b=2.0)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def forward(self, x):
        if self.pos_embed == "conv":
            x = self.patch_embeddings(x).flatten(2).transpose(-1, -2)
        else:
            x = self.patch_embeddings(x)
        return self.dropout(x + self.position_embeddings)


class OracleSABlock(nn.Module):
    def __init__(self, hidden_size, num_heads, dropout_rate=0.0):
        super().__init__()
        if not (0 <= dropout_rate <= 1):
            raise AssertionError("dropout_rate should be between 0 and 1.")
        if hidden_size % num_heads != 0:
            raise AssertionError("hidden size should be divisible by num_heads.")
        self.num_heads = num_heads
        self.out_proj = nn.Linear(hidden_size, hidden_size)
        self.qkv = nn.Linear(hidden_size, hidden_size * 3, bias=False)
        self.drop_output = nn.Dropout(dropout_rate)
        self.drop_weights = nn.Dropout(dropout_rate)
        self.head_dim = hidden_size // num_heads
        self.scale = self.head_dim ** -0.5

    def forward(self, x):
        b, n, hdim = x.shape
        # "b h (qkv l d) -> qkv b l h d": feature = which*H + head*d + j
        qkv = self.qkv(x).view(b, n, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        att = (torch.einsum("blxd,blyd->blxy", q, k) * self.scale).softmax(dim=-1)
        att = self.drop_weights(att)
        x = torch.einsum("bhxy,bhyd->bhxd", att, v)
        x = x.permute(0, 2, 1, 3).reshape(b, n, hdim)  # "b h l d -> b l (h d)"
        return self.drop_output(self.out_proj(x))


class OracleMLPBlock(nn.Module):
    def __init__(self, hidden_size, mlp_dim, dropout_rate=0.0):
        super().__init__()
        if not (0 <= dropout_rate <= 1):
            raise AssertionError("dropout_rate should be between 0 and 1.")
        self.linear1 = nn.Linear(hidden_size, mlp_dim)
        self.linear2 = nn.Linear(mlp_dim, hidden_size)
        self.fn = nn.GELU()
        self.drop1 = nn.Dropout(dropout_rate)
        self.drop2 = nn.Dropout(dropout_rate)

    def forward(self, x):
        return self.drop2(self.linear2(self.drop1(self.fn(self.linear1(x)))))


class OracleTransformerBlock(nn.Module):
    def __init__(self, hidden_size, mlp_dim, num_heads, dropout_rate=0.0):
        super().__init__()
        self.mlp = OracleMLPBlock(hidden_size, mlp_dim, dropout_rate)
        self.norm1 = nn.LayerNorm(hidden_size)
        self.attn = OracleSABlock(hidden_size, num_heads, dropout_rate)
        self.norm2 = nn.LayerNorm(hidden_size)

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        x = x + self.mlp(self.norm2(x))
        return x


class OracleViT(nn.Module):
    def __init__(self, in_channels, img_size, patch_size, hidden_size=768, mlp_dim=3072, num_layers=12,
                 num_heads=12, pos_embed="perceptron", classification=False, num_classes=2, dropout_rate=0.0):
        super().__init__()
        if not (0 <= dropout_rate <= 1):
            raise AssertionError("dropout_rate should be between 0 and 1.")
        if hidden_size % num_heads != 0:
            raise AssertionError("hidden size should be divisible by num_heads.")
        if pos_embed not in ["conv", "perceptron"]:
            raise KeyError(f"Position embedding layer of type {pos_embed} is not supported.")
        self.classification = classification
        self.patch_embedding = OraclePatchEmbeddingBlock(
            in_channels, img_size, patch_size, hidden_size, num_heads, pos_embed, dropout_rate)
        self.blocks = nn.ModuleList(
            [OracleTransformerBlock(hidden_size, mlp_dim, num_heads, dropout_rate) for _ in range(num_layers)])
        self.norm = nn.LayerNorm(hidden_size)
        if classification:
            self.classification_head = nn.Linear(hidden_size, num_classes)

    def forward(self, x):
        x = self.patch_embedding(x)
        hidden_states_out = []
        for blk in self.blocks:
            x = blk(x)
            hidden_states_out.append(x)
        x = self.norm(x)
        if self.classification:
            x = self.classification_head(x[:, 0])
        return x, hidden_states_out


# ----------------------------------------------------------------------------------------------
# conv blocks (monai.networks.blocks.{dynunet_block,unetr_block} @0.6.0, called at unetr.py:90-175)
# ----------------------------------------------------------------------------------------------
class _Conv(nn.Sequential):
    """monai Convolution(conv_only=True): a Sequential holding one sub-module named ``conv``."""

    def __init__(self, in_ch, out_ch, kernel_size, stride=1, bias=False, transposed=False):
        super().__init__()
        pad = (kernel_size - stride + 1) // 2
        if transposed:
            out_pad = 2 * pad + stride - kernel_size
            conv = nn.ConvTranspose3d(in_ch, out_ch, kernel_size, stride, pad, out_pad, bias=bias)
        else:
            conv = nn.Conv3d(in_ch, out_ch, kernel_size, stride, pad, bias=bias)
        self.add_module("conv", conv)


def _norm(norm_name, ch):
    name = norm_name[0] if isinstance(norm_name, (tuple, list)) else norm_name
    if str(name).lower() == "instance":
        return nn.InstanceNorm3d(ch)
    if str(name).lower() == "batch":
        return nn.BatchNorm3d(ch)
    raise KeyError(f"norm {norm_name} not restated in the oracle")


class OracleUnetResBlock(nn.Module):
    def __init__(self, in_ch, out_ch, kernel_size, stride, norm_name):
        super().__init__()
        self.conv1 = _Conv(in_ch, out_ch, kernel_size, stride)
        self.conv2 = _Conv(out_ch, out_ch, kernel_size, 1)
        self.conv3 = _Conv(in_ch, out_ch, 1, stride)
        self.lrelu = nn.LeakyReLU(negative_slope=0.01, inplace=False)
        self.norm1 = _norm(norm_name, out_ch)
        self.norm2 = _norm(norm_name, out_ch)
        self.norm3 = _norm(norm_name, out_ch)
        self.downsample = in_ch != out_ch or stride != 1

    def forward(self, inp):
        residual = inp
        out = self.lrelu(self.norm1(self.conv1(inp)))
        out = self.norm2(self.conv2(out))
        if self.downsample:
            residual = self.norm3(self.conv3(residual))
        out = out + residual
        return self.lrelu(out)


class OracleUnetBasicBlock(nn.Module):
    def __init__(self, in_ch, out_ch, kernel_size, stride, norm_name):
        super().__init__()
        self.conv1 = _Conv(in_ch, out_ch, kernel_size, stride)
        self.conv2 = _Conv(out_ch, out_ch, kernel_size, 1)
        self.lrelu = nn.LeakyReLU(negative_slope=0.01, inplace=False)
        self.norm1 = _norm(norm_name, out_ch)
        self.norm2 = _norm(norm_name, out_ch)

    def forward(self, inp):
        out = self.lrelu(self.norm1(self.conv1(inp)))
        return self.lrelu(self.norm2(self.conv2(out)))


class OracleUnetrBasicBlock(nn.Module):
    def __init__(self, in_ch, out_ch, kernel_size, stride, norm_name, res_block):
        super().__init__()
        blk = OracleUnetResBlock if res_block else OracleUnetBasicBlock
        self.layer = blk(in_ch, out_ch, kernel_size, stride, norm_name)

    def forward(self, x):
        return self.layer(x)


class OracleUnetrPrUpBlock(nn.Module):
    def __init__(self, in_ch, out_ch, num_layer, kernel_size, stride, upsample_kernel_size, norm_name,
                 conv_block, res_block):
        super().__init__()
        k = upsample_kernel_size
        self.transp_conv_init = _Conv(in_ch, out_ch, k, k, transposed=True)
        if conv_block:
            blk = OracleUnetResBlock if res_block else OracleUnetBasicBlock
            self.blocks = nn.ModuleList([
                nn.Sequential(_Conv(out_ch, out_ch, k, k, transposed=True),
                              blk(out_ch, out_ch, kernel_size, stride, norm_name))
                for _ in range(num_layer)])
        else:
            self.blocks = nn.ModuleList([_Conv(out_ch, out_ch, k, k, transposed=True) for _ in range(num_layer)])

    def forward(self, x):
        x = self.transp_conv_init(x)
        for blk in self.blocks:
            x = blk(x)
        return x


class OracleUnetrUpBlock(nn.Module):
    def __init__(self, in_ch, out_ch, kernel_size, upsample_kernel_size, norm_name, res_block):
        super().__init__()
        k = upsample_kernel_size
        self.transp_conv = _Conv(in_ch, out_ch, k, k, transposed=True)
        blk = OracleUnetResBlock if res_block else OracleUnetBasicBlock
        self.conv_block = blk(out_ch + out_ch, out_ch, kernel_size, 1, norm_name)

    def forward(self, inp, skip):
        out = self.transp_conv(inp)
        out = torch.cat((out, skip), dim=1)
        return self.conv_block(out)


class OracleUnetOutBlock(nn.Module):
    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.conv = _Conv(in_ch, out_ch, 1, 1, bias=True)

    def forward(self, x):
        return self.conv(x)


# ----------------------------------------------------------------------------------------------
# UNETR (unetr.py:21-208)
# ----------------------------------------------------------------------------------------------
class OracleUNETR(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, img_size: Tuple[int, int, int], feature_size: int,
                 hidden_size: int, mlp_dim: int, num_heads: int, pos_embed: str, norm_name: Union[Tuple, str],
                 conv_block: bool = False, res_block: bool = False, dropout_rate: float = 0.0) -> None:
        super().__init__()
        if not (0 <= dropout_rate <= 1):                      # unetr.py:60-61
            raise AssertionError("dropout_rate should be between 0 and 1.")
        if hidden_size % num_heads != 0:                      # unetr.py:63-64
            raise AssertionError("hidden size should be divisible by num_heads.")
        if pos_embed not in ["conv", "perceptron"]:           # unetr.py:66-67
            raise KeyError(f"Position embedding layer of type {pos_embed} is not supported.")
        self.num_layers = 12                                  # unetr.py:69
        self.patch_size = (16, 16, 16)                        # unetr.py:70
        self.feat_size = tuple(img_size[i] // self.patch_size[i] for i in range(3))
        self.hidden_size = hidden_size
        self.classification = False
        self.vit = OracleViT(in_channels, img_size, self.patch_size, hidden_size, mlp_dim, self.num_layers,
                             num_heads, pos_embed, self.classification, dropout_rate=dropout_rate)
        f = feature_size
        self.encoder1 = OracleUnetrBasicBlock(in_channels, f, 3, 1, norm_name, res_block)
        self.encoder2 = OracleUnetrPrUpBlock(hidden_size, f * 2, 2, 3, 1, 2, norm_name, conv_block, res_block)
        self.encoder3 = OracleUnetrPrUpBlock(hidden_size, f * 4, 1, 3, 1, 2, norm_name, conv_block, res_block)
        self.encoder4 = OracleUnetrPrUpBlock(hidden_size, f * 8, 0, 3, 1, 2, norm_name, conv_block, res_block)
        self.decoder5 = OracleUnetrUpBlock(hidden_size, f * 8, 3, 2, norm_name, res_block)
        self.decoder4 = OracleUnetrUpBlock(f * 8, f * 4, 3, 2, norm_name, res_block)
        self.decoder3 = OracleUnetrUpBlock(f * 4, f * 2, 3, 2, norm_name, res_block)
        self.decoder2 = OracleUnetrUpBlock(f * 2, f, 3, 2, norm_name, res_block)
        self.out = OracleUnetOutBlock(f, out_channels)

    def proj_feat(self, x, hidden_size, feat_size):          # unetr.py:177-180
        x = x.view(x.size(0), feat_size[0], feat_size[1], feat_size[2], hidden_size)
        return x.permute(0, 4, 1, 2, 3).contiguous()

    def _encode(self, x_in):
        x, hs = self.vit(x_in)
        enc1 = self.encoder1(x_in)
        enc2 = self.encoder2(self.proj_feat(hs[3], self.hidden_size, self.feat_size))
        enc3 = self.encoder3(self.proj_feat(hs[6], self.hidden_size, self.feat_size))
        enc4 = self.encoder4(self.proj_feat(hs[9], self.hidden_size, self.feat_size))
        return x, enc1, enc2, enc3, enc4

    def forward(self, x_in, freeze_encoder=False):           # unetr.py:182-208
        if freeze_encoder:
            with torch.no_grad():
                x, enc1, enc2, enc3, enc4 = self._encode(x_in)
        else:
            x, enc1, enc2, enc3, enc4 = self._encode(x_in)
        dec4 = self.proj_feat(x, self.hidden_size, self.feat_size)
        dec3 = self.decoder5(dec4, enc4)
        dec2 = self.decoder4(dec3, enc3)
        dec1 = self.decoder3(dec2, enc2)
        out = self.decoder2(dec1, enc1)
        logits = self.out(out)
        return enc4, logits


# ----------------------------------------------------------------------------------------------
# DiceCELoss (monai.losses.DiceCELoss @0.6.0, unetr_segmentation_3d.py:404 and :480)
# ----------------------------------------------------------------------------------------------
def oracle_dice_ce_terms(logits, target, to_onehot_y=True, softmax=True, sigmoid=False,
                         smooth_nr=1e-5, smooth_dr=1e-5):
    """Returns (dice_term, ce_term); loss = dice_term + ce_term (lambda_dice = lambda_ce = 1)."""
    n_pred_ch = logits.shape[1]
    if sigmoid:
        p = torch.sigmoid(logits)
    elif softmax:
        p = torch.softmax(logits, 1)
    else:
        p = logits
    if to_onehot_y:
        y = F.one_hot(target.squeeze(1).long(), n_pred_ch).movedim(-1, 1).to(p.dtype)
    else:
        y = target.to(p.dtype)
    axes = tuple(range(2, logits.dim()))
    inter = (y * p).sum(axes)
    den = y.sum(axes) + p.sum(axes)
    dice = (1.0 - (2.0 * inter + smooth_nr) / (den + smooth_dr)).mean()
    if n_pred_ch == target.shape[1]:
        ce_t = torch.argmax(target, dim=1)
    else:
        ce_t = target.squeeze(1)
    ce = F.cross_entropy(logits, ce_t.long(), reduction="mean")
    return dice, ce


def oracle_dice_ce_loss(logits, target, **kw):
    d, c = oracle_dice_ce_terms(logits, target, **kw)
    return d + c


def oracle_train_step(model, optimizer, x, y):
    """unetr_segmentation_3d.py:220-226 with the monai.networks.nets.UNETR (logits-only) convention."""
    _, logits = model(x)
    loss = oracle_dice_ce_loss(logits, y)
    loss.backward()
    optimizer.step()
    optimizer.zero_grad()
    return loss.detach()


# ----------------------------------------------------------------------------------------------
# ranking pre-training losses (unetr_ranking_pretraining_3d.py:59-133, 202-236) -- fully in-repo,
# so this part IS a line-by-line restatement.
# ----------------------------------------------------------------------------------------------
def oracle_cosine_sim_171(a, b, eps=1e-6):
    """nn.CosineSimilarity(dim=-1, eps) with the torch 1.7.1 formula x.y / max(|x||y|, eps)."""
    num = (a * b).sum(-1)
    den = (a.norm(dim=-1) * b.norm(dim=-1)).clamp_min(eps)
    return num / den


def oracle_slices(f1, f2, slice_dimension, init_idx, num_partitions=4):
    """unetr_ranking_pretraining_3d.py:69-118: 4 partitions x [x1, x1_trans, x2, x2_trans]."""
    dims = f1.shape
    part = int(dims[slice_dimension] / num_partitions)
    out = []
    for pidx in range(num_partitions):
        s = init_idx + pidx * part
        idx = [slice(None)] * 4
        idx[slice_dimension - 1] = s
        out.append([f1[0][tuple(idx)].reshape(dims[1], -1), f1[1][tuple(idx)].reshape(dims[1], -1),
                    f2[0][tuple(idx)].reshape(dims[1], -1), f2[1][tuple(idx)].reshape(dims[1], -1)])
    return out


def oracle_extract_triplets(f1, f2, slice_dimension, init_idx, num_partitions=4):
    """unetr_ranking_pretraining_3d.py:59-133 with np.random.choice replaced by an injected init_idx."""
    slices_list = oracle_slices(f1, f2, slice_dimension, init_idx, num_partitions)
    reference, similar, dissimilar = [], [], []
    for pidx in range(num_partitions):
        cur = slices_list[pidx]
        others = []
        for o in range(num_partitions):
            if o != pidx:
                others.extend(slices_list[o])
        for (pair, dis) in product(permutations(cur, 2), others):
            reference.append(pair[0])
            similar.append(pair[1])
            dissimilar.append(dis)
    return reference, similar, dissimilar


def _cos_memo(a, b, memo):
    """cos(a, b) memoised on the identity of the two slice tensors: the 576 triplets are built from only 16 distinct
    slices, so the reference's 2 x 576 (BT) / 576 x 577 (contrastive) cosine calls take 256 distinct values.  A pure
    function evaluated once per distinct argument pair gives bit-identical results in the same summation order; it
    only keeps the CPU checker at seconds instead of minutes."""
    k = (id(a), id(b))
    v = memo.get(k)
    if v is None:
        v = memo[k] = oracle_cosine_sim_171(a, b)
    return v


def oracle_bt_loss(reference, similar, dissimilar, temperature):
    """unetr_ranking_pretraining_3d.py:202-212 (the loss value; backward/step is the caller's)."""
    loss = 0
    memo = {}
    for ref, sim, dis in zip(reference, similar, dissimilar):
        comp = _cos_memo(ref, sim, memo) / temperature - _cos_memo(ref, dis, memo) / temperature
        loss = loss + torch.mean(torch.log(1 + torch.exp(-comp)))
    return loss


def oracle_contrastive_loss(reference, similar, dissimilar, temperature):
    """unetr_ranking_pretraining_3d.py:219-231."""
    loss = 0
    memo, ememo = {}, {}

    def ecos(a, b):
        k = (id(a), id(b))
        v = ememo.get(k)
        if v is None:
            v = ememo[k] = torch.exp(_cos_memo(a, b, memo) / temperature)
        return v

    for ref, sim in zip(reference, similar):
        num = ecos(ref, sim)
        den_list = [ecos(ref, dis) for dis in dissimilar]
        den_list.append(num)
        den = torch.stack(den_list, dim=0).sum(dim=0)
        loss = loss + (-torch.mean(torch.log(num / den)))
    return loss


# ----------------------------------------------------------------------------------------------
# synthetic CT-like data shared by tests and bench so both sides see identical inputs
# ----------------------------------------------------------------------------------------------
# ----------------------------------------------------------------------------------------------------------------
# Validation side (unetr_segmentation_3d.py:103-132).  MONAI 0.6.0 is not installed: these restate
# monai.inferers.sliding_window_inference (mode="constant", the reference's default), monai.data.utils.dense_patch_slices,
# AsDiscrete(argmax/to_onehot) and monai.metrics.DiceMetric + do_metric_reduction from the published 0.6.0 sources.
# PARITY UNPINNED (no MONAI here to check against); the reference's call sites fix the arguments.
def oracle_sliding_window_inference(inputs, roi_size, sw_batch_size, predictor, overlap=0.25):
    import math
    num_spatial_dims = inputs.dim() - 2
    batch_size = inputs.shape[0]
    image_size_ = list(inputs.shape[2:])
    roi_size = tuple(roi_size)
    image_size = tuple(max(image_size_[i], roi_size[i]) for i in range(num_spatial_dims))
    pad_size = []
    for k in range(inputs.dim() - 1, 1, -1):
        diff = max(roi_size[k - 2] - inputs.shape[k], 0)
        half = diff // 2
        pad_size.extend([half, diff - half])
    inputs = F.pad(inputs, pad=pad_size, mode="constant", value=0.0)
    scan_interval = []
    for i in range(num_spatial_dims):
        if roi_size[i] == image_size[i]:
            scan_interval.append(int(roi_size[i]))
        else:
            interval = int(roi_size[i] * (1 - overlap))
            scan_interval.append(interval if interval > 0 else 1)
    # dense_patch_slices
    scan_num = []
    for i in range(num_spatial_dims):
        num = int(math.ceil(float(image_size[i]) / scan_interval[i]))
        scan_dim = next((d for d in range(num) if d * scan_interval[i] + roi_size[i] >= image_size[i]), None)
        scan_num.append(scan_dim + 1 if scan_dim is not None else 1)
    starts = []
    for dim in range(num_spatial_dims):
        dim_starts = []
        for idx in range(scan_num[dim]):
            start_idx = idx * scan_interval[dim]
            start_idx -= max(start_idx + roi_size[dim] - image_size[dim], 0)
            dim_starts.append(start_idx)
        starts.append(dim_starts)
    grid = torch.stack(torch.meshgrid(*[torch.tensor(s) for s in starts], indexing="ij"), -1).reshape(-1, num_spatial_dims)
    slices = [tuple(slice(int(s), int(s) + roi_size[d]) for d, s in enumerate(x)) for x in grid]
    num_win = len(slices)
    total_slices = num_win * batch_size
    importance_map = torch.ones(roi_size, dtype=inputs.dtype)
    output_image = count_map = None
    for slice_g in range(0, total_slices, sw_batch_size):
        slice_range = range(slice_g, min(slice_g + sw_batch_size, total_slices))
        unravel_slice = [[slice(int(idx / num_win), int(idx / num_win) + 1), slice(None)] + list(slices[idx % num_win])
                         for idx in slice_range]
        window_data = torch.cat([inputs[tuple(win_slice)] for win_slice in unravel_slice])
        seg_prob = predictor(window_data)
        if output_image is None:
            output_shape = [batch_size, seg_prob.shape[1]] + list(image_size)
            output_image = torch.zeros(output_shape, dtype=seg_prob.dtype)
            count_map = torch.zeros(output_shape, dtype=seg_prob.dtype)
        for idx, original_idx in zip(slice_range, unravel_slice):
            output_image[tuple(original_idx)] += importance_map * seg_prob[idx - slice_g]
            count_map[tuple(original_idx)] += importance_map
    output_image = output_image / count_map
    final_slicing = []
    for sp in range(num_spatial_dims):
        slice_dim = slice(pad_size[sp * 2], image_size_[num_spatial_dims - sp - 1] + pad_size[sp * 2])
        final_slicing.insert(0, slice_dim)
    while len(final_slicing) < output_image.dim():
        final_slicing.insert(0, slice(None))
    return output_image[tuple(final_slicing)]


def oracle_post_pred(logits, n_classes):
    """AsDiscrete(argmax=True, to_onehot=True, n_classes) on a batched [B,C,...] tensor (unetr_segmentation_3d.py:406)"""
    return F.one_hot(torch.argmax(logits, dim=1), n_classes).movedim(-1, 1).float()


def oracle_post_label(label, n_classes):
    """AsDiscrete(to_onehot=True, n_classes) (unetr_segmentation_3d.py:405)"""
    return F.one_hot(label.squeeze(1).long(), n_classes).movedim(-1, 1).float()


def oracle_dice_metric(y_pred, y, reduction="mean"):
    """monai.metrics.compute_meandice (include_background=True) + do_metric_reduction for "mean" / "mean_batch";
    returns (per-item per-class dice with NaN where the class is absent from y, reduced value)"""
    axes = list(range(2, y_pred.dim()))
    inter = torch.sum(y * y_pred, dim=axes)
    y_o = torch.sum(y, dim=axes)
    denom = y_o + torch.sum(y_pred, dim=axes)
    f = torch.where(y_o > 0, (2.0 * inter) / denom, torch.tensor(float("nan")))
    raw = f.clone()
    nans = torch.isnan(f)
    not_nans = (~nans).float()
    f = f.clone()
    f[nans] = 0
    zero = torch.zeros(1)
    if reduction == "mean":
        nn_c = not_nans.sum(dim=1)
        f = torch.where(nn_c > 0, f.sum(dim=1) / nn_c, zero)
        nn_b = (nn_c > 0).float().sum(dim=0)
        f = torch.where(nn_b > 0, f.sum(dim=0) / nn_b, zero)
    elif reduction == "mean_batch":
        nn_b = not_nans.sum(dim=0)
        f = torch.where(nn_b > 0, f.sum(dim=0) / nn_b, zero)
    else:
        raise ValueError(reduction)
    return raw, f


# synthetic CT-like data lives in tools/synthetic.py (neither oracle nor product: bench.py's GPU leg must not need
# oracle/ to make its inputs); re-exported here so the tests keep one import for "the inputs both sides see"
from tools.synthetic import synthetic_volume  # noqa: E402,F401
