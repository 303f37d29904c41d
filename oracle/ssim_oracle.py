"""CPU restatement (torch, fp64 by default, autograd) of the image term of the loss RadegsModel inherits.

TEST INFRASTRUCTURE ONLY (never imported by collab_splats_amd).  PARITY UNPINNED: the term lives in third-party
code that is absent from /root/reference and not version-pinned by it -- `super().get_loss_dict` at
/root/reference/collab_splats/models/rade_gs_model.py:289 resolves to nerfstudio's SplatfactoModel
(`nerfstudio @ git+https://github.com/BasisResearch/nerfstudio.git`, /root/reference/pyproject.toml:40), whose
SSIM is `pytorch_msssim.SSIM(data_range=1.0, size_average=True, channel=3)`.  The reference's tests hold no
vector for it.  What is restated here is the published algorithm of those two packages:

  pytorch_msssim (1.0.0) ssim.py
      _fspecial_gauss_1d(size=11, sigma=1.5)   float32 exp(-(c^2) / (2 sigma^2)), normalised
      gaussian_filter                          separable F.conv2d, stride 1, NO padding, groups = channels
      _ssim                                    K = (0.01, 0.03); C = (K * data_range)^2; ssim_map = luminance * cs_map,
                                               per-channel mean over the valid (H - 10) x (W - 10) positions
      SSIM.forward (size_average=True)         mean over channels (and batch)
  nerfstudio models/splatfacto.py  SplatfactoModel.get_loss_dict (ssim_lambda = 0.2)
      Ll1 = |gt - pred|.mean();  simloss = 1 - ssim(gt[None, CHW], pred[None, CHW])
      main_loss = (1 - ssim_lambda) * Ll1 + ssim_lambda * simloss
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def gauss_window(size: int = 11, sigma: float = 1.5) -> torch.Tensor:
    coords = torch.arange(size, dtype=torch.float32) - size // 2
    g = torch.exp(-(coords ** 2) / (2 * sigma ** 2))
    return g / g.sum()


def _filter(x: torch.Tensor, win: torch.Tensor) -> torch.Tensor:
    """x [1,C,H,W]; the window along H, then along W (valid)."""
    C = x.shape[1]
    w = win.to(x.dtype)
    x = F.conv2d(x, w.view(1, 1, -1, 1).repeat(C, 1, 1, 1), groups=C)
    return F.conv2d(x, w.view(1, 1, 1, -1).repeat(C, 1, 1, 1), groups=C)


def ssim(gt: torch.Tensor, pred: torch.Tensor, data_range: float = 1.0) -> torch.Tensor:
    """Mean SSIM of two [H,W,C] images (the argument order of the call in splatfacto.py: X = gt, Y = pred; the index is
    symmetric)."""
    X = gt.permute(2, 0, 1)[None]
    Y = pred.permute(2, 0, 1)[None]
    win = gauss_window()
    C1, C2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    mu1, mu2 = _filter(X, win), _filter(Y, win)
    mu1_sq, mu2_sq, mu1_mu2 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = _filter(X * X, win) - mu1_sq
    s2 = _filter(Y * Y, win) - mu2_sq
    s12 = _filter(X * Y, win) - mu1_mu2
    cs_map = (2 * s12 + C2) / (s1 + s2 + C2)
    ssim_map = ((2 * mu1_mu2 + C1) / (mu1_sq + mu2_sq + C1)) * cs_map
    return ssim_map.flatten(2).mean(-1).mean()


def main_loss(pred: torch.Tensor, gt: torch.Tensor, ssim_lambda: float = 0.2) -> torch.Tensor:
    l1 = torch.abs(gt - pred).mean()
    return (1 - ssim_lambda) * l1 + ssim_lambda * (1 - ssim(gt, pred))
