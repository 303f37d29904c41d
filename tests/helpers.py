"""Shared test helpers (tests only)."""
import os

import numpy as np
import torch


def rel_err(a, b) -> float:
    """tensor-inf-norm relative error  max|a-b| / max|b|  (SURVEY.md section 8(d))."""
    a = np.asarray(a.detach().cpu() if torch.is_tensor(a) else a, dtype=np.float64)
    b = np.asarray(b.detach().cpu() if torch.is_tensor(b) else b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)) if a.size else 0.0


def small_scene(n=300, W=48, H=40, seed=0, dtype=torch.float64, sh_k=16):
    g = torch.Generator().manual_seed(seed)
    means = torch.rand(n, 3, generator=g, dtype=dtype) * torch.tensor([3.0, 2.4, 6.0], dtype=dtype) + \
        torch.tensor([-1.5, -1.2, 1.5], dtype=dtype)
    quats = torch.randn(n, 4, generator=g, dtype=dtype)
    scales = torch.exp(torch.rand(n, 3, generator=g, dtype=dtype) * 2.5 - 4.0)
    opac = torch.sigmoid(torch.rand(n, generator=g, dtype=dtype) * 6 - 2)
    sh = torch.randn(n, sh_k, 3, generator=g, dtype=dtype) * 0.3
    th = 0.3
    R = torch.tensor([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]], dtype=dtype)
    V = torch.eye(4, dtype=dtype)
    V[:3, :3] = R
    V[:3, 3] = torch.tensor([0.1, -0.2, 0.3], dtype=dtype)
    K = torch.tensor([[60.0, 0, W / 2], [0, 55.0, H / 2], [0, 0, 1]], dtype=dtype)
    return dict(means=means, quats=quats, scales=scales, opacities=opac, sh=sh, viewmat=V, K=K, W=W, H=H)


def upstream(shapes, seed=5, dtype=torch.float64):
    g = torch.Generator().manual_seed(seed)
    return [torch.rand(s, generator=g, dtype=dtype) for s in shapes]


MARGIN_TOL = 8 * 2.0 ** -24   # 8 units of the fp32 roundoff in the sigma evaluation (cr_blend_margin): below this two
                              # correct fp32 implementations may branch differently.  Measured on MI355X
                              # (scripts/flip_margins.py): every differing pixel sits at <= 1.2 units.


class FlipProof:
    """Evidence that an out-of-tolerance element is a THRESHOLD FLIP and nothing else.

    Built from the C restatement's per-pixel margin map (oracle/craster.c::cr_blend_margin: the smallest rounding
    error of the sigma evaluation, in units of the size of its terms, that could move alpha across alpha_min, T'
    across t_stop or T across median_t for any Gaussian the pixel traverses).
    A pixel may only miss the tolerance if its margin is below MARGIN_TOL.  A per-Gaussian gradient row may only miss
    it if the Gaussian's screen-space box (mean2d +- radii) contains a pixel that ACTUALLY DIFFERS between the two
    implementations -- an image value out of tolerance, or unequal ``last_ids`` / ``median_ids`` -- and that pixel has
    itself been proven to sit on a threshold (``check_pixels`` records every differing pixel it has accepted; call it
    for the five images and the two index maps BEFORE the gradient tensors).  A box that merely covers some
    low-margin pixel where nothing flipped explains nothing and fails."""

    def __init__(self, margin, means2d=None, radii=None, margin_tol=MARGIN_TOL):
        self.margin = np.asarray(margin, dtype=np.float64)
        self.low = self.margin < margin_tol
        self.margin_tol = margin_tol
        self.means2d = None if means2d is None else np.asarray(means2d, dtype=np.float64)
        self.radii = None if radii is None else np.asarray(radii, dtype=np.int64)
        self.flipped = np.zeros_like(self.low)           # pixels that differ AND were proven to sit on a threshold
        self.pixel_checks = 0
        self.rows_checked = 0                            # out-of-tolerance gradient rows seen / explained by a flip
        self.rows_loose_only = 0                         # ... that only the old rule (any low-margin pixel) would pass

    @staticmethod
    def _sat(mask):
        sat = np.zeros((mask.shape[0] + 1, mask.shape[1] + 1), dtype=np.int64)
        sat[1:, 1:] = mask.astype(np.int64).cumsum(0).cumsum(1)
        return sat

    def check_pixels(self, bad_mask, name):
        """bad_mask [H, W] bool (pixels that differ): every one must sit on a threshold; they are remembered as the
        flips that may explain gradient rows."""
        self.pixel_checks += 1
        unexplained = bad_mask & ~self.low
        assert not unexplained.any(), (
            f"{name}: {int(unexplained.sum())} pixel(s) miss the tolerance with threshold margin >= {self.margin_tol} "
            f"(first at {tuple(np.argwhere(unexplained)[0])}, margin {self.margin[unexplained].min():.3e}): not a flip")
        self.flipped |= bad_mask

    def check_ids(self, got_last, ref_last, got_med, ref_med, max_frac=1e-4):
        """The contributor index maps agree except at proven flips (and those pixels count as differing)."""
        for key, got, ref in (("last_ids", got_last, ref_last), ("median_ids", got_med, ref_med)):
            diff = np.asarray(got) != np.asarray(ref)
            assert diff.sum() <= max(2, max_frac * diff.size), (key, int(diff.sum()))
            self.check_pixels(diff, key)

    def _box_counts(self, sat, rows):
        H, W = self.low.shape
        mx, my = self.means2d[rows, 0], self.means2d[rows, 1]
        rx, ry = self.radii[rows, 0], self.radii[rows, 1]
        x0 = np.clip(np.floor(mx - rx - 1), 0, W).astype(np.int64); x1 = np.clip(np.ceil(mx + rx + 1), 0, W).astype(np.int64)
        y0 = np.clip(np.floor(my - ry - 1), 0, H).astype(np.int64); y1 = np.clip(np.ceil(my + ry + 1), 0, H).astype(np.int64)
        return sat[y1, x1] - sat[y0, x1] - sat[y1, x0] + sat[y0, x0]

    def check_rows(self, bad_rows, name):
        """bad_rows: indices of Gaussians whose gradient misses the tolerance."""
        assert self.means2d is not None, "FlipProof needs means2d/radii for gradient tensors"
        assert self.pixel_checks > 0, "FlipProof: check the images / index maps before the gradient rows"
        rows = np.asarray(bad_rows, dtype=np.int64).reshape(-1)
        if rows.size == 0:
            return
        strict = self._box_counts(self._sat(self.flipped), rows) > 0
        loose = self._box_counts(self._sat(self.low), rows) > 0
        self.rows_checked += int(rows.size)
        self.rows_loose_only += int((loose & ~strict).sum())
        msg = (f"[FlipProof] {name}: {rows.size} out-of-tolerance row(s), {int(strict.sum())} cover a pixel that differs, "
               f"{int((loose & ~strict).sum())} cover only a low-margin pixel where nothing flipped (rejected), "
               f"{int((~loose).sum())} cover no threshold pixel at all; {int(self.flipped.sum())} differing pixel(s) on record, "
               f"{int(self.low.sum())} low-margin pixel(s) in the image")
        print(msg)
        if os.environ.get("FLIPPROOF_LOG"):                  # (pytest swallows the output of passing tests)
            with open(os.environ["FLIPPROOF_LOG"], "a") as f:
                f.write(msg + "\n")
        assert strict.all(), (
            f"{name}: Gaussian {int(rows[~strict][0])} (+{int((~strict).sum()) - 1} more) misses the tolerance but its screen box "
            f"covers no pixel that differs between the two implementations: not a flip")


def assert_close_flips(got, ref, name="", tol=1e-4, outlier_frac=2e-5, outlier_tol=None, min_outliers=2, proof=None):
    """Tensor-inf-norm relative comparison that tolerates fp32 THRESHOLD FLIPS -- and only those.

    The algorithm is discontinuous at alpha == 1/255 (skip), T == 1e-4 (stop) and T == 0.5 (median):
    two correct fp32 implementations (v_exp_f32 vs libm expf) can take different branches for an
    isolated (pixel, Gaussian) pair sitting on a threshold, which moves that pixel by up to
    ~1/255 of a contribution (hard part 3).  So: all but `outlier_frac` of the
    elements must meet `tol`, and the rest must stay within `outlier_tol` (both relative to max|ref|).
    On small images one flipped pixel already exceeds the fraction, so `min_outliers` pixels (rows of the last
    dimension) are always allowed; a flip of the MEDIAN depth moves that pixel to the depth of a neighbouring Gaussian, i.e. by
    anything inside the depth range, so tensors named "*med_depth*" get `outlier_tol` = 1.

    With ``proof`` (a FlipProof) the blanket allowance becomes a demonstrated one: every out-of-tolerance pixel of an
    image [H, W, ...] must have a threshold margin < MARGIN_TOL, and every out-of-tolerance row of a per-Gaussian
    tensor [N, ...] must cover such a pixel; anything else fails.
    """
    if outlier_tol is None:
        outlier_tol = 1.0 if "med_depth" in name else 2e-2
    a = np.asarray(got.detach().cpu() if torch.is_tensor(got) else got, dtype=np.float64)
    b = np.asarray(ref.detach().cpu() if torch.is_tensor(ref) else ref, dtype=np.float64)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    scale = max(np.abs(b).max(), 1e-30)
    d = np.abs(a - b) / scale
    frac = float((d > tol).mean())
    width = a.shape[-1] if a.ndim > 1 else 1                     # a flipped pixel moves all its channels
    allowed = max(outlier_frac, (min_outliers * width + 0.5) / max(d.size, 1))
    if proof is not None:
        # every outlier is checked individually below; one flipped pixel moves the gradients of all the Gaussians
        # composited behind it, so the count of (proven) outliers only gets a loose sanity bound
        allowed = max(allowed, 20 * outlier_frac)
    assert frac <= allowed, f"{name}: {frac:.2e} of elements exceed {tol} (max {d.max():.3e})"
    assert d.max() <= outlier_tol, f"{name}: max rel err {d.max():.3e} > {outlier_tol}"
    if proof is not None and frac > 0:
        bad = d > tol
        if a.shape[:2] == proof.low.shape:                       # image [H, W] or [H, W, C]
            proof.check_pixels(bad.reshape(bad.shape[0], bad.shape[1], -1).any(-1), name)
        else:                                                    # per-Gaussian [N] or [N, ...]
            proof.check_rows(np.nonzero(bad.reshape(bad.shape[0], -1).any(-1))[0], name)
    return float(d.max()), frac
