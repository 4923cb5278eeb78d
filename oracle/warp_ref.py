"""CPU ORACLE (test infrastructure only) for the homography warp and point projection.

Restates, with a pinned fp32 operation order, what the reference obtains from the
third-party package Kornia (``kornia>=0.5.0``, requirements.txt:1 - not vendored, not
installed here, so parity with Kornia itself is UNPINNED; see oracle/torch_ref.py):

* ``kornia.geometry.transform.HomographyWarper(h, w, mode, normalized_coordinates=True)``
  as used at models/reconstructor.py:100-107,116
* ``kornia.geometry.linalg.transform_points`` as used at models/reconstructor.py:124

Published Kornia 0.5/0.6 algorithm that is followed:
  create_meshgrid(h, w, normalized): xs = linspace(0, w-1, w); xs = (xs/(w-1) - 0.5)*2
  transform_points(T, p):            p_h = [p, 1];  q_h = T @ p_h;  q = from_homogeneous(q_h)
  convert_points_from_homogeneous:   s = 1/(z + 1e-8) if |z| > 1e-8 else 1;  q = s * q_h[:2]
  F.grid_sample(src, grid, mode, padding_mode='zeros', align_corners=False)

Every product/sum below is a separate, individually rounded fp32 torch op (no FMA, no
BLAS), in the order  ((t0*x + t1*y) + t2)  - this is the order the HIP kernel reproduces
bit for bit, which makes nearest-mode output comparisons exact.  The sampling stage
restates ATen's CPU grid sampler: unnormalise ``fma(u + 1, size/2, -0.5)``, nearest =
round-half-to-even, bilinear = floor-based 4 taps with zero contribution out of range;
``tests/test_oracle.py`` checks it against ``torch.nn.functional.grid_sample`` itself.
"""
import torch

EPS = 1e-8  # kornia convert_points_from_homogeneous default


def normalized_axis(n):
    """create_meshgrid axis: linspace(0, n-1, n) is exactly 0..n-1 in fp32; then (v/(n-1) - 0.5)*2."""
    v = torch.arange(n, dtype=torch.float32)
    return (v / float(n - 1) - 0.5) * 2.0


def _homography_apply(t, x, y):
    """rows of t applied to (x, y, 1) with the pinned order, then Kornia's de-homogenisation."""
    X = (t[0] * x + t[1] * y) + t[2]
    Y = (t[3] * x + t[4] * y) + t[5]
    Z = (t[6] * x + t[7] * y) + t[8]
    one = torch.ones_like(Z)
    scale = torch.where(torch.abs(Z) > EPS, one / (Z + EPS), one)
    return scale * X, scale * Y


def warp_grid(theta, h, w):
    """(B,1,3,3) or (B,3,3) homographies -> sampling grid (B,h,w,2) in normalised coords."""
    B = theta.shape[0]
    t = theta.reshape(B, 9).to(torch.float32)
    tt = [t[:, k].reshape(B, 1, 1) for k in range(9)]
    xn = normalized_axis(w).reshape(1, 1, w).expand(B, h, w)
    yn = normalized_axis(h).reshape(1, h, 1).expand(B, h, w)
    u, v = _homography_apply(tt, xn, yn)
    return torch.stack([u, v], dim=-1)


def unnormalize(coord, size):
    """ATen CPU grid sampler, align_corners=False: ``(coord + 1) * (size/2) - 0.5`` where the
    multiply-subtract is ONE fused operation (the vectorised kernel is compiled with FMA
    contraction).  Established empirically against torch 2.10 CPU: of the candidate
    roundings only fma(fl(coord + 1), size/2, -0.5) reproduces ``F.grid_sample(mode='nearest')``
    on every coordinate within +-9 ulp of every half-pixel boundary (sizes 97/360/640/720/1280).
    The fp64 product of two fp32 values is exact, so rounding the fp64 expression once to fp32
    is the fused result."""
    c1 = (coord + 1.0).to(torch.float64)
    return (c1 * (float(size) / 2.0) - 0.5).to(torch.float32)


def sample(src, grid, mode):
    """grid_sample(src (B,C,Hs,Ws), grid (B,h,w,2), mode, 'zeros', align_corners=False)."""
    B, C, Hs, Ws = src.shape
    px = unnormalize(grid[..., 0], Ws)
    py = unnormalize(grid[..., 1], Hs)
    flat = src.reshape(B, C, Hs * Ws)

    def tap(ix, iy):
        ok = (ix >= 0) & (ix < Ws) & (iy >= 0) & (iy < Hs)
        lin = (iy.clamp(0, Hs - 1) * Ws + ix.clamp(0, Ws - 1)).to(torch.int64)
        val = torch.gather(flat, 2, lin.reshape(B, 1, -1).expand(B, C, -1)).reshape(B, C, *ix.shape[1:])
        return val * ok.unsqueeze(1).to(src.dtype)

    if mode == "nearest":
        # values whose magnitude exceeds the int range are out of bounds anyway
        ix = torch.round(px).clamp(-2.0, Ws + 1.0).to(torch.int64)  # torch.round = half-to-even
        iy = torch.round(py).clamp(-2.0, Hs + 1.0).to(torch.int64)
        bad = ~(torch.isfinite(px) & torch.isfinite(py))
        out = tap(ix, iy)
        return out * (~bad).unsqueeze(1).to(src.dtype)
    if mode == "bilinear":
        pxc = torch.nan_to_num(px, nan=-10.0).clamp(-4.0, Ws + 3.0)
        pyc = torch.nan_to_num(py, nan=-10.0).clamp(-4.0, Hs + 3.0)
        x0 = torch.floor(pxc)
        y0 = torch.floor(pyc)
        wx1 = pxc - x0
        wx0 = 1.0 - wx1
        wy1 = pyc - y0
        wy0 = 1.0 - wy1
        x0i, y0i = x0.to(torch.int64), y0.to(torch.int64)
        out = tap(x0i, y0i) * (wy0 * wx0).unsqueeze(1)
        out = out + tap(x0i + 1, y0i) * (wy0 * wx1).unsqueeze(1)
        out = out + tap(x0i, y0i + 1) * (wy1 * wx0).unsqueeze(1)
        out = out + tap(x0i + 1, y0i + 1) * (wy1 * wx1).unsqueeze(1)
        return out
    raise ValueError(mode)


def homography_warp(theta, template, h, w, mode="bilinear"):
    """HomographyWarper(h, w, mode, normalized_coordinates=True)(template, theta).squeeze(1)
    - the call at models/reconstructor.py:116-118."""
    grid = warp_grid(theta, h, w)
    return sample(template.to(torch.float32), grid, mode).squeeze(1)


def transform_points(trans, points):
    """kornia.geometry.linalg.transform_points for trans (B,1,3,3)|(B,3,3), points (B,N,2)."""
    B = trans.shape[0]
    t = trans.reshape(B, 9).to(torch.float32)
    tt = [t[:, k].reshape(B, 1) for k in range(9)]
    u, v = _homography_apply(tt, points[..., 0], points[..., 1])
    return torch.stack([u, v], dim=-1)
