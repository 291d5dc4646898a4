"""TEST INFRASTRUCTURE ONLY -- float64 CPU oracle of the TRASE rasterizer path.

What it restates (reference file:line it follows):

* call contract of ``GaussianRasterizer(raster_settings)(means3D, means2D, shs,
  sh_objs, colors_precomp, opacities, scales, rotations, cov3D_precomp) ->
  (image, radii, feats, depth)`` -- gaussian_renderer/__init__.py:58-73,137-146
* matrix conventions (row-vector, transposed storage) -- scene/cameras.py:76-79,
  utils/graphics_utils.py:38-71
* SH basis constants / polynomial order -- utils/sh_utils.py:26-43,57-112 and
  the ``+0.5, clamp_min 0`` colour rule -- gaussian_renderer/__init__.py:105-108
* quaternion (r,x,y,z) -> rotation matrix and Sigma = R S S^T R^T --
  utils/general_utils.py:122-154, scene/gaussian_model.py:37-41
* everything that lives inside the absent CUDA extension follows SURVEY.md
  Appendix A (public 3DGS lineage + gaussian-grouping ``sh_objs`` channels +
  Deformable-3DGS blended depth).  PARITY UNPINNED for that part: no source,
  no tests, no golden vectors exist in /root/reference.

Every lineage assumption is a keyword switch of :class:`OracleOptions`.

Gradients come from ``torch.autograd`` on this restatement (float64); the two
places where the lineage's hand-written backward is *not* the derivative of
its forward are reproduced with detach tricks (see ``lineage_grads``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import torch

TILE = 16  # the lineage's BLOCK_X == BLOCK_Y == 16; part of the visible semantics
            # because tile-rect membership decides which Gaussians a pixel sees.

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
         -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658,
         0.3731763325901154, -0.4570457994644658, 1.445305721320277,
         -0.5900435899266435)


@dataclass
class OracleOptions:
    # Appendix A switches -----------------------------------------------------
    near_cull_z: float = 0.2          # cull if view-space z <= this
    lowpass: float = 0.3              # added to both diagonal entries of cov2D
    alpha_max: float = 0.99
    alpha_min: float = 1.0 / 255.0
    t_stop: float = 1e-4
    feats_bg: bool = False            # False: features get no background term; True: feats[c] += T_final * feat_bg_value
    feat_bg_value: float = 0.0
    depth_normalised: bool = False    # False: blended depth as is; True: depth = sum(w z) / (1 - T_final)
    lineage_grads: bool = True        # straight-through 0.99 clamp; frozen clamped tx/tz
    # fragility margins (relative) used to flag pixels whose discrete decisions
    # may legitimately flip between float32 and float64 arithmetic
    frag_rel: float = 2e-5
    # with the device's own float32 per-Gaussian state at hand (device_view incl. conic_opacity) only the device's
    # *arithmetic* in the blend loop is unknown: exponent polynomial + exp2 in float32, ~2e-6 relative in alpha
    frag_rel_arith: float = 5e-6


def _f64(x):
    return None if x is None else x.to(torch.float64)


def build_rotation_unnormalised(q: torch.Tensor) -> torch.Tensor:
    """Entries as utils/general_utils.py:134-142, but the quaternion is used as
    given (Appendix A.3: the CUDA path does not normalise)."""
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=-1)
    return R.reshape(-1, 3, 3)


def cov3d_from_scale_rot(scales, rotations, scale_modifier):
    """Sigma = R S S R^T, stored as (xx,xy,xz,yy,yz,zz) like strip_lowerdiag
    (utils/general_utils.py:108-119)."""
    R = build_rotation_unnormalised(rotations)
    L = R * (scale_modifier * scales)[:, None, :]
    S = L @ L.transpose(1, 2)
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=-1)


def eval_sh_colors(deg: int, shs: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """shs: (N, M, 3) coefficient-major (scene/gaussian_model.py:194-197);
    same polynomial as utils/sh_utils.py:74-100.  Returns (N,3) before +0.5."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = SH_C0 * shs[:, 0]
    if deg > 0:
        res = res - SH_C1 * y * shs[:, 1] + SH_C1 * z * shs[:, 2] - SH_C1 * x * shs[:, 3]
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * shs[:, 4] + SH_C2[1] * yz * shs[:, 5]
                   + SH_C2[2] * (2.0 * zz - xx - yy) * shs[:, 6]
                   + SH_C2[3] * xz * shs[:, 7] + SH_C2[4] * (xx - yy) * shs[:, 8])
            if deg > 2:
                res = (res + SH_C3[0] * y * (3 * xx - yy) * shs[:, 9]
                       + SH_C3[1] * xy * z * shs[:, 10]
                       + SH_C3[2] * y * (4 * zz - xx - yy) * shs[:, 11]
                       + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * shs[:, 12]
                       + SH_C3[4] * x * (4 * zz - xx - yy) * shs[:, 13]
                       + SH_C3[5] * z * (xx - yy) * shs[:, 14]
                       + SH_C3[6] * x * (xx - 3 * yy) * shs[:, 15])
    return res


@dataclass
class Geom:
    """Per-Gaussian forward state (float64, differentiable where meaningful)."""
    valid: torch.Tensor        # (N,) bool: survives all culls
    radii: torch.Tensor        # (N,) int32 (0 where culled)
    xy: torch.Tensor           # (N,2) pixel-space centre
    depth: torch.Tensor        # (N,) view-space z
    conic: torch.Tensor        # (N,3) (A,B,C) of the inverse 2D covariance
    rgb: torch.Tensor          # (N,3)
    rect: torch.Tensor         # (N,4) int64 tile rect x0,y0,x1,y1 (half-open)
    tiles_touched: torch.Tensor
    frag_gauss: torch.Tensor   # (N,) bool: a discrete preprocess decision is borderline
    cov3d: torch.Tensor        # (N,6)
    ndc: torch.Tensor          # (N,2) NDC centre; carries the means2D gradient hook
    frag_radius: torch.Tensor = None   # (N,) bool: ceil() of the radius is borderline
    frag_rect: torch.Tensor = None     # (N,) bool: a tile-rect edge / the near cull is borderline
    r_real: torch.Tensor = None        # (N,) un-rounded radius


def preprocess(settings, means3D, shs, colors_precomp, opacities, scales, rotations,
               cov3D_precomp, means2D=None, opt: OracleOptions = OracleOptions()) -> Geom:
    """Appendix A 'Preprocess' items 1-9."""
    W, H = int(settings.image_width), int(settings.image_height)
    vm = _f64(settings.viewmatrix).reshape(4, 4)      # stored transposed (row-vector form)
    pm = _f64(settings.projmatrix).reshape(4, 4)
    campos = _f64(settings.campos).reshape(3)
    tanx, tany = float(settings.tanfovx), float(settings.tanfovy)
    fx, fy = W / (2.0 * tanx), H / (2.0 * tany)
    p = means3D
    N = p.shape[0]
    # 1. view space (row-vector convention == column-major read of the flat matrix)
    t = p @ vm[:3, :3] + vm[3, :3]
    in_front = t[:, 2] > opt.near_cull_z
    # 2. clip space
    ph = p @ pm[:3, :] + pm[3, :]
    pw = 1.0 / (ph[:, 3] + 1e-7)
    ndc = ph[:, :2] * pw[:, None]
    if means2D is not None:
        # means2D is the zero dummy whose .grad must receive dL/d(ndc)
        # (gaussian_renderer/__init__.py:48-52; Appendix A 'Render bwd')
        ndc = ndc + means2D[:, :2]
    # 3. covariance
    if cov3D_precomp is not None:
        cov3d = cov3D_precomp
    else:
        cov3d = cov3d_from_scale_rot(scales, rotations, float(settings.scale_modifier))
    Sig = torch.stack([cov3d[:, 0], cov3d[:, 1], cov3d[:, 2],
                       cov3d[:, 1], cov3d[:, 3], cov3d[:, 4],
                       cov3d[:, 2], cov3d[:, 4], cov3d[:, 5]], dim=-1).reshape(N, 3, 3)
    # 4. EWA
    tz = t[:, 2]
    tz_safe = torch.where(in_front, tz, torch.ones_like(tz))
    limx, limy = 1.3 * tanx, 1.3 * tany
    txtz, tytz = t[:, 0] / tz_safe, t[:, 1] / tz_safe
    cx = (txtz < -limx) | (txtz > limx)
    cy = (tytz < -limy) | (tytz > limy)
    txc = txtz.clamp(-limx, limx) * tz_safe
    tyc = tytz.clamp(-limy, limy) * tz_safe
    if opt.lineage_grads:
        # the lineage replaces t.x by the clamped value and then treats it as an
        # independent variable with zero incoming gradient when clamped
        txc = torch.where(cx, txc.detach(), t[:, 0])
        tyc = torch.where(cy, tyc.detach(), t[:, 1])
    zero = torch.zeros_like(tz_safe)
    J = torch.stack([fx / tz_safe, zero, -fx * txc / (tz_safe * tz_safe),
                     zero, fy / tz_safe, -fy * tyc / (tz_safe * tz_safe)], dim=-1).reshape(N, 2, 3)
    Wm = vm[:3, :3].T                                   # math rotation: t = Wm p + trans
    A = J @ Wm
    cov2 = A @ Sig @ A.transpose(1, 2)
    a = cov2[:, 0, 0] + opt.lowpass
    b = cov2[:, 0, 1]
    c = cov2[:, 1, 1] + opt.lowpass
    # 5. conic + radius
    det = a * c - b * b
    det_ok = det != 0
    det_safe = torch.where(det_ok, det, torch.ones_like(det))
    conic = torch.stack([c / det_safe, -b / det_safe, a / det_safe], dim=-1)
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
    r_real = 3.0 * torch.sqrt(torch.clamp_min(lam, 0.0))
    radius = torch.ceil(r_real.detach())
    # 6. pixel centre and tile rect
    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    pxd, pyd = px.detach(), py.detach()

    def _rect(lo_real, hi_real, g):
        lo = torch.trunc(lo_real / TILE).clamp(0, g)
        hi = torch.trunc(hi_real / TILE).clamp(0, g)
        return lo.to(torch.int64), hi.to(torch.int64)

    x0, x1 = _rect(pxd - radius, pxd + radius + (TILE - 1), gx)
    y0, y1 = _rect(pyd - radius, pyd + radius + (TILE - 1), gy)
    area = (x1 - x0) * (y1 - y0)
    finite = torch.isfinite(r_real.detach()) & torch.isfinite(pxd) & torch.isfinite(pyd)
    valid = in_front & det_ok & (area > 0) & finite
    radii = torch.where(valid, radius, torch.zeros_like(radius)).to(torch.int32)
    # fragility of the discrete decisions: distance of the real-valued quantities from the integer
    # boundaries that ceil()/trunc() cut at, measured against a float32 error model
    # (pixel centre: a few ulp of the NDC value scaled by W/2; radius: a few ulp, relative)
    rr = r_real.detach()
    frag_radius = torch.abs(rr - torch.round(rr)) < (4e-6 * rr + 1e-6)
    err_px = 6e-7 * max(W, H) + 1e-6
    frag_rect = torch.zeros_like(frag_radius)
    for v in (pxd - radius, pxd + radius + (TILE - 1), pyd - radius, pyd + radius + (TILE - 1)):
        q = v / TILE
        frag_rect = frag_rect | (torch.abs(q - torch.round(q)) < err_px / TILE)
    frag_rect = frag_rect | (torch.abs(tz.detach() - opt.near_cull_z) < 1e-6)
    frag_radius = frag_radius & in_front & finite
    frag_rect = frag_rect & in_front & finite
    frag = frag_radius | frag_rect
    # 7. colour
    if colors_precomp is not None:
        rgb = colors_precomp
    else:
        d = p - campos[None, :]
        d = d / torch.sqrt((d * d).sum(-1, keepdim=True))
        rgb = torch.clamp_min(eval_sh_colors(int(settings.sh_degree), shs, d) + 0.5, 0.0)
    rect = torch.stack([x0, y0, x1, y1], dim=-1)
    rect = torch.where(valid[:, None], rect, torch.zeros_like(rect))
    return Geom(valid=valid, radii=radii, xy=torch.stack([px, py], -1), depth=tz, conic=conic,
                rgb=rgb, rect=rect, tiles_touched=torch.where(valid, area, torch.zeros_like(area)),
                frag_gauss=frag, cov3d=cov3d, ndc=ndc, frag_radius=frag_radius, frag_rect=frag_rect, r_real=rr)


def blend_tile(xy, conic, opac, depth, chans, pixx, pixy, opt: OracleOptions, dev=None):
    """Appendix A 'Render fwd' for one tile.

    xy (G,2), conic (G,3), opac (G,), depth (G,), chans (G,Cc) already in blend
    order; pixx/pixy (P,) pixel coordinates.  Returns (out (P,Cc), T_final (P,),
    n_contrib (P,), fragile (P,) bool, blended (G,P) bool).

    dev: optional (xy_d (G,2), conic_d (G,3), opac_d (G,)) -- the DEVICE's float32 per-Gaussian state,
    promoted to float64.  Used only for the fragility flags: a pixel is fragile when the same gates,
    evaluated exactly on the device's rounded inputs, decide differently from the oracle's, or when the
    device-input value lies within float32 *arithmetic* rounding of a threshold.  Without it the flags
    fall back to a fixed relative margin around the oracle's own values."""
    G = xy.shape[0]
    P = pixx.shape[0]
    dx = xy[:, 0:1] - pixx[None, :]
    dy = xy[:, 1:2] - pixy[None, :]
    power = -0.5 * (conic[:, 0:1] * dx * dx + conic[:, 2:3] * dy * dy) - conic[:, 1:2] * dx * dy
    keep_p = power <= 0
    raw = opac[:, None] * torch.exp(torch.where(keep_p, power, torch.zeros_like(power)))
    if opt.lineage_grads:
        alpha = raw + (torch.clamp_max(raw, opt.alpha_max) - raw).detach()   # straight-through clamp
    else:
        alpha = torch.clamp_max(raw, opt.alpha_max)
    keep = keep_p & (alpha.detach() >= opt.alpha_min)
    a_eff = torch.where(keep, alpha, torch.zeros_like(alpha))
    one_m = 1.0 - a_eff
    cp = torch.cumprod(one_m, dim=0)                       # test_T after each Gaussian
    alive = cp.detach() >= opt.t_stop                       # prefix property (cp is monotone)
    T_excl = torch.cat([torch.ones(1, P, dtype=cp.dtype), cp[:-1]], dim=0)
    w = torch.where(alive & keep, a_eff * T_excl, torch.zeros_like(a_eff))
    out = w.transpose(0, 1) @ chans                         # (P,Cc)
    T_final = torch.where(alive, cp, torch.ones_like(cp)).amin(dim=0) if G > 0 else torch.ones(P, dtype=torch.float64)
    idx = torch.arange(1, G + 1, dtype=torch.int64)[:, None]
    n_contrib = torch.where(alive & keep, idx, torch.zeros_like(idx)).amax(dim=0) if G > 0 else torch.zeros(P, dtype=torch.int64)
    # fragility: decisions that may flip under float32 evaluation, only where they matter
    # (before the pixel is done)
    rel = opt.frag_rel
    live_before = torch.cat([torch.ones(1, P, dtype=torch.bool), alive[:-1]], dim=0)
    if dev is None:
        ad = alpha.detach()
        f_alpha = (torch.abs(ad - opt.alpha_min) < 4 * rel * opt.alpha_min) & keep_p
        f_pow = (torch.abs(power.detach()) < 1e-6) & (opac[:, None].detach() >= opt.alpha_min)
        f_stop = (torch.abs(cp.detach() - opt.t_stop) < 8 * rel * opt.t_stop) & keep
        fragile = ((f_alpha | f_pow | f_stop) & live_before).any(dim=0)
    else:
        with torch.no_grad():
            xy_d, conic_d, opac_d = dev
            dxd = xy_d[:, 0:1] - pixx[None, :]
            dyd = xy_d[:, 1:2] - pixy[None, :]
            power_d = -0.5 * (conic_d[:, 0:1] * dxd * dxd + conic_d[:, 2:3] * dyd * dyd) - conic_d[:, 1:2] * dxd * dyd
            keep_pd = power_d <= 0
            alpha_d = torch.clamp_max(opac_d[:, None] * torch.exp(torch.where(keep_pd, power_d, torch.zeros_like(power_d))),
                                      opt.alpha_max)
            keep_d = keep_pd & (alpha_d >= opt.alpha_min)
            cp_d = torch.cumprod(1.0 - torch.where(keep_d, alpha_d, torch.zeros_like(alpha_d)), dim=0)
            alive_d = cp_d >= opt.t_stop
            live_before_d = torch.cat([torch.ones(1, P, dtype=torch.bool), alive_d[:-1]], dim=0)
            lb = live_before | live_before_d
            # The device evaluates e = log2(e) * power + log2(opacity) as a polynomial about the origin of the pixel's
            # 8x8 block (trase_amd/csrc/common.h pair_poly / poly_eval); its float32 rounding error scales with the size
            # of the cancelling terms, i.e. with the exponent's value at the block origin, not with e itself.
            L2E = 1.4426950408889634
            bxo = torch.floor(pixx / 8.0) * 8.0
            byo = torch.floor(pixy / 8.0) * 8.0
            RX = (xy_d[:, 0:1] - bxo[None, :]).abs()
            RY = (xy_d[:, 1:2] - byo[None, :]).abs()
            lo2 = torch.log2(opac_d.clamp_min(1e-30))[:, None]
            M = L2E * (0.5 * (conic_d[:, 0:1].abs() * RX * RX + conic_d[:, 2:3].abs() * RY * RY)
                       + conic_d[:, 1:2].abs() * RX * RY) + lo2.abs()
            marg = torch.clamp(8 * 5.96e-8 * (M + 8.0), min=4 * opt.frag_rel_arith * L2E)     # absolute, in log2 units
            e_d = L2E * power_d + lo2
            f_alpha = (torch.abs(e_d - math.log2(opt.alpha_min)) < marg) & keep_pd
            f_pow = (torch.abs(L2E * power_d) < marg) & (opac_d[:, None] >= opt.alpha_min)
            f_stop = (torch.abs(cp_d - opt.t_stop) < 8 * rel * opt.t_stop) & keep_d
            flip = (keep != keep_d) | (alive != alive_d)
            fragile = ((f_alpha | f_pow | f_stop | flip) & lb).any(dim=0)
    return out, T_final, n_contrib, fragile, (alive & keep)


@dataclass
class OracleOut:
    image: torch.Tensor      # (3,H,W)
    radii: torch.Tensor      # (N,) int32
    feats: torch.Tensor      # (F,H,W)
    depth: torch.Tensor      # (1,H,W)
    fragile: torch.Tensor    # (H,W) bool: pixel has a borderline discrete decision
    frag_gauss: torch.Tensor # (N,) bool
    num_rendered: int        # R = sum of tiles touched (lineage definition)
    geom: Geom
    final_T: torch.Tensor    # (H,W)
    n_contrib: torch.Tensor  # (H,W)
    pairs_done: int = 0      # (tile,Gaussian) pairs actually composited (== num_rendered unless sampled)
    tile_mask: torch.Tensor = None   # (H,W) bool: pixels of the tiles that were composited (all, unless sampled)
    frag_stats: dict = None          # fragile-pixel counts by cause: gate / order / gauss


def rasterize(settings, means3D, means2D=None, shs=None, sh_objs=None, colors_precomp=None,
              opacities=None, scales=None, rotations=None, cov3D_precomp=None,
              opt: OracleOptions = OracleOptions(), sort_depth: Optional[torch.Tensor] = None,
              radii_override: Optional[torch.Tensor] = None, tile_step: int = 1,
              max_seconds: Optional[float] = None, device_view: Optional[dict] = None,
              tiles: Optional[list] = None) -> OracleOut:
    """Full forward.  All float inputs are promoted to float64 (gradients flow
    back to the caller's leaves through the promotion).

    sort_depth: optional (N,) float32 keys defining the blend order (e.g. the
    device's own view-space depths) -- removes order flips between float32 and
    float64 depth evaluation from the comparison.
    radii_override: optional (N,) int radii from the device, used *only* for
    Gaussians the oracle marks fragile.
    tile_step: render only every tile_step-th tile (bounded sample for bench.py's
    cpu_baseline timing); skipped tiles keep the background.
    max_seconds: stop compositing further tiles once this much wall time has been
    spent in the tile loop (bounded sample); OracleOut.pairs_done says how far it got.
    device_view: optional {"radii": (N,) int, "xy": (N,2) float32, "depth": (N,) float32[, "conic_opacity":
    (N,4) float32]} -- the device's own per-Gaussian forward state.  Used ONLY to settle decisions that are legitimately
    ambiguous between float32 and float64 evaluation: the blend order (float32 depth keys), and --
    for Gaussians this oracle marks borderline -- the ceil() of the radius, the truncated tile-rect
    edges and the near cull, each adopted only when it is one of the admissible neighbours of the
    oracle's own value.  Every value that is compared (maps, gradients) is still the oracle's.
    tiles: optional explicit list of (tx, ty) 16x16 tiles to composite (others keep the background);
    OracleOut.tile_mask marks their pixels."""
    if (shs is None) == (colors_precomp is None):
        raise ValueError("Please provide excatly one of either SHs or precomputed colors!")
    if ((scales is None or rotations is None) and cov3D_precomp is None) or \
       ((scales is not None or rotations is not None) and cov3D_precomp is not None):
        raise ValueError("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
    W, H = int(settings.image_width), int(settings.image_height)
    means3D = _f64(means3D)
    N = means3D.shape[0]
    shs, colors_precomp, opacities = _f64(shs), _f64(colors_precomp), _f64(opacities)
    scales, rotations, cov3D_precomp = _f64(scales), _f64(rotations), _f64(cov3D_precomp)
    means2D = _f64(means2D)
    F = 0 if sh_objs is None else sh_objs.shape[-1]
    feats_in = None if sh_objs is None else _f64(sh_objs).reshape(N, F)
    g = preprocess(settings, means3D, shs, colors_precomp, opacities, scales, rotations,
                   cov3D_precomp, means2D, opt)
    if radii_override is not None and bool(g.frag_radius.any()):
        g = _apply_radii_override(g, radii_override, W, H)
    if device_view is not None:
        g = _apply_device_view(g, device_view, W, H, opt)
        if sort_depth is None:
            dv_depth = device_view["depth"].detach().cpu().to(torch.float32)
            dv_vis = device_view["radii"].detach().cpu() > 0
            # culled-on-device Gaussians have no stored depth; they only matter if the oracle keeps them (then unresolved)
            sort_depth = torch.where(dv_vis, dv_depth, g.depth.detach().to(torch.float32))
    bg = _f64(settings.bg).reshape(3)
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    C = 3 + F + 1
    out = torch.zeros(H, W, C, dtype=torch.float64)
    out[..., :3] = out[..., :3] + bg              # pixels of empty tiles show background
    if opt.feats_bg and F:
        out[..., 3:3 + F] = opt.feat_bg_value
    final_T = torch.ones(H, W, dtype=torch.float64)
    n_contrib = torch.zeros(H, W, dtype=torch.int64)
    fragile = torch.zeros(H, W, dtype=torch.bool)
    tile_mask = torch.zeros(H, W, dtype=torch.bool)
    tile_set = None if tiles is None else {(int(a), int(b)) for a, b in tiles}
    frag_stats = {"gate": 0, "order": 0, "gauss": 0}
    opac = opacities.reshape(N)
    chans_all = torch.cat([g.rgb] + ([feats_in] if F else []) + [g.depth[:, None]], dim=-1)
    key = g.depth.detach() if sort_depth is None else sort_depth.to(torch.float64)
    key32 = key.to(torch.float32)
    rect = g.rect
    vidx = torch.nonzero(g.valid).reshape(-1)
    import time as _time
    t_loop = _time.perf_counter()
    pairs_done = 0
    stop = False
    dev_state = None
    if device_view is not None and device_view.get("conic_opacity") is not None:
        # the device's float32 (centre, conic, opacity), promoted; Gaussians the device culled keep the oracle's values
        dvis = (device_view["radii"].detach().cpu() > 0)[:, None]
        co = device_view["conic_opacity"].detach().cpu().to(torch.float64)
        dev_state = (torch.where(dvis, device_view["xy"].detach().cpu().to(torch.float64), g.xy.detach()),
                     torch.where(dvis, co[:, :3], g.conic.detach()),
                     torch.where(dvis[:, 0], co[:, 3], opac.detach()))
    rows_needed = None if tile_set is None else {b for _, b in tile_set}
    for ty in range(gy):
        if stop:
            break
        if rows_needed is not None and ty not in rows_needed:
            continue
        in_row = vidx[(rect[vidx, 1] <= ty) & (rect[vidx, 3] > ty)]
        if in_row.numel() == 0:
            continue
        for tx in range(gx):
            if tile_step > 1 and (ty * gx + tx) % tile_step != 0:
                continue
            if tile_set is not None and (tx, ty) not in tile_set:
                continue
            tile_mask[ty * TILE:min(ty * TILE + TILE, H), tx * TILE:min(tx * TILE + TILE, W)] = True
            ids = in_row[(rect[in_row, 0] <= tx) & (rect[in_row, 2] > tx)]
            if ids.numel() == 0:
                continue
            if max_seconds is not None and _time.perf_counter() - t_loop > max_seconds:
                stop = True
                break
            pairs_done += int(ids.numel())
            # stable sort on the float32 depth bits, ties -> ascending Gaussian index
            order = torch.sort(key32[ids], stable=True).indices
            ids = ids[order]
            x_lo, y_lo = tx * TILE, ty * TILE
            x_hi, y_hi = min(x_lo + TILE, W), min(y_lo + TILE, H)
            ys, xs = torch.meshgrid(torch.arange(y_lo, y_hi), torch.arange(x_lo, x_hi), indexing="ij")
            pixx = xs.reshape(-1).to(torch.float64)
            pixy = ys.reshape(-1).to(torch.float64)
            dev = None
            if dev_state is not None:
                dev = (dev_state[0][ids], dev_state[1][ids], dev_state[2][ids])
            o, Tf, nc, fr, contrib = blend_tile(g.xy[ids], g.conic[ids], opac[ids], g.depth[ids],
                                       chans_all[ids], pixx, pixy, opt, dev=dev)
            hh, ww = y_hi - y_lo, x_hi - x_lo
            o = o.reshape(hh, ww, C)
            Tf2 = Tf.reshape(hh, ww)
            o_ft, o_dp = o[..., 3:3 + F], o[..., 3 + F:]
            if opt.feats_bg:                        # lineage switch: features over a background value
                o_ft = o_ft + Tf2[..., None] * opt.feat_bg_value
            if opt.depth_normalised:                # lineage switch: depth / accumulated alpha
                A = (1.0 - Tf2)[..., None]
                o_dp = torch.where(A > 1e-10, o_dp / A.clamp_min(1e-10), torch.zeros_like(o_dp))
            o = torch.cat([o[..., :3] + Tf2[..., None] * bg, o_ft, o_dp], dim=-1)
            out[y_lo:y_hi, x_lo:x_hi] = o          # CopySlices: differentiable, O(tile)
            final_T[y_lo:y_hi, x_lo:x_hi] = Tf2.detach()
            n_contrib[y_lo:y_hi, x_lo:x_hi] = nc.reshape(hh, ww)
            # depth-order fragility: near-equal float32 keys of neighbours in the list
            frag_stats["gate"] += int(fr.sum())
            k = key[ids]
            if k.numel() > 1:
                near = (k[1:] - k[:-1]).abs() < 4e-7 * k[1:].abs().clamp_min(1e-3)
                if sort_depth is None and bool(near.any()):
                    # swapping two neighbours only changes pixels that blend both of them
                    fr0 = fr
                    for j in torch.nonzero(near).reshape(-1).tolist():
                        fr = fr | (contrib[j] & contrib[j + 1])
                    frag_stats["order"] += int((fr & ~fr0).sum())
            fragile[y_lo:y_hi, x_lo:x_hi] = fr.reshape(hh, ww)
            # a Gaussian whose membership of this tile is itself ambiguous (unresolved borderline rect / radius /
            # near cull) makes the whole tile fragile -- with a device view these are adopted instead (see above)
            if bool(g.frag_gauss[ids].any()):
                frag_stats["gauss"] += int((~fr).sum())
                fragile[y_lo:y_hi, x_lo:x_hi] = True
    canvas = out
    img = canvas[..., :3].permute(2, 0, 1)
    feats = canvas[..., 3:3 + F].permute(2, 0, 1)
    depth = canvas[..., 3 + F:].permute(2, 0, 1)
    return OracleOut(image=img, radii=g.radii, feats=feats, depth=depth, fragile=fragile,
                     frag_gauss=g.frag_gauss, num_rendered=int(g.tiles_touched.sum()), geom=g,
                     final_T=final_T, n_contrib=n_contrib, pairs_done=pairs_done,
                     tile_mask=tile_mask if (tile_set is not None or tile_step > 1) else torch.ones(H, W, dtype=torch.bool),
                     frag_stats=frag_stats)


def _apply_radii_override(g: Geom, radii_override, W, H) -> Geom:
    """For Gaussians whose ceil() is borderline adopt the device's radius, provided it is one of
    the two admissible neighbours; their fragility is then resolved."""
    ro = radii_override.to(torch.float64).cpu()
    own = torch.ceil(g.r_real)
    admissible = ((ro - own).abs() <= 1.0) & (ro > 0)
    take = g.frag_radius & admissible & g.valid
    radius = torch.where(take, ro, g.radii.to(torch.float64))
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    pxd, pyd = g.xy[:, 0].detach(), g.xy[:, 1].detach()
    x0 = torch.trunc((pxd - radius) / TILE).clamp(0, gx).to(torch.int64)
    x1 = torch.trunc((pxd + radius + (TILE - 1)) / TILE).clamp(0, gx).to(torch.int64)
    y0 = torch.trunc((pyd - radius) / TILE).clamp(0, gy).to(torch.int64)
    y1 = torch.trunc((pyd + radius + (TILE - 1)) / TILE).clamp(0, gy).to(torch.int64)
    area = (x1 - x0) * (y1 - y0)
    rect = torch.stack([x0, y0, x1, y1], -1)
    rect = torch.where(take[:, None], rect, g.rect)
    g.rect = rect
    g.radii = torch.where(take, radius, g.radii.to(torch.float64)).to(torch.int32)
    g.tiles_touched = torch.where(take, area, g.tiles_touched)
    g.frag_gauss = g.frag_rect | (g.frag_radius & ~take)
    return g


def _rect_f32(px, py, radius, gx, gy):
    """The device's tile rect (trase_amd/csrc/gs_math.h tile_rect): float32 quotients, C truncation."""
    import numpy as np
    px, py, r = (np.asarray(v, dtype=np.float32) for v in (px, py, radius))
    t = np.float32(TILE)
    def edge(v, g):
        return np.clip(np.trunc((v / t).astype(np.float32)).astype(np.int64), 0, g)
    x0, y0 = edge(px - r, gx), edge(py - r, gy)
    x1, y1 = edge(px + r + np.float32(TILE - 1), gx), edge(py + r + np.float32(TILE - 1), gy)
    return torch.from_numpy(np.stack([x0, y0, x1, y1], -1))


def _apply_device_view(g: Geom, dv: dict, W: int, H: int, opt: OracleOptions) -> Geom:
    """Adopt the device's discrete preprocess decisions for the Gaussians this oracle marks borderline,
    provided each adopted value is an admissible neighbour of the oracle's own (radius +-1, every rect
    edge +-1, near cull only when |z - near| is inside the float32 error band).  The device's centre and
    depth must agree with the oracle's to float32 accuracy for the adoption to be admissible."""
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    dr = dv["radii"].detach().cpu().to(torch.float64)
    dxy = dv["xy"].detach().cpu().to(torch.float32)
    ddepth = dv["depth"].detach().cpu().to(torch.float32)
    dvis = dr > 0
    cand = g.frag_gauss.clone()
    if not bool(cand.any()):
        return g
    own_r = torch.ceil(g.r_real)
    xy = g.xy.detach()
    # (a) device keeps it: radius, centre and depth must be float32-close to ours
    close = ((dxy.to(torch.float64) - xy).abs().amax(dim=1) < 2e-6 * max(W, H) + 1e-4) & \
            ((ddepth.to(torch.float64) - g.depth.detach()).abs() < 1e-5 * g.depth.detach().abs().clamp_min(1.0))
    rad_ok = ((dr - own_r).abs() <= 1.0)
    drect = _rect_f32(dxy[:, 0].numpy(), dxy[:, 1].numpy(), dr.numpy(), gx, gy)
    own_rect_r = []
    pxd, pyd = xy[:, 0], xy[:, 1]
    for lo_hi in (pxd - dr, pyd - dr, pxd + dr + (TILE - 1), pyd + dr + (TILE - 1)):
        own_rect_r.append(torch.trunc(lo_hi / TILE))
    own_rect = torch.stack([own_rect_r[0].clamp(0, gx), own_rect_r[1].clamp(0, gy),
                            own_rect_r[2].clamp(0, gx), own_rect_r[3].clamp(0, gy)], -1).to(torch.int64)
    rect_ok = ((drect - own_rect).abs() <= 1).all(dim=1)
    area = (drect[:, 2] - drect[:, 0]) * (drect[:, 3] - drect[:, 1])
    take_vis = cand & dvis & close & rad_ok & rect_ok & (area > 0) & (g.depth.detach() > opt.near_cull_z)
    # (b) device culls it: admissible when our own decision is borderline for a reason that can cull
    #     (near plane, or a rect that may be empty with the neighbouring radius / edge)
    own_area_min = torch.ones_like(area)
    for k in range(2):
        lo = torch.minimum(g.rect[:, k], g.rect[:, k] + 1)
        hi = torch.maximum(g.rect[:, k + 2] - 1, lo)
        own_area_min = own_area_min * (hi - lo).clamp_min(0)
    near_band = (g.depth.detach() - opt.near_cull_z).abs() < 1e-6
    take_cull = cand & ~dvis & (near_band | (own_area_min == 0))
    radius = torch.where(take_vis, dr, g.radii.to(torch.float64))
    g.rect = torch.where(take_vis[:, None], drect, g.rect)
    g.rect = torch.where(take_cull[:, None], torch.zeros_like(g.rect), g.rect)
    g.valid = (g.valid | take_vis) & ~take_cull
    g.radii = torch.where(g.valid, radius, torch.zeros_like(radius)).to(torch.int32)
    new_area = (g.rect[:, 2] - g.rect[:, 0]) * (g.rect[:, 3] - g.rect[:, 1])
    g.tiles_touched = torch.where(g.valid, new_area, torch.zeros_like(new_area))
    g.frag_gauss = cand & ~(take_vis | take_cull)
    return g
