"""Seeded synthetic scenes + cameras shaped like the reference's inputs.

No datasets exist on the GPU box, so bench/tests render synthetic Gaussians with
the value distributions SURVEY.md 8(d) prescribes.  The camera record carries
exactly the fields ``gaussian_renderer.render()`` reads from a reference
``Camera`` (scene/cameras.py:70-79): transposed (row-vector) matrices,
``FoVx/FoVy``, ``image_width/height``, ``camera_center``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

SH_C0 = 0.28209479177387814


def rgb2sh(rgb):
    """utils/sh_utils.py:114-115"""
    return (rgb - 0.5) / SH_C0


def world2view(Rc2w: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """Restates getWorld2View2 with zero translate / unit scale
    (utils/graphics_utils.py:45-57): R is stored camera-to-world, t world-to-camera."""
    Rt = torch.zeros(4, 4, dtype=torch.float64)
    Rt[:3, :3] = Rc2w.T
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    return Rt.to(torch.float32)


def projection_matrix(znear, zfar, fovx, fovy) -> torch.Tensor:
    """Restates getProjectionMatrix (utils/graphics_utils.py:59-77)."""
    ty, tx = math.tan(fovy / 2), math.tan(fovx / 2)
    top, right = ty * znear, tx * znear
    bottom, left = -top, -right
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


@dataclass
class SynthCamera:
    image_width: int
    image_height: int
    FoVx: float
    FoVy: float
    world_view_transform: torch.Tensor   # (4,4) transposed
    projection_matrix: torch.Tensor      # (4,4) transposed
    full_proj_transform: torch.Tensor    # (4,4)
    camera_center: torch.Tensor          # (3,)
    fid: torch.Tensor                    # (1,) time in [0,1]
    znear: float = 0.01
    zfar: float = 100.0

    def to(self, device):
        return SynthCamera(self.image_width, self.image_height, self.FoVx, self.FoVy,
                           self.world_view_transform.to(device), self.projection_matrix.to(device),
                           self.full_proj_transform.to(device), self.camera_center.to(device),
                           self.fid.to(device), self.znear, self.zfar)


def orbit_camera(width: int, height: int, angle: float = 0.0, radius: float = 4.0,
                 elevation: float = 0.15, focal_mult: float = 1.2, fid: float = 0.0,
                 znear: float = 0.01, zfar: float = 100.0) -> SynthCamera:
    """Camera on an orbit looking at the origin; focal = focal_mult * W.  znear / zfar default to the
    reference's Camera constants (scene/cameras.py:70-71).  radius < scene extent puts the camera INSIDE the cloud
    (the reference's GUIs fly through the scene, gui.py:927-1128): Gaussians behind the image plane, at the near cull
    and with clamped tx/tz then dominate."""
    focal = focal_mult * width
    fovx = 2 * math.atan(width / (2 * focal))
    fovy = 2 * math.atan(height / (2 * focal))
    eye = torch.tensor([radius * math.cos(elevation) * math.sin(angle),
                        radius * math.sin(elevation),
                        radius * math.cos(elevation) * math.cos(angle)], dtype=torch.float64)
    fwd = -eye / eye.norm()                      # camera +z looks at the origin
    up = torch.tensor([0.0, -1.0, 0.0], dtype=torch.float64)   # image y points down
    right = torch.linalg.cross(up, fwd)
    right = right / right.norm()
    down = torch.linalg.cross(fwd, right)
    Rc2w = torch.stack([right, down, fwd], dim=1)          # columns = camera axes in world
    t = -(Rc2w.T @ eye)
    wvt = world2view(Rc2w, t).transpose(0, 1).contiguous()
    proj = projection_matrix(znear, zfar, fovx, fovy).transpose(0, 1).contiguous()
    full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0).contiguous()
    center = wvt.inverse()[3, :3].contiguous()
    return SynthCamera(width, height, fovx, fovy, wvt, proj, full, center,
                       torch.tensor([fid], dtype=torch.float32), znear, zfar)


@dataclass
class SynthScene:
    """Raw (pre-activation) parameters laid out like scene/gaussian_model.py:56-63."""
    xyz: torch.Tensor            # (N,3)
    features_dc: torch.Tensor    # (N,1,3)
    features_rest: torch.Tensor  # (N,15,3)
    scaling: torch.Tensor        # (N,3) log-scale
    rotation: torch.Tensor       # (N,4) raw quaternion
    opacity: torch.Tensor        # (N,1) logit
    gaussian_features: torch.Tensor  # (N,1,F)

    def to(self, device):
        return SynthScene(*[getattr(self, k).to(device) for k in
                            ("xyz", "features_dc", "features_rest", "scaling", "rotation",
                             "opacity", "gaussian_features")])

    # activations exactly as scene/gaussian_model.py:43-51,183-205
    def activated(self, norm_features: bool = True):
        scales = torch.exp(self.scaling)
        rot = torch.nn.functional.normalize(self.rotation)
        opac = torch.sigmoid(self.opacity)
        shs = torch.cat((self.features_dc, self.features_rest), dim=1)
        f = self.gaussian_features
        if norm_features:
            f = f / (f.norm(dim=2, keepdim=True) + 1e-9)   # gaussian_renderer/__init__.py:120-121
        return dict(means3D=self.xyz, scales=scales, rotations=rot, opacities=opac, shs=shs, sh_objs=f)


def make_scene(n: int, feat_dim: int = 32, seed: int = 0, extent: float = 1.3,
               scale_mult: float = 0.3, opacity_mode: str = "trained", layout: str = "cube",
               world_scale: float = 1.0) -> SynthScene:
    """xyz ~ U(-extent, extent)^3 (scene/dataset_readers.py:409), scales from the mean
    point spacing (stand-in for log(sqrt(distCUDA2)), scene/gaussian_model.py:237-238)
    jittered by +-0.5 in log space, normalised N(0,1) quaternions, opacity logits
    N(0,2) ('trained') or inverse_sigmoid(0.1) ('init'), SH dc = RGB2SH(U(0,1)),
    rest ~ N(0,0.05), features = RGB2SH(U(0,1)) (scene/gaussian_model.py:232).

    layout="clusters" (SURVEY 8d names clustered blobs): the same cube draws are pulled towards 2 + seed % 6 blob
    centres (Gaussian blobs of different widths over a sparse uniform background), so that tile lists range from empty
    to very deep inside one image.  world_scale multiplies positions AND sizes (the image is unchanged when the camera
    radius is scaled alike; view depth is not): world_scale 5..20 with a radius-4*world_scale camera gives
    z in [20, 100], the reference's zfar (scene/cameras.py:70)."""
    g = torch.Generator().manual_seed(seed)
    xyz = (torch.rand(n, 3, generator=g) * 2 - 1) * extent
    spacing = ((2 * extent) ** 3 / max(n, 1)) ** (1.0 / 3.0)
    base = math.log(scale_mult * spacing)
    scaling = base + (torch.rand(n, 3, generator=g) - 0.5)
    if layout == "clusters":
        g2 = torch.Generator().manual_seed(seed * 7919 + 13)
        k = 2 + seed % 6
        centres = (torch.rand(k, 3, generator=g2) * 2 - 1) * (0.75 * extent)
        widths = 0.04 + 0.25 * torch.rand(k, generator=g2)
        which = torch.randint(0, k, (n,), generator=g2)
        in_blob = torch.rand(n, generator=g2) < 0.85
        blob = centres[which] + (xyz / extent) * 0.6 * extent * widths[which][:, None] * 3.0
        xyz = torch.where(in_blob[:, None], blob, xyz)
        # denser blobs hold smaller splats (what densification produces)
        scaling = scaling + torch.where(in_blob, torch.log(widths[which] * 3.0).clamp(max=0.0), torch.zeros(n))[:, None]
    elif layout != "cube":
        raise ValueError(f"unknown scene layout {layout!r}")
    if world_scale != 1.0:
        xyz = xyz * world_scale
        scaling = scaling + math.log(world_scale)
    rotation = torch.randn(n, 4, generator=g)
    if opacity_mode == "trained":
        opacity = torch.randn(n, 1, generator=g) * 2.0
    else:
        opacity = torch.full((n, 1), math.log(0.1 / 0.9))
    dc = rgb2sh(torch.rand(n, 1, 3, generator=g))
    rest = torch.randn(n, 15, 3, generator=g) * 0.05
    feats = rgb2sh(torch.rand(n, 1, feat_dim, generator=g))
    return SynthScene(xyz.float(), dc.float(), rest.float(), scaling.float(), rotation.float(),
                      opacity.float(), feats.float())


class SynthGaussianModel:
    """The slice of the reference's GaussianModel that render() reads (scene/gaussian_model.py:56-63 raw
    parameters, :43-51,183-205 activated getters), over a SynthScene."""

    def __init__(self, scene: SynthScene, sh_degree: int = 3, requires_grad: bool = True):
        self.max_sh_degree = 3
        self.active_sh_degree = sh_degree
        mk = lambda t: t.detach().clone().requires_grad_(requires_grad)
        self._xyz = mk(scene.xyz)
        self._features_dc = mk(scene.features_dc)
        self._features_rest = mk(scene.features_rest)
        self._scaling = mk(scene.scaling)
        self._rotation = mk(scene.rotation)
        self._opacity = mk(scene.opacity)
        self._gaussian_features = mk(scene.gaussian_features)

    def parameters(self):
        return [self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation, self._opacity,
                self._gaussian_features]

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_scaling(self):
        return torch.exp(self._scaling)

    @property
    def get_rotation(self):
        return torch.nn.functional.normalize(self._rotation)

    @property
    def get_opacity(self):
        return torch.sigmoid(self._opacity)

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_gaussian_features(self):
        return self._gaussian_features

    def get_covariance(self, scaling_modifier=1):
        """Sigma = (R S)(R S)^T as six floats, R from the NORMALISED raw quaternion (scene/gaussian_model.py:37-41,
        216-217 with utils/general_utils.py:108-156) -- what ``pipe.compute_cov3D_python`` feeds the rasterizer."""
        q = torch.nn.functional.normalize(self._rotation)
        r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                         2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                         2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=-1).reshape(-1, 3, 3)
        L = R * (scaling_modifier * self.get_scaling)[:, None, :]
        S = L @ L.transpose(1, 2)
        return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=-1)


class SynthPipe:
    convert_SHs_python = False
    compute_cov3D_python = False
    debug = False


class SynthDeformNetwork(torch.nn.Module):
    """Random-init network with the layer shapes and parameter names of the reference's ``DeformNetwork``
    (utils/time_utils.py:60-104, default TRASE config: D=8, W=256, multires=10, t_multires=10, skip at layer 4's
    output).  A parameter holder for benches and tests; its eager fp32 ``forward`` is the "what the reference
    runs" comparison leg, never the product path (that is ``trase_amd.deform.deform_forward``)."""

    def __init__(self, is_blender: bool = False, is_6dof: bool = False):
        super().__init__()
        self.is_blender, self.is_6dof = bool(is_blender), bool(is_6dof)
        emb = 84
        if is_blender:                      # utils/time_utils.py:74-86: t_multires = 6, timenet 13 -> 256 -> 30
            emb = 93
            self.timenet = torch.nn.Sequential(torch.nn.Linear(13, 256), torch.nn.ReLU(inplace=True), torch.nn.Linear(256, 30))
        self.linear = torch.nn.ModuleList(
            [torch.nn.Linear(emb, 256)] + [torch.nn.Linear(emb + 256 if i == 4 else 256, 256) for i in range(7)])
        if is_6dof:                         # utils/time_utils.py:100-102
            self.branch_w = torch.nn.Linear(256, 3)
            self.branch_v = torch.nn.Linear(256, 3)
        else:
            self.gaussian_warp = torch.nn.Linear(256, 3)
        self.gaussian_rotation = torch.nn.Linear(256, 4)
        self.gaussian_scaling = torch.nn.Linear(256, 3)

    @staticmethod
    def embed(v, nf):
        out = [v]
        for f in range(nf):
            out += [torch.sin(v * 2.0 ** f), torch.cos(v * 2.0 ** f)]
        return torch.cat(out, -1)

    def time_block(self, t):
        return self.timenet(self.embed(t, 6)) if self.is_blender else self.embed(t, 10)

    def forward(self, x, t):
        e = torch.cat([self.embed(x, 10), self.time_block(t)], -1)
        h = e
        for i, l in enumerate(self.linear):
            h = torch.relu(l(h))
            if i == 4:
                h = torch.cat([e, h], -1)
        if self.is_6dof:                    # utils/time_utils.py:111-118
            from .deform import exp_se3
            w, v = self.branch_w(h), self.branch_v(h)
            theta = torch.norm(w, dim=-1, keepdim=True)
            d_xyz = exp_se3(torch.cat([w / theta + 1e-5, v / theta + 1e-5], dim=-1), theta)
        else:
            d_xyz = self.gaussian_warp(h)
        return d_xyz, self.gaussian_rotation(h), self.gaussian_scaling(h)
