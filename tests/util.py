"""Shared helpers of the test-suite (test infrastructure, may import oracle/)."""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess
import tempfile

import numpy as np
import torch

from trase_amd.rasterizer import GaussianRasterizationSettings
from trase_amd.synthetic import make_scene, orbit_camera

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def settings_for(cam, sh_degree=3, bg=(0.0, 0.0, 0.0), scale_modifier=1.0, device="cpu", debug=False):
    return GaussianRasterizationSettings(
        image_height=int(cam.image_height), image_width=int(cam.image_width),
        tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
        bg=torch.tensor(bg, dtype=torch.float32, device=device), scale_modifier=scale_modifier,
        viewmatrix=cam.world_view_transform.to(device), projmatrix=cam.full_proj_transform.to(device),
        sh_degree=sh_degree, campos=cam.camera_center.to(device), prefiltered=False, debug=debug)


def small_case(n=400, w=96, h=64, feat=32, seed=0, scale_mult=0.9, angle=0.4, d_rot=0.0, opacity_mode="trained",
               layout="cube", world_scale=1.0, radius=4.0, elevation=0.15, focal_mult=1.2):
    """Activated inputs of one rasterizer call (CPU tensors).  radius is given in units of the UNSCALED scene (the
    camera moves out with world_scale, so the image stays the same and only the view depth grows)."""
    scene = make_scene(n, feat_dim=max(feat, 1), seed=seed, scale_mult=scale_mult, opacity_mode=opacity_mode,
                       layout=layout, world_scale=world_scale)
    cam = orbit_camera(w, h, angle=angle, radius=radius * world_scale, elevation=elevation, focal_mult=focal_mult)
    act = scene.activated()
    if d_rot:
        g = torch.Generator().manual_seed(seed + 7)
        act["rotations"] = act["rotations"] + d_rot * torch.randn(n, 4, generator=g)   # non-unit quaternions
    if feat == 0:
        act["sh_objs"] = None
    return act, cam


class HsView(C.Structure):
    _fields_ = [("V", C.c_float * 16), ("PM", C.c_float * 16), ("cam", C.c_float * 3),
                ("tanx", C.c_float), ("tany", C.c_float), ("mod", C.c_float),
                ("W", C.c_int), ("H", C.c_int), ("deg", C.c_int)]


def build_hostsim():
    out = os.path.join(tempfile.gettempdir(), f"libtrase_hostsim_{os.getpid()}.so")
    src = os.path.join(ROOT, "tests", "hostsim", "hostsim.cpp")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", out, src])
    lib = C.CDLL(out)
    lib.hs_forward.restype = None
    lib.hs_backward.restype = None
    return lib


def hs_view(settings):
    v = HsView()
    vm = settings.viewmatrix.reshape(-1).tolist()
    pm = settings.projmatrix.reshape(-1).tolist()
    for i in range(16):
        v.V[i] = vm[i]
        v.PM[i] = pm[i]
    for i in range(3):
        v.cam[i] = float(settings.campos[i])
    v.tanx, v.tany, v.mod = settings.tanfovx, settings.tanfovy, settings.scale_modifier
    v.W, v.H, v.deg = settings.image_width, settings.image_height, settings.sh_degree
    return v


def fptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def np32(t):
    return None if t is None else np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float32))
