"""Prototype: two views per step, phase-locked -- the list-building chains of the two views run side by side on two streams, the
compositing kernels (transposing LDS reads) run alone, the per-Gaussian tails side by side again.  Bitwise check against the serial
product path + throughput.  python scratch/phase_locked.py [steps]"""
import sys, os, math, time, ctypes as C, torch
sys.path.insert(0, os.getcwd())
from trase_amd.synthetic import make_scene, orbit_camera, SynthGaussianModel, SynthPipe
from trase_amd import rasterizer as R, _lib
from trase_amd.rasterizer import GaussianRasterizationSettings, _fill_settings, _sizes, _bytes
from gaussian_renderer import render
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
MODE = os.environ.get("PL_MODE", "locked")
N, W, H, F = 300_000, 1920, 1080, 32
dev = torch.device("cuda", 0)
pc = SynthGaussianModel(make_scene(N, feat_dim=F, seed=0, scale_mult=0.27).to(dev))
pipe = SynthPipe(); params = pc.parameters()
cams = [orbit_camera(W, H, angle=2 * math.pi * k / 16, fid=k / 16).to(dev) for k in range(16)]
bg = torch.zeros(3, device=dev)
g = torch.Generator().manual_seed(1234)
gi = (torch.randn(3, H, W, generator=g) / (W * H)).to(dev); gf = (torch.randn(F, H, W, generator=g) / (W * H)).to(dev)
lib = _lib.load()
lib.trase_rast_render_raw_phase.restype = C.c_int
lib.trase_rast_render_raw_phase.argtypes = [C.POINTER(_lib.RastSettings), C.POINTER(_lib.RastRawInputs), C.POINTER(_lib.RastOutputs),
                                            C.POINTER(_lib.RastWorkspace), C.c_int32, C.c_void_p]

def serial(i):
    for p in params: p.grad = None
    o = render(cams[i % 16], pc, pipe, bg, 0.0, 0.0, 0.0)
    torch.autograd.backward([o["render"], o["render_gaussian_features"]], [gi, gf])
    return [o["render"].detach().clone(), o["render_gaussian_features"].detach().clone(), o["radii"].clone()] + [p.grad.clone() for p in params if p.grad is not None]
R.set_sync(True)
caps = []
for i in range(16):
    serial(i); caps.append(R.last_status()[2])
CAP = int(max(caps) * 1.25) + 1024
R.set_sync(False, capacity=CAP)
refs = [serial(i) for i in range(16)]
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps): serial(i)
torch.cuda.synchronize()
print("serial product path: %.1f views/s" % (steps / (time.perf_counter() - t0)), flush=True)

def st(s): return C.c_void_p(s.cuda_stream)

class View:
    def __init__(self, cam):
        rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), bg=bg,
                                           scale_modifier=1.0, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
                                           sh_degree=pc.active_sh_degree, campos=cam.camera_center, prefiltered=False, debug=False)
        self.keep = []
        self.s = _fill_settings(rs, dev, self.keep)
        raw = self.raw = _lib.RastRawInputs()
        raw.P, raw.F, raw.norm_features = N, F, 1
        raw.xyz, raw.d_xyz = _lib.ptr(pc._xyz), None
        raw.features_dc, raw.features_rest, raw.opacity = _lib.ptr(pc._features_dc), _lib.ptr(pc._features_rest), _lib.ptr(pc._opacity)
        raw.scaling, raw.d_scaling, raw.rotation, raw.d_rotation = _lib.ptr(pc._scaling), None, _lib.ptr(pc._rotation), None
        raw.gaussian_features = _lib.ptr(pc._gaussian_features)
        self.featn = torch.empty(N, F, device=dev); raw.featn = _lib.ptr(self.featn)
        self.buf = torch.empty(3 + F + 1, H, W, device=dev)
        self.image, self.feats, self.depth = self.buf[:3], self.buf[3:3 + F], self.buf[3 + F:]
        self.radii = torch.empty(N, dtype=torch.int32, device=dev)
        out = self.out = _lib.RastOutputs()
        out.image, out.radii, out.depth, out.feats = _lib.ptr(self.image), _lib.ptr(self.radii), _lib.ptr(self.depth), _lib.ptr(self.feats)
        gb, bb, ib, pb, tb, btb = _sizes(lib, N, W, H, F, CAP)
        self.ws_t = [_bytes(x, dev) for x in (gb, bb, ib, pb, max(tb, btb))]
        ws = self.ws = _lib.RastWorkspace()
        ws.geom, ws.geom_bytes = _lib.ptr(self.ws_t[0]), self.ws_t[0].numel()
        ws.bin, ws.bin_bytes = _lib.ptr(self.ws_t[1]), self.ws_t[1].numel()
        ws.img, ws.img_bytes = _lib.ptr(self.ws_t[2]), self.ws_t[2].numel()
        ws.pre, ws.pre_bytes = _lib.ptr(self.ws_t[3]), self.ws_t[3].numel()
        ws.tmp, ws.tmp_bytes = _lib.ptr(self.ws_t[4]), self.ws_t[4].numel()
        ws.capacity = CAP
        self.grads = [torch.empty_like(p) for p in (pc._xyz, pc._features_dc, pc._features_rest, pc._opacity, pc._scaling, pc._rotation, pc._gaussian_features)]
        self.m2d = torch.empty(N, 3, device=dev)
        gr = self.gr = _lib.RastRawGrads()
        gr.dL_dimage, gr.dL_dfeats = _lib.ptr(gi), _lib.ptr(gf)
        gr.dL_dxyz, gr.dL_dmeans2D = _lib.ptr(self.grads[0]), _lib.ptr(self.m2d)
        gr.dL_dfeatures_dc, gr.dL_dfeatures_rest, gr.dL_dopacity = _lib.ptr(self.grads[1]), _lib.ptr(self.grads[2]), _lib.ptr(self.grads[3])
        gr.dL_dscaling, gr.dL_drotation, gr.dL_dgaussian_features = _lib.ptr(self.grads[4]), _lib.ptr(self.grads[5]), _lib.ptr(self.grads[6])
    def front(self, s):
        _lib.check(lib.trase_rast_preprocess_raw(C.byref(self.s), C.byref(self.raw), C.byref(self.out), C.byref(self.ws), st(s)), "pre")
        _lib.check(lib.trase_rast_render_raw_phase(C.byref(self.s), C.byref(self.raw), C.byref(self.out), C.byref(self.ws), 1, st(s)), "bin")
    def compose(self, s):
        _lib.check(lib.trase_rast_render_raw_phase(C.byref(self.s), C.byref(self.raw), C.byref(self.out), C.byref(self.ws), 2, st(s)), "fwd")
    def bwd_compose(self, s):
        _lib.check(lib.trase_rast_backward_raw_compose(C.byref(self.s), C.byref(self.raw), C.byref(self.out), C.byref(self.ws), C.byref(self.gr), st(s)), "bwdc")
    def tail(self, s):
        _lib.check(lib.trase_rast_backward_raw_gaussians(C.byref(self.s), C.byref(self.raw), C.byref(self.out), C.byref(self.ws), C.byref(self.gr), 0, N, st(s)), "tail")
    def result(self):
        return [self.image.clone(), self.feats.clone(), self.radii.clone()] + [x.clone() for x in (self.grads[0], self.grads[1], self.grads[2], self.grads[3], self.grads[4], self.grads[5], self.grads[6])]

V = 2
views = [[View(cams[i]) for i in range(16)] for _ in range(1)][0]
side = [torch.cuda.Stream() for _ in range(V)]
main = torch.cuda.current_stream()
def batch(ids, mode):
    vs = [views[i % 16] for i in ids]
    if mode == "serial":
        for v in vs:
            v.front(main); v.compose(main); v.bwd_compose(main); v.tail(main)
        return
    for k, v in enumerate(vs):
        side[k].wait_stream(main)
        v.front(side[k])
    for k in range(len(vs)): main.wait_stream(side[k])
    for v in vs: v.compose(main)
    for v in vs: v.bwd_compose(main)
    for k, v in enumerate(vs):
        side[k].wait_stream(main)
        v.tail(side[k])
    for k in range(len(vs)): main.wait_stream(side[k])

def order(res):   # grads of the harness in the product's parameter order
    return res
for mode in ("serial", "locked"):
    for i in range(0, 8, V): batch(list(range(i, i + V)), mode)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(0, steps, V): batch(list(range(i, i + V)), mode)
    torch.cuda.synchronize()
    print("stage calls, %s: %.1f views/s" % (mode, steps / (time.perf_counter() - t0)), flush=True)
# bitwise check of the locked mode against the product's serial path
names = ["image", "feats", "radii"] + [f"grad{k}" for k in range(7)]
bad = 0
for rnd in range(3):
    for i in range(0, 16, V):
        batch([i, i + 1], "locked")
        torch.cuda.synchronize()
        for k in (i, i + 1):
            res = views[k].result()
            ref = refs[k]
            # product grads order = params order; compare as multisets by shape
            for n_, a in zip(names[:3], res[:3]):
                if not torch.equal(a, ref[names.index(n_)]): bad += 1; print("view", k, n_, "differs")
            for a in res[3:]:
                if not any(a.shape == b.shape and torch.equal(a, b) for b in ref[3:]): bad += 1; print("view", k, "a gradient", tuple(a.shape), "differs")
print("phase-locked vs serial product path: mismatching tensors", bad)
