"""Size edge cases of the widened rows against their PyTorch compositions: L1 + SSIM, KNN smoothing, NNFM, Adam (many tensors, odd
sizes), mask statistics, the pair head.  Prints each case first."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as Fn
SEED = 1000 * int(sys.argv[1]) if len(sys.argv) > 1 else 0
dev = torch.device("cuda", 0)
bad = 0
def rel(a, b):
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-12)
def check(name, ok, info=""):
    global bad
    if not ok:
        bad += 1; print("   PROBLEM", name, info, flush=True)

# ---- L1 + SSIM
from trase_amd.losses import l1_ssim, photometric_loss, loss_nnfm_style
def ref_ssim(x, y):
    g = torch.tensor([math.exp(-(i - 5) ** 2 / (2 * 1.5 ** 2)) for i in range(11)])
    g = (g / g.sum()).unsqueeze(1)
    win = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0).expand(x.shape[0], 1, 11, 11).contiguous().to(x.device)
    conv = lambda t: Fn.conv2d(t, win, padding=5, groups=x.shape[0])
    mu1, mu2 = conv(x), conv(y)
    s1, s2, s12 = conv(x * x) - mu1 * mu1, conv(y * y) - mu2 * mu2, conv(x * y) - mu1 * mu2
    return (((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))).mean()
for shape in [(3, 1, 1), (3, 2, 3), (3, 5, 7), (1, 11, 11), (3, 31, 33), (3, 32, 32), (3, 33, 31), (4, 64, 65), (2, 200, 1), (3, 1, 300), (3, 270, 480)]:
    print("loss", shape, flush=True)
    torch.manual_seed(sum(shape) + SEED)
    x = torch.rand(*shape, device=dev); y = (x + 0.1 * torch.randn_like(x)).clamp(0, 1)
    xa = x.clone().requires_grad_(True)
    (0.8 * (xa - y).abs().mean() + 0.2 * (1 - ref_ssim(xa, y))).backward()
    xb = x.clone().requires_grad_(True)
    l = photometric_loss(xb, y, 0.2); l.backward()
    want = 0.8 * (x - y).abs().mean() + 0.2 * (1 - ref_ssim(x, y))
    check("loss value", abs(float(l) - float(want)) < 2e-5, (float(l), float(want)))
    check("loss grad", rel(xb.grad, xa.grad) < 5e-5, rel(xb.grad, xa.grad))

# ---- smoothing
from trase_amd.smooth import smooth_features
from pytorch3d.ops import knn_points
for n, K, S in [(17, 16, 8), (20, 4, 4), (100, 16, 1), (1000, 16, 16), (5000, 8, 3), (33, 2, 1)]:
    print("smooth", n, K, S, flush=True)
    g = torch.Generator().manual_seed(n + SEED)
    xyz = torch.rand(n, 3, generator=g).to(dev)
    feats = torch.randn(n, 1, 32, generator=g).to(dev)
    idx = knn_points(xyz.unsqueeze(0), xyz.unsqueeze(0), K=K).idx.squeeze(0)
    sel = torch.randperm(K, generator=g)[:S]
    fa = feats.clone().requires_grad_(True)
    ra = Fn.normalize(fa, dim=-1, p=2)[idx[:, sel.to(dev)], 0, :].mean(dim=1).unsqueeze(1)
    go = torch.randn(n, 1, 32, generator=g).to(dev)
    ra.backward(go)
    fb = feats.clone().requires_grad_(True)
    rb = smooth_features(fb, idx, sel)
    rb.backward(go)
    check("smooth fwd", rel(rb, ra) < 1e-5, rel(rb, ra)); check("smooth bwd", rel(fb.grad, fa.grad) < 1e-4, rel(fb.grad, fa.grad))

# ---- NNFM
for C_, n1, n2 in [(128, 1, 1), (64, 7, 5), (256, 100, 33), (512, 1000, 777), (64, 50, 2000), (192, 1, 300)]:
    print("nnfm", C_, n1, n2, flush=True)
    g = torch.Generator().manual_seed(n1 + n2 + SEED)
    a = torch.randn(C_, n1, generator=g).to(dev); b = torch.randn(C_, n2, generator=g).to(dev)
    aa = a.clone().requires_grad_(True)
    an = aa / (aa.norm(dim=0, keepdim=True) + 1e-8) if False else aa
    # utils/loss_utils.py:223-228 restated
    def ref(f1, f2):
        n_1 = f1 / f1.norm(dim=0, keepdim=True).clamp_min(1e-12) if False else Fn.normalize(f1, dim=0)
        n_2 = Fn.normalize(f2, dim=0)
        d = 1.0 - n_1.t() @ n_2
        return d.min(dim=1).values.mean()
    lr = ref(aa, b); lr.backward()
    ab = a.clone().requires_grad_(True)
    lb = loss_nnfm_style(ab, b); lb.backward()
    check("nnfm value", abs(float(lb) - float(lr)) < 2e-5 * max(1.0, abs(float(lr))), (float(lb), float(lr)))
    check("nnfm grad", rel(ab.grad, aa.grad) < 2e-3, rel(ab.grad, aa.grad))

# ---- Adam: many tensors, odd sizes
from trase_amd.optim import FusedAdam
print("adam", flush=True)
torch.manual_seed(SEED)
shapes = [(1,), (2,), (3,), (5, 1), (7, 3), (31,), (32,), (33,), (255,), (257, 3), (1000, 15, 3)] + [(k + 1, 2) for k in range(40)]
a = [torch.randn(*s, device=dev).requires_grad_(True) for s in shapes]
b = [t.detach().clone().requires_grad_(True) for t in a]
ref = torch.optim.Adam([{"params": [p], "lr": 1e-3 * (1 + i % 5)} for i, p in enumerate(a)], eps=1e-15)
opt = FusedAdam([{"params": [p], "lr": 1e-3 * (1 + i % 5)} for i, p in enumerate(b)], eps=1e-15)
for it in range(4):
    for pa, pb in zip(a, b):
        gr = torch.randn_like(pa); pa.grad = gr.clone(); pb.grad = gr.clone()
    ref.step(); opt.step()
check("adam", all(rel(pb, pa) < 5e-6 for pa, pb in zip(a, b)), max(rel(pb, pa) for pa, pb in zip(a, b)))

# ---- mask statistics + pair head on odd shapes
from trase_amd.feature_head import mask_stats, contrastive_head
for N, H, W, rate in [(1, 8, 8, 0.5), (3, 17, 29, 0.3), (64, 33, 65, 0.1), (256, 40, 50, 0.05), (100, 270, 480, 0.002)]:
    print("head", N, H, W, flush=True)
    g = torch.Generator().manual_seed(N + H + SEED)
    sam = (torch.rand(N, H, W, generator=g) < 0.3).to(dev)
    cover, size = mask_stats(sam)
    check("cover", torch.equal(cover.to(torch.int64), sam.sum(0).to(torch.int64)))
    check("size", torch.equal(size.to(torch.int64).reshape(-1), sam.reshape(N, -1).sum(1).to(torch.int64)))
    sp = torch.logical_and(torch.rand(H, W, generator=g).to(dev) < rate, cover != 0)
    sm = torch.ones(N, dtype=torch.bool, device=dev)
    f = torch.randn(32, H, W, generator=g).to(dev).requires_grad_(True)
    for mode in ("soft", "all", "hard"):
        for use_w in (False, True):
            out = contrastive_head(f, sam, sp, sm, mode, 0.75, 0.5, mask_size=size, use_weights=use_w, with_norm_reg=True)
            tot = out[0] + out[1] + out[4]
            f.grad = None
            tot.backward()
            check("head finite", bool(torch.isfinite(tot)) and bool(torch.isfinite(f.grad).all()), (mode, use_w, float(tot)))
torch.cuda.synchronize()
print("done; problems:", bad)
