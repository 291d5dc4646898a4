"""How many rows of loss_nnfm_style follow a neighbour other than the float64 arg-min, and by what margin."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trase_amd.losses import loss_nnfm_style
dev = torch.device("cuda", 0)
G = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "nnfm.npz"))
cases = [(n, torch.from_numpy(G[f"{n}_f1"]), torch.from_numpy(G[f"{n}_f2"])) for n in ("small", "vgg")]
for C, n1, n2, relu in [(512, 1000, 777, False), (512, 4000, 3000, True), (64, 2000, 5000, False), (256, 3000, 31, True), (128, 500, 1, False), (512, 8000, 8000, True)]:
    g = torch.Generator().manual_seed(C + n1 + n2)
    a, b = torch.randn(C, n1, generator=g), torch.randn(C, n2, generator=g)
    if relu:
        a, b = torch.relu(a + 0.3), torch.relu(b + 0.3)
    cases.append((f"rand C={C} {n1}x{n2} relu={relu}", a, b))
for name, a, b in cases:
    a = a.to(dev); b = b.to(dev)
    ad, bd = a.double(), b.double()
    an, bn = ad / ad.norm(dim=0, keepdim=True), bd / bd.norm(dim=0, keepdim=True)
    cos = an.t() @ bn                                    # (n1, n2) float64
    top = cos.topk(min(2, cos.shape[1]), dim=1)
    margin = (top.values[:, 0] - top.values[:, -1]) if cos.shape[1] > 1 else torch.ones(cos.shape[0], device=dev, dtype=torch.float64)
    x = a.clone().requires_grad_(True)
    loss = loss_nnfm_style(x, b); loss.backward()
    # which neighbour did the gradient follow?  d/df1 = -(b_j/(|a||b_j|) - cos a/|a|^2)/n1: recover cos of the followed neighbour
    xr = ad.clone().requires_grad_(True)
    lr = (1.0 - ((xr / xr.norm(dim=0, keepdim=True)).t() @ bn).max(dim=1).values).mean(); lr.backward()
    scale = float(xr.grad.abs().max())
    rowdiff = (x.grad.double() - xr.grad).abs().amax(dim=0) / scale
    wrong = rowdiff > 1e-4
    print(f"{name}: rows {a.shape[1]}, loss diff {abs(float(loss) - float(lr)):.2e}, rows following another neighbour: {int(wrong.sum())}"
          f" (largest float64 margin among them {float(margin[wrong].max()) if wrong.any() else 0.0:.2e});"
          f" rows with margin < 1e-3: {int((margin < 1e-3).sum())}, < 1e-6: {int((margin < 1e-6).sum())}", flush=True)
