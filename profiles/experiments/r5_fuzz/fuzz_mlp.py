"""Size / dead-row sweep of the deformation MLP's training pair: sizes around every tile and grid boundary, both row orders, dead
masks from none to all, default and is_blender networks -- outputs bit-identical across row orders, gradients equal to the index-order
all-rows computation within fp32 reassociation.  Prints each case first (a GPU fault names its case)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trase_amd import deform as D
from trase_amd.deform import deform_forward
from trase_amd.synthetic import SynthDeformNetwork
dev = torch.device("cuda", 0)
sizes = [1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1119, 1120, 1121, 1152, 1153, 1184, 4063, 4064, 4095, 4096, 4097, 4127, 4128,
         4129, 5000, 8191, 8192, 8193, 20011, 65536, 65537]
torch.manual_seed(0)
bad = 0
for blender in (False, True):
    net = (SynthDeformNetwork(is_blender=True) if blender else SynthDeformNetwork()).to(dev)
    live = dict(net.named_parameters())
    for n in sizes:
        x = (torch.rand(n, 3, device=dev) * 2 - 1) * 1.3
        t = torch.tensor([[0.4]], device=dev).expand(n, -1)
        g0 = [torch.randn(n, c, device=dev) for c in (3, 4, 3)]
        masks = {"none": torch.ones(n, device=dev), "slab": (x[:, 1].abs() <= 0.6).float(), "one": torch.zeros(n, device=dev), "all": torch.zeros(n, device=dev)}
        masks["one"][n // 2] = 1.0
        for mname, m in masks.items():
            g = [v * m[:, None] for v in g0]
            res = {}
            for mode in ("none", "morton"):
                print(f"blender={blender} n={n} mask={mname} order={mode}", flush=True)
                D.set_row_order(mode)
                net.zero_grad(set_to_none=True)
                outs = deform_forward(live, x, t, is_blender=blender) if blender else deform_forward(live, x, t)
                torch.autograd.backward(outs, g)
                torch.cuda.synchronize()
                res[mode] = ([o.detach().clone() for o in outs], {k: (p.grad.clone() if p.grad is not None else None) for k, p in live.items()})
            for a, b in zip(res["none"][0], res["morton"][0]):
                if not torch.equal(a, b):
                    bad += 1; print("   OUTPUT MISMATCH", float((a - b).abs().max()), flush=True)
            for k in res["none"][1]:
                a, b = res["none"][1][k], res["morton"][1][k]
                if a is None or b is None:
                    if not (a is None and b is None):
                        bad += 1; print("   NONE MISMATCH", k, flush=True)
                    continue
                s = float(a.abs().max())
                d = float((a - b).abs().max())
                if not (d <= 2e-5 * max(s, 1e-20) + 1e-12) or not torch.isfinite(b).all():
                    bad += 1; print("   GRAD MISMATCH", k, d, s, flush=True)
                if mname == "all" and float(b.abs().max()) != 0.0:
                    bad += 1; print("   NONZERO gradient for all-zero cotangents", k, flush=True)
D.set_row_order("morton")
print("done; problems:", bad)
