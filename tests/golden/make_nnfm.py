"""Generates nnfm.npz from the imported reference's loss_nnfm_style (utils/loss_utils.py:223-228): the loss value and
its gradient w.r.t. the rendered-frame features for two seeded cases (C = 64 and C = 512 channels).  Data only.

    python tests/golden/make_nnfm.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference  # noqa: E402


def main():
    import_reference()
    from utils.loss_utils import loss_nnfm_style
    out = {}
    for name, (c, n1, n2, seed) in {"small": (64, 300, 211, 0), "vgg": (512, 384, 256, 1)}.items():
        g = torch.Generator().manual_seed(seed)
        # post-ReLU-like features (non-negative, sparse-ish) as VGG conv outputs are
        f1 = torch.relu(torch.randn(c, n1, generator=g) + 0.3).requires_grad_(True)
        f2 = torch.relu(torch.randn(c, n2, generator=g) + 0.3)
        loss = loss_nnfm_style(f1, f2)
        loss.backward()
        # the matched neighbour and the margin to the runner-up (for tie-aware comparisons)
        with torch.no_grad():
            cosm = (f1 / torch.linalg.norm(f1, dim=0)).T @ (f2 / torch.linalg.norm(f2, dim=0))
            top2 = cosm.topk(2, dim=1)
        out.update({f"{name}_f1": f1.detach().numpy(), f"{name}_f2": f2.numpy(), f"{name}_loss": loss.item(),
                    f"{name}_grad": f1.grad.numpy(), f"{name}_argmin": top2.indices[:, 0].numpy(),
                    f"{name}_margin": (top2.values[:, 0] - top2.values[:, 1]).numpy()})
        print(name, "loss", loss.item(), "min margin", float((top2.values[:, 0] - top2.values[:, 1]).min()))
    np.savez_compressed(os.path.join(HERE, "nnfm.npz"), **out)
    print("wrote nnfm.npz", os.path.getsize(os.path.join(HERE, "nnfm.npz")))


if __name__ == "__main__":
    main()
