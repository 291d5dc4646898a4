"""Launch sequences on more than one HIP stream of a GPU (round 4, profiles/r4_two_streams.md).

* a view rendered on a side stream is bit-identical to the same view on the default stream;
* views issued round-robin on three streams WITHOUT any synchronisation by the caller are bit-identical to the serial run: the
  library keeps its launch sequences in order across the streams of a device (``trase_amd.rasterizer._stream``).  Without that
  guard this test fails on MI355X: the preprocess kernel writes wrong colours for lanes 48..63 of some waves while a compositing
  kernel of another view shares its CU.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(n, w, h, feat):
    from trase_amd.synthetic import SynthGaussianModel, SynthPipe, make_scene, orbit_camera
    dev = torch.device("cuda")
    pc = SynthGaussianModel(make_scene(n, feat_dim=feat, seed=0, scale_mult=0.27).to(dev))
    cams = [orbit_camera(w, h, angle=2 * math.pi * k / 8, fid=k / 8).to(dev) for k in range(8)]
    g = torch.Generator().manual_seed(7)
    gi = (torch.randn(3, h, w, generator=g) / (w * h)).to(dev)
    gf = (torch.randn(feat, h, w, generator=g) / (w * h)).to(dev)
    return dev, pc, SynthPipe(), cams, gi, gf


def _view(pc, pipe, cam, bg, gi, gf):
    from gaussian_renderer import render
    for p in pc.parameters():
        p.grad = None
    o = render(cam, pc, pipe, bg, 0.0, 0.0, 0.0)
    torch.autograd.backward([o["render"], o["render_gaussian_features"]], [gi, gf])
    out = [o["render"].detach().clone(), o["render_gaussian_features"].detach().clone(), o["depth"].detach().clone(), o["radii"].clone()]
    return out + [p.grad.clone() for p in pc.parameters() if p.grad is not None]


def test_views_on_several_streams_match_the_serial_run():
    from trase_amd import rasterizer as R
    n, w, h, feat = 300_000, 1920, 1080, 32
    dev, pc, pipe, cams, gi, gf = _setup(n, w, h, feat)
    bg = torch.zeros(3, device=dev)
    R.set_sync(True)
    try:
        caps = []
        for c in cams:
            _view(pc, pipe, c, bg, gi, gf)
            caps.append(R.last_status()[2])
        R.set_sync(False, capacity=int(max(caps) * 1.25) + 1024)
        ref = []
        for c in cams:
            ref.append(_view(pc, pipe, c, bg, gi, gf))
            torch.cuda.synchronize()
        # one view on a side stream
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            got = _view(pc, pipe, cams[3], bg, gi, gf)
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(ref[3], got)), "side stream differs from the default stream"
        # round robin over three streams, nothing synchronised by the caller
        streams = [torch.cuda.Stream() for _ in range(3)]
        torch.cuda.synchronize()
        for rnd in range(2):
            got = []
            for i in range(16):
                with torch.cuda.stream(streams[i % 3]):
                    got.append(_view(pc, pipe, cams[i % 8], bg, gi, gf))
            torch.cuda.synchronize()
            bad = [i for i in range(16) if not all(torch.equal(a, b) for a, b in zip(ref[i % 8], got[i]))]
            assert not bad, f"views {bad} issued on alternating streams differ from the serial run"
    finally:
        R.set_sync(True)


def test_unordered_streams_are_bit_identical_since_the_library_has_no_packed_fp32():
    """Round 6: the corruption that made the cross-stream ordering necessary hit compiler-generated packed-FP32 VALU instructions
    (v_pk_*_f32) of a wave beside a kernel feeding transposing LDS reads into MFMAs; the library is built with
    -fno-slp-vectorize since.  With the ordering OFF (rasterizer.set_stream_ordering(False)) views round-robin on two and three
    streams -- gradients through torch.autograd.grad: no shared .grad across streams -- must equal the serial run bit for bit."""
    from gaussian_renderer import render
    from trase_amd import rasterizer as R
    n, w, h, feat = 300_000, 1920, 1080, 32
    dev, pc, pipe, cams, gi, gf = _setup(n, w, h, feat)
    bg = torch.zeros(3, device=dev)
    params = pc.parameters()

    def view(i):
        o = render(cams[i % 8], pc, pipe, bg, 0.0, 0.0, 0.0)
        gr = torch.autograd.grad([o["render"], o["render_gaussian_features"]], params + [o["viewspace_points"]], [gi, gf], allow_unused=True)
        ts = [o["render"], o["render_gaussian_features"], o["depth"], o["radii"]] + [t for t in gr if t is not None]
        return torch.stack([t.contiguous().view(torch.int32).to(torch.int64).sum() for t in ts])
    R.set_sync(True)
    try:
        caps = []
        for i in range(8):
            view(i)
            caps.append(R.last_status()[2])
        R.set_sync(False, capacity=int(max(caps) * 1.25) + 1024)
        ref = [view(i).cpu() for i in range(8)]
        R.set_stream_ordering(False)
        for ns in (2, 3):
            streams = [torch.cuda.Stream() for _ in range(ns)]
            torch.cuda.synchronize()
            got = []
            for i in range(48):
                with torch.cuda.stream(streams[i % ns]):
                    got.append(view(i))
            torch.cuda.synchronize()
            bad = [i for i, d in enumerate(got) if not torch.equal(ref[i % 8], d.cpu())]
            assert not bad, f"{ns} streams, ordering off: views {bad} differ from the serial run"
    finally:
        R.set_stream_ordering(True)
        R.set_sync(True)
