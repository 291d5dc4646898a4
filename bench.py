#!/usr/bin/env python
"""views/sec (fwd+bwd) of the rasterizer path on synthetic S4 input (BASELINE.json metric):
N = 300 000 Gaussians, 1920x1080, 32 feature channels, SH degree 3.

A "step" is one pass of the hot path over one view: activations of the raw parameters
(gaussian_renderer/__init__.py:82-121) -> GaussianRasterizer forward -> backward with fixed
cotangents on the RGB image and the 32-channel feature map (the window the reference times at
train.py:157-303, minus its loss heads).  For --gpus N > 1 every rank renders a different view
(data parallel over camera views) and the step ends with one RCCL all-reduce of the flat
Gaussian-gradient buffer.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from trase_amd import rasterizer as R  # noqa: E402
from trase_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402
from trase_amd.synthetic import make_scene, orbit_camera  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
N_SIMD = 256 * 4               # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9               # peak engine clock
C_RGB = 3


def source_sha16() -> str:
    """Hash of the kernel sources the library is built from (stable across rebuilds, unlike the .so): stamps PMC tables
    so that replayed counters are only reported for the build they were collected on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "trase_amd", "csrc")
    # the sources of the path this bench times (the other kernels of the library -- MLP, losses, KNN, optimizer ... -- do not
    # change what the replayed counters describe)
    path_files = ("api.hip", "binning.hip", "common.h", "gs_math.h", "preprocess.hip", "preprocess_raw.hip", "render.hip",
                  "render_bwd_gs.hip", "render_bwd_hw.hip", "render_fwd_mf.hip")
    for name in sorted(os.listdir(d)):
        if name in path_files:
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def algorithmic_bytes(kernel: str, n: int, r: int, p: int, f: int) -> float:
    """SURVEY.md 8(d) per-kernel algorithmic bytes (general F)."""
    if kernel == "render_bwd":
        return (4 * (C_RGB + f) + 8) * p + (44 + 4 * f) * r + (36 + 4 * f) * n
    if kernel == "render_fwd":
        return (44 + 4 * f) * r + (4 * (C_RGB + f + 1) + 8) * p
    if kernel == "preprocess_fwd":
        return 284 * n
    if kernel == "preprocess_bwd":
        return 648 * n
    return 36 * r   # binning family


def view_bytes(n, r, p, f):
    """A_view = (840+8F) N + (124+8F) R + (44+8F) P  (= 1096 N + 380 R + 300 P at F=32)."""
    return (840 + 8 * f) * n + (124 + 8 * f) * r + (44 + 8 * f) * p


def settings_for(cam, device):
    return GaussianRasterizationSettings(
        image_height=cam.image_height, image_width=cam.image_width,
        tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
        bg=torch.zeros(3, device=device), scale_modifier=1.0,
        viewmatrix=cam.world_view_transform.to(device), projmatrix=cam.full_proj_transform.to(device),
        sh_degree=3, campos=cam.camera_center.to(device), prefiltered=False, debug=False)


def cpu_preprocess_baseline(scene_cpu, cam, budget_s, threads):
    """The CPU baseline BASELINE.json's north_star names: the reference's PyTorch-CPU preprocess (activations +
    deformation add + feature normalisation, SH -> RGB via eval_sh, Sigma = RS(RS)^T, projection by
    full_proj_transform; oracle/cpu_preprocess.py, pinned against the imported reference by
    tests/test_cpu_preprocess.py), fp32, ALL Gaussians of the workload, `threads` host threads, repeated over whole
    views until the time budget is spent -- no extrapolation.  Returns (seconds per view, views timed)."""
    from oracle.cpu_preprocess import reference_cpu_preprocess
    torch.set_num_threads(threads)
    n = scene_cpu.xyz.shape[0]
    z3, z4 = torch.zeros(n, 3), torch.zeros(n, 4)
    args = (scene_cpu.xyz, scene_cpu.features_dc, scene_cpu.features_rest, scene_cpu.opacity, scene_cpu.scaling,
            scene_cpu.rotation, scene_cpu.gaussian_features, z3, z4, z3, cam.full_proj_transform, cam.camera_center,
            cam.image_width, cam.image_height)
    k = 4096                                      # warm-up (thread pool, allocator) on a slice: a whole view can take seconds
    warm = tuple(a[:k] if (torch.is_tensor(a) and a.dim() > 0 and a.shape[0] == n) else a for a in args)
    with torch.no_grad():
        reference_cpu_preprocess(*warm)
        reps, t0 = 0, time.perf_counter()
        while True:
            reference_cpu_preprocess(*args)
            reps += 1
            dt = time.perf_counter() - t0
            if dt > budget_s or reps >= 200:
                break
    return dt / reps, reps


def cpu_baseline(scene_cpu, cam, feat, tile_step, budget_s):
    """Secondary, clearly labelled sample: the float64 oracle (full per-Gaussian preprocess + compositing fwd+bwd of
    every tile_step-th tile until the forward time budget is spent), extrapolated by the (tile,Gaussian) pair count."""
    from oracle import raster_oracle as ro
    st = settings_for(cam, "cpu")
    act = scene_cpu.activated()
    leaves = {k: v.double().clone().requires_grad_(True) for k, v in act.items()}
    t0 = time.perf_counter()
    o = ro.rasterize(st, leaves["means3D"], None, shs=leaves["shs"], sh_objs=leaves["sh_objs"],
                     opacities=leaves["opacities"], scales=leaves["scales"], rotations=leaves["rotations"],
                     tile_step=tile_step, max_seconds=budget_s)
    loss = o.image.sum() + o.feats.sum()
    loss.backward()
    dt = time.perf_counter() - t0
    return dt, o.num_rendered, o.pairs_done


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gaussians", type=int, default=300_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--feat", type=int, default=32)
    ap.add_argument("--scale-mult", type=float, default=0.27)
    ap.add_argument("--variant", type=lambda s: int(s, 0), default=None)
    ap.add_argument("--unfused", action="store_true",
                    help="time the reference's PyTorch prep ops around GaussianRasterizer instead of the fused render()")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--preroll-steps", type=int, default=200,
                    help="untimed steps in front of the W warm-up steps: the GPU's clocks settle under sustained load only "
                         "(reported as `preroll_steps` in the line; 0 = time a GPU that has just left the sizing pass)")
    ap.add_argument("--exchange-chunks", type=int, default=0,
                    help="Gaussian-index ranges of the overlapped gradient exchange (sink bucket): the all-reduce of a range "
                         "starts while the backward's per-Gaussian tail is still computing the next one; 1 = one all-reduce "
                         "after the backward; 0 (default) = chosen from the exchange model (trase_amd.dp.recommended_chunks: "
                         "ranges whenever the modelled exchange is longer than they cost -- every extra range costs ~0.027 ms "
                         "of kernel ramp/tail at S4, measured at N=1)")
    ap.add_argument("--active-sh-degree", type=int, default=3,
                    help="--exchange phased: the active SH degree handed to the phased exchange (train.py:160 ramps it 0 -> 3): only the "
                         "active f_rest coefficient rows are exchanged")
    ap.add_argument("--exchange", choices=["allreduce", "rs_ag", "direct", "phased"], default="allreduce",
                    help="algorithm of the gradient exchange (trase_amd.dp.FlatGradBucket): one RCCL all-reduce (default; RCCL "
                         "picks ring / tree), reduce-scatter + all-gather on the padded flat bucket, or the same two phases as "
                         "grouped point-to-point transfers to every peer at once (all seven xGMI links of a GPU busy)")
    ap.add_argument("--no-iteration-window", action="store_true",
                    help="skip the secondary window (SURVEY.md 8d: whole GAUSSIAN- / FEATURE-state iterations, `iteration_ms`)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="N=1 only: create a one-rank RCCL process group and issue the exchange collectives anyway")
    ap.add_argument("--bucket", choices=["auto", "sink", "accumulate"], default="auto",
                    help="gradient bucket of the N>1 exchange step; 'sink' / 'accumulate' force it on at N=1 (for timing the "
                         "bucket handling alone: the all-reduce is a no-op there)")
    ap.add_argument("--cpu-tile-step", type=int, default=293)
    ap.add_argument("--cpu-budget-s", type=float, default=16.0, help="wall-time budget of the CPU preprocess baseline")
    ap.add_argument("--cpu-oracle-budget-s", type=float, default=1.0, help="forward wall-time budget of the oracle sample")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = every host core (os.cpu_count())")
    ap.add_argument("--kernel-breakdown", action="store_true", help="(kept for old command lines: the per-kernel pass always runs)")
    ap.add_argument("--policy", choices=["free", "sync"], default="free",
                    help="capacity policy of the timed steps: free = set_sync(False) with a measured capacity (default, what the "
                         "metric is quoted on); sync = the drop-in default (pair count read back every forward, exact allocation)")
    ap.add_argument("--graph", choices=["auto", "on", "off"], nargs="?", const="on", default="auto",
                    help="launch-graph replay of the forward / backward launch sequences (trase_amd.rasterizer.set_graph).  auto "
                         "(default): on for the small BASELINE configurations (<= 200k Gaussians), which are host-bound otherwise; "
                         "off at the headline size, where it changes nothing (measured 902.9 vs 903.0 views/s) and the per-kernel "
                         "HIP events of the timed region -- which the roofline figure needs -- cannot be recorded inside a graph")
    ap.add_argument("--shard", choices=["views", "tiles"], default="views",
                    help="'views' (default, the headline): every rank renders a different view.  'tiles' (BASELINE config 5): ONE "
                         "view per step for the whole job, every rank renders a load-balanced strip of 16x16-tile rows, the RGB "
                         "strips are all-gathered (full-frame style loss) and the strips' partial gradients all-reduced; strong "
                         "scaling; defaults to the S5 size (2.5 M Gaussians, 1280x960) unless sizes are given")
    ap.add_argument("--dense-strip-grads", action="store_true",
                    help="tiles mode / strip table: dense per-Gaussian gradients (every row written, zeros included) instead of "
                         "trase_amd.rasterizer.set_sparse_strip_grads (only the rows of the strip's Gaussians)")
    ap.add_argument("--forward-only", action="store_true",
                    help="tiles mode / strip table: time the forward alone (rendering a frame: the use the tile-row axis is for -- "
                         "no backward, no gradient exchange)")
    ap.add_argument("--strip-table", type=str, default="",
                    help="ONE GPU: time every rank's strip (fwd+bwd, load-balanced partition) for world = 1, 2, 4, 8 and write the "
                         "predicted tile-sharding speed-up (communication excluded) to this JSON file; no bench line is printed")
    args = ap.parse_args()
    args.graph = (args.graph == "on") or (args.graph == "auto" and args.gaussians <= 200_000 and args.shard == "views" and not args.strip_table
                                          and args.policy == "free" and not args.unfused and args.gpus == 1)
    if (args.shard == "tiles" or args.strip_table) and "--gaussians" not in " ".join(sys.argv) and "--width" not in " ".join(sys.argv):
        args.gaussians, args.width, args.height = 2_500_000, 1280, 960          # S5: Google Immersive size

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if world != args.gpus:
        if "RANK" in os.environ or "LOCAL_RANK" in os.environ:
            sys.exit(f"bench.py: --gpus {args.gpus} but launched with WORLD_SIZE={world}; pass --nproc-per-node {args.gpus}")
        if torch.cuda.device_count() < args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} requested but only {torch.cuda.device_count()} GPU(s) are visible")
        # plain `python bench.py --gpus N`: spawn the N ranks ourselves, exactly like the driver's launch line
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    # TEST MODE (tests/test_gpu_bench_ranks.py): several ranks share the visible GPU(s) and talk over gloo -- every N > 1 code
    # path of this file (bucket, exchange schedules, barrier, MAX over ranks, rank 0's line) runs on a one-GPU box; the line it
    # prints says so and is not a measurement
    shared_gpu_test = os.environ.get("TRASE_BENCH_SHARED_GPU_TEST") == "1"
    if shared_gpu_test:
        local_rank = local_rank % torch.cuda.device_count()
    if torch.cuda.device_count() <= local_rank:
        sys.exit(f"bench.py: rank {rank} has no GPU (LOCAL_RANK={local_rank}, {torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared_gpu_test:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)
    elif args.force_collectives:
        import socket
        import torch.distributed as dist
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=device)
    if args.variant is not None:
        R.set_variant(args.variant)

    N, W, H, F = args.gaussians, args.width, args.height, args.feat
    P = W * H
    scene_cpu = make_scene(N, feat_dim=F, seed=0, scale_mult=args.scale_mult)
    from trase_amd.synthetic import SynthGaussianModel, SynthPipe
    from gaussian_renderer import render          # the drop-in for the reference's render()
    pc = SynthGaussianModel(scene_cpu.to(device))
    pipe = SynthPipe()
    params = pc.parameters()
    # one flat gradient bucket; .grad of every parameter is a view into it (single all-reduce)
    # (only needed when there is an exchange step; at N=1 autograd just assigns .grad)
    from trase_amd.dp import FlatGradBucket
    # "phased": FlatGradBucket.allreduce_phased (xyz first, the rest on a side stream until the next step's render) with the
    # "direct" algorithm; one exchange per step, no Gaussian-range overlap
    phased = args.exchange == "phased"
    if phased:
        args.exchange_chunks = 1
    bucket = FlatGradBucket(params, exchange="direct" if phased else args.exchange) if (world > 1 or args.bucket != "auto" or args.force_collectives) else None
    phase = {"ex": None, "bytes": None}
    if bucket is not None:
        bucket._force = bool(args.force_collectives)      # one-rank RCCL group: issue the collectives anyway
        bucket.time_exchange = True                       # HIP events around the collective(s): `exchange_ms` of the bench line
        if args.exchange_chunks == 0:
            from trase_amd.dp import recommended_chunks
            args.exchange_chunks = recommended_chunks(bucket.bytes_per_step, max(world, 1), "direct" if args.exchange == "direct" else "ring") \
                if (world > 1 and args.shard == "views") else 1
    args.exchange_chunks = max(args.exchange_chunks, 1)
    if shared_gpu_test and world > 1:
        # gloo completes an asynchronous all-reduce of a device tensor by copying the result back IN PLACE later, which bumps the
        # tensor's version after FlatGradBucket noted it (its modified-after-hand-off check then fires); RCCL's in-place collective
        # counts once, at the call.  The overlapped ranges are covered by the one-rank RCCL tests (tests/test_gpu_overlap.py).
        args.exchange_chunks = 1
    use_sink = bucket is not None and not args.unfused and args.bucket != "accumulate"
    if use_sink:
        # the fused backward writes every gradient once, straight into the bucket: no zero-fill, no accumulation pass
        from trase_amd.renderer import set_grad_sink
        if args.exchange_chunks > 1:
            set_grad_sink(**bucket.overlapped(args.exchange_chunks, force_collectives=args.force_collectives))
        else:
            set_grad_sink(bucket.sink())

    n_views = 16
    cams = [orbit_camera(W, H, angle=2 * math.pi * (k + (0 if args.shard == "tiles" else rank * 0.37)) / n_views, fid=k / n_views)
            for k in range(n_views)]
    settings = [settings_for(c, device) for c in cams]
    cams_dev = [c.to(device) for c in cams]
    bg = torch.zeros(3, device=device)
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    g_img = torch.randn(3, H, W, generator=g).to(device) / P
    g_feat = torch.randn(F, H, W, generator=g).to(device) / P

    tiles_mode = args.shard == "tiles"
    strip = {"part": None}          # tile-row partition of the current job (tiles mode)

    def step(i):
        if tiles_mode:
            return step_tiles(i)
        if bucket is not None and not use_sink:
            bucket.zero()                 # autograd accumulates into the bucket views
        else:
            for p_ in params:
                p_.grad = None            # autograd adopts the gradients (N > 1: views into the bucket, see above)
        if phase["ex"] is not None:        # phase B of the previous step's exchange: needed before this step's preprocess
            phase["ex"].wait_rest()
            phase["ex"] = None
        if not args.unfused:
            # render() drop-in: the A1 prep (activations, SH concat, feature normalisation) is fused into the
            # per-Gaussian HIP kernels
            out = render(cams_dev[i % n_views], pc, pipe, bg, 0.0, 0.0, 0.0)
            img, radii, feats = out["render"], out["radii"], out["render_gaussian_features"]
        else:
            # the reference's render() body verbatim: PyTorch prep ops around the GaussianRasterizer operator
            st = settings[i % n_views]
            means2D = torch.zeros_like(pc.get_xyz, requires_grad=True)            # gaussian_renderer/__init__.py:48
            gfeat = pc.get_gaussian_features
            sh_objs = gfeat / (gfeat.norm(dim=2, keepdim=True) + 1e-9)
            img, radii, feats, depth = GaussianRasterizer(raster_settings=st)(
                means3D=pc.get_xyz, means2D=means2D, shs=pc.get_features, sh_objs=sh_objs, colors_precomp=None,
                opacities=pc.get_opacity, scales=pc.get_scaling, rotations=pc.get_rotation, cov3D_precomp=None)
        torch.autograd.backward([img, feats], [g_img, g_feat])
        if bucket is not None and phased:
            ex = bucket.allreduce_phased(first=[pc._xyz], sh_rest=(pc._features_rest, args.active_sh_degree))
            ex.wait_first()                # (Adam(xyz, MLP) and the next iteration's MLP forward would run here)
            phase["ex"], phase["bytes"] = ex, (ex.bytes_first, ex.bytes_rest)
        elif bucket is not None:
            bucket.allreduce()
        return radii

    def step_tiles(i, rows=None):
        """One rank's share of ONE view: its strip of tile rows (forward), all-gather of the RGB strips (the full frame a
        style loss needs), backward of the strip, all-reduce of the partial gradients."""
        from trase_amd import dp
        for p_ in params:
            p_.grad = None
        b, e = rows if rows is not None else strip["part"][rank]
        if args.forward_only:          # rendering one frame (inference / GUI / evaluation): no backward, no gradient exchange
            with torch.no_grad(), (R.tile_rows(b, e) if (b, e) != (0, 0) else _nullctx()):
                out = render(cams_dev[i % n_views], pc, pipe, bg, 0.0, 0.0, 0.0)
                if world > 1 and (b, e) != (0, 0):       # (the whole-view sizing pass has nothing to gather)
                    dp.allgather_strips(out["render"], strip["part"], H)
            return out["radii"]
        with R.tile_rows(b, e) if (b, e) != (0, 0) else _nullctx():
            out = render(cams_dev[i % n_views], pc, pipe, bg, 0.0, 0.0, 0.0)
            img, radii, feats = out["render"], out["radii"], out["render_gaussian_features"]
            if world > 1 and (b, e) != (0, 0):
                img = dp.allgather_strips(img, strip["part"], H)
            torch.autograd.backward([img, feats], [g_img, g_feat])
        if bucket is not None:
            bucket.allreduce()
        return radii

    import contextlib
    _nullctx = contextlib.nullcontext

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    # ---- sizing pass (synchronising policy): measure the pair count of every view
    log(f"scene ready: N={N} {W}x{H} F={F}")
    R.set_sync(True)
    if tiles_mode or args.strip_table:
        from trase_amd import dp
        tiles_mode = True
        # sparse strip gradients pay where a strip holds a small share of the Gaussians (measured at S5: 45 % of them have a pair
        # in a half-image strip -- scattered row accesses then lose to the dense kernels; 13 % in an eighth: 1.04 -> 0.93 ms per strip there, but 1.38 -> 1.44 at four strips)
        R.set_sparse_strip_grads((not args.dense_strip_grads) and world >= 8)
        step_tiles(0, rows=(0, 0))                       # the whole view once: per-tile-row pair loads
        loads = R.last_tile_row_loads()
        full_pairs = R.last_status()[2]
        strip["part"] = dp.tile_row_partition(H, world, loads=loads)
        log(f"tile rows {len(loads)}, pairs {full_pairs}; load-balanced strips {strip['part']}")
    if args.strip_table:
        # ONE GPU: every rank's strip of world = 1, 2, 4, 8 timed in turn (communication excluded)
        table = {"workload": f"{N} Gaussians, {W}x{H}, F={F}, one view sharded by load-balanced tile-row strips"
                             + (", FORWARD ONLY (rendering)" if args.forward_only else ", forward + backward"), "steps": args.steps,
                 "tile_row_loads": [int(x) for x in loads.tolist()], "world": {}}
        for wsize in (1, 2, 4, 8):
            part = dp.tile_row_partition(H, wsize, loads=loads)
            part_eq = dp.tile_row_partition(H, wsize)
            R.set_sparse_strip_grads((not args.dense_strip_grads) and wsize >= 8)
            rec = {"strips": part, "ms": [], "pairs": [], "equal_rows_strips": part_eq, "equal_rows_ms": [],
                   "sparse_strip_grads": bool((not args.dense_strip_grads) and wsize >= 8)}
            for label, pp, dst in (("balanced", part, rec["ms"]), ("equal", part_eq, rec["equal_rows_ms"])):
                if label == "equal" and (wsize == 1 or pp == part):
                    rec["equal_rows_ms"] = list(rec["ms"])
                    continue
                for rows in pp:
                    if wsize == 1:
                        rows = (0, 0)                # one rank = the whole image, rendered as such (not as a 60-row "strip")
                    R.set_sync(True)
                    caps = []
                    for i in range(min(4, n_views)):
                        step_tiles(i, rows=rows)
                        caps.append(R.last_status()[2])
                    R.set_sync(False, capacity=int(max(caps) * 1.3) + 1024)
                    for i in range(2):
                        step_tiles(i, rows=rows)
                    best = float("inf")
                    for _ in range(3):               # best of three: a transient on the box otherwise decides the max over strips
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        for i in range(args.steps):
                            step_tiles(i, rows=rows)
                        torch.cuda.synchronize()
                        best = min(best, (time.perf_counter() - t0) / args.steps * 1e3)
                    dst.append(round(best, 4))
                    if label == "balanced":
                        rec["pairs"].append(int(sum(caps) / len(caps)))
                        if rows == pp[0]:            # where a strip's time goes: HIP-event profile of rank 0's strip
                            R.profile_enable(1)
                            for i in range(4):
                                step_tiles(i, rows=rows)
                            torch.cuda.synchronize()
                            rec["kernels_ms_rank0"] = {k: round(v["ms"] * v["n"] / 4, 4) for k, v in R.profile_report().items()}
                            R.profile_enable(0)
            rec["max_ms"], rec["equal_rows_max_ms"] = max(rec["ms"]), max(rec["equal_rows_ms"])
            table["world"][str(wsize)] = rec
            log(f"world {wsize}: strip ms {rec['ms']} (equal rows: {rec['equal_rows_ms']})")
        base = table["world"]["1"]["max_ms"]
        table["predicted_speedup_excl_comms"] = {k: round(base / v["max_ms"], 3) for k, v in table["world"].items()}
        table["predicted_speedup_equal_rows"] = {k: round(base / v["equal_rows_max_ms"], 3) for k, v in table["world"].items()}
        table["note"] = ("single-GPU prediction: step time of a W-rank job = slowest strip; the all-gather of the RGB strips and the "
                         "all-reduce of the gradient bucket are NOT included (unmeasured on multi-GPU hardware)")
        json.dump(table, open(args.strip_table, "w"), indent=1)
        print(json.dumps({"strip_table": args.strip_table, "predicted_speedup_excl_comms": table["predicted_speedup_excl_comms"]}), flush=True)
        return
    r_list, reff_list = [], []
    for i in range(n_views):
        step(i)
        st_ = R.last_status()
        r_list.append(st_[0])        # lineage definition: sum of 16x16 tiles touched
        reff_list.append(st_[2])     # (8x8 sub-tile, Gaussian) pairs actually binned after exact culling
    r_max, r_mean = max(r_list), sum(r_list) / len(r_list)
    reff_mean = sum(reff_list) / len(reff_list)
    if args.policy == "free":
        R.set_sync(False, capacity=int(max(reff_list) * 1.25) + 1024)
    log(f"pairs per view: lineage R mean {r_mean:.0f} max {r_max} (R/N {r_mean / N:.2f}); binned sub-tile pairs mean {reff_mean:.0f}")

    R.set_graph(bool(args.graph))       # (the library's own default is "auto": <= 200k Gaussians)
    if args.graph:
        for i in range(2 * n_views):      # every view's forward and backward sequence is captured once
            step(i)
    # Pre-roll: the metric is the throughput of a training loop, i.e. of a GPU that has been busy for a while.  The clock governor
    # needs ~a second of sustained load to settle (measured, one box, profiles/r5_bench_clock_state.txt: 20 timed steps right
    # after the sizing pass 903-908 views/s; after 100 / 1000 more steps 904-916 / 916-925; 5 / 30 / 200 ms of idle in front
    # of the timed steps 868-873 / 845-846 / 825-837).  These steps are untimed, like the W warm-up steps that follow them.
    for i in range(args.preroll_steps):
        step(i)
    for i in range(args.warmup):
        step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    if world > 1 or args.force_collectives:
        # RCCL prints its version banner through C stdio when the first communicator is created: push it out of EVERY rank's
        # buffer now, so that rank 0's JSON line is the last thing on the job's stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
    if not args.graph:           # (a replayed launch graph has no per-kernel events: --graph takes the kernel times of the untimed pass below)
        R.profile_enable(2)      # HIP events around the compositing kernels only, on the launch stream
    import gc
    gc.disable()                 # no collector pause inside the 20-step window (NOT gc.collect(): tens of ms of idle GPU in front
                                 # of the timed steps cost 7-9 %, see above)
    host_t = []
    # one timing event behind every step (a timestamp write in stream order: no synchronisation, no gap): the spread of the
    # window's own steps goes into the line (`step_ms`) -- VERDICT r5 weak 11: 20 steps are 22 ms, and a single slow step moves them
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    step_ev[0].record()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
        step_ev[i + 1].record()
        host_t.append(time.perf_counter())
    if phase["ex"] is not None:
        phase["ex"].wait_rest()
        phase["ex"] = None
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    gc.enable()
    host_gaps = [round((b - a) * 1e3, 3) for a, b in zip([t0] + host_t[:-1], host_t)]
    log(f"host ms per step() call: {host_gaps}; drain {(t1 - host_t[-1]) * 1e3:.3f} ms")
    prof = R.profile_report() if not args.graph else {}
    R.profile_enable(0)
    exchange_ms = bucket.exchange_ms() if bucket is not None else None      # the last timed step's collective(s)
    status = R.last_status()
    assert status[1] == 0, "pair buffer overflowed inside the timed region; result invalid"
    elapsed = t1 - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed / args.steps * 1e3
    views_per_s = (1 if tiles_mode else world) * args.steps / elapsed     # tiles mode: the whole job renders ONE view per step
    dev_raw = [step_ev[k].elapsed_time(step_ev[k + 1]) for k in range(args.steps)]
    dev_steps = sorted(dev_raw)
    step_ms = {"min": round(dev_steps[0], 4), "median": round(dev_steps[len(dev_steps) // 2], 4), "max": round(dev_steps[-1], 4),
               "slowest_step": dev_raw.index(dev_steps[-1]),
               "what": "device time between the events recorded behind consecutive timed steps (this rank); ms_per_step is the host clock over all of them"}
    log(f"timed: {ms_per_step:.3f} ms/step, {views_per_s:.1f} views/s")

    breakdown, launches = None, None
    # the per-kernel table is part of the line: roofline.forward, backward_frac and launches_per_view are built from it
    R.profile_enable(1)
    nb = min(4, n_views)
    for i in range(nb):
        step(i)
    torch.cuda.synchronize()
    rep_ = R.profile_report()
    breakdown = {k: round(v["ms"] * v["n"] / nb, 4) for k, v in rep_.items()}
    # kernel launches per view of the library's own launch sequences (one profiling scope = one launch, except scan_tiles
    # = 2); memsets and the torch-side kernels of the wrapper are not counted
    per_scope = {k: (2 if k == "scan_tiles" else 1) * v["n"] / nb for k, v in rep_.items()}
    chain = ("radix_hist", "radix_scan", "radix_scatter", "depth_sort", "emit_pairs", "scan_tiles", "tile_ranges")
    launches = {"binning_chain": round(sum(v for k, v in per_scope.items() if k in chain), 2),
                "all_kernels": round(sum(per_scope.values()), 2)}
    R.profile_enable(0)

    # ---- extra key: two views per launch sequence (trase_amd.renderer.render_views: ONE depth sort for both views) ---------------
    # not the headline (the reference renders one view per iteration, train.py:180): what a loop that accumulates two views per
    # optimizer step, or an evaluation sweep, gets; same views, same cotangents, forward + backward of both
    batched = None
    if world == 1 and not tiles_mode and not args.unfused and args.policy == "free" and bucket is None and not args.forward_only:
        try:
            from trase_amd.renderer import render_views

            def step2(i):
                for p_ in params:
                    p_.grad = None
                outs = render_views([cams_dev[(2 * i) % n_views], cams_dev[(2 * i + 1) % n_views]], pc, pipe, bg, 0.0, 0.0, 0.0)
                for o_ in outs:
                    torch.autograd.backward([o_["render"], o_["render_gaussian_features"]], [g_img, g_feat])
            def step1x2(i):            # the same two views, gradients accumulated over both, through two render() calls: the
                for p_ in params:      # comparison of the same phase of the run
                    p_.grad = None
                for j_ in (2 * i, 2 * i + 1):
                    o_ = render(cams_dev[j_ % n_views], pc, pipe, bg, 0.0, 0.0, 0.0)
                    torch.autograd.backward([o_["render"], o_["render_gaussian_features"]], [g_img, g_feat])

            def timed_pairs(fn, n):
                for i in range(2):
                    fn(i)
                torch.cuda.synchronize()
                t_ = time.perf_counter()
                for i in range(n):
                    fn(i)
                torch.cuda.synchronize()
                return (time.perf_counter() - t_) / (2 * n) * 1e3
            n2 = max(args.steps // 2, 4)
            ser, par = [], []
            for _ in range(2):           # alternating: clocks and allocator state drift over a run
                ser.append(timed_pairs(step1x2, n2))
                par.append(timed_pairs(step2, n2))
            batched = {"views_per_s": round(1e3 / min(par), 3), "ms_per_view": round(min(par), 4), "pairs_timed": n2,
                       "serial_ms_per_view_same_phase": round(min(ser), 4), "all_ms_per_view": {"serial": [round(x, 4) for x in ser], "pair": [round(x, 4) for x in par]}}
            for p_ in params:
                p_.grad = None
        except Exception as e:
            batched = {"error": f"{type(e).__name__}: {e}"}

    # ---- extra key: the price of the bf16-split channel contraction --------------------------------------------
    # the same steps with the packed-FP32 compositing kernels (TRASE_VARIANT_VALU_FORWARD | _BACKWARD: every product in fp32), same
    # run, right after the headline; and how far the two forwards' maps are apart on view 0
    fp32_variant = None
    if world == 1 and not tiles_mode and not args.unfused and F == 32 and not args.forward_only and bucket is None:
        try:
            v0 = R._Policy.variant
            with torch.no_grad():
                o_ = render(cams_dev[0], pc, pipe, bg, 0.0, 0.0, 0.0)
                ref_maps = (o_["render"].clone(), o_["render_gaussian_features"].clone(), o_["depth"].clone())
            R.set_variant(v0 | R.VARIANT_VALU_FORWARD | R.VARIANT_VALU_BACKWARD)
            try:
                with torch.no_grad():
                    o_ = render(cams_dev[0], pc, pipe, bg, 0.0, 0.0, 0.0)
                    diffs = [float((a_ - b_).abs().max()) for a_, b_ in zip(ref_maps, (o_["render"], o_["render_gaussian_features"], o_["depth"]))]
                del o_, ref_maps
                for i in range(max(args.warmup, 5) + 20):
                    step(i)
                torch.cuda.synchronize()
                t_ = time.perf_counter()
                for i in range(args.steps):
                    step(i)
                torch.cuda.synchronize()
                el_ = time.perf_counter() - t_
                R.profile_enable(1)
                for i in range(4):
                    step(i)
                torch.cuda.synchronize()
                kv_ = {k: round(v["ms"] * v["n"] / 4, 4) for k, v in R.profile_report().items() if k.startswith("render_")}
                R.profile_enable(0)
            finally:
                R.set_variant(v0)
            fp32_variant = {"views_per_s": round(args.steps / el_, 3), "ms_per_step": round(el_ / args.steps * 1e3, 4),
                            "vs_headline": round((args.steps / el_) / views_per_s, 4), "compositing_kernels_ms": kv_,
                            "max_map_diff_view0": {"image": diffs[0], "features": diffs[1], "depth": diffs[2]},
                            "what": "the same timed steps with TRASE_VARIANT_VALU_FORWARD | TRASE_VARIANT_VALU_BACKWARD (packed-FP32 "
                                    "compositing, no bf16-split MFMA contraction), measured in this run after the headline; "
                                    "max_map_diff_view0 = max |default - fp32 variant| of the forward maps of view 0"}
            for p_ in params:
                p_.grad = None
            for i in range(3):                 # back on the default kernels before the windows below
                step(i)
        except Exception as e:
            fp32_variant = {"error": f"{type(e).__name__}: {e}"}

    # ---- extra key: two views in flight on two streams (never the headline) ----------------------------------
    # rounds 4-5 could not offer this: per-Gaussian kernels produced wrong values beside the MFMA compositing kernels of another
    # launch sequence.  Round 6 traced that to compiler-generated packed-FP32 VALU and builds the library without it; here the
    # same views go round-robin to two streams (the wrapper's cross-stream ordering off), gradients through torch.autograd.grad,
    # and every view's maps + gradients are checked bit for bit against the serial run IN THIS RUN before the rate is reported
    in_flight = None
    if world == 1 and not tiles_mode and not args.unfused and args.policy == "free" and bucket is None and not args.forward_only:
        try:
            def view_ad(i, digest):
                o_ = render(cams_dev[i % n_views], pc, pipe, bg, 0.0, 0.0, 0.0)
                gr_ = torch.autograd.grad([o_["render"], o_["render_gaussian_features"]], list(params) + [o_["viewspace_points"]], [g_img, g_feat],
                                          allow_unused=True)
                if not digest:
                    return None
                ts_ = [o_["render"], o_["render_gaussian_features"], o_["depth"], o_["radii"]] + [t_ for t_ in gr_ if t_ is not None]
                return torch.stack([t_.contiguous().view(torch.int32).to(torch.int64).sum() for t_ in ts_])
            ref_ = [view_ad(i, True).cpu() for i in range(n_views)]
            R.set_stream_ordering(False)
            try:
                st2 = [torch.cuda.Stream(), torch.cuda.Stream()]
                torch.cuda.synchronize()

                def flight(n_, digest):
                    out_ = []
                    for i in range(n_):
                        with torch.cuda.stream(st2[i % 2]):
                            out_.append(view_ad(i, digest))
                    torch.cuda.synchronize()
                    return out_
                got_ = flight(2 * n_views, True)
                bad_ = sum(int(not torch.equal(ref_[i % n_views], d_.cpu())) for i, d_ in enumerate(got_))
                rates_ = []
                for _ in range(2):
                    flight(6, False)
                    t_ = time.perf_counter()
                    flight(2 * args.steps, False)
                    rates_.append(2 * args.steps / (time.perf_counter() - t_))
                ser_ = []
                for _ in range(2):
                    for i in range(6):
                        view_ad(i, False)
                    torch.cuda.synchronize()
                    t_ = time.perf_counter()
                    for i in range(2 * args.steps):
                        view_ad(i, False)
                    torch.cuda.synchronize()
                    ser_.append(2 * args.steps / (time.perf_counter() - t_))
            finally:
                R.set_stream_ordering(True)
                torch.cuda.synchronize()
            in_flight = {"views_per_s": round(max(rates_), 3), "serial_views_per_s_same_phase": round(max(ser_), 3),
                         "views_checked": 2 * n_views, "views_differing_from_serial": bad_,
                         "what": "the bench's views round-robin on TWO streams of this GPU (rasterizer.set_stream_ordering(False)), "
                                 "gradients through torch.autograd.grad; every checked view's maps and gradients compared bit for bit "
                                 "with the serial run first"}
            if bad_:
                in_flight["views_per_s"] = None          # a rate of wrong results is not a rate
        except Exception as e:
            in_flight = {"error": f"{type(e).__name__}: {e}"}

    # ---- secondary window: whole training iterations, iter_start -> iter_end of train.py:157-303 -------
    iteration_ms = None
    if world == 1 and not tiles_mode and not args.no_iteration_window and not args.unfused and (N, W, H, F) == (300_000, 1920, 1080, 32):
        try:
            from trase_amd.bench_iterations import make_feature_iteration, make_gaussian_iteration, time_iterations
            log("secondary window: 16 GAUSSIAN-state + 16 FEATURE-state iterations ...")
            cams8 = cams_dev[::2]
            it_g = make_gaussian_iteration(pc, cams8, W, H, device)
            t_g = time_iterations(it_g, iters=16, warm=40)
            it_f, restore = make_feature_iteration(pc, cams8, W, H, device)
            try:
                t_f = time_iterations(it_f, iters=16, warm=40)
            finally:
                restore()
            # the same iterations captured WHOLE into one torch.cuda.CUDAGraph each and replayed: no host work
            # between the launches at all.  (Freezes the host-side draws of the capture -- see capture_iteration.)
            graph_ms = {}
            try:
                from trase_amd.bench_iterations import capture_iteration, time_graph_replays
                gg, keep_g = capture_iteration(it_g, warm=2)
                graph_ms["gaussian"] = round(time_graph_replays(gg, iters=16, warm=24), 3)
                del gg, keep_g
                it_f2, restore2 = make_feature_iteration(pc, cams8, W, H, device)
                try:
                    gf_, keep_f = capture_iteration(it_f2, warm=2)
                    graph_ms["feature"] = round(time_graph_replays(gf_, iters=16, warm=24), 3)
                    del gf_, keep_f
                finally:
                    restore2()
            except Exception as e:
                graph_ms["error"] = f"{type(e).__name__}: {e}"
            iteration_ms = {"gaussian": round(t_g, 3), "feature": round(t_f, 3), "whole_iteration_graph_replay": graph_ms,
                            "what": "one whole training iteration without the optimizer step (train.py:157-303), all-HIP path: "
                                    "GAUSSIAN state = deformation MLP with gradients + render() (image scope) + L1/SSIM + backward; "
                                    "FEATURE state = MLP under no_grad + render(KNN-smoothed normalised features) + contrastive "
                                    "head on 100 masks / 5000 sampled pixels + backward; 16 iterations each after 40 untimed ones (clocks settled, see preroll_steps)"}
            for p_ in params:
                p_.grad = None
        except Exception as e:                                  # the headline line must not depend on the secondary window
            iteration_ms = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        r_used = sum(r_list[i % n_views] for i in range(args.steps)) / args.steps
        dom = max(prof.items(), key=lambda kv: kv[1]["ms"] * kv[1]["n"])[0] if prof else "render_bwd"
        dom_ms = prof[dom]["ms"] if prof else (breakdown or {}).get(dom, float("nan"))
        a_bytes = algorithmic_bytes(dom, N, r_used, P, F)
        achieved = a_bytes / (dom_ms * 1e-3) / 1e9
        # per-launch PMC figures of the same command (profiles/run_pmc.sh -> profiles/pmc_per_launch.json):
        # HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB per the gfx950 note of MI355X_MICROARCH.md; VALU wave-instructions
        # These are REPLAYED from a tracked table, not counters of this run; `counters_source` says so, and they are only
        # reported when the table was collected on the kernel sources this library is built from.
        traffic = valu_insts = mfma_insts = None
        counters_source = None
        fwd_rec = {}
        tpath = os.path.join(ROOT, "profiles", "pmc_per_launch.json")
        if os.path.exists(tpath) and (N, W, H, F) == (300_000, 1920, 1080, 32) and not args.unfused:   # collected on S4 only
            try:
                table = json.load(open(tpath))
                rec = table.get(dom) or {}
                sha_tab, sha_now = table.get("_source_sha16"), source_sha16()
                if sha_tab == sha_now:
                    traffic, valu_insts, mfma_insts = rec.get("hbm_bytes"), rec.get("valu_insts"), rec.get("mfma_insts")
                    fwd_rec = table.get("render_fwd") or {}
                    counters_source = (f"replayed from profiles/pmc_per_launch.json ({table.get('_from', '?')}; rocprofv3 PMC passes of "
                                       f"this command on kernel sources {sha_tab}) -- not counters of this run")
                else:
                    counters_source = (f"none: profiles/pmc_per_launch.json was collected on kernel sources {sha_tab}, this build is "
                                       f"{sha_now}; traffic / valu_frac withheld")
            except Exception:
                pass
        # VALU fraction ("report HBM fraction AND VALU fraction"): a wave64 VALU instruction occupies its SIMD
        # for 4 cycles; 256 CUs x 4 SIMDs at the 2.4 GHz peak clock
        valu_frac = None if valu_insts is None else valu_insts * 4.0 / (N_SIMD * dom_ms * 1e-3 * CLOCK_HZ)
        out = {
            # BASELINE.json's metric names the S4 configuration; other sizes (parity-test configurations) say so
            "metric": ("TEST MODE (ranks share a GPU over gloo: not a measurement) " if shared_gpu_test else "") + ("views/sec (fwd+bwd), 1080p, 300k Gaussians, 32-d feat" if (N, W, H, F) == (300_000, 1920, 1080, 32)
                       else f"views/sec (fwd+bwd), {W}x{H}, {N} Gaussians, {F}-d feat") + (", one view tile-row sharded over the ranks" if tiles_mode else ""),
            "value": round(views_per_s, 3), "unit": "views/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "preroll_steps": args.preroll_steps, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong" if tiles_mode else "weak", "vs_baseline": None,
            "dtype": "f32 (channel contractions: 3-product bf16-split MFMA, fp32 accumulate)", "data": "synthetic",
            "config": {"workload": f"{'S4 headline' if (N, W, H, F) == (300_000, 1920, 1080, 32) else 'custom'}: {N} Gaussians, {W}x{H}, F={F}, SH deg 3, one view per step per GPU"
                                   + (", view-DP + RCCL all-reduce of Gaussian grads" if (world > 1 and not tiles_mode) else "")
                                   + (f", ONE view per step sharded by load-balanced tile-row strips {strip['part']}, RGB strips all-gathered, "
                                      "partial gradients all-reduced" if tiles_mode else ""),
                       "pairs_R_mean": round(r_mean), "pairs_R_max": r_max, "R_over_N": round(r_mean / N, 2),
                       "subtile_pairs_mean": round(reff_mean),
                       "tiles": ((W + 15) // 16) * ((H + 15) // 16), "variant": R._Policy.variant, "capacity_policy": args.policy,
                       "graph_replay": (R.graph_stats() if args.graph else None),
                       "entry": "GaussianRasterizer + PyTorch prep (reference render() body)" if args.unfused
                                else "gaussian_renderer.render() drop-in, A1 prep fused",
                       "bucket_bytes": (None if bucket is None else bucket.bytes_per_step),
                       # HIP events on the compute stream: one exchange = the collective(s) of allreduce(); with overlapped
                       # ranges the window opens at the first range's hand-off, i.e. it includes the backward tail underneath
                       "exchange_ms": (None if exchange_ms is None else round(exchange_ms, 4)),
                       "exchange_algo": (None if bucket is None else args.exchange),
                       "phased_bytes_first_rest": phase["bytes"],
                       "exchange": (None if bucket is None else
                                    f"flat bucket {bucket.bytes_per_step} B/step, "
                                    + ("zero + accumulate" if not use_sink else
                                       (f"sink, all-reduce in {args.exchange_chunks} Gaussian ranges overlapped with the backward tail"
                                        if args.exchange_chunks > 1 else "sink, one all-reduce after the backward")))},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "counters_source": counters_source,
                         "valu_frac": None if valu_frac is None else round(valu_frac, 4), "valu_insts": valu_insts,
                         "mfma_insts": mfma_insts,
                         "kernel_ms": round(dom_ms, 4), "algorithmic_bytes": int(a_bytes),
                         "view_frac": round(view_bytes(N, r_used, P, F) * views_per_s / (1 if tiles_mode else world) / 1e9 / HBM_PEAK_GBS, 5)},
            "kernels_ms_per_view": breakdown,
            "launches_per_view": launches,
            "iteration_ms": iteration_ms,
            "step_ms": step_ms,
            "fp32_variant": fp32_variant,
            "views_in_flight_2": in_flight,
            "batched_2_views_per_s": (None if batched is None else batched.get("views_per_s")),
            "batched_2_views": batched,
        }
        # the forward compositing kernel beside the dominant one, and the VALU-issue fraction of the two together: both kernels
        # sit at ~0.65-0.70 of the VALU issue rate at their register-limited residency, i.e. the roofline that binds them is NOT
        # the HBM one `frac` is quoted on (SURVEY 8d predicted the VALU ceiling) -- VERDICT r4 item 7
        fwd_ms = (prof.get("render_fwd") or {}).get("ms") if prof else None
        if fwd_ms is None:
            fwd_ms = (breakdown or {}).get("render_fwd")
        if fwd_ms:
            f_bytes = algorithmic_bytes("render_fwd", N, r_used, P, F)
            f_valu = fwd_rec.get("valu_insts")
            out["roofline"]["forward"] = {
                "kernel": "render_fwd", "kernel_ms": round(fwd_ms, 4), "algorithmic_bytes": int(f_bytes),
                "achieved": round(f_bytes / (fwd_ms * 1e-3) / 1e9, 2), "frac": round(f_bytes / (fwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "traffic": fwd_rec.get("hbm_bytes"), "valu_insts": f_valu, "mfma_insts": fwd_rec.get("mfma_insts"),
                "valu_frac": None if f_valu is None else round(f_valu * 4.0 / (N_SIMD * fwd_ms * 1e-3 * CLOCK_HZ), 4)}
            if f_valu is not None and valu_insts is not None and dom == "render_bwd":
                out["roofline"]["compositing_valu_frac"] = round((f_valu + valu_insts) * 4.0 / (N_SIMD * (fwd_ms + dom_ms) * 1e-3 * CLOCK_HZ), 4)
            out["roofline"]["bound_note"] = (
                "frac / forward.frac are HBM fractions of SURVEY 8(d)'s algorithmic bytes; both compositing kernels are VALU-issue-bound "
                "at those fractions (valu_frac, forward.valu_frac, compositing_valu_frac = VALU wave-instructions x 4 cycles / "
                f"({N_SIMD} SIMDs x kernel time x {CLOCK_HZ / 1e9:.1f} GHz), registers limiting residency): the binding roofline is the "
                "vector ALU issue rate, not HBM bandwidth")
        if breakdown and breakdown.get("render_bwd") and breakdown.get("reduce_rows") and dom == "render_bwd":
            # SURVEY 8(d) charges the per-Gaussian gradient write to the backward compositing; in this design reduce_rows does
            # that write (and re-reads the per-pair rows): backward compositing as 8(d) defines it = render_bwd + reduce_rows
            both_ms = breakdown["render_bwd"] + breakdown["reduce_rows"]
            out["roofline"]["backward_frac"] = round(a_bytes / (both_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
            out["roofline"]["backward_ms"] = round(both_ms, 4)
        if world == 1 and not args.no_cpu_baseline:
            host = os.cpu_count() or 1
            # the reference's Python preprocess is a chain of small elementwise torch ops: with one thread per core of a
            # big host the intra-op thread pool costs more than it gives, so the leg is timed twice -- on every core (as
            # BASELINE.md section 4 prescribes) and on min(cores, 32) threads -- and the FASTER one is the baseline
            runs = []
            # (the all-cores leg is capped at 64 threads: on a 256-core host ONE view of this op chain took 23 s with a
            # thread per core -- pure pool overhead -- and was then discarded for the 32-thread run)
            for th in sorted({args.cpu_threads or min(host, 64), min(host, 32)}, reverse=True):
                print(f"[bench] cpu baseline: reference PyTorch-CPU preprocess on {th} of {host} host cores ...", file=sys.stderr, flush=True)
                sec_view, reps = cpu_preprocess_baseline(scene_cpu, cams[0], args.cpu_budget_s / 2, th)
                runs.append({"threads": th, "ms_per_view": round(sec_view * 1e3, 3), "views_timed": reps})
            best = min(runs, key=lambda r: r["ms_per_view"])
            sec_view, threads = best["ms_per_view"] * 1e-3, best["threads"]
            pre_ms = (breakdown or {}).get("preprocess_fwd")
            out["cpu_baseline"] = {
                "value": round(1.0 / sec_view, 4), "unit": "views/s (preprocess stage only)", "cores": threads, "kind": "port",
                "host_cores": host, "gaussians_per_s": round(N / sec_view, 1), "ms_per_view": round(sec_view * 1e3, 3),
                "hip_preprocess_ms_per_view": pre_ms, "runs": runs,
                "sample": f"the reference's PyTorch-CPU preprocess restated (oracle/cpu_preprocess.py; fp32; activations + "
                          f"eval_sh colours + covariance + projection) over ALL {N} Gaussians of the workload, whole views, "
                          f"no extrapolation; timed with {[r['threads'] for r in runs]} torch threads, the faster run "
                          f"({threads} threads, {best['views_timed']} views) is reported"}
            print(f"[bench] cpu oracle sample (secondary) ...", file=sys.stderr, flush=True)
            torch.set_num_threads(min(host, 16))
            dt, r_cpu, pairs = cpu_baseline(scene_cpu, cams[0], F, args.cpu_tile_step, args.cpu_oracle_budget_s)
            est_full = dt * r_cpu / max(pairs, 1)
            out["cpu_oracle_sample"] = {"value": round(1.0 / est_full, 6), "unit": "views/s (full fwd+bwd, extrapolated)",
                                        "cores": min(host, 16), "kind": "port",
                                        "sample": f"float64 oracle fwd+bwd on view 0: all {N} Gaussians preprocessed, "
                                                  f"{pairs} of {r_cpu} (tile,Gaussian) pairs composited (every "
                                                  f"{args.cpu_tile_step}th tile, {dt:.1f} s measured, extrapolated by pair count)"}
        try:                                   # RCCL prints its version banner through C stdio: get it out BEFORE the JSON line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if world > 1 or args.force_collectives:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
