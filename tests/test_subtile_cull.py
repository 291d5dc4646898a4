"""Exact sub-tile culling (gs_math.h subtile_live) must be conservative: it may keep a block no
pixel of which passes the blend gate, but it must never drop a block in which some pixel does."""
import ctypes as C

import numpy as np


def test_subtile_cull_is_conservative_and_tight(hostsim):
    hostsim.hs_subtile_live.restype = C.c_int
    hostsim.hs_subtile_live.argtypes = [C.c_float] * 6 + [C.c_int] * 4
    rng = np.random.default_rng(0)
    W, H = 200, 120
    kept = dropped = wrongly_dropped = kept_dead = 0
    for _ in range(12000):
        gx, gy = rng.uniform(-30, W + 30), rng.uniform(-30, H + 30)
        # random SPD covariance -> conic
        s1, s2, th = rng.uniform(0.6, 25.0), rng.uniform(0.6, 25.0), rng.uniform(0, np.pi)
        R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        cov = R @ np.diag([s1 * s1, s2 * s2]) @ R.T
        con = np.linalg.inv(cov)
        A, B, Cc = np.float32(con[0, 0]), np.float32(con[0, 1]), np.float32(con[1, 1])
        op = np.float32(rng.choice([rng.uniform(0.001, 0.02), rng.uniform(0.02, 1.0)]))
        # a block in the neighbourhood of the Gaussian (that is where the decision is non-trivial)
        reach = 3.5 * max(s1, s2)
        bx = 8 * int((gx + rng.uniform(-reach, reach)) // 8)
        by = 8 * int((gy + rng.uniform(-reach, reach)) // 8)
        if bx < 0 or by < 0 or bx >= W or by >= H:
            continue
        xs = np.arange(bx, min(bx + 8, W), dtype=np.float64)
        ys = np.arange(by, min(by + 8, H), dtype=np.float64)
        dx = np.float64(np.float32(gx)) - xs[None, :]
        dy = np.float64(np.float32(gy)) - ys[:, None]
        power = -0.5 * (float(A) * dx * dx + float(Cc) * dy * dy) - float(B) * dx * dy
        alpha = np.minimum(0.99, float(op) * np.exp(np.minimum(power, 0)))
        passes = bool(((power <= 0) & (alpha >= 1.0 / 255.0 * (1 - 1e-6))).any())
        live = bool(hostsim.hs_subtile_live(np.float32(gx), np.float32(gy), A, B, Cc, op, int(bx), int(by), W, H))
        if live:
            kept += 1
            kept_dead += (not passes)
        else:
            dropped += 1
            wrongly_dropped += passes
    assert wrongly_dropped == 0
    assert dropped > 500 and kept > 500
    # the continuous-box bound is tight: few kept blocks are dead (the slack is the gap between
    # the pixel lattice and the continuous box)
    assert kept_dead < 0.25 * kept, (kept_dead, kept)


def test_row_span_prefilter_never_drops_a_live_subtile(hostsim):
    """subtile_row_span (gs_math.h) only pre-filters the candidates that subtile_cull_live decides: over random
    splats -- elongated, rotated, near and beyond the image border, tiny and huge, low opacity -- the pruned
    enumeration must find exactly the live set of the full-rect enumeration, and should test far fewer candidates."""
    hostsim.hs_subtile_enumerate.restype = C.c_int
    hostsim.hs_subtile_enumerate.argtypes = [C.c_float] * 6 + [C.c_int] * 3 + [C.POINTER(C.c_int)] * 3
    rng = np.random.default_rng(1)
    W, H = 333, 205                                   # not multiples of 8 or 16
    tot_live = tot_full = tot_pruned = 0
    for it in range(40000):
        gx, gy = rng.uniform(-40, W + 40), rng.uniform(-40, H + 40)
        big = it % 7 == 0
        s1 = rng.uniform(0.55, 60.0 if big else 12.0)
        s2 = s1 * rng.uniform(0.02, 1.0) if it % 3 == 0 else rng.uniform(0.55, 60.0 if big else 12.0)
        th = rng.uniform(0, np.pi)
        R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        cov = R @ np.diag([s1 * s1, max(s2, 0.55) ** 2]) @ R.T
        con = np.linalg.inv(cov)
        A, B, Cc = np.float32(con[0, 0]), np.float32(con[0, 1]), np.float32(con[1, 1])
        op = np.float32(rng.choice([rng.uniform(0.004, 0.02), rng.uniform(0.02, 1.0)]))
        lam = 0.5 * (cov[0, 0] + cov[1, 1]) + np.sqrt(max(0.1, (0.5 * (cov[0, 0] + cov[1, 1])) ** 2 - np.linalg.det(cov)))
        radius = int(np.ceil(3.0 * np.sqrt(lam)))
        mm, nf, npr = C.c_int(), C.c_int(), C.c_int()
        live = hostsim.hs_subtile_enumerate(np.float32(gx), np.float32(gy), A, B, Cc, op, radius, W, H, C.byref(mm),
                                            C.byref(nf), C.byref(npr))
        assert mm.value == 0, (it, gx, gy, float(A), float(B), float(Cc), float(op), radius, live, mm.value)
        tot_live += live; tot_full += nf.value; tot_pruned += npr.value
    assert tot_live > 100000
    assert tot_pruned < 0.75 * tot_full, (tot_pruned, tot_full)       # the pre-filter does prune
    assert tot_pruned >= tot_live


def test_row_interval_rule_is_conservative_and_matches_the_block_rule(hostsim):
    """subtile_row_live (gs_math.h) decides a whole sub-tile row at once from the x-extent of the ellipse inside the
    row's band.  Over random splats (elongated, rotated, crossing the image border, tiny, huge, low opacity) its live set
    must (a) contain every block in which some PIXEL passes the blend gate -- checked against a float64 pixel-level
    evaluation --, (b) contain the per-block rule's set up to a vanishing number of boundary blocks, and (c) be only
    marginally larger than it."""
    fn = hostsim.hs_subtile_rows
    fn.restype = C.c_int
    fn.argtypes = [C.c_float] * 6 + [C.c_int] * 3 + [C.POINTER(C.c_int)] * 2 + [C.POINTER(C.c_ubyte), C.c_int, C.POINTER(C.c_int)]
    rng = np.random.default_rng(2)
    W, H = 333, 205
    tot_row = tot_only_row = tot_only_blk = missed = checked = 0
    mask = (C.c_ubyte * 65536)()
    rect = (C.c_int * 4)()
    for it in range(30000):
        gx, gy = rng.uniform(-40, W + 40), rng.uniform(-40, H + 40)
        big = it % 7 == 0
        s1 = rng.uniform(0.55, 60.0 if big else 12.0)
        s2 = s1 * rng.uniform(0.02, 1.0) if it % 3 == 0 else rng.uniform(0.55, 60.0 if big else 12.0)
        th = rng.uniform(0, np.pi)
        R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        cov = R @ np.diag([s1 * s1, max(s2, 0.55) ** 2]) @ R.T
        con = np.linalg.inv(cov)
        A, B, Cc = np.float32(con[0, 0]), np.float32(con[0, 1]), np.float32(con[1, 1])
        op = np.float32(rng.choice([rng.uniform(0.004, 0.02), rng.uniform(0.02, 1.0)]))
        lam = 0.5 * (cov[0, 0] + cov[1, 1]) + np.sqrt(max(0.1, (0.5 * (cov[0, 0] + cov[1, 1])) ** 2 - np.linalg.det(cov)))
        radius = int(np.ceil(3.0 * np.sqrt(lam)))
        ob, orow = C.c_int(), C.c_int()
        live = fn(np.float32(gx), np.float32(gy), A, B, Cc, op, radius, W, H, C.byref(ob), C.byref(orow), mask, 65536, rect)
        tot_row += live; tot_only_row += orow.value; tot_only_blk += ob.value
        if it % 10 == 0:                                # pixel-level truth on a subset (it is the slow part)
            sx0, sy0, sx1, sy1 = rect[0], rect[1], rect[2], rect[3]
            ncol = sx1 - sx0
            if ncol <= 0 or (sy1 - sy0) * ncol > 65536:
                continue
            xs = np.arange(sx0 * 8, min(sx1 * 8, W), dtype=np.float64)
            ys = np.arange(sy0 * 8, min(sy1 * 8, H), dtype=np.float64)
            if xs.size == 0 or ys.size == 0:
                continue
            dx = np.float64(np.float32(gx)) - xs[None, :]
            dy = np.float64(np.float32(gy)) - ys[:, None]
            power = -0.5 * (float(A) * dx * dx + float(Cc) * dy * dy) - float(B) * dx * dy
            ok = (power <= 0) & (float(op) * np.exp(np.minimum(power, 0)) >= 1.0 / 255.0 * (1 - 1e-6))
            for r in range(ys.size // 8 + (ys.size % 8 > 0)):
                for c in range(xs.size // 8 + (xs.size % 8 > 0)):
                    if ok[8 * r:8 * r + 8, 8 * c:8 * c + 8].any():
                        checked += 1
                        missed += not (mask[r * ncol + c] & 2)
    assert missed == 0 and checked > 20000, (missed, checked)
    assert tot_row > 100000
    assert tot_only_blk <= 1e-4 * tot_row, (tot_only_blk, tot_row)      # boundary blocks at rounding level only
    assert tot_only_row <= 0.01 * tot_row, (tot_only_row, tot_row)      # and the row rule is as tight as the block rule
