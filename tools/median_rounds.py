#!/usr/bin/env python3
"""How many parallel rounds does the recursive (in-place) 3x3 median need?  (analysis tool, CPU only; uses the oracle's dumps)

The reference filters in place in raster order (adcensus_util.cpp MedianFilter on disp_left itself): the window of (x, y) holds
the FILTERED values of (x-1,y-1), (x,y-1), (x+1,y-1), (x-1,y) and the unfiltered values of the other five -- a triangular system,
whose unique solution any chaotic iteration reaches (like the region voting).  K11 resolves it as a wavefront
(W + 2H = 4080 levels at 1080p, one wave per band of 64 rows: 0.68 ms).  This tool measures the alternative on real maps:
  jacobi        every pixel from the previous iterate: rounds until a whole round changes nothing (= the sequential result)
  blocked T     tiles of S x S pixels, each loaded with a halo of T (top, left, right), T local rounds in the tile, the core
                written back IN PLACE while other tiles read it (tiles in random order = any interleaving): "kernels" until one
                changes nothing.
    python tools/median_rounds.py [noise|structured] [W H D seed]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PRED = [(-1, -1), (-1, 0), (-1, 1), (0, -1)]          # filtered when (x, y) is computed
REST = [(0, 0), (0, 1), (1, -1), (1, 0), (1, 1)]      # still unfiltered
BIG = np.float32(3.0e38)                               # stands for the invalid value (+inf) so that it sorts below the padding


def padded(a, fill):
    p = np.full((a.shape[0] + 2, a.shape[1] + 2), fill, np.float32)
    p[1:-1, 1:-1] = a
    return p


def window(cur_p, inp_p, y0, y1, x0, x1, iy=0, ix=0):
    """stack of the nine window values of the pixels [y0,y1) x [x0,x1) (coordinates of cur_p's array; the unfiltered values come
    from inp_p at the same pixels shifted by (iy, ix)); outside the image a side-centre neighbour counts as -inf and a corner
    neighbour as +inf (= the reference's `wnd[n/2]` of the in-image values)"""
    out = []
    for dy, dx in PRED:
        out.append(cur_p[0 if dy and dx else 1][1 + y0 + dy:1 + y1 + dy, 1 + x0 + dx:1 + x1 + dx])
    for dy, dx in REST:
        out.append(inp_p[0 if dy and dx else 1][1 + iy + y0 + dy:1 + iy + y1 + dy, 1 + ix + x0 + dx:1 + ix + x1 + dx])
    return np.stack(out)


def F(cur, inp, y0, y1, x0, x1):
    cp = (padded(cur, np.inf), padded(cur, -np.inf))     # [0]: corner padding, [1]: side padding
    ip = (padded(inp, np.inf), padded(inp, -np.inf))
    return np.sort(window(cp, ip, y0, y1, x0, x1), axis=0)[4]


def ring_padded(a):
    """a with one ring of cells around it whose values make the PLAIN median of nine reproduce the window rule on the four image
    edges: among any three consecutive ring cells along an edge one is -inf and two are +inf (period 3), so an edge pixel's three
    missing neighbours count as one value below and two above everything -- exactly what the side-centre / corner substitution
    does.  (The four corner pixels miss five neighbours and keep the explicit rule.)"""
    H, W = a.shape
    p = np.full((H + 2, W + 2), np.inf, np.float32)
    p[1:-1, 1:-1] = a
    lo = np.float32(-np.inf)
    for c in range(0, W, 3):
        p[0, 1 + c] = lo
        p[H + 1, 1 + c] = lo
    for r in range(0, H, 3):
        p[1 + r, 0] = lo
        p[1 + r, W + 1] = lo
    return p


def F_ring(cur, inp):
    """one Jacobi round with the ring-padded maps and NO border logic except the four corner pixels"""
    H, W = inp.shape
    cp, ip = ring_padded(cur), ring_padded(inp)
    out = []
    for dy, dx in PRED:
        out.append(cp[1 + dy:1 + H + dy, 1 + dx:1 + W + dx])
    for dy, dx in REST:
        out.append(ip[1 + dy:1 + H + dy, 1 + dx:1 + W + dx])
    new = np.sort(np.stack(out), axis=0)[4]
    ref = F(cur, inp, 0, H, 0, W)  # (the explicit rule, used for the four corner pixels only)
    for y in (0, H - 1):
        for x in (0, W - 1):
            new[y, x] = ref[y, x]
    return new


def fin(m):
    return np.where(np.isinf(m), BIG, m).astype(np.float32)


def jacobi(inp):
    """-> (fixed point, rounds incl. the one that changes nothing)"""
    H, W = inp.shape
    cur, rounds = inp.copy(), 0
    while True:
        new = F(cur, inp, 0, H, 0, W)
        rounds += 1
        ch = int((new != cur).sum())
        cur = new
        if ch == 0:
            return cur, rounds


def blocked(inp, S, T, rng):
    """tiles of S x S, halo T (top, left, right), T local rounds per tile and kernel, cores written back in place in random tile
    order -> (fixed point, kernels incl. the one that changes nothing)"""
    H, W = inp.shape
    ip_all = (padded(inp, np.inf), padded(inp, -np.inf))
    cur, kernels = inp.copy(), 0
    tiles = [(ty, tx) for ty in range(0, H, S) for tx in range(0, W, S)]
    while True:
        kernels += 1
        changed = 0
        for k in rng.permutation(len(tiles)):
            ty, tx = tiles[k]
            # region whose values the core can depend on after T rounds: T rows above, T columns left and right
            y0, x0, x1 = max(ty - T, 0), max(tx - T, 0), min(tx + S + T, W)
            y1 = min(ty + S, H)
            loc_c = cur[y0:y1, x0:x1].copy()
            for _ in range(T):
                cp = (padded(loc_c, np.inf), padded(loc_c, -np.inf))   # (wrong along cut edges: those pixels are held fixed)
                h, w = loc_c.shape
                new = np.sort(window(cp, ip_all, 0, h, 0, w, y0, x0), axis=0)[4]
                # pixels on a cut edge keep their snapshot value (their window is incomplete); after T rounds their influence
                # has just reached the core's border -- the shrinking valid region of temporal blocking
                if y0 > 0: new[0, :] = loc_c[0, :]
                if x0 > 0: new[:, 0] = loc_c[:, 0]
                if x1 < W: new[:, -1] = loc_c[:, -1]
                loc_c = new
            core = loc_c[ty - y0:ty - y0 + S, tx - x0:tx - x0 + S]
            dst = cur[ty:ty + S, tx:tx + S]
            changed += int((core != dst).sum())
            dst[...] = core
        if changed == 0:
            return cur, kernels


def main():
    from adcensus_amd import workloads
    from oracle import pyoracle
    kind = sys.argv[1] if len(sys.argv) > 1 else "noise"
    a = sys.argv[2:6]
    W, H, D, seed = (int(v) for v in (a + ["960", "540", "128", "12345" if kind == "noise" else "777"][len(a):]))
    l, r = (workloads.noise_pair(W, H, seed=seed) if kind == "noise" else workloads.structured_pair(W, H, D, seed=seed))
    o = pyoracle.load("auto").run(l, r, pyoracle.Option(max_disparity=D))
    inp, ref = fin(o["disp_after_dda"]), fin(o["disp_final"])
    print("%s %dx%d D=%d" % (kind, W, H, D))
    cur, rounds = jacobi(inp)
    print("  jacobi: %d rounds (the last one changes nothing); equals the reference's in-place result: %s" % (rounds, bool((cur == ref).all())))
    rng = np.random.default_rng(1)
    for S, T in ((64, 8), (32, 8), (64, 16), (32, 4)):
        cur, kernels = blocked(inp, S, T, rng)
        print("  blocked: tiles %2dx%-2d, %2d local rounds: %3d kernels (the last one changes nothing); equals the reference: %s"
              % (S, S, T, kernels, bool((cur == ref).all())))


if __name__ == "__main__":
    main()
