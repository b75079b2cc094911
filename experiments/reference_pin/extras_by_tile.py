#!/usr/bin/env python3
"""VERDICT r4 #9c, one bounded look from the other side: are the reference's "extra" pixels (8-spp documentation render brighter than the oracle's) a property
of SCHEDULING — the 16x16 tile a pixel lies in, the order the 28 worker threads took the tiles in — rather than of path logic?  If a tile-level state were
involved (a per-tile sampler clone, arena or film tile carrying something over), extras would cluster by tile: the count per tile would be over-dispersed against
a binomial with the picture's mean rate, and neighbouring tiles in Morton (= hand-out) order would correlate.  usage: python experiments/reference_pin/extras_by_tile.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import pyoracle
from rs_pbrt_amd import scenes
from tests.test_reference_pin import G, to_u8

sc = scenes.cornell_box_docs(pyoracle.bvh_build)
r = pyoracle.render(sc, scenes.cornell_docs_render_desc(8), threads=os.cpu_count() or 8)
ours = to_u8(scenes.film_to_rgb(r["film"])).reshape(G["spp8"].shape)
ref = G["spp8"].astype(np.int32)
extra = (ref - ours).max(-1) > 0            # the reference is brighter in some channel
lower = (ours - ref).max(-1) > 0
h, w = extra.shape
print("pixels %d, extras %d (%.4f), ours above %d" % (extra.size, extra.sum(), extra.mean(), lower.sum()))
ts = 16
nty, ntx = (h + ts - 1) // ts, (w + ts - 1) // ts
cnt = np.zeros((nty, ntx)); npx = np.zeros((nty, ntx))
for ty in range(nty):
    for tx in range(ntx):
        blk = extra[ty * ts:(ty + 1) * ts, tx * ts:(tx + 1) * ts]
        cnt[ty, tx] = blk.sum(); npx[ty, tx] = blk.size
p = extra.mean()
# dispersion against a binomial per tile with the tile's own lit fraction removed: restrict to tiles whose pixels are all non-black in the reference
lit = np.zeros((nty, ntx), bool)
for ty in range(nty):
    for tx in range(ntx):
        lit[ty, tx] = (ref[ty * ts:(ty + 1) * ts, tx * ts:(tx + 1) * ts].max(-1) > 8).all() and npx[ty, tx] == ts * ts
c = cnt[lit]; n = ts * ts
pl = c.sum() / (len(c) * n)
var_binom = n * pl * (1 - pl)
print("fully lit tiles %d: mean extras per tile %.2f, variance %.2f, binomial variance %.2f -> dispersion index %.2f" % (len(c), c.mean(), c.var(), var_binom, c.var() / var_binom))
# the same statistic for a spatially smooth rate: compare with the variance explained by the local mean (3x3 tile neighbourhood)
from scipy.ndimage import uniform_filter
rate = cnt / np.maximum(npx, 1)
smooth = uniform_filter(rate, 3, mode="nearest")
resid = (rate - smooth)[lit]
print("residual of a tile's rate against its 3x3 neighbourhood mean: std %.4f; a binomial tile of this rate has std %.4f" % (resid.std(), np.sqrt(pl * (1 - pl) / n * (1 + 1 / 9))))
# Morton (hand-out) order: lag-1 autocorrelation of the tile counts in BlockQueue order
def morton2(x, y):
    m = 0
    for b in range(16):
        m |= ((x >> b) & 1) << (2 * b) | ((y >> b) & 1) << (2 * b + 1)
    return m
order = sorted(((ty, tx) for ty in range(nty) for tx in range(ntx)), key=lambda t: morton2(t[1], t[0]))
seq = np.array([rate[t] for t in order if lit[t]])
ac = np.corrcoef(seq[:-1], seq[1:])[0, 1]
rng = np.random.default_rng(1)
perm = [np.corrcoef(q[:-1], q[1:])[0, 1] for q in (rng.permutation(seq) for _ in range(2000))]
print("lag-1 autocorrelation of the per-tile rate in Morton order %.3f (shuffled: mean %.3f, 99th percentile %.3f)" % (ac, np.mean(perm), np.percentile(perm, 99)))
# position inside the tile: is any pixel position (first pixel of a tile, first row, ...) over-represented?
pos = np.zeros((ts, ts)); tot = np.zeros((ts, ts))
for ty in range(nty):
    for tx in range(ntx):
        if not lit[ty, tx]: continue
        pos += extra[ty * ts:(ty + 1) * ts, tx * ts:(tx + 1) * ts]; tot += 1
pr = pos / tot
print("rate by position inside a tile: min %.4f max %.4f mean %.4f; binomial std of one position %.4f; first pixel of the tile %.4f, first row %.4f, last row %.4f" % (
    pr.min(), pr.max(), pr.mean(), np.sqrt(pl * (1 - pl) / tot[0, 0]), pr[0, 0], pr[0].mean(), pr[-1].mean()))
