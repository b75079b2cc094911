import os, sys, numpy as np
mode = sys.argv[1]; os.environ['ORC_RR_MODE'] = mode; os.environ['ORC_RR_START'] = '3'
HERE = os.path.dirname(os.path.abspath(__file__)); REPO = os.environ.get('REPO', '/root/repo')   # run from the scratch copy (README.md)
sys.path.insert(0, HERE); sys.path.insert(1, REPO); sys.path.insert(2, os.path.join(REPO, 'tools'))
from oracle import pyoracle
import recover_cornell_docs as r
sc = r.build(r.FROZEN_TRIS, r.FROZEN_LIGHT, quads=r.FROZEN_QUADS)
u5 = r.to_u8(r.render(sc, 8)); ref = r.U8[8]
def hist(d): return np.bincount(np.minimum(d, 12).ravel(), minlength=13)[1:] / d.size
print('reference - depth 5          :', np.round(hist(np.maximum(ref - u5, 0).max(-1)), 4), ' share>0 %.4f' % ((ref - u5).max(-1) > 0).mean())
for depth in (6, 7, 100):
    ud = r.to_u8(r.render(sc, 8, max_depth=depth))
    print('mode %s depth %3d - depth 5    :' % (mode, depth), np.round(hist(np.maximum(ud - u5, 0).max(-1)), 4), ' share>0 %.4f' % ((ud - u5).max(-1) > 0).mean())
