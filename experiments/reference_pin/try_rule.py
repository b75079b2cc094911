import os, sys, numpy as np
mode, start = sys.argv[1], sys.argv[2]
os.environ['ORC_RR_MODE'] = mode; os.environ['ORC_RR_START'] = start
HERE = os.path.dirname(os.path.abspath(__file__)); REPO = os.environ.get('REPO', '/root/repo')   # run from the scratch copy (README.md)
sys.path.insert(0, HERE); sys.path.insert(1, REPO); sys.path.insert(2, os.path.join(REPO, 'tools'))
from oracle import pyoracle
assert pyoracle.__file__.startswith(HERE), pyoracle.__file__
import recover_cornell_docs as r
assert r.oracle is pyoracle
sc = r.build(r.FROZEN_TRIS, r.FROZEN_LIGHT, quads=r.FROZEN_QUADS)
unsat = r.U8[8].max(-1) < 250
for depth in [int(a) for a in sys.argv[3:]]:
    img = r.render(sc, 8, max_depth=depth); s = r.to_u8(img) - r.U8[8]
    print('mode', mode, 'start', start, 'depth', depth, r.byte_stats(img), 'signed %.3f' % s[unsat].mean(), 'ours/ref %.5f' % (np.minimum(img,1)[unsat].mean() / r.R8[unsat].mean()), flush=True)
