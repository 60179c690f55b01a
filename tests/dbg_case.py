import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # debug helper, run from the repo root
import numpy as np, oracle
from csnappy_amd import api
P = oracle.best()
urls = open('tests/golden/urls.10K','rb').read()
bad = 0
for (blk, p, mode) in ((32768,15,1),(65536,16,0),(4096,13,1)):
    for k in range(0, len(urls), blk):
        x = urls[k:k+blk]
        got = api.compress_fragment(x, p) if mode else api.compress(x, p)
        want = P.compress_fragment(np.frombuffer(x,np.uint8), p) if mode else P.compress(np.frombuffer(x,np.uint8), p)
        if got != want:
            bad += 1
            if bad <= 3:
                n = min(len(got), len(want))
                d = next((i for i in range(n) if got[i] != want[i]), n)
                print('MISMATCH block', k//blk, 'blk', blk, 'p', p, 'len got/want', len(got), len(want), 'first diff at', d)
                print(' got ', got[max(0,d-8):d+24].hex()); print(' want', want[max(0,d-8):d+24].hex())
print('bad', bad)
