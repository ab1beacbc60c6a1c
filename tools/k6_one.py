#!/usr/bin/env python3
"""One pool size, repeated: for counter passes.  python tools/k6_one.py K"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from massivedatans_amd.clustering import neighbors as nb
K = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
rng = np.random.RandomState(2)
pts = rng.uniform(size=(K, 3))
np.random.seed(K)
masks = nb.draw_bootstrap_masks(K, 10)
s = nb.MemberSet(pts)
for _ in range(30):
    r = s.bootstrap_radius_packed(masks, 10)
print(K, r)
