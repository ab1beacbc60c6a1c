"""cProfile of an end-to-end GPU run: python tools/e2e_profile.py <horns|nothing> <ndata> <nlive> <cap>"""
import cProfile, os, pstats, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from massivedatans_amd import gen, sample

kind, ndata, nlive, cap = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
data = (gen.horns if kind == "horns" else gen.nothing)(ndata)

def go():
    with np.errstate(all="ignore"):
        return sample.run(data["x"], data["y"], nlive_points=nlive, max_samples=cap, use_graph=False)

cProfile.run("res = go()", "/tmp/e2e.prof")
pstats.Stats("/tmp/e2e.prof").sort_stats("tottime").print_stats(28)
