import json, sys, os
sys.path.insert(0, "/root/repo"); os.chdir(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from massivedatans_amd import _lib
lib = _lib.require_device()
out = bench.muse_leg(lib, _lib)
print(json.dumps({k: {kk: out[k][kk] for kk in ("kernel", "launch_us", "achieved", "frac")} for k in ("b1", "b64", "b64_filter")}))
