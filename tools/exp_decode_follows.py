"""mgh_decompress from host memory with the decoder following the arriving record (MGH_HL_DECODE_FOLLOWS=1)
against copy-then-decode (=0), alternating inside one process; pageable and registered buffers. Dev tool."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgard_amd as mg
from mgard_amd import highlevel as hl
from tests.util import smooth_field
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
u = smooth_field((n, n, n), np.float32)
c = hl.compress(u, 1e-3, np.inf, mg.REL).copy()
back = np.zeros_like(u)
def t(reps=6):
    hl.decompress(c, out=back)
    t0 = time.perf_counter()
    for _ in range(reps): hl.decompress(c, out=back)
    return (time.perf_counter() - t0) / reps * 1e3
for kind in ("pageable", "registered"):
    if kind == "registered":
        hl.pin(c); hl.pin(back)
    res = {"0": [], "1": []}
    for rnd in range(4):
        for f in ("1", "0"):
            os.environ["MGH_HL_DECODE_FOLLOWS"] = f
            res[f].append(t())
    print(kind, "follows: %s   copy-then-decode: %s" % (" ".join("%.2f" % x for x in res["1"]), " ".join("%.2f" % x for x in res["0"])))
hl.unpin(c); hl.unpin(back)
