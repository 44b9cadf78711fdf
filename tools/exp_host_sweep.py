"""Host-buffer mgh_compress / mgh_decompress (pageable, touched buffers) at 512^3 f32 for one setting of
MGH_HL_COPY_THREADS / MGH_HL_RING_MB (read when the pool / ring are first used). Dev tool."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgard_amd as mg
from mgard_amd import highlevel as hl
from tests.util import smooth_field
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
u = smooth_field((n, n, n), np.float32)
cbuf = np.zeros(u.nbytes + 1000000, np.uint8); back = np.zeros_like(u)
def best(f, reps=4, batches=3):
    f(); ts = []
    for _ in range(batches):
        t0 = time.perf_counter()
        for _ in range(reps): r = f()
        ts.append((time.perf_counter() - t0) / reps * 1e3)
    return min(ts), r
c_ms, s = best(lambda: hl.compress(u, 1e-3, np.inf, mg.REL, out=cbuf))
x_ms, v = best(lambda: hl.decompress(s, out=back))
print("threads %s ring %s MB: compress %.2f ms  decompress %.2f ms" % (os.environ.get("MGH_HL_COPY_THREADS", "5"), os.environ.get("MGH_HL_RING_MB", "16"), c_ms, x_ms))
