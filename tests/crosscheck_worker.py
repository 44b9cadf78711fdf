"""Worker of test_gpu_highlevel.py::test_alternative_kernels_give_the_same_result: compress and
decompress a few arrays and print digests; the parent runs it with and without the developer
switches that select the older / simpler kernels (they are read once per process)."""
import hashlib
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tests/", 1)[0])
import mgard_amd as mg  # noqa: E402
from mgard_amd import highlevel as hl  # noqa: E402
from tests.util import inside_field, smooth_field  # noqa: E402

out = []
# (the last 3-D shape is big enough for the long-march class of the level kernel: 4 x 64 tiles,
# chunks of 16 coarse planes; the 4-D one runs the slice-by-slice path in both directions)
for shape, dt, tol in [((65, 97, 130), np.float32, 1e-2), ((40, 129, 66), np.float64, 1e-3), ((300, 70), np.float32, 1e-2),
                       ((257, 260, 300), np.float32, 1e-3), ((6, 66, 70, 129), np.float32, 1e-3)]:
    u = smooth_field(shape, dt)
    buf = hl.compress(u, tol, np.inf, mg.REL)
    v = hl.decompress(buf)
    out.append(str(len(buf)))  # (the bytes themselves vary: the outlier list is in atomic order)
    out.append(hashlib.sha256(v.tobytes()).hexdigest()[:16])
# round 6: a short fastest extent (64 x 4 tiles), many t-slices (batched way back), long strided pencils
# (chunked solves), D = 5 (row kernels, device-resident norm) -- on fields that compress
for shape, dt in [((70, 300, 9), np.float32), ((20, 40, 40, 40), np.float32), ((2500, 40, 48), np.float64),
                  ((5, 5, 20, 20, 40), np.float32), ((140, 260, 300), np.float32)]:
    u = inside_field(shape, dt)
    buf = hl.compress(u, 1e-3, np.inf, mg.REL)
    v = hl.decompress(buf)
    out.append(str(len(buf)))
    out.append(hashlib.sha256(v.tobytes()).hexdigest()[:16])
print("DIGESTS " + " ".join(out))
