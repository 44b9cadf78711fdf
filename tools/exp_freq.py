"""Dump the symbol histogram of the 512^3 bench field (input of the Huffman code construction)."""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import mgard_amd
from tests.util import smooth_field
shape = (512, 512, 512)
u = smooth_field(shape, np.float32)
d = torch.from_numpy(u).cuda()
h = mgard_amd.Hierarchy(shape, np.float32)
sym = h.decompose_quantize_sym16(d, mgard_amd.REL, 1e-3, float('inf'))[0]
f = torch.bincount(sym.view(-1).to(torch.int64), minlength=8192).cpu().numpy().astype(np.uint32)
np.save('gpurun_out/freq_512.npy', f)
print("nonzero bins", int((f > 0).sum()), "max", int(f.max()))
