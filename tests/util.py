"""Shared helpers for the tests: seeded synthetic fields."""
import numpy as np


def smooth_field(shape, dtype=np.float32, seed=20260101, noise=1e-3):
    """u = sin(2*pi*3x) cos(2*pi*2y) + 0.5 sin(2*pi*5z) + noise*xi on the unit cube; for other
    dimensionalities the same recipe is applied to the trailing dims that exist."""
    rng = np.random.default_rng(seed)
    D = len(shape)
    ax = [np.arange(n, dtype=np.float64) / max(n - 1, 1) for n in shape]
    grids = np.meshgrid(*ax, indexing="ij", sparse=True)
    f = [3.0, 2.0, 5.0, 1.0, 4.0]
    u = np.zeros(shape, dtype=np.float64)
    u = u + np.sin(2 * np.pi * f[0] * grids[D - 1])
    if D >= 2:
        u = u * np.cos(2 * np.pi * f[1] * grids[D - 2])
    if D >= 3:
        u = u + 0.5 * np.sin(2 * np.pi * f[2] * grids[D - 3])
    for d in range(D - 3):
        u = u + 0.25 * np.cos(2 * np.pi * f[3 + (d % 2)] * grids[d])
    u = u + noise * rng.uniform(-1, 1, size=shape)
    return np.ascontiguousarray(u.astype(dtype))


def nonuniform_coords(shape, dtype=np.float64, seed=7):
    """x_i = (i + 0.3*zeta_i)/(n-1), zeta ~ U(-1,1): strictly increasing."""
    rng = np.random.default_rng(seed)
    out = []
    for n in shape:
        z = rng.uniform(-1, 1, size=n)
        out.append(((np.arange(n) + 0.3 * z) / (n - 1)).astype(dtype))
    return out


def inside_field(shape, dtype=np.float32, seed=1):
    """A field whose quantized coefficients stay inside an 8192-entry dictionary at 1e-3 in 4-D / 5-D and on
    short extents too (the quantizer's bins shrink with 1 + 3^D; smooth_field puts whole periods across
    extents of a few nodes): one period of a sine across every extent of 32 and more points, 1 % of that
    across the short ones, 1e-3 uniform noise."""
    ax = np.meshgrid(*[np.arange(n, dtype=np.float64) / max(n - 1, 1) for n in shape], indexing="ij", sparse=True)
    g = sum((1.0 if shape[k] >= 32 else 0.01) * np.sin(2 * np.pi * a + 0.3 * k) for k, a in enumerate(ax))
    g = g + 1e-3 * np.random.default_rng(seed).uniform(-1, 1, size=shape)
    return np.ascontiguousarray(g.astype(dtype))

