"""Builds the HIP C-ABI library (mgard_amd/libmgard_hip.so) for gfx950 with hipcc.

Cross-compiles without a GPU. -ffp-contract=off is part of the numerical contract: the
reference arithmetic is the non-FMA branch (MGARD_X_FMA is never defined upstream).

The two translation units are compiled to objects side by side (an object is rebuilt only when
the source or a header it includes, directly or not, is newer) and linked into the shared
library."""
import os
import re
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJDIR = os.path.join(HERE, "_obj")
# MGARD_HIP_LIB: developer override to A/B two builds of the library in one session
LIB = os.environ.get("MGARD_HIP_LIB", os.path.join(HERE, "libmgard_hip.so"))
SOURCES = ["capi.hip", "highlevel.hip"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall"]
LINK_FLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-ldl"]

_INC = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)


def _closure(path, seen=None):
    """The file and every quoted include it reaches."""
    seen = set() if seen is None else seen
    path = os.path.normpath(path)
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    with open(path, errors="replace") as f:
        for inc in _INC.findall(f.read()):
            _closure(os.path.join(os.path.dirname(path), inc), seen)
    return seen


def _obj(src):
    return os.path.join(OBJDIR, os.path.splitext(src)[0] + ".o")


def _stale(src):
    o = _obj(src)
    if not os.path.exists(o):
        return True
    t = os.path.getmtime(o)
    deps = _closure(os.path.join(CSRC, src)) | {os.path.abspath(__file__)}
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    if any(_stale(s) for s in SOURCES):
        return True
    return any(os.path.getmtime(_obj(s)) > t for s in SOURCES)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(OBJDIR, exist_ok=True)
    procs = []
    for s in SOURCES:
        if force or _stale(s):
            cmd = [hipcc] + HIPCC_FLAGS + ["-c", os.path.join(CSRC, s), "-o", _obj(s)]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [hipcc] + LINK_FLAGS + ["-o", LIB] + [_obj(s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB
