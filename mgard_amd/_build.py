"""Builds the HIP C-ABI library (mgard_amd/libmgard_hip.so) for gfx950 with hipcc.

Cross-compiles without a GPU. -ffp-contract=off is part of the numerical contract: the
reference arithmetic is the non-FMA branch (MGARD_X_FMA is never defined upstream)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# MGARD_HIP_LIB: developer override to A/B two builds of the library in one session
LIB = os.environ.get("MGARD_HIP_LIB", os.path.join(HERE, "libmgard_hip.so"))
SOURCES = ["capi.hip", "highlevel.hip"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
               "-shared", "-Wall", "-ldl"]


def _deps():
    out = [os.path.join(HERE, "..", "include", "mgard_hip.h"),
           os.path.join(HERE, "..", "include", "mgard_hip_compress.h")]
    for f in os.listdir(CSRC):
        if f.endswith((".hip", ".hpp", ".h", ".cpp")):
            out.append(os.path.join(CSRC, f))
    return out


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in _deps())


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + HIPCC_FLAGS + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB
