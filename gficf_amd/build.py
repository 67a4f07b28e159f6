"""In-tree build of libgficf_hip.so with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")


def build_extension(force: bool = False, jobs: int = 3) -> str:
    cmd = ["make", "-C", CSRC, f"-j{jobs}"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    out = os.path.join(_HERE, "libgficf_hip.so")
    if not os.path.exists(out):
        raise RuntimeError("hipcc build did not produce libgficf_hip.so")
    return out
