"""Build libqbm_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m qbm_image_classification_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels with the tree."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libqbm_hip.so")
SOURCES = ["rbm_kernels.hip", "qubo_kernels.hip", "data_kernels.hip", "coreset_kernels.hip"]
HEADERS = ["qbm_device.hip.h", "strip_mma.hip.h", "qbm_host.hpp", os.path.join("..", "..", "include", "qbm_hip.h")]
# -ffp-contract=off: the numerical contract places every FMA explicitly
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


def _stale() -> bool:
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = [hipcc] + FLAGS + srcs + ["-o", SO]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
