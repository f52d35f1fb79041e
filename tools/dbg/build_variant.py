"""Build a variant of libqbm_hip.so with extra -D flags into tools/microbench/bin/<name>.so (for QBM_HIP_LIB=... A/B runs).
    python tools/dbg/build_variant.py NAME -DQBM_EXP_X=1 ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from qbm_image_classification_amd.build import CSRC, FLAGS, SOURCES
out = os.path.join(ROOT, "tools", "microbench", "bin", sys.argv[1] + ".so")
os.makedirs(os.path.dirname(out), exist_ok=True)
cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + FLAGS + sys.argv[2:] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", out]
subprocess.check_call(cmd)
print(out)
