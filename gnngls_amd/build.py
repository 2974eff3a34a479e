"""Builds libgnngls_hip.so (the C-ABI shared library of hand-written HIP kernels) for gfx950.

    python -m gnngls_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is kept in-tree (gnngls_amd/libgnngls_hip.so,
git-ignored) so that it travels with the repo snapshot to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libgnngls_hip.so")
SOURCES = ["gls_kernels.hip", "model_kernels.hip", "train_kernels.hip", "capi.hip"]
# -ffp-contract=off: the guided matrix D + k*P (gnngls/algorithms.py:164) rounds twice and np.isclose
# (operators.py:42) is evaluated literally; a fused multiply-add would change move selection.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-Wno-unused-result"]


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.isfile(os.path.join(CSRC, s))]


def needs_build():
    if not os.path.isfile(SO):
        return True
    t = os.path.getmtime(SO)
    deps = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "gnngls_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps if os.path.isfile(d))


def build(force=False, verbose=False):
    if not force and not needs_build():
        return SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + os.environ.get("GNNGLS_EXTRA_FLAGS", "").split() + sources() + ["-o", SO]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
