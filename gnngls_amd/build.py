"""Builds libgnngls_hip.so (the C-ABI shared library of hand-written HIP kernels) for gfx950.

    python -m gnngls_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is kept in-tree (gnngls_amd/libgnngls_hip.so,
git-ignored) so that it travels with the repo snapshot to the GPU box.  Every translation unit is
compiled to an object of its own (gnngls_amd/build/*.o, git-ignored), stale ones in parallel, then
linked: the search kernels take ~90 s, the other three units seconds each.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
SO = os.path.join(HERE, "libgnngls_hip.so")
SOURCES = ["gls_kernels.hip", "model_kernels.hip", "train_kernels.hip", "capi.hip"]
# -ffp-contract=off: the guided matrix D + k*P (gnngls/algorithms.py:164) rounds twice and np.isclose
# (operators.py:42) is evaluated literally; a fused multiply-add would change move selection.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result"]
# headers a unit includes (a unit is rebuilt when it, one of these or the flags change)
HEADERS = {
    "gls_kernels.hip": ["gls_common.h", "gls_descent_scans.h", "gls_perturbation.h", "gls_kernels.h"],
    "model_kernels.hip": ["model_kernels.h"],
    "train_kernels.hip": ["train_kernels.h", "model_kernels.h"],
    "capi.hip": ["gls_kernels.h", "model_kernels.h", "train_kernels.h", os.path.join("..", "..", "include", "gnngls_hip.h")],
}


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.isfile(os.path.join(CSRC, s))]


def _extra():
    return os.environ.get("GNNGLS_EXTRA_FLAGS", "").split()


def _obj(src):
    return os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")


def _stamp():
    return " ".join(FLAGS + _extra())


def _unit_stale(src):
    o = _obj(src)
    if not os.path.isfile(o) or not os.path.isfile(o + ".flags"):
        return True
    with open(o + ".flags") as f:
        if f.read() != _stamp():
            return True
    t = os.path.getmtime(o)
    deps = [src] + [os.path.join(CSRC, h) for h in HEADERS.get(os.path.basename(src), [])]
    # any header of csrc/ the table above does not know about counts for every unit
    known = {os.path.normpath(os.path.join(CSRC, h)) for hs in HEADERS.values() for h in hs}
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC)
             if f.endswith(".h") and os.path.normpath(os.path.join(CSRC, f)) not in known]
    return any(os.path.getmtime(d) > t for d in deps if os.path.isfile(d))


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
    os.makedirs(OBJ, exist_ok=True)

    def compile_unit(src):
        cmd = [hipcc] + FLAGS + _extra() + ["-c", src, "-o", _obj(src)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=CSRC)
        with open(_obj(src) + ".flags", "w") as f:
            f.write(_stamp())

    stale = [s for s in sources() if force or _unit_stale(s)]
    with ThreadPoolExecutor(max_workers=max(1, len(stale))) as ex:
        list(ex.map(compile_unit, stale))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [_obj(s) for s in sources()] + ["-o", SO]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
