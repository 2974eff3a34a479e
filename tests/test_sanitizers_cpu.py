"""CPU-side hardening (no GPU sanitizers exist on this pool): the C oracle and the host side of the C ABI run under
AddressSanitizer + UndefinedBehaviorSanitizer.

* oracle/gls_oracle.c, oracle/held_karp.c, oracle/one_tree.c: rebuilt with gcc -fsanitize=address,undefined and driven through the
  golden-vector tests in a child interpreter (the sanitizer runtime has to be preloaded into python);
* gnngls_amd/csrc/capi.hip: compiled HOST-ONLY with the ROCm clang and the same sanitizers, linked with the fuzz driver
  tests/native/capi_fuzz.cpp (hostile arguments: B < 0, n > 65535, NULL pointers, undersized workspaces, bad enums).
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]


def test_oracle_under_asan_ubsan(tmp_path):
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isfile(libasan):
        pytest.skip("gcc has no libasan here")
    so = str(tmp_path / "libgls_oracle_san.so")
    hk = str(tmp_path / "libheld_karp_san.so")
    one = str(tmp_path / "libone_tree_san.so")
    flags = ["-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wextra"] + SAN
    subprocess.check_call(["gcc"] + flags + [os.path.join(ROOT, "oracle", "gls_oracle.c"), "-o", so, "-lm"])
    subprocess.check_call(["gcc"] + flags + [os.path.join(ROOT, "oracle", "held_karp.c"), "-o", hk])
    subprocess.check_call(["gcc"] + flags + [os.path.join(ROOT, "oracle", "one_tree.c"), "-o", one])
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", GLS_ORACLE_SO=so,
               HELD_KARP_SO=hk, ONE_TREE_SO=one, UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider",
                          os.path.join(ROOT, "tests", "test_oracle_golden.py"), os.path.join(ROOT, "tests", "test_held_karp_cpu.py"),
                          os.path.join(ROOT, "tests", "test_one_tree_cpu.py")],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "passed" in out.stdout and "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr


def test_c_abi_argument_checks_under_asan_ubsan(tmp_path):
    so_dir = os.path.join(ROOT, "gnngls_amd")
    if not os.path.isfile(os.path.join(so_dir, "libgnngls_hip.so")):
        pytest.skip("libgnngls_hip.so not built")
    obj, exe = str(tmp_path / "capi_san.o"), str(tmp_path / "capi_fuzz")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--cuda-host-only", "-std=c++17", "-c",
                           os.path.join(so_dir, "csrc", "capi.hip"), "-o", obj] + SAN)
    subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang++", "-std=c++17", os.path.join(ROOT, "tests", "native", "capi_fuzz.cpp"), obj,
                           "-L" + so_dir, "-lgnngls_hip", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + so_dir,
                           "-Wl,-rpath,/opt/rocm/lib", "-o", exe] + SAN)
    out = subprocess.run([exe, "60000"], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1",
                                  HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES=""))      # host-side checks only: no GPU
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "capi_fuzz: 60003 calls" in out.stdout and "runtime error" not in out.stderr
