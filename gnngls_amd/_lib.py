"""ctypes binding of libgnngls_hip.so (C ABI declared in include/gnngls_hip.h).

The product path has no CPU fallback: if the HIP library is missing or a call fails, this raises.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# GNNGLS_HIP_SO: another build of the same library (A/B kernel experiments, diagnostic -DGLS_STAMPS builds)
SO = os.environ.get("GNNGLS_HIP_SO") or os.path.join(HERE, "libgnngls_hip.so")

_vp = ctypes.c_void_p
_int = ctypes.c_int
_i64 = ctypes.c_int64
_f64 = ctypes.c_double
_f32 = ctypes.c_float

# name -> argtypes; every function returns int (0 = ok) unless listed in _RESTYPES
SIGNATURES = {
    "gnngls_abi_version": [],
    "gnngls_last_error": [],
    "gnngls_gls_resident_capacity": [_int],
    "gnngls_gls_describe_config": [_int, _int, _int, _vp, _vp, _vp, _vp],
    "gnngls_gls_describe_run": [_int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "gnngls_gls_kernel_resources": [_int, _int, _int, _int, _int, _vp, _vp],
    "gnngls_two_opt_delta_all": [_vp, _vp, _int, _int, _vp, _vp],
    "gnngls_relocate_delta_all": [_vp, _vp, _int, _int, _vp, _vp],
    "gnngls_best_move": [_vp, _vp, _int, _int, _int, _vp, _int, _vp, _vp, _vp, _vp],
    "gnngls_tour_cost": [_vp, _vp, _int, _int, _vp, _vp],
    "gnngls_nearest_neighbor": [_vp, _int, _int, _int, _vp, _vp],
    "gnngls_gls_run": [_vp, _vp, _int, _int, _int, _vp, _vp, _int, _int, _int, _i64, _f64, _f64,
                       _vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _vp, _vp],
    "gnngls_model_packed_floats": [_int, _int],
    "gnngls_regret_forward_workspace_bytes": [_int, _int],
    "gnngls_regret_forward": [_vp, _vp, _int, _int, _int, _int, _vp, _vp, _i64, _vp],
    "gnngls_regret_prepared_bytes": [_int],
    "gnngls_regret_prepare": [_vp, _int, _int, _vp, _i64, _vp],
    "gnngls_regret_forward_prepared": [_vp, _vp, _vp, _i64, _int, _int, _int, _int, _vp, _vp, _i64, _vp],
    "gnngls_regret_train_workspace_bytes": [_int, _int, _int],
    "gnngls_regret_train_forward": [_vp, _vp, _int, _int, _int, _int, _f32, _vp, _vp, _vp, _i64, _vp],
    "gnngls_regret_train_backward": [_vp, _vp, _vp, _int, _int, _int, _int, _vp, _vp, _i64, _vp],
    "gnngls_pack_features": [_vp, _int, _int, _f64, _f64, _vp, _vp],
    "gnngls_unpack_regret": [_vp, _int, _int, _f64, _f64, _vp, _vp],
    "gnngls_debug_set_penalty16_limit": [_int],
    "gnngls_debug_set_stamp_buffer": [_vp],
    "gnngls_debug_set_gls_threads": [_int],
    "gnngls_debug_set_gls_team": [_int],
    "gnngls_debug_set_gls_prune": [_int],
    "gnngls_gls_uses_team": [_int, _int, _int],
    "gnngls_gls_waves_per_simd": [_int, _int, _int],
    "gnngls_profile_enable": [_int],
    "gnngls_profile_collect": [_vp, _vp],
    "gnngls_profile_set_executed_evals": [_vp],
}

PROF_KINDS = ["pack_features", "embed", "gemm_fc", "gat_rows", "gat_rows_rank1", "gemm_ffn1(unused)", "gemm_ffn2(unused)",
              "decision", "unpack_regret", "nearest_neighbor", "tour_cost", "gls", "ffn_fused",
              "train_colsum", "train_elementwise", "train_gemm_bwd", "train_gemm_tn", "train_gat_bwd"]


def profile_enable(on=True):
    check(load().gnngls_profile_enable(int(on)), "profile_enable")


def profile_collect():
    """-> {kind: (milliseconds, launches)} since profile_enable(True)."""
    n = len(PROF_KINDS)
    ms = (ctypes.c_double * n)()
    cnt = (ctypes.c_int64 * n)()
    check(load().gnngls_profile_collect(ctypes.cast(ms, ctypes.c_void_p), ctypes.cast(cnt, ctypes.c_void_p)),
          "profile_collect")
    return {k: (ms[i], cnt[i]) for i, k in enumerate(PROF_KINDS)}
_RESTYPES = {"gnngls_last_error": ctypes.c_char_p, "gnngls_model_packed_floats": ctypes.c_int64,
             "gnngls_regret_forward_workspace_bytes": ctypes.c_int64,
             "gnngls_regret_prepared_bytes": ctypes.c_int64,
             "gnngls_regret_train_workspace_bytes": ctypes.c_int64}

_ABI4 = ("gnngls_regret_prepared_bytes", "gnngls_regret_prepare", "gnngls_regret_forward_prepared")
_lib = None


class GnnglsHipError(RuntimeError):
    pass


def load():
    """Loads the shared library (building nothing: run `python -m gnngls_amd.build` or
    __graft_entry__.build() first).  Raises if it is absent -- there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.isfile(SO):
            raise GnnglsHipError(
                f"{SO} not found: build the HIP extension with `python -m gnngls_amd.build` "
                "(this package has no CPU fallback)")
        # PyTorch-ROCm bundles its own libamdhip64.so.7 (same SONAME as /opt/rocm's).  The process must
        # hold exactly ONE HIP runtime, and it has to be the one torch's streams/allocations live in:
        # import torch first so the dynamic linker binds this library to the runtime torch loaded.
        import torch  # noqa: F401
        L = ctypes.CDLL(SO)
        for name, argtypes in SIGNATURES.items():
            if name in _ABI4 and "GNNGLS_HIP_SO" in os.environ and not hasattr(L, name):
                continue              # an older build of the library named by GNNGLS_HIP_SO (same-box A/B of kernel variants)
            f = getattr(L, name)      # AttributeError if the symbol is missing
            f.argtypes = argtypes
            f.restype = _RESTYPES.get(name, ctypes.c_int)
        _lib = L
    return _lib


def check(code, what=""):
    if code != 0:
        msg = load().gnngls_last_error()
        raise GnnglsHipError(f"{what} failed ({code}): {msg.decode() if msg else ''}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device-resident contiguous tensor required"
    return ctypes.c_void_p(t.data_ptr())


def current_stream():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
