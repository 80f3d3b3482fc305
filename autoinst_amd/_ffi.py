"""ctypes binding of ``libautoinst_hip.so`` (C ABI in ``include/autoinst_hip.h``).

There is no CPU fallback: if the library is missing or cannot be loaded the import of the
compute entry points raises, and every compute call needs a gfx950 device.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AUTOINST_HIP_LIB") or os.path.join(_HERE, "libautoinst_hip.so")   # the override is for A/B runs of two builds

AI_OK = 0
AI_MEM_HOST, AI_MEM_DEVICE = 0, 1
NUM_CUTS = 10

# every symbol include/autoinst_hip.h declares (tests check the library exports each one)
SYMBOLS = (
    "ai_version", "ai_last_error", "ai_ctx_create", "ai_ctx_destroy", "ai_affinity_build",
    "ai_csr_from_host", "ai_csr_dims", "ai_csr_export", "ai_csr_free", "ai_ncut", "ai_fiedler",
    "ai_sweep", "ai_lsym_apply", "ai_bench_spmv", "ai_eigs_smallest",
    "ai_radius_mean_pool", "ai_nn1_project", "ai_ncut_batch",
    "ai_label_pairs", "ai_merge_associate", "ai_unique_points", "ai_affinity_build_sam", "ai_affinity_apply_camera",
    "ai_ctx_mem_info", "ai_abi_version", "ai_abi_sizeof", "ai_bench_copy",
)
ABI_VERSION = 6   # AI_ABI_VERSION of the header this binding was written against


class NcutOpts(C.Structure):
    _fields_ = [("tol", C.c_double), ("max_iter", C.c_int32), ("check_every", C.c_int32), ("reserved", C.c_int32), ("window_rows", C.c_int64)]


class NcutStats(C.Structure):
    _fields_ = [
        ("levels", C.c_int64), ("lanczos_solves", C.c_int64), ("null_solves", C.c_int64),
        ("lanczos_steps", C.c_int64), ("spmv_rows", C.c_int64), ("spmv_nnz", C.c_int64),
        ("unconverged", C.c_int64), ("n_groups", C.c_int64),
        ("ms_total", C.c_double), ("ms_eigen", C.c_double), ("ms_spmv", C.c_double),
        ("ms_sweep", C.c_double), ("ms_rebuild", C.c_double), ("max_resid", C.c_double),
        ("restarted_solves", C.c_int64), ("hist_retries", C.c_int64),
        ("max_true_resid", C.c_double), ("true_resid_limit", C.c_double),
        ("accepted_above_limit", C.c_int64), ("check_timeouts", C.c_int64),
        ("spmv_blocks", C.c_int64), ("spmv_blocks_idle", C.c_int64),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class AutoinstHipError(RuntimeError):
    pass


class NoConvergence(AutoinstHipError):
    """A Lanczos solve hit max_iter (status -3); the reference raises scipy's ArpackNoConvergence there."""


_lib = None


def load():
    """Load the shared library (once).  Raises ``AutoinstHipError`` when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AutoinstHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C autoinst_amd/csrc`).  autoinst_amd has no CPU fallback.")
    # torch ships its own HIP runtime; a process that uses both must let torch initialise its runtime FIRST (the library then
    # resolves to the runtime already loaded).  The other order leaves torch without devices ("No HIP GPUs are available").
    # So: if torch is already imported, make it initialise now; a program that imports torch later must do so before this call.
    import sys
    if "torch" in sys.modules:
        try:
            t = sys.modules["torch"]
            if t.cuda.is_available():
                t.cuda.init()
        except Exception:  # noqa: BLE001 -- torch without a device: the library reports its own error below
            pass
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # e.g. the ROCm runtime is absent
        raise AutoinstHipError(f"cannot load {LIB_PATH}: {e}") from e
    vp, i64, i32, dbl = C.c_void_p, C.c_int64, C.c_int32, C.c_double
    P = C.POINTER
    lib.ai_version.restype = C.c_int
    lib.ai_last_error.restype = C.c_char_p
    lib.ai_ctx_create.argtypes = [C.c_int, P(vp)]
    lib.ai_ctx_destroy.argtypes = [vp]
    lib.ai_ctx_mem_info.argtypes = [vp, P(i64)]
    lib.ai_affinity_build.argtypes = [vp, vp, i64, vp, i32, vp, i32, dbl, dbl, dbl, dbl, C.c_int, P(vp)]
    lib.ai_affinity_build_sam.argtypes = [vp, vp, i64, vp, i32, vp, i32, vp, i32, dbl, dbl, dbl, dbl, dbl, C.c_int, P(vp)]
    lib.ai_affinity_apply_camera.argtypes = [vp, vp, vp, i32, vp, i32, dbl, dbl, C.c_int]
    lib.ai_csr_from_host.argtypes = [vp, i64, vp, vp, vp, P(vp)]
    lib.ai_csr_dims.argtypes = [vp, P(i64), P(i64)]
    lib.ai_csr_export.argtypes = [vp, vp, vp, vp, vp]
    lib.ai_csr_free.argtypes = [vp, vp]
    lib.ai_ncut.argtypes = [vp, vp, i64, dbl, dbl, P(NcutOpts), vp, P(i32), P(NcutStats)]
    lib.ai_fiedler.argtypes = [vp, vp, P(NcutOpts), P(dbl), vp, P(i32), P(dbl)]
    lib.ai_sweep.argtypes = [vp, vp, vp, vp, vp, P(dbl)]
    lib.ai_lsym_apply.argtypes = [vp, vp, vp, vp]
    lib.ai_bench_spmv.argtypes = [vp, vp, i32, P(dbl), P(dbl)]
    lib.ai_eigs_smallest.argtypes = [vp, vp, i32, P(NcutOpts), vp, vp, P(i32), P(dbl)]
    lib.ai_radius_mean_pool.argtypes = [vp, vp, i64, vp, i64, vp, i32, dbl, C.c_int, vp, vp]
    lib.ai_nn1_project.argtypes = [vp, vp, i64, vp, i64, C.c_int, vp, vp]
    lib.ai_ncut_batch.argtypes = [vp, P(vp), i32, P(i64), dbl, dbl, P(NcutOpts), P(vp), P(i32), P(NcutStats)]
    lib.ai_label_pairs.argtypes = [vp, vp, vp, i64, C.c_int, i64, vp, vp, vp, P(i64)]
    lib.ai_merge_associate.argtypes = [vp, vp, vp, i64, vp, vp, i64, vp, dbl, i32, i32, C.c_int, vp, vp, vp, vp, vp]
    lib.ai_unique_points.argtypes = [vp, vp, i64, C.c_int, vp, P(i64)]
    if os.environ.get("AUTOINST_HIP_LIB") and not hasattr(lib, "ai_abi_version"):
        # an older build selected for an A/B run (tools/): no ABI check, and the entry points it lacks stay unbound.  Only on request:
        # a stale build picked up by accident must fail here with "rebuild", not later with an AttributeError or a wrong struct layout
        if os.environ.get("AUTOINST_HIP_ALLOW_OLD_ABI") != "1":
            raise AutoinstHipError(
                f"{LIB_PATH} (selected through AUTOINST_HIP_LIB) has no ai_abi_version: it predates ABI 4.  Rebuild it, or set "
                "AUTOINST_HIP_ALLOW_OLD_ABI=1 to load it unchecked for an A/B run (stats fields it lacks read as garbage)")
        sys.stderr.write(f"autoinst_amd: WARNING: {LIB_PATH} loaded WITHOUT the ABI check (AUTOINST_HIP_ALLOW_OLD_ABI=1)\n")
        for name in SYMBOLS:
            if hasattr(lib, name) and name not in ("ai_version", "ai_last_error"):
                getattr(lib, name).restype = C.c_int
        _lib = lib
        return lib
    lib.ai_abi_sizeof.argtypes = [C.c_int]
    lib.ai_bench_copy.argtypes = [vp, i64, i32, P(dbl)]
    for name in SYMBOLS:
        if name not in ("ai_version", "ai_last_error", "ai_abi_sizeof"):
            getattr(lib, name).restype = C.c_int
    lib.ai_abi_sizeof.restype = i64
    # a library built from another header would read past (or short of) the structs this binding passes
    if lib.ai_abi_version() != ABI_VERSION or lib.ai_abi_sizeof(0) != C.sizeof(NcutOpts) or lib.ai_abi_sizeof(1) != C.sizeof(NcutStats):
        raise AutoinstHipError(
            f"{LIB_PATH}: ABI version {lib.ai_abi_version()} / struct sizes {lib.ai_abi_sizeof(0)}, {lib.ai_abi_sizeof(1)} do not match this "
            f"binding ({ABI_VERSION} / {C.sizeof(NcutOpts)}, {C.sizeof(NcutStats)}): rebuild the library (make -C autoinst_amd/csrc)")
    _lib = lib
    return lib


def check(status: int, what: str):
    if status == AI_OK:
        return
    msg = load().ai_last_error().decode("utf-8", "replace")
    if status == -1:
        raise ValueError(f"{what}: {msg}")
    if status == -2:
        raise MemoryError(f"{what}: {msg}")
    if status == -3:
        raise NoConvergence(f"{what}: {msg}")
    raise AutoinstHipError(f"{what} failed (status {status}): {msg}")
