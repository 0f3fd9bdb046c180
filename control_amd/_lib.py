"""ctypes binding of ``libkkt.so`` (``include/kkt.h``).  No PyTorch, no fallbacks.

The library is built in-tree by ``make -C control_amd/csrc`` (``__graft_entry__.build``).
Loading fails loudly when it is missing; creating a system fails loudly when there is no
GPU -- the product has no CPU path.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (KKT_LIB: another build of the same library, for same-box A/B measurements of a kernel change)
LIB_PATH = os.environ.get("KKT_LIB") or os.path.join(_HERE, "libkkt.so")

c_i32p = C.POINTER(C.c_int32)
c_f64p = C.POINTER(C.c_double)


class KktError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libkkt error {code}: {msg}")
        self.code = code


class PcDesc(C.Structure):
    _fields_ = [("kind", C.c_int), ("n_t", C.c_int), ("tau", C.c_double),
                ("beta", C.c_double), ("epsilon", C.c_double), ("nx", C.c_int64),
                ("m_indptr", c_i32p), ("m_indices", c_i32p), ("m_values", c_f64p),
                ("n_bc", C.c_int64), ("bc_idx", c_i32p),
                ("mass_its", C.c_int), ("mass_emin", C.c_double), ("mass_emax", C.c_double),
                ("schur_its", C.c_int), ("schur_emin", C.c_double),
                ("schur_emax", C.c_double), ("schur_eimag", C.c_double),
                ("coarse_cycles", C.c_int), ("n_coarse", C.c_int64),
                ("p_indptr", c_i32p), ("p_indices", c_i32p), ("p_values", c_f64p)]


class PcStokesDesc(C.Structure):
    _fields_ = [("n_p_blocks", C.c_int), ("cn", C.c_int), ("nv", C.c_int64), ("np", C.c_int64),
                ("b_scale", C.c_double), ("post_scale", C.c_double),
                ("b_indptr", c_i32p), ("b_indices", c_i32p), ("b_values", c_f64p),
                ("kp_indptr", c_i32p), ("kp_indices", c_i32p), ("kp_values", c_f64p),
                ("mp_indptr", c_i32p), ("mp_indices", c_i32p), ("mp_values", c_f64p),
                ("kp_its", C.c_int), ("kp_emin", C.c_double), ("kp_emax", C.c_double),
                ("mp_its", C.c_int), ("mp_emin", C.c_double), ("mp_emax", C.c_double),
                ("kp_coarse_cycles", C.c_int), ("kp_n_coarse", C.c_int64),
                ("kp_p_indptr", c_i32p), ("kp_p_indices", c_i32p), ("kp_p_values", c_f64p)]


class Info(C.Structure):
    _fields_ = [(n, C.c_int64) for n in
                ("n_local", "n_blocks_stored", "n_value_arrays", "n_patterns",
                 "nnz_blocks", "rows_blocks", "bytes_algorithmic",
                 "bytes_device_values", "bytes_device_index", "bytes_streamed")] + \
               [("last_solve_ms", C.c_double), ("last_pc_applies", C.c_int64),
                ("last_op_applies", C.c_int64), ("program_fallbacks", C.c_int64)] + \
               [(n, C.c_int64) for n in
                ("sweep_form", "sweep_tiles", "sweep_threads", "sweep_depth", "sweep_row_slots",
                 "sweep_its", "apply_launches", "apply_switched")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class StageTimes(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("operator_ms", "pc_ms", "orth_ms", "allreduce_ms",
                                          "other_ms", "total_ms")] + \
               [(n, C.c_int64) for n in ("iterations", "operator_applies", "pc_applies")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class PcStageTimes(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("sweeps_ms", "batched_ms", "comm_ms", "total_ms")] + \
               [(n, C.c_int64) for n in ("sweep_launches", "sweep_phases", "batched_launches",
                                         "comm_steps")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class StepLock(C.Structure):
    _fields_ = [("n_steps", C.c_int), ("restart", C.c_int), ("V", c_f64p), ("h", c_f64p),
                ("v_next", c_f64p)]


PC_CALLBACK = C.CFUNCTYPE(C.c_int, C.c_void_p, c_f64p, c_f64p, c_f64p, c_f64p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, c_f64p, C.c_int, C.c_int)
SENDRECV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, c_f64p, C.c_int64, C.c_int,
                          c_f64p, C.c_int64, C.c_int)

# name -> (restype, argtypes); every symbol include/kkt.h declares
SIGNATURES = {
    "kkt_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "kkt_destroy": (C.c_int, [C.c_void_p]),
    "kkt_last_error": (C.c_char_p, [C.c_void_p]),
    "kkt_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p]),
    "kkt_set_tile_coordinates": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, c_f64p]),
    "kkt_set_layout": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                 C.c_int, C.c_int, C.c_int]),
    "kkt_set_shard": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "kkt_set_shard_families": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "kkt_shard_range": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int),
                                  C.POINTER(C.c_int)]),
    "kkt_add_block": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64,
                                C.c_int64, c_i32p, c_i32p, c_f64p, C.c_int64]),
    "kkt_update_block_values": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, c_f64p]),
    "kkt_set_bc": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, c_i32p, C.c_double]),
    "kkt_set_const_nullspace": (C.c_int, [C.c_void_p, C.c_int, C.c_double]),
    "kkt_finalize": (C.c_int, [C.c_void_p]),
    "kkt_set_pc_schur": (C.c_int, [C.c_void_p, C.POINTER(PcDesc)]),
    "kkt_set_pc_stokes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.POINTER(PcStokesDesc)]),
    "kkt_set_pc_callback": (C.c_int, [C.c_void_p, PC_CALLBACK, C.c_void_p]),
    "kkt_set_pc_identity": (C.c_int, [C.c_void_p]),
    "kkt_set_krylov": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double,
                                 C.c_double, C.c_double, C.c_int]),
    "kkt_apply": (C.c_int, [C.c_void_p, c_f64p, c_f64p]),
    "kkt_pc_apply": (C.c_int, [C.c_void_p, c_f64p, c_f64p]),
    "kkt_solve": (C.c_int, [C.c_void_p, c_f64p, c_f64p, C.POINTER(C.c_int),
                            C.POINTER(C.c_int), c_f64p, c_f64p, C.c_int,
                            C.POINTER(C.c_int)]),
    "kkt_local_size": (C.c_int64, [C.c_void_p]),
    "kkt_vec_alloc": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "kkt_vec_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "kkt_vec_upload": (C.c_int, [C.c_void_p, C.c_void_p, c_f64p]),
    "kkt_vec_download": (C.c_int, [C.c_void_p, C.c_void_p, c_f64p]),
    "kkt_apply_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "kkt_pc_apply_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "kkt_solve_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int),
                                   C.POINTER(C.c_int), c_f64p, c_f64p, C.c_int,
                                   C.POINTER(C.c_int)]),
    "kkt_sync": (C.c_int, [C.c_void_p]),
    "kkt_time_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                 C.POINTER(C.c_float)]),
    "kkt_time_pc_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                    C.POINTER(C.c_float)]),
    "kkt_time_pc_sweeps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float),
                                     C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "kkt_get_stage_times": (C.c_int, [C.c_void_p, C.POINTER(StageTimes)]),
    "kkt_time_pc_stages": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.POINTER(PcStageTimes)]),
    "kkt_debug_set_steplock": (C.c_int, [C.c_void_p, C.POINTER(StepLock)]),
    "kkt_get_info": (C.c_int, [C.c_void_p, C.POINTER(Info)]),
    "kkt_comm_unique_id": (C.c_int, [C.c_void_p]),
    "kkt_comm_init_rccl": (C.c_int, [C.c_void_p, C.c_void_p]),
    "kkt_comm_init_callbacks": (C.c_int, [C.c_void_p, ALLREDUCE_FN, SENDRECV_FN,
                                          C.c_void_p]),
    "kkt_comm_barrier": (C.c_int, [C.c_void_p]),
    "kkt_comm_max": (C.c_int, [C.c_void_p, c_f64p]),
}

_lib = None


def load():
    """The loaded library with typed entry points.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C control_amd/csrc` "
            "(hipcc --offload-arch=gfx950).  control_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def f64(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(c_f64p)


def i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(c_i32p)
