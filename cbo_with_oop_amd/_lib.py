"""ctypes binding of libcbo_hip.so (include/cbo_hip.h).  No PyTorch, no fallback: if the shared
library is missing, or no gfx950 GPU is visible when a context is requested, this raises."""
from __future__ import annotations

import atexit
import ctypes
import os
import threading
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CBO_HIP_LIB", os.path.join(_HERE, "libcbo_hip.so"))

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int64_p = ctypes.POINTER(ctypes.c_int64)
c_int_p = ctypes.POINTER(ctypes.c_int)
c_void_pp = ctypes.POINTER(ctypes.c_void_p)

ABI_VERSION = 5          # include/cbo_hip.h: CBO_HIP_ABI_VERSION
ABI_DIAG_BASE = 1000     # CBO_HIP_ABI_DIAG_BASE: timing-only builds report ABI_DIAG_BASE + version
CBO_OK = 0
CBO_ERR_INVALID = -1
CBO_ERR_HIP = -2
CBO_ERR_NOT_PD = -3
CBO_ERR_NONPOS_DIAG = -4
CBO_ERR_NOT_FITTED = -5
CBO_ERR_UNSUPPORTED = -6
CBO_ERR_NO_DEVICE = -7
CBO_ERR_COMM = -8

TASK_CODE = {"min": 0, "max": 1}
DTYPE_CODE = {"f64": 0, "f32": 1}


class CboTimers(ctypes.Structure):
    _fields_ = [("ms_kxx", ctypes.c_double), ("ms_chol", ctypes.c_double), ("ms_alpha", ctypes.c_double),
                ("ms_kstar", ctypes.c_double), ("ms_trsm", ctypes.c_double), ("ms_acq", ctypes.c_double),
                ("n_fit", ctypes.c_int64), ("n_sweep", ctypes.c_int64), ("n_trsm_launches", ctypes.c_int64),
                ("trsm_flops", ctypes.c_double), ("ms_f32_convert", ctypes.c_double)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


SEM_MAX_NODES = 16
SEM_MAX_TERMS = 64
SEM_FN_CODE = {"id": 0, "square": 1, "exp": 2, "cos": 3, "sin": 4}


class CboSemSpec(ctypes.Structure):
    """struct cbo_sem_spec of include/cbo_hip.h."""
    _fields_ = [("n_nodes", ctypes.c_int), ("eps_index", ctypes.c_int * SEM_MAX_NODES),
                ("term_begin", ctypes.c_int * (SEM_MAX_NODES + 1)), ("term_parent", ctypes.c_int * SEM_MAX_TERMS),
                ("term_fn", ctypes.c_int * SEM_MAX_TERMS), ("term_a", ctypes.c_double * SEM_MAX_TERMS),
                ("term_c", ctypes.c_double * SEM_MAX_TERMS)]


class CboHipError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libcbo_hip error {code}: {message}")
        self.code = code


# every exported symbol of include/cbo_hip.h: name -> (restype, argtypes)
SIGNATURES = {
    "cbo_abi_version": (ctypes.c_int, []),
    "cbo_last_error": (ctypes.c_char_p, []),
    "cbo_device_count": (ctypes.c_int, [c_int_p]),
    "cbo_init": (ctypes.c_int, [ctypes.c_int, c_void_pp]),
    "cbo_shutdown": (None, [ctypes.c_void_p]),
    "cbo_synchronize": (ctypes.c_int, [ctypes.c_void_p]),
    "cbo_set_profiling": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "cbo_reset_timers": (ctypes.c_int, [ctypes.c_void_p]),
    "cbo_get_timers": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(CboTimers)]),
    "cbo_region_begin": (ctypes.c_int, [ctypes.c_void_p]),
    "cbo_region_end": (ctypes.c_int, [ctypes.c_void_p, c_double_p]),
    "cbo_device_name": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]),
    "cbo_gp_create": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int, c_double_p,
                                     c_double_p, c_double_p, c_double_p, ctypes.c_double, c_double_p, ctypes.c_int,
                                     ctypes.c_double, ctypes.c_int, c_void_pp]),
    "cbo_gp_destroy": (None, [ctypes.c_void_p]),
    "cbo_gp_fit": (ctypes.c_int, [ctypes.c_void_p, c_int_p, c_double_p]),
    "cbo_gp_set_data": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, c_double_p, c_double_p, c_double_p,
                                       c_double_p]),
    "cbo_gp_upload_data": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, c_double_p, c_double_p, c_double_p,
                                          c_double_p]),
    "cbo_gp_predict": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, c_double_p, c_double_p, c_double_p,
                                      ctypes.c_int, c_double_p, c_double_p]),
    "cbo_gp_set_hyper": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_double, c_double_p, ctypes.c_double]),
    "cbo_gp_log_marginal": (ctypes.c_int, [ctypes.c_void_p, c_double_p]),
    "cbo_gp_lml_gradients": (ctypes.c_int, [ctypes.c_void_p, c_double_p, c_double_p, c_double_p, c_double_p]),
    "cbo_gp_predict_gradients": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, c_double_p, c_double_p, c_double_p,
                                                c_double_p]),
    "cbo_gp_predict_grouped": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, c_double_p, c_double_p,
                                              c_double_p, ctypes.c_int, c_double_p, c_double_p]),
    "cbo_gp_predict_do": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, c_double_p, ctypes.c_int,
                                         c_double_p, c_int_p, ctypes.c_int, c_double_p, c_double_p]),
    "cbo_gp_get_posterior": (ctypes.c_int, [ctypes.c_void_p, c_double_p, c_double_p]),
    "cbo_gp_assemble_kxx": (ctypes.c_int, [ctypes.c_void_p, c_double_p]),
    "cbo_gp_n": (ctypes.c_int64, [ctypes.c_void_p]),
    "cbo_gp_dtype": (ctypes.c_int, [ctypes.c_void_p]),
    "cbo_gp_jitter": (ctypes.c_int, [ctypes.c_void_p, c_int_p, c_double_p]),
    "cbo_cands_create": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, c_double_p, c_double_p,
                                        c_double_p, ctypes.c_int64, c_void_pp]),
    "cbo_cands_destroy": (None, [ctypes.c_void_p]),
    "cbo_cands_keep_solution": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "cbo_gp_append": (ctypes.c_int, [ctypes.c_void_p, c_double_p, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                     c_int_p]),
    "cbo_acq_sweep": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_int,
                                     ctypes.c_double, ctypes.c_double, c_double_p, c_double_p, c_double_p,
                                     c_double_p, c_int64_p]),
    "cbo_gp_fit_sweep": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_int,
                                        ctypes.c_double, ctypes.c_double, c_double_p, c_double_p, c_double_p,
                                        c_double_p, c_int64_p, c_int_p, c_double_p]),
    "cbo_acq_sweep_sets": (ctypes.c_int, [ctypes.c_int, c_void_pp, c_void_pp, c_double_p, ctypes.c_int, ctypes.c_double,
                                          c_double_p, c_double_p, c_int64_p]),
    "cbo_gp_fit_level": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, c_int_p, c_double_p]),
    "cbo_comm_gather_i64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, c_int64_p]),
    "cbo_comm_share_factor": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, c_int_p, ctypes.c_int,
                                              c_int_p, ctypes.c_int]),
    "cbo_gp_take_factor_slices": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
    "cbo_schedule_report": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64]),
    "cbo_trial_step": (ctypes.c_int, [ctypes.c_int, c_void_pp, c_void_pp, ctypes.c_int, ctypes.c_int64, c_double_p, c_double_p,
                                      c_double_p, c_double_p, c_double_p, ctypes.c_int, ctypes.c_double, c_double_p, c_double_p,
                                      c_int64_p, c_int_p]),
    "cbo_acq_sweep_host": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, c_double_p, c_double_p, c_double_p,
                                          ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                          c_double_p, c_double_p, c_int64_p]),
    "cbo_argmax_sets": (ctypes.c_int, [c_double_p, ctypes.c_int, c_int_p]),
    "cbo_argmax_pairs": (ctypes.c_int, [c_double_p, c_int64_p, ctypes.c_int, c_double_p, c_int64_p]),
    "cbo_sem_create": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(CboSemSpec), ctypes.c_int64, ctypes.c_int,
                                      c_double_p, c_void_pp]),
    "cbo_sem_destroy": (None, [ctypes.c_void_p]),
    "cbo_sem_target": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int, c_int_p,
                                      c_double_p, c_double_p]),
    "cbo_selftest_mfma": (ctypes.c_int, [ctypes.c_void_p, c_double_p]),
    "cbo_comm_unique_id": (ctypes.c_int, [ctypes.c_void_p]),
    "cbo_comm_init_rank": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, c_void_pp]),
    "cbo_comm_init_all": (ctypes.c_int, [ctypes.c_int, c_void_pp, c_void_pp]),
    "cbo_comm_destroy": (None, [ctypes.c_void_p]),
    "cbo_comm_size": (ctypes.c_int, [ctypes.c_void_p, c_int_p, c_int_p]),
    "cbo_comm_argmax": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_double, ctypes.c_int64, c_double_p, c_int64_p]),
    "cbo_comm_argmax_all": (ctypes.c_int, [ctypes.c_int, c_void_pp, c_double_p, c_int64_p, c_double_p, c_int64_p]),
    "cbo_comm_max_f64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_double, c_double_p]),
    "cbo_comm_barrier": (ctypes.c_int, [ctypes.c_void_p]),
}

_lib = None
_lock = threading.Lock()


def load():
    """Load libcbo_hip.so (once) and declare every prototype.  Raises if the library is missing."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise ImportError(
                    f"{LIB_PATH} not found: build it with `make -C cbo_with_oop_amd/csrc` (or "
                    f"`python -c 'import __graft_entry__ as g; g.build()'`).  There is no CPU fallback.")
            lib = ctypes.CDLL(LIB_PATH)
            for name, (restype, argtypes) in SIGNATURES.items():
                fn = getattr(lib, name)       # AttributeError if the .so lacks a declared symbol
                fn.restype = restype
                fn.argtypes = argtypes
            version = lib.cbo_abi_version()
            if version >= ABI_DIAG_BASE and os.environ.get("CBO_HIP_ALLOW_DIAG") != "1":
                raise ImportError(
                    f"{LIB_PATH} is a timing-only diagnostic build (cbo_abi_version() = {version}): its results may be "
                    f"wrong by construction.  Measurement scripts opt in with CBO_HIP_ALLOW_DIAG=1.")
            if version % ABI_DIAG_BASE != ABI_VERSION:
                raise ImportError(f"{LIB_PATH} has ABI version {version}, this package binds version {ABI_VERSION}: "
                                  f"rebuild it (`make -C cbo_with_oop_amd/csrc`)")
            _lib = lib
    return _lib


def check(rc):
    if rc != CBO_OK:
        msg = load().cbo_last_error()
        raise_for(rc, msg.decode() if msg else "")


def raise_for(rc, msg):
    if rc in (CBO_ERR_NOT_PD, CBO_ERR_NONPOS_DIAG):
        # GPy's jitchol raises numpy.linalg.LinAlgError (SURVEY.md §8b conventions)
        raise np.linalg.LinAlgError(msg)
    raise CboHipError(rc, msg)


def dptr(a):
    """float64 C-contiguous ndarray (or None) -> double*; the array must be kept alive by the caller."""
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags.c_contiguous
    n = a.size
    if 0 < n <= 65536 and a.flags.writeable:
        # a ctypes array over the same memory: accepted where double* is declared, and four times cheaper to make than
        # ndarray.ctypes.data_as (0.8 against 3.5 us -- a reference-scale trial is 50 us)
        t = _array_types.get(n)
        if t is None:
            t = _array_types[n] = ctypes.c_double * n
        return t.from_buffer(a)
    return a.ctypes.data_as(c_double_p)


_array_types = {}


def as_f64(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


class Context:
    """One HIP device + stream (cbo_ctx).  Contexts are cached per device id."""
    _cache = {}
    _live = weakref.WeakSet()      # every context not yet shut down, cached or not

    def __init__(self, device_id=0):
        lib = load()
        h = ctypes.c_void_p()
        check(lib.cbo_init(int(device_id), ctypes.byref(h)))
        self.handle = h
        self.device_id = int(device_id)
        self.closed = False
        Context._live.add(self)

    def close(self):
        """cbo_shutdown.  Handles created on this context must not be used (or destroyed) afterwards; the
        wrappers check ``closed`` before they free anything."""
        if not self.closed:
            self.closed = True
            load().cbo_shutdown(self.handle)
            self.handle = None

    @classmethod
    def get(cls, device_id=None):
        if device_id is None:
            device_id = int(os.environ.get("CBO_HIP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
            n = device_count()
            if n > 0:
                device_id %= n
        ctx = cls._cache.get(device_id)
        if ctx is None:
            ctx = cls(device_id)
            cls._cache[device_id] = ctx
        return ctx

    def synchronize(self):
        check(load().cbo_synchronize(self.handle))

    def schedule_report(self):
        """(shapes still exploring, text): what the context has measured and chosen for ``cbo_gp_fit_sweep``, one line per
        shape (``cbo_schedule_report``)."""
        buf = ctypes.create_string_buffer(1 << 16)
        n = load().cbo_schedule_report(self.handle, buf, len(buf))
        if n < 0:
            check(n)
        return n, buf.value.decode()

    def set_profiling(self, enabled):
        check(load().cbo_set_profiling(self.handle, int(bool(enabled))))

    def reset_timers(self):
        check(load().cbo_reset_timers(self.handle))

    def timers(self):
        t = CboTimers()
        check(load().cbo_get_timers(self.handle, ctypes.byref(t)))
        return t.as_dict()

    def region_begin(self):
        check(load().cbo_region_begin(self.handle))

    def region_end(self):
        ms = ctypes.c_double(0.0)
        check(load().cbo_region_end(self.handle, ctypes.byref(ms)))
        return ms.value

    def name(self):
        buf = ctypes.create_string_buffer(256)
        check(load().cbo_device_name(self.handle, buf, 256))
        return buf.value.decode()

    def selftest_mfma(self):
        err = ctypes.c_double(-1.0)
        check(load().cbo_selftest_mfma(self.handle, ctypes.byref(err)))
        return err.value


@atexit.register
def _close_contexts():
    # Tear the device state down while the HIP runtime is still fully alive: streams created with a CU mask that
    # survive into the runtime's own exit handlers crash profiling tools that hook finalisation (rocprofv3).
    for ctx in list(Context._live):
        try:
            ctx.close()
        except Exception:
            pass
    Context._cache.clear()


def device_count():
    n = ctypes.c_int(0)
    check(load().cbo_device_count(ctypes.byref(n)))
    return n.value
