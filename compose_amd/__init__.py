"""compose_amd -- MI355X-native CEDR limiter hot path (QLT, CAAS, BFB tree all-reduce).

This package is a thin ctypes binding of ``libcedr_hip.so`` (hand-written HIP
kernels for gfx950 + C++ host classes), whose C ABI is declared in
``include/cedr_hip.h``. The Python classes mirror the reference's plugin
interface (cedr::CDR protocol, reference cedr/cedr_cdr.hpp:16-112): same method
names, argument meaning and error behaviour (errors surface as ``CedrError``
carrying the C++ ``std::logic_error`` text).

There is NO CPU fallback: if the shared library is missing the import fails,
and compute calls fail if no GPU is present.
"""
import ctypes as C
import os

import numpy as np

__all__ = ["CedrError", "ProblemType", "Parallel", "Tree", "QLT", "CAAS", "BfbTreeAllReducer",
           "DeviceArray", "lib", "LIB_PATH", "build", "device_count"]

_HERE = os.path.dirname(os.path.abspath(__file__))
# CEDR_HIP_LIB selects another build of the same library (kernel tuning experiments).
LIB_PATH = os.environ.get("CEDR_HIP_LIB") or os.path.join(_HERE, "libcedr_hip.so")


class CedrError(RuntimeError):
    pass


class ProblemType:
    """reference cedr/cedr.hpp:29-39"""
    conserve = 1
    shapepreserve = 2
    consistent = 4
    nonnegative = 8


def build():
    """Compile libcedr_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "csrc")])


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_lp = C.POINTER(C.c_long)
_vp = C.c_void_p

USER_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, _vp, _vp, _vp, C.c_int, C.c_int)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, _vp, _vp, _vp, C.c_size_t, _vp)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, _vp, _vp, _vp, C.c_size_t, _vp)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, _vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                          C.POINTER(C.c_int), _vp)


class _QltDeviceOp(C.Structure):
    _fields_ = [("l2r_data", _vp), ("r2l_data", _vp), ("stride", C.c_int64),
                ("trcr2prob", _vp), ("trcr2bl2r", _vp), ("trcr2br2l", _vp)]


class _CaasDeviceOp(C.Structure):
    _fields_ = [("d", _vp), ("nlclcells", C.c_int), ("nt", C.c_int), ("need_conserve", C.c_int)]


_SIGS = {
    # name: (restype, argtypes)
    "cedr_last_error": (C.c_char_p, []),
    "cedr_parallel_create_self": (_vp, []),
    "cedr_rccl_get_unique_id": (C.c_int, [_vp]),
    "cedr_parallel_create_rccl": (_vp, [_vp, C.c_int, C.c_int]),
    "cedr_parallel_create_rehearsal": (_vp, [_vp, C.c_int]),
    "cedr_peer_mailbox_create": (_vp, [C.c_int, C.c_size_t, _vp]),
    "cedr_peer_mailbox_destroy": (None, [_vp]),
    "cedr_parallel_create_peer_flags": (_vp, [_vp, _vp, C.c_int, C.c_int]),
    "cedr_parallel_create_callbacks": (_vp, [C.c_int, C.c_int, ALLGATHER_FN, ALLREDUCE_FN, _vp]),
    "cedr_parallel_create_callbacks_p2p": (_vp, [C.c_int, C.c_int, ALLGATHER_FN, ALLREDUCE_FN, EXCHANGE_FN, _vp]),
    "cedr_parallel_exchange": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                         C.POINTER(C.c_int), _vp]),
    "cedr_parallel_allgather": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp]),
    "cedr_parallel_allreduce_sum": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp]),
    "cedr_parallel_rank": (C.c_int, [_vp]),
    "cedr_parallel_size": (C.c_int, [_vp]),
    "cedr_parallel_free": (None, [_vp]),
    "cedr_tree_create": (_vp, [C.c_int, C.c_int, _ip, _ip, _ip, _lp]),
    "cedr_tree_create_levels": (_vp, [C.c_int, C.c_int, _ip, _ip, _ip, _lp, _ip]),
    "cedr_tree_prune": (_vp, [_vp, C.c_int]),
    "cedr_tree_nnodes": (C.c_int, [_vp]),
    "cedr_tree_make_1d": (_vp, [C.c_int] * 4),
    "cedr_tree_free": (None, [_vp]),
    "cedr_tree_analyze": (_vp, [_vp, C.c_int, C.c_int]),
    "cedr_ns_free": (None, [_vp]),
    "cedr_ns_nslots": (C.c_int, [_vp]),
    "cedr_ns_nnode": (C.c_int, [_vp]),
    "cedr_ns_nlevels": (C.c_int, [_vp]),
    "cedr_ns_level_size": (C.c_int, [_vp, C.c_int]),
    "cedr_ns_level_nme": (C.c_int, [_vp, C.c_int]),
    "cedr_ns_level_nkids": (C.c_int, [_vp, C.c_int]),
    "cedr_ns_nodes": (None, [_vp, _ip]),
    "cedr_ns_level_nodes": (None, [_vp, C.c_int, _ip]),
    "cedr_ns_level_msgs": (None, [_vp, C.c_int, C.c_int, _ip]),
    "cedr_ns_check_comm": (C.c_int, [_vp]),
    "cedr_qlt_create": (_vp, [_vp, C.c_int, _vp, C.c_int]),
    "cedr_qlt_free": (None, [_vp]),
    "cedr_qlt_declare_tracer": (C.c_int, [_vp, C.c_int, C.c_int]),
    "cedr_qlt_end_tracer_declarations": (C.c_int, [_vp]),
    "cedr_qlt_get_buffers_sizes": (C.c_int, [_vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "cedr_qlt_get_buffers_sizes_tight": (C.c_int, [_vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "cedr_qlt_set_buffers": (C.c_int, [_vp, _vp, _vp]),
    "cedr_qlt_finish_setup": (C.c_int, [_vp]),
    "cedr_qlt_get_problem_type": (C.c_int, [_vp, C.c_int]),
    "cedr_qlt_get_num_tracers": (C.c_int, [_vp]),
    "cedr_qlt_nlclcells": (C.c_int, [_vp]),
    "cedr_qlt_get_owned_glblcells": (C.c_int, [_vp, _lp]),
    "cedr_qlt_gci2lci": (C.c_int, [_vp, C.c_int]),
    "cedr_qlt_run": (C.c_int, [_vp]),
    "cedr_qlt_set_stream": (C.c_int, [_vp, _vp]),
    "cedr_qlt_placement_report": (C.c_char_p, [_vp]),
    "cedr_qlt_sweep_down_event": (_vp, [_vp]),
    "cedr_qlt_sweep_mid_event": (_vp, [_vp]),
    "cedr_stream_wait_event": (C.c_int, [_vp, _vp]),
    "cedr_qlt_get_device_op": (C.c_int, [_vp, C.POINTER(_QltDeviceOp)]),
    "cedr_qlt_set_rhom_bulk": (C.c_int, [_vp, _vp]),
    "cedr_qlt_set_Qm_bulk": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "cedr_qlt_get_Qm_bulk": (C.c_int, [_vp, _vp]),
    "cedr_qlt_metadata": (C.c_int, [_vp] + [_ip] * 8),
    "cedr_qlt_nodesets": (_vp, [_vp]),
    "cedr_plan_create": (_vp, [_vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "cedr_plan_free": (None, [_vp]),
    "cedr_qlt_plan": (_vp, [_vp]),
    "cedr_plan_summary": (C.c_int, [_vp, C.POINTER(C.c_int64)]),
    "cedr_plan_stage_sizes": (C.c_int, [_vp, C.c_int, C.c_int, _ip]),
    "cedr_plan_stage": (C.c_int, [_vp, C.c_int, C.c_int, _ip, C.POINTER(C.c_uint16), _ip]),
    "cedr_plan_stage_is_top": (C.c_int, [_vp, C.c_int, C.c_int]),
    "cedr_plan_stage_micro": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(C.c_uint16), C.POINTER(C.c_uint16)]),
    "cedr_plan_exchange": (C.c_int, [_vp, _ip, _ip, _ip]),
    "cedr_plan_levels": (C.c_int, [_vp, _ip]),
    "cedr_plan_level_sizes": (C.c_int, [_vp, C.c_int, _ip]),
    "cedr_plan_level": (C.c_int, [_vp, C.c_int, _ip, _ip, _ip, _ip, _ip]),
    "cedr_caas_create": (_vp, [_vp, C.c_int, USER_ALLREDUCE_FN, C.c_int, _vp]),
    "cedr_caas_free": (None, [_vp]),
    "cedr_caas_declare_tracer": (C.c_int, [_vp, C.c_int, C.c_int]),
    "cedr_caas_end_tracer_declarations": (C.c_int, [_vp]),
    "cedr_caas_get_buffers_sizes": (C.c_int, [_vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "cedr_caas_set_buffers": (C.c_int, [_vp, _vp, _vp]),
    "cedr_caas_finish_setup": (C.c_int, [_vp]),
    "cedr_caas_get_problem_type": (C.c_int, [_vp, C.c_int]),
    "cedr_caas_get_num_tracers": (C.c_int, [_vp]),
    "cedr_caas_run": (C.c_int, [_vp]),
    "cedr_caas_set_stream": (C.c_int, [_vp, _vp]),
    "cedr_caas_set_Qm_source": (C.c_int, [_vp, _vp]),
    "cedr_caas_get_device_op": (C.c_int, [_vp, C.POINTER(_CaasDeviceOp)]),
    "cedr_caas_get_recv": (C.c_int, [_vp, C.POINTER(_vp)]),
    "cedr_caas_create_bfb": (_vp, [_vp, C.c_int, _vp, C.c_int]),
    "cedr_caas_fused_with_bfb": (C.c_int, [_vp]),
    "cedr_bfb_create": (_vp, [_vp, _vp, C.c_int, C.c_int]),
    "cedr_bfb_free": (None, [_vp]),
    "cedr_bfb_nlocal": (C.c_int, [_vp]),
    "cedr_bfb_allreduce": (C.c_int, [_vp, _vp, _vp, C.c_int]),
    "cedr_bfb_set_stream": (C.c_int, [_vp, _vp]),
    "cedr_device_count": (C.c_int, []),
    "cedr_malloc": (_vp, [C.c_size_t]),
    "cedr_free": (None, [_vp]),
    "cedr_caas_bind_arrays": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "cedr_qlt_bind_arrays": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cedr_memcpy_h2d": (C.c_int, [_vp, _vp, C.c_size_t]),
    "cedr_memcpy_d2h": (C.c_int, [_vp, _vp, C.c_size_t]),
    "cedr_memcpy_d2d_async": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "cedr_stream_synchronize": (C.c_int, [C.c_void_p]),
    "cedr_memcpy_h2d_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "cedr_memcpy_d2h_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "cedr_stream_create_cu_mask": (C.c_int, [C.POINTER(C.c_uint32), C.c_int32, C.POINTER(C.c_void_p)]),
    "cedr_stream_destroy": (C.c_int, [C.c_void_p]),
    "cedr_device_synchronize": (C.c_int, []),
    # kernel level (layer K) entry points used directly by tests / bench
    "cedr_hip_level_up": (C.c_int, [_vp, C.c_int32, C.c_int32, _vp, C.c_int64, _vp, C.c_int, _vp]),
    "cedr_hip_qlt_level_down": (C.c_int, [_vp, _vp, C.c_int32, _vp, _vp, _vp]),
    "cedr_hip_caas_nchunks": (C.c_int32, [C.c_int32]),
    "cedr_hip_caas_reduce_local": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int, _vp, _vp, _vp, _vp]),
    "cedr_hip_caas_sum_partials": (C.c_int, [C.c_int32, C.c_int32, _vp, _vp, _vp]),
    "cedr_hip_caas_finish_local": (C.c_int, [C.c_int32, C.c_int32, _vp, _vp, _vp]),
    "cedr_hip_profile_start": (None, []),
    "cedr_hip_profile_stop": (C.c_int, []),
    "cedr_hip_profile_get": (C.c_int, [C.c_int, _ip, C.POINTER(C.c_int64), C.POINTER(C.c_float)]),
    "cedr_hip_profile_name": (C.c_char_p, [C.c_int]),
    "cedr_local_solve": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, C.c_double, _dp, _dp, _dp, _dp]),
    "cedr_hip_node_problem_batch": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cedr_hip_local_batch": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
}

_LIB = None


def lib():
    """Load libcedr_hip.so (once). Raises CedrError if it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise CedrError("%s is missing: run compose_amd.build() (or `make -C compose_amd/csrc`). "
                        "There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in _SIGS.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _LIB = L
    return L


def _err():
    return lib().cedr_last_error().decode(errors="replace")


def _chk(rc, what=""):
    if rc != 0:
        msg = _err() if rc == -1 else "HIP error %d" % rc
        raise CedrError("%s: %s" % (what, msg) if what else msg)


def _handle(h, what):
    if not h:
        raise CedrError("%s: %s" % (what, _err()))
    return h


def device_count():
    return lib().cedr_device_count()


def kernels_sha256():
    """sha256 of the cedr_kernels.hip the loaded library was built from (stamped in at build time)."""
    f = lib().cedr_hip_kernels_sha256
    f.restype = C.c_char_p
    return f().decode()


class MaskedStream:
    """A HIP stream restricted to the compute units whose bit is set in `words` (32 per word;
    cedr_stream_create_cu_mask). `.handle` is what set_stream() of a CDR takes."""

    def __init__(self, words):
        arr = (C.c_uint32 * len(words))(*[int(w) & 0xFFFFFFFF for w in words])
        h = C.c_void_p()
        _chk(lib().cedr_stream_create_cu_mask(arr, len(words), C.byref(h)), "cedr_stream_create_cu_mask")
        self.handle = h.value

    def close(self):
        if self.handle:
            lib().cedr_stream_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass


def _ptr(x, need=None):
    """Device pointer of a DeviceArray, torch tensor, int or None. Arrays are checked before their
    address is handed to a kernel: 8-byte elements (float64), contiguous, on the GPU, and at least
    'need' elements long -- a float32, strided, host or short array would otherwise be silent
    corruption or a GPU fault."""
    if x is None:
        return None
    if isinstance(x, DeviceArray):
        if x.dtype.itemsize != 8:
            raise CedrError("expected an array of doubles, got %s" % x.dtype)
        if need is not None and x.n < need:
            raise CedrError("device array has %d elements, %d are needed" % (x.n, need))
        return x.ptr
    if isinstance(x, int):
        return x
    if hasattr(x, "data_ptr"):
        if str(getattr(x, "dtype", "")) not in ("torch.float64", "float64"):
            raise CedrError("expected a float64 tensor, got %s" % getattr(x, "dtype", None))
        if hasattr(x, "is_contiguous") and not x.is_contiguous():
            raise CedrError("expected a contiguous tensor")
        if hasattr(x, "is_cuda") and not x.is_cuda:
            raise CedrError("expected a tensor on the GPU")
        if need is not None and x.numel() < need:
            raise CedrError("tensor has %d elements, %d are needed" % (x.numel(), need))
        return x.data_ptr()
    raise TypeError("expected a device array, got %r" % type(x))


class DeviceArray:
    """A hipMalloc'ed array of doubles (or ints) with numpy upload/download."""

    def __init__(self, n, dtype=np.float64):
        self.dtype = np.dtype(dtype)
        self.n = int(n)
        self.ptr = _handle(lib().cedr_malloc(max(1, self.n) * self.dtype.itemsize), "cedr_malloc")

    @staticmethod
    def from_numpy(a):
        a = np.ascontiguousarray(a)
        d = DeviceArray(a.size, a.dtype)
        d.upload(a)
        return d

    def upload(self, a):
        a = np.ascontiguousarray(a, self.dtype)
        assert a.size <= self.n
        _chk(lib().cedr_memcpy_h2d(self.ptr, a.ctypes.data, a.size * self.dtype.itemsize), "h2d")

    def numpy(self, shape=None):
        out = np.empty(self.n, self.dtype)
        _chk(lib().cedr_memcpy_d2h(out.ctypes.data, self.ptr, self.n * self.dtype.itemsize), "d2h")
        return out if shape is None else out.reshape(shape)

    def offset(self, nelem):
        return self.ptr + int(nelem) * self.dtype.itemsize

    def __del__(self):
        try:
            if self.ptr:
                lib().cedr_free(self.ptr)
        except Exception:
            pass


LOCAL_BC_QP, LOCAL_BC_QP_2D, LOCAL_CAAS, LOCAL_NONNEG_LS, LOCAL_NONNEG_CAAS = range(5)


def local_solve(kind, w, a, b, xlo, xhi, y, clip=True, early_exit_on_tol=True, max_its=100):
    """cedr::local::* evaluated on the HOST by libcedr_hip.so (cedr_local.hpp; reference
    cedr/cedr_local.hpp:24-60). Returns (info, x)."""
    arrs = [np.ascontiguousarray(v, np.float64) for v in (w, a, xlo, xhi, y)]
    n = len(arrs[1])
    x = np.zeros(n)
    flags = (1 if clip else 0) | (2 if early_exit_on_tol else 0)
    info = lib().cedr_local_solve(kind, n, flags, max_its, arrs[0].ctypes.data_as(_dp), arrs[1].ctypes.data_as(_dp),
                                  float(b), arrs[2].ctypes.data_as(_dp), arrs[3].ctypes.data_as(_dp),
                                  arrs[4].ctypes.data_as(_dp), x.ctypes.data_as(_dp))
    return info, x


def local_batch(kind, w, a, b, xlo, xhi, y, clip=True, early_exit_on_tol=True, max_its=100):
    """The same functions on the GPU, one problem per thread. Arrays are [nprob][n], b is [nprob].
    Returns (info[nprob], x[nprob][n])."""
    w, a, xlo, xhi, y = (np.ascontiguousarray(v, np.float64) for v in (w, a, xlo, xhi, y))
    nprob, n = a.shape
    rec = np.concatenate([w, a, xlo, xhi, y, np.asarray(b, np.float64).reshape(nprob, 1)], axis=1)
    din = DeviceArray.from_numpy(rec)
    dx = DeviceArray(nprob * n)
    di = DeviceArray(nprob, np.int32)
    flags = (1 if clip else 0) | (2 if early_exit_on_tol else 0)
    _chk(lib().cedr_hip_local_batch(kind, nprob, n, flags, max_its, din.ptr, dx.ptr, di.ptr, None), "local_batch")
    synchronize()
    return di.numpy(), dx.numpy((nprob, n))


def node_problem_batch(problem_type, rhom, pd, Qm, rhom0, k0d, rhom1, k1d, prefer=False, pair=False):
    """solve_node_problem (reference cedr_qlt_inl.hpp:120-203) as the sweep kernels evaluate it, one
    problem per GPU thread. pd, k0d, k1d are [nprob][3]; the rest [nprob]. Returns [nprob][2]."""
    c = lambda v: np.asarray(v, np.float64).reshape(len(Qm), -1)
    rec = np.ascontiguousarray(np.concatenate([c(rhom), c(pd), c(Qm), c(rhom0), c(k0d), c(rhom1), c(k1d)], axis=1))
    assert rec.shape[1] == 13
    din = DeviceArray.from_numpy(rec)
    dout = DeviceArray(2 * len(rec))
    _chk(lib().cedr_hip_node_problem_batch(problem_type, len(rec), int(prefer), int(pair), din.ptr, dout.ptr, None),
         "node_problem_batch")
    synchronize()
    return dout.numpy((len(rec), 2))


class PeerMailbox:
    """A rank's mailbox of the peer-flag transport (Parallel.peer_mailbox)."""

    def __init__(self, handle):
        self.h = _handle(handle, "PeerMailbox")

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().cedr_peer_mailbox_destroy(self.h)
                self.h = None
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass


class Parallel:
    """cedr::mpi::Parallel (reference cedr/cedr_mpi.hpp:17-27)."""

    def __init__(self, handle, keep=None):
        self.h = _handle(handle, "Parallel")
        self._keep = keep

    @staticmethod
    def self_():
        return Parallel(lib().cedr_parallel_create_self())

    @staticmethod
    def rccl(unique_id, rank, size):
        buf = C.create_string_buffer(bytes(unique_id), 128)
        return Parallel(lib().cedr_parallel_create_rccl(buf, rank, size))

    @staticmethod
    def rehearsal(inner, pretended_size):
        """Rank 0 of pretended_size ranks over `inner`, a Parallel of ONE real rank (self_() or rccl(id, 0, 1)):
        collectives run through `inner` for real, the other ranks' all-gather blocks are copies of this rank's.
        What one GPU can run of an N-GPU job (measurements, tests)."""
        return Parallel(lib().cedr_parallel_create_rehearsal(inner.h, int(pretended_size)), keep=inner)

    PEER_HANDLE_BYTES = 96

    @staticmethod
    def peer_mailbox(size, slot_doubles=4096):
        """This rank's mailbox for the peer-flag transport (cedr_mpi.hpp): returns (mailbox, handle bytes). The launcher
        gathers every rank's handle (rank order) and hands them to peer_flags()."""
        buf = C.create_string_buffer(Parallel.PEER_HANDLE_BYTES)
        return PeerMailbox(lib().cedr_peer_mailbox_create(int(size), int(slot_doubles), buf)), buf.raw

    @staticmethod
    def peer_flags(mailbox, handles, rank, size):
        """One process per GPU of one node over stream-ordered stores into the peers' mailboxes (no communicator
        library): `handles` = the ranks' handle bytes in rank order (a list, or already concatenated)."""
        blob = b"".join(handles) if isinstance(handles, (list, tuple)) else bytes(handles)
        if len(blob) != Parallel.PEER_HANDLE_BYTES * int(size):
            raise CedrError("peer_flags: %d bytes of handles for %d ranks" % (len(blob), size))
        buf = C.create_string_buffer(blob, len(blob))
        return Parallel(lib().cedr_parallel_create_peer_flags(mailbox.h, buf, int(rank), int(size)), keep=mailbox)

    @staticmethod
    def rccl_unique_id():
        buf = C.create_string_buffer(128)
        rc = lib().cedr_rccl_get_unique_id(buf)
        if rc != 0:
            raise CedrError("ncclGetUniqueId failed (%d)" % rc)
        return buf.raw

    @staticmethod
    def callbacks(rank, size, allgather, allreduce_sum):
        """allgather(send_ptr, recv_ptr, count, stream) / allreduce_sum(...) -> int, on device pointers."""
        ag = ALLGATHER_FN(lambda user, s, r, n, st: int(allgather(s, r, n, st) or 0))
        ar = ALLREDUCE_FN(lambda user, s, r, n, st: int(allreduce_sum(s, r, n, st) or 0))
        return Parallel(lib().cedr_parallel_create_callbacks(rank, size, ag, ar, None), keep=(ag, ar))

    @staticmethod
    def callbacks_p2p(rank, size, allgather, allreduce_sum, exchange):
        """The same plus exchange(ops, stream) -> int with ops = [(peer, device_ptr, count, is_send), ...]:
        one round of point-to-point messages (plans in level mode: partial trees, scattered ownership)."""
        ag = ALLGATHER_FN(lambda user, s, r, n, st: int(allgather(s, r, n, st) or 0))
        ar = ALLREDUCE_FN(lambda user, s, r, n, st: int(allreduce_sum(s, r, n, st) or 0))

        def x(user, nops, peers, bufs, counts, sends, st):
            ops = [(peers[i], bufs[i], counts[i], bool(sends[i])) for i in range(nops)]
            return int(exchange(ops, st) or 0)
        xf = EXCHANGE_FN(x)
        return Parallel(lib().cedr_parallel_create_callbacks_p2p(rank, size, ag, ar, xf, None), keep=(ag, ar, xf))

    @staticmethod
    def host_staged_p2p(rank, size, allgather_host, allreduce_host, exchange_host):
        """host_staged plus exchange_host(sends, recvs) with sends = [(peer, array)], recvs = [(peer, count)]
        -> list of received arrays in the order of recvs."""
        L = lib()
        base = Parallel._staged_ops(allgather_host, allreduce_host)

        def xch(ops, st):
            sends, recvs, rptr = [], [], []
            for peer, ptr, n, is_send in ops:
                if is_send:
                    a = np.empty(n)
                    _chk(L.cedr_memcpy_d2h_async(a.ctypes.data, ptr, a.nbytes, st), "d2h")
                    sends.append((peer, a))
                else:
                    recvs.append((peer, n))
                    rptr.append(ptr)
            _chk(L.cedr_stream_synchronize(st), "sync")
            got = [np.ascontiguousarray(g, np.float64) for g in exchange_host(sends, recvs)]
            for ptr, g, (_, n) in zip(rptr, got, recvs):
                assert g.size == n
                _chk(L.cedr_memcpy_h2d_async(ptr, g.ctypes.data, g.nbytes, st), "h2d")
            _chk(L.cedr_stream_synchronize(st), "sync")
            return 0
        return Parallel.callbacks_p2p(rank, size, base[0], base[1], xch)

    @staticmethod
    def _staged_ops(allgather_host, allreduce_host):
        L = lib()

        def staged(host_op):
            def f(s, r, n, st):
                a = np.empty(n)
                _chk(L.cedr_memcpy_d2h_async(a.ctypes.data, s, a.nbytes, st), "d2h")
                _chk(L.cedr_stream_synchronize(st), "sync")
                out = np.ascontiguousarray(host_op(a), np.float64)
                _chk(L.cedr_memcpy_h2d_async(r, out.ctypes.data, out.nbytes, st), "h2d")
                _chk(L.cedr_stream_synchronize(st), "sync")  # 'out' must outlive the copy
                return 0
            return f
        return staged(allgather_host), staged(allreduce_host)

    @staticmethod
    def host_staged(rank, size, allgather_host, allreduce_host):
        """Rehearsal transport: the two collectives run on HOST numpy arrays supplied by the
        caller (e.g. torch.distributed/gloo); device buffers are staged through hipMemcpy on the
        stream the library hands over (it may be a non-blocking stream that the null stream does
        not synchronise with: the send buffer was filled by kernels on it, the readers follow on it).
        allgather_host(send[count]) -> recv[size*count]; allreduce_host(send[count]) -> recv[count]."""
        return Parallel.callbacks(rank, size, *Parallel._staged_ops(allgather_host, allreduce_host))

    def exchange(self, ops, stream=None):
        """One round of point-to-point messages, ops = [(peer, device_ptr, count, is_send), ...]."""
        n = len(ops)
        peers = (C.c_int * n)(*[o[0] for o in ops])
        bufs = (C.c_void_p * n)(*[o[1] for o in ops])
        counts = (C.c_size_t * n)(*[o[2] for o in ops])
        sends = (C.c_int * n)(*[int(o[3]) for o in ops])
        return lib().cedr_parallel_exchange(self.h, n, peers, bufs, counts, sends, stream)

    def allgather(self, send, recv, count, stream=None):
        """recv[r*count:(r+1)*count] = rank r's send[:count] (DeviceArrays or device pointers), enqueued on stream."""
        _chk(lib().cedr_parallel_allgather(self.h, _ptr(send), _ptr(recv), int(count), stream), "allgather")

    def allreduce_sum(self, send, recv, count, stream=None):
        _chk(lib().cedr_parallel_allreduce_sum(self.h, _ptr(send), _ptr(recv), int(count), stream), "allreduce_sum")

    def rank(self):
        return lib().cedr_parallel_rank(self.h)

    def size(self):
        return lib().cedr_parallel_size(self.h)

    def root(self):
        return 0

    def amroot(self):
        return self.rank() == 0

    def __del__(self):
        try:
            lib().cedr_parallel_free(self.h)
        except Exception:
            pass


class Tree:
    """The caller's tree (reference cedr/cedr_tree_caller.hpp:12-30), in array form."""

    def __init__(self, handle):
        self.h = _handle(handle, "Tree")

    @staticmethod
    def make_1d(ncells, nranks=1, pseudorandom=False, imbalanced=False):
        """tree::oned::make_tree over Mesh(ncells) (reference cedr/cedr_tree.cpp:366-420)."""
        return Tree(lib().cedr_tree_make_1d(ncells, nranks, int(pseudorandom), int(imbalanced)))

    @staticmethod
    def from_arrays(root, kid0, kid1, rank, cellidx):
        kid0 = np.ascontiguousarray(kid0, np.int32)
        kid1 = np.ascontiguousarray(kid1, np.int32)
        rank = np.ascontiguousarray(rank, np.int32)
        cellidx = np.ascontiguousarray(cellidx, np.int64)
        return Tree(lib().cedr_tree_create(len(kid0), int(root), kid0.ctypes.data_as(_ip),
                                           kid1.ctypes.data_as(_ip), rank.ctypes.data_as(_ip),
                                           cellidx.ctypes.data_as(_lp)))

    @staticmethod
    def from_arrays_levels(root, kid0, kid1, rank, cellidx, level):
        """A PARTIAL tree (tree::Node::level, reference cedr/cedr_tree_caller.hpp:20-22): nodes above
        the cells may lack kids; level[i] >= 0 wherever something was left out, every rank given."""
        a = [np.ascontiguousarray(v, np.int32) for v in (kid0, kid1, rank)]
        cellidx = np.ascontiguousarray(cellidx, np.int64)
        level = np.ascontiguousarray(level, np.int32)
        return Tree(lib().cedr_tree_create_levels(len(a[0]), int(root), a[0].ctypes.data_as(_ip),
                                                  a[1].ctypes.data_as(_ip), a[2].ctypes.data_as(_ip),
                                                  cellidx.ctypes.data_as(_lp), level.ctypes.data_as(_ip)))

    def prune(self, rank):
        """The partial tree rank `rank` needs of this (full) tree: its nodes, their kids, its ancestors."""
        return Tree(lib().cedr_tree_prune(self.h, int(rank)))

    def nnodes(self):
        return lib().cedr_tree_nnodes(self.h)

    def analyze(self, ncells, rank):
        """tree::analyze -> dict of integer arrays (NodeSets, reference cedr/cedr_tree.hpp:16-97)."""
        h = _handle(lib().cedr_tree_analyze(self.h, ncells, rank), "analyze")
        try:
            return dump_nodesets(h)
        finally:
            lib().cedr_ns_free(h)

    def plan(self, ncells, rank, nranks, max_tile_leaves=0):
        h = _handle(lib().cedr_plan_create(self.h, ncells, rank, nranks, max_tile_leaves), "plan")
        try:
            return dump_plan(h, nranks)
        finally:
            lib().cedr_plan_free(h)

    def __del__(self):
        try:
            lib().cedr_tree_free(self.h)
        except Exception:
            pass


def dump_nodesets(h):
    L = lib()
    nn = L.cedr_ns_nnode(h)
    nodes = np.zeros((nn, 7), np.int32)
    L.cedr_ns_nodes(h, nodes.ctypes.data_as(_ip))
    levels = []
    for il in range(L.cedr_ns_nlevels(h)):
        ln = np.zeros(L.cedr_ns_level_size(h, il), np.int32)
        L.cedr_ns_level_nodes(h, il, ln.ctypes.data_as(_ip))
        me = np.zeros((L.cedr_ns_level_nme(h, il), 3), np.int32)
        kids = np.zeros((L.cedr_ns_level_nkids(h, il), 3), np.int32)
        if len(me):
            L.cedr_ns_level_msgs(h, il, 0, me.ctypes.data_as(_ip))
        if len(kids):
            L.cedr_ns_level_msgs(h, il, 1, kids.ctypes.data_as(_ip))
        levels.append(dict(nodes=ln, me=me, kids=kids))
    return dict(nslots=L.cedr_ns_nslots(h), nodes=nodes, levels=levels,
                check_comm=L.cedr_ns_check_comm(h))


def dump_plan(h, nranks=1):
    L = lib()
    s = (C.c_int64 * 8)()
    L.cedr_plan_summary(h, s)
    out = dict(nlcl=s[0], nslots=s[1], root_slot=s[4], nb_max=s[5], nmyroots=s[6], stride=s[7],
               local=[], top=[])
    for top, key, cnt in ((0, "local", s[2]), (1, "top", s[3])):
        for i in range(cnt):
            sz = (C.c_int * 6)()
            _chk(L.cedr_plan_stage_sizes(h, top, i, sz))
            tiles = np.zeros((sz[0], 10), np.int32)
            kids = np.zeros((sz[2], 2), np.uint16)
            lvlptr = np.zeros(sz[3], np.int32)
            _chk(L.cedr_plan_stage(h, top, i, tiles.ctypes.data_as(_ip),
                                   kids.ctypes.data_as(C.POINTER(C.c_uint16)), lvlptr.ctypes.data_as(_ip)))
            micro = np.zeros(max(sz[4], 1), np.uint16)
            outs = np.zeros(max(sz[5], 1), np.uint16)
            _chk(L.cedr_plan_stage_micro(h, top, i, micro.ctypes.data_as(C.POINTER(C.c_uint16)),
                                         outs.ctypes.data_as(C.POINTER(C.c_uint16))))
            out[key].append(dict(tiles=tiles, kids=kids, lvlptr=lvlptr, max_nodes=sz[1],
                                 micro=micro[:sz[4]], outs=outs[:sz[5]],
                                 top=bool(L.cedr_plan_stage_is_top(h, top, i))))
    out["my_root_slots"], out["my_top_slots"], out["gather_slots"] = plan_exchange(h, nranks)
    info = (C.c_int * 3)()
    L.cedr_plan_levels(h, info)
    out["level_mode"], out["root_owner"], out["steps"] = bool(info[0]), int(info[2]), []
    for i in range(info[1]):  # per-level point-to-point steps (partial trees, scattered ownership)
        sz = (C.c_int * 6)()
        _chk(L.cedr_plan_level_sizes(h, i, sz))
        arrs = [np.zeros(max(n, 1), np.int32) for n in (3 * sz[1], sz[2], sz[3], 3 * sz[4], 3 * sz[5])]
        _chk(L.cedr_plan_level(h, i, *[a.ctypes.data_as(_ip) for a in arrs]))
        out["steps"].append(dict(level=sz[0], nodes=arrs[0][:3 * sz[1]].reshape(-1, 3), out_slots=arrs[1][:sz[2]],
                                 in_slots=arrs[2][:sz[3]], out_msgs=arrs[3][:3 * sz[4]].reshape(-1, 3),
                                 in_msgs=arrs[4][:3 * sz[5]].reshape(-1, 3)))
    out["apex"] = None  # the replicated top as one plain tile, if it is small enough
    sz = (C.c_int * 6)()
    if L.cedr_plan_stage_sizes(h, 2, 0, sz) == 0:
        tiles = np.zeros((sz[0], 10), np.int32)
        kids = np.zeros((sz[2], 2), np.uint16)
        lvlptr = np.zeros(sz[3], np.int32)
        _chk(L.cedr_plan_stage(h, 2, 0, tiles.ctypes.data_as(_ip), kids.ctypes.data_as(C.POINTER(C.c_uint16)),
                               lvlptr.ctypes.data_as(_ip)))
        micro = np.zeros(max(sz[4], 1), np.uint16)
        outs = np.zeros(max(sz[5], 1), np.uint16)
        _chk(L.cedr_plan_stage_micro(h, 2, 0, micro.ctypes.data_as(C.POINTER(C.c_uint16)),
                                     outs.ctypes.data_as(C.POINTER(C.c_uint16))))
        out["apex"] = dict(tiles=tiles, kids=kids, lvlptr=lvlptr, max_nodes=sz[1], micro=micro[:sz[4]],
                           outs=outs[:sz[5]])
    return out


def plan_exchange(h, nranks):
    """(my_root_slots, my_top_slots, gather_slots[nranks][nb_max]) of a plan handle."""
    L = lib()
    s = (C.c_int64 * 8)()
    L.cedr_plan_summary(h, s)
    nm, nb = int(s[6]), int(s[5])
    mr = np.zeros(nm, np.int32)
    mt = np.zeros(nm if s[3] else 0, np.int32)
    gs = np.zeros(nranks * nb if s[3] else 0, np.int32)
    # arrays with zero length still need valid pointers
    pad = np.zeros(1, np.int32)
    L.cedr_plan_exchange(h, (mr if nm else pad).ctypes.data_as(_ip),
                         (mt if len(mt) else pad).ctypes.data_as(_ip),
                         (gs if len(gs) else pad).ctypes.data_as(_ip))
    return mr, mt, gs.reshape(nranks, nb) if len(gs) else gs


class _CDR:
    """Shared CDR protocol (reference cedr/cedr_cdr.hpp:38-107)."""
    _pfx = ""

    def _f(self, name):
        return getattr(lib(), self._pfx + name)

    def declare_tracer(self, problem_type, rhomidx=0):
        _chk(self._f("declare_tracer")(self.h, int(problem_type), int(rhomidx)), "declare_tracer")

    def end_tracer_declarations(self):
        _chk(self._f("end_tracer_declarations")(self.h), "end_tracer_declarations")

    def get_buffers_sizes(self):
        b1, b2 = C.c_size_t(), C.c_size_t()
        _chk(self._f("get_buffers_sizes")(self.h, C.byref(b1), C.byref(b2)), "get_buffers_sizes")
        return b1.value, b2.value

    def set_buffers(self, buf1, buf2):
        self._bufs = (buf1, buf2)
        _chk(self._f("set_buffers")(self.h, _ptr(buf1), _ptr(buf2)), "set_buffers")

    def finish_setup(self):
        _chk(self._f("finish_setup")(self.h), "finish_setup")

    def get_problem_type(self, tracer_idx):
        rc = self._f("get_problem_type")(self.h, int(tracer_idx))
        if rc < 0:
            raise CedrError(_err())
        return rc

    def get_num_tracers(self):
        return self._f("get_num_tracers")(self.h)

    def run(self):
        _chk(self._f("run")(self.h), "run")

    def set_stream(self, stream):
        """hipStream_t (int / None) run() and the bulk set_* / get_Qm calls are enqueued on."""
        _chk(self._f("set_stream")(self.h, stream), "set_stream")
        self._stream = stream


def stream_wait_event(stream, event):
    """hipStreamWaitEvent(stream, event)."""
    _chk(lib().cedr_stream_wait_event(stream, event), "stream_wait_event")


class QLT(_CDR):
    """cedr::qlt::QLT (reference cedr/cedr_qlt.hpp:129-197)."""
    _pfx = "cedr_qlt_"

    def sweep_down_event(self):
        """hipEvent_t run() records just before its first-stage root->leaves sweep (fp64-issue bound): another
        stream that waits for it runs memory-bound work side by side with that sweep."""
        return _handle(lib().cedr_qlt_sweep_down_event(self.h), "sweep_down_event")

    def sweep_mid_event(self):
        """hipEvent_t run() records after its first-stage leaves->root sweep: what follows until the first-stage
        root->leaves sweep are small, latency-bound launches that leave most of the GPU idle."""
        return _handle(lib().cedr_qlt_sweep_mid_event(self.h), "sweep_mid_event")

    def __init__(self, parallel, ncells, tree, prefer_mass_con_to_bounds=False):
        self.p, self.tree = parallel, tree
        self.h = _handle(lib().cedr_qlt_create(parallel.h, int(ncells), tree.h,
                                               int(prefer_mass_con_to_bounds)), "QLT")

    def nlclcells(self):
        return lib().cedr_qlt_nlclcells(self.h)

    def get_owned_glblcells(self):
        g = np.zeros(self.nlclcells(), np.int64)
        _chk(lib().cedr_qlt_get_owned_glblcells(self.h, g.ctypes.data_as(_lp)))
        return g

    def gci2lci(self, gci):
        rc = lib().cedr_qlt_gci2lci(self.h, int(gci))
        if rc < 0:
            raise CedrError(_err())
        return rc

    def get_buffers_sizes_tight(self):
        """What this build touches (doubles): enough for set_buffers, about half of get_buffers_sizes."""
        b1, b2 = C.c_size_t(), C.c_size_t()
        _chk(lib().cedr_qlt_get_buffers_sizes_tight(self.h, C.byref(b1), C.byref(b2)), "get_buffers_sizes_tight")
        return b1.value, b2.value

    def get_device_op(self):
        op = _QltDeviceOp()
        lib().cedr_qlt_get_device_op(self.h, C.byref(op))
        return op

    def set_rhom(self, rhom):
        """Bulk DeviceOp::set_rhom over all local cells; rhom is a device array [cell]."""
        _chk(lib().cedr_qlt_set_rhom_bulk(self.h, _ptr(rhom, self.nlclcells())), "set_rhom")

    def set_Qm(self, Qm, Qm_min, Qm_max, Qm_prev):
        """Bulk DeviceOp::set_Qm; device arrays [tracer][cell] in caller tracer order."""
        n = self.nlclcells() * self.get_num_tracers()
        _chk(lib().cedr_qlt_set_Qm_bulk(self.h, _ptr(Qm, n), _ptr(Qm_min, n), _ptr(Qm_max, n), _ptr(Qm_prev, n)),
             "set_Qm")

    def get_Qm(self, out):
        _chk(lib().cedr_qlt_get_Qm_bulk(self.h, _ptr(out, self.nlclcells() * self.get_num_tracers())), "get_Qm")

    def bind_arrays(self, rhom, Qm, Qm_min, Qm_max, Qm_prev, Qm_out):
        """Zero-copy mode: run() reads the caller's device arrays (rhom[cell]; the others
        [tracer][cell]) and writes the limited masses to Qm_out (may be Qm). rhom=None switches back."""
        nc = self.nlclcells()
        n = nc * self.get_num_tracers()
        f = lambda x, k: _ptr(x, k) if x is not None else None
        _chk(lib().cedr_qlt_bind_arrays(self.h, f(rhom, nc), f(Qm, n), f(Qm_min, n), f(Qm_max, n), f(Qm_prev, n),
                                        f(Qm_out, n)), "bind_arrays")

    def metadata(self):
        nt = self.get_num_tracers()
        a = [np.zeros(nt, np.int32) for _ in range(5)] + [np.zeros(7, np.int32) for _ in range(3)]
        lib().cedr_qlt_metadata(self.h, *[x.ctypes.data_as(_ip) for x in a])
        keys = ["trcr2prob", "bidx2trcr", "trcr2bidx", "trcr2bl2r", "trcr2br2l",
                "prob2trcrptr", "prob2bl2r", "prob2br2l"]
        return dict(zip(keys, a))

    def placement_report(self):
        """What finish_setup did about the placement of its own buffers (CEDR_HIP_PLACE_TRIES)."""
        return (lib().cedr_qlt_placement_report(self.h) or b"").decode()

    def nodesets(self):
        return dump_nodesets(lib().cedr_qlt_nodesets(self.h))

    def plan(self):
        return dump_plan(lib().cedr_qlt_plan(self.h), self.p.size())

    def __del__(self):
        try:
            lib().cedr_qlt_free(self.h)
        except Exception:
            pass


class CAAS(_CDR):
    """cedr::caas::CAAS (reference cedr/cedr_caas.hpp:51-96).

    user_reducer(send_ptr, recv_ptr, nlocal, nfld) -> int mirrors
    CAAS::UserAllReducer::operator() (cedr_caas.hpp:31-40) on device pointers.
    user_reducer may also be a BfbTreeAllReducer object: the production pairing as a type
    (caas::BfbUserAllReducer), which run() fuses with the reducer's first tree stage.
    """
    _pfx = "cedr_caas_"

    def __init__(self, parallel, nlclcells, user_reducer=None, n_accum_in_place=1):
        self.p = parallel
        self.n = int(nlclcells)
        if isinstance(user_reducer, BfbTreeAllReducer):
            self._bfb = user_reducer  # shared with the C++ object; kept alive here as well
            self.h = _handle(lib().cedr_caas_create_bfb(parallel.h, self.n, user_reducer.h,
                                                        int(n_accum_in_place)), "CAAS")
            return
        if user_reducer is not None:
            self._cb = USER_ALLREDUCE_FN(lambda user, s, r, nl, nf: int(user_reducer(s, r, nl, nf) or 0))
        else:
            self._cb = C.cast(None, USER_ALLREDUCE_FN)
        self.h = _handle(lib().cedr_caas_create(parallel.h, int(nlclcells), self._cb,
                                                int(n_accum_in_place), None), "CAAS")
        self.n = int(nlclcells)

    def fused_with_bfb(self):
        """After finish_setup: whether run() fuses the reduce pass with the BFB reducer's first stage."""
        return bool(lib().cedr_caas_fused_with_bfb(self.h))

    def get_device_op(self):
        op = _CaasDeviceOp()
        lib().cedr_caas_get_device_op(self.h, C.byref(op))
        return op

    def set_Qm_source(self, qm_src):
        self._src = qm_src
        _chk(lib().cedr_caas_set_Qm_source(self.h, _ptr(qm_src)))

    def bind_arrays(self, Qm, Qm_min, Qm_max, Qm_prev=None):
        """Zero-copy mode: run() limits the caller's [tracer][cell] device arrays in place
        (Qm is overwritten with the result); bind_arrays(None, None, None) switches back."""
        _chk(lib().cedr_caas_bind_arrays(self.h, _ptr(Qm) if Qm is not None else None,
                                         _ptr(Qm_min) if Qm_min is not None else None,
                                         _ptr(Qm_max) if Qm_max is not None else None,
                                         _ptr(Qm_prev) if Qm_prev is not None else None), "bind_arrays")

    def _copy(self, row0, src, nrows):
        op = self.get_device_op()
        nbytes = 8 * self.n * nrows
        _chk(lib().cedr_memcpy_d2d_async(op.d + 8 * self.n * row0, _ptr(src), nbytes,
                                         getattr(self, "_stream", None)), "copy")

    def set_rhom(self, rhom):
        """Bulk DeviceOp::set_rhom (cedr_caas_inl.hpp:14-19)."""
        self._copy(0, rhom, 1)

    def set_Qm(self, Qm, Qm_min, Qm_max, Qm_prev=None):
        """Bulk DeviceOp::set_Qm (cedr_caas_inl.hpp:22-34): the layouts coincide, so plain copies."""
        op = self.get_device_op()
        nt = op.nt
        self._copy(1, Qm, nt)
        self._copy(1 + nt, Qm_min, nt)
        self._copy(1 + 2 * nt, Qm_max, nt)
        if op.need_conserve:
            if Qm_prev is None:
                raise CedrError("Qm_prev was not provided to set_Qm.")
            self._copy(1 + 3 * nt, Qm_prev, nt)

    def get_Qm(self, out):
        op = self.get_device_op()
        _chk(lib().cedr_memcpy_d2d_async(_ptr(out), op.d + 8 * self.n, 8 * self.n * op.nt,
                                         getattr(self, "_stream", None)))

    def recv_ptr(self):
        p = _vp()
        lib().cedr_caas_get_recv(self.h, C.byref(p))
        return p.value

    def __del__(self):
        try:
            lib().cedr_caas_free(self.h)
        except Exception:
            pass


class BfbTreeAllReducer:
    """cedr::BfbTreeAllReducer (reference cedr/cedr_bfb_tree_allreduce.hpp:14-57)."""

    def __init__(self, parallel, tree, nleaf, nfield):
        self.p, self.tree, self.nfield = parallel, tree, nfield
        self.h = _handle(lib().cedr_bfb_create(parallel.h, tree.h, int(nleaf), int(nfield)), "Bfb")

    def nlocal(self):
        return lib().cedr_bfb_nlocal(self.h)

    def allreduce(self, send, recv, transpose=False):
        _chk(lib().cedr_bfb_allreduce(self.h, _ptr(send), _ptr(recv), int(transpose)), "allreduce")

    def set_stream(self, stream):
        """hipStream_t (int / None) all work is enqueued on; inside CAAS.run use CAAS's stream."""
        _chk(lib().cedr_bfb_set_stream(self.h, stream), "set_stream")

    def __del__(self):
        try:
            lib().cedr_bfb_free(self.h)
        except Exception:
            pass


def synchronize():
    _chk(lib().cedr_device_synchronize(), "synchronize")


def profile_start():
    """Start recording HIP-event timings of every kernel launch of this library."""
    lib().cedr_hip_profile_start()


def profile_stop():
    """Stop and return a list of (kernel name, aux, milliseconds), one per launch."""
    L = lib()
    n = L.cedr_hip_profile_stop()
    out = []
    kid, aux, ms = C.c_int(), C.c_int64(), C.c_float()
    for i in range(n):
        _chk(L.cedr_hip_profile_get(i, C.byref(kid), C.byref(aux), C.byref(ms)), "profile_get")
        out.append((L.cedr_hip_profile_name(kid.value).decode(), aux.value, ms.value))
    return out
