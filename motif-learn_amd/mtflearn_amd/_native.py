"""ctypes binding of ``libzernike_hip.so`` (C ABI: ``include/zernike_hip.h``).

The library is built in-tree by ``__graft_entry__.build()`` / ``csrc/Makefile`` into
``mtflearn_amd/lib/``.  There is deliberately no CPU fallback here: if the shared object is
missing, or no HIP device is visible, every compute call raises ``RuntimeError``.
"""
from __future__ import annotations

import ctypes
import os
import threading
import weakref
from ctypes import POINTER, byref, c_char_p, c_double, c_int, c_int32, c_int64, c_void_p

import numpy as np

ZK_F32, ZK_F64 = 0, 1
ZK_U8, ZK_U16, ZK_I16 = 2, 3, 4   # host-buffer entry points only: widened to float32 on the device (exact)
PATH_AUTO, PATH_GENERIC, PATH_FOLDED, PATH_SEPARABLE, PATH_STREAM, PATH_DIRECT = 0, 1, 2, 3, 4, 5
OP_POINTS, OP_MAPS = 1, 2
XFER_SEND, XFER_RECV, XFER_ALLGATHER, XFER_BCAST = 1, 2, 3, 4
COMM_AUTO, COMM_P2P, COMM_ALLGATHER, COMM_BCAST = 0, 1, 2, 3
COMM_ALGOS = {"": COMM_AUTO, "auto": COMM_AUTO, "p2p": COMM_P2P, "allgather": COMM_ALLGATHER, "bcast": COMM_BCAST}
PATH_NAMES = {PATH_GENERIC: "generic", PATH_FOLDED: "folded", PATH_SEPARABLE: "separable", PATH_STREAM: "stream",
              PATH_DIRECT: "direct"}



class Xfer(ctypes.Structure):
    """``zk_xfer``: one RCCL call of the all-gather schedule (include/zernike_hip.h)."""
    _fields_ = [("op", c_int32), ("peer", c_int32), ("group", c_int32), ("plane", c_int32),
                ("offset", c_int64), ("count", c_int64)]


# MTFLEARN_AMD_LIB: alternative build of the same ABI (e.g. a timing-only ablation variant)
LIB_PATH = os.environ.get("MTFLEARN_AMD_LIB") or os.path.join(
    os.path.dirname(os.path.abspath(__file__)), "lib", "libzernike_hip.so")

# every symbol include/zernike_hip.h declares: (restype, argtypes)
SYMBOLS = {
    "zk_abi_version": (c_int, []),
    "zk_device_count": (c_int, []),
    "zk_last_error_string": (c_char_p, []),
    "zk_plan_create": (c_int, [c_int, c_int, POINTER(c_int32), POINTER(c_int32), POINTER(c_double),
                               c_int, POINTER(c_void_p)]),
    "zk_plan_destroy": (None, [c_void_p]),
    "zk_plan_has_path": (c_int, [c_void_p, c_int, c_int, c_int]),
    "zk_plan_resolved_path": (c_int, [c_void_p, c_int, c_int, c_int64]),
    "zk_plan_supports": (c_int, [c_void_p, c_int, c_int]),
    "zk_plan_disk_pixels": (c_int, [c_void_p]),
    "zk_plan_set_path": (c_int, [c_void_p, c_int]),
    "zk_transform_patches": (c_int, [c_void_p, c_void_p, c_int, c_int64, POINTER(c_double)]),
    "zk_transform_patches_dev": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p]),
    "zk_transform_frame": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, POINTER(c_double)]),
    "zk_transform_frame_dev": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_int64, c_int64,
                                       c_void_p, c_void_p]),
    "zk_transform_points": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, POINTER(c_int32), c_int64,
                                    POINTER(c_double)]),
    "zk_transform_points_dev": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_void_p, c_int64, c_void_p,
                                        c_void_p]),
    "zk_frame_maps": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, POINTER(c_int32), c_int,
                              POINTER(c_int32), c_int, c_int, POINTER(c_double), c_int, c_void_p, c_void_p,
                              c_void_p]),
    "zk_frame_maps_dev": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_int64, c_int64, POINTER(c_int32),
                                  c_int, POINTER(c_int32), c_int, c_int, POINTER(c_double), c_int, c_void_p,
                                  c_void_p, c_void_p, c_void_p]),
    "zk_moment_maps": (c_int, [c_void_p, POINTER(c_double), c_int64, POINTER(c_int32), c_int, POINTER(c_int32), c_int, c_int,
                               POINTER(c_double), c_int, c_void_p, c_void_p, c_void_p]),
    "zk_moment_maps_dev": (c_int, [c_void_p, c_void_p, c_int64, POINTER(c_int32), c_int, POINTER(c_int32), c_int, c_int,
                                   POINTER(c_double), c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "zk_points_maps": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, POINTER(c_int32), c_int64, POINTER(c_int32), c_int,
                               POINTER(c_int32), c_int, c_int, POINTER(c_double), c_int, c_void_p, c_void_p, c_void_p]),
    "zk_transform_frame_dev_strided": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_int64, c_int64,
                                               c_void_p, c_int64, c_void_p]),
    "zk_frame_maps_dev_strided": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_int64, c_int64,
                                          POINTER(c_int32), c_int, POINTER(c_int32), c_int, c_int, POINTER(c_double),
                                          c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "zk_plan_set_host_chunk": (c_int, [c_void_p, c_int64]),
    "zk_plan_release_staging": (c_int, [c_void_p]),
    "zk_host_alloc": (c_int, [c_int64, POINTER(c_void_p)]),
    "zk_host_free": (c_int, [c_void_p]),
    "zk_comm_unique_id": (c_int, [c_void_p]),
    "zk_comm_init_rank": (c_int, [c_int, c_int, c_int, c_void_p, POINTER(c_void_p)]),
    "zk_comm_init_file": (c_int, [c_int, c_int, c_int, c_char_p, c_double, POINTER(c_void_p)]),
    "zk_comm_init_tcp": (c_int, [c_int, c_int, c_int, c_char_p, c_int, c_double, POINTER(c_void_p)]),
    "zk_comm_destroy": (c_int, [c_void_p]),
    "zk_comm_rank": (c_int, [c_void_p]),
    "zk_comm_world": (c_int, [c_void_p]),
    "zk_allgather_rows": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_void_p]),
    "zk_allgather_rows_plan": (c_int, [c_int, c_int, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_int,
                                       POINTER(Xfer), c_int64, POINTER(c_int64)]),
    "zk_comm_join": (c_int, [c_void_p, c_void_p]),
    "zk_comm_allgather_host": (c_int, [c_void_p, c_void_p, c_void_p, c_int64]),
    "zk_autocorr_mean": (c_int, [c_int, c_void_p, c_int, c_int64, c_int64, c_int64, POINTER(c_int32), c_int, c_int,
                                 POINTER(c_double)]),
    "zk_polar_radii": (c_int64, [c_int64, c_int64]),
    "zk_polar_profile": (c_int, [c_int, POINTER(c_double), c_int64, c_int64, c_int64, c_int64, c_int64, c_int,
                                 POINTER(c_double)]),
    "zk_power_spectra": (c_int, [c_int, c_void_p, c_int, c_int64, c_int64, c_int64, POINTER(c_int32), c_int,
                                 POINTER(c_double), POINTER(c_double)]),
    "zk_denoise_fft": (c_int, [c_int, c_void_p, c_int, c_int64, c_int64, c_double, POINTER(c_double)]),
    "zk_wavelet_sigma": (c_int, [c_int, c_void_p, c_int, c_int64, c_int64, POINTER(c_double)]),
    "zk_gram": (c_int, [c_int, POINTER(c_double), c_int64, c_int, POINTER(c_double), POINTER(c_void_p)]),
    "zk_project": (c_int, [c_int, c_void_p, c_int64, c_int, POINTER(c_double), POINTER(c_double), c_int, POINTER(c_double),
                           c_int]),
    "zk_gram_dev": (c_int, [c_int, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "zk_project_dev": (c_int, [c_int, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "zk_rows_create": (c_int, [c_int, POINTER(c_double), c_int64, c_int, POINTER(c_void_p)]),
    "zk_rows_adopt": (c_int, [c_int, c_void_p, c_int64, c_int, POINTER(c_void_p)]),
    "zk_rows_destroy": (c_int, [c_void_p]),
    "zk_rows_data": (c_void_p, [c_void_p]),
    "zk_rows_center": (c_int, [c_void_p, POINTER(c_double), POINTER(c_double), POINTER(c_int64)]),
    "zk_rows_colsum": (c_int, [c_void_p, POINTER(c_double)]),
    "zk_rows_center_at": (c_int, [c_void_p, POINTER(c_double), POINTER(c_double), POINTER(c_int64)]),
    "zk_rows_fetch": (c_int, [c_void_p, POINTER(c_int64), c_int, c_int, POINTER(c_double)]),
    "zk_rows_reset_labels": (c_int, [c_void_p]),
    "zk_rows_labels": (c_int, [c_void_p, POINTER(c_int32)]),
    "zk_rows_labels_dev": (c_void_p, [c_void_p]),
    "zk_kmeans_seed_step": (c_int, [c_void_p, POINTER(c_double), POINTER(c_double), c_int, c_int, POINTER(c_double)]),
    "zk_kmeans_seed_pick": (c_int, [c_void_p, c_int, POINTER(c_double), c_int, POINTER(c_int64)]),
    "zk_kmeans_step": (c_int, [c_void_p, POINTER(c_double), c_int, c_int, POINTER(c_double), POINTER(c_double), POINTER(c_int64)]),
    "zk_rows_profile": (c_int, [c_void_p, c_int]),
    "zk_rows_last_kernel_ms": (c_double, [c_void_p]),
    "zk_kmeans_own_distance": (c_int, [c_void_p, POINTER(c_double), c_int, POINTER(c_double)]),
    "zk_gmm_estep": (c_int, [c_void_p, POINTER(c_double), POINTER(c_double), POINTER(c_double), POINTER(c_double), c_int, c_int, POINTER(c_double)]),
    "zk_gmm_resp_from_labels": (c_int, [c_void_p, c_int]),
    "zk_gmm_moments": (c_int, [c_void_p, c_int, c_int, POINTER(c_double), POINTER(c_double)]),
    "zk_rows_gram": (c_int, [c_void_p, POINTER(c_double), POINTER(c_double)]),
    "zk_rows_knn_correlation": (c_int, [c_void_p, c_int, c_int, c_double, POINTER(c_int64), POINTER(c_double), POINTER(c_double)]),
    "zk_uniform_choice_index": (c_int, [c_int64, c_double, POINTER(c_int64)]),
    "zk_uniform_choice_index_sequential": (c_int, [c_int64, c_double, POINTER(c_int64), POINTER(c_double)]),
    "zk_repeated_sum_f64": (c_double, [c_double, c_int64]),
    "zk_force_layout_stage": (c_int, [POINTER(c_double), c_int64, POINTER(c_int64), POINTER(c_int64), POINTER(c_double), c_int64,
                                      POINTER(c_int64), c_int, c_int64, POINTER(c_double), c_int, c_double, POINTER(c_int64),
                                      POINTER(c_double)]),
    "zk_device_malloc": (c_int, [c_int, c_int64, POINTER(c_void_p)]),
    "zk_device_free": (c_int, [c_int, c_void_p]),
    "zk_device_copy": (c_int, [c_int, c_void_p, c_void_p, c_int64, c_int]),
    "zk_device_synchronize": (c_int, [c_int]),
    "zk_hbm_probe": (c_int, [c_int, c_void_p, c_void_p, c_int64, c_int64, c_int, POINTER(c_double)]),
    "zk_debug_strip3_launches": (c_int64, []),
    "zk_clock_monitor_start": (c_int, [c_int, c_double, POINTER(c_void_p)]),
    "zk_clock_monitor_stop": (c_int, [c_void_p, POINTER(c_double), POINTER(c_double)]),
    "zk_plan_profile": (c_int, [c_void_p, c_int]),
    "zk_plan_profile_read": (c_int, [c_void_p, POINTER(c_int64), POINTER(c_double)]),
    "zk_plan_profile_read_launches": (c_int, [c_void_p, POINTER(c_double), c_int64, POINTER(c_int64)]),
}

_lib = None


def _preload_torch_hip():
    """When PyTorch-ROCm is installed it ships its own ``libamdhip64.so`` (SONAME libamdhip64.so.7).
    Two HIP runtimes in one process do not coexist (the second one finds no device), so bind our
    library to torch's copy: loading it first satisfies our NEEDED libamdhip64.so.7 by SONAME, and a
    later ``import torch`` resolves to the same file.  Without torch the system ROCm runtime is used."""
    if os.environ.get("MTFLEARN_AMD_SYSTEM_HIP"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        for base in (spec.submodule_search_locations or []) if spec else []:
            cand = os.path.join(base, "lib", "libamdhip64.so")
            if os.path.exists(cand):
                ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
                return
    except Exception:
        pass


def load():
    """Load the shared library once; raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C motif-learn_amd/csrc` (hipcc, --offload-arch=gfx950). "
            "mtflearn_amd has no CPU fallback.")
    _preload_torch_hip()
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    msg = load().zk_last_error_string()
    return msg.decode() if msg else ""


def check(code, what):
    if code != 0:
        raise RuntimeError(f"{what} failed with code {code}: {last_error()}")


def device_count():
    n = load().zk_device_count()
    if n < 0:
        raise RuntimeError(f"zk_device_count failed with code {n}: {last_error()}")
    return n


class ClockMonitor:
    """``with ClockMonitor(device) as m: ...`` -- the mean shader clock (``m.ghz``) over the ``m.ms`` the block took, sampled by
    one resident wave (``zk_clock_monitor_start / _stop``).  Inside the block synchronise the STREAM the work runs on, not
    the device: a device-wide synchronisation waits for the monitor's own wave, i.e. for its whole time budget."""

    def __init__(self, device=0, max_ms=1500.0):
        self.device, self.max_ms, self.ghz, self.ms, self._h = device, max_ms, None, None, c_void_p()

    def __enter__(self):
        check(load().zk_clock_monitor_start(int(self.device), float(self.max_ms), byref(self._h)), "zk_clock_monitor_start")
        return self

    def __exit__(self, *exc):
        ghz, ms = c_double(), c_double()
        check(load().zk_clock_monitor_stop(self._h, byref(ghz), byref(ms)), "zk_clock_monitor_stop")
        self.ghz, self.ms = ghz.value, ms.value
        return False


def hbm_probe(device, src_ptr, nbytes, dst_ptr=None, store_per_group=0, reps=5):
    """Milliseconds per pass of a plain stream over device buffers (``zk_hbm_probe``): LDS-DMA read of ``nbytes`` at
    ``src_ptr`` (``dst_ptr`` None), a 16-B-per-lane copy, or the read stream plus ``store_per_group`` bytes written per
    256-KiB group."""
    ms = c_double()
    check(load().zk_hbm_probe(int(device), c_void_p(src_ptr), c_void_p(dst_ptr) if dst_ptr else None, int(nbytes),
                              int(store_per_group), int(reps), byref(ms)), "zk_hbm_probe")
    return ms.value


def dtype_code(dtype):
    """ZK_F32 / ZK_F64 for the two element types the kernels read natively, ZK_U8 / ZK_U16 / ZK_I16 for the narrow
    detector formats the host-buffer entry points widen on the device, else None."""
    if dtype == np.float32:
        return ZK_F32
    if dtype == np.float64:
        return ZK_F64
    return {np.dtype(np.uint8): ZK_U8, np.dtype(np.uint16): ZK_U16, np.dtype(np.int16): ZK_I16}.get(np.dtype(dtype))


class PinnedPool:
    """Page-locked host arrays for results, recycled.

    ``ZPs.transform`` returns arrays of hundreds of MB to GB; filling a fresh ``np.empty`` over PCIe pays a page
    fault per 4 KiB and the runtime's pageable staging.  Arrays from this pool are page-locked (``zk_host_alloc``
    = ``hipHostMalloc``): the device writes them by DMA at link speed.  A block returns to the pool when the last
    NumPy view of it is garbage-collected and is handed out again for the next result of a fitting size, so
    repeated calls neither allocate nor fault.  ``MTFLEARN_AMD_PINNED_MB`` caps the idle bytes the pool keeps
    (default 16384; 0 disables the pool: results are plain ``np.empty`` arrays)."""

    GRAIN = 2 << 20

    def __init__(self):
        self._free = []            # [(capacity, ptr)]
        self._lock = threading.Lock()
        self.max_idle = int(os.environ.get("MTFLEARN_AMD_PINNED_MB", "16384")) << 20
        self.min_bytes = 1 << 20   # smaller results are not worth a pinned block

    def empty(self, shape, dtype=np.float64):
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        if self.max_idle <= 0 or nbytes < self.min_bytes:
            return np.empty(shape, dtype=dtype)
        cap = -(-nbytes // self.GRAIN) * self.GRAIN
        ptr = None
        with self._lock:
            fits = [k for k, (c, _) in enumerate(self._free) if cap <= c <= 2 * cap]
            if fits:
                cap, ptr = self._free.pop(min(fits, key=lambda k: self._free[k][0]))
        if ptr is None:
            handle = c_void_p()
            if load().zk_host_alloc(cap, byref(handle)) != 0:     # no pinned memory to be had: pageable result
                return np.empty(shape, dtype=dtype)
            ptr = handle.value
        buf = (ctypes.c_char * cap).from_address(ptr)
        weakref.finalize(buf, self._release, cap, ptr)
        return np.ndarray(shape, dtype=dtype, buffer=buf)

    def _release(self, cap, ptr):
        with self._lock:
            if sum(c for c, _ in self._free) + cap <= self.max_idle:
                self._free.append((cap, ptr))
                return
        if _lib is not None:
            _lib.zk_host_free(c_void_p(ptr))

    def trim(self):
        """Give every idle block back to the system."""
        with self._lock:
            blocks, self._free = self._free, []
        for _, ptr in blocks:
            load().zk_host_free(c_void_p(ptr))


pinned = PinnedPool()


class Plan:
    """Owns one ``zk_plan`` (device tables for one basis on one GPU)."""

    def __init__(self, basis, n, m, device=0):
        lib = load()
        basis = np.ascontiguousarray(basis, dtype=np.float64)
        self.n_poly, self.size = int(basis.shape[0]), int(basis.shape[1])
        self.device = int(device)
        n32 = np.ascontiguousarray(n, dtype=np.int32)
        m32 = np.ascontiguousarray(m, dtype=np.int32)
        if device_count() == 0:
            raise RuntimeError("no HIP device visible: mtflearn_amd computes on MI355X only "
                               "(there is no CPU fallback)")
        handle = c_void_p()
        check(lib.zk_plan_create(self.size, self.n_poly, n32.ctypes.data_as(POINTER(c_int32)),
                                 m32.ctypes.data_as(POINTER(c_int32)),
                                 basis.ctypes.data_as(POINTER(c_double)), self.device, byref(handle)),
              "zk_plan_create")
        self._h = handle
        self._lib = lib

    def close(self):
        if getattr(self, "_h", None):
            self._lib.zk_plan_destroy(self._h)
            self._h = None

    __del__ = close

    # -- introspection / control -------------------------------------------------------
    def has_path(self, mode, dtype_code_, path):
        """True if kernel family ``path`` exists for mode (0 batch, 1 frame) and element type."""
        return bool(self._lib.zk_plan_has_path(self._h, mode, dtype_code_, path))

    def supports(self, op, dtype_code_):
        """True if the plan has the key-point (OP_POINTS) / fused-maps (OP_MAPS) kernel for the element type."""
        return bool(self._lib.zk_plan_supports(self._h, op, dtype_code_))

    def best_path(self, mode, dtype_code_, n_units=1 << 20):
        """The kernel family a transform runs on with the plan's current setting (what PATH_AUTO resolves to for a
        batch of ``n_units`` patches / a frame)."""
        return self._lib.zk_plan_resolved_path(self._h, mode, dtype_code_, n_units)

    @property
    def disk_pixels(self):
        return self._lib.zk_plan_disk_pixels(self._h)

    def set_path(self, path):
        check(self._lib.zk_plan_set_path(self._h, path), "zk_plan_set_path")

    def profile(self, enable=True):
        check(self._lib.zk_plan_profile(self._h, int(enable)), "zk_plan_profile")

    def profile_read(self):
        launches, ms = c_int64(), c_double()
        check(self._lib.zk_plan_profile_read(self._h, byref(launches), byref(ms)), "zk_plan_profile_read")
        return launches.value, ms.value

    def profile_read_launches(self, cap=65536):
        """Per-launch kernel times (ms) recorded since the last read, in launch order; raises if more than ``cap``
        launches were recorded (the library drops what does not fit -- a truncated list would skew every statistic)."""
        buf = (c_double * cap)()
        n = c_int64()
        check(self._lib.zk_plan_profile_read_launches(self._h, buf, cap, byref(n)), "zk_plan_profile_read_launches")
        if n.value > cap:
            raise RuntimeError(f"profile_read_launches: {n.value} launches recorded, room for {cap}")
        return list(buf[:n.value])

    # -- host-buffer entry points --------------------------------------------------------
    def transform_patches(self, patches):
        code = dtype_code(patches.dtype)
        out = pinned.empty((patches.shape[0], self.n_poly))
        check(self._lib.zk_transform_patches(self._h, patches.ctypes.data_as(c_void_p), code,
                                             patches.shape[0], out.ctypes.data_as(POINTER(c_double))),
              "zk_transform_patches")
        return out

    def transform_frame(self, image):
        code = dtype_code(image.dtype)
        h, w = image.shape
        out = pinned.empty((self.n_poly, h, w))
        check(self._lib.zk_transform_frame(self._h, image.ctypes.data_as(c_void_p), code, h, w,
                                           out.ctypes.data_as(POINTER(c_double))),
              "zk_transform_frame")
        return out

    def transform_points(self, image, points):
        """Moments of the windows at integer (x, y) points of a host frame -> (N, n_poly) float64."""
        code = dtype_code(image.dtype)
        h, w = image.shape
        pts = np.ascontiguousarray(points, dtype=np.int32).reshape(-1, 2)
        out = pinned.empty((pts.shape[0], self.n_poly))
        check(self._lib.zk_transform_points(self._h, image.ctypes.data_as(c_void_p), code, h, w,
                                            pts.ctypes.data_as(POINTER(c_int32)), pts.shape[0],
                                            out.ctypes.data_as(POINTER(c_double))), "zk_transform_points")
        return out

    def frame_maps(self, image, n_complex, folds=None, m_unselect=(0, 1), p=2, theta=None, want_abs=True):
        """Fused frame -> (rot_maps, |Z^c|, mirror_map); any of them None when not requested."""
        code = dtype_code(image.dtype)
        h, w = image.shape
        folds32 = np.ascontiguousarray(folds if folds is not None else [], dtype=np.int32)
        unsel32 = np.ascontiguousarray(m_unselect, dtype=np.int32)
        rot = pinned.empty((len(folds32), h, w)) if len(folds32) else None
        ab = pinned.empty((n_complex, h, w)) if want_abs else None
        th = None if theta is None else np.ascontiguousarray(theta, dtype=np.float64)
        mir = pinned.empty((h, w)) if th is not None else None
        ptr = lambda a: a.ctypes.data_as(c_void_p) if a is not None else None
        check(self._lib.zk_frame_maps(
            self._h, image.ctypes.data_as(c_void_p), code, h, w,
            folds32.ctypes.data_as(POINTER(c_int32)), len(folds32),
            unsel32.ctypes.data_as(POINTER(c_int32)), len(unsel32), 2 if p == 2 else 0,
            th.ctypes.data_as(POINTER(c_double)) if th is not None else None, 0 if th is None else len(th),
            ptr(rot), ptr(ab), ptr(mir)), "zk_frame_maps")
        return rot, ab, mir

    def _rows_maps(self, n_rows, n_complex, folds, m_unselect, p, theta, want_abs, call):
        folds32 = np.ascontiguousarray(folds if folds is not None else [], dtype=np.int32)
        unsel32 = np.ascontiguousarray(m_unselect, dtype=np.int32)
        rot = pinned.empty((n_rows, len(folds32))) if len(folds32) else None
        ab = pinned.empty((n_rows, n_complex)) if want_abs else None
        th = None if theta is None else np.ascontiguousarray(theta, dtype=np.float64)
        mir = pinned.empty((n_rows,)) if th is not None else None
        ptr = lambda a: a.ctypes.data_as(c_void_p) if a is not None else None
        call(folds32.ctypes.data_as(POINTER(c_int32)), len(folds32),
             unsel32.ctypes.data_as(POINTER(c_int32)), len(unsel32), 2 if p == 2 else 0,
             th.ctypes.data_as(POINTER(c_double)) if th is not None else None, 0 if th is None else len(th),
             ptr(rot), ptr(ab), ptr(mir))
        return rot, ab, mir

    def moment_maps(self, moments, n_complex, folds=None, m_unselect=(0, 1), p=2, theta=None, want_abs=True):
        """(N, n_poly) host moments -> (rot (N, n_folds), |Z^c| (N, N_c), mirror (N)); None when not requested."""
        mom = np.ascontiguousarray(moments, dtype=np.float64)
        return self._rows_maps(mom.shape[0], n_complex, folds, m_unselect, p, theta, want_abs,
                               lambda *tail: check(self._lib.zk_moment_maps(
                                   self._h, mom.ctypes.data_as(POINTER(c_double)), mom.shape[0], *tail), "zk_moment_maps"))

    def points_maps(self, image, points, n_complex, folds=None, m_unselect=(0, 1), p=2, theta=None, want_abs=True):
        """Windows at integer (x, y) points of a host frame -> the same three outputs; the moments stay on the device."""
        code = dtype_code(image.dtype)
        h, w = image.shape
        pts = np.ascontiguousarray(points, dtype=np.int32).reshape(-1, 2)
        return self._rows_maps(pts.shape[0], n_complex, folds, m_unselect, p, theta, want_abs,
                               lambda *tail: check(self._lib.zk_points_maps(
                                   self._h, image.ctypes.data_as(c_void_p), code, h, w,
                                   pts.ctypes.data_as(POINTER(c_int32)), pts.shape[0], *tail), "zk_points_maps"))

    def frame_maps_dev(self, image_ptr, code, height, width, row0, n_rows, folds, m_unselect, p, theta,
                       rot_ptr, abs_ptr, mirror_ptr, stream=0, plane_stride=None):
        folds32 = np.ascontiguousarray(folds if folds is not None else [], dtype=np.int32)
        unsel32 = np.ascontiguousarray(m_unselect, dtype=np.int32)
        th = None if theta is None else np.ascontiguousarray(theta, dtype=np.float64)
        head = (self._h, c_void_p(image_ptr), code, height, width, row0, n_rows,
                folds32.ctypes.data_as(POINTER(c_int32)), len(folds32),
                unsel32.ctypes.data_as(POINTER(c_int32)), len(unsel32), 2 if p == 2 else 0,
                th.ctypes.data_as(POINTER(c_double)) if th is not None else None, 0 if th is None else len(th),
                c_void_p(rot_ptr), c_void_p(abs_ptr), c_void_p(mirror_ptr))
        if plane_stride is None:
            check(self._lib.zk_frame_maps_dev(*head, c_void_p(stream)), "zk_frame_maps_dev")
        else:
            check(self._lib.zk_frame_maps_dev_strided(*head, plane_stride, c_void_p(stream)),
                  "zk_frame_maps_dev_strided")

    # -- device-pointer entry points (bench / multi-GPU harness) ------------------------------
    def transform_patches_dev(self, patches_ptr, code, n_patches, out_ptr, stream=0):
        check(self._lib.zk_transform_patches_dev(self._h, c_void_p(patches_ptr), code, n_patches,
                                                 c_void_p(out_ptr), c_void_p(stream)),
              "zk_transform_patches_dev")

    def transform_frame_dev(self, image_ptr, code, height, width, row0, n_rows, out_ptr, stream=0, plane_stride=None):
        """Row band -> (n_poly, n_rows, W) at ``out_ptr``; with ``plane_stride`` (doubles) the planes are that far
        apart, i.e. the band is written in place into a larger (n_poly, H, W) array."""
        if plane_stride is None:
            check(self._lib.zk_transform_frame_dev(self._h, c_void_p(image_ptr), code, height, width, row0,
                                                   n_rows, c_void_p(out_ptr), c_void_p(stream)),
                  "zk_transform_frame_dev")
        else:
            check(self._lib.zk_transform_frame_dev_strided(self._h, c_void_p(image_ptr), code, height, width, row0,
                                                           n_rows, c_void_p(out_ptr), plane_stride, c_void_p(stream)),
                  "zk_transform_frame_dev_strided")

    def set_host_chunk(self, chunk_bytes):
        check(self._lib.zk_plan_set_host_chunk(self._h, int(chunk_bytes)), "zk_plan_set_host_chunk")

    def release_staging(self):
        """Free the staging buffers the host-buffer entry points have grown (re-created on demand)."""
        check(self._lib.zk_plan_release_staging(self._h), "zk_plan_release_staging")


def allgather_rows_plan(rank, world, n_planes, height, width, rows_per_rank, row_off, n_rows, algo=COMM_AUTO):
    """The schedule ``zk_allgather_rows`` executes on rank ``rank`` of ``world``, as a list of
    ``(op, peer, group, plane, offset, count)`` tuples in issue order (``zk_allgather_rows_plan``; pure host code)."""
    lib = load()
    if isinstance(algo, str):
        algo = COMM_ALGOS[algo]
    need = c_int64()
    args = (int(rank), int(world), int(n_planes), int(height), int(width), int(rows_per_rank), int(row_off), int(n_rows),
            int(algo))
    check(lib.zk_allgather_rows_plan(*args, None, 0, byref(need)), "zk_allgather_rows_plan")
    buf = (Xfer * max(1, need.value))()
    check(lib.zk_allgather_rows_plan(*args, buf, need.value, byref(need)), "zk_allgather_rows_plan")
    return [(x.op, x.peer, x.group, x.plane, x.offset, x.count) for x in buf[:need.value]]


class Comm:
    """Owns one ``zk_comm``: this process's endpoint of the RCCL communicator (one process per GPU).

    Rendezvous: ``Comm(device, rank, world, path=...)`` -- ranks of one node meet through a file rank 0
    writes; ``host=/port=`` -- rank 0 listens on a TCP port; ``unique_id=`` -- the caller moved the
    128-byte id from rank 0 (``Comm.unique_id()``) to the others by its own means."""

    ID_BYTES = 128

    def __init__(self, device, rank, world, path=None, host=None, port=None, unique_id=None, timeout=120.0):
        lib = load()
        handle = c_void_p()
        self.device, self.rank, self.world = int(device), int(rank), int(world)
        if unique_id is not None:
            buf = ctypes.create_string_buffer(bytes(unique_id), self.ID_BYTES)
            check(lib.zk_comm_init_rank(self.device, self.rank, self.world, buf, byref(handle)), "zk_comm_init_rank")
        elif path is not None:
            check(lib.zk_comm_init_file(self.device, self.rank, self.world, os.fsencode(path), float(timeout),
                                        byref(handle)), "zk_comm_init_file")
        elif port is not None:
            check(lib.zk_comm_init_tcp(self.device, self.rank, self.world, (host or "127.0.0.1").encode(), int(port),
                                       float(timeout), byref(handle)), "zk_comm_init_tcp")
        else:
            raise ValueError("Comm needs one of path=, port= or unique_id=")
        self._h, self._lib = handle, lib

    def ranks_seen(self):
        """World size as the communicator inside the library holds it (``zk_comm_world``)."""
        return int(self._lib.zk_comm_world(self._h))

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(Comm.ID_BYTES)
        check(load().zk_comm_unique_id(buf), "zk_comm_unique_id")
        return buf.raw

    def close(self):
        if getattr(self, "_h", None):
            self._lib.zk_comm_destroy(self._h)
            self._h = None

    __del__ = close

    def allgather_rows(self, full_ptr, n_planes, height, width, rows_per_rank, row_off, n_rows, stream=0):
        """In-place all-gather of the row window ``[row_off, row_off+n_rows)`` of every rank's block of the
        ``(n_planes, height, width)`` float64 array at ``full_ptr`` (see ``zk_allgather_rows``)."""
        check(self._lib.zk_allgather_rows(self._h, c_void_p(full_ptr), n_planes, height, width, rows_per_rank,
                                          row_off, n_rows, c_void_p(stream)), "zk_allgather_rows")

    def join(self, stream=0):
        check(self._lib.zk_comm_join(self._h, c_void_p(stream)), "zk_comm_join")

    def allgather_host(self, payload: bytes):
        """Blocking all-gather of the same number of host bytes from every rank; returns the list of every rank's bytes."""
        n = len(payload)
        send = ctypes.create_string_buffer(payload, n)
        recv = ctypes.create_string_buffer(n * self.world)
        check(self._lib.zk_comm_allgather_host(self._h, send, recv, n), "zk_comm_allgather_host")
        return [recv.raw[r * n:(r + 1) * n] for r in range(self.world)]


class _DeviceRef:
    """What ``array.device`` answers for a :class:`DeviceArray` (``.index``, ``.type``: the attributes the drivers read)."""
    type = "cuda"

    def __init__(self, index):
        self.index = int(index)

    def __repr__(self):
        return f"device(index={self.index})"


class DeviceArray:
    """A C-contiguous array in device memory, allocated through the library (``zk_device_malloc``) -- what the device entry
    points and the sharded drivers of :mod:`mtflearn_amd.distributed` run on when torch is not used at all.  It offers the few
    attributes those functions read from an operand (``shape``, ``dtype``, ``device.index``, ``data_ptr()``, ``numel()``,
    ``element_size()``, ``is_cuda``, ``is_contiguous()``) and views of whole leading-axis items (``a[i]``, ``a[lo:hi]``).
    Work on it runs on the device's default stream (stream 0)."""
    is_cuda = True

    def __init__(self, shape, dtype=np.float64, device=0, _base=None, _ptr=0):
        self.shape = tuple(int(v) for v in (shape if hasattr(shape, "__len__") else (shape,)))
        self.dtype = np.dtype(dtype)
        self.device = _DeviceRef(device)
        self._base = _base                       # the owning array of a view
        if _base is None:
            ptr = c_void_p()
            check(load().zk_device_malloc(self.device.index, max(self.nbytes, 16), byref(ptr)), "zk_device_malloc")
            self._ptr = int(ptr.value)
        else:
            self._ptr = int(_ptr)

    # ---- the tensor-like surface ----
    @property
    def nbytes(self):
        return self.numel() * self.dtype.itemsize

    def numel(self):
        return int(np.prod(self.shape, dtype=np.int64)) if self.shape else 1

    def element_size(self):
        return self.dtype.itemsize

    def data_ptr(self):
        return self._ptr

    def is_contiguous(self):
        return True

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, key):
        if not self.shape:
            raise IndexError("a 0-d DeviceArray has no items")
        n = self.shape[0]
        item = int(np.prod(self.shape[1:], dtype=np.int64)) * self.dtype.itemsize
        owner = self if self._base is None else self._base
        if isinstance(key, (int, np.integer)):
            i = int(key) + (n if key < 0 else 0)
            if not 0 <= i < n:
                raise IndexError(f"index {key} out of range for axis 0 with size {n}")
            return DeviceArray(self.shape[1:], self.dtype, self.device.index, _base=owner, _ptr=self._ptr + i * item)
        if isinstance(key, slice):
            lo, hi, step = key.indices(n)
            if step != 1:
                raise IndexError("only contiguous slices of the leading axis are views")
            hi = max(hi, lo)
            return DeviceArray((hi - lo,) + self.shape[1:], self.dtype, self.device.index, _base=owner, _ptr=self._ptr + lo * item)
        raise IndexError("DeviceArray views take an integer or a contiguous slice of the leading axis")

    # ---- host <-> device ----
    @classmethod
    def from_numpy(cls, array, device=0):
        array = np.ascontiguousarray(array)
        out = cls(array.shape, array.dtype, device)
        if array.nbytes:
            check(load().zk_device_copy(out.device.index, c_void_p(out._ptr), array.ctypes.data_as(c_void_p), array.nbytes, 1), "zk_device_copy")
        return out

    def numpy(self):
        """A host copy (the device is synchronised first: pending work on any stream of it has written its results)."""
        check(load().zk_device_synchronize(self.device.index), "zk_device_synchronize")
        out = np.empty(self.shape, dtype=self.dtype)
        if out.nbytes:
            check(load().zk_device_copy(self.device.index, out.ctypes.data_as(c_void_p), c_void_p(self._ptr), out.nbytes, 2), "zk_device_copy")
        return out

    def fill_(self, value):
        """Fill with a constant (through a host array: a test / set-up helper, not a fast path)."""
        host = np.full(self.shape, value, dtype=self.dtype)
        if host.nbytes:
            check(load().zk_device_copy(self.device.index, c_void_p(self._ptr), host.ctypes.data_as(c_void_p), host.nbytes, 1), "zk_device_copy")
        return self

    def close(self):
        if self._base is None and getattr(self, "_ptr", 0):
            load().zk_device_free(self.device.index, c_void_p(self._ptr))
            self._ptr = 0

    __del__ = close
