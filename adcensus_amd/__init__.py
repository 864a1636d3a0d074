"""adcensus_amd -- MI355X-native AD-Census stereo matcher (host-side Python mirror).

The product is the C-ABI shared library ``adcensus_amd/lib/libadcensus_hip.so`` (hand-written HIP
kernels for gfx950, built by ``adcensus_amd/csrc/Makefile``) plus the C++ facade
``include/ADCensusStereo.h``.  This module is a thin ctypes mirror of that facade
(``ADCensusStereo.Initialize / Match / Reset``, ``ADCensusOption`` -- same names, argument meaning
and error behaviour as the reference's ADCensusStereo.h:14-41 / adcensus_types.h:45-75) used by the
tests and by bench.py.  There is NO CPU fallback: if the HIP library is missing or no GPU is
visible, construction / Initialize fails loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ADC_HIP_LIB") or os.path.join(_HERE, "lib", "libadcensus_hip.so")  # (override: A/B builds, tools/)

# stage / buffer ids (include/adcensus_c_api.h)
STAGES = ["cost", "arms", "aggregate", "scanline", "wta", "refine"]
(BUF_GRAY_LEFT, BUF_GRAY_RIGHT, BUF_CENSUS_LEFT, BUF_CENSUS_RIGHT, BUF_ARMS, BUF_SUPCOUNT_H, BUF_SUPCOUNT_V,
 BUF_VOLUME_A, BUF_DISP_LEFT, BUF_DISP_RIGHT, BUF_OUTLIER_LABEL) = range(11)
(RUN_GRAY_CENSUS, RUN_COST, RUN_ARMS, RUN_AGGREGATE, RUN_SCANLINE, RUN_WTA, RUN_LRCHECK, RUN_REGION_VOTING,
 RUN_INTERPOLATION, RUN_DISCONTINUITY, RUN_MEDIAN) = range(11)
MAX_DISP_RANGE = 2047
PAPER_CENSUS5X5, PAPER_SO_SUM, PAPER_RIGHT_ARMS = 1, 2, 4  # adc_set_paper_modes (opt-in, not the reference's behaviour)


class ADCensusOption(C.Structure):
    """Mirror of struct ADCensusOption (adcensus_types.h:45-75) == adc_option of the C ABI."""
    _fields_ = [
        ("min_disparity", C.c_int32), ("max_disparity", C.c_int32),
        ("lambda_ad", C.c_int32), ("lambda_census", C.c_int32),
        ("cross_L1", C.c_int32), ("cross_L2", C.c_int32),
        ("cross_t1", C.c_int32), ("cross_t2", C.c_int32),
        ("so_p1", C.c_float), ("so_p2", C.c_float),
        ("so_tso", C.c_int32), ("irv_ts", C.c_int32),
        ("irv_th", C.c_float), ("lrcheck_thres", C.c_float),
        ("do_lr_check", C.c_uint8), ("do_filling", C.c_uint8),
        ("do_discontinuity_adjustment", C.c_uint8), ("reserved_", C.c_uint8),
    ]

    def __init__(self, **kw):
        super().__init__()
        lib().adc_option_default(C.byref(self))
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError(k)
            setattr(self, k, v)


_lib = None


def lib():
    """Loads the C-ABI library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("HIP library missing: %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i32, u8p = C.c_void_p, C.c_int32, C.c_void_p
    L.adc_option_default.argtypes = [C.POINTER(ADCensusOption)]
    L.adc_option_default.restype = None
    L.adc_device_count.restype = C.c_int
    L.adc_version.restype = C.c_char_p
    L.adc_last_error.restype = C.c_char_p
    L.adc_create.argtypes = [i32, i32, C.POINTER(ADCensusOption), C.c_int]
    L.adc_create.restype = vp
    L.adc_destroy.argtypes = [vp]
    L.adc_destroy.restype = None
    for name in ("adc_match", "adc_match_async"):
        getattr(L, name).argtypes = [vp, u8p, u8p, vp]
        getattr(L, name).restype = C.c_int
    L.adc_match_device.argtypes = [vp, vp, vp, vp]
    L.adc_match_device.restype = C.c_int
    L.adc_wait.argtypes = [vp]
    L.adc_wait.restype = C.c_int
    L.adc_stage_name.argtypes = [C.c_int]
    L.adc_stage_name.restype = C.c_char_p
    L.adc_set_profiling.argtypes = [vp, C.c_int]
    L.adc_set_profiling.restype = None
    L.adc_set_verbose.argtypes = [vp, C.c_int]
    L.adc_set_verbose.restype = None
    L.adc_get_stage_ms.argtypes = [vp, C.POINTER(C.c_float), C.c_int]
    L.adc_get_stage_ms.restype = C.c_int
    L.adc_get_aggregate_pass_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    L.adc_get_aggregate_pass_ms.restype = C.c_int
    L.adc_get_aggregate_info.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.adc_get_aggregate_info.restype = C.c_int
    L.adc_get_aggregate_kernel.argtypes = [vp]
    L.adc_get_aggregate_kernel.restype = C.c_char_p
    L.adc_host_register.argtypes = [vp, C.c_size_t]
    L.adc_host_register.restype = C.c_int
    L.adc_host_unregister.argtypes = [vp]
    L.adc_host_unregister.restype = C.c_int
    L.adc_set_paper_modes.argtypes = [vp, C.c_uint32]
    L.adc_set_paper_modes.restype = C.c_int
    L.adc_get_stream.argtypes = [vp]
    L.adc_get_stream.restype = vp
    L.adc_device_synchronize.restype = C.c_int
    L.adc_device_malloc.argtypes = [C.c_size_t]
    L.adc_device_malloc.restype = vp
    L.adc_device_free.argtypes = [vp]
    L.adc_device_free.restype = None
    L.adc_memcpy_h2d.argtypes = [vp, vp, C.c_size_t]
    L.adc_memcpy_h2d.restype = C.c_int
    L.adc_memcpy_d2h.argtypes = [vp, vp, C.c_size_t]
    L.adc_memcpy_d2h.restype = C.c_int
    L.adc_device_copy_ms.argtypes = [vp, vp, C.c_size_t, C.c_int]
    L.adc_device_copy_ms.restype = C.c_double
    if hasattr(L, "adc_device_copy_kernel_ms"):  # (absent from A/B builds of older revisions, ADC_HIP_LIB)
        L.adc_device_copy_kernel_ms.argtypes = [vp, vp, C.c_size_t, C.c_int]
        L.adc_device_copy_kernel_ms.restype = C.c_double
    L.adc_debug_read.argtypes = [vp, C.c_int, vp]
    L.adc_debug_read.restype = C.c_int
    L.adc_debug_write.argtypes = [vp, C.c_int, vp]
    L.adc_debug_write.restype = C.c_int
    L.adc_debug_set_images.argtypes = [vp, u8p, u8p]
    L.adc_debug_set_images.restype = C.c_int
    L.adc_debug_run.argtypes = [vp, C.c_int, C.c_int]
    L.adc_debug_run.restype = C.c_int
    L.adc_farm_create.argtypes = [i32, i32, C.POINTER(ADCensusOption), C.c_int, C.c_int]
    L.adc_farm_create.restype = vp
    L.adc_farm_destroy.argtypes = [vp]
    L.adc_farm_destroy.restype = None
    L.adc_farm_submit.argtypes = [vp, u8p, u8p, vp, C.POINTER(C.c_uint64)]
    L.adc_farm_submit.restype = C.c_int
    L.adc_farm_wait.argtypes = [vp, C.c_uint64]
    L.adc_farm_wait.restype = C.c_int
    L.adc_farm_drain.argtypes = [vp]
    L.adc_farm_drain.restype = C.c_int64
    L.adc_debug_counter.argtypes = [vp, C.c_int]
    L.adc_debug_counter.restype = C.c_int64
    L.adc_debug_voting_stats.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.adc_debug_voting_stats.restype = C.c_int
    _lib = L
    return L


def device_count():
    return lib().adc_device_count()


def last_error():
    return lib().adc_last_error().decode()


def host_register(arr):
    """Page-locks a numpy array for DMA straight from / to it (adc_host_register); call host_unregister before it dies."""
    if lib().adc_host_register(arr.ctypes.data, arr.nbytes) != 0:
        raise RuntimeError("adc_host_register failed: " + last_error())


def host_unregister(arr):
    lib().adc_host_unregister(arr.ctypes.data)


def _img(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a


class PreviousPairFailed(RuntimeError):
    """adc_farm_submit returned ADC_FARM_PREVIOUS_FAILED: the pair that occupied the pipeline before (`failed_ticket`) failed
    while it was collected; the NEW pair is in flight all the same and `ticket` is its ticket."""

    def __init__(self, ticket, failed_ticket, message):
        RuntimeError.__init__(self, message)
        self.ticket, self.failed_ticket = int(ticket), int(failed_ticket)


class PairFarm:
    """Persistent farm of `pipelines` matchers of one geometry on one device (adc_farm_* of the C ABI): submit() enqueues
    a whole Match asynchronously from host buffers, results land in the caller's arrays in submission order."""

    def __init__(self, width, height, option, device=-1, pipelines=3):
        self.width, self.height = int(width), int(height)
        self.pipelines = int(pipelines)
        self._f = lib().adc_farm_create(self.width, self.height, C.byref(option), int(device), int(pipelines))
        if not self._f:
            raise RuntimeError("adc_farm_create failed: " + last_error())
        self._keep = {}

    def submit(self, img_left, img_right, disp_left):
        l, r = _img(img_left), _img(img_right)
        assert l.size == self.width * self.height * 3 and r.size == l.size
        assert disp_left.dtype == np.float32 and disp_left.flags["C_CONTIGUOUS"] and disp_left.size == self.width * self.height
        t = C.c_uint64(0)
        rc = lib().adc_farm_submit(self._f, l.ctypes.data, r.ctypes.data, disp_left.ctypes.data, C.byref(t))
        if rc not in (0, 3):
            raise RuntimeError("adc_farm_submit failed (%d): %s" % (rc, last_error()))
        ticket = int(t.value)
        self._keep[ticket] = disp_left  # the output array must stay alive until the pair is delivered
        self._keep.pop(ticket - self.pipelines, None)  # submit() has just delivered the pair that held this pipeline before
        if rc == 3:  # ADC_FARM_PREVIOUS_FAILED: the new pair IS in flight (ticket), the pipeline's previous pair failed
            raise PreviousPairFailed(ticket, ticket - self.pipelines, "adc_farm_submit: " + last_error())
        return ticket

    def wait(self, ticket):
        if lib().adc_farm_wait(self._f, int(ticket)) != 0:
            raise RuntimeError("adc_farm_wait failed: " + last_error())
        self._keep.pop(int(ticket), None)

    def drain(self):
        n = int(lib().adc_farm_drain(self._f))
        if n < 0:
            raise RuntimeError("adc_farm_drain failed: " + last_error())
        self._keep.clear()
        return n

    def close(self):
        if self._f:
            lib().adc_farm_destroy(self._f)
            self._f = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ADCensusStereo:
    """Mirror of class ADCensusStereo (ADCensusStereo.h:14-41).

    Initialize(width, height, option) -> bool ; Match(img_left, img_right, disp_left) -> bool ;
    Reset(width, height, option) -> bool.  Images: uint8 [H][W][3] BGR; disp_left: float32 [H][W],
    caller-allocated, filled in place.
    """

    def __init__(self, device=-1):
        self._h = None
        self._device = device
        self.width = self.height = 0
        self.option = None

    # -- lifetime ------------------------------------------------------------------------------
    def Initialize(self, width, height, option):
        self.Release()
        self.width, self.height, self.option = int(width), int(height), option
        h = lib().adc_create(int(width), int(height), C.byref(option), int(self._device))
        self._h = h
        return bool(h)

    def Reset(self, width, height, option):
        self.Release()
        return self.Initialize(width, height, option)

    def Release(self):
        if self._h:
            lib().adc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.Release()
        except Exception:
            pass

    @property
    def is_initialized(self):
        return bool(self._h)

    @property
    def disp_range(self):
        return self.option.max_disparity - self.option.min_disparity

    # -- the drop-in path ----------------------------------------------------------------------
    def Match(self, img_left, img_right, disp_left):
        if not self._h:
            return False  # ADCensusStereo.cpp:71-73
        if img_left is None or img_right is None or disp_left is None:
            return False  # :74-76
        l, r = _img(img_left), _img(img_right)
        assert l.size == self.width * self.height * 3 and r.size == l.size
        assert disp_left.dtype == np.float32 and disp_left.flags["C_CONTIGUOUS"] and disp_left.size == self.width * self.height
        return lib().adc_match(self._h, l.ctypes.data, r.ctypes.data, disp_left.ctypes.data) == 0

    def match(self, img_left, img_right):
        """Convenience: returns a new float32 [H][W] map; raises on failure."""
        d = np.empty((self.height, self.width), dtype=np.float32)
        if not self.Match(img_left, img_right, d):
            raise RuntimeError("Match failed: " + last_error())
        return d

    # -- additive API --------------------------------------------------------------------------
    def match_device(self, d_left, d_right, d_disp):
        """Device pointers (ints); asynchronous; call wait().  The two image buffers are BORROWED until wait() returns: do
        not overwrite or free them before (the handle keeps no pointer to them afterwards)."""
        return lib().adc_match_device(self._h, d_left, d_right, d_disp) == 0

    def match_async(self, img_left, img_right, disp_left):
        l, r = _img(img_left), _img(img_right)
        self._keep = (l, r, disp_left)
        return lib().adc_match_async(self._h, l.ctypes.data, r.ctypes.data, disp_left.ctypes.data) == 0

    def wait(self):
        return lib().adc_wait(self._h) == 0

    def set_profiling(self, on=True):
        """False / 0: off; True / 1: stage timers + aggregation launch marks; 2: aggregation launch marks only (adcensus_c_api.h)"""
        lib().adc_set_profiling(self._h, int(on) if not isinstance(on, bool) else (1 if on else 0))

    def set_verbose(self, on=True):
        lib().adc_set_verbose(self._h, 1 if on else 0)

    def stage_ms(self):
        ms = (C.c_float * len(STAGES))()
        lib().adc_get_stage_ms(self._h, ms, len(STAGES))
        return {STAGES[i]: float(ms[i]) for i in range(len(STAGES))}

    def aggregate_pass_ms(self):
        ms, n = C.c_float(0), C.c_int(0)
        lib().adc_get_aggregate_pass_ms(self._h, C.byref(ms), C.byref(n))
        return float(ms.value), int(n.value)

    def aggregate_info(self):
        """(average ms of a regular aggregation launch, launches, algorithmic passes they covered, first pass fused?)"""
        ms, n, p, f = C.c_float(0), C.c_int(0), C.c_int(0), C.c_int(0)
        lib().adc_get_aggregate_info(self._h, C.byref(ms), C.byref(n), C.byref(p), C.byref(f))
        return float(ms.value), int(n.value), int(p.value), bool(f.value)

    def set_paper_modes(self, modes):
        """Opt-in paper features (PAPER_CENSUS5X5 | PAPER_SO_SUM | PAPER_RIGHT_ARMS); 0 = the reference's behaviour."""
        if lib().adc_set_paper_modes(self._h, int(modes)) != 0:
            raise RuntimeError("adc_set_paper_modes failed: " + last_error())

    def aggregate_kernel(self):
        return lib().adc_get_aggregate_kernel(self._h).decode()

    # -- test-only debug surface ---------------------------------------------------------------
    def _buf_spec(self, which):
        h, w, d = self.height, self.width, self.disp_range
        return {
            BUF_GRAY_LEFT: (np.uint8, (h, w)), BUF_GRAY_RIGHT: (np.uint8, (h, w)),
            BUF_CENSUS_LEFT: (np.uint64, (h, w)), BUF_CENSUS_RIGHT: (np.uint64, (h, w)),
            BUF_ARMS: (np.uint8, (h, w, 4)),
            BUF_SUPCOUNT_H: (np.uint16, (h, w)), BUF_SUPCOUNT_V: (np.uint16, (h, w)),
            BUF_VOLUME_A: (np.float32, (h, w, d)),
            BUF_DISP_LEFT: (np.float32, (h, w)), BUF_DISP_RIGHT: (np.float32, (h, w)),
            BUF_OUTLIER_LABEL: (np.uint8, (h, w)),
        }[which]

    def debug_read(self, which):
        dt, shp = self._buf_spec(which)
        out = np.empty(shp, dtype=dt)
        rc = lib().adc_debug_read(self._h, which, out.ctypes.data)
        if rc != 0:
            raise RuntimeError("adc_debug_read(%d) failed: %s" % (which, last_error()))
        return out

    def debug_write(self, which, arr):
        dt, shp = self._buf_spec(which)
        a = np.ascontiguousarray(arr, dtype=dt)
        assert a.shape == tuple(shp), (a.shape, shp)
        rc = lib().adc_debug_write(self._h, which, a.ctypes.data)
        if rc != 0:
            raise RuntimeError("adc_debug_write(%d) failed: %s" % (which, last_error()))

    def debug_set_images(self, img_left, img_right):
        l, r = _img(img_left), _img(img_right)
        if lib().adc_debug_set_images(self._h, l.ctypes.data, r.ctypes.data) != 0:
            raise RuntimeError("adc_debug_set_images failed: " + last_error())

    def debug_run(self, stage, arg=0):
        rc = lib().adc_debug_run(self._h, stage, arg)
        if rc != 0:
            raise RuntimeError("adc_debug_run(%d) failed: %s" % (stage, last_error()))

    def debug_set_budget(self, kernels):
        """Test hook: the launch budget of the NEXT Match's voting chain (adc_debug_run ADC_RUN_REGION_VOTING with arg < 0)."""
        rc = lib().adc_debug_run(self._h, RUN_REGION_VOTING, -int(kernels))
        if rc != 0:
            raise RuntimeError("adc_debug_run failed: " + last_error())

    def debug_counter(self, which):
        return int(lib().adc_debug_counter(self._h, which))

    def voting_stats(self):
        r, e = C.c_int64(0), C.c_int64(0)
        lib().adc_debug_voting_stats(self._h, C.byref(r), C.byref(e))
        return int(r.value), int(e.value)
