"""ctypes loader for the CPU oracles -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (adcensus_amd/) must never import it.

Two interchangeable oracles export the ABI of oracle/oracle_abi.h:
  kind "reference": oracle/_ref/libadcensus_ref.so  (reference sources compiled in place)
  kind "port":      oracle/_port/libadcensus_port.so (oracle/adcensus_port.c restatement)
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(_HERE, "_ref", "libadcensus_ref.so")
PORT_SO = os.path.join(_HERE, "_port", "libadcensus_port.so")


class Option(C.Structure):
    """Plain-C mirror of ADCensusOption (adcensus_types.h:45-75) == adc_option."""
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
        # defaults: adcensus_types.h:67-74
        self.min_disparity, self.max_disparity = 0, 64
        self.lambda_ad, self.lambda_census = 10, 30
        self.cross_L1, self.cross_L2, self.cross_t1, self.cross_t2 = 34, 17, 20, 6
        self.so_p1, self.so_p2, self.so_tso = 1.0, 3.0, 15
        self.irv_ts, self.irv_th, self.lrcheck_thres = 20, 0.4, 1.0
        self.do_lr_check, self.do_filling, self.do_discontinuity_adjustment = 1, 1, 0
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError(k)
            setattr(self, k, v)


# (field name, dtype, per-pixel shape) ; 'D' is replaced by the disparity range
_DUMP_FIELDS = [
    ("gray_left", np.uint8, ()), ("gray_right", np.uint8, ()),
    ("census_left", np.uint64, ()), ("census_right", np.uint64, ()),
    ("cost_init", np.float32, ("D",)),
    ("arms", np.uint8, (4,)),
    ("sup_count_h", np.uint16, ()), ("sup_count_v", np.uint16, ()),
    ("cost_aggr", np.float32, ("D",)), ("cost_so", np.float32, ("D",)),
    ("disp_left_wta", np.float32, ()), ("disp_right_wta", np.float32, ()),
    ("outlier_label", np.uint8, ()),
    ("disp_after_lr", np.float32, ()), ("disp_after_irv", np.float32, ()),
    ("disp_after_interp", np.float32, ()), ("disp_after_dda", np.float32, ()),
    ("disp_final", np.float32, ()),
]
DUMP_NAMES = [f[0] for f in _DUMP_FIELDS]


class _Dump(C.Structure):
    _fields_ = [(name, C.c_void_p) for name, _, _ in _DUMP_FIELDS]


class Oracle:
    def __init__(self, path):
        self.path = path
        self.lib = C.CDLL(path)
        self.lib.adc_oracle_kind.restype = C.c_char_p
        self.lib.adc_oracle_run.restype = C.c_int
        self.lib.adc_oracle_run.argtypes = [C.c_int32, C.c_int32, C.POINTER(Option), C.c_void_p, C.c_void_p,
                                            C.POINTER(_Dump)]
        self.lib.adc_oracle_match.restype = C.c_int
        self.lib.adc_oracle_match.argtypes = [C.c_int32, C.c_int32, C.POINTER(Option), C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.POINTER(C.c_double)]
        self.lib.adc_oracle_median3_inplace.restype = None
        self.lib.adc_oracle_median3_inplace.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        self.kind = self.lib.adc_oracle_kind().decode()
        try:
            self.lib.adc_oracle_build_info.restype = C.c_char_p
            self.build_info = self.lib.adc_oracle_build_info().decode()
        except AttributeError:  # a checker built before the symbol existed
            self.build_info = "unknown"

    @staticmethod
    def _check_images(left, right):
        left = np.ascontiguousarray(left, dtype=np.uint8)
        right = np.ascontiguousarray(right, dtype=np.uint8)
        assert left.ndim == 3 and left.shape[2] == 3 and left.shape == right.shape
        return left, right

    def run(self, left, right, opt=None, stages=None, paper_modes=0):
        """Runs the whole pipeline stage by stage; returns {stage name: ndarray}.
        `stages`: iterable of names from DUMP_NAMES (default: all).  paper_modes != 0: the opt-in paper features of the
        product's adc_set_paper_modes (port oracle only -- the reference does not implement them)."""
        opt = opt or Option()
        left, right = self._check_images(left, right)
        h, w = left.shape[:2]
        d = opt.max_disparity - opt.min_disparity
        want = set(DUMP_NAMES if stages is None else stages)
        unknown = want - set(DUMP_NAMES)
        assert not unknown, unknown
        out, dump = {}, _Dump()
        if w > 0 and h > 0 and d > 0:
            for name, dt, shp in _DUMP_FIELDS:
                if name in want:
                    shape = (h, w) + tuple(d if s == "D" else s for s in shp)
                    out[name] = np.zeros(shape, dtype=dt)
                    setattr(dump, name, out[name].ctypes.data)
        if paper_modes:
            if self.kind != "port":
                raise RuntimeError("paper modes exist in the port oracle only (the reference has no such code)")
            self.lib.adc_oracle_run_paper.restype = C.c_int
            self.lib.adc_oracle_run_paper.argtypes = [C.c_int32, C.c_int32, C.POINTER(Option), C.c_uint32, C.c_void_p, C.c_void_p,
                                                      C.POINTER(_Dump)]
            rc = self.lib.adc_oracle_run_paper(w, h, C.byref(opt), int(paper_modes), left.ctypes.data, right.ctypes.data, C.byref(dump))
        else:
            rc = self.lib.adc_oracle_run(w, h, C.byref(opt), left.ctypes.data, right.ctypes.data, C.byref(dump))
        if rc != 0:
            raise RuntimeError("oracle Initialize failed (rc=%d)" % rc)
        return out

    def match(self, left, right, opt=None):
        """Public-API Initialize+Match; returns (disparity, seconds spent in Match)."""
        opt = opt or Option()
        left, right = self._check_images(left, right)
        h, w = left.shape[:2]
        disp = np.zeros((h, w), dtype=np.float32)
        secs = C.c_double(0.0)
        rc = self.lib.adc_oracle_match(w, h, C.byref(opt), left.ctypes.data, right.ctypes.data,
                                       disp.ctypes.data, C.byref(secs))
        if rc != 0:
            raise RuntimeError("oracle match failed (rc=%d)" % rc)
        return disp, secs.value

    def median3_inplace(self, disp):
        disp = np.array(disp, dtype=np.float32, order="C", copy=True)
        self.lib.adc_oracle_median3_inplace(disp.ctypes.data, disp.shape[1], disp.shape[0])
        return disp


def have_ref():
    return os.path.exists(REF_SO)


def have_port():
    return os.path.exists(PORT_SO)


def load(kind="auto"):
    """kind: 'reference', 'port' or 'auto' (reference if its .so exists, else port)."""
    if kind == "auto":
        kind = "reference" if have_ref() else "port"
    path = REF_SO if kind == "reference" else PORT_SO
    if not os.path.exists(path):
        raise FileNotFoundError("%s oracle not built: %s (run `make -C oracle`)" % (kind, path))
    return Oracle(path)
