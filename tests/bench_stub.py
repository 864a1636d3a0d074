"""Stand-in for the `adcensus_amd` package, used ONLY by the CPU-tier test of bench.py's multi-rank path
(tests/test_bench_contract.py::test_bench_two_ranks_gloo_stub, env ADC_BENCH_MATCHER_MODULE=tests.bench_stub): the same
surface bench.py drives -- lib() with the device-memory helpers, ADCensusStereo with match_device / wait / stage timers,
PairFarm -- on host memory, with a deterministic fake "disparity map" (a hash of the two images), so that the torchrun
re-exec, the rank -> device binding, the collectives, the digest cross-check and the JSON line of the N > 1 path run
end-to-end on a machine without a GPU.  Never imported by the product or by a real bench run."""
import ctypes as C
import hashlib
import os

import numpy as np

BOUND_DEVICES = []  # device index every ADCensusStereo / PairFarm of this process was created on (the test reads it via the JSON line)


class ADCensusOption(C.Structure):
    _fields_ = [("min_disparity", C.c_int32), ("max_disparity", C.c_int32)]


def _fake_disp(left_bytes, right_bytes, w, h):
    seed = int.from_bytes(hashlib.sha256(left_bytes[:4096] + right_bytes[:4096]).digest()[:4], "little")
    return np.random.default_rng(seed).random((h, w), dtype=np.float32)


class _Lib:
    def __init__(self):
        self._mem, self._next = {}, 4096

    def adc_version(self):
        return b"bench stub (no device)"

    def adc_device_malloc(self, nbytes):
        p = self._next
        self._next += (int(nbytes) + 4095) // 4096 * 4096 + 4096
        self._mem[p] = bytearray(int(nbytes))
        return p

    def adc_device_free(self, p):
        self._mem.pop(p, None)

    def adc_memcpy_h2d(self, dst, src, nbytes):
        self._mem[dst][:nbytes] = C.string_at(src, nbytes)
        return 0

    def adc_memcpy_d2h(self, dst, src, nbytes):
        C.memmove(dst, bytes(self._mem[src][:nbytes]), nbytes)
        return 0

    def adc_device_synchronize(self):
        return 0

    def adc_device_copy_ms(self, dst, src, nbytes, reps):
        return -1.0

    def adc_device_copy_kernel_ms(self, dst, src, nbytes, reps):
        return -1.0


_LIB = _Lib()


def lib():
    return _LIB


def device_count():
    return int(os.environ.get("ADC_STUB_DEVICES", "8"))


def last_error():
    return ""


def host_register(arr):
    return None


def host_unregister(arr):
    return None


class PreviousPairFailed(RuntimeError):
    pass


class ADCensusStereo:
    def __init__(self, device=-1):
        self.device = int(device)
        BOUND_DEVICES.append(self.device)
        self._pending = None

    def Initialize(self, w, h, opt):
        self.w, self.h = int(w), int(h)
        return 0 <= self.device < device_count()

    def match_device(self, dl, dr, dd):
        self._pending = (dl, dr, dd)
        return True

    def wait(self):
        dl, dr, dd = self._pending
        out = _fake_disp(bytes(_LIB._mem[dl]), bytes(_LIB._mem[dr]), self.w, self.h)
        _LIB._mem[dd][:] = out.tobytes()
        return True

    def Match(self, left, right, disp):
        disp[:] = _fake_disp(left.tobytes(), right.tobytes(), self.w, self.h)
        return True

    def set_profiling(self, on=True):
        pass

    def stage_ms(self):
        return {"cost": 0.01, "arms": 0.01, "aggregate": 0.1, "scanline": 0.1, "wta": 0.01, "refine": 0.1}

    def aggregate_info(self):
        return (0.1, 4, 8, 1)

    def aggregate_kernel(self):
        return "stub"

    def debug_counter(self, which):
        return 0

    def Release(self):
        pass


class PairFarm:
    def __init__(self, width, height, option, device=-1, pipelines=3):
        self.w, self.h, self.pipelines = int(width), int(height), int(pipelines)
        BOUND_DEVICES.append(int(device))
        self._t = 0

    def submit(self, l, r, disp):
        disp[:] = _fake_disp(np.ascontiguousarray(l).tobytes(), np.ascontiguousarray(r).tobytes(), self.w, self.h)
        self._t += 1
        return self._t

    def wait(self, ticket):
        pass

    def drain(self):
        return self._t

    def close(self):
        pass
