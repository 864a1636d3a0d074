"""Runs inside a subprocess of tests/test_gpu_faults.py with ADC_HIP_LIB = libadcensus_hip_faultinj.so (the TEST build of the C ABI:
the n-th HIP call of capi.hip's object-lifetime / Match path is not executed and reports hipErrorOutOfMemory).  Prints one JSON
object; the test asserts on it.  Contract under test (ADCensusStereo.cpp:31-40,71-76; SURVEY.md 8b "HIP failure -> false"):
Initialize / Match / adc_farm_submit report the failure, nothing leaks, and the object is usable afterwards."""
import ctypes as C
import json
import sys

import numpy as np

import adcensus_amd as A
from adcensus_amd import workloads


def free_bytes(hip):
    free, total = C.c_size_t(0), C.c_size_t(0)
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return int(free.value)


def main():
    L = A.lib()
    assert hasattr(L, "adc_test_fail_at"), "not the fault-injection build"
    L.adc_test_fail_at.argtypes = [C.c_long]
    L.adc_test_fail_at.restype = None
    L.adc_test_hip_calls.restype = C.c_long
    hip = C.CDLL("libamdhip64.so")
    W, H, D = 256, 144, 64
    left, right = workloads.structured_pair(W, H, D, seed=31)
    left2, right2 = workloads.noise_pair(W, H, seed=32)
    opt = A.ADCensusOption(max_disparity=D)
    out = {}

    # ---- reference results from an undisturbed object of the same (fault-injection) library
    L.adc_test_fail_at(0)
    st = A.ADCensusStereo(device=0)
    assert st.Initialize(W, H, opt)
    create_calls = int(L.adc_test_hip_calls())
    want = st.match(left, right)
    want2 = st.match(left2, right2)
    L.adc_test_fail_at(0)
    st.match(left, right)
    match_calls = int(L.adc_test_hip_calls())
    st.Release()
    out["create_calls"], out["match_calls"] = create_calls, match_calls

    # ---- Initialize: every single HIP call of adc_create fails once -> NULL + message, and the device memory comes back
    L.adc_device_synchronize()
    base = free_bytes(hip)
    bad_create, leaks = [], []
    for n in range(1, create_calls + 1):
        L.adc_test_fail_at(n)
        h = L.adc_create(W, H, C.byref(opt), 0)
        if h or not A.last_error():
            bad_create.append(n)
            if h:
                L.adc_destroy(h)
        L.adc_test_fail_at(0)
        L.adc_device_synchronize()
        if abs(free_bytes(hip) - base) > (2 << 20):  # (the runtime's own pools move by less)
            leaks.append((n, base - free_bytes(hip)))
    out["create_not_failed"], out["create_leaks"] = bad_create, leaks
    # one step past the end: nothing fails
    L.adc_test_fail_at(create_calls + 1)
    h = L.adc_create(W, H, C.byref(opt), 0)
    out["create_past_end_ok"] = bool(h)
    L.adc_test_fail_at(0)
    if h:
        L.adc_destroy(h)

    # ---- Match: fail the n-th HIP call of a Match, then run two clean Matches on the SAME handle
    st = A.ADCensusStereo(device=0)
    assert st.Initialize(W, H, opt)
    st.match(left2, right2)  # (history: the handle has learnt arm maxima / a chain budget from another image)
    step = max(1, match_calls // 40)
    picks = sorted(set(list(range(1, min(match_calls, 12) + 1)) + list(range(1, match_calls + 1, step)) + [match_calls - 1, match_calls]))
    not_failed, wrong_after = [], []
    d = np.empty((H, W), np.float32)
    for n in picks:
        L.adc_test_fail_at(n)
        ok = st.Match(left, right, d)
        L.adc_test_fail_at(0)
        if ok or not A.last_error():
            not_failed.append(n)
        got = st.match(left, right)
        got2 = st.match(left2, right2)
        if not (np.array_equal(got.view(np.uint32), want.view(np.uint32)) and np.array_equal(got2.view(np.uint32), want2.view(np.uint32))):
            wrong_after.append(n)
    out["match_picks"], out["match_not_failed"], out["match_wrong_after"] = len(picks), not_failed, wrong_after

    # ---- asynchronous entry points: the failure surfaces in adc_match_async or in adc_wait
    async_bad = []
    for n in picks[:: max(1, len(picks) // 12)]:
        L.adc_test_fail_at(n)
        ok = st.match_async(left, right, d) and st.wait()
        L.adc_test_fail_at(0)
        if ok:
            async_bad.append(n)
        if not np.array_equal(st.match(left, right).view(np.uint32), want.view(np.uint32)):
            async_bad.append(-n)
    out["async_bad"] = async_bad
    st.Release()

    # ---- farm: a failing submit returns nonzero, the farm goes on
    farm = A.PairFarm(W, H, opt, device=0, pipelines=2)
    outs = [np.empty((H, W), np.float32) for _ in range(6)]
    farm_rc = []
    for i in range(6):
        if i == 2:
            L.adc_test_fail_at(3)
        t = C.c_uint64(0)
        rc = L.adc_farm_submit(farm._f, left.ctypes.data, right.ctypes.data, outs[i].ctypes.data, C.byref(t))
        L.adc_test_fail_at(0)
        farm_rc.append(int(rc))
    delivered = farm.drain()
    out["farm_rc"], out["farm_delivered"] = farm_rc, int(delivered)
    out["farm_outputs_ok"] = [bool(np.array_equal(outs[i].view(np.uint32), want.view(np.uint32))) for i in range(6)]
    farm.close()
    L.adc_device_synchronize()
    out["final_leak_bytes"] = base - free_bytes(hip)
    print("FAULT_PROBE " + json.dumps(out))


if __name__ == "__main__":
    sys.exit(main())
