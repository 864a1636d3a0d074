"""CPU tier, sanitizer builds (round-3 review item 7d): the host C++ layer -- facade + CLI with its
hand-written PNG / zlib codec -- and both oracles run under ASAN + UBSAN (-fno-sanitize-recover: any report is a non-zero exit).
The CLI is linked against a stub of the C ABI (tests/sanitize/stub_capi.c: no HIP, a synthetic disparity map), so the whole
main() path runs here: PNG decoding of every colour type the decoder accepts, Initialize / Match through the facade,
SaveDisparityMap / SaveDisparityCloud writers, and the rejection of damaged files."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")


def _make(target_dir):
    r = subprocess.run(["make", "-C", target_dir, "asan"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.fail("make asan failed in %s:\n%s" % (target_dir, r.stdout[-2000:] + r.stderr[-2000:]))


@pytest.fixture(scope="module")
def cli_asan():
    _make(os.path.join(ROOT, "adcensus_amd", "host"))
    return os.path.join(ROOT, "adcensus_amd", "build", "asan", "adcensus_cli_asan")


def _run(cmd, ok=True):
    r = subprocess.run(cmd, env=ENV, capture_output=True, text=True, timeout=300)
    report = "ERROR: AddressSanitizer" in r.stderr or "runtime error:" in r.stderr or "LeakSanitizer" in r.stderr
    assert not report, r.stderr[-3000:]
    if ok:
        assert r.returncode == 0, (r.returncode, r.stdout[-1000:], r.stderr[-2000:])
    return r


def test_oracles_under_sanitizers():
    _make(os.path.join(ROOT, "oracle"))
    out = _run([os.path.join(ROOT, "oracle", "_port", "sanitize_port")]).stdout
    assert "match == staged run: yes" in out
    ref = os.path.join(ROOT, "oracle", "_ref", "sanitize_ref")
    if os.path.isdir("/root/reference/AD-Census"):
        assert os.path.exists(ref)
        assert "match == staged run: yes" in _run([ref]).stdout


def test_cli_and_facade_under_sanitizers(cli_asan, tmp_path):
    from PIL import Image
    rng = np.random.default_rng(3)
    w, h = 83, 57  # odd sizes: every filter type / row length path of the decoder
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    rgb[:, :, 1] = (np.arange(w)[None, :] * 3 + np.arange(h)[:, None]) % 256  # smooth channel: Sub / Up / Paeth filters get chosen
    variants = {"rgb": Image.fromarray(rgb), "gray": Image.fromarray(rgb[:, :, 1]), "rgba": Image.fromarray(np.dstack([rgb, rgb[:, :, :1]])),
                "pal": Image.fromarray(rgb).quantize(64)}
    for name, img in variants.items():
        img.save(tmp_path / ("%s.png" % name), optimize=(name == "rgb"))
    # decode + re-encode of every colour type; the re-encoded file decodes (PIL) to what PIL reads from the original
    for name in variants:
        out = tmp_path / ("%s_conv.png" % name)
        _run([cli_asan, "--convert", str(tmp_path / ("%s.png" % name)), str(out)])
        want = np.array(Image.open(tmp_path / ("%s.png" % name)).convert("RGB"))
        assert np.array_equal(np.array(Image.open(out).convert("RGB")), want), name
    # binary PPM input
    with open(tmp_path / "in.ppm", "wb") as f:
        f.write(b"P6\n# comment\n%d %d\n255\n" % (w, h) + rgb.tobytes())
    _run([cli_asan, "--convert", str(tmp_path / "in.ppm"), str(tmp_path / "ppm_conv.png")])
    assert np.array_equal(np.array(Image.open(tmp_path / "ppm_conv.png").convert("RGB")), rgb)
    # the whole main(): load, Initialize / Match through the facade (stub C ABI), all four writers
    pref = tmp_path / "out"
    r = _run([cli_asan, str(tmp_path / "rgb.png"), str(tmp_path / "rgba.png"), "-3", "29", str(pref)])
    assert "Matching...Done" in r.stdout
    d = np.array(Image.open(str(pref) + "-d.png"))
    c = np.array(Image.open(str(pref) + "-c.png"))
    assert d.shape == (h, w) and c.shape == (h, w, 3)
    _run([cli_asan, "--colormap", str(pref) + "-d.png", str(tmp_path / "cm.png")])
    assert np.array_equal(np.array(Image.open(tmp_path / "cm.png").convert("RGB")), c)
    with open(str(pref) + ".pfm", "rb") as f:
        assert f.readline().strip() == b"Pf" and f.readline().split() == [str(w).encode(), str(h).encode()]
        f.readline()
        disp = np.frombuffer(f.read(), "<f4").reshape(h, w)[::-1]
    rows = open(str(pref) + "-cloud.txt").read().splitlines()
    assert len(rows) == int(np.isfinite(disp).sum()) and (disp < 0).any() and np.isinf(disp).any()
    # error paths: mismatching sizes, empty disparity range (Initialize -> false), missing file
    Image.fromarray(rgb[:-1]).save(tmp_path / "short.png")
    assert _run([cli_asan, str(tmp_path / "rgb.png"), str(tmp_path / "short.png")], ok=False).returncode != 0
    assert _run([cli_asan, str(tmp_path / "rgb.png"), str(tmp_path / "rgb.png"), "5", "5", str(pref)], ok=False).returncode != 0
    assert _run([cli_asan, str(tmp_path / "nope.png"), str(tmp_path / "rgb.png")], ok=False).returncode != 0
    # damaged files must be rejected (or decoded) without touching memory they do not own: truncations at every 37th byte, a
    # corrupted length field, a corrupted IDAT payload, an IHDR claiming a huge image
    blob = open(tmp_path / "rgb.png", "rb").read()
    bad = {"trunc_%d" % n: blob[:n] for n in range(1, len(blob), 37)}
    bad["len"] = blob[:33] + b"\x7f\xff\xff\xf0" + blob[37:]
    bad["idat"] = blob[:60] + bytes((b ^ 0x5A) for b in blob[60:120]) + blob[120:]
    bad["huge"] = blob[:16] + (60000).to_bytes(4, "big") + (60000).to_bytes(4, "big") + blob[24:]
    for name, data in bad.items():
        p = tmp_path / ("bad_%s.png" % name)
        with open(p, "wb") as f:
            f.write(data)
        _run([cli_asan, "--convert", str(p), str(tmp_path / "bad_out.png")], ok=False)
