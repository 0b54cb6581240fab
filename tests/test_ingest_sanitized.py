"""The text readers (amof_amd/csrc/ingest.hip, plain host C++) under AddressSanitizer + UBSan on
the CPU: well-formed fixtures, every truncation of a small trajectory, and corrupted bytes.  The
parser works on an mmap of the file, so an off-by-one at the end of the mapping would fault here."""

import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _build(tmp_path_factory, flags):
    out = str(tmp_path_factory.mktemp("san") / "ingest_driver")
    cmd = ["g++", "-std=c++17", "-O1", "-g"] + flags + [
           "-x", "c++", os.path.join(ROOT, "amof_amd", "csrc", "ingest.hip"),
           os.path.join(ROOT, "tests", "native", "ingest_sanitize_driver.cpp"), "-lpthread", "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer build unavailable: " + r.stderr[-400:])
    return out


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    return _build(tmp_path_factory, ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"])


@pytest.fixture(scope="module")
def tsan_driver(tmp_path_factory):
    return _build(tmp_path_factory, ["-fsanitize=thread"])


def _run(driver, paths):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([driver] + paths, capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-2000:]
    assert "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-2000:]
    return r.stdout


def test_fixtures_clean_under_sanitizers(driver):
    out = _run(driver, [os.path.join(GOLD, "ZIF-4.xyz"), os.path.join(GOLD, "toy_trajectory_200.cell")])
    assert "rc=0 F=1 N=272 lattice=1" in out and "cell rc=0 rows=" in out


def test_truncations_and_corruptions(driver, tmp_path):
    rng = np.random.default_rng(5)
    lines = []
    for f in range(3):
        lines.append("5\n")
        lines.append('Lattice="10.0 0.0 0.0 0.0 11.0 0.0 0.0 0.0 12.5" Properties=species:S:1:pos:R:3 i = %d\n' % f)
        for k in range(5):
            x, y, z = rng.normal(size=3) * 10.0 ** int(rng.integers(-3, 4))
            lines.append("%-2s %.10g %.12e %r\n" % (["Zn", "N", "C", "H", "H"][k], x, y, float(z)))
    text = "".join(lines).encode()
    paths = []
    for cut in range(0, len(text) + 1):                 # every truncation, with and without a final newline
        p = tmp_path / ("cut%04d.xyz" % cut)
        p.write_bytes(text[:cut])
        paths.append(str(p))
    for k in range(200):                                # random byte corruption
        b = bytearray(text)
        for _ in range(int(rng.integers(1, 6))):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        p = tmp_path / ("bad%04d.xyz" % k)
        p.write_bytes(bytes(b))
        paths.append(str(p))
    cell = b"#   Step   Time [fs]       Ax [Angstrom]  Ay Az Bx By Bz Cx Cy Cz Volume\n" + b"".join(
        b"%8d %12.3f " % (i, 0.5 * i) + b" ".join(b"%.10f" % v for v in rng.normal(size=9)) + b" 1234.5\n" for i in range(4))
    for cut in range(0, len(cell) + 1, 3):
        p = tmp_path / ("cut%04d.cell" % cut)
        p.write_bytes(cell[:cut])
        paths.append(str(p))
    out = _run(driver, paths)
    full = [l for l in out.splitlines() if ("cut%04d.xyz" % len(text)) in l and "read(threads=1)" in l]
    assert full and "rc=0 F=3 N=5 lattice=1" in full[0]


def test_threaded_read_clean_under_tsan(tsan_driver, tmp_path):
    # the frame-parallel reader (worker threads writing disjoint frames, shared error slots)
    rng = np.random.default_rng(6)
    with open(tmp_path / "many.xyz", "w") as fh:
        for f in range(64):
            fh.write("7\nframe %d\n" % f)
            for k in range(7):
                fh.write("C %.8f %.8f %.8f\n" % tuple(rng.normal(size=3)))
    out = _run(tsan_driver, [str(tmp_path / "many.xyz"), os.path.join(GOLD, "ZIF-4.xyz")])
    assert "rc=0 F=64 N=7" in out
