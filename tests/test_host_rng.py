"""numpy's legacy Gaussian stream in blocks (csrc/host_constrainer.cpp: gauss_fill, BandLook) against the deviate-by-deviate
forms, in a small native program: values, numpy's cached second deviate and the position of the stream, bit for bit; with the
blocks made by the caller and by the two helper threads (MDNS_BAND_THREADS=1)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    rng_o = os.path.join(ROOT, "massivedatans_amd", "csrc", "host_rng.o")
    if not os.path.exists(rng_o):
        pytest.skip("csrc/host_rng.o is not built")
    exe = str(tmp_path_factory.mktemp("native") / "host_rng_check")
    cmd = ["g++", "-O2", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-std=c++17", "-fno-exceptions", "-w",
           os.path.join(ROOT, "tests", "native", "host_rng_check.cpp"), rng_o, "-o", exe, "-lm", "-lpthread"]
    subprocess.run(cmd, check=True, timeout=300)
    return exe


@pytest.mark.parametrize("threads", ["0", "1"])
def test_blocked_gaussian_stream_is_the_deviate_by_deviate_one(checker, threads):
    out = subprocess.run([checker], env=dict(os.environ, MDNS_BAND_THREADS=threads), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "gauss_fill mismatches: 0" in out.stdout and "advance mismatches: 0" in out.stdout
