"""Host-side sanitizer run of the HIP kernels (tests/cpu_sim): the kernel sources compiled for the CPU with
AddressSanitizer + UBSan, one thread per lane, checked against the oracle.  Catches out-of-bounds accesses and
logic errors before a kernel reaches a GPU; it is test infrastructure, never a product path."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SIM = os.path.join(HERE, "cpu_sim")


@pytest.fixture(scope="module")
def sim_main():
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    subprocess.check_call(["make", "-C", SIM, "-j8", "-s"])
    return os.path.join(SIM, "sim_main")


def _run(exe, *args):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    p = subprocess.run([exe, *args], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return p.stdout


def test_record_kernels_under_asan(sim_main):
    assert "records: ok" in _run(sim_main, "records")


def test_rans_kernels_under_asan(sim_main):
    assert "rans: ok" in _run(sim_main, "rans", "quick")


def test_gunzip_kernel_under_asan(sim_main):
    """Valid and mutated gzip members, every stream and output slot a heap block of its exact size: zlib's verdict, and not a
    byte read or written outside."""
    assert "gunzip: ok" in _run(sim_main, "gunzip")


def test_gzip_kernel_under_asan(sim_main):
    """Members written by the gzip kernel from inputs and into slots of exact size (block boundaries, an alphabet whose
    optimal code is deeper than 15 bits, incompressible and empty inputs, BGZF): zlib and the gunzip kernel return the input,
    a slot one byte short is refused, and not a byte is read or written outside."""
    assert "gzip: ok" in _run(sim_main, "gzip")


def test_bam_decoder_under_asan(sim_main):
    """Stripes of BAM records with damaged fields in a heap block of their exact size: flagged, nothing touched outside."""
    out = _run(sim_main, "bam")
    assert "bam fuzz: ok" in out and "bam encode: ok" in out

