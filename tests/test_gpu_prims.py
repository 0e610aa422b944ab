"""The wave / workgroup primitives of csrc/afq_prims.h (register bitonic sorts, block scans) against std::sort and a
serial scan: tests/prims_check.hip is compiled for gfx950 (by __graft_entry__.build(), or here when the binary is
missing) and run on the device.  Every kernel that sorts - cr-like buckets, parsimony reads, EM labels, ATAC
fragments - goes through these, so a wrong comparator network shows up here in isolation."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "prims_check.hip")
BIN = os.path.join(HERE, "_build", "prims_check")


def build_prims_check():
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    hdr = os.path.join(HERE, "..", "alevin-fry_amd", "csrc", "afq_prims.h")
    if os.path.exists(BIN) and os.path.getmtime(BIN) >= max(os.path.getmtime(SRC), os.path.getmtime(hdr)):
        return
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    r = subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-o", BIN + ".new", SRC], capture_output=True, text=True)
    if r.returncode == 0:
        os.replace(BIN + ".new", BIN)
    elif not os.path.exists(BIN):   # (a binary built by build() that only looks older after a copy of the tree is still the one to run)
        raise RuntimeError("cannot build tests/prims_check.hip:\n" + r.stderr)


@pytest.mark.gpu
def test_sort_and_scan_primitives_match_the_host():
    build_prims_check()
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().startswith("ok "), r.stdout
