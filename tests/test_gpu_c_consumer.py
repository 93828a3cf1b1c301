"""The C ABI from plain C: tests/c_abi/consumer.c (gcc, HIP runtime only; no Python, no torch in that process) creates
the network, packs synthetic weights, and checks forward reproducibility, danra_sample == the hand-composed loop
(bit for bit) and the error convention."""
import os
import subprocess

import pytest

from conftest import ROOT

BINARY = os.path.join(ROOT, "tests", "c_abi", "consumer")


def test_header_is_plain_c():
    """include/danra_hip.h compiles as C99 on its own (no C++, no HIP, no torch types in the signatures)."""
    src = '#include "danra_hip.h"\nint main(void) { danra_config_t c; (void)c; return DANRA_OK; }\n'
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-I",
                        os.path.join(ROOT, "include"), "-x", "c", "-"], input=src.encode(), capture_output=True)
    assert r.returncode == 0, r.stderr.decode()


@pytest.mark.gpu
def test_c_consumer_runs_on_the_gpu():
    assert os.path.exists(BINARY), "build it with `make -C tests/c_abi` (or __graft_entry__.build())"
    r = subprocess.run([BINARY], capture_output=True, timeout=240)
    assert r.returncode == 0, (r.returncode, r.stdout.decode(), r.stderr.decode())
    assert r.stdout.decode().startswith("OK ")
