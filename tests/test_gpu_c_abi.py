"""The drop-in boundary from C: tests/c/abi_threads.c includes nothing but include/ka9q_hip.h, links libka9q_hip.so and
drives create_filter_input / create_filter_output / set_filter / execute_filter_input / execute_filter_output from a
producer and a consumer thread, as radio.c:186-195 and fm.c:43 do.  It checks the known answers SURVEY.md 8(c) recorded
from the unmodified reference (noise_gain, transient power, unity passband gain, y[0] of blocks 1 and 2)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_caller_with_producer_and_consumer_threads():
    exe = os.path.join(ROOT, "tests", "c", "abi_threads")
    assert os.path.exists(exe), "tests/c/abi_threads is built by __graft_entry__.build() (make -C tests/c)"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "abi_threads: ok" in r.stdout
