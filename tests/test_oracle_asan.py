"""the CPU restatement (the checker of every parity test) under AddressSanitizer and
UndefinedBehaviorSanitizer.  `make -C oracle asan` builds oracle/_asan/libka9q_oracle_asan.so from the same source and runs
tests/test_oracle_*.py against it with the sanitizer runtimes preloaded; any report aborts that run."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_is_clean_under_asan_and_ubsan():
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None or os.environ.get("KA9Q_ORACLE_SO"):  # (no compiler, or already inside the sanitizer run)
        pytest.skip("no C compiler here")
    rt = subprocess.run([cc, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(rt):
        pytest.skip("no AddressSanitizer runtime here")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert " passed" in r.stdout


def test_oracle_threaded_use_is_clean_under_tsan():
    """the one threaded use of the oracle (oracle/pipeline.py threads=: one forward transform, the channels' filter outputs
    and demodulators in parallel) under ThreadSanitizer: `make -C oracle tsan` runs oracle/tsan_driver.c"""
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None or os.environ.get("KA9Q_ORACLE_SO"):
        pytest.skip("no C compiler here")
    rt = subprocess.run([cc, "-print-file-name=libtsan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(rt):
        pytest.skip("no ThreadSanitizer runtime here")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "tsan"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "tsan_driver:" in r.stdout and "WARNING: ThreadSanitizer" not in r.stdout + r.stderr
