"""The channel bank's HOST code under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers; GPU
code cannot be instrumented on this pool, host code can).  tests/c/bank_san.c, a plain C caller that sees only
include/ka9q_hip.h, walks the bank through the raw ring's read offset and its compaction, refused writes and launches,
launches queued without a sync with retunes and mode switches in between, and the streaming ring -- once against the
shipped library and once against `make -C tests/c san`'s instrumented build, each with the back halves queued by the caller
and by the bank's own thread (KA9Q_HIP_ENQ_THREAD).  No sanitizer report, and one PCM checksum for all four runs."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c")


def _run(exe, thread):
    env = dict(os.environ, KA9Q_HIP_ENQ_THREAD=str(thread), ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=300)
    # (an instrumented binary that cannot set up its shadow memory on this machine is an environment matter, not a finding)
    if "_san" in exe and r.returncode != 0 and not r.stdout.strip() and re.search(
            r"ReserveShadowMemoryRange|Shadow memory range interleaves|failed to (mmap|allocate)|ASan runtime does not come first", r.stderr):
        pytest.skip("the AddressSanitizer runtime does not start here: " + r.stderr.strip().splitlines()[0][:200])
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    m = re.search(r"bank_san: ok, checksum ([0-9a-f]{16})", r.stdout)
    assert m, r.stdout[-500:]
    return m.group(1)


def test_bank_host_paths_plain_and_under_asan_ubsan():
    plain = os.path.join(HERE, "bank_san")
    if not os.path.exists(plain):
        pytest.skip("tests/c/bank_san not built (make -C tests/c)")
    sums = {("plain", t): _run(plain, t) for t in (0, 1)}
    san = os.path.join(HERE, "_san", "bank_san_asan")
    if os.path.exists(san):
        sums.update({("asan", t): _run(san, t) for t in (0, 1)})
    assert len(set(sums.values())) == 1, sums
    if not os.path.exists(san):
        pytest.skip("instrumented build absent (make -C tests/c san): plain runs only, checksums agree")
