"""The hand-overs between the bank's two streams under load (tools/stress_handover.py): bursts of launches queued without a
host sync, every launch fed another signal, against the same sequence run on one stream with a sync after each launch --
bit for bit.  The events that order the streams carry no system-scope fence, and the host (not the front stream) waits for
the last reader of the spectrum slot a launch overwrites: this is the test that would notice if either were wrong, and the
second case shows that it would (a planted hazard: the back stream does not wait for the front half)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(env_extra, bursts, seed, *more):
    env = dict(os.environ, **env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_handover.py"), str(bursts), str(seed), *more], env=env,
                          capture_output=True, text=True, timeout=200)


@pytest.mark.parametrize("env", [{}, {"KA9Q_HIP_ENQ_THREAD": "1"}, {"KA9Q_HIP_SLOT_HOSTWAIT": "0", "KA9Q_HIP_EVENT_FENCE": "1"}])
def test_bursts_of_launches_match_synchronised_ones(env):
    r = _run(env, 400, 11)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 bad" in r.stdout


def test_a_planted_hazard_is_found():
    r = _run({"KA9Q_HIP_DIAG_SKIP": "4"}, 40, 12)
    assert r.returncode != 0 and "DIFFERENT" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_bursts_of_launches_on_a_mixed_bank():
    """the linear / AM demodulators on their own stream beside the FM demodulator: the fork and the join are two more events
    of the same kind"""
    r = _run({}, 300, 13, "mixed")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 bad" in r.stdout
