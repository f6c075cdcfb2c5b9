"""GPU: a short, fixed-seed run of every randomised parity sweep in tools/fuzz_*.py (the long runs and what they found are in
profiles/r01_fuzz_*.txt and DESIGN.md section 2).  Each sweep draws shapes and parameters at random, runs the HIP path through
the C ABI and the oracle on the same inputs and applies the bars of the fixed tests; a sweep returns non-zero if any case is
outside them without one of the diagnosed, documented causes (the reference's own fp32 running sums for n0 / IF power, a
hang-AGC branch tie, fm_snr's threshold, a noise-only channel 50 dB under its neighbours)."""
import importlib
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
TOOLS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")


@pytest.mark.parametrize("name,cases,seed", [("fuzz_filter", 40, 2), ("fuzz_bank", 30, 1), ("fuzz_linear", 40, 1),
                                             ("fuzz_mixed", 30, 1), ("fuzz_ingest", 30, 1), ("fuzz_afsk", 8, 1),
                                             ("fuzz_tones", 10, 1), ("fuzz_frontend", 45, 1), ("fuzz_rtp", 30, 1),
                                             # seed 137, case 36, session 25: a bit clock decision on its knife edge INSIDE a signal
                                             # (round 2's one open case); settled by the replay over the GPU's own tone products
                                             ("fuzz_afsk", 37, 137)])
def test_sweep(hip, monkeypatch, capsys, name, cases, seed):
    if TOOLS not in sys.path:
        sys.path.insert(0, TOOLS)
    mod = importlib.import_module(name)
    monkeypatch.setattr(sys, "argv", [name, str(cases), str(seed)])
    rc = mod.main()
    out = capsys.readouterr().out
    assert rc == 0, out[-3000:]
    assert f"{cases} cases" in out
