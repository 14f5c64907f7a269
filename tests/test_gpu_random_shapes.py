"""Random sizes and settings, engine against oracle (tools/stress_parity.py runs
hundreds of these; the suite keeps a short, fixed-seed sample)."""
import os
import random
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_random_shapes_and_settings():
    import stress_parity
    rng = random.Random(2024)
    for idx in range(16):
        n1, n2, kw, k_or, found = stress_parity.one_case(rng, idx)
        print("case %d: %dx%d %s orientations=%d found=%s" % (idx, n1, n2, kw, k_or, found), flush=True)
