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


def test_random_alternate_solutions():
    import stress_parity
    rng = random.Random(4048)
    for idx in range(12):
        n1, n2, method, S, k_or, found = stress_parity.alt_case(rng, idx)
        print("alt %d: %dx%d -alt-%s x%d orientations=%d found=%s" % (idx, n1, n2, method, S, k_or, found), flush=True)
