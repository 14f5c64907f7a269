"""Build a variant of the library: python tools/exp/build_variant.py <tag> [-D...]  ->  tools/exp/lib_<tag>.so"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from minrms_amd import build  # noqa: E402

out = os.path.join(ROOT, "tools", "exp", "lib_%s.so" % sys.argv[1])
build.build_engine(force=True, extra_flags=sys.argv[2:], lib=out, guard=False)
print(out)
