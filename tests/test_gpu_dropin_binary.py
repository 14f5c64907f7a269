"""bin/minrms with the HIP engine behind the SearchOr interface
(oracle/_ref/minrms_hip = the reference's own objects + integration/*.cc +
libminrms_hip.so, built in the container that has the reference tree) must
write the same files as the stock binary: align<N>.msf byte for byte except the
date in the MSF header, align_chimera.plot and align_chimera.info byte for byte;
and print the same "<count> updates for orientation <o>" lines while it searches."""
import json
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "minrms_hip")


def _mask(text):
    return re.sub(r"(MSF:\s*\d+\s+Type:\s*\S+\s+).*?(\s+Check:)", r"\1<DATE>\2", text)


@pytest.mark.parametrize("name", ["ship41x13", "syn60x50_c1", "syn70x80", "syn60x50_of", "syn60x50_alt3", "syn60x50_altr3"])
def test_same_files_as_stock_minrms(name, tmp_path):
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/minrms_hip was not built (needs the reference tree)")
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "msf_%s.json" % name)))
    (tmp_path / "a.pdb").write_text(g["a_pdb"])
    (tmp_path / "b.pdb").write_text(g["b_pdb"])
    print("case", name, flush=True)
    p = subprocess.run([EXE] + g["flags"].split() + ["a.pdb", "b.pdb"], cwd=tmp_path,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=240)
    assert p.returncode == 0, p.stdout[-3000:]
    # what the search reports per orientation that improved the table (first pass)
    updates = []
    for line in p.stdout.splitlines():
        m = re.match(r"^(\d+) updates for orientation (\d+)", line)
        if m:
            updates.append([int(m.group(1)), int(m.group(2))])
    assert updates == g["updates"], "different 'updates for orientation' lines"
    got = sorted(f for f in os.listdir(tmp_path) if f.startswith("align"))
    assert got == sorted(g["files"]), "different set of output files"
    for f in got:
        assert _mask((tmp_path / f).read_text()) == g["files"][f], "%s differs" % f
