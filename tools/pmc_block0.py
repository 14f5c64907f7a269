#!/usr/bin/env python3
"""Counter summary of the dominant sweep kernel (layer block 0 of the default
bench workload) from the per-kernel sums tools/profile_round.sh writes.

  python tools/pmc_block0.py <dir> <tag>  >  <tag>_pmc_dp_block0.json

FETCH_SIZE / WRITE_SIZE are in KB.  FETCH_SIZE is taken at factor 1.0: the
kernel's reads are scalar loads and 4-byte-per-lane vector loads, for which the
counter was calibrated against a known byte count on this access pattern (an
earlier build whose two launches read the 21 609 x 90 kB distance tables once
plus the 150 x 64 floats per orientation the second panel takes over from the
first: 2.82e9 B against FETCH_SIZE x 1024 = 2.905e9 B); the factor 2 of wide
coalesced 16-byte-per-lane reads does not apply."""
import csv
import json
import os
import sys


def rows(path):
    if not os.path.exists(path):
        return []
    return list(csv.DictReader(open(path, newline="")))


def main():
    root, tag = sys.argv[1], sys.argv[2]
    out = {"workload": "synthetic 150x150 full sweep, 21609 orientations; bench.py --steps 1 --warmup 0"}
    kernel = None
    for r in rows(os.path.join(root, "%s_pmc_fetch_size.csv" % tag)):
        if "k_dp_panel<80, false, true, true>" in r["kernel"]:
            kernel = r["kernel"]
            out["dispatches"] = int(r["dispatches"])
            out["FETCH_SIZE_KB_total"] = float(r["sum_FETCH_SIZE"])
    for r in rows(os.path.join(root, "%s_pmc_write_size.csv" % tag)):
        if r["kernel"] == kernel:
            out["WRITE_SIZE_KB_total"] = float(r["sum_WRITE_SIZE"])
    out["kernel"] = kernel
    if kernel and "WRITE_SIZE_KB_total" in out:
        out["hbm_bytes_per_launch"] = (out["FETCH_SIZE_KB_total"] + out["WRITE_SIZE_KB_total"]) * 1024.0 / out["dispatches"]
    sq = {}
    for c in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_BUSY_CYCLES"):
        for r in rows(os.path.join(root, "%s_pmc_%s.csv" % (tag, c))):
            if r["kernel"] == kernel:
                sq[c] = float(r["sum_" + c])
    if sq:
        out["sq_counters_sum_over_dispatches"] = sq
        if sq.get("SQ_WAVE_CYCLES"):
            out["share_of_wave_lifetime_in_s_waitcnt"] = sq.get("SQ_WAIT_ANY", 0.0) / sq["SQ_WAVE_CYCLES"]
        if sq.get("SQ_INSTS_VALU") and sq.get("SQ_ACTIVE_INST_VALU"):
            # SQ_ACTIVE_INST_VALU counts quad-cycles (4 shader cycles each)
            out["shader_cycles_per_vector_instruction"] = 4.0 * sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_INSTS_VALU"]
    # the cooperative DP kernel of the refinement rounds (the kernel with the largest share of GPU time)
    coop = {}
    for r in rows(os.path.join(root, "%s_pmc_fetch_size.csv" % tag)):
        if "k_dp_coop<" in r["kernel"] and int(r["dispatches"]) > coop.get("dispatches", 0):
            coop = {"kernel": r["kernel"], "dispatches": int(r["dispatches"]), "FETCH_SIZE_KB_total": float(r["sum_FETCH_SIZE"])}
    for r in rows(os.path.join(root, "%s_pmc_write_size.csv" % tag)):
        if coop and r["kernel"] == coop["kernel"]:
            coop["WRITE_SIZE_KB_total"] = float(r["sum_WRITE_SIZE"])
    if coop and "WRITE_SIZE_KB_total" in coop:
        coop["hbm_bytes_per_launch"] = (coop["FETCH_SIZE_KB_total"] + coop["WRITE_SIZE_KB_total"]) * 1024.0 / coop["dispatches"]
        out["refinement_dp_kernel"] = coop
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
