#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the real reference -- TEST INFRASTRUCTURE.

Runs oracle/_ref/ref_dump_{strict,fast} (the reference compiled in place from
/root/reference by oracle/Makefile) on the PDB fixtures the reference ships and
on seeded synthetic chains, and stores inputs + expected outputs as small numpy
archives.  Only this container can run it (the reference does not travel); the
archives it writes are committed and are all the GPU box sees.

Usage: python3 oracle/make_golden.py [case ...]     (default: all cases)
"""
import hashlib
import os
import re
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from minrms_amd.synth import synth_pdb_text, synth_backbone_pdb_text  # noqa: E402

REF = os.environ.get("MINRMS_REF", "/root/reference")
GOLD = os.path.join(ROOT, "tests", "golden")

SHIPPED_A = os.path.join(REF, "bin/fssp2msf/test_cases/insert_28to30_2_w_chainZ.pdb")
SHIPPED_B = os.path.join(REF, "bin/pdb_select/test_cases/d1.pdb")

# name -> (structure A, structure B, minrms flags, trace level, keep og?)
#   structure = ("file", path) | ("synth", n, seed) | ("synth3", n, seed)  (N, CA, C records)
#   an optional sixth element is the text of atoms_used.txt (minrms_parser.cc:81-130)
BB3 = " N\n CA\n C\n"
CASES = {
    # the only PDB fixtures the reference ships (41 x 13 residues)
    "ship41x13_full":   (("file", SHIPPED_A), ("file", SHIPPED_B), "-fm 4 -HS -ir -r", 2, True),
    "ship41x13_noir":   (("file", SHIPPED_A), ("file", SHIPPED_B), "-fm 4 -HS -IR -r", 2, True),
    "ship41x13_minN5":  (("file", SHIPPED_A), ("file", SHIPPED_B), "-fm 4 -HS -minN 5 -max-rmsd 3.0 -ir -r", 2, True),
    "syn60x50_full":    (("synth", 60, 1234), ("synth", 50, 5678), "-fm 4 -HS -ir -r", 1, True),
    "syn60x50_c1":      (("synth", 60, 1234), ("synth", 50, 5678), "-fm 4 -HS -minN 0.33 -max-rmsd 3.5 -ir -r", 1, True),
    "syn60x50_maxdist": (("synth", 60, 1234), ("synth", 50, 5678), "-fm 4 -HS -max-dist 6.0 -ir -r", 1, True),
    "syn50x70_maxN":    (("synth", 50, 4321), ("synth", 70, 8765), "-fm 4 -HS -maxN 30 -ir -r", 1, True),
    "syn90x100_full":   (("synth", 90, 11), ("synth", 100, 22), "-fm 4 -HS -ir -r", 1, False),
    "syn150x150_c1":    (("synth", 150, 1234), ("synth", 150, 5678), "-fm 4 -HS -minN 0.33 -max-rmsd 3.5 -ir -r", 1, False),
    "syn150x150_full":  (("synth", 150, 1234), ("synth", 150, 5678), "-fm 4 -HS -ir -r", 1, False),
    # Needleman-Wunsch orientation filter (-of dist,min_matches)
    "syn90x100_noir":   (("synth", 90, 11), ("synth", 100, 22), "-fm 4 -HS -IR -r", 1, False),
    "syn60x50_of_noir": (("synth", 60, 1234), ("synth", 50, 5678), "-fm 4 -HS -of 8.0,0.33 -IR -r", 1, True),
    "ship41x13_of":     (("file", SHIPPED_A), ("file", SHIPPED_B), "-fm 4 -HS -of 5.0,6 -ir -r", 2, True),
    "syn60x50_of":      (("synth", 60, 1234), ("synth", 50, 5678), "-fm 4 -HS -of 8.0,0.33 -ir -r", 1, True),
    "syn60x50_of_md":   (("synth", 60, 1234), ("synth", 50, 5678), "-fm 4 -HS -of 7.0,12 -max-dist 6.5 -minN 8 -ir -r", 1, True),
    "syn90x100_of":     (("synth", 90, 11), ("synth", 100, 22), "-fm 4 -HS -of 8.0,0.33 -ir -r", 1, False),
    # three atoms per residue
    "bb3_40x45_full":   (("synth3", 40, 31), ("synth3", 45, 32), "-fm 4 -HS -ir -r", 2, True, BB3),
    "bb3_40x45_of_md":  (("synth3", 40, 31), ("synth3", 45, 32), "-fm 4 -HS -of 6.1,8 -max-dist 3.9 -ir -r", 1, True, BB3),
    # helix/sheet interval masks (-hs: HELIX x HELIX and SHEET x SHEET pairs), explicit intervals
    "ss60x50_hs":       (("synth_ss", 60, 1234), ("synth_ss", 50, 5678), "-fm 4 -hs -ir -r", 1, True),
    "ss60x50_hs_minN":  (("synth_ss", 60, 1234), ("synth_ss", 50, 5678), "-fm 5 -hs -minN 0.5 -max-rmsd 4.0 -ir -r", 1, True),
    "syn60x50_fmint":   (("synth", 60, 1234), ("synth", 50, 5678), "-fm 4 -HS -fm-interval 5-40,10-45 -fm-interval 30-60,1-25 -ir -r", 1, True),
    "syn60x50_fmcut":   (("synth", 60, 1234), ("synth", 50, 5678), "-fm 5,1.2 -HS -ir -r", 1, True),
    # alternate solutions by orientation (-alt-o count,min_distance[,rmsd_ratio])
    "ship41x13_alt3":   (("file", SHIPPED_A), ("file", SHIPPED_B), "-fm 4 -HS -alt-o 3,2.0 -ir -r", 1, True),
    "syn60x50_alt3":    (("synth", 60, 1234), ("synth", 50, 5678), "-fm 4 -HS -alt-o 3,3.0,1.6 -ir -r", 1, True),
    "syn60x50_alt4all": (("synth", 60, 1234), ("synth", 50, 5678), "-fm 4 -HS -alt-o 4,2.5,2.0 -ir -r-all", 1, True),
    "syn60x50_alt2nr":  (("synth", 60, 1234), ("synth", 50, 5678), "-fm 4 -HS -alt-o 2,4.0 -minN 10 -max-rmsd 4.5 -ir -R", 1, True),
    # ... compared by the orientations the alignments were found at (-alt-o-orig, implies -IR)
    "ship41x13_alto3":  (("file", SHIPPED_A), ("file", SHIPPED_B), "-fm 4 -HS -alt-o-orig 3,2.0 -r", 1, True),
    "syn60x50_alto3all": (("synth", 60, 1234), ("synth", 50, 5678), "-fm 4 -HS -alt-o-orig 3,3.0,1.6 -r-all", 1, True),
    # alternate solutions by matched pairs (-alt-r count,max_pairs_in_common[,rmsd_ratio])
    "ship41x13_altr3":  (("file", SHIPPED_A), ("file", SHIPPED_B), "-fm 4 -HS -alt-r 3,4 -ir -r", 1, True),
    "syn60x50_altr3":   (("synth", 60, 1234), ("synth", 50, 5678), "-fm 4 -HS -alt-r 3,10 -ir -r", 1, True),
    "syn60x50_altr3tol": (("synth", 60, 1234), ("synth", 50, 5678), "-fm 4 -HS -alt-r 3,6,1.5 -ir -r-all", 1, True),
    "syn60x50_altr2IR": (("synth", 60, 1234), ("synth", 50, 5678), "-fm 4 -HS -alt-r 2,8,2.0 -max-rmsd 5.0 -IR -r", 1, True),
    # the larger benchmark sizes: only the first `limit` orientations of the
    # shuffled list are swept (the full sweeps take the reference 0.5 h / 5 h)
    "syn250x250_first400": (("synth", 250, 1234), ("synth", 250, 5678), "-fm 4 -HS -ir -r", 1, False, None, 400),
    "syn400x400_first150": (("synth", 400, 1234), ("synth", 400, 5678), "-fm 4 -HS -ir -r", 1, False, None, 150),
    "syn300x220_first200": (("synth", 300, 77), ("synth", 220, 88), "-fm 4 -HS -minN 20 -max-rmsd 6.0 -ir -r", 1, False, None, 200),
}


# Full sweeps of the large benchmark sizes.  name -> (A, B, flags, n_ranges, refine?)
#   refine = True : ONE reference process, Search() over everything, then Refine()
#                   (strict build only; ~0.5 h at 250x250)
#   refine = False: Search() only, the o_id list cut into n_ranges contiguous
#                   ranges, one reference process each (Search(og, start, stop),
#                   search_or.h:92-103); the per-N tables are merged with the
#                   rule of FirstPass (search_or_dyn3d.cc:1825-1835): a later
#                   orientation replaces an entry only if strictly smaller
FULL_CASES = {
    "syn250x250_full": (("synth", 250, 1234), ("synth", 250, 5678), "-fm 4 -HS -ir -r", 1, True),
    "syn400x400_fullsearch": (("synth", 400, 1234), ("synth", 400, 5678), "-fm 4 -HS -ir -r", 6, False),
}


def run_full_case(name, n_ranges=None):
    specA, specB, flags, default_ranges, refine = FULL_CASES[name]
    n_ranges = n_ranges or default_ranges
    exe = os.path.join(HERE, "_ref", "ref_dump_strict")
    with tempfile.TemporaryDirectory() as tmp:
        _stage(specA, os.path.join(tmp, "a.pdb"))
        _stage(specB, os.path.join(tmp, "b.pdb"))
        # how many orientations: one bounded run that searches nothing
        env = dict(os.environ, REF_DUMP_OUT=os.path.join(tmp, "count.txt"), REF_DUMP_TRACE="0",
                   REF_DUMP_BENCH="1")
        subprocess.run([exe] + flags.split() + ["a.pdb", "b.pdb"], cwd=tmp, env=env, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT)
        head = parse_dump(os.path.join(tmp, "count.txt"))
        n_og = head["og"].shape[0]
        procs = []
        for r in range(n_ranges):
            lo, hi = n_og * r // n_ranges, n_og * (r + 1) // n_ranges
            env = dict(os.environ, REF_DUMP_OUT=os.path.join(tmp, "dump%d.txt" % r), REF_DUMP_TRACE="0",
                       REF_DUMP_START=str(lo), REF_DUMP_LIMIT=str(hi))
            if not refine:
                env["REF_DUMP_SEARCH_ONLY"] = "1"
            log = open(os.path.join(tmp, "log%d.txt" % r), "w")
            procs.append((lo, hi, subprocess.Popen([exe] + flags.split() + ["a.pdb", "b.pdb"], cwd=tmp,
                                                   env=env, stdout=log, stderr=subprocess.STDOUT)))
        out = None
        for r, (lo, hi, p) in enumerate(procs):
            if p.wait() != 0:
                raise RuntimeError("%s range %d failed:\n%s" % (name, r, open(os.path.join(tmp, "log%d.txt" % r)).read()[-2000:]))
            d = parse_dump(os.path.join(tmp, "dump%d.txt" % r))
            if out is None:
                out = d
                out["search_range"] = np.full(d["search_ssd"].shape[0], 0, np.int32)
                out["range_lo"] = [lo]
                out["range_hi"] = [hi]
            else:
                better = d["search_ssd"] < out["search_ssd"]          # strict: ties stay with the earlier range
                for key in ("search_ssd", "search_xform", "search_pairs"):
                    out[key][better] = d[key][better]
                out["search_range"][better] = r
                out["range_lo"].append(lo)
                out["range_hi"].append(hi)
            if refine:
                text = open(os.path.join(tmp, "log%d.txt" % r)).read()
                m = re.search(r"During refinement, (\d+) new orientations were tested,\s+and (\d+) instances", text)
                out["refine_iters"], out["refine_new"] = int(m.group(1)), int(m.group(2))
    out["flags"] = flags
    og = out.pop("og")
    out["og_sha256"] = hashlib.sha256(np.ascontiguousarray(og).tobytes()).hexdigest()
    out["og_count"] = og.shape[0]
    out["searched"] = og.shape[0]
    out["range_lo"] = np.array(out["range_lo"], np.int64)
    out["range_hi"] = np.array(out["range_hi"], np.int64)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print("%-22s og=%d ranges=%d refine_iters=%s -> %d kB" % (
        name, out["og_count"], n_ranges, out.get("refine_iters"),
        os.path.getsize(os.path.join(GOLD, name + ".npz")) // 1024))


def run_refine_from_search(name, check_only=False):
    """Refine() on top of the merged Search() tables of a full-sweep golden.

    The sweep of FULL_CASES[name] was split over several reference processes, so
    no single process holds the tables Refine() starts from.  They are handed
    back to ONE reference process (ref_dump's REF_DUMP_LOAD_SEARCH), which runs
    the reference's own Refine() on them; the result joins the archive as
    refine_* / refine_iters / refine_new.  With check_only the archive must
    already hold a Refine() result from a one-process run (syn250x250_full): the
    reloaded run has to reproduce it bit for bit, which validates the method."""
    specA, specB, flags = FULL_CASES[name][:3]
    path = os.path.join(GOLD, name + ".npz")
    g = dict(np.load(path))
    exe = os.path.join(HERE, "_ref", "ref_dump_strict")
    n_min, n_max = int(g["n_min"]), int(g["n_max"])
    with tempfile.TemporaryDirectory() as tmp:
        _stage(specA, os.path.join(tmp, "a.pdb"))
        _stage(specB, os.path.join(tmp, "b.pdb"))
        with open(os.path.join(tmp, "tables.txt"), "w") as f:
            for k in range(n_max - n_min + 1):
                n = n_min + k
                have = g["search_ssd"][k] < np.float32(1e26)
                words = ["%d" % n, "%08x" % g["search_ssd"][k:k + 1].view(np.uint32)[0]]
                words += ["%08x" % w for w in g["search_xform"][k].view(np.uint32)]
                words.append("%d" % (n if have else 0))
                if have:
                    words += ["%d %d" % (int(p[0]), int(p[1])) for p in g["search_pairs"][k, :n]]
                f.write(" ".join(words) + "\n")
        env = dict(os.environ, REF_DUMP_OUT=os.path.join(tmp, "dump.txt"), REF_DUMP_TRACE="0",
                   REF_DUMP_LOAD_SEARCH=os.path.join(tmp, "tables.txt"))
        p = subprocess.run([exe] + flags.split() + ["a.pdb", "b.pdb"], cwd=tmp, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if p.returncode != 0:
            raise RuntimeError("%s failed on %s:\n%s" % (exe, name, p.stdout[-2000:]))
        d = parse_dump(os.path.join(tmp, "dump.txt"))
        m = re.search(r"During refinement, (\d+) new orientations were tested,\s+and (\d+) instances", p.stdout)
        d["refine_iters"], d["refine_new"] = int(m.group(1)), int(m.group(2))
    for key in ("search_ssd", "search_xform", "search_pairs"):          # the tables went in unchanged
        assert np.array_equal(d[key].view(np.uint8), g[key].view(np.uint8)), key
    keys = ("refine_ssd", "refine_xform", "refine_pairs", "refine_iters", "refine_new")
    if check_only:
        for key in keys:
            assert np.array_equal(np.asarray(d[key]), np.asarray(g[key])), "%s: %s differs from the one-process run" % (name, key)
        print("%-22s Refine() on reloaded Search() tables reproduces the one-process run (%d iterations)" % (
            name, d["refine_iters"]))
        return
    for key in keys:
        g[key] = d[key]
    np.savez_compressed(path, **g)
    print("%-22s refine_iters=%d refine_new=%d -> %d kB" % (name, d["refine_iters"], d["refine_new"],
                                                          os.path.getsize(path) // 1024))


def _hexf(tokens):
    return np.array([int(t, 16) for t in tokens], dtype=np.uint32).view(np.float32)


def parse_dump(path, n_max_hint=None):
    d = {}
    with open(path) as f:
        lines = f.read().split("\n")
    i = 0
    trace_rows = []
    while i < len(lines):
        t = lines[i].split()
        i += 1
        if not t:
            continue
        if t[0] == "DIMS":
            d["N1"], d["N2"], d["nbb"] = int(t[1]), int(t[2]), int(t[3])
        elif t[0] == "SETTINGS":
            kv = dict(zip(t[1::2], t[2::2]))
            for k in ("n_min", "n_max", "rws", "refine_method", "refine_max_iters",
                      "num_solutions", "of", "of_min", "fm", "alt_method", "alt_pairs"):
                if k in kv:
                    d[k] = int(kv[k])
            for k in ("max_rmsd", "max_dist", "of_dist", "fm_cut", "alt_min_3d", "alt_tol"):
                if k in kv:
                    d[k] = _hexf([kv[k]])[0]
        elif t[0] in ("XYZA", "XYZB"):
            d[t[0].lower()] = _hexf(t[1:]).reshape(-1, 3)
        elif t[0] == "IV":
            cnt = int(t[1])
            d["intervals"] = np.array([lines[i + k].split() for k in range(cnt)], np.int32).reshape(cnt, 4)
            i += cnt
        elif t[0] == "OG":
            cnt = int(t[1])
            og = np.empty((cnt, 12), np.float32)
            for k in range(cnt):
                og[k] = _hexf(lines[i + k].split())
            i += cnt
            d["og"] = og
        elif t[0] == "PAXIS":
            v = _hexf(t[1:])
            d["paxis_eigenvalues"], d["paxis_matrix"] = v[:3], v[3:]
        elif t[0] in ("SEARCH", "REFINE", "SEARCH_ALT", "REFINE_ALT"):
            rn = int(t[2]) if t[0].endswith("_ALT") else int(t[1])
            n_max = d["n_max"]
            ssd = np.empty(rn, np.float32)
            xf = np.empty((rn, 12), np.float32)
            pairs = np.full((rn, n_max, 2), -1, np.int16)
            for k in range(rn):
                r = lines[i + k].split()
                ssd[k] = _hexf([r[1]])[0]
                xf[k] = _hexf(r[2:14])
                nm = int(r[14])
                rest = r[15:]
                if rest:
                    pairs[k, :nm] = np.array(rest, dtype=np.int16).reshape(nm, 2)
            i += rn
            tag = t[0].lower()
            if tag.endswith("_alt"):
                tag = "%s%s" % (tag, t[1])          # search_alt1, refine_alt2, ...
            d[tag + "_ssd"], d[tag + "_xform"], d[tag + "_pairs"] = ssd, xf, pairs
        elif t[0] == "TRACE":
            pass
        elif t[0] == "T":
            trace_rows.append(t[1:])
        elif t[0] == "F":
            d.setdefault("filtered_out", []).append(int(t[1]))
        elif t[0] == "SEARCHED":
            d["searched"] = int(t[1])
        elif t[0] == "CLIP":
            d["clip_after_search"] = _hexf([t[2]])[0]
        elif t[0] == "WINNERS":
            rn = int(t[1])
            d["passed"] = int(t[3])
            w = np.empty(rn, np.int64)
            for k in range(rn):
                r = lines[i + k].split()
                w[k] = int(r[1])
            i += rn
            d["winner_oid"] = w
    if trace_rows:
        # per (o, n): ssd_orig bits, ssd bits, pairs
        n_max = d["n_max"]
        m = len(trace_rows)
        to = np.empty(m, np.int32)
        tn = np.empty(m, np.int16)
        tso = np.empty(m, np.float32)
        ts = np.empty(m, np.float32)
        tp = np.full((m, n_max, 2), -1, np.int16)
        for k, r in enumerate(trace_rows):
            to[k], tn[k] = int(r[0]), int(r[1])
            tso[k], ts[k] = _hexf(r[2:4])
            tp[k, :tn[k]] = np.array(r[4:], dtype=np.int16).reshape(tn[k], 2)
        d.update(trace_o=to, trace_n=tn, trace_ssd_orig=tso, trace_ssd=ts, trace_pairs=tp)
    if "filtered_out" in d:
        d["filtered_out"] = np.array(d["filtered_out"], np.int64)
    return d


def _stage(spec, dst):
    if spec[0] == "file":
        shutil.copy(spec[1], dst)
    elif spec[0] == "synth_ss":
        with open(dst, "w") as f:
            f.write(synth_pdb_text(spec[1], spec[2], with_secondary_structure=True))
    elif spec[0] == "synth3":
        with open(dst, "w") as f:
            f.write(synth_backbone_pdb_text(spec[1], spec[2]))
    else:
        with open(dst, "w") as f:
            f.write(synth_pdb_text(spec[1], spec[2]))


def run_case(name):
    specA, specB, flags, trace, keep_og = CASES[name][:5]
    atoms_used = CASES[name][5] if len(CASES[name]) > 5 else None
    limit = CASES[name][6] if len(CASES[name]) > 6 else None
    out = {}
    for variant in ("strict", "fast"):
        exe = os.path.join(HERE, "_ref", "ref_dump_" + variant)
        with tempfile.TemporaryDirectory() as tmp:
            _stage(specA, os.path.join(tmp, "a.pdb"))
            _stage(specB, os.path.join(tmp, "b.pdb"))
            if atoms_used:
                with open(os.path.join(tmp, "atoms_used.txt"), "w") as f:
                    f.write(atoms_used)
            env = dict(os.environ, REF_DUMP_OUT=os.path.join(tmp, "dump.txt"),
                       REF_DUMP_TRACE=str(trace if variant == "strict" else 0))
            if limit:
                env["REF_DUMP_LIMIT"] = str(limit)
            p = subprocess.run([exe] + flags.split() + ["a.pdb", "b.pdb"], cwd=tmp, env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if p.returncode != 0:
                raise RuntimeError("%s failed on %s:\n%s" % (exe, name, p.stdout[-2000:]))
            d = parse_dump(os.path.join(tmp, "dump.txt"))
            m = re.search(r"During refinement, (\d+) new orientations were tested,\s+and (\d+) instances",
                          p.stdout)
            if m:
                d["refine_iters"], d["refine_new"] = int(m.group(1)), int(m.group(2))
            plot = os.path.join(tmp, "align_chimera.plot")
            if os.path.exists(plot):
                d["plot_text"] = open(plot).read()
            if variant == "strict":
                out = d
            else:
                # the project's own -O3 -ffast-math build: keep what the 1e-5 A
                # tolerance is judged against
                for k in ("search_ssd", "refine_ssd", "refine_pairs", "search_pairs", "plot_text"):
                    if k in d:
                        out["fast_" + k] = d[k]
    out["flags"] = flags
    og = out["og"]
    out["og_sha256"] = hashlib.sha256(np.ascontiguousarray(og).tobytes()).hexdigest()
    out["og_count"] = og.shape[0]
    if not keep_og:
        del out["og"]
    os.makedirs(GOLD, exist_ok=True)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print("%-20s og=%d  refine_iters=%s  -> %s.npz (%d kB)" % (
        name, out["og_count"], out.get("refine_iters"), name,
        os.path.getsize(os.path.join(GOLD, name + ".npz")) // 1024))


# ---- a batch of domains, all against all (one reference run per pair) ----
def _bench_domains(count, seed=64):
    """Lengths and seeds of the first `count` of the synthetic domains that
    tools/allvsall_bench.py aligns all against all (lengths U[80, 200])."""
    import random
    rng = random.Random(seed)
    return [rng.randint(80, 200) for _ in range(count)], [2000 + k for k in range(count)]


# name -> (lengths, seed per domain, flags)
BATCHES = {
    "doms6": ([34, 48, 27, 55, 41, 38], [9001 + k for k in range(6)], "-fm 4 -HS -ir -r"),
    # the sizes of the all-against-all benchmark: its first 12 domains, 66 pairs
    "doms12_80to200": _bench_domains(12) + ("-fm 4 -HS -ir -r",),
}


def _batch_pair(args):
    exe, flags, la, sa, lb, sb = args
    with tempfile.TemporaryDirectory() as tmp:
        _stage(("synth", la, sa), os.path.join(tmp, "a.pdb"))
        _stage(("synth", lb, sb), os.path.join(tmp, "b.pdb"))
        env = dict(os.environ, REF_DUMP_OUT=os.path.join(tmp, "dump.txt"), REF_DUMP_TRACE="0")
        r = subprocess.run([exe] + flags.split() + ["a.pdb", "b.pdb"], cwd=tmp, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stdout[-2000:])
        d = parse_dump(os.path.join(tmp, "dump.txt"))
        m = re.search(r"During refinement, (\d+) new orientations were tested", r.stdout)
        return int(m.group(1)), d


def run_batch_case(name, workers=None):
    """One reference process per pair, `workers` of them at a time (the largest pairs first)."""
    from concurrent.futures import ThreadPoolExecutor
    lengths, seeds, flags = BATCHES[name]
    exe = os.path.join(HERE, "_ref", "ref_dump_strict")
    pairs = [(i, j) for i in range(len(lengths)) for j in range(i + 1, len(lengths))]
    width = max(lengths)
    ssd = np.full((len(pairs), width), np.float32(1e26), np.float32)
    al = np.full((len(pairs), width, width, 2), -1, np.int16)
    iters = np.zeros(len(pairs), np.int64)
    ogs = np.zeros(len(pairs), np.int64)
    xyz = {}
    order = sorted(range(len(pairs)), key=lambda p: -lengths[pairs[p][0]] * lengths[pairs[p][1]])
    jobs = [(exe, flags, lengths[pairs[p][0]], seeds[pairs[p][0]], lengths[pairs[p][1]], seeds[pairs[p][1]]) for p in order]
    workers = workers or int(os.environ.get("MAKE_GOLDEN_WORKERS", "1"))
    with ThreadPoolExecutor(max_workers=workers) as ex:
        results = list(ex.map(_batch_pair, jobs))
    for p, (it, d) in zip(order, results):
        i, j = pairs[p]
        iters[p] = it
        ogs[p] = d["og"].shape[0]
        rn = d["refine_ssd"].shape[0]
        ssd[p, d["n_min"] - 1:d["n_min"] - 1 + rn] = d["refine_ssd"]
        al[p, d["n_min"] - 1:d["n_min"] - 1 + rn, :d["n_max"]] = d["refine_pairs"]
        xyz[i], xyz[j] = d["xyza"], d["xyzb"]
    out = dict(lengths=np.array(lengths), seeds=np.array(seeds), flags=flags, pair_i=np.array([q[0] for q in pairs]),
               pair_j=np.array([q[1] for q in pairs]), refine_ssd=ssd, refine_pairs=al, refine_iters=iters,
               og_count=ogs)
    for k in range(len(lengths)):
        out["xyz%d" % k] = xyz[k]
    np.savez_compressed(os.path.join(GOLD, "batch_" + name + ".npz"), **out)
    print("batch_%-14s %d pairs -> %d kB" % (name, len(pairs),
                                             os.path.getsize(os.path.join(GOLD, "batch_" + name + ".npz")) // 1024))


# ---- expected output FILES of the stock binary (byte-compatibility test) ----
MSF_CASES = {
    "ship41x13": (("file", SHIPPED_A), ("file", SHIPPED_B), "-fm 4 -HS -ir -r"),
    "syn60x50_c1": (("synth", 60, 1234), ("synth", 50, 5678), "-fm 4 -HS -minN 0.33 -max-rmsd 3.5 -ir -r"),
    "syn70x80": (("synth", 70, 101), ("synth", 80, 202), "-fm 4 -HS -ir -r"),
    "syn60x50_of": (("synth", 60, 1234), ("synth", 50, 5678), "-fm 4 -HS -of 7.0,12 -max-dist 6.5 -minN 8 -ir -r"),
    "syn60x50_alt3": (("synth", 60, 1234), ("synth", 50, 5678), "-fm 4 -HS -alt-o 3,3.0,1.6 -ir -r-all"),
    "syn60x50_altr3": (("synth", 60, 1234), ("synth", 50, 5678), "-fm 4 -HS -alt-r 3,6,1.5 -ir -r-all"),
}


def mask_msf_date(text):
    """The MSF header embeds wall-clock date and time (pair_alignment.cc:1553-1574)."""
    return re.sub(r"(MSF:\s*\d+\s+Type:\s*\S+\s+).*?(\s+Check:)", r"\1<DATE>\2", text)


def run_msf_case(name):
    import json
    specA, specB, flags = MSF_CASES[name]
    exe = os.path.join(HERE, "_ref", "minrms_strict")
    with tempfile.TemporaryDirectory() as tmp:
        _stage(specA, os.path.join(tmp, "a.pdb"))
        _stage(specB, os.path.join(tmp, "b.pdb"))
        p = subprocess.run([exe] + flags.split() + ["a.pdb", "b.pdb"], cwd=tmp,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if p.returncode != 0:
            raise RuntimeError(p.stdout[-2000:])
        files = {}
        for f in sorted(os.listdir(tmp)):
            if f.startswith("align"):
                files[f] = mask_msf_date(open(os.path.join(tmp, f)).read())
        # what FirstPass prints per orientation that improved the table ("3 updates for orientation 17",
        # search_or_dyn3d.cc:1885-1889), without the time estimate a run of more than a second appends;
        # the later passes of -alt-r print the same kind of line and are left out (everything from the
        # first orientation number that does not exceed the one before it)
        updates = []
        for line in p.stdout.splitlines():
            m = re.match(r"^(\d+) updates for orientation (\d+)", line)
            if m:
                if updates and int(m.group(2)) <= updates[-1][1]:
                    break
                updates.append([int(m.group(1)), int(m.group(2))])
        out = {"flags": flags, "a_pdb": open(os.path.join(tmp, "a.pdb")).read(),
               "b_pdb": open(os.path.join(tmp, "b.pdb")).read(), "files": files, "updates": updates}
    with open(os.path.join(GOLD, "msf_" + name + ".json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("msf_%-16s %d files" % (name, len(files)))


if __name__ == "__main__":
    names = sys.argv[1:] or (list(CASES) + ["msf:" + k for k in MSF_CASES] + ["batch:" + k for k in BATCHES if k == "doms6"])
    for nm in names:
        if nm.startswith("full:"):              # never part of the default list: hours of CPU
            part = nm[5:].split(",")
            run_full_case(part[0], int(part[1]) if len(part) > 1 else None)
        elif nm.startswith("refine_from_search:"):      # Refine() on a split sweep's merged tables
            run_refine_from_search(nm.split(":", 1)[1])
        elif nm.startswith("check_refine_from_search:"):
            run_refine_from_search(nm.split(":", 1)[1], check_only=True)
        elif nm.startswith("msf:"):
            run_msf_case(nm[4:])
        elif nm.startswith("batch:"):
            run_batch_case(nm[6:])
        else:
            run_case(nm)
