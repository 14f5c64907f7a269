"""Seeded synthetic CA-only protein-like chains.

The real PDB entries the benchmark configurations name (1HHO/1MBO, a TIM-barrel
pair, a kinase pair, SCOP domains) cannot be fetched offline, so every workload
here uses chains made by this generator: alternating helix-like (rise 1.5 A,
radius 2.3 A, 100 deg/residue) and strand-like (rise 3.3 A) segments of 6-14
residues with random axes that are pulled back toward the origin beyond 15 A.
Coordinates are rounded through the PDB text format (%8.3f), exactly what a PDB
reader would hand to the search.
"""
import math
import random

import numpy as np

_AA = ['GLY', 'ALA', 'SER', 'CYS', 'VAL', 'THR', 'ILE', 'PRO', 'MET', 'ASP',
       'ASN', 'LEU', 'LYS', 'GLU', 'GLN', 'ARG', 'HIS', 'PHE', 'TYR', 'TRP']


def synth_pdb_text(n, seed, chain='A', with_secondary_structure=False):
    """PDB text (ATOM records, CA only) of an n-residue synthetic chain.
    with_secondary_structure: also emit a HELIX / SHEET record for every
    helix-like / strand-like segment of at least 4 residues (same ATOM records)."""
    rng = random.Random(seed)
    pts = []
    segments = []                      # (helix?, first residue number, last residue number)
    x = [0.0, 0.0, 0.0]
    while len(pts) < n:
        seglen = rng.randint(6, 14)
        helix = rng.random() < 0.5
        ax = [rng.gauss(0, 1) for _ in range(3)]
        r = math.sqrt(sum(c * c for c in x))
        if r > 15:
            ax = [ax[k] - x[k] / r * 2.0 for k in range(3)]
        na = math.sqrt(sum(c * c for c in ax))
        ax = [c / na for c in ax]
        t = [1, 0, 0] if abs(ax[0]) < 0.9 else [0, 1, 0]
        u = [ax[1] * t[2] - ax[2] * t[1], ax[2] * t[0] - ax[0] * t[2], ax[0] * t[1] - ax[1] * t[0]]
        nu = math.sqrt(sum(c * c for c in u))
        u = [c / nu for c in u]
        v = [ax[1] * u[2] - ax[2] * u[1], ax[2] * u[0] - ax[0] * u[2], ax[0] * u[1] - ax[1] * u[0]]
        seg_first = len(pts) + 1
        for k in range(seglen):
            if len(pts) >= n:
                break
            if helix:
                rise, rad, ang = 1.5, 2.3, math.radians(100.0) * k
            else:
                rise, rad, ang = 3.3, 0.9, math.pi * k
            pts.append([x[j] + ax[j] * rise * k + rad * (math.cos(ang) * u[j] + math.sin(ang) * v[j])
                        for j in range(3)])
        x = [pts[-1][j] + ax[j] * 3.0 for j in range(3)]
        segments.append((helix, seg_first, len(pts)))
    lines = []
    names = []
    for i, p in enumerate(pts):
        names.append(_AA[rng.randrange(20)])
        lines.append("ATOM  %5d  CA  %3s %1s%4d    %8.3f%8.3f%8.3f  1.00  0.00           C\n"
                     % (i + 1, names[-1], chain, i + 1, p[0], p[1], p[2]))
    lines.append("END\n")
    if with_secondary_structure:
        head = []
        nh = ns = 0
        for helix, a, b in segments:
            if b - a + 1 < 4:
                continue
            if helix:
                nh += 1
                head.append("HELIX  %3d %3d %3s %1s %4d  %3s %1s %4d  1%36d\n"
                            % (nh, nh, names[a - 1], chain, a, names[b - 1], chain, b, b - a + 1))
            else:
                ns += 1
                head.append("SHEET  %3d %3s%2d %3s %1s%4d  %3s %1s%4d  0\n"
                            % (ns, "S1", 1, names[a - 1], chain, a, names[b - 1], chain, b))
        lines = head + lines
    return "".join(lines)


def synth_backbone_pdb_text(n, seed, chain='A'):
    """PDB text with N, CA and C records per residue: the CA trace of
    synth_pdb_text(n, seed) plus two satellites at seeded offsets of about 1.5 A.
    Only meant to exercise more than one atom per residue; not real geometry."""
    ca = pdb_text_to_ca(synth_pdb_text(n, seed, chain))
    rng = random.Random(seed * 7919 + 13)
    lines = []
    serial = 1
    for i in range(n):
        res = _AA[rng.randrange(20)]
        offs = {}
        for name in (" N  ", " C  "):
            v = [rng.gauss(0, 1) for _ in range(3)]
            nv = math.sqrt(sum(c * c for c in v))
            offs[name] = [1.5 * c / nv for c in v]
        for name in (" N  ", " CA ", " C  "):
            o = offs.get(name, [0.0, 0.0, 0.0])
            p = [float(ca[i][k]) + o[k] for k in range(3)]
            lines.append("ATOM  %5d %4s %3s %1s%4d    %8.3f%8.3f%8.3f  1.00  0.00           %s\n"
                         % (serial, name, res, chain, i + 1, p[0], p[1], p[2], name.strip()[0]))
            serial += 1
    lines.append("END\n")
    return "".join(lines)


def pdb_text_to_ca(text):
    """float32 [n,3] CA coordinates of the ATOM records in PDB text."""
    out = []
    for line in text.splitlines():
        if line.startswith("ATOM") and line[12:16] == " CA ":
            out.append([float(line[30:38]), float(line[38:46]), float(line[46:54])])
    return np.asarray(out, dtype=np.float32)


def synth_ca(n, seed):
    """float32 [n,3] coordinates of the synthetic chain (n, seed)."""
    return pdb_text_to_ca(synth_pdb_text(n, seed))


def synth_pair(n1, n2, seed1=1234, seed2=5678):
    """The benchmark pair: chains (n1, seed 1234) and (n2, seed 5678)."""
    return synth_ca(n1, seed1), synth_ca(n2, seed2)
