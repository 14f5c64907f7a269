"""Interval pairs for fragment matching from HELIX / SHEET records.

Host-side mirror of what the reference does before orientation generation with
`-hs` (bin/minrms/minrms_parser.cc:641-675): every HELIX interval of structure 1
is paired with every HELIX interval of structure 2, then the same for SHEET
strands (ConvertHelixSheetRecordsToIntervalNtuples, lib/biopolymer/intervals.cc:
335-568); an interval shorter than the fragment size is widened first
(ExtendInterval, :225-330).  The result is the `interval_pairs` argument of
`minrms_engine_gen_orientations`: rows {first1, last1, first2, last2} of 0-based
inclusive residue indices into the structures handed to the engine.
"""
import numpy as np


def residue_ids(pdb_text, atom_name=" CA "):
    """(chain, seqNum, insertCode) of every residue, in file order, taken from
    the ATOM records of one atom per residue."""
    ids = []
    for line in pdb_text.splitlines():
        if line.startswith("ATOM") and line[12:16] == atom_name:
            ids.append((line[21], int(line[22:26]), line[26]))
    return ids


def helix_sheet_records(pdb_text):
    """([helix intervals], [sheet intervals]) as pairs of residue ids, in file order
    (lib/biopolymer/intervals.cc:115-134: initial and terminal residue of a record)."""
    helices, sheets = [], []
    for line in pdb_text.splitlines():
        line = line.ljust(40)
        if line.startswith("HELIX"):
            helices.append(((line[19], int(line[21:25]), line[25]), (line[31], int(line[33:37]), line[37])))
        elif line.startswith("SHEET"):
            sheets.append(((line[21], int(line[22:26]), line[26]), (line[32], int(line[33:37]), line[37])))
    return helices, sheets


def extend_interval(first, last, n_res, min_size):
    """ExtendInterval (lib/biopolymer/intervals.cc:225-330) on residue indices."""
    if last < first:
        raise ValueError("Subset-Match Interval cannot have start > end.")
    if first <= last - min_size + 1:
        return first, last
    needed = min_size - (last - first + 1)
    pad = needed // 2 if needed % 2 == 0 else needed // 2 + 1
    shift_right = shift_left = 0
    if first >= pad:
        first -= pad
    else:
        shift_right = pad - first
        first = 0
    if n_res - 1 - last >= pad:
        last += pad
    else:
        shift_left = pad - (n_res - 1 - last)
        last = n_res - 1
    if shift_right and shift_left:
        first, last = 0, n_res - 1
    else:
        first -= shift_left
        last += shift_right
    return first, last


def _usable(records, ids, min_size, what):
    index = {rid: k for k, rid in enumerate(ids)}
    out = []
    for a, b in records:
        ia, ib = index.get(a), index.get(b)
        if ia is not None and ib is not None:
            out.append(extend_interval(ia, ib, len(ids), min_size))
        elif (ia is None) != (ib is None):       # intervals.cc:420-448
            raise ValueError("Invalid %s record: one end of %r - %r is not among the residues in use"
                             % (what, a, b))
    return out


def helix_sheet_interval_pairs(pdb_text_1, pdb_text_2, ids_1=None, ids_2=None, fragment_size=4):
    """int32 [P][4] interval pairs for `-hs`; raises if a structure has neither
    HELIX nor SHEET records usable with the other (minrms_parser.cc:662-669)."""
    ids_1 = residue_ids(pdb_text_1) if ids_1 is None else ids_1
    ids_2 = residue_ids(pdb_text_2) if ids_2 is None else ids_2
    h1, s1 = helix_sheet_records(pdb_text_1)
    h2, s2 = helix_sheet_records(pdb_text_2)
    pairs = []
    for what, r1, r2 in (("HELIX", h1, h2), ("SHEET", s1, s2)):
        u1 = _usable(r1, ids_1, fragment_size, what)
        u2 = _usable(r2, ids_2, fragment_size, what)
        for q in range(len(u1) * len(u2)):       # structure 1's index runs fastest (:491-499)
            a = u1[q % len(u1)]
            b = u2[q // len(u1)]
            pairs.append([a[0], a[1], b[0], b[1]])
    if not pairs:
        raise ValueError("At least one of the PDB files contains no HELIX or SHEET records. "
                         "Disable helix-sheet matching (-HS) or add the records.")
    return np.asarray(pairs, np.int32)
