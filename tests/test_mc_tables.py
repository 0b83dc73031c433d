"""Structural checks of include/disn_mc_tables.h (data extracted from the
reference tool's tables) via the oracle library."""
import re
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _table(name):
    txt = open(os.path.join(ROOT, "include", "disn_mc_tables.h")).read()
    m = re.search(name + r"\[\d+\] = \{(.*?)\};", txt, re.S)
    return np.array([int(v) for v in m.group(1).replace("\n", "").split(",") if v.strip()])


def test_sizes_match_survey():
    assert len(_table("DISN_MC_CASE")) == 256
    assert len(_table("DISN_MC_FACE_TESTS")) == 256 * 7
    assert len(_table("DISN_MC_AMBIGUITY")) == 15 * 64
    assert len(_table("DISN_MC_SUB_OFFSET")) == 632          # 632 strings
    assert len(_table("DISN_MC_TRI")) == 11140               # ... 11 140 bytes


def test_case_table():
    case = _table("DISN_MC_CASE")
    assert case[0] == 0 and case[255] == 0
    assert (case[1:255] > 0).all() and case.max() == 14
    # complementary codes share a base case
    assert all(case[c] == case[255 - c] for c in range(256))


def test_subtable_histogram():
    start = _table("DISN_MC_SUB_START")
    nsub = np.diff(start)
    hist = {int(k): int(v) for k, v in zip(*np.unique(nsub[nsub > 0], return_counts=True))}
    assert hist == {1: 126, 2: 80, 4: 24, 8: 16, 5: 6, 46: 2}
    assert nsub[90] == 46 and nsub[165] == 46


def test_every_reachable_ambiguity_has_a_subtable():
    case, face, interior = _table("DISN_MC_CASE"), _table("DISN_MC_FACE_TESTS"), _table("DISN_MC_INTERIOR_TEST")
    amb, start = _table("DISN_MC_AMBIGUITY"), _table("DISN_MC_SUB_START")
    n_amb = 0
    n_face = 0
    for code in range(1, 255):
        nbits = face[7 * code] + (1 if interior[code] else 0)
        n_amb += nbits > 0
        n_face += face[7 * code] > 0
        nsub = start[code + 1] - start[code]
        targets = {amb[64 * case[code] + a] for a in range(1 << nbits)}
        assert max(targets) < nsub
        # -1 ("impossible") may only appear where the interior bit cannot vary freely
        assert min(targets) >= -1
    assert n_amb == 128          # = codes with more than one sub-table
    assert n_face == 120         # 120 codes carry face tests


def test_triangles_use_only_crossed_edges():
    """Every triangle corner of every sub-table lies on an edge whose two cube
    corners have different signs in that code (or is the centre vertex)."""
    edge = [(0, 1), (1, 2), (3, 2), (0, 3), (4, 5), (5, 6), (7, 6), (4, 7), (0, 4), (1, 5), (2, 6), (3, 7)]
    start, off, tri = _table("DISN_MC_SUB_START"), _table("DISN_MC_SUB_OFFSET"), _table("DISN_MC_TRI")
    for code in range(1, 255):
        crossed = {e for e, (a, b) in enumerate(edge) if ((code >> a) & 1) != ((code >> b) & 1)}
        for s in range(start[code], start[code + 1]):
            n, has_c = tri[off[s]], tri[off[s] + 1]
            ids = set(tri[off[s] + 2: off[s] + 2 + n].tolist())
            if code in (90, 165) and s - start[code] == 15:
                # placeholder in the reference's table (ten (1,1,1) triangles) for the
                # case-13 configuration whose interior test the tool does not implement
                assert len(ids) == 1
                continue
            assert ids - {12} <= crossed, (code, s)
            assert (12 in ids) <= bool(has_c)
            if not has_c:
                assert ids == crossed, (code, s)   # plain sub-tables use every crossed edge


def test_impossible_ambiguity_entries_are_exactly_the_contradictory_case13_patterns():
    """Over every (code, face/interior test outcome) the run-time can form -- 686 entries of the 15 x 64 table --
    `sub < 0` (App. B.3 "internal error", the voxel emits nothing) occurs only for the fully alternating codes 90
    and 165, and there exactly for the 18 outcomes in which one pair of opposite faces is joined through both faces
    while another pair is separated on both.  With L_f = [product of the face's positive corners > product of its
    negative corners], opposite faces satisfy L1*L3 = L2*L4 = L5*L6 (each equals the product of all eight
    magnitudes), so no real corner values produce such an outcome; float32 rounding can, within a few ulps of a
    four-way tie (tools/mc_bad_voxel_search.py, tests/test_gpu_mc_geometry.py).  mc.hip relies on "only 90 / 165"."""
    import itertools
    case, it = _table("DISN_MC_CASE"), _table("DISN_MC_INTERIOR_TEST")
    ft = _table("DISN_MC_FACE_TESTS").reshape(256, 7)
    amb = _table("DISN_MC_AMBIGUITY").reshape(15, 64)
    fv = [(0, 1, 5, 4), (1, 2, 6, 5), (3, 2, 6, 7), (0, 3, 7, 4), (0, 1, 2, 3), (4, 5, 6, 7)]   # App. A.3
    total, bad_codes = 0, {}
    for code in range(256):
        if case[code] == 0:
            continue
        bits = ft[code, 0] + (1 if it[code] else 0)
        for a in range(1 << bits):
            total += 1
            if amb[case[code], a] < 0:
                bad_codes.setdefault(code, []).append(a)
    assert total == 686 and sorted(bad_codes) == [90, 165] and all(len(v) == 18 for v in bad_codes.values())
    for code, bad in bad_codes.items():
        pos = [(code >> v) & 1 for v in range(8)]
        contradictory = set()
        for L in itertools.product([False, True], repeat=6):       # L[f]: positives' product > negatives' product
            a = 0
            for t in range(1, 7):
                f = int(ft[code, t])
                face = abs(f) - 1
                a_pos = pos[fv[face][0]] == 1
                raw = L[face] if a_pos else not L[face]                # A*C > B*D, A = the face's first corner
                a |= int((raw == (f > 0)) == a_pos) << (t - 1)
            pairs = [(L[0], L[2]), (L[1], L[3]), (L[4], L[5])]
            if any(x and y for x, y in pairs) and any(not x and not y for x, y in pairs):
                contradictory.add(a)
        assert contradictory == set(bad)
