"""Extracts the reference's frozen policy tables into tests/golden/policy_tables.npz (run in the build container only).

  * LABELS (crazyhouse / lichess / chess) ... /root/reference/engine/tests/legacyconstants.h:162-6737
  * FLAT_PLANE_IDX (same three modes) ....... /root/reference/engine/src/environments/chess_related/policymaprepresentation.h:39-6602
"""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/engine"


def blocks(text, start_pat):
    out = []
    for m in re.finditer(start_pat, text):
        end = text.index("};", m.end())
        out.append(text[m.end():end])
    return out


def main():
    legacy = open(os.path.join(REF, "tests/legacyconstants.h")).read()
    lab = blocks(legacy, r"const std::string LABELS\[\] = \{")
    assert len(lab) == 3
    labels = [re.findall(r'"([^"]+)"', b) for b in lab]          # order in file: crazyhouse, lichess, chess
    pm = open(os.path.join(REF, "src/environments/chess_related/policymaprepresentation.h")).read()
    fl = blocks(pm, r"const unsigned long FLAT_PLANE_IDX\[\] = \{")
    assert len(fl) == 3
    flat = [np.array([int(x) for x in re.findall(r"\d+", b)], dtype=np.uint16) for b in fl]
    for l, f, n in zip(labels, flat, (2272, 2316, 1968)):
        assert len(l) == n and len(f) == n, (len(l), len(f), n)
    out = os.path.join(ROOT, "tests", "golden", "policy_tables.npz")
    np.savez_compressed(out, labels_crazyhouse=np.array(labels[0]), labels_lichess=np.array(labels[1]),
                        labels_chess=np.array(labels[2]), flat_crazyhouse=flat[0], flat_lichess=flat[1], flat_chess=flat[2])
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
