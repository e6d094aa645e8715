"""Test-side FASTA reader with the rules of read_input (fbg.cpp:151-201)."""
import numpy as np


def read_fasta(path, elastic=True, gap_limit=1):
    with open(path, "rb") as fh:
        lines = fh.read().split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    ids, rows, cur = [lines[0][1:]], [], b""
    seqs = []
    for line in lines[1:]:
        if line[:1] == b">":
            seqs.append(cur)
            cur = b""
            ids.append(line[1:])
        else:
            cur += line
    seqs.append(cur)
    expected = len(seqs[0])
    for s in seqs:
        if len(s) != expected:
            continue
        if not elastic and gap_limit > 0:
            runs = [len(r) for r in s.split(b"-")]  # noqa: F841
            longest, run = 0, 0
            for c in s:
                run = run + 1 if c == 0x2D else 0
                longest = max(longest, run)
            if longest >= gap_limit:
                continue
        rows.append(s)
    if not rows:
        return None, ids
    return np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(len(rows), expected).copy(), ids


def write_fasta(path, msa, ids, width=None):
    with open(path, "wb") as fh:
        for i, row in enumerate(msa):
            fh.write(b">" + ids[i].encode() + b"\n")
            s = row.tobytes()
            if width:
                for k in range(0, len(s), width):
                    fh.write(s[k:k + width] + b"\n")
            else:
                fh.write(s + b"\n")
