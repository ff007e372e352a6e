"""The oracle against every known-answer vector available for the path."""
import numpy as np


def hx(b):
    return " ".join("%02x" % x for x in b)


def test_run_write_block_rule(oracle, golden):
    g = golden["run_write"]
    for row in g["rows"]:
        for off in (0, 61, 62, 63):
            got = oracle.run_write(g["comp"], row["len"], prefill=off)
            assert hx(got) == row["off%d" % off], (row["len"], off)
            # Run::read decodes every row back to the same total length.
            runs = oracle.run_decode(got)
            assert all(c == g["comp"] for c, _ in runs)
            assert sum(l for _, l in runs) == row["len"]


def test_bytecode(oracle, golden):
    for row in golden["bytecode_write"]["rows"]:
        enc = oracle.bytecode_write(row["value"])
        assert hx(enc) == row["hex"]
        assert oracle.bytecode_read(enc) == (row["value"], len(enc))


def test_get_bounds(oracle, golden):
    rows = golden["get_bounds"]["rows"]
    assert oracle.get_bounds(rows[0]["first"], rows[0]["last"], rows[0]["blocks"]) == [tuple(x) for x in rows[0]["bounds"]]
    b = oracle.get_bounds(rows[1]["first"], rows[1]["last"], rows[1]["blocks"])
    assert len(b) == rows[1]["count"]
    assert b[:3] == [tuple(x) for x in rows[1]["head"]]
    assert b[-1:] == [tuple(x) for x in rows[1]["tail"]]
    # contiguous cover
    assert all(b[k][1] + 1 == b[k + 1][0] for k in range(len(b) - 1))


def test_fnv(oracle, golden):
    assert "%016x" % oracle.fnv1a(bytes(golden["fnv1a"]["bytes"])) == golden["fnv1a"]["hash"]


def test_worked_merge(oracle, golden):
    g = golden["worked_merge"]
    a = oracle.FMI.from_text(oracle.text_from_strings(g["A"]))
    b = oracle.FMI.from_text(oracle.text_from_strings(g["B"]))
    assert "".join(oracle.CHARS[c] for c in a.symbols) == g["bwt_A"]
    assert "".join(oracle.CHARS[c] for c in b.symbols) == g["bwt_B"]

    for kw in (dict(threads=1), dict(threads=2, sequence_blocks=3), dict(threads=1, run_buffer_size=1, merge_buffers=2)):
        ranks, counts, _ = oracle.search(a, b, **kw)
        assert [[int(r), int(c)] for r, c in zip(ranks, counts)] == g["ra_runs"]
        assert list(oracle.ra_from_runs(ranks, counts)) == g["ra"]

    ra = np.array(g["ra"], dtype=np.uint64)
    bits = np.zeros(a.bases + b.bases, dtype=np.uint8)
    bits[np.arange(b.bases, dtype=np.uint64) + ra] = 1
    assert "".join(str(x) for x in bits) == g["interleaving_bits"]

    m, _ = oracle.merge(a.clone(), b.clone(), threads=1)
    assert "".join(oracle.CHARS[c] for c in m.symbols) == g["bwt_AB"]
    assert (m.sequences, m.bases) == (g["sequences"], g["bases"])
    assert hx(m.data) == g["data_hex"]
    assert list(m.C) == g["C"]
    assert "%016x" % m.hash == g["hash"]

    # == direct BWT of the concatenated collection
    ab = oracle.FMI.from_text(oracle.text_from_strings(g["A"] + g["B"]))
    assert hx(ab.data) == g["data_hex"]
