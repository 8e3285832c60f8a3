"""The scan stage of pa_process_reads on the CPU tier (pa_fastq_scan_host, csrc/fastq.cpp): record count, header / sequence
extents against a Python reading of the same text — the acceptance rules of bio 1.5's fastq reader as process_reads applies
them (src/pseudoaligner.rs:430-447): LF / CRLF, no final line break, trailing blank lines, empty sequences (also as the LAST
record), wrapped records, gzip members, and what is refused."""
import gzip

import numpy as np
import pytest

import helpers

pa = helpers.pa


def _records(n, rng, empty_every=0):
    ids, seqs = [], []
    for i in range(n):
        ids.append("r%d/%s" % (i, "".join("abXY:_"[int(j)] for j in rng.integers(0, 6, int(rng.integers(0, 9))))))
        length = 0 if (empty_every and i % empty_every == 0) else int(rng.integers(1, 260))
        seqs.append("".join("ACGTNacgt"[int(j)] for j in rng.integers(0, 9, length)))
    return ids, seqs


def _text(ids, seqs, nl="\n", wrap=0, tail=" extra"):
    def w(t):
        return t if not wrap or not t else nl.join(t[j:j + wrap] for j in range(0, len(t), wrap))
    return "".join("@%s%s%s%s%s+%s%s%s" % (i, tail, nl, w(s), nl, nl, w("I" * len(s)), nl) for i, s in zip(ids, seqs))


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_scan_matches_python_reading(tmp_path, threads):
    rng = np.random.default_rng(11 + threads)
    ids, seqs = _records(3000, rng, empty_every=97)
    variants = {
        "lf": _text(ids, seqs),
        "crlf": _text(ids, seqs, nl="\r\n"),
        "no_final_newline": _text(ids, seqs)[:-1],
        "trailing_blank": _text(ids, seqs) + "\n\n\n",
        "crlf_trailing_blank": _text(ids, seqs, nl="\r\n") + "\r\n\r\n",
    }
    for name, text in variants.items():
        p = tmp_path / (name + ".fq")
        p.write_text(text, newline="")
        starts, hdr, seq, kind = pa.fastq_scan(str(p), threads)
        assert kind == 0 and len(starts) == len(ids), name
        raw = text.encode()
        for i in (0, 1, 2, 96, 97, 98, 1499, 2998, 2999):
            s0 = int(starts[i])
            assert raw[s0:s0 + 1] == b"@" and int(seq[i]) == len(seqs[i]), (name, i)
            header = raw[s0:s0 + int(hdr[i])].decode().rstrip("\r")
            assert header == "@" + ids[i] + " extra", (name, i)
            s1 = s0 + int(hdr[i]) + 1
            assert raw[s1:s1 + int(seq[i])].decode() == seqs[i], (name, i)
        assert np.array_equal(seq, np.array([len(s) for s in seqs], np.uint32)), name
        assert np.all(np.diff(starts.astype(np.int64)) > 0)


@pytest.mark.parametrize("window", [1, 300, 5000])
def test_scan_in_windows(tmp_path, monkeypatch, window):
    """the scan walks a mapped file a window at a time (PA_INGEST_WINDOW; pa_process_reads packs a window's records while the next is
    still unread): the records found are those of the one-window scan, whatever the window cuts through"""
    rng = np.random.default_rng(3)
    ids, seqs = _records(2000, rng, empty_every=53)
    for name, text in (("lf", _text(ids, seqs)), ("crlf", _text(ids, seqs, nl="\r\n")), ("no_final_newline", _text(ids, seqs)[:-1]),
                       ("trailing_blank", _text(ids, seqs) + "\n\n\n"), ("empty_last", _text(ids, seqs) + "@last\n\n+\n\n\n")):
        p = tmp_path / (name + ".fq")
        p.write_text(text, newline="")
        monkeypatch.delenv("PA_INGEST_WINDOW", raising=False)
        s0, h0, q0, k0 = pa.fastq_scan(str(p), 3)
        monkeypatch.setenv("PA_INGEST_WINDOW", str(window))
        s1, h1, q1, k1 = pa.fastq_scan(str(p), 3)
        assert k0 == k1 == 0 and np.array_equal(s0, s1) and np.array_equal(h0, h1) and np.array_equal(q0, q1), (name, window)
        assert len(s0) == len(ids) + (name == "empty_last")
    # four-line records first, wrapped ones behind them: the rest of the file is rewritten from the first window that is not in shape
    half = 1000
    mixed = _text(ids[:half], seqs[:half]) + _text(ids[half:], seqs[half:], wrap=30)
    p = tmp_path / "mixed.fq"
    p.write_text(mixed, newline="")
    starts, hdr, seq, kind = pa.fastq_scan(str(p), 3)
    assert kind == 2 and len(starts) == len(ids)
    assert np.array_equal(seq, np.array([len(s) for s in seqs], np.uint32))


def test_scan_last_record_with_empty_sequence(tmp_path):
    rng = np.random.default_rng(5)
    ids, seqs = _records(50, rng)
    base = _text(ids, seqs)
    for name, ending in (("nl_nl", "@last\n\n+\n\n"), ("crlf", "@last\r\n\r\n+\r\n\r\n"), ("no_quality_line", "@last\n\n+"), ("blank_lines_after", "@last\n\n+\n\n\n\n")):
        p = tmp_path / (name + ".fq")
        p.write_text(base + ending, newline="")
        starts, hdr, seq, kind = pa.fastq_scan(str(p), 2)
        assert len(starts) == 51 and seq[50] == 0 and int(starts[50]) == len(base.encode()), name


def test_scan_wrapped_and_gzip(tmp_path):
    rng = np.random.default_rng(6)
    ids, seqs = _records(400, rng)
    seqs = [s or "A" for s in seqs]
    p = tmp_path / "wrapped.fq"
    p.write_text(_text(ids, seqs, wrap=23), newline="")
    starts, hdr, seq, kind = pa.fastq_scan(str(p), 3)
    assert kind == 2 and len(starts) == 400 and np.array_equal(seq, np.array([len(s) for s in seqs], np.uint32))
    raw = _text(ids, seqs).encode()
    g = tmp_path / "two_members.fq.gz"
    g.write_bytes(gzip.compress(raw[: len(raw) // 3]) + gzip.compress(raw[len(raw) // 3:]))
    starts, hdr, seq, kind = pa.fastq_scan(str(g), 3)
    assert kind == 1 and len(starts) == 400 and np.array_equal(seq, np.array([len(s) for s in seqs], np.uint32))
    assert raw[int(starts[7]):int(starts[7]) + int(hdr[7])].decode() == "@" + ids[7] + " extra"
    empty = tmp_path / "empty.fq"
    empty.write_text("")
    assert len(pa.fastq_scan(str(empty), 2)[0]) == 0


@pytest.mark.parametrize("name,text", [
    ("no_at", "r1\nACGT\n+\nIIII\n"),
    ("blank_inside", "@r1\nACGT\n+\nIIII\n\n@r2\nACGT\n+\nIIII\n"),
    ("ends_in_sequence", "@r1\nACGT\n+\nIIII\n@r2\nACGT\n"),
    ("ends_after_plus_with_sequence", "@r1\nACGT\n+\nIIII\n@r2\nACGT\n+"),
    ("wrapped_truncated", "@r1\nACGT\nACGT\n+\nIIII\nIIII\n@r2\nAC\nGT\n"),
])
def test_scan_refuses(tmp_path, name, text):
    p = tmp_path / (name + ".fq")
    p.write_text(text)
    with pytest.raises(pa.PaError):
        pa.fastq_scan(str(p), 2)
    with pytest.raises(pa.PaError):
        pa.fastq_scan(str(tmp_path / "missing.fq"), 2)
