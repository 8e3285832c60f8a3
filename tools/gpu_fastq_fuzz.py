"""Fuzz of pa_process_reads' host stages (csrc/fastq.cpp: line-break scan, record positions, 16-base packing, formatter fast
paths) against the oracle's tuples: random slices of small.fq rewritten with random ids (quotes, backslashes, control bytes,
tabs, trailing blanks), random read lengths (0..300: word counts change between batches), lower case / N / IUPAC letters,
LF or CRLF, with or without a final line break, trailing blank lines, wrapped records, gzip; random batch sizes and thread
counts. Usage (GPU box): python tools/gpu_fastq_fuzz.py [files [first seed]]"""
import gzip, importlib, os, sys, tempfile
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import helpers
pa = helpers.pa

def rust_debug(s):   # impl Debug for str, for the bytes this fuzz produces (ASCII)
    out = ['"']
    for ch in s:
        c = ord(ch)
        if ch == '"': out.append('\\"')
        elif ch == "\\": out.append("\\\\")
        elif ch == "\t": out.append("\\t")
        elif ch == "\r": out.append("\\r")
        elif ch == "\n": out.append("\\n")
        elif c < 0x20 or c == 0x7f: out.append("\\u{%x}" % c)
        else: out.append(ch)
    out.append('"')
    return "".join(out)

def run(nfiles, first=0):
    _, base = helpers.read_fastq()
    bad = 0
    aligners = {}
    with tempfile.TemporaryDirectory() as td:
        for seed in range(first, first + nfiles):
            rng = np.random.default_rng(7000 + seed)
            k = (20, 24, 31)[seed % 3]
            if k not in aligners:
                aligners[k] = pa.Pseudoaligner(pa.build_index(str(helpers.FASTA), k, 8), 0)
            a = aligners[k]
            n = int(rng.integers(0, 900))
            ids, seqs = [], []
            for i in range(n):
                s = base[int(rng.integers(0, len(base)))]
                kind = int(rng.integers(0, 8))
                if kind == 0: s = s[: int(rng.integers(0, len(s) + 1))]
                elif kind == 1: s = (s + base[int(rng.integers(0, len(base)))] + s)[: int(rng.integers(1, 301))]
                elif kind == 2: s = s.lower()
                elif kind == 3 and len(s) > 30:
                    p = int(rng.integers(0, len(s))); s = s[:p] + "NRYKMnry-."[int(rng.integers(0, 10))] + s[p + 1:]
                alphabet = 'abcXYZ019_:/#"\\\x01\x1f\x7f\t|'
                rid = "".join(alphabet[int(j)] for j in rng.integers(0, len(alphabet), int(rng.integers(1, 24))))
                rid = rid.replace(" ", "_")
                if rid[0] in "\t": rid = "x" + rid
                ids.append(rid); seqs.append(s)
            nl = "\r\n" if seed % 4 == 1 else "\n"
            wrap = seed % 5 == 2
            def w(t, width):   # bio reads as many quality lines as sequence lines: both wrapped alike
                if not wrap or not t: return t
                return nl.join(t[j:j + width] for j in range(0, len(t), width))
            widths = rng.integers(7, 80, max(n, 1))
            tail = ["", " extra words", "\tkept tab", "  "][seed % 4]
            text = "".join("@%s%s%s%s%s+%s%s%s" % (i, tail if tail != "\tkept tab" else "", nl, w(s, int(wd)), nl, nl, w("I" * len(s), int(wd)), nl)
                           for i, s, wd in zip(ids, seqs, widths))
            if seed % 7 == 3 and text: text = text[: -len(nl)]
            if seed % 7 == 5: text += nl * int(rng.integers(1, 4))
            # expected tuples: ids are cut at the first space after trailing whitespace is trimmed; other bytes -> A
            o_res, o_coff, o_ids, _ = helpers.Oracle(a.host).map_reads(["".join(c if c in "ACGT" else "A" for c in s.upper()) for s in seqs], 2, 4)
            want = []
            for i, rid in enumerate(ids):
                shown = (rid + (tail if tail != "\tkept tab" else "")).rstrip(" \t\r\n").split(" ")[0]
                cl = o_ids[int(o_coff[i]):int(o_coff[i + 1])].tolist()
                flag = bool(o_res["mapped"][i]) and o_res["coverage"][i] >= 32 and not cl
                want.append("(%s, %s, [%s], %d)" % ("true" if flag else "false", rust_debug(shown), ", ".join(map(str, cl)), o_res["coverage"][i] if o_res["mapped"][i] else 0))
            if wrap and any(len(s) == 0 for s in seqs):
                continue   # bio's reader cannot tell an empty wrapped sequence from a missing one: not a case
            path = os.path.join(td, "f%d.fq" % seed)
            raw = text.encode("latin-1")
            if seed % 6 == 4:
                path += ".gz"; raw = gzip.compress(raw[: len(raw) // 2]) + gzip.compress(raw[len(raw) // 2:])
            open(path, "wb").write(raw)
            os.environ["PA_INGEST_BATCH"] = str(int(rng.choice([64, 128, 320, 1024, 1 << 22])))
            out = os.path.join(td, "o.txt")
            try:
                got_n, flagged = pa.process_reads(path, a, out, int(rng.integers(1, 9)))
                got = open(out, "rb").read().decode("latin-1").split("\n")[:-1] if n else []
                ok = got_n == n and got == want
            except pa.PaError as e:
                ok = False; print("  error:", e)
            if not ok:
                bad += 1
                print("MISMATCH seed %d (n=%d k=%d nl=%r wrap=%s gz=%s)" % (seed, n, k, nl, wrap, path.endswith(".gz")))
                print("   lines got %d want %d, batch %s" % (len(got), len(want), os.environ["PA_INGEST_BATCH"]))
                for j, (x, y) in enumerate(zip(got, want)):
                    if x != y: print("   line", j, "seq len", len(seqs[j]), "\n   got ", x[:150], "\n   want", y[:150]); break
    print("fastq fuzz: %d files, mismatching %d" % (nfiles, bad))
    return bad

if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 150, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
