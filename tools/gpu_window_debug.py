#!/usr/bin/env python3
"""pa_process_reads under small scan windows (PA_INGEST_WINDOW) against the default: first differing output line. Run on the GPU box."""
import os, sys, importlib
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import helpers
pa = helpers.pa
host = pa.HostIndex.build_fasta(str(helpers.FASTA), 20, 4)
al = pa.Pseudoaligner(host)
fq = str(helpers.FASTQ)
os.environ.pop("PA_INGEST_WINDOW", None)
pa.process_reads(fq, al, "/tmp/ref.txt", 3)
ref = open("/tmp/ref.txt").read().splitlines()
for w in sys.argv[1:] or ["300"]:
    os.environ["PA_INGEST_WINDOW"] = w
    n, fl = pa.process_reads(fq, al, "/tmp/got.txt", 3)
    got = open("/tmp/got.txt").read().splitlines()
    print("window", w, "n", n, "lines", len(got), "ref", len(ref))
    for i, (a, b) in enumerate(zip(got, ref)):
        if a != b:
            print("  first difference at line", i, "\n   got", got[max(0, i - 1):i + 2], "\n   ref", ref[max(0, i - 1):i + 2])
            break
    else:
        if len(got) != len(ref): print("  prefix equal; got ends", got[-2:], "ref ends", ref[-2:])
