import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import helpers, numpy as np
pa = helpers.pa
hi = pa.build_index(str(helpers.FASTA), 20, 8)
al = pa.Pseudoaligner(hi, 0)
o = helpers.Oracle(hi)
ids, seqs = helpers.read_fastq()
def run(reads, tag):
    res, coff, cids = al.map_batch(reads)
    ores, ocoff, oids, _ = o.map_reads(reads, 2, 1)
    bad = [(i, int(res['class_len'][i]), int(ores['class_len'][i])) for i in range(len(reads)) if res['class_len'][i] != ores['class_len'][i]]
    print(tag, "bad:", bad[:10])
filler = "A"*60
for pos in (0, 7, 8, 31, 32, 36, 63):
    reads = [filler]*64; reads[pos] = seqs[36]
    run(reads, "heavy at lane %d, fillers unmapped" % pos)
light = seqs[0]
for pos in (6, 36):
    reads = [light]*64; reads[pos] = seqs[36]
    run(reads, "heavy at lane %d, fillers light" % pos)
reads = [seqs[36]]*64
run(reads, "all heavy")
reads = [seqs[36]]*9
run(reads, "9 heavy")
