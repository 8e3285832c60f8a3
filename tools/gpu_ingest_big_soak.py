"""pa_process_reads at PRODUCTION sizes, content-checked (the bench's FASTQ leg writes to /dev/null): 8 M reads of config 3 as a 2.5 GB FASTQ file,
mapped over and over — default 64 MiB windows, 16 / 128 MiB windows, 1 / 2 / 3 lanes, every window through the host's scan — every output byte-identical
to the first, and the first and last 200 k tuples of it equal to the oracle's on the same reads. A race between the lane's streams, a window seam, or a page
of the file's mapping dropped too early would show as a different sha256. usage (GPU box): python tools/gpu_ingest_big_soak.py [repeats]"""
import hashlib, importlib, os, sys, time
from pathlib import Path
import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import helpers   # noqa: E402
pa = helpers.pa
import bench     # noqa: E402  (write_fastq: the file the bench's leg maps)


def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        while True:
            b = f.read(1 << 24)
            if not b:
                break
            h.update(b)
    return h.hexdigest()


def main():
    repeats = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    n, L, wpr, seed = 8_000_000, 150, 5, 2
    tx = pa.Txome.synthesize(58000, 203000, 7)
    host = pa.HostIndex.from_txome_device(tx, 24, 0)
    al = pa.Pseudoaligner(host)
    d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    fq, out = os.path.join(d, "pa_big_soak.fq"), os.path.join(d, "pa_big_soak.out")
    size = bench.write_fastq(tx, fq, n, L, seed, wpr, np)
    print("file: %d reads, %.2f GB" % (n, size / 1e9), flush=True)
    bad = 0
    try:
        got, flagged = pa.process_reads(fq, al, out, 16)
        assert got == n
        want = sha(out)
        # the first and last 200 k tuples against the oracle on the same reads
        oracle = helpers.Oracle(host)
        lines = open(out, "rb").read().split(b"\n")
        assert lines[-1] == b"" and len(lines) == n + 1
        for first in (0, n - 200_000):
            m = 200_000
            tiles, lens = tx.simulate_host(L, seed, m, 0, first, wpr)
            res, coff, cids, _ = oracle.map_tiles(tiles, lens, wpr, 2, 16)
            for i in range(m):
                cl = cids[int(coff[i]):int(coff[i + 1])].tolist()
                mapped = bool(res["mapped"][i])
                flag = mapped and res["coverage"][i] >= 32 and not cl
                exp = '(%s, "r%09d", [%s], %d)' % ("true" if flag else "false", first + i, ", ".join(map(str, cl)), res["coverage"][i] if mapped else 0)
                if lines[first + i].decode() != exp:
                    bad += 1
                    if bad < 5:
                        print("MISMATCH read %d: %r vs %r" % (first + i, lines[first + i], exp))
        print("first / last 200 k tuples against the oracle: mismatching %d" % bad, flush=True)
        del lines

        def again(what, lanes=1, env=None):
            nonlocal bad
            old = {}
            for k, v in (env or {}).items():
                old[k] = os.environ.get(k); os.environ[k] = v
            try:
                t0 = time.time()
                g, _ = pa.process_reads(fq, al, out, 16) if lanes == 1 else pa.process_reads_multi(fq, [al] * lanes, out, 16)
                dt = time.time() - t0
            finally:
                for k, v in old.items():
                    if v is None:
                        del os.environ[k]
                    else:
                        os.environ[k] = v
            ok = g == n and sha(out) == want
            if not ok:
                bad += 1
                print("DIFFERENT OUTPUT: %s" % what, flush=True)
            return dt
        ts = [again("default, call %d" % i) for i in range(repeats)]
        print("default windows x %d: identical so far %s; %.1f - %.1f M reads/s to a file in %s" % (repeats, bad == 0, n / max(ts) / 1e6, n / min(ts) / 1e6, d), flush=True)
        for lanes in (2, 3):
            for i in range(max(repeats // 6, 2)):
                again("%d lanes, call %d" % (lanes, i), lanes)
        for w in (16 << 20, 128 << 20, 5_000_000):
            again("window %d" % w, 1, {"PA_INGEST_WINDOW": str(w)})
            again("window %d, 2 lanes" % w, 2, {"PA_INGEST_WINDOW": str(w)})
        again("host scan", 1, {"PA_INGEST_HOST_SCAN": "1"})
        print("big-file soak: %d calls, problems %d" % (repeats + 2 * max(repeats // 6, 2) + 7, bad), flush=True)
    finally:
        for p in (fq, out):
            if os.path.exists(p):
                os.unlink(p)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
