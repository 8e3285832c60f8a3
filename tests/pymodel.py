"""An INDEPENDENT model of Pseudoaligner::map_read, written from the behavioural spec of SURVEY.md §3.2 (which was read off
/root/reference/src/pseudoaligner.rs:64-418) — NOT from oracle/pa_oracle.c and not from the product. TEST INFRASTRUCTURE ONLY.

Plain Python on plain Python data: node sequences as str, the k-mer dictionary as a dict from k-mer strings, edges resolved the
way the debruijn crate resolves them (appendix A: by looking the neighbour k-mer up), classes as sorted lists, one base compared
per loop iteration exactly as the spec words it. Slow and obvious on purpose: it exists so that the C oracle's restatement of the
walk — left extension incl. the quirk of :129, re-seek, premature-break accounting — is pinned by a second, structurally
different statement of the same spec (tests/test_independent_model.py diffs the two on thousands of reads with errors).
"""

BASES = "ACGT"


class Model:
    def __init__(self, arrays):
        """arrays: HostIndex.arrays() — the flat graph (node sequences, Exts bytes, colours) and the class table"""
        import numpy as np
        a = arrays
        self.k = int(a["k"])
        self.seq, self.exts, self.colour = [], [], []
        lut = np.frombuffer(BASES.encode(), np.uint8)
        for n in range(int(a["num_nodes"])):
            s, l = int(a["node_start"][n]), int(a["node_len"][n])
            pos = np.arange(s, s + l, dtype=np.int64)
            codes = ((a["node_seq"][pos >> 5] >> ((pos & 31) * 2).astype(np.uint64)) & np.uint64(3)).astype(np.uint8)
            self.seq.append(lut[codes].tobytes().decode())
            self.exts.append(int(a["node_exts"][n]))
            self.colour.append(int(a["node_colour"][n]))
        off = a["ec_offset"]
        self.classes = [a["ec_ids"][int(off[c]):int(off[c + 1])].tolist() for c in range(int(a["num_classes"]))]
        k = self.k
        self.index, self.first, self.last = {}, {}, {}          # dbg_index: k-mer -> (node, offset); first / last k-mer -> node
        for n, s in enumerate(self.seq):
            for o in range(len(s) - k + 1):
                self.index[s[o:o + k]] = (n, o)
            self.first[s[:k]] = n
            self.last[s[-k:]] = n
        self.events = dict(left_ext=0, q1=0, left_hops=0, reseek=0, premature_fwd=0, premature_left=0, hops=0)

    # Exts: low nibble = right extensions by base code, high nibble = left extensions
    def has_right(self, n, b):
        return (self.exts[n] >> BASES.index(b)) & 1 == 1

    def has_left(self, n, b):
        return (self.exts[n] >> (4 + BASES.index(b))) & 1 == 1

    def r_edge(self, n, b):      # the node whose first k-mer is (last k-mer of n)[1:] + b
        return self.first[self.seq[n][-(self.k - 1):] + b]

    def l_edge(self, n, b):      # the node whose last k-mer is b + (first k-mer of n)[:-1]
        return self.last[b + self.seq[n][:self.k - 1]]

    def map_read_to_nodes(self, read, allowed):
        """spec steps 1-6 -> (None | (coverage, mismatches), nodes)"""
        K, L, ev = self.k, len(read), self.events
        nodes = []
        thr = int(0.2 * L)                                        # step 1
        if L < K:
            return None, nodes
        last = L - K
        coverage = mismatches = 0

        def find(kp):                                             # step 2
            while kp <= last:
                hit = self.index.get(read[kp:kp + K])
                if hit is not None:
                    return hit, kp
                kp += 3
            return None, kp

        hit, kp = find(0)                                         # step 3
        if hit is not None and kp >= thr:                         # step 4: left extension
            ev["left_ext"] += 1
            nid, off = hit
            lp, pn = kp - 1, nid
            po = off - 1 if off > 0 else 0
            if off == 0:
                ev["q1"] += 1
            while True:
                mx = min(lp + 1, po + 1)
                matched = snp = 0
                premature = False
                for idx in range(mx):
                    if self.seq[pn][po - idx] != read[lp - idx]:
                        mismatches += 1
                        snp += 1
                        if snp > allowed:
                            premature = True
                            break
                    matched += 1
                    coverage += 1
                if premature:
                    ev["premature_left"] += 1
                if lp + 1 - matched == 0 or premature:
                    break
                lp -= matched
                b = read[lp]
                if self.has_left(pn, b):
                    pn = self.l_edge(pn, b)
                    po = len(self.seq[pn]) - K
                    nodes.append(pn)
                    ev["left_hops"] += 1
                else:
                    break
        if hit is not None:                                       # step 5: forward search (kp <= last holds for a hit)
            nid, off = hit
            while True:
                kp += K
                coverage += K
                nodes.append(nid)
                ro = off + K
                mx = min(L - kp, len(self.seq[nid]) - ro)
                matched = snp = 0
                premature = False
                for idx in range(mx):
                    if self.seq[nid][ro + idx] != read[kp + idx]:
                        mismatches += 1
                        snp += 1
                        if snp > allowed:
                            premature = True
                            break
                    matched += 1
                    coverage += 1
                kp += matched
                if premature:
                    ev["premature_fwd"] += 1
                if kp >= L:
                    break
                b = read[kp]
                if not premature and self.has_right(nid, b):
                    nid = self.r_edge(nid, b)
                    off = 0
                    kp -= K - 1
                    coverage -= K - 1
                    ev["hops"] += 1
                else:
                    if kp > last:
                        break
                    ev["reseek"] += 1
                    hit, kp = find(kp)
                    if hit is None:
                        break
                    nid, off = hit
        if not nodes:                                             # step 6
            assert coverage == 0
            return None, nodes
        return (coverage, mismatches), nodes

    def nodes_to_eq_class(self, nodes):
        """sorted intersection of the colour lists of all visited nodes (order-independent, may be empty)"""
        if not nodes:
            return []
        order = sorted(range(len(nodes)), key=lambda i: len(self.classes[self.colour[nodes[i]]]))   # stable, by class length
        out = list(self.classes[self.colour[nodes[order[0]]]])
        for i in order[1:]:
            other = set(self.classes[self.colour[nodes[i]]])
            out = [v for v in out if v in other]
        return out

    def map_read(self, read, allowed=2):
        """-> None | (class ids, coverage, mismatches, nodes)"""
        r, nodes = self.map_read_to_nodes(read, allowed)
        if r is None:
            return None
        return self.nodes_to_eq_class(nodes), r[0], r[1], nodes
