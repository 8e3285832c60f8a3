"""The C-ABI library loads, exports every symbol include/pseudoaligner_amd.h declares, and its host-side entry points
behave (no GPU compute here)."""
import ctypes as C
import re

import numpy as np
import pytest

import helpers

pa = helpers.pa


def declared_symbols():
    text = (helpers.ROOT / "include" / "pseudoaligner_amd.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pa_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(built):
    names = declared_symbols()
    assert len(names) > 40
    lib = C.CDLL(str(pa._ffi.library_path()))
    for n in names:
        assert hasattr(lib, n), "library does not export %s" % n
    assert set(names) == set(pa._ffi.SIGNATURES), set(names) ^ set(pa._ffi.SIGNATURES)
    assert pa.lib().pa_abi_version() == 1


def test_product_never_references_the_oracle():
    """the oracle is test infrastructure: nothing under the package may mention it"""
    pkg = helpers.ROOT / "rust-pseudoaligner_amd"
    for f in list(pkg.glob("*.py")) + list((pkg / "csrc").glob("*")):
        text = f.read_text(errors="ignore")
        assert "pa_oracle" not in text and "oracle/" not in text and "libpa_emu" not in text, f


def test_shipped_library_reads_no_tuning_knob(built):
    """the A/B knobs of DESIGN.md §8 exist only in -DPA_DEBUG_KNOBS builds (tools/build_variant.sh): the shipped library does not
    even hold their names, so an exported variable cannot make it skip result stores or counts"""
    blob = pa._ffi.library_path().read_bytes()
    for knob in (b"PA_MAP_ABLATE", b"PA_MAP_STATS", b"PA_POOL_SLOTS", b"PA_MAP_BLOCKS_PER_CU", b"PA_MAP_GREAD", b"PA_DICT_LOAD", b"PA_SIM_TX_LIMIT"):
        assert knob not in blob, knob


def test_no_gpu_means_loud_failure(built, small_index):
    if pa.lib().pa_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pa.PaError) as e:
        pa.Pseudoaligner(small_index(20))
    assert e.value.code == pa._ffi.PA_ERR_NO_DEVICE and "no CPU fallback" in str(e.value)
    p = C.c_void_p()
    assert pa.lib().pa_device_malloc(0, 1024, C.byref(p)) == pa._ffi.PA_ERR_NO_DEVICE


def test_encode_reads_host_layout(built):
    reads = ["ACGT", "", "TTTTGGGGCCCCAAAATTTTGGGGCCCCAAAATTTTG", "acgtn"]   # 4, 0, 37 (two words), lower case + N
    tiles, lens, wpr = pa.encode_reads_host(reads)
    assert wpr == 2 and lens.tolist() == [4, 0, 37, 5] and len(tiles) == 128
    t = tiles.reshape(1, 2, 64)
    assert int(t[0, 0, 0]) == 0b11100100                       # A=0 C=1 G=2 T=3, base j at bits 2j
    assert int(t[0, 0, 1]) == 0 and int(t[0, 1, 2]) == 0b10_11_11_11_11   # TTTTG: bases 32..36
    assert int(t[0, 0, 3]) == 0b00_11_10_01_00                 # n -> A
    assert pa.unpack_tiles(tiles, lens, wpr) == ["ACGT", "", reads[2], "ACGTA"]
    assert helpers.pack_read("ACGT")[0] == 0b11100100


def test_checker_packer_is_independent_and_agrees(built):
    """helpers.pack_reads_tiles (numpy, what the oracle is fed in the batch parity tests) vs the product's host encoder:
    two implementations, same tiles — on ragged, empty, lower-case, non-ACGT reads and across the 64-read tile seam"""
    _, seqs = helpers.read_fastq()
    reads = seqs[:200] + ["", "A", "acgtnNRY" * 9, "T" * 33, "G" * 64, "C" * 65, seqs[0][:31], seqs[1][:32]] + seqs[200:260]
    t1, l1, w1 = pa.encode_reads_host(reads)
    t2, l2, w2 = helpers.pack_reads_tiles(reads)
    assert w1 == w2 and np.array_equal(l1, l2) and np.array_equal(t1, t2)
    t3, _, _ = helpers.pack_reads_tiles(reads, 4)                                  # explicit words per read
    t4, _, _ = pa.encode_reads_host(reads, 4)
    assert np.array_equal(t3, t4)
    assert [int(x) for x in helpers.pack_reads_tiles(["ACGT"])[0][:1]] == [0b11100100]


def test_simulator_is_a_pure_function_of_seed_and_index(built, small_index):
    tx = pa.Txome.from_host_index(small_index(24))
    a, la = tx.simulate_host(100, 1, 1000)
    b, lb = tx.simulate_host(100, 1, 300, first_read=700)
    ra, rb = pa.unpack_tiles(a, la, 4), pa.unpack_tiles(b, lb, 4)
    assert ra[700:] == rb                                      # sharding by read index is rank-count independent
    c, _ = tx.simulate_host(100, 2, 1000)
    assert pa.unpack_tiles(c, la, 4) != ra
    _, seqs = helpers.read_fasta()
    joined = set()
    for r in ra[:50]:
        assert any(r in s for s in seqs)                       # error-free reads are substrings of transcripts
    e, _ = tx.simulate_host(100, 1, 1000, sub_rate_ppm=10000)
    diffs = sum(x != y for r1, r2 in zip(ra, pa.unpack_tiles(e, la, 4)) for x, y in zip(r1, r2))
    assert 700 < diffs < 1300                                  # ~1 % of 100 000 bases
    with pytest.raises(pa.PaError):
        tx.simulate_host(100000, 1, 10)                        # no transcript that long


def test_synthetic_transcriptome_shape(built):
    tx = pa.Txome.synthesize(2000, 7000, 7)
    packed, tx_start = tx.arrays()
    lens = np.diff(tx_start.astype(np.int64))
    assert 6000 < len(lens) < 8000 and 1200 < lens.mean() < 1800 and lens.min() >= 60
    tx2 = pa.Txome.synthesize(2000, 7000, 7)
    assert np.array_equal(tx2.arrays()[0], packed)


def test_counts_reference_definition(small_index):
    host = small_index(24)
    _, seqs = helpers.read_fastq()
    res, coff, cids, _ = helpers.Oracle(host).map_reads(seqs[:2000], 2, 2)
    counts = helpers.counts_reference(res, coff, cids, host)
    nc = host.arrays()["num_classes"]
    assert counts.sum() == 2000 and counts[nc + 2] == int((res["mapped"] == 0).sum())


def _header_functions():
    import re
    h = (helpers.ROOT / "include" / "pseudoaligner_amd.h").read_text()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(m.group(1) for m in re.finditer(r"\b(pa_\w+)\s*\([^;{]*\)\s*;", h)))


def test_c_client_calls_every_entry_point(built, tmp_path):
    """integration/c/abi_check.c: compiled with gcc -Wall -Werror against the header (a prototype that drifts from what a
    foreign binding assumes does not build) and run: host half everywhere, device half on a GPU box"""
    import re
    import subprocess
    exe = helpers._build.build_abi_check()
    src = (helpers.ROOT / "integration" / "c" / "abi_check.c").read_text()
    missing = [f for f in _header_functions() if f + "(" not in src]
    assert not missing, "abi_check.c does not call: %s" % missing
    rust = (helpers.ROOT / "integration" / "rust" / "src" / "amd_ffi.rs").read_text()
    unknown = [f for f in re.findall(r"pub fn (pa_\w+)\(", rust) if f not in _header_functions()]
    assert not unknown, "amd_ffi.rs binds symbols the header does not declare: %s" % unknown
    out = subprocess.run([str(exe), str(helpers.FASTA), str(helpers.FASTQ), str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 failures" in out.stdout
    # the #[repr(C)] structs of the Rust binding against what the C compiler laid out (sizeof / offsetof printed by abi_check.c)
    import abi_sigs
    got = {}
    for line in out.stdout.splitlines():
        f = line.split()
        if len(f) >= 3 and f[0] == "layout":
            got[f[1] if f[2] != "sizeof" else f[1] + ".sizeof"] = int(f[-1])
    rs = abi_sigs.rust_structs(rust)
    assert set(rs) == {"pa_flat_index", "pa_read_result", "pa_index_stats"}
    for name, fields in rs.items():
        offs, size = abi_sigs.layout(fields)
        assert got[name + ".sizeof"] == size, (name, got[name + ".sizeof"], size)
        for fld, off in offs.items():
            assert got["%s.%s" % (name, fld)] == off, (name, fld, got.get("%s.%s" % (name, fld)), off)
    if pa.lib().pa_device_count() < 1:
        assert "no device" in out.stdout
    else:
        assert "device halves ok" in out.stdout


def test_rust_binding_matches_the_header_signature_by_signature():
    """integration/rust/src/amd_ffi.rs cannot be compiled here (no rustc): every `pub fn pa_*` prototype — arity, each parameter's
    C type through a fixed Rust -> C map, the return type — every #[repr(C)] struct (field order, names, types) and every constant is
    compared with include/pseudoaligner_amd.h; a drifted u32 / u64, a swapped argument or a missing `const` fails here"""
    import abi_sigs
    header = (helpers.ROOT / "include" / "pseudoaligner_amd.h").read_text()
    rust = (helpers.ROOT / "integration" / "rust" / "src" / "amd_ffi.rs").read_text()
    hp, rp = abi_sigs.header_prototypes(header), abi_sigs.rust_prototypes(rust)
    assert len(rp) > 45 and len(hp) >= len(rp)
    for name, sig in rp.items():
        assert name in hp, "amd_ffi.rs binds %s, which the header does not declare" % name
        assert hp[name] == sig, "%s: Rust says %s, the header %s" % (name, sig, hp[name])
    hs, rs = abi_sigs.header_structs(header), abi_sigs.rust_structs(rust)
    for name, fields in rs.items():
        assert hs[name] == fields, "struct %s: Rust %s, header %s" % (name, fields, hs[name])
    hc, rc = abi_sigs.header_consts(header), abi_sigs.rust_consts(rust)
    for name, value in rc.items():
        assert hc.get(name) == value, "constant %s: Rust %r, header %r" % (name, value, hc.get(name))
    # the comparison is not vacuous: a drifted binding is caught
    for bad_from, bad_to in (("n_reads: u64, allowed_mismatches: u32", "n_reads: u32, allowed_mismatches: u32"),
                             ("pub coverage: u32, pub mismatches: u32", "pub mismatches: u32, pub coverage: u32"),
                             ("d_tiles: *const u64, d_lens: *const u32, n_reads: u64,", "d_lens: *const u32, d_tiles: *const u64, n_reads: u64,")):
        assert bad_from in rust
        drift = rust.replace(bad_from, bad_to, 1)
        same = all(hp[n] == s for n, s in abi_sigs.rust_prototypes(drift).items()) and all(hs[n] == f for n, f in abi_sigs.rust_structs(drift).items())
        assert not same, "a drifted binding went unnoticed: %s" % bad_to
