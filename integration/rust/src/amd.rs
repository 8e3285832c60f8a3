//! Exporter + drop-in `process_reads` for the reference crate. Add `mod amd_ffi; mod amd;` to src/lib.rs.
//! The exporter only reads `pub` fields of `Pseudoaligner<K>` (src/pseudoaligner.rs:27-33); `dbg_index` (the boomphf MPHF,
//! :30) is not exported: every hit is verified against the node sequence (:99-107), which makes it an exact dictionary that
//! the library rebuilds. Not compiled in the image of this repository (no rustc): kept in step with
//! include/pseudoaligner_amd.h by integration/c/abi_check.c, which exercises the same calls from C.
use std::collections::HashMap;
use std::ffi::{CStr, CString};
use std::fmt::Debug;
use std::fs::File;
use std::io::{self, Write};
use std::path::Path;
use std::sync::{Arc, Mutex, OnceLock};

use anyhow::{anyhow, Error};                 // the crate's error type (src/pseudoaligner.rs:15)
use bio::io::fastq;
use debruijn::dna_string::DnaString;
use debruijn::{Kmer, Mer, Vmer};
use log::info;

use crate::amd_ffi::*;
use crate::config::DEFAULT_ALLOWED_MISMATCHES;
use crate::pseudoaligner::Pseudoaligner;

fn check(rc: i32) -> Result<i32, Error> {
    if rc >= 0 { return Ok(rc); }
    let msg = unsafe { CStr::from_ptr(pa_last_error()) }.to_string_lossy().into_owned();
    Err(anyhow!("pseudoaligner_amd error {}: {}", rc, msg))
}

/// The arrays of the flat index (include/pseudoaligner_amd.h, pa_flat_index), owned on the Rust side for the duration of a call.
pub struct FlatArrays {
    pub k: u32, pub num_transcripts: u32,
    pub seq: Vec<u64>, pub start: Vec<u64>, pub len: Vec<u32>, pub exts: Vec<u8>, pub colour: Vec<u32>,
    pub ec_off: Vec<u64>, pub ec_ids: Vec<u32>,
}

impl FlatArrays {
    pub fn from_pseudoaligner<K: Kmer>(al: &Pseudoaligner<K>) -> FlatArrays {
        let (mut seq, mut start, mut len, mut exts, mut colour) = (vec![0u64; 1], vec![0u64], vec![], vec![], vec![]);
        let mut pos = 0u64;
        for node in al.dbg.iter_nodes() {                       // debruijn::graph::DebruijnGraph
            let s = node.sequence();
            for i in 0..s.len() {                                // A0 C1 G2 T3, LSB-first 2-bit packing
                if pos % 32 == 0 && pos > 0 { seq.push(0); }
                *seq.last_mut().unwrap() |= (s.get(i) as u64) << (2 * (pos % 32));
                pos += 1;
            }
            start.push(pos); len.push(s.len() as u32); exts.push(node.exts().val); colour.push(*node.data());
        }
        seq.push(0); seq.push(0);                               // pad words: 32-base windows may read past the last base
        let (mut ec_off, mut ec_ids) = (vec![0u64], vec![]);
        for c in &al.eq_classes { ec_ids.extend_from_slice(c); ec_off.push(ec_ids.len() as u64); }
        FlatArrays { k: K::k() as u32, num_transcripts: al.tx_names.len() as u32, seq, start, len, exts, colour, ec_off, ec_ids }
    }

    pub fn view(&self) -> PaFlatIndex {
        PaFlatIndex { k: self.k, num_nodes: self.len.len() as u32, num_classes: (self.ec_off.len() - 1) as u32,
            num_transcripts: self.num_transcripts, seq_bases: *self.start.last().unwrap(), node_seq: self.seq.as_ptr(),
            node_start: self.start.as_ptr(), node_len: self.len.as_ptr(), node_exts: self.exts.as_ptr(),
            node_colour: self.colour.as_ptr(), ec_offset: self.ec_off.as_ptr(), ec_ids: self.ec_ids.as_ptr(),
            node_redge: std::ptr::null(), node_ledge: std::ptr::null() }
    }

    /// Write the index in the library's own container, e.g. to diff it against the index the library builds from the same
    /// FASTA: `pa_host_index_compare` (numbering and unitig break points are free; k-mer -> id-list map must agree).
    pub fn save(&self, path: &str) -> Result<(), Error> {
        let mut h = std::ptr::null_mut();
        check(unsafe { pa_host_index_from_flat(&self.view(), &mut h) })?;
        let p = CString::new(path)?;
        let rc = unsafe { pa_host_index_save(h, p.as_ptr()) };
        unsafe { pa_host_index_destroy(h) };
        check(rc).map(|_| ())
    }

    /// 0 = equivalent to the index the library builds from `fasta` at the same k, 1 = different, 2 = undecided
    pub fn compare_with_fasta(&self, fasta: &str) -> Result<(i32, String), Error> {
        let (mut mine, mut theirs) = (std::ptr::null_mut(), std::ptr::null_mut());
        check(unsafe { pa_host_index_from_flat(&self.view(), &mut mine) })?;
        let p = CString::new(fasta)?;
        check(unsafe { pa_host_index_build_fasta(p.as_ptr(), self.k, 0, &mut theirs) })?;
        let mut report = vec![0u8; 512];
        let rc = unsafe { pa_host_index_compare(mine, theirs, 1 << 28, report.as_mut_ptr() as *mut _, report.len()) };
        unsafe { pa_host_index_destroy(mine); pa_host_index_destroy(theirs); }
        let text = CStr::from_bytes_until_nul(&report).map(|c| c.to_string_lossy().into_owned()).unwrap_or_default();
        check(rc).map(|rc| (rc, text))
    }
}

pub struct AmdIndex { raw: *mut PaIndex, max_class_len: usize }
unsafe impl Send for AmdIndex {}
unsafe impl Sync for AmdIndex {}

impl AmdIndex {
    pub fn from_pseudoaligner<K: Kmer>(al: &Pseudoaligner<K>, device: i32) -> Result<AmdIndex, Error> {
        let flat = FlatArrays::from_pseudoaligner(al);
        let mut raw = std::ptr::null_mut();
        check(unsafe { pa_index_create(&flat.view(), device, &mut raw) })?;   // arrays are only borrowed during the call
        let mut stats = PaIndexStats::default();
        if let Err(e) = check(unsafe { pa_index_get_stats(raw, &mut stats) }) { unsafe { pa_index_destroy(raw) }; return Err(e); }
        Ok(AmdIndex { raw, max_class_len: stats.max_class_len as usize })
    }

    /// map_read_with_mismatch (src/pseudoaligner.rs:361) on the read as the caller holds it: the DnaString's bases go over as
    /// 2-bit words (LSB-first, packed here through `Mer::get`, so nothing depends on the crate's storage order)
    pub fn map_read_with_mismatch(&self, read_seq: &DnaString, allowed_mismatches: usize) -> Result<Option<(Vec<u32>, usize, usize)>, Error> {
        let len = read_seq.len();
        let mut words = vec![0u64; (len + 31) / 32 + 1];
        for i in 0..len { words[i / 32] |= (read_seq.get(i) as u64) << (2 * (i % 32)); }
        // a read's class is an intersection of index classes: never longer than the longest of them
        let mut class = vec![0u32; self.max_class_len.max(1)];
        let (mut n, mut cov, mut mm) = (0u32, 0u32, 0u32);
        let rc = check(unsafe { pa_map_read_packed(self.raw, words.as_ptr(), len as u32, PA_PACKED_LSB_FIRST, allowed_mismatches as u32,
                                                   class.as_mut_ptr(), class.len() as u32, &mut n, &mut cov, &mut mm) })?;
        if rc == 0 { return Ok(None); }
        class.truncate(n as usize);
        Ok(Some((class, cov as usize, mm as usize)))
    }
}
impl Drop for AmdIndex { fn drop(&mut self) { unsafe { pa_index_destroy(self.raw) } } }

/// One replica of the index per GPU (`pa_index_create_multi`): what `process_reads_path` deals its windows of text to.
pub struct AmdIndexSet { raw: Vec<*mut PaIndex> }
unsafe impl Send for AmdIndexSet {}
unsafe impl Sync for AmdIndexSet {}
impl AmdIndexSet {
    pub fn from_pseudoaligner<K: Kmer>(al: &Pseudoaligner<K>, devices: &[i32]) -> Result<AmdIndexSet, Error> {
        let flat = FlatArrays::from_pseudoaligner(al);
        let mut raw = vec![std::ptr::null_mut(); devices.len()];
        check(unsafe { pa_index_create_multi(&flat.view(), devices.as_ptr(), devices.len() as i32, raw.as_mut_ptr()) })?;   // all or nothing
        Ok(AmdIndexSet { raw })
    }
}
impl Drop for AmdIndexSet { fn drop(&mut self) { for r in &self.raw { unsafe { pa_index_destroy(*r) } } } }

/// the GPUs to use: `PSEUDOALIGNER_AMD_DEVICES=0,1,2,3` (default: every GPU the process sees)
fn devices_of_env() -> Vec<i32> {
    if let Ok(v) = std::env::var("PSEUDOALIGNER_AMD_DEVICES") {
        let d: Vec<i32> = v.split(',').filter_map(|x| x.trim().parse().ok()).collect();
        if !d.is_empty() { return d; }
    }
    (0..unsafe { pa_device_count() }.max(1)).collect()
}

/// `process_reads` for a caller that has the PATH of the FASTQ file — the CLI does (`fastq::Reader::from_file(args.arg_reads_fastq)`,
/// src/bin/pseudoaligner.rs:139): the file never passes through `bio`'s reader. The library reads windows of it into pinned memory,
/// the GPUs find the records and map them (pa_process_reads_multi: one lane per GPU, tuples on stdout in input order), 150 M reads/s
/// per GPU from text in the page cache — against a few M records/s for ANY loop over `fastq::Reader::records()` on one thread, which
/// is what bounds the reader-fed `process_reads` below whatever runs behind it.
pub fn process_reads_path<K: Kmer + Sync + Send, P: AsRef<Path> + Debug, Q: AsRef<Path>>(
    reads_fastq: Q,
    index: &Pseudoaligner<K>,
    outdir: P,
    num_threads: usize,
) -> Result<(), Error> {
    info!("Done Reading index");
    info!("Starting Multi-threaded Mapping");
    info!("Output directory: {:?}", outdir);
    let set = AmdIndexSet::from_pseudoaligner(index, &devices_of_env())?;
    io::stdout().flush()?;                                            // (the library writes to the process's stdout through C stdio)
    let path = CString::new(reads_fastq.as_ref().to_string_lossy().into_owned())?;
    let dash = CString::new("-")?;
    let (mut n_reads, mut n_flagged) = (0u64, 0u64);
    // the progress line of :497-503 (exactly every 10^6-th read, f32 Display) and the `eprintln!()` behind it come from the library
    check(unsafe { pa_process_reads_multi(set.raw.as_ptr(), set.raw.len() as i32, path.as_ptr(), dash.as_ptr(), num_threads as i32,
                                          &mut n_reads, &mut n_flagged) })?;
    info!("Done Mapping Reads");
    info!("Mapped {} reads, {} flagged", n_reads, n_flagged);
    Ok(())
}

// ---------------------------------------------------------------------------------------------------------------------------
// Drop-in entry points with the reference's EXACT signatures. The GPU copy of an index is made on first use and cached by the
// CONTENT of the `Pseudoaligner`: k, node / class / transcript counts and a fingerprint of EVERY base of every node, every node's
// colour and extension byte and EVERY id of every class — two indexes of the same shape with different sequence (SNP-personalised
// transcriptomes) or classes such as [1,5,9] / [1,6,9] never share a GPU copy; an index that was dropped and another one
// allocated in its place never meets a stale copy; a moved or cloned one finds its copy again. The cache holds at most
// CACHE_MAX GPU copies (least recently used goes: `pa_index_destroy` frees its HBM when the last `Arc` is dropped).
// ---------------------------------------------------------------------------------------------------------------------------
const CACHE_MAX: usize = 2;
type CacheKey = (usize, usize, usize, usize, u64);
fn fingerprint<K: Kmer>(index: &Pseudoaligner<K>) -> u64 {
    let mut h = 0xcbf2_9ce4_8422_2325u64;                                  // FNV-1a over 64-bit words
    let mut mix = |v: u64| { h ^= v; h = h.wrapping_mul(0x0000_0100_0000_01b3); };
    for c in &index.eq_classes {                                           // every class: its length and every id
        mix(c.len() as u64);
        for pair in c.chunks(2) { mix(((pair[0] as u64) << 32) | *pair.get(1).unwrap_or(&u32::MAX) as u64); }
    }
    for node in index.dbg.iter_nodes() {                                   // every node: length, colour, extensions, every base
        let s = node.sequence();
        mix(((s.len() as u64) << 32) | *node.data() as u64);
        mix(node.exts().val as u64);
        let (mut w, mut n) = (0u64, 0u32);
        for i in 0..s.len() {
            w |= (s.get(i) as u64) << (2 * n); n += 1;
            if n == 32 { mix(w); w = 0; n = 0; }
        }
        if n > 0 { mix(w); }
    }
    h
}
struct Cache { map: HashMap<CacheKey, Arc<AmdIndex>>, order: Vec<CacheKey> }   // order: least recently used first
fn cache() -> &'static Mutex<Cache> {
    static CACHE: OnceLock<Mutex<Cache>> = OnceLock::new();
    CACHE.get_or_init(|| Mutex::new(Cache { map: HashMap::new(), order: Vec::new() }))
}
fn key_of<K: Kmer>(index: &Pseudoaligner<K>) -> CacheKey {
    (index.dbg.len(), index.eq_classes.len(), index.tx_names.len(), K::k(), fingerprint(index))
}
// (the fingerprint walks every base once per call — tenths of a second at 200 k transcripts; `process_reads` and `map_reads` look
// their index up once per call, a host that calls `map_read` per read keeps the `Arc<AmdIndex>` of `gpu_index` instead)
fn device_of_env() -> i32 { std::env::var("PSEUDOALIGNER_AMD_DEVICE").ok().and_then(|v| v.parse().ok()).unwrap_or(0) }

/// the GPU copy of `index` (built once: about half a second for a 200 k-transcript index)
pub fn gpu_index<K: Kmer>(index: &Pseudoaligner<K>) -> Result<Arc<AmdIndex>, Error> {
    let key = key_of(index);
    {
        let mut c = cache().lock().unwrap();
        if let Some(hit) = c.map.get(&key).cloned() {
            c.order.retain(|k| *k != key); c.order.push(key);
            return Ok(hit);
        }
    }
    let made = Arc::new(AmdIndex::from_pseudoaligner(index, device_of_env())?);
    let mut c = cache().lock().unwrap();
    let got = c.map.entry(key).or_insert(made).clone();
    c.order.retain(|k| *k != key); c.order.push(key);
    while c.order.len() > CACHE_MAX { let old = c.order.remove(0); c.map.remove(&old); }   // the GPU copy goes with its last Arc
    Ok(got)
}
/// call before dropping a `Pseudoaligner` whose GPU copy should go too
pub fn forget<K: Kmer>(index: &Pseudoaligner<K>) {
    let key = key_of(index);
    let mut c = cache().lock().unwrap();
    c.map.remove(&key); c.order.retain(|k| *k != key);
}

/// MANY reads in one launch — what a caller that loops over `map_read` should call instead (the reference's own test maps every
/// transcript one call at a time, src/build_index.rs:309). One `map_read` call is one H2D + one kernel launch + one D2H: tens of
/// microseconds per read, slower than the CPU path it replaces; a batch amortises them over all its reads.
/// Result i is what `map_read(&reads[i])` returns (src/pseudoaligner.rs:381-384).
pub fn map_reads<K: Kmer + Sync + Send>(index: &Pseudoaligner<K>, reads: &[DnaString]) -> Result<Vec<Option<(Vec<u32>, usize)>>, Error> {
    let gpu = gpu_index(index)?;
    let (mut words, mut word_off, mut lens) = (Vec::new(), vec![0u64], Vec::with_capacity(reads.len()));
    for r in reads {                                                       // every read starts on a word boundary, LSB-first words
        let base = words.len();
        words.resize(base + (r.len() + 31) / 32, 0u64);
        for i in 0..r.len() { words[base + i / 32] |= (r.get(i) as u64) << (2 * (i % 32)); }
        word_off.push(words.len() as u64);
        lens.push(r.len() as u32);
    }
    words.push(0);
    let mut results = vec![PaReadResult::default(); reads.len()];
    let mut class_off = vec![0u64; reads.len() + 1];
    let mut class_ids: *const u32 = std::ptr::null();
    check(unsafe { pa_map_batch_packed(gpu.raw, words.as_ptr(), word_off.as_ptr(), lens.as_ptr(), reads.len() as u64, PA_PACKED_LSB_FIRST,
                                       DEFAULT_ALLOWED_MISMATCHES as u32, results.as_mut_ptr(), class_off.as_mut_ptr(), &mut class_ids) })?;
    Ok((0..reads.len()).map(|i| {
        if results[i].mismatches & PA_MAPPED_BIT == 0 { return None; }
        let ids = unsafe { std::slice::from_raw_parts(class_ids.add(class_off[i] as usize), (class_off[i + 1] - class_off[i]) as usize) };
        Some((ids.to_vec(), results[i].coverage as usize))                 // (library-owned ids: copied before the next call on this index)
    }).collect())
}

/// `Pseudoaligner::map_read` (src/pseudoaligner.rs:381), same arguments and result: replace its body by
/// `crate::amd::map_read(self, read_seq)`. Panics where the reference panics (it has no error path, :307,446).
pub fn map_read<K: Kmer + Sync + Send>(index: &Pseudoaligner<K>, read_seq: &DnaString) -> Option<(Vec<u32>, usize)> {
    let gpu = gpu_index(index).expect("pseudoaligner_amd: GPU index");
    gpu.map_read_with_mismatch(read_seq, DEFAULT_ALLOWED_MISMATCHES).expect("pseudoaligner_amd: map_read")
        .map(|(eq_class, read_coverage, _mismatches)| (eq_class, read_coverage))
}

/// `process_reads` (src/pseudoaligner.rs:420-425) with the reference's signature. The worker pool, the reader mutex
/// (utils.rs:152-157) and the sync_channel (:430) become: read a chunk of records -> pa_records_push (packs and launches full
/// batches on the GPU while this thread goes on reading) -> pa_records_pull (the Debug tuples of :490, in input order) -> stdout.
/// BOUND BY ITS READER: the signature hands over an open `fastq::Reader`, so ONE thread iterates `records()` (a few M records/s:
/// a line scan, two allocations and a UTF-8 check per record) — one GPU maps 150 M reads/s from text and 13 G reads/s from HBM
/// behind it. A caller that knows the file's path (the CLI does) calls `process_reads_path` above: two lines of
/// src/bin/pseudoaligner.rs.
pub fn process_reads<K: Kmer + Sync + Send, P: AsRef<Path> + Debug>(
    reader: fastq::Reader<io::BufReader<File>>,
    index: &Pseudoaligner<K>,
    outdir: P,
    num_threads: usize,
) -> Result<(), Error> {
    info!("Done Reading index");
    info!("Starting Multi-threaded Mapping");
    info!("Output directory: {:?}", outdir);
    let gpu = gpu_index(index)?;
    let mut stream = std::ptr::null_mut();
    check(unsafe { pa_record_stream_create(gpu.raw, num_threads as i32, 0, &mut stream) })?;
    struct Guard(*mut PaRecordStream);
    impl Drop for Guard { fn drop(&mut self) { unsafe { pa_record_stream_destroy(self.0) } } }
    let _guard = Guard(stream);

    let stdout = io::stdout();
    let mut out = io::BufWriter::with_capacity(1 << 22, stdout.lock());
    let mut text = vec![0u8; 1 << 26];   // (big pulls are copied by the library's worker pool)
    let mut millions_reported = 0u64;
    let mut drain = |out: &mut dyn Write| -> Result<(), Error> {
        loop {
            let mut n = 0usize;
            let rc = unsafe { pa_records_pull(stream, text.as_mut_ptr() as *mut _, text.len(), &mut n) };
            if rc == PA_ERR_BUFFER_TOO_SMALL && text.len() < (1usize << 31) {
                // one tuple longer than the buffer (a class of hundreds of thousands of ids): n = the bytes it needs; grow and ask again
                text.resize(n.max(text.len() * 2), 0);
                continue;
            }
            check(rc)?;
            if n == 0 { break; }
            out.write_all(&text[..n])?;
        }
        // the reference's progress line (src/pseudoaligner.rs:497-503), whenever another million reads have been printed
        // (batch granular: the counts are those of the batches rendered so far)
        let (mut n_reads, mut n_flagged) = (0u64, 0u64);
        check(unsafe { pa_record_stream_stats(stream, &mut n_reads, &mut n_flagged) })?;
        if n_reads / 1_000_000 > millions_reported {
            millions_reported = n_reads / 1_000_000;
            eprint!("\rDone Mapping {} reads w/ Rate: {}", n_reads, n_flagged as f32 * 100.0 / n_reads as f32);
            io::stderr().flush().expect("Could not flush stdout");
        }
        Ok(())
    };
    let mut records = reader.records();
    loop {
        let (mut ids, mut id_off, mut seqs, mut seq_off) = (Vec::new(), vec![0u64], Vec::new(), vec![0u64]);
        for r in records.by_ref().take(1 << 16) {
            let r = r?;
            ids.extend_from_slice(r.id().as_bytes()); id_off.push(ids.len() as u64);       // record.id() (:456)
            seqs.extend_from_slice(r.seq()); seq_off.push(seqs.len() as u64);             // record.seq() (:449)
        }
        let n = id_off.len() - 1;
        if n == 0 { break; }
        check(unsafe { pa_records_push(stream, ids.as_ptr(), id_off.as_ptr(), seqs.as_ptr(), seq_off.as_ptr(), n as u64) })?;
        drain(&mut out)?;
    }
    check(unsafe { pa_records_flush(stream) })?;
    drain(&mut out)?;
    out.flush()?;
    let (mut n_reads, mut n_flagged) = (0u64, 0u64);
    check(unsafe { pa_record_stream_stats(stream, &mut n_reads, &mut n_flagged) })?;
    eprintln!();
    info!("Done Mapping Reads");
    info!("Mapped {} reads, {} flagged", n_reads, n_flagged);
    Ok(())
}
