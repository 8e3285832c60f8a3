//! Exporter + drop-in `process_reads` for the reference crate. Add `mod amd_ffi; mod amd;` to src/lib.rs.
//! The exporter only reads `pub` fields of `Pseudoaligner<K>` (src/pseudoaligner.rs:27-33); `dbg_index` (the boomphf MPHF,
//! :30) is not exported: every hit is verified against the node sequence (:99-107), which makes it an exact dictionary that
//! the library rebuilds. Not compiled in the image of this repository (no rustc): kept in step with
//! include/pseudoaligner_amd.h by integration/c/abi_check.c, which exercises the same calls from C.
use std::ffi::{CStr, CString};
use std::fmt::Debug;
use std::fs::File;
use std::io;
use std::path::Path;

use bio::io::fastq;
use debruijn::{Kmer, Mer, Vmer};
use failure::{format_err, Error};
use log::info;

use crate::amd_ffi::*;
use crate::config::READ_COVERAGE_THRESHOLD;
use crate::pseudoaligner::Pseudoaligner;

fn check(rc: i32) -> Result<i32, Error> {
    if rc >= 0 { return Ok(rc); }
    let msg = unsafe { CStr::from_ptr(pa_last_error()) }.to_string_lossy().into_owned();
    Err(format_err!("pseudoaligner_amd error {}: {}", rc, msg))
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

pub struct AmdIndex { raw: *mut PaIndex }
unsafe impl Send for AmdIndex {}
unsafe impl Sync for AmdIndex {}

impl AmdIndex {
    pub fn from_pseudoaligner<K: Kmer>(al: &Pseudoaligner<K>, device: i32) -> Result<AmdIndex, Error> {
        let flat = FlatArrays::from_pseudoaligner(al);
        let mut raw = std::ptr::null_mut();
        check(unsafe { pa_index_create(&flat.view(), device, &mut raw) })?;   // arrays are only borrowed during the call
        Ok(AmdIndex { raw })
    }

    /// map_read (src/pseudoaligner.rs:381): Some((eq_class, coverage)) | None
    pub fn map_read(&self, read: &[u8]) -> Result<Option<(Vec<u32>, usize)>, Error> {
        let mut class = vec![0u32; 1 << 16];
        let (mut n, mut cov) = (0u32, 0u32);
        let rc = check(unsafe { pa_map_read(self.raw, read.as_ptr(), read.len() as u32, class.as_mut_ptr(),
                                            class.len() as u32, &mut n, &mut cov) })?;
        if rc == 0 { return Ok(None); }
        class.truncate(n as usize);
        Ok(Some((class, cov as usize)))
    }
}
impl Drop for AmdIndex { fn drop(&mut self) { unsafe { pa_index_destroy(self.raw) } } }

/// process_reads (src/pseudoaligner.rs:420-425) with the same signature; the worker pool, the reader mutex
/// (utils.rs:152-157) and the sync_channel (:430) become: fill a batch -> pa_map_batch -> print the tuples.
pub fn process_reads<P: AsRef<Path> + Debug>(reader: fastq::Reader<io::BufReader<File>>, index: &AmdIndex, outdir: P,
                                             _num_threads: usize) -> Result<(), Error> {
    info!("Output directory: {:?}", outdir);
    let mut records = reader.records();
    loop {
        let (mut ascii, mut offsets, mut ids) = (Vec::new(), vec![0u64], Vec::new());
        for r in records.by_ref().take(1 << 20) {
            let r = r?; ascii.extend_from_slice(r.seq()); offsets.push(ascii.len() as u64); ids.push(r.id().to_owned());
        }
        if ids.is_empty() { break; }
        let n = ids.len();
        let mut res = vec![PaReadResult::default(); n];
        let mut coff = vec![0u64; n + 1];
        let mut cls: *const u32 = std::ptr::null();
        check(unsafe { pa_map_batch(index.raw, ascii.as_ptr(), offsets.as_ptr(), n as u64, 2, res.as_mut_ptr(), coff.as_mut_ptr(), &mut cls) })?;
        for i in 0..n {
            let mapped = res[i].mismatches & PA_MAPPED_BIT != 0;
            let class: Vec<u32> = unsafe { std::slice::from_raw_parts(cls.add(coff[i] as usize), res[i].class_len as usize) }.to_vec();
            let cov = if mapped { res[i].coverage as usize } else { 0 };
            let flag = mapped && cov >= READ_COVERAGE_THRESHOLD && class.is_empty();          // :455
            println!("{:?}", (flag, ids[i].clone(), class, cov));                             // :490
        }
    }
    Ok(())
}
