//! cbl-gpu — `CBL<K, T, PREFIX_BITS>` with the reference's method names, argument meaning and panics for the bulk-insert path
//! (`/root/reference/src/cbl.rs:71-79,127-177,219-228,293-339,358-373,433-449`), every body a call into `libcblx`
//! (`include/cblx.h`). The index lives in the HBM of one MI355X; nothing here computes on the host.
//!
//! NOT compiled in the image this repository was built in (no Rust toolchain there): shipped as source for the maintainer
//! who wires the MI355X path into the reference. Differences from the reference type that a caller can see:
//!   * `T` is the reference's `T` (`build.rs:34-41`: the narrowest integer that holds 2K + POS_BITS bits — `u64` up to K = 29,
//!     `u128` from K = 31 on) and is checked by the reference's own assert (`src/cbl.rs:87-91`); k-mers are packed in it, the
//!     word layout behind the ABI follows from K alone.
//!   * `insert_seq` may only enqueue; every observer (`count`, `contains*`, `save_to_file`, `|=`, `iter`) flushes first, so
//!     results are those of the reference after the same calls.
//!   * out of scope, as in SURVEY.md §8f: `remove`, `remove_seq`, `&=`, `-=`, `^=`, `merge` / `intersect` on vectors of sets,
//!     and the Trie node statistics (`buckets_nodes`, `buckets_node_count`).
//!
//! Drop-in use in `examples/cbl.rs`: replace `use cbl::CBL;` by `use cbl_gpu::CBL;` — nothing else. `write_index(&cbl, path)` /
//! `read_index(path)` (`examples/cbl.rs:117-142`) go through this type's `Serialize` / `Deserialize`, which speak the reference's
//! serde data model (`serde_model.rs`), so bincode `DefaultOptions` + varint writes and reads the reference's bytes.
//! `cbl.save_to_file(path)` / `CBL::load_from_file(path)` (`src/cbl.rs:127-160`) do the same inside the library, on the device
//! (9.4 GB in 0.2 s instead of a host walk). `Build`, `Insert`, `Merge`, `Count`, `Query` run unchanged.
use std::collections::BTreeMap;
use std::ffi::{CStr, CString};
use std::marker::PhantomData;
use std::ops::BitOrAssign;
use std::os::raw::c_int;
use std::path::Path;

use cblx_sys as sys;

mod serde_model;

/// The integer types a k-mer is packed in (`IntKmer::to_int`, `src/kmer.rs:200-202`: 2K bits, first base most significant).
pub trait PackedInt: Copy {
    const BITS: u32;
    fn split(self) -> (u64, u64);
    fn join(lo: u64, hi: u64) -> Self;
}
impl PackedInt for u64 {
    const BITS: u32 = 64;
    #[inline]
    fn split(self) -> (u64, u64) {
        (self, 0)
    }
    #[inline]
    fn join(lo: u64, _hi: u64) -> Self {
        lo
    }
}
impl PackedInt for u128 {
    const BITS: u32 = 128;
    #[inline]
    fn split(self) -> (u64, u64) {
        (self as u64, (self >> 64) as u64)
    }
    #[inline]
    fn join(lo: u64, hi: u64) -> Self {
        ((hi as u128) << 64) | lo as u128
    }
}

/// A k-mer packed into an integer, as the reference's `IntKmer<K, T>` (with the feature `reference-types` the reference's own
/// type converts from / into this one).
#[derive(Debug, Clone, Copy, PartialEq, Eq, PartialOrd, Ord, Hash)]
#[repr(transparent)]
pub struct IntKmer<const K: usize, T: PackedInt>(pub T);

impl<const K: usize, T: PackedInt> IntKmer<K, T> {
    #[inline]
    pub fn from_int(s: T) -> Self {
        Self(s)
    }
    #[inline]
    pub fn to_int(self) -> T {
        self.0
    }
}

#[cfg(feature = "reference-types")]
mod reference_types {
    use super::{IntKmer, PackedInt};
    use cbl::kmer::{Base, IntKmer as RefKmer, Kmer};

    impl<const K: usize, T: PackedInt + Base> From<RefKmer<K, T>> for IntKmer<K, T> {
        fn from(k: RefKmer<K, T>) -> Self {
            IntKmer(k.to_int())
        }
    }
    impl<const K: usize, T: PackedInt + Base> From<IntKmer<K, T>> for RefKmer<K, T> {
        fn from(k: IntKmer<K, T>) -> Self {
            RefKmer::<K, T>::from_int(k.0)
        }
    }
}

/// A set of k-mers resident in the HBM of one MI355X (`CBL<K, T, PREFIX_BITS>`, `src/cbl.rs:40-54`).
pub struct CBL<const K: usize, T: PackedInt, const PREFIX_BITS: usize = 24> {
    ctx: *mut sys::cblx_ctx,
    _t: PhantomData<T>,
}

// The reference type is not Send / Sync either (UniquePtr members, `src/cbl.rs:50-53`); one ctx = one owner thread.

impl<const K: usize, T: PackedInt, const PREFIX_BITS: usize> CBL<K, T, PREFIX_BITS> {
    /// Number of bits of a k-mer (`src/cbl.rs:16-18`).
    pub const KMER_BITS: usize = 2 * K;
    /// Number of bits of a position inside a k-mer (`src/cbl.rs:20-22`: ilog2 of 2K rounded up to a power of two).
    pub const POS_BITS: usize = (2 * K).next_power_of_two().trailing_zeros() as usize;

    fn last_error(&self) -> String {
        unsafe { CStr::from_ptr(sys::cblx_last_error(self.ctx)) }.to_string_lossy().into_owned()
    }

    /// A non-zero status becomes the panic the reference would raise, with the library's text
    /// (`CBLX_ESHORT` carries "Sequence size (n) is smaller than K (k)", `src/cbl.rs:329-334`).
    #[inline]
    fn check(&self, rc: c_int) {
        if rc != sys::CBLX_OK {
            panic!("{}", self.last_error());
        }
    }

    fn with(canonical: bool) -> Self {
        // the reference's assert, same condition and text (`src/cbl.rs:87-91`): the word — 2K bits + position — must fit `T`,
        // so `CBL::<31, u64>` (68 bits) panics here exactly as it does there
        assert!(
            Self::KMER_BITS + Self::POS_BITS <= T::BITS as usize,
            "Cannot fit a {K}-mer and its length in a {}-bit integer",
            T::BITS
        );
        let p = sys::cblx_params {
            k: K as u32,
            prefix_bits: PREFIX_BITS as u32,
            canonical: canonical as u32,
            device: -1,
            flags: 0,
            reserved: 0,
        };
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { sys::cblx_create(&p, &mut ctx) };
        if rc != sys::CBLX_OK {
            // PREFIX_BITS / SUFFIX_BITS asserts of `src/wordset/mod.rs:37-41`, or no usable GPU (there is no CPU fallback)
            panic!("{}", unsafe { CStr::from_ptr(sys::cblx_last_global_error()) }.to_string_lossy());
        }
        Self { ctx, _t: PhantomData }
    }

    /// Creates an empty set (`src/cbl.rs:71-74`).
    pub fn new() -> Self {
        Self::with(false)
    }

    /// Creates an empty set of canonical k-mers (`src/cbl.rs:77-79`).
    pub fn new_canonical() -> Self {
        Self::with(true)
    }

    fn cpath<P: AsRef<Path>>(path: P) -> CString {
        CString::new(path.as_ref().to_str().expect("path is not valid UTF-8")).expect("path contains a NUL byte")
    }

    /// Saves the set to a file: bincode `DefaultOptions` + varint, the bytes of the reference (`src/cbl.rs:127-142`).
    pub fn save_to_file<P: AsRef<Path> + Copy>(&self, path: P) {
        let p = Self::cpath(path);
        let rc = unsafe { sys::cblx_save_to_file(self.ctx, p.as_ptr()) };
        if rc != sys::CBLX_OK {
            panic!("Failed to open {}: {}", path.as_ref().to_str().unwrap(), self.last_error());
        }
    }

    /// Loads a set from a file (`src/cbl.rs:145-160`; trailing bytes are rejected). K / PREFIX_BITS are not in the file:
    /// they are this type's parameters, as in the reference. The canonical flag comes from the file.
    pub fn load_from_file<P: AsRef<Path> + Copy>(path: P) -> Self {
        let s = Self::new();
        let p = Self::cpath(path);
        let rc = unsafe { sys::cblx_load_from_file(s.ctx, p.as_ptr()) };
        if rc != sys::CBLX_OK {
            panic!("Failed to load {}: {}", path.as_ref().to_str().unwrap(), s.last_error());
        }
        s
    }

    /// The exact bytes `save_to_file` writes (for callers that hold a writer instead of a path).
    pub fn to_bytes(&self) -> Vec<u8> {
        let mut n = 0u64;
        self.check(unsafe { sys::cblx_serialized_size(self.ctx, &mut n) });
        let mut buf = vec![0u8; n as usize];
        let mut written = 0u64;
        self.check(unsafe { sys::cblx_serialize(self.ctx, buf.as_mut_ptr(), n, &mut written) });
        buf.truncate(written as usize);
        buf
    }

    /// Inverse of `to_bytes` (`Deserialize`, `src/wordset/mod.rs:398-437`).
    pub fn from_bytes(data: &[u8]) -> Self {
        Self::try_from_bytes(data).unwrap_or_else(|e| panic!("{}", e))
    }

    /// `from_bytes` that reports malformed bytes instead of panicking (what `Deserialize` needs).
    pub fn try_from_bytes(data: &[u8]) -> Result<Self, String> {
        let s = Self::new();
        let rc = unsafe { sys::cblx_load(s.ctx, data.as_ptr(), data.len() as u64) };
        if rc != sys::CBLX_OK {
            return Err(s.last_error());
        }
        Ok(s)
    }

    /// Returns `true` if the set stores canonical k-mers (`src/cbl.rs:164-166`).
    #[inline]
    pub fn is_canonical(&self) -> bool {
        let mut c: c_int = 0;
        self.check(unsafe { sys::cblx_is_canonical(self.ctx, &mut c) });
        c != 0
    }

    /// Counts the k-mers of the set (`src/cbl.rs:169-171`).
    pub fn count(&self) -> usize {
        let mut n = 0u64;
        self.check(unsafe { sys::cblx_count(self.ctx, &mut n) });
        n as usize
    }

    /// Returns `true` if there are no k-mers in the set (`src/cbl.rs:175-177`).
    #[inline]
    pub fn is_empty(&self) -> bool {
        let mut e: c_int = 0;
        self.check(unsafe { sys::cblx_is_empty(self.ctx, &mut e) });
        e != 0
    }

    /// Returns `true` if the set contains the k-mer (`src/cbl.rs:219-221`).
    #[inline]
    pub fn contains(&self, kmer: IntKmer<K, T>) -> bool {
        let (lo, hi) = kmer.to_int().split();
        let mut out = 0u8;
        self.check(unsafe { sys::cblx_contains_kmers(self.ctx, &lo, &hi, 1, &mut out) });
        out != 0
    }

    /// Adds a k-mer; returns `true` if it was absent (`src/cbl.rs:226-228`). One device round trip per call: a caller with
    /// many k-mers uses `insert_kmers`.
    #[inline]
    pub fn insert(&mut self, kmer: IntKmer<K, T>) -> bool {
        let (lo, hi) = kmer.to_int().split();
        let mut absent = 0u8;
        self.check(unsafe { sys::cblx_insert_kmers(self.ctx, &lo, &hi, 1, &mut absent) });
        absent != 0
    }

    /// `kmers.iter().map(|k| self.insert(*k)).collect()` in one call.
    pub fn insert_kmers(&mut self, kmers: &[IntKmer<K, T>]) -> Vec<bool> {
        let (lo, hi): (Vec<u64>, Vec<u64>) = kmers.iter().map(|k| k.to_int().split()).unzip();
        let mut absent = vec![0u8; kmers.len()];
        self.check(unsafe { sys::cblx_insert_kmers(self.ctx, lo.as_ptr(), hi.as_ptr(), kmers.len() as u64, absent.as_mut_ptr()) });
        absent.into_iter().map(|b| b != 0).collect()
    }

    fn assert_len(seq: &[u8]) {
        // the reference's assert, before anything is handed over (`src/cbl.rs:294-299,312-317,329-334`)
        assert!(seq.len() >= K, "Sequence size ({}) is smaller than K ({})", seq.len(), K);
    }

    /// Returns `true` if the set contains all the k-mers of a sequence (`src/cbl.rs:293-307`).
    #[inline]
    pub fn contains_all(&mut self, seq: &[u8]) -> bool {
        Self::assert_len(seq);
        let mut out: c_int = 0;
        self.check(unsafe { sys::cblx_contains_all(self.ctx, seq.as_ptr(), seq.len() as u64, &mut out) });
        out != 0
    }

    /// For each k-mer of a sequence, `true` if it is in the set (`src/cbl.rs:311-324`): chunk by chunk, forward-strand words
    /// before reverse-strand ones inside a chunk of a canonical index, as `get_seq_words` orders them.
    #[inline]
    pub fn contains_seq(&mut self, seq: &[u8]) -> Vec<bool> {
        Self::assert_len(seq);
        let cap = (seq.len() - K + 1) as u64;
        let mut out = vec![0u8; cap as usize];
        let mut n = 0u64;
        self.check(unsafe { sys::cblx_contains_seq(self.ctx, seq.as_ptr(), seq.len() as u64, out.as_mut_ptr(), cap, &mut n) });
        out.truncate(n as usize);
        out.into_iter().map(|b| b != 0).collect()
    }

    /// Adds all the k-mers of a sequence (`src/cbl.rs:328-339`). Enqueues; the kernels run at the next observer or `flush`.
    #[inline]
    pub fn insert_seq(&mut self, seq: &[u8]) {
        Self::assert_len(seq);
        self.check(unsafe { sys::cblx_insert_seq(self.ctx, seq.as_ptr(), seq.len() as u64) });
    }

    /// The reader loop `for record in reader { cbl.insert_seq(record.seq()) }` (`examples/cbl.rs:154-163`) inside the
    /// library: FASTA / FASTQ, plain or gzip, records in file order. Returns the number of records.
    pub fn insert_fastx_file<P: AsRef<Path> + Copy>(&mut self, path: P) -> u64 {
        let p = Self::cpath(path);
        let mut n = 0u64;
        self.check(unsafe { sys::cblx_insert_fastx_file(self.ctx, p.as_ptr(), &mut n) });
        n
    }

    /// Materialises everything enqueued so far (idempotent; every observer does it first).
    pub fn flush(&mut self) {
        self.check(unsafe { sys::cblx_flush(self.ctx) });
    }

    /// Iterates over the k-mers of the set in the reference's order (`src/cbl.rs:358-361`): prefixes ascending, a Vec bucket in
    /// stored order, a Trie bucket ascending; every word through `revert_necklace_pos`. The k-mers are exported in one call.
    pub fn iter(&self) -> impl Iterator<Item = IntKmer<K, T>> + '_ {
        let n = self.count();
        let (mut lo, mut hi) = (vec![0u64; n], vec![0u64; n]);
        let mut got = 0u64;
        self.check(unsafe { sys::cblx_export_kmers(self.ctx, lo.as_mut_ptr(), hi.as_mut_ptr(), n as u64, &mut got) });
        lo.truncate(got as usize);
        hi.truncate(got as usize);
        lo.into_iter().zip(hi).map(|(l, h)| IntKmer::from_int(T::join(l, h)))
    }

    fn bucket_table(&self) -> (Vec<u32>, Vec<u32>) {
        let mut nb = 0u64;
        self.check(unsafe { sys::cblx_num_buckets(self.ctx, &mut nb) });
        let (mut prefix, mut len) = (vec![0u32; nb as usize], vec![0u32; nb as usize]);
        let mut got = 0u64;
        self.check(unsafe { sys::cblx_bucket_sizes(self.ctx, prefix.as_mut_ptr(), len.as_mut_ptr(), std::ptr::null_mut(), nb, &mut got) });
        prefix.truncate(got as usize);
        len.truncate(got as usize);
        (prefix, len)
    }

    /// Proportion of the available prefixes in use (`src/cbl.rs:364-367`).
    #[inline]
    pub fn prefix_load(&self) -> f64 {
        let mut nb = 0u64;
        self.check(unsafe { sys::cblx_num_buckets(self.ctx, &mut nb) });
        nb as f64 / (1u64 << PREFIX_BITS) as f64
    }

    /// The prefixes of the set with the sizes of their buckets (`src/cbl.rs:370-373`).
    #[inline]
    pub fn buckets_sizes(&self) -> impl Iterator<Item = (usize, usize)> + '_ {
        let (prefix, len) = self.bucket_table();
        prefix.into_iter().zip(len).map(|(p, l)| (p as usize, l as usize))
    }

    /// Number of buckets of each size (`src/cbl.rs:376-379`).
    pub fn buckets_size_count(&self) -> BTreeMap<usize, usize> {
        let mut m = BTreeMap::new();
        for (_, size) in self.buckets_sizes() {
            *m.entry(size).or_insert(0usize) += 1;
        }
        m
    }

    /// Proportion of the k-mers held by the buckets of each size (`src/cbl.rs:382-385`).
    pub fn buckets_load_repartition(&self) -> BTreeMap<usize, f64> {
        let total = self.count() as f64;
        self.buckets_size_count().into_iter().map(|(size, n)| (size, (size * n) as f64 / total)).collect()
    }
}

impl<const K: usize, T: PackedInt, const PREFIX_BITS: usize> Default for CBL<K, T, PREFIX_BITS> {
    fn default() -> Self {
        Self::new()
    }
}

impl<const K: usize, T: PackedInt, const PREFIX_BITS: usize> Clone for CBL<K, T, PREFIX_BITS> {
    /// `#[derive(Clone)]` of the reference (`src/cbl.rs:40`): `|=` into an empty set clones every bucket as stored (kind and
    /// order kept, `src/wordset/set_ops.rs:123-157`), on the device.
    fn clone(&self) -> Self {
        let s = Self::with(self.is_canonical());
        s.check(unsafe { sys::cblx_merge_assign(s.ctx, self.ctx) });
        s
    }
}

impl<const K: usize, T: PackedInt, const PREFIX_BITS: usize> BitOrAssign<&mut Self> for CBL<K, T, PREFIX_BITS> {
    /// Union of `self` and `other` in place (`src/cbl.rs:433-449`). As in the reference, `other`'s Vec buckets that also exist in
    /// `self` end up sorted (`iter_sorted`, `src/trievec/mod.rs:209-220`).
    fn bitor_assign(&mut self, other: &mut Self) {
        assert_eq!(self.is_canonical(), other.is_canonical(), "One of the index is canonical while the other isn't");
        self.check(unsafe { sys::cblx_merge_assign(self.ctx, other.ctx) });
    }
}

impl<const K: usize, T: PackedInt, const PREFIX_BITS: usize> CBL<K, T, PREFIX_BITS> {
    /// `let mut c = self.clone(); c |= other; c` without the clone's copy (no reference counterpart as a method: the device merge writes a new
    /// arena anyway, `cblx_merge_from`). `self` is untouched; `other` is left as `|=` leaves it (its Vec buckets that meet a bucket of `self` sorted).
    pub fn merged_with(&self, other: &mut Self) -> Self {
        assert_eq!(self.is_canonical(), other.is_canonical(), "One of the index is canonical while the other isn't");
        let s = Self::with(self.is_canonical());
        s.check(unsafe { sys::cblx_merge_from(s.ctx, self.ctx, other.ctx) });
        s
    }
}

impl<const K: usize, T: PackedInt, const PREFIX_BITS: usize> Drop for CBL<K, T, PREFIX_BITS> {
    fn drop(&mut self) {
        unsafe { sys::cblx_destroy(self.ctx) }
    }
}

#[cfg(test)]
mod tests {
    //! The reference's own membership tests for this surface (`src/cbl.rs:591-773`), restated; they need an MI355X.
    use super::*;

    const K: usize = 31;
    type T = u128; // 2K + POS_BITS = 68 bits (`build.rs:34-41`)

    fn seq(n: usize, seed: u64) -> Vec<u8> {
        let mut x = seed;
        (0..n)
            .map(|_| {
                x = x.wrapping_add(0x9E3779B97F4A7C15);
                let mut z = x;
                z = (z ^ (z >> 30)).wrapping_mul(0xBF58476D1CE4E5B9);
                z = (z ^ (z >> 27)).wrapping_mul(0x94D049BB133111EB);
                b"ACGT"[((z ^ (z >> 31)) & 3) as usize]
            })
            .collect()
    }

    #[test]
    fn insert_seq_then_contains_all() {
        let s = seq(10_000, 42);
        let mut cbl = CBL::<K, T>::new();
        cbl.insert_seq(&s);
        assert!(cbl.contains_all(&s));
        assert_eq!(cbl.contains_seq(&s).len(), s.len() - K + 1);
        assert!(cbl.count() <= s.len() - K + 1);
    }

    #[test]
    fn union_and_round_trip() {
        let (a, b) = (seq(5_000, 1), seq(5_000, 2));
        let (mut x, mut y) = (CBL::<K, T>::new(), CBL::<K, T>::new());
        x.insert_seq(&a);
        y.insert_seq(&b);
        x |= &mut y;
        assert!(x.contains_all(&a) && x.contains_all(&b));
        let z = CBL::<K, T>::from_bytes(&x.to_bytes());
        assert_eq!(z.count(), x.count());
        assert_eq!(z.to_bytes(), x.to_bytes());
    }

    /// `write_index` / `read_index` of `examples/cbl.rs:117-142` through this type's serde impls: the bytes bincode writes are the
    /// bytes `save_to_file` writes, and they read back to the same index (needs the dev-dependency `bincode = "1.3"`).
    #[test]
    fn serde_model_is_the_file_format() {
        use bincode::{DefaultOptions, Options};
        let mut cbl = CBL::<K, T>::new();
        for seed in 0..40 {
            cbl.insert_seq(&seq(3_000, seed)); // enough words under one prefix for both kinds of bucket
        }
        let opts = || DefaultOptions::new().with_varint_encoding().reject_trailing_bytes();
        let via_serde = opts().serialize(&cbl).unwrap();
        assert_eq!(via_serde, cbl.to_bytes());
        let back: CBL<K, T> = opts().deserialize(&via_serde).unwrap();
        assert_eq!(back.count(), cbl.count());
        assert_eq!(back.to_bytes(), via_serde);
    }

    #[test]
    #[should_panic(expected = "Cannot fit a 31-mer and its length in a 64-bit integer")]
    fn word_must_fit_the_integer_type() {
        let _ = CBL::<31, u64>::new(); // `src/cbl.rs:87-91`
    }

    #[test]
    #[should_panic(expected = "is smaller than K")]
    fn short_sequence_panics() {
        CBL::<K, T>::new().insert_seq(b"ACGT");
    }
}
