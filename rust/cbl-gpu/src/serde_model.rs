//! `Serialize` / `Deserialize` of the facade `CBL` in the REFERENCE'S serde data model, so that the reference's own
//! `write_index(&cbl, path)` / `read_index(path)` (`/root/reference/examples/cbl.rs:117-142`: bincode `DefaultOptions` + varint)
//! produce and accept the reference's bytes with nothing but the `use` line of `examples/cbl.rs` changed.
//!
//! The model, as the reference's derives and impls define it:
//!   `CBL`        struct { canonical: bool, wordset: WordSet }                  (`src/cbl.rs:40-54`; the two queues are `#[serde(skip)]`)
//!   `WordSet`    map<u32 prefix, TrieVec>, len = `tiered.len()`, ascending      (`src/wordset/mod.rs:382-396`)
//!   `TrieVec`    newtype struct of enum `TrieOrVec`                              (`src/trievec/mod.rs:8-15`)
//!                  variant 0 `Vec(Vec<SlicedInt>)`  — newtype variant holding a seq, STORED order
//!                  variant 1 `Trie(Trie, usize)`    — tuple variant (trie, len)
//!   `SlicedInt`  `serialize_bytes` of its BYTES little-endian bytes              (`src/sliced_int.rs:110-114`)
//!   `Trie`       newtype struct of `Box<TrieNode>`                               (`src/trie.rs:8-9`)
//!   `TrieNode`   struct { bv: TinyBitvector, children: Vec<Trie> }               (`src/trie.rs:53-57`)
//!   `TinyBitvector` seq of the set indices as `u8`, ascending, len = count       (`src/bitvector/tiny/mod.rs:97-105`)
//! Level d of a trie consumes big-endian byte d of the suffix (`src/sliced_int.rs:50-54`, `src/trie.rs:118-131`); the last level
//! has no children. A Trie is therefore a pure function of the bucket's ascending suffix list, which is what libcblx stores.
//!
//! Serialize walks `cblx_export_buckets` (ascending prefixes; a Vec bucket in stored order, a Trie bucket ascending) and feeds the
//! caller's serializer bucket by bucket. Deserialize visits the same model with ANY deserializer, re-encodes what it sees as the
//! index file format (SURVEY.md Appendix A — for bincode input that reproduces the input bytes) and hands it to `cblx_load`,
//! which keeps kinds and stored order exactly as `WordSetVisitor` does (`src/wordset/mod.rs:398-437`).
//!
//! NOT compiled in the image this repository was built in (no Rust toolchain there).
use std::fmt;
use std::marker::PhantomData;
use std::os::raw::{c_int, c_void};

use serde::de::{self, DeserializeSeed, Deserializer, EnumAccess, MapAccess, SeqAccess, VariantAccess, Visitor};
use serde::ser::{Error as SerError, Serialize, SerializeMap, SerializeSeq, SerializeStruct, SerializeTupleVariant, Serializer};

use super::{sys, PackedInt, CBL};

// ---- Serialize ------------------------------------------------------------------------------------------------------------------

/// One bucket as the export callback hands it over: suffix i = (hi[i] << 64) | lo[i].
#[derive(Clone, Copy)]
struct Bucket<'a> {
    kind: c_int,
    lo: &'a [u64],
    hi: Option<&'a [u64]>,
    bytes: usize,
}
impl<'a> Bucket<'a> {
    #[inline]
    fn value(&self, i: usize) -> u128 {
        ((self.hi.map_or(0, |h| h[i]) as u128) << 64) | self.lo[i] as u128
    }
    /// big-endian byte `level` of suffix i (what trie level `level` consumes)
    #[inline]
    fn be_byte(&self, i: usize, level: usize) -> u8 {
        (self.value(i) >> (8 * (self.bytes - 1 - level))) as u8
    }
}

struct Sliced {
    le: [u8; 16],
    bytes: usize,
}
impl Serialize for Sliced {
    fn serialize<S: Serializer>(&self, s: S) -> Result<S::Ok, S::Error> {
        s.serialize_bytes(&self.le[..self.bytes])
    }
}

struct VecBody<'a>(Bucket<'a>);
impl<'a> Serialize for VecBody<'a> {
    fn serialize<S: Serializer>(&self, s: S) -> Result<S::Ok, S::Error> {
        let b = &self.0;
        let mut seq = s.serialize_seq(Some(b.lo.len()))?;
        for i in 0..b.lo.len() {
            seq.serialize_element(&Sliced { le: b.value(i).to_le_bytes(), bytes: b.bytes })?;
        }
        seq.end()
    }
}

/// The trie node of suffixes [a, z) of an ascending bucket at `level` (they share their first `level` big-endian bytes).
#[derive(Clone, Copy)]
struct Node<'a> {
    b: Bucket<'a>,
    level: usize,
    a: usize,
    z: usize,
}
impl<'a> Node<'a> {
    /// the sub-ranges of equal byte `level`, in ascending byte order
    fn runs(&self) -> impl Iterator<Item = (u8, usize, usize)> + '_ {
        let mut i = self.a;
        std::iter::from_fn(move || {
            if i >= self.z {
                return None;
            }
            let v = self.b.be_byte(i, self.level);
            let start = i;
            while i < self.z && self.b.be_byte(i, self.level) == v {
                i += 1;
            }
            Some((v, start, i))
        })
    }
}
struct NodeBv<'a>(Node<'a>);
impl<'a> Serialize for NodeBv<'a> {
    fn serialize<S: Serializer>(&self, s: S) -> Result<S::Ok, S::Error> {
        let mut seq = s.serialize_seq(Some(self.0.runs().count()))?;
        for (v, _, _) in self.0.runs() {
            seq.serialize_element(&v)?;
        }
        seq.end()
    }
}
struct NodeChildren<'a>(Node<'a>);
impl<'a> Serialize for NodeChildren<'a> {
    fn serialize<S: Serializer>(&self, s: S) -> Result<S::Ok, S::Error> {
        let n = self.0;
        if n.level + 1 == n.b.bytes {
            return s.serialize_seq(Some(0))?.end(); // the last level holds no children (`src/trie.rs:118-131`)
        }
        let mut seq = s.serialize_seq(Some(n.runs().count()))?;
        for (_, a, z) in n.runs() {
            seq.serialize_element(&TrieSer(Node { b: n.b, level: n.level + 1, a, z }))?;
        }
        seq.end()
    }
}
struct NodeSer<'a>(Node<'a>);
impl<'a> Serialize for NodeSer<'a> {
    fn serialize<S: Serializer>(&self, s: S) -> Result<S::Ok, S::Error> {
        let mut st = s.serialize_struct("TrieNode", 2)?;
        st.serialize_field("bv", &NodeBv(self.0))?;
        st.serialize_field("children", &NodeChildren(self.0))?;
        st.end()
    }
}
struct TrieSer<'a>(Node<'a>);
impl<'a> Serialize for TrieSer<'a> {
    fn serialize<S: Serializer>(&self, s: S) -> Result<S::Ok, S::Error> {
        s.serialize_newtype_struct("Trie", &NodeSer(self.0)) // Box<TrieNode> is transparent
    }
}

struct TrieOrVecSer<'a>(Bucket<'a>);
impl<'a> Serialize for TrieOrVecSer<'a> {
    fn serialize<S: Serializer>(&self, s: S) -> Result<S::Ok, S::Error> {
        let b = self.0;
        if b.kind == 0 {
            s.serialize_newtype_variant("TrieOrVec", 0, "Vec", &VecBody(b))
        } else {
            let mut tv = s.serialize_tuple_variant("TrieOrVec", 1, "Trie", 2)?;
            tv.serialize_field(&TrieSer(Node { b, level: 0, a: 0, z: b.lo.len() }))?;
            tv.serialize_field(&b.lo.len())?; // usize
            tv.end()
        }
    }
}
struct TrieVecSer<'a>(Bucket<'a>);
impl<'a> Serialize for TrieVecSer<'a> {
    fn serialize<S: Serializer>(&self, s: S) -> Result<S::Ok, S::Error> {
        s.serialize_newtype_struct("TrieVec", &TrieOrVecSer(self.0))
    }
}

struct MapSink<M: SerializeMap> {
    map: M,
    bytes: usize,
    err: Option<M::Error>,
}
unsafe extern "C" fn entry_cb<M: SerializeMap>(user: *mut c_void, prefix: u32, kind: c_int, len: u64, lo: *const u64, hi: *const u64) -> c_int {
    let sink = &mut *(user as *mut MapSink<M>);
    let n = len as usize;
    let lo: &[u64] = if n == 0 { &[] } else { std::slice::from_raw_parts(lo, n) };
    let hi: Option<&[u64]> = if n == 0 || hi.is_null() { None } else { Some(std::slice::from_raw_parts(hi, n)) };
    match sink.map.serialize_entry(&prefix, &TrieVecSer(Bucket { kind, lo, hi, bytes: sink.bytes })) {
        Ok(()) => 0,
        Err(e) => {
            sink.err = Some(e);
            1 // stops the walk
        }
    }
}

struct WordSetSer<'a, const K: usize, T: PackedInt, const PREFIX_BITS: usize>(&'a CBL<K, T, PREFIX_BITS>);
impl<'a, const K: usize, T: PackedInt, const PREFIX_BITS: usize> Serialize for WordSetSer<'a, K, T, PREFIX_BITS> {
    fn serialize<S: Serializer>(&self, s: S) -> Result<S::Ok, S::Error> {
        let cbl = self.0;
        let mut nb = 0u64;
        let mut consts = sys::cblx_consts::default();
        unsafe {
            if sys::cblx_num_buckets(cbl.ctx, &mut nb) != sys::CBLX_OK || sys::cblx_get_consts(cbl.ctx, &mut consts) != sys::CBLX_OK {
                return Err(S::Error::custom(cbl.last_error()));
            }
        }
        let mut sink = MapSink::<S::SerializeMap> { map: s.serialize_map(Some(nb as usize))?, bytes: consts.bytes as usize, err: None };
        let cb: unsafe extern "C" fn(*mut c_void, u32, c_int, u64, *const u64, *const u64) -> c_int = entry_cb::<S::SerializeMap>;
        let rc = unsafe { sys::cblx_export_buckets(cbl.ctx, Some(cb), &mut sink as *mut _ as *mut c_void) };
        if let Some(e) = sink.err.take() {
            return Err(e);
        }
        if rc != sys::CBLX_OK {
            return Err(S::Error::custom(cbl.last_error()));
        }
        sink.map.end()
    }
}

impl<const K: usize, T: PackedInt, const PREFIX_BITS: usize> Serialize for CBL<K, T, PREFIX_BITS> {
    fn serialize<S: Serializer>(&self, s: S) -> Result<S::Ok, S::Error> {
        let mut st = s.serialize_struct("CBL", 2)?;
        st.serialize_field("canonical", &self.is_canonical())?;
        st.serialize_field("wordset", &WordSetSer(self))?;
        st.end()
    }
}

// ---- Deserialize: visit the model, re-encode as the index file format, cblx_load -------------------------------------------------

/// bincode 1.3 varint (`DefaultOptions::with_varint_encoding`): every length, variant index and integer wider than a byte.
fn varint(out: &mut Vec<u8>, v: u64) {
    if v <= 250 {
        out.push(v as u8);
    } else if v < 1 << 16 {
        out.push(0xFB);
        out.extend_from_slice(&(v as u16).to_le_bytes());
    } else if v < 1 << 32 {
        out.push(0xFC);
        out.extend_from_slice(&(v as u32).to_le_bytes());
    } else {
        out.push(0xFD);
        out.extend_from_slice(&v.to_le_bytes());
    }
}

struct SlicedSeed<'o> {
    out: &'o mut Vec<u8>,
    bytes: usize,
}
impl<'de, 'o> DeserializeSeed<'de> for SlicedSeed<'o> {
    type Value = ();
    fn deserialize<D: Deserializer<'de>>(self, d: D) -> Result<(), D::Error> {
        d.deserialize_bytes(self)
    }
}
impl<'de, 'o> Visitor<'de> for SlicedSeed<'o> {
    type Value = ();
    fn expecting(&self, f: &mut fmt::Formatter) -> fmt::Result {
        f.write_str("an integer sliced into bytes")
    }
    fn visit_bytes<E: de::Error>(self, b: &[u8]) -> Result<(), E> {
        if b.len() != self.bytes {
            return Err(E::invalid_length(b.len(), &self));
        }
        varint(self.out, b.len() as u64);
        self.out.extend_from_slice(b);
        Ok(())
    }
    fn visit_byte_buf<E: de::Error>(self, b: Vec<u8>) -> Result<(), E> {
        self.visit_bytes(&b)
    }
}

struct VecSeed<'o> {
    out: &'o mut Vec<u8>,
    bytes: usize,
}
impl<'de, 'o> DeserializeSeed<'de> for VecSeed<'o> {
    type Value = ();
    fn deserialize<D: Deserializer<'de>>(self, d: D) -> Result<(), D::Error> {
        d.deserialize_seq(self)
    }
}
impl<'de, 'o> Visitor<'de> for VecSeed<'o> {
    type Value = ();
    fn expecting(&self, f: &mut fmt::Formatter) -> fmt::Result {
        f.write_str("a sequence of sliced integers")
    }
    fn visit_seq<A: SeqAccess<'de>>(self, mut seq: A) -> Result<(), A::Error> {
        let mut body = Vec::new();
        let mut n = 0u64;
        while seq.next_element_seed(SlicedSeed { out: &mut body, bytes: self.bytes })?.is_some() {
            n += 1;
        }
        varint(self.out, n);
        self.out.extend_from_slice(&body);
        Ok(())
    }
}

struct BvSeed<'o>(&'o mut Vec<u8>);
impl<'de, 'o> DeserializeSeed<'de> for BvSeed<'o> {
    type Value = ();
    fn deserialize<D: Deserializer<'de>>(self, d: D) -> Result<(), D::Error> {
        d.deserialize_seq(self)
    }
}
impl<'de, 'o> Visitor<'de> for BvSeed<'o> {
    type Value = ();
    fn expecting(&self, f: &mut fmt::Formatter) -> fmt::Result {
        f.write_str("a small bitvector")
    }
    fn visit_seq<A: SeqAccess<'de>>(self, mut seq: A) -> Result<(), A::Error> {
        let mut idx = Vec::new();
        while let Some(i) = seq.next_element::<u8>()? {
            idx.push(i);
        }
        varint(self.0, idx.len() as u64);
        self.0.extend_from_slice(&idx); // u8 is one raw byte
        Ok(())
    }
}

struct ChildrenSeed<'o>(&'o mut Vec<u8>);
impl<'de, 'o> DeserializeSeed<'de> for ChildrenSeed<'o> {
    type Value = ();
    fn deserialize<D: Deserializer<'de>>(self, d: D) -> Result<(), D::Error> {
        d.deserialize_seq(self)
    }
}
impl<'de, 'o> Visitor<'de> for ChildrenSeed<'o> {
    type Value = ();
    fn expecting(&self, f: &mut fmt::Formatter) -> fmt::Result {
        f.write_str("the children of a trie node")
    }
    fn visit_seq<A: SeqAccess<'de>>(self, mut seq: A) -> Result<(), A::Error> {
        let mut body = Vec::new();
        let mut n = 0u64;
        while seq.next_element_seed(TrieSeed(&mut body))?.is_some() {
            n += 1;
        }
        varint(self.0, n);
        self.0.extend_from_slice(&body);
        Ok(())
    }
}

/// `Trie` = newtype struct of `TrieNode` = struct { bv, children }
struct TrieSeed<'o>(&'o mut Vec<u8>);
impl<'de, 'o> DeserializeSeed<'de> for TrieSeed<'o> {
    type Value = ();
    fn deserialize<D: Deserializer<'de>>(self, d: D) -> Result<(), D::Error> {
        d.deserialize_newtype_struct("Trie", self)
    }
}
impl<'de, 'o> Visitor<'de> for TrieSeed<'o> {
    type Value = ();
    fn expecting(&self, f: &mut fmt::Formatter) -> fmt::Result {
        f.write_str("a trie")
    }
    fn visit_newtype_struct<D: Deserializer<'de>>(self, d: D) -> Result<(), D::Error> {
        d.deserialize_struct("TrieNode", &["bv", "children"], NodeVisitor(self.0))
    }
    // formats that skip the newtype wrapper hand the node over directly
    fn visit_seq<A: SeqAccess<'de>>(self, seq: A) -> Result<(), A::Error> {
        NodeVisitor(self.0).visit_seq(seq)
    }
    fn visit_map<A: MapAccess<'de>>(self, map: A) -> Result<(), A::Error> {
        NodeVisitor(self.0).visit_map(map)
    }
}
struct NodeVisitor<'o>(&'o mut Vec<u8>);
impl<'de, 'o> Visitor<'de> for NodeVisitor<'o> {
    type Value = ();
    fn expecting(&self, f: &mut fmt::Formatter) -> fmt::Result {
        f.write_str("a trie node")
    }
    fn visit_seq<A: SeqAccess<'de>>(self, mut seq: A) -> Result<(), A::Error> {
        seq.next_element_seed(BvSeed(&mut *self.0))?.ok_or_else(|| de::Error::invalid_length(0, &"a trie node with 2 fields"))?;
        seq.next_element_seed(ChildrenSeed(&mut *self.0))?.ok_or_else(|| de::Error::invalid_length(1, &"a trie node with 2 fields"))?;
        Ok(())
    }
    fn visit_map<A: MapAccess<'de>>(self, mut map: A) -> Result<(), A::Error> {
        // self-describing formats: the fields by name, `bv` before `children` in the encoding whatever their order here
        let (mut bv, mut ch) = (None, None);
        while let Some(key) = map.next_key::<String>()? {
            match key.as_str() {
                "bv" => {
                    let mut b = Vec::new();
                    map.next_value_seed(BvSeed(&mut b))?;
                    bv = Some(b);
                }
                "children" => {
                    let mut c = Vec::new();
                    map.next_value_seed(ChildrenSeed(&mut c))?;
                    ch = Some(c);
                }
                other => return Err(de::Error::unknown_field(other, &["bv", "children"])),
            }
        }
        self.0.extend_from_slice(&bv.ok_or_else(|| de::Error::missing_field("bv"))?);
        self.0.extend_from_slice(&ch.ok_or_else(|| de::Error::missing_field("children"))?);
        Ok(())
    }
}

/// variant of `TrieOrVec` by index (bincode) or by name (self-describing formats)
struct VariantIdx(u32);
impl<'de> de::Deserialize<'de> for VariantIdx {
    fn deserialize<D: Deserializer<'de>>(d: D) -> Result<Self, D::Error> {
        struct V;
        impl<'de> Visitor<'de> for V {
            type Value = VariantIdx;
            fn expecting(&self, f: &mut fmt::Formatter) -> fmt::Result {
                f.write_str("variant `Vec` or `Trie`")
            }
            fn visit_u64<E: de::Error>(self, v: u64) -> Result<VariantIdx, E> {
                if v < 2 {
                    Ok(VariantIdx(v as u32))
                } else {
                    Err(E::invalid_value(de::Unexpected::Unsigned(v), &"variant index 0 <= i < 2"))
                }
            }
            fn visit_str<E: de::Error>(self, v: &str) -> Result<VariantIdx, E> {
                match v {
                    "Vec" => Ok(VariantIdx(0)),
                    "Trie" => Ok(VariantIdx(1)),
                    _ => Err(E::unknown_variant(v, &["Vec", "Trie"])),
                }
            }
        }
        d.deserialize_identifier(V)
    }
}

struct TrieTuple<'o>(&'o mut Vec<u8>);
impl<'de, 'o> Visitor<'de> for TrieTuple<'o> {
    type Value = ();
    fn expecting(&self, f: &mut fmt::Formatter) -> fmt::Result {
        f.write_str("tuple variant TrieOrVec::Trie")
    }
    fn visit_seq<A: SeqAccess<'de>>(self, mut seq: A) -> Result<(), A::Error> {
        seq.next_element_seed(TrieSeed(&mut *self.0))?.ok_or_else(|| de::Error::invalid_length(0, &"tuple variant TrieOrVec::Trie with 2 elements"))?;
        let len: u64 = seq.next_element()?.ok_or_else(|| de::Error::invalid_length(1, &"tuple variant TrieOrVec::Trie with 2 elements"))?;
        varint(self.0, len); // usize travels as u64
        Ok(())
    }
}

/// `TrieVec` = newtype struct of enum `TrieOrVec`
struct TrieVecSeed<'o> {
    out: &'o mut Vec<u8>,
    bytes: usize,
}
impl<'de, 'o> DeserializeSeed<'de> for TrieVecSeed<'o> {
    type Value = ();
    fn deserialize<D: Deserializer<'de>>(self, d: D) -> Result<(), D::Error> {
        d.deserialize_newtype_struct("TrieVec", self)
    }
}
impl<'de, 'o> Visitor<'de> for TrieVecSeed<'o> {
    type Value = ();
    fn expecting(&self, f: &mut fmt::Formatter) -> fmt::Result {
        f.write_str("a bucket (Vec or Trie)")
    }
    fn visit_newtype_struct<D: Deserializer<'de>>(self, d: D) -> Result<(), D::Error> {
        d.deserialize_enum("TrieOrVec", &["Vec", "Trie"], self)
    }
    fn visit_enum<A: EnumAccess<'de>>(self, data: A) -> Result<(), A::Error> {
        let (VariantIdx(i), variant) = data.variant::<VariantIdx>()?;
        varint(self.out, i as u64);
        if i == 0 {
            variant.newtype_variant_seed(VecSeed { out: self.out, bytes: self.bytes })
        } else {
            variant.tuple_variant(2, TrieTuple(self.out))
        }
    }
}

/// `WordSet` = map<u32, TrieVec>; yields (entries, their encoding)
struct WordSetSeed {
    bytes: usize,
}
impl<'de> DeserializeSeed<'de> for WordSetSeed {
    type Value = (u64, Vec<u8>);
    fn deserialize<D: Deserializer<'de>>(self, d: D) -> Result<Self::Value, D::Error> {
        d.deserialize_map(self)
    }
}
impl<'de> Visitor<'de> for WordSetSeed {
    type Value = (u64, Vec<u8>);
    fn expecting(&self, f: &mut fmt::Formatter) -> fmt::Result {
        f.write_str("a wordset")
    }
    fn visit_map<A: MapAccess<'de>>(self, mut access: A) -> Result<Self::Value, A::Error> {
        let mut body = Vec::new();
        let mut n = 0u64;
        while let Some(prefix) = access.next_key::<u32>()? {
            varint(&mut body, prefix as u64);
            access.next_value_seed(TrieVecSeed { out: &mut body, bytes: self.bytes })?;
            n += 1;
        }
        Ok((n, body))
    }
}

struct CblVisitor<const K: usize, T: PackedInt, const PREFIX_BITS: usize>(PhantomData<T>);
impl<const K: usize, T: PackedInt, const PREFIX_BITS: usize> CblVisitor<K, T, PREFIX_BITS> {
    /// BYTES of the reference's `SlicedInt`: ceil(SUFFIX_BITS / 8), SUFFIX_BITS = 2K + POS_BITS - PREFIX_BITS (`src/cbl.rs:16-32`)
    fn suffix_bytes() -> usize {
        let pos_bits = (2 * K).next_power_of_two().trailing_zeros() as usize;
        (2 * K + pos_bits - PREFIX_BITS + 7) / 8
    }
    fn assemble<E: de::Error>(canonical: bool, wordset: (u64, Vec<u8>)) -> Result<CBL<K, T, PREFIX_BITS>, E> {
        let mut file = Vec::with_capacity(wordset.1.len() + 10);
        file.push(canonical as u8);
        varint(&mut file, wordset.0);
        file.extend_from_slice(&wordset.1);
        CBL::<K, T, PREFIX_BITS>::try_from_bytes(&file).map_err(E::custom)
    }
}
impl<'de, const K: usize, T: PackedInt, const PREFIX_BITS: usize> Visitor<'de> for CblVisitor<K, T, PREFIX_BITS> {
    type Value = CBL<K, T, PREFIX_BITS>;
    fn expecting(&self, f: &mut fmt::Formatter) -> fmt::Result {
        f.write_str("struct CBL")
    }
    fn visit_seq<A: SeqAccess<'de>>(self, mut seq: A) -> Result<Self::Value, A::Error> {
        let canonical: bool = seq.next_element()?.ok_or_else(|| de::Error::invalid_length(0, &"struct CBL with 2 elements"))?;
        let ws = seq.next_element_seed(WordSetSeed { bytes: Self::suffix_bytes() })?.ok_or_else(|| de::Error::invalid_length(1, &"struct CBL with 2 elements"))?;
        Self::assemble(canonical, ws)
    }
    fn visit_map<A: MapAccess<'de>>(self, mut map: A) -> Result<Self::Value, A::Error> {
        let (mut canonical, mut ws) = (None, None);
        while let Some(key) = map.next_key::<String>()? {
            match key.as_str() {
                "canonical" => canonical = Some(map.next_value::<bool>()?),
                "wordset" => ws = Some(map.next_value_seed(WordSetSeed { bytes: Self::suffix_bytes() })?),
                other => return Err(de::Error::unknown_field(other, &["canonical", "wordset"])),
            }
        }
        Self::assemble(canonical.ok_or_else(|| de::Error::missing_field("canonical"))?, ws.ok_or_else(|| de::Error::missing_field("wordset"))?)
    }
}

impl<'de, const K: usize, T: PackedInt, const PREFIX_BITS: usize> de::Deserialize<'de> for CBL<K, T, PREFIX_BITS> {
    fn deserialize<D: Deserializer<'de>>(d: D) -> Result<Self, D::Error> {
        d.deserialize_struct("CBL", &["canonical", "wordset"], CblVisitor::<K, T, PREFIX_BITS>(PhantomData))
    }
}
