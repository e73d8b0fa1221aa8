"""CPU-side checks of the product: the C-ABI library loads and exports every symbol include/cblx.h declares (no
compute calls without a GPU), fails loudly without a device, and the scalar device/host code shared with the
kernels (cbl_amd/csrc/necklace.hpp) agrees with the definition."""
import ctypes as C
import re
import subprocess
from pathlib import Path

import pytest

import cbl_amd

ROOT = Path(__file__).resolve().parent.parent


def _header_functions():
    text = (ROOT / "include" / "cblx.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cblx_[a-z0-9_]+)\s*\(", text)) - {"cblx_bucket_cb"})


def test_library_exports_every_declared_symbol():
    L = cbl_amd.lib()
    names = _header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/cblx.h but not exported by libcblx.so"
    assert sorted(cbl_amd.SIGNATURES) == names
    assert L.cblx_abi_version() == 3


def test_rust_sys_crate_declares_every_symbol():
    """rust/cblx-sys/src/lib.rs (shipped as source: no Rust toolchain in this image) binds exactly the functions of the header, with
    the same number of parameters each, and repeats its constants."""
    header = (ROOT / "include" / "cblx.h").read_text()
    header_nc = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    rs = (ROOT / "rust" / "cblx-sys" / "src" / "lib.rs").read_text()
    rs_nc = re.sub(r"//.*", "", rs)
    extern = rs_nc[rs_nc.index('extern "C" {'):]
    rust_fns = {m.group(1): m.group(2) for m in re.finditer(r"pub fn (cblx_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*[^;]+)?;", extern, flags=re.S)}
    assert sorted(rust_fns) == _header_functions()

    def nparams(arglist):
        arglist = arglist.strip()
        if arglist in ("", "void"):
            return 0
        depth, n = 0, 1
        for ch in arglist:  # commas at depth 0 (callback types nest parentheses)
            depth += ch in "(<["
            depth -= ch in ")>]"
            n += ch == "," and depth == 0
        return n - (1 if arglist.rstrip().endswith(",") else 0)

    for name, rargs in rust_fns.items():
        m = re.search(r"\b%s\s*\((.*?)\)\s*;" % name, header_nc, flags=re.S)
        assert m, name
        assert nparams(m.group(1)) == nparams(rargs), f"{name}: the header and the Rust binding disagree on the number of parameters"
    for cname, val in re.findall(r"#define (CBLX_[A-Z_]+) (\d+)u?\b", header_nc):
        assert re.search(r"pub const %s: \w+ = %s;" % (cname, val), rs), f"{cname} = {val} missing from the Rust crate"
    for cname, val in re.findall(r"(CBLX_E[A-Z]+|CBLX_OK) = (\d+)", header_nc):
        assert re.search(r"pub const %s: c_int = %s;" % (cname, val), rs), cname
    # the facade binds only functions that exist
    facade = (ROOT / "rust" / "cbl-gpu" / "src" / "lib.rs").read_text()
    used = set(re.findall(r"sys::(cblx_[a-z0-9_]+)\(", facade))
    assert used and used <= set(rust_fns)
    # ... and carries every method of the reference surface for this path (src/cbl.rs:71-79,127-177,219-228,293-339,358-385,433-449)
    for meth in ("new", "new_canonical", "save_to_file", "load_from_file", "is_canonical", "count", "is_empty", "contains", "insert", "contains_all", "contains_seq",
                 "insert_seq", "iter", "prefix_load", "buckets_sizes", "buckets_size_count", "buckets_load_repartition"):
        assert re.search(r"pub fn %s\b" % meth, facade), meth
    assert "BitOrAssign<&mut Self>" in facade and "impl<const K: usize, T: PackedInt, const PREFIX_BITS: usize> Drop" in facade
    # the reference's assert, not a weaker one (/root/reference/src/cbl.rs:87-91): CBL::<31, u64> must panic as it does there
    assert re.search(r"assert!\(\s*Self::KMER_BITS \+ Self::POS_BITS <= T::BITS as usize,\s*\"Cannot fit a \{K\}-mer and its length in a \{\}-bit integer\"", facade)
    assert "Self::KMER_BITS <= T::BITS" not in facade
    # Serialize / Deserialize in the reference's serde data model (src/cbl.rs:40-54, src/wordset/mod.rs:382-437, src/trievec/mod.rs:8-15,
    # src/trie.rs:8-9,53-57, src/sliced_int.rs:110-114, src/bitvector/tiny/mod.rs:97-105): the calls that fix the bytes under bincode
    model = (ROOT / "rust" / "cbl-gpu" / "src" / "serde_model.rs").read_text()
    assert "mod serde_model;" in facade and "serialize_bytes(&self.to_bytes())" not in facade
    for call in ('serialize_struct("CBL", 2)', 'serialize_field("canonical"', 'serialize_field("wordset"', "serialize_map(Some(nb as usize))",
                 'serialize_newtype_struct("TrieVec"', 'serialize_newtype_variant("TrieOrVec", 0, "Vec"', 'serialize_tuple_variant("TrieOrVec", 1, "Trie", 2)',
                 'serialize_newtype_struct("Trie"', 'serialize_struct("TrieNode", 2)', 'serialize_field("bv"', 'serialize_field("children"', "serialize_bytes(&self.le[..self.bytes])",
                 "impl<const K: usize, T: PackedInt, const PREFIX_BITS: usize> Serialize for CBL<K, T, PREFIX_BITS>",
                 "impl<'de, const K: usize, T: PackedInt, const PREFIX_BITS: usize> de::Deserialize<'de> for CBL<K, T, PREFIX_BITS>",
                 'deserialize_enum("TrieOrVec", &["Vec", "Trie"]', "try_from_bytes(&file)"):
        assert call in model, call
    assert set(re.findall(r"sys::(cblx_[a-z0-9_]+)\(", model)) <= set(rust_fns)


def test_no_cpu_fallback_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(cbl_amd.CblxError) as e:
        cbl_amd.CBL(31, 24)
    assert e.value.code == cbl_amd.EDEVICE


def test_parameter_validation_needs_no_gpu():
    for k, pb in ((30, 24), (3, 2), (61, 24), (31, 0), (31, 33), (7, 18)):
        with pytest.raises(cbl_amd.CblxError) as e:
            cbl_amd.CBL(k, pb)
        assert e.value.code == cbl_amd.EINVAL, (k, pb)


def test_necklace_host_unit(tmp_path):
    exe = tmp_path / "necklace_unit"
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", str(exe), str(ROOT / "tests" / "host" / "necklace_unit.cpp")], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_pack_planes_host_unit(tmp_path):
    """The host packer of the bit planes (3 bits per base over PCIe, cbl_amd/csrc/xfer.hpp) against the scalar definition."""
    exe = tmp_path / "pack_planes_unit"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "--offload-arch=gfx950", "-o", str(exe), str(ROOT / "tests" / "host" / "pack_planes_unit.cpp")], check=True,
                   capture_output=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_fastx_planes_host_unit(tmp_path):
    """The parser threads' side of the FASTA / FASTQ file insert (cbl_amd/csrc/fastx_parse.hpp: regions, counting pass, PlaneSink)
    against a byte-by-byte definition, under AddressSanitizer + UBSan (the sanitizers run on the CPU build only)."""
    exe = tmp_path / "fastx_planes_unit"
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-o", str(exe),
                    str(ROOT / "tests" / "host" / "fastx_planes_unit.cpp")], check=True, capture_output=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr

def test_cut_plan_host_unit(tmp_path):
    """Host logic of the grouped receiver (cbl_amd/csrc/cuts.hpp: the key of the cut table, group cuts from the sampled histogram, the bin
    map) against the definitions, under AddressSanitizer + UBSan: table lookup == counting the cuts, bins of a rank consecutive, a
    segment value at most once per group, cuts multiples of 64 inside their range."""
    exe = tmp_path / "cut_plan_unit"
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-o", str(exe),
                    str(ROOT / "tests" / "host" / "cut_plan_unit.cpp")], check=True, capture_output=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.parametrize("k,pb,nreads,L", [(31, 24, 3000, 150), (9, 4, 1500, 100), (59, 28, 400, 250), (15, 6, 1500, 150), (11, 8, 40, 3000)])
def test_index_shard_cuts_speculative_equals_sequential(k, pb, nreads, L, tmp_path):
    """Host-only half of cblx_load_shard_from_file: the speculative search for the entry starts of `world` prefix ranges
    (an entry start can only be recognised; a fake one — the tail of a Vec whose elements end in 00 00 01 — must not be
    trusted) finds exactly what a walk over every entry finds, for byte-balanced cuts and for given prefix bounds."""
    import numpy as np

    from cbl_amd import synth
    from oracle import Oracle

    o = Oracle(k, pb)
    b, off = synth.reads(5, nreads, L)
    o.insert_seqs(b, off)
    path = tmp_path / "x.cbl"
    path.write_bytes(o.serialize())
    size = path.stat().st_size
    for world in (1, 2, 3, 8, 16):
        a = cbl_amd.index_shard_cuts(path, k, pb, world)
        s = cbl_amd.index_shard_cuts(path, k, pb, world, sequential=True)
        assert a[2] and s[2]
        assert (a[0] == s[0]).all() and (a[1] == s[1]).all()
        assert s[0][world] == size and all(s[0][i] <= s[0][i + 1] for i in range(world))
        bounds = s[1][1:world]
        a2 = cbl_amd.index_shard_cuts(path, k, pb, world, bounds)
        s2 = cbl_amd.index_shard_cuts(path, k, pb, world, bounds, sequential=True)
        assert a2[2] and (a2[0] == s2[0]).all() and (a2[1] == s2[1]).all()
        assert (s2[0] == s[0]).all()  # cutting at the first prefixes of the byte-balanced runs gives the same runs
        if world > 1:  # bounds between / beyond the stored prefixes
            odd = np.array(sorted({int(x) + 1 for x in bounds} | {0, (1 << pb) - 1}))[: world - 1].astype(np.uint32)
            odd = np.pad(odd, (0, world - 1 - len(odd)), constant_values=(1 << pb) - 1)
            a3 = cbl_amd.index_shard_cuts(path, k, pb, world, odd)
            s3 = cbl_amd.index_shard_cuts(path, k, pb, world, odd, sequential=True)
            assert a3[2] and (a3[0] == s3[0]).all() and (a3[1] == s3[1]).all()


def test_index_shard_cuts_rejects_garbage(tmp_path):
    p = tmp_path / "bad.cbl"
    p.write_bytes(bytes([0, 3]) + bytes(range(200)))
    with pytest.raises(cbl_amd.CblxError) as e:
        cbl_amd.index_shard_cuts(p, 31, 24, 2, sequential=True)
    assert e.value.code == cbl_amd.EFORMAT
