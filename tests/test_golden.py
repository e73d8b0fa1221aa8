"""Golden vectors (tests/golden/index_vectors.json, made by tests/golden/make_golden.py): the oracle must keep
reproducing them (CPU), and the HIP path must hit the same bytes without the oracle in the loop (GPU)."""
import hashlib
import json
import os

import pytest

from cbl_amd import synth

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "index_vectors.json")))


@pytest.mark.parametrize("v", GOLD["synthetic"], ids=lambda v: v["name"])
def test_oracle_reproduces_synthetic_vectors(v):
    from oracle import Oracle

    bases, offsets = synth.reads(v["seed"], v["n_reads"], v["read_len"])
    assert bases[: v["read_len"]].tobytes().decode() == v["first_read"]  # the generator itself is pinned
    o = Oracle(v["k"], v["prefix_bits"], v["canonical"])
    o.insert_seqs(bases, offsets)
    blob = o.serialize()
    assert (o.count(), o.n_buckets(), len(blob)) == (v["count"], v["n_buckets"], v["index_bytes"])
    assert hashlib.sha256(blob).hexdigest() == v["sha256"]


@pytest.mark.parametrize("v", GOLD["literal"], ids=lambda v: v["name"])
def test_oracle_reproduces_literal_vectors(v):
    from oracle import Oracle

    o = Oracle(v["k"], v["prefix_bits"], v["canonical"])
    for s in v["sequences"]:
        o.insert_seq(s.encode())
    assert o.serialize().hex() == v["index_hex"] and o.count() == v["count"]


@pytest.mark.gpu
@pytest.mark.parametrize("v", GOLD["synthetic"], ids=lambda v: v["name"])
def test_gpu_reproduces_synthetic_vectors(v):
    import cbl_amd

    bases, offsets = synth.reads(v["seed"], v["n_reads"], v["read_len"])
    g = cbl_amd.CBL(v["k"], v["prefix_bits"], canonical=v["canonical"])
    g.insert_seqs(bases, offsets)
    blob = g.serialize()
    assert (g.count(), g.num_buckets(), len(blob)) == (v["count"], v["n_buckets"], v["index_bytes"])
    assert hashlib.sha256(blob).hexdigest() == v["sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("v", GOLD["literal"], ids=lambda v: v["name"])
def test_gpu_reproduces_literal_vectors(v):
    import cbl_amd

    g = cbl_amd.CBL(v["k"], v["prefix_bits"], canonical=v["canonical"])
    for s in v["sequences"]:
        g.insert_seq(s.encode())
    assert g.serialize().hex() == v["index_hex"] and g.count() == v["count"]
    # load -> serialize identity through the ABI
    h = cbl_amd.CBL(v["k"], v["prefix_bits"], canonical=v["canonical"])
    h.load(bytes.fromhex(v["index_hex"]))
    assert h.serialize().hex() == v["index_hex"]


# ---- single-k-mer surface: insert return values, membership flags of a query sequence, iteration order ---------------------
@pytest.mark.parametrize("v", GOLD["kmers"], ids=lambda v: v["name"])
def test_oracle_reproduces_kmer_vectors(v):
    from oracle import Oracle

    o = Oracle(v["k"], v["prefix_bits"], v["canonical"])
    o.insert_seq(v["sequence"].encode())
    assert [o.insert_kmer(int(x, 16)) for x in v["kmers"]] == v["was_absent"]
    assert o.serialize().hex() == v["index_hex"] and o.count() == v["count"]
    assert [o.contains_word(w) for w in o.seq_words(v["query"].encode())] == v["query_flags"]
    assert [hex(o.kmer_of_word(w)) for w in o.iter_words()] == v["iter"]


@pytest.mark.gpu
@pytest.mark.parametrize("v", GOLD["kmers"], ids=lambda v: v["name"])
def test_gpu_reproduces_kmer_vectors(v):
    import cbl_amd

    g = cbl_amd.CBL(v["k"], v["prefix_bits"], canonical=v["canonical"])
    g.insert_seq(v["sequence"].encode())
    assert g.insert_kmers([int(x, 16) for x in v["kmers"]]).tolist() == v["was_absent"]
    assert g.serialize().hex() == v["index_hex"] and g.count() == v["count"]
    assert g.contains_seq(v["query"].encode()) == v["query_flags"]
    assert [hex(x) for x in g.iter()] == v["iter"]
    assert g.contains_kmers([int(x, 16) for x in v["iter"]]).all()
    # one k-mer at a time gives the same answers as the batch
    h = cbl_amd.CBL(v["k"], v["prefix_bits"], canonical=v["canonical"])
    h.insert_seq(v["sequence"].encode())
    assert [h.insert(int(x, 16)) for x in v["kmers"][:25]] == v["was_absent"][:25]
