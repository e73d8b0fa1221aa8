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
