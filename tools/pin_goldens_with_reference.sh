#!/bin/bash
# Pins tests/golden/index_vectors.json against the REAL reference (imartayan/CBL), for anyone with `cargo +nightly`.
#
# The image this repository was built in has no Rust toolchain, so the serialized bytes of the golden vectors come from the
# C++ oracle (oracle/cbl_oracle.hpp) cross-checked by an independent Python restatement (oracle/pyref.py): "parity unpinned"
# for the bincode / serde layer (DESIGN_HISTORY.md §7). This script closes that gap: for every golden case it writes the case's FASTA
# with the generator the vectors were made with (cbl_amd.synth, numpy only), builds the reference's CLI for the case's K /
# PREFIX_BITS as its README prescribes (README.md:112-129: K=.. PREFIX_BITS=.. cargo +nightly build --release --examples),
# runs `cbl build [-c] <fasta> -o <index>` (examples/cbl.rs:147-167) and compares SHA-256 with the golden file.
#
# Usage: tools/pin_goldens_with_reference.sh /path/to/CBL-checkout   (git clone --recursive https://github.com/imartayan/CBL.git)
# Nothing of the reference is copied into this repository; outputs go to a temporary directory.
set -euo pipefail
REF=${1:?usage: $0 /path/to/CBL-checkout}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
command -v cargo > /dev/null || { echo "cargo not found (the reference needs Rust nightly)"; exit 2; }
[ -f "$REF/Cargo.toml" ] || { echo "$REF is not a checkout of imartayan/CBL"; exit 2; }
[ -f "$REF/cxx/sux/sux/bits/Rank9Sel.hpp" ] || [ -d "$REF/cxx/sux/sux" ] || { echo "submodules missing: git submodule update --init --recursive"; exit 2; }
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
# one line per golden case: name k prefix_bits canonical sha256, and its FASTA under $TMP/<name>.fa
python3 - "$ROOT" "$TMP" <<'PY'
import hashlib, json, os, sys
root, tmp = sys.argv[1], sys.argv[2]
sys.path.insert(0, root)
from cbl_amd import synth  # the generator the tests use (numpy only, no GPU)
g = json.load(open(os.path.join(root, "tests", "golden", "index_vectors.json")))
with open(os.path.join(tmp, "cases.txt"), "w") as out:
    for case in g["synthetic"]:   # reads = synth.reads(seed, n_reads, read_len): iid ACGT, splitmix64
        bases, offsets = synth.reads(case["seed"], case["n_reads"], case["read_len"])
        raw = bases.tobytes()
        with open(os.path.join(tmp, case["name"] + ".fa"), "wb") as f:
            for i in range(case["n_reads"]):
                f.write(b">r%d\n" % i + raw[int(offsets[i]):int(offsets[i + 1])] + b"\n")
        out.write("%s %d %d %d %s\n" % (case["name"], case["k"], case["prefix_bits"], int(case["canonical"]), case["sha256"]))
    for case in g["literal"]:     # literal sequences; the expected bytes are in the file (index_hex)
        with open(os.path.join(tmp, case["name"] + ".fa"), "wb") as f:
            for i, sq in enumerate(case["sequences"]):
                f.write(b">r%d\n" % i + sq.encode() + b"\n")
        sha = hashlib.sha256(bytes.fromhex(case["index_hex"])).hexdigest()
        out.write("%s %d %d %d %s\n" % (case["name"], case["k"], case["prefix_bits"], int(case["canonical"]), sha))
PY
fail=0
while read -r name k pb canon sha; do
  echo "== $name (K=$k PREFIX_BITS=$pb canonical=$canon)"
  (cd "$REF" && K=$k PREFIX_BITS=$pb cargo +nightly build --release --examples > "$TMP/build_$name.log" 2>&1) || { echo "   build failed: see $TMP/build_$name.log"; fail=1; continue; }
  flag=""; [ "$canon" = 1 ] && flag="-c"
  "$REF/target/release/examples/cbl" build $flag "$TMP/$name.fa" -o "$TMP/$name.cbl"
  got=$(sha256sum "$TMP/$name.cbl" | cut -d' ' -f1)
  if [ "$got" = "$sha" ]; then echo "   OK  $got"; else echo "   MISMATCH reference $got, golden $sha"; fail=1; fi
done < "$TMP/cases.txt"
[ $fail = 0 ] && echo "all golden vectors match the reference: the serialized bytes are pinned" || { echo "NOT pinned"; exit 1; }
