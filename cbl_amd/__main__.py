"""`python -m cbl_amd <command>` — the build / insert / merge / count / query / list subcommands of the reference CLI
(/root/reference/examples/cbl.rs:147-167,230-249,270-279,168-229) on the MI355X path.

K and PREFIX_BITS are compile-time constants of the reference (env K / PREFIX_BITS at cargo build time, build.rs:9-56);
here they are flags with the same defaults (K=25, PREFIX_BITS=24). Index files are interchangeable with the reference's.
"""
import argparse
import sys

from . import CBL


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m cbl_amd")
    ap.add_argument("-k", type=int, default=25, help="k-mer size (odd, <= 59); the reference bakes it in at build time")
    ap.add_argument("--prefix-bits", type=int, default=24)
    ap.add_argument("--device", type=int, default=-1)
    sub = ap.add_subparsers(dest="cmd", required=True)
    b = sub.add_parser("build", help="Build an index containing the k-mers of a FASTA/Q file")
    b.add_argument("input")
    b.add_argument("-o", "--output")
    b.add_argument("-c", "--canonical", action="store_true")
    i = sub.add_parser("insert", help="Add the k-mers of a FASTA/Q file to an index")
    i.add_argument("index")
    i.add_argument("input")
    i.add_argument("-o", "--output")
    m = sub.add_parser("merge", help="Compute the union of two indexes")
    m.add_argument("first_index")
    m.add_argument("second_index")
    m.add_argument("-o", "--output")
    c = sub.add_parser("count", help="Count the k-mers contained in an index")
    c.add_argument("index")
    q = sub.add_parser("query", help="Query an index for every k-mer contained in a FASTA/Q file")
    q.add_argument("index")
    q.add_argument("input")
    ls = sub.add_parser("list", help="List the k-mers contained in an index")
    ls.add_argument("index")
    ls.add_argument("-o", "--output")
    a = ap.parse_args(argv)

    if a.cmd == "build":
        cbl = CBL(a.k, a.prefix_bits, canonical=a.canonical, device=a.device)
        print(f"Building the index of {'canonical ' if a.canonical else ''}{a.k}-mers contained in {a.input}", file=sys.stderr)
        cbl.insert_fastx_file(a.input)
        if a.output:
            print(f"Writing the index to {a.output}", file=sys.stderr)
            cbl.save_to_file(a.output)
    elif a.cmd == "insert":
        print(f"Reading the index stored in {a.index}", file=sys.stderr)
        cbl = CBL.load_from_file(a.index, a.k, a.prefix_bits, device=a.device)
        print(f"Adding the {'canonical ' if cbl.is_canonical() else ''}{a.k}-mers contained in {a.input} to the index", file=sys.stderr)
        cbl.insert_fastx_file(a.input)
        if a.output:
            print(f"Writing the index to {a.output}", file=sys.stderr)
            cbl.save_to_file(a.output)
    elif a.cmd == "merge":
        cbl = CBL.load_from_file(a.first_index, a.k, a.prefix_bits, device=a.device)
        cbl2 = CBL.load_from_file(a.second_index, a.k, a.prefix_bits, device=a.device)
        cbl |= cbl2
        if a.output:
            print(f"Writing the index to {a.output}", file=sys.stderr)
            cbl.save_to_file(a.output)
    elif a.cmd == "count":
        cbl = CBL.load_from_file(a.index, a.k, a.prefix_bits, device=a.device)
        print(f"It contains {cbl.count()} {a.k}-mers", file=sys.stderr)
        print(cbl.count())
    elif a.cmd == "query":  # examples/cbl.rs:205-228
        cbl = CBL.load_from_file(a.index, a.k, a.prefix_bits, device=a.device)
        print(f"Querying the {'canonical ' if cbl.is_canonical() else ''}{a.k}-mers contained in {a.input}", file=sys.stderr)
        _, total, positive = cbl.query_fastx_file(a.input)
        print(f"# queries: {total}", file=sys.stderr)
        print(f"# positive queries: {positive} ({positive * 100 / total if total else float('nan'):.2f}%)", file=sys.stderr)
        print(total, positive)
    elif a.cmd == "list":  # examples/cbl.rs:177-203: one k-mer per line, IntKmer::to_nucs (first base most significant)
        import numpy as np

        cbl = CBL.load_from_file(a.index, a.k, a.prefix_bits, device=a.device)
        print(f"Listing {'canonical ' if cbl.is_canonical() else ''}{a.k}-mers contained in {a.index}", file=sys.stderr)
        lo, hi = cbl.kmers_np()
        nuc = np.frombuffer(b"ACTG", dtype=np.uint8)  # src/kmer.rs:26-27
        out = open(a.output, "wb") if a.output else sys.stdout.buffer
        for c0 in range(0, len(lo), 1 << 20):
            l = lo[c0 : c0 + (1 << 20)]
            h = hi[c0 : c0 + (1 << 20)] if hi is not None else None
            lines = np.empty((len(l), a.k + 1), dtype=np.uint8)
            lines[:, a.k] = ord("\n")
            for j in range(a.k):  # base j sits 2 * (k - 1 - j) bits up
                sh = 2 * (a.k - 1 - j)
                code = (l >> np.uint64(sh)) if sh < 64 else (h >> np.uint64(sh - 64))
                lines[:, j] = nuc[(code & np.uint64(3)).astype(np.intp)]
            out.write(lines.tobytes())
        if a.output:
            out.close()


if __name__ == "__main__":
    main()
