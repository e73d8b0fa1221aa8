"""bench.py's command line and accounting tables (no GPU): every --config resolves to a complete workload, the defaults are
the ones the contract names, and the per-stage algorithmic bytes add up to what DESIGN_HISTORY.md §4 states for cfg 2."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_every_config_resolves():
    for name, cfg in bench.CONFIGS.items():
        a = bench.parse_args(["--config", name])
        assert (a.k, a.prefix_bits, a.reads, a.read_len) == (cfg["k"], cfg["prefix_bits"], cfg["reads"], cfg["read_len"])
        assert a.kind in ("build", "merge")
        assert a.genome == cfg.get("genome", 0)
    assert bench.parse_args(["--config", "dup"]).genome == 40_000_000


def test_defaults_follow_the_contract():
    a = bench.parse_args([])
    assert (a.gpus, a.config, a.kind) == (1, "cfg2", "build")
    assert bench.parse_args(["--gpus", "8"]).config == "cfg3"  # BASELINE.json's multi-GPU configuration
    a = bench.parse_args(["--gpus", "2", "--steps", "3", "--warmup", "1"])
    assert (a.steps, a.warmup) == (3, 1)


def test_algorithmic_bytes_of_cfg2():
    alg = bench.stage_alg_bytes(31, 24, 150)
    assert alg["radix_scatter"] == 18 + 17 + 16  # pass A reads 9 + writes 8 (+1 side channel), B 8 + 8 + 1, C 8 + 8
    assert alg["encode"] == 150 / 120 + 9
    assert alg["bucket_medium"] == 16 and alg["radix_hist"] == 2.0
    kb, wb, hi, sfx, nbytes = bench.word_layout(31, 24)
    assert (kb, wb, hi, sfx, nbytes) == (62, 68, 1, 8, 6)
    assert bench.word_layout(59, 28)[1:] == (125, 8, 16, 13)


def test_no_stage_is_priced_for_work_it_does_not_do():
    # PREFIX_BITS > 24: the last scatter stores the bucket starts itself, the directory stage reads no records
    for k, pb, L in ((31, 28, 150), (59, 28, 250), (31, 24, 150), (25, 24, 150)):
        assert bench.stage_alg_bytes(k, pb, L)["directory"] == 0.0
    assert bench.stage_alg_bytes(31, 8, 150)["directory"] == 8  # no LSD pass at all: boundaries from a scan of the sorted records


def test_multi_gpu_defaults_take_the_native_bins_path():
    a = bench.parse_args(["--gpus", "8"])
    assert (a.transport, a.protocol, a.protocol_resolved, a.slices) == ("native", "auto", "bins", 3)
    # 2 - 4 ranks: one link per pair of GPUs bounds every protocol that ships words — the library picks "replicate" on 2 - 3 ranks (the reads cross as
    # bit planes), "sorted" on 4 (rehearsed: profiles/r06_wire_emulated.md)
    for n in (2, 3):
        b = bench.parse_args(["--gpus", str(n)])
        assert (b.protocol, b.protocol_resolved, b.slices) == ("auto", "replicate", 1)
    b = bench.parse_args(["--gpus", "4"])
    assert (b.protocol, b.protocol_resolved, b.slices) == ("auto", "sorted", 4)
    assert bench.parse_args(["--gpus", "4", "--protocol", "bins"]).protocol_resolved == "bins"
    assert bench.parse_args(["--gpus", "8", "--transport", "torch"]).protocol == "sorted"
    # the CPU leg stays bounded (about 20-30 s) whatever the configuration
    assert bench.parse_args(["--config", "cfg3"]).cpu_sample_reads == 125_000
    assert bench.parse_args(["--config", "cfg4"]).cpu_sample_reads == 60_000
    assert bench.parse_args([]).cpu_sample_reads == 1_000_000


def test_source_hash_is_stable_and_covers_csrc():
    h = bench.source_hash()
    assert len(h) == 16 and h == bench.source_hash()


def _child(code: str, stdout=None):
    import subprocess

    return subprocess.Popen([sys.executable, "-c", code], stdout=stdout)


def test_launcher_stops_the_other_ranks_when_one_fails(capsys):
    import subprocess
    import time

    t0 = time.monotonic()
    procs = [_child("import time,sys; print('partial'); sys.stdout.flush(); time.sleep(120)", stdout=subprocess.PIPE),
             _child("import sys; sys.exit(3)"), _child("import time; time.sleep(120)")]
    rc = bench.supervise_ranks(procs, 600.0)
    assert rc == 3 and time.monotonic() - t0 < 30
    assert all(p.poll() is not None for p in procs)  # nobody is left waiting in a collective
    assert "partial" in capsys.readouterr().out  # what rank 0 had printed is still relayed


def test_launcher_deadline_and_clean_exit(capsys):
    import subprocess
    import time

    t0 = time.monotonic()
    procs = [_child("import time; time.sleep(120)", stdout=subprocess.PIPE), _child("import time; time.sleep(120)")]
    assert bench.supervise_ranks(procs, 1.0) == 124 and time.monotonic() - t0 < 30
    assert all(p.poll() is not None for p in procs)
    # a big line from rank 0 (more than a pipe buffer) while another rank is still running must not deadlock
    procs = [_child("import sys; sys.stdout.write('x' * 300000 + '\\n{\"ok\": 1}\\n')", stdout=subprocess.PIPE), _child("import time; time.sleep(1)")]
    assert bench.supervise_ranks(procs, 60.0) == 0
    assert capsys.readouterr().out.rstrip().endswith('{"ok": 1}')
