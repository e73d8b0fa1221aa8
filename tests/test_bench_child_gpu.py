"""bench.py run the way the driver runs it — a child process, the launcher's own rank children — on one GPU: `--gpus 2 --shared-gpu`
is the dry run of the N-rank line (ranks share GPU 0, exchange staged through gloo), `--cpu-full` the full-size parity leg at a size
the oracle finishes in seconds. First contact of `python bench.py --gpus N` must not be the 8-GPU box."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("config", ["cfg3", "cfg2", "merge"])
def test_two_rank_line_through_the_launcher(config):
    out = _run("--gpus", "2", "--shared-gpu", "--config", config, "--reads", "20000", "--steps", "1", "--warmup", "0",
               "--cpu-sample-reads", "2000", "--launch-timeout", "600")
    assert out["n_gpus"] == 2 and out["steps"] == 1 and out["scaling"] == "weak" and out["value"] > 0
    assert out["roofline"] and out["roofline"]["bound"] == "hbm" and out["roofline"]["kernels"]
    assert out["exchange"] and out["exchange"]["world_size"] == 2
    if config == "merge":
        assert out["merge"]["words_union"] <= out["merge"]["words_self"] + out["merge"]["words_other"]
    else:
        assert out["cpu_baseline"] and out["cpu_baseline"]["cores"] == 1 and out["cpu_baseline"]["value"] > 0
        k = 31
        assert out["distinct_kmers_in_index"] <= 2 * 20000 * (150 - k + 1)
        assert all(s > 0 for s in out["exchange"]["sent_bytes_per_rank_step"])
        # the line carries its own denominator: the same configuration on one GPU (rank 0's share, built directly after the timed steps) ...
        one = out["one_gpu_same_config"]
        assert one["value"] > 0 and one["ms_per_step"] > 0 and abs(out["scaling_vs_one_gpu_same_config"] - out["value"] / one["value"]) < 0.01
        # ... and says what crossed the links: the library's choice at two ranks is "replicate" (one link per pair of GPUs bounds every protocol
        # that ships words: the reads cross as bit planes instead)
        assert out["exchange"]["protocol_asked"] == "auto" and out["exchange"]["protocol"] == "replicate"
        # ... and carries its own correctness verdict: the sharded index against every rank's word stream (count, set checksum, structure, every
        # inserted k-mer found in exactly one share), and the CPU leg's sample rebuilt on the GPU with the oracle's bytes
        sc = out["set_check"]
        assert sc["ok"] is True and sc["validate_violations"] == 0 and sc["members_ok"] is True and sc["members_found"] == sc["kmers_inserted"] == 2 * 20000 * (150 - k + 1)
        assert sc["count"] == out["distinct_kmers_in_index"] and sc["repeats_implied"] == sc["kmers_inserted"] - sc["count"] >= 0
        assert sc["repeats_implied"] != 0 or (sc["checksums_equal"] is True and sc["checksum_index"] == sc["checksum_word_streams"])
        ps = out["parity_sample"]
        assert ps["equal"] is True and ps["sha256"] == ps["sha256_oracle"] and ps["bytes"] == ps["bytes_oracle"] > 0 and ps["reads"] == 2000 and ps["first_difference_at"] is None
    # an algorithmic rate above the HBM peak is a bookkeeping error (bench.py asserts it too)
    assert all(r["frac"] is None or r["frac"] <= 1.0 for r in out["roofline"]["kernels"])


def test_default_line_carries_parity_sample():
    """The driver's N = 1 line: the CPU leg's sample is built once more on the GPU after the timed region and byte-compared with the oracle's index of it."""
    out = _run("--config", "cfg2", "--reads", "40000", "--steps", "1", "--warmup", "0", "--cpu-sample-reads", "5000", "--no-h2d")
    ps = out["parity_sample"]
    assert ps["equal"] is True and ps["reads"] == 5000 and ps["sha256"] == ps["sha256_oracle"] and ps["bytes"] == ps["bytes_oracle"] > 0
    assert ps["distinct_kmers"] == ps["distinct_kmers_oracle"] <= 5000 * 120
    assert "set_check" not in out and "roofline_error" not in out


def test_launcher_reports_a_failing_rank():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    # K = 32 is not a supported word layout: every rank fails after the rendezvous; the launcher must come back non-zero, promptly
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--shared-gpu", "--reads", "2000", "--k", "32", "--steps", "1",
                        "--warmup", "0", "--no-cpu-baseline", "--launch-timeout", "300"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_full_size_parity_leg_at_a_small_size():
    out = _run("--config", "cfg2", "--reads", "30000", "--steps", "1", "--warmup", "0", "--cpu-full", "--no-h2d")
    p = out["parity_full_size"]
    assert p["equal"] is True and p["sha256"] == p["sha256_oracle"] and p["bytes"] == p["bytes_oracle"] > 0
    assert p["distinct_kmers"] == p["distinct_kmers_oracle"] == out["distinct_kmers_in_index"]
    assert p["first_difference_at"] is None


def test_full_size_parity_leg_of_the_merge_at_a_small_size():
    """`--config merge --cpu-full`: the oracle builds both operands, merges them, and BOTH resulting indexes are byte-compared (the merged one
    and other, whose Vec buckets the merge sorted); the CPU baseline of that line is the oracle's merge."""
    out = _run("--config", "merge", "--reads", "30000", "--steps", "2", "--warmup", "1", "--cpu-full", "--no-h2d")
    p = out["parity_full_size"]
    assert p["equal"] is True and p["sha256"] == p["sha256_oracle"] and p["bytes"] == p["bytes_oracle"] > 0
    o = p["other_after_merge"]
    assert o["equal"] is True and o["sha256"] == o["sha256_oracle"] and 0 < o["bytes"] < p["bytes"]
    assert p["distinct_kmers"] == p["distinct_kmers_oracle"] == out["merge"]["words_union"]
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["value"] > 0 and "merge" in out["cpu_baseline"]["sample"]
