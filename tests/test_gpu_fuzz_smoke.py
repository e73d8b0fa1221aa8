"""A short, fixed slice of the differential fuzzer (tests/fuzz_parity.py) in the GPU suite: 60 cases of seed 401. The fuzzer
(random K / PREFIX_BITS / alphabets / operation sequences incl. load, merge, sorted batches, shard loads, bucket-batch export /
install, batches at high coverage of a short genome) caught the round-2 encode regression every shape-based test had missed (test_dirty_chunk_next_to_clean_ones_...).
The full fuzzer (hundreds of cases per seed) is run by hand on the GPU box."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fuzz_slice_seed_401():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the -m gpu tests must run on the MI355X box")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_parity.py"), "--cases", "60", "--seed", "401"], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, CBLX_FUZZ_DIAG="1"))
    assert r.returncode == 0 and "fuzz ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
