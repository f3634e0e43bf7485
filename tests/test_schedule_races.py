"""Race check "the strong way" (tools/race_check.py): the same training steps from the same initial state under every SCHEDULE
the engine can run -- three-lane / two-lane / one-lane / Python-replayed / untaped launch sequences, events attached to launches or recorded
behind them, the deferred-update wait on the side stream or on the training queue -- must end on the same BITS: final loss,
every dense parameter, the cache rows' checksum, the tags, the running statistics.  Same kernels, same inputs, same stream
dependencies; only who issues which call, and when, differs -- so any difference is a missing dependency.  Run at the per-rank
batch of an 8-GPU run and at the c3 batch, each under BOTH take schedules (two aux regions: the next batch's take at the head of
the step; chained take: one aux region, take and sort behind the embedding update -- `gather_alone_min` picks by batch size in
production), across a window boundary, on c3's shapes with the tables capped at 2 M rows."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("batch,steps", [(1024, 400), (8192, 120)])
def test_every_schedule_ends_on_the_same_bits(batch, steps):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "race_check.py"), "--batch", str(batch), "--steps", str(steps)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import race_check
    n = len(race_check.VARIANTS) - 1           # every variant but the first (the reference for the others)
    assert n >= 7 and r.stdout.count("bit-identical") == n and "DIFFERS" not in r.stdout, r.stdout


def test_the_check_notices_a_step_that_does_not_wait_for_its_slice_sort():
    """Negative control of the variants with late slice sorts: when nothing waits for a look-ahead slice's sort (the takes'
    streams skip the slice event) and the sort arrives late, steps read unsorted lists and stale once-only flags -- the run must
    leave the reference's bits, i.e. the check above is not blind to this dependency."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "race_check.py"), "--batch", "8192", "--steps", "60", "--negative"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DIFFERS" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
