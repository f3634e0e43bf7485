"""`bench.py --gpus N --plan-only`: what every rank of an N-GPU run will ask the node for -- host / pinned / HBM bytes, threads,
streams, hardware queues, port -- printed WITHOUT a HIP call or a process group, so that the first run on a real 8-GPU node can
be read against it (first-trace checklist).  bench.build_workload takes its thread counts from the same
function (bench.rank_resources), so the plan and the run cannot disagree.  CPU only."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    return env


@pytest.mark.parametrize("n", [1, 2, 4, 8])
def test_plan_only_prints_one_record_per_rank_and_touches_no_gpu(n):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "20", "--warmup", "5",
                        "--plan-only"], env=_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    recs = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert [r["rank"] for r in recs] == list(range(n)) and '"metric"' not in p.stdout
    r0 = recs[0]
    assert r0["world"] == n and r0["local_batch"] * n == r0["global_batch"] == 8192 and r0["lookahead"] == 3000
    assert r0["host_tables_bytes_one_mapping_per_node"] == 187767399 * 128 * 4
    assert n == 1 or r0["host_tables_mode"] in ("shared", "replicas")
    assert r0["gpu_max_hw_queues"] == 4 and len({r["master_port"] for r in recs}) == 1
    # shared mapping / one rank: rank 0 writes the evictions back; private replicas: every rank writes to its own copy
    replicas = n > 1 and r0["host_tables_mode"] == "replicas"
    assert [r["writes_evictions_back"] for r in recs] == ([True] * n if replicas else [True] + [False] * (n - 1))
    assert r0["hbm_bytes_per_rank"]["cache_rows"] > 10e9 and r0["hbm_bytes_per_rank"]["sum_without_growth"] < 200e9
    assert (r0["projection"] is not None) and r0["projection"]["projected_scaling"] >= 1.0
    assert len(r0["hip_streams"]) == (4 if n == 1 else 6)


def test_rank_resources_follow_the_cpu_share():
    import bench
    roomy = bench.rank_resources("c3", 8, cpus=128, mem_limit=0, shm_free=1 << 40)
    assert roomy["cpus_per_rank"] == 16 and roomy["tape_lanes"] == 3 and roomy["plan_gather_threads"] == 13 and not roomy["notes"]
    tight = bench.rank_resources("c3", 8, cpus=16, mem_limit=0, shm_free=1 << 40)
    assert tight["cpus_per_rank"] == 2 and tight["tape_lanes"] == 1 and tight["plan_gather_threads"] == 1 and len(tight["notes"]) == 2
    one = bench.rank_resources("c3", 1, cpus=16, mem_limit=0)
    assert one["tape_lanes"] == 1 and one["plan_gather_threads"] == 13      # (a local batch of 8192 replays in one lane)
    assert bench.rank_resources("c3", 8, cpus=128, mem_limit=0, shm_free=1 << 40)["omp_num_threads"] == 16


def test_rank_resources_refuse_what_cannot_fit():
    import bench
    assert bench.rank_resources("c3", 8, cpus=128, mem_limit=64 << 30)["refused"], "96 GB of host tables in a 64 GB job"
    assert not bench.rank_resources("c3", 8, cpus=128, mem_limit=512 << 30, shm_free=1 << 40)["refused"]
    small = bench.rank_resources("c3", 8, cpus=128, mem_limit=0, shm_free=64 << 30)
    assert small["host_tables_mode"] == "replicas" and not small["refused"], "96 GB of tables, 64 GB of /dev/shm: private copies"
    assert bench.rank_resources("c3", 8, cpus=128, mem_limit=400 << 30, shm_free=64 << 30)["refused"], "8 x 96 GB in a 400 GB job"
    assert bench.rank_resources("c3", 8, cpus=128, mem_limit=0, shm_free=1 << 40)["host_tables_mode"] == "shared"
    assert any("divide" in r for r in bench.rank_resources("c3", 3, cpus=128, mem_limit=0, shm_free=1 << 40)["refused"])
    huge = bench.rank_resources("c4", 1, cpus=16, mem_limit=300 << 30)
    assert huge["refused"], "c4 uncapped pins 192 GB"
    assert not bench.rank_resources("c4", 1, cpus=16, mem_limit=300 << 30, max_ind_range=2000000)["refused"]


def test_host_tables_mode_is_one_decision_for_plan_and_run(monkeypatch):
    """hostmem.host_tables_mode is what BOTH make_host_tables (the run) and bench.py --plan-only (the plan) call: the tmpfs
    decides shared / replicas, the job's memory limit refuses replicas that do not fit (with the staging counted), the
    environment forces either."""
    from cdlrm_amd import hostmem
    GB = 1 << 30
    monkeypatch.delenv("CDLRM_HOST_TABLES", raising=False)
    assert hostmem.host_tables_mode(96 * GB, 1) == ("private", None)
    assert hostmem.host_tables_mode(96 * GB, 8, shm_free=200 * GB, limit=None) == ("shared", None)
    mode, err = hostmem.host_tables_mode(96 * GB, 8, shm_free=32 * GB, limit=2000 * GB)
    assert mode == "replicas" and err is None
    mode, err = hostmem.host_tables_mode(96 * GB, 8, shm_free=32 * GB, limit=800 * GB)
    assert mode == "replicas" and "exceed" in err
    mode, err = hostmem.host_tables_mode(96 * GB, 2, staging_bytes=30 * GB, shm_free=1 * GB, limit=int(2 * 126 * GB / 0.85) - GB)
    assert mode == "replicas" and err is not None          # the staging tips it over
    monkeypatch.setenv("CDLRM_HOST_TABLES", "shared")
    mode, err = hostmem.host_tables_mode(96 * GB, 8, shm_free=32 * GB, limit=None)
    assert mode == "shared" and "do not fit" in err
    monkeypatch.setenv("CDLRM_HOST_TABLES", "bogus")
    with pytest.raises(ValueError):
        hostmem.host_tables_mode(96 * GB, 8, shm_free=32 * GB, limit=None)


def test_work_skipping_debug_switches_are_refused_by_the_shipped_library():
    """cdlrm_debug_set(6, 1 | 2) -- no embedding update / no slot sort, timing experiments -- exist in -DCDLRM_DEV builds only:
    a bench line cannot have been produced by a library that skipped work."""
    from cdlrm_amd import _lib
    r = _lib.raw()
    assert r.cdlrm_debug_set(6, 1) != 0 and r.cdlrm_debug_set(6, 2) != 0 and r.cdlrm_debug_set(6, 35) != 0
    assert b"CDLRM_DEV" in r.cdlrm_last_error()
    assert r.cdlrm_debug_set(6, 32) == 0 and r.cdlrm_debug_set(6, 0) == 0
