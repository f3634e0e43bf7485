"""`python bench.py --gpus N` and `python -m cdlrm_amd.main_no_ddp ... --world-size=N` start their N ranks themselves (the
reference's `mp.spawn(Run, nprocs=args.world_size)`, main_no_ddp.py:638-643) and can never report a world size they did not
run at.  CPU part: the launcher itself and the refusals; GPU part: both entry points typed WITHOUT torchrun, two ranks
emulated on the one GPU of the test box (CDLRM_BENCH_EMULATE=1: both ranks on device 0, collectives over gloo)."""
import json
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CLI_FLAGS = ["--arch-sparse-feature-size=16", "--arch-mlp-bot=13-32-16", "--arch-mlp-top=32-1",
             "--arch-embedding-size=3000-50-7-1200-40000", "--mini-batch-size=64", "--lookahead=4", "--cache-size=40",
             "--num-ways=4", "--loss-function=bce", "--round-targets=True", "--learning-rate=0.1", "--lr-embeds=0.3",
             "--print-freq=1", "--numpy-rand-seed=11", "--table-agg-freq=5", "--data-generation=criteo-synthetic",
             "--num-batches=10"]


def _env(**kw):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "GROUP_RANK", "LOCAL_WORLD_SIZE",
              "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env.update(kw)
    return env


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, out[-3000:]
    return json.loads(lines[0])


# ------------------------------------------------------------------------------------------------ CPU

def test_launcher_command_shape():
    from cdlrm_amd import launch
    cmd = launch.launcher_command(4, ["--gpus", "4"], script="/x/bench.py", port=4711)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "4711" and cmd[-3:] == ["/x/bench.py", "--gpus", "4"]
    cmd = launch.launcher_command(2, ["--world-size=2"], module="cdlrm_amd.main_no_ddp", port=1)
    assert cmd[-3:] == ["-m", "cdlrm_amd.main_no_ddp", "--world-size=2"]


def test_spawn_ranks_starts_one_child_per_rank(tmp_path):
    from cdlrm_amd import launch
    script = tmp_path / "rank.py"
    script.write_text("import os, sys\n"
                      "open(os.path.join(sys.argv[1], 'r' + os.environ['RANK']), 'w').write(os.environ['WORLD_SIZE'])\n"
                      "sys.exit(7 if os.environ['RANK'] == '1' and len(sys.argv) > 2 else 0)\n")
    assert launch.spawn_ranks(3, [str(tmp_path)], script=str(script)) == 0
    assert sorted(p.name for p in tmp_path.iterdir() if p.name.startswith("r") and p.name != "rank.py") == ["r0", "r1", "r2"]
    assert (tmp_path / "r2").read_text() == "3"
    assert launch.spawn_ranks(2, [str(tmp_path), "fail"], script=str(script)) != 0         # a failing rank fails the launch


def test_bench_refuses_a_world_it_was_not_launched_with():
    """`--gpus 2` inside a 1-rank launch (or any mismatch) exits before the first HIP call, without a JSON line."""
    for gpus, world in ((2, 1), (1, 2), (4, 2)):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--config", "c1"],
                           env=_env(RANK="0", LOCAL_RANK="0", WORLD_SIZE=str(world)), capture_output=True, text=True,
                           timeout=300)
        assert p.returncode != 0 and "does not match" in (p.stderr + p.stdout), (gpus, world, p.stderr[-500:])
        assert '"metric"' not in p.stdout


def test_cli_refuses_a_world_it_was_not_launched_with():
    p = subprocess.run([sys.executable, "-m", "cdlrm_amd.main_no_ddp"] + CLI_FLAGS + ["--world-size=4"],
                       env=_env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="2"), capture_output=True, text=True, timeout=300,
                       cwd=ROOT)
    assert p.returncode != 0 and "does not match" in (p.stderr + p.stdout)


def test_check_world():
    from cdlrm_amd import launch
    launch.check_world(1)
    with pytest.raises(SystemExit):
        launch.check_world(2)


# ------------------------------------------------------------------------------------------------ GPU

@pytest.mark.gpu
def test_bench_gpus_2_as_typed_runs_two_ranks():
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--config", "c1", "--steps", "8", "--warmup", "2",
                        "--prewarm-ms", "0"], env=_env(CDLRM_BENCH_EMULATE="1"), capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    line = _json_line(p.stdout)
    assert line["n_gpus"] == 2 and line["config"]["dist_world_size"] == 2 and line["config"]["parallelism"] == "dp2"
    assert line["config"]["dist_backend"] == "gloo" and line["config"]["local_batch"] * 2 == line["config"]["global_batch"]
    assert line["steps"] == 8 and 0.0 < line["config"]["final_loss"] < 2.0


@pytest.mark.gpu
def test_private_host_table_copies_equal_the_shared_mapping():
    """hostmem.make_host_tables at world > 1: ONE tmpfs mapping registered by every rank (rank 0 writes evictions back), or -- when
    the tmpfs is too small for the tables -- a private pinned copy per rank to which every rank applies the write-backs.  Same
    bits either way: 400 steps over 50 small windows of uniform indices (sets fill, rows are evicted and written back, later
    windows fetch them again), two ranks, final loss equal to the last bit."""
    losses = {}
    for mode in ("shared", "replicas"):
        p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--config", "c1", "--steps", "400", "--warmup", "2",
                            "--lookahead", "8", "--alpha", "0", "--prewarm-ms", "0"],
                           env=_env(CDLRM_BENCH_EMULATE="1", CDLRM_HOST_TABLES=mode), capture_output=True, text=True,
                           timeout=600, cwd=ROOT)
        assert p.returncode == 0, p.stderr[-3000:]
        line = _json_line(p.stdout)
        assert line["n_gpus"] == 2 and line["config"]["refills_in_timed_region"]["window_commits"] >= 40
        losses[mode] = line["config"]["final_loss"]
    assert losses["shared"] == losses["replicas"], losses


@pytest.mark.gpu
def test_cli_world_size_2_as_typed_equals_the_torchrun_launch():
    """The README command shape (`python main_no_ddp.py ... --world-size=N`) works as typed and prints what the same ranks
    print under an explicit torchrun."""
    from cdlrm_amd import launch
    outs = []
    for cmd in ([sys.executable, "-m", "cdlrm_amd.main_no_ddp"] + CLI_FLAGS + ["--world-size=2"],
                launch.launcher_command(2, CLI_FLAGS + ["--world-size=2"], module="cdlrm_amd.main_no_ddp")):
        p = subprocess.run(cmd, env=_env(CDLRM_BENCH_EMULATE="1"), capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert p.returncode == 0, p.stderr[-3000:]
        losses = [float(x) for x in re.findall(r"Loss = ([0-9.eE+-]+),", p.stdout)]
        assert len(losses) == 9 and all(0.0 < x < 2.0 for x in losses), p.stdout[-2000:]
        outs.append(losses)
    assert outs[0] == outs[1]


def test_cli_without_a_gpu_says_there_is_no_cpu_fallback():
    """no CPU product path -- and the CLI says so instead of dying in a torch traceback."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the CLI would train")
    p = subprocess.run([sys.executable, "-m", "cdlrm_amd.main_no_ddp"] + CLI_FLAGS + ["--world-size=1"], env=_env(),
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode != 0 and "no CPU fallback" in p.stderr and "Traceback" not in p.stderr, p.stderr[-800:]
    p = subprocess.run([sys.executable, "bench.py", "--config", "c1", "--steps", "2", "--warmup", "1"], env=_env(),
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode != 0 and "no CPU fallback" in p.stderr and "Traceback" not in p.stderr, p.stderr[-800:]


@pytest.mark.gpu
def test_cli_world_size_2_tests_on_rank_0_inside_the_training_loop(tmp_path):
    """`--test-freq` at world > 1: the test loop runs on rank 0 only (main_no_ddp.py:478-494) while rows of a table-agg merge may
    still be travelling in deadline order -- every rank drains them (collectives) before rank 0 tests alone; neither rank may
    hang or issue an exchange the other does not.  Day files in the reference's format, two ranks emulated on the one GPU."""
    import numpy as np
    rng = np.random.RandomState(3)
    counts = np.array([900, 40, 7, 300, 1500])
    sizes = [64 * 9 + 9, 64 * 8 + 30, 64 * 2]
    for day, n in enumerate(sizes):
        np.savez(os.path.join(tmp_path, "day_%d_reordered.npz" % day), X_int=rng.randint(0, 500, size=(n, 13)).astype(np.int32),
                 X_cat=np.stack([rng.randint(0, c, size=n) for c in counts], axis=1).astype(np.int32),
                 y=rng.randint(0, 2, size=n).astype(np.int32))
    np.savez(os.path.join(tmp_path, "day_day_count.npz"), total_per_file=np.array(sizes))
    np.savez(os.path.join(tmp_path, "day_fea_count.npz"), counts=counts)
    flags = [f for f in CLI_FLAGS if not f.startswith(("--arch-embedding-size", "--data-generation", "--num-batches",
                                                       "--table-agg-freq", "--lookahead"))]
    flags += ["--data-generation=dataset", "--raw-data-file=" + os.path.join(tmp_path, "day"), "--lookahead=8",
              "--table-agg-freq=3", "--test-freq=4", "--world-size=2"]
    p = subprocess.run([sys.executable, "-m", "cdlrm_amd.main_no_ddp"] + flags, env=_env(CDLRM_BENCH_EMULATE="1"),
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    n_train = (sizes[0] + sizes[1]) // 64
    assert p.stdout.count("Testing at") == (n_train - 1) // 4 + (0 if (n_train - 1) % 4 == 0 else 1), p.stdout[-2000:]
    assert p.stdout.count("Test accuracy = ") == p.stdout.count("Testing at")
    losses = [float(x) for x in re.findall(r"Loss = ([0-9.eE+-]+),", p.stdout)]
    assert len(losses) == n_train - 1 and all(0.0 < x < 10.0 for x in losses) and losses[-1] < 1.0, losses
