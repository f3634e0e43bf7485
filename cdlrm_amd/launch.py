"""Self-contained multi-rank launch: the counterpart of the reference's `mp.spawn(Run, nprocs=args.world_size)`
(main_no_ddp.py:638-643), so that `python -m cdlrm_amd.main_no_ddp ... --world-size=8` and `python bench.py --gpus 8`
work as typed, without `torchrun` on the command line.

One process per GPU: the calling process -- BEFORE its first HIP call -- starts `python -m torch.distributed.run
--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P <entry point> <argv>` as a CHILD, lets the ranks
write to its own stdout / stderr, and exits with the child's return code.  Never a re-exec of a process that touched the
GPU (on this pool an `exec` from a GPU-initialised process takes the machine down).

`check_world(n)` is the other half: once the process group is up, a rank refuses to run when torch.distributed's world
size is not the one the command line asked for -- an `--gpus 8` that silently measures one GPU cannot happen.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys

EMULATE_ENV = "CDLRM_BENCH_EMULATE"     # development: N ranks on ONE GPU, collectives over gloo (tests on a 1-GPU box)


def emulated() -> bool:
    return os.environ.get(EMULATE_ENV, "0") == "1"


def under_launcher() -> bool:
    """True in a rank process some launcher (torchrun, this module) started."""
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher_command(n: int, argv, *, script: str | None = None, module: str | None = None, port: int | None = None):
    assert (script is None) != (module is None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n)),
           "--master-addr", "127.0.0.1", "--master-port", str(port or free_port())]
    cmd += ["-m", module] if module else [script]
    return cmd + list(argv)


def spawn_ranks(n: int, argv, *, script: str | None = None, module: str | None = None, port: int | None = None,
                cwd: str | None = None) -> int:
    """Start the N rank processes as children and wait for them.  Returns their return code."""
    import torch
    if torch.cuda.is_initialized():
        raise RuntimeError("cdlrm_amd.launch.spawn_ranks: this process has already initialised the GPU; the ranks must "
                           "be started before the first HIP call")
    env = dict(os.environ)
    env.setdefault("GPU_MAX_HW_QUEUES", "4")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs between the ranks of this host
    # the CPUs this job may USE (cgroup quota / affinity), not the machine's: os.cpu_count() says 256 on a box that grants 16
    from .hostmem import cpu_share
    env.setdefault("OMP_NUM_THREADS", str(max(1, cpu_share() // max(1, n))))
    cmd = launcher_command(n, argv, script=script, module=module, port=port)
    sys.stdout.flush()
    sys.stderr.flush()
    return subprocess.call(cmd, env=env, cwd=cwd)


def check_world(requested: int) -> None:
    """Hard failure when the process group is not the size the command line asked for."""
    import torch.distributed as dist
    have = dist.get_world_size() if dist.is_initialized() else 1
    if have != int(requested):
        raise SystemExit("ERROR: %d ranks were requested but torch.distributed reports a world size of %d"
                         % (int(requested), have))
