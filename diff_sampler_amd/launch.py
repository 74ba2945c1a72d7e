"""One-process-per-GPU launching, self-checked (the reference's `torchrun --standalone --nproc_per_node=N`,
diff-solvers-main/launch.sh, and the env:// rendezvous of torch_utils/distributed.py:13-28).

``bench.py --gpus N`` and the CLIs use this so that the flag can never silently lie:

  * ``WORLD_SIZE`` unset and N > 1  -> re-exec the same command under ``python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port <free port>`` (one rank per GPU, RCCL/gloo rendezvous on the
    loopback address: the container hostname may not resolve);
  * ``WORLD_SIZE`` set (the driver's own ``torch.distributed.run`` launch) -> it must equal ``--gpus``;
  * after ``init_process_group`` the world size is taken from the communicator itself (an all-reduce of ones), and every
    rank must own a distinct device.

Anything inconsistent raises ``LaunchError`` (the callers exit non-zero).
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import List, Optional, Sequence, Tuple


class LaunchError(RuntimeError):
    pass


COMM_INFO = {}      # filled by init_group: backend, world size from the communicator, RCCL's version line (NCCL_DEBUG=VERSION), IPC mode


def ipc_default(env) -> None:
    """Multi-process GPU work on the deployment pool needs dmabuf IPC: its host driver has no legacy IPC, and without
    ``HSA_ENABLE_IPC_MODE_LEGACY=0`` RCCL / cross-process tensor sharing fail with ``hipIpcGetMemHandle: invalid argument`` (the pool's
    images export the variable; a launcher that builds its own environment must keep it).  The value already in the environment
    always wins; ``DS_KEEP_HSA_IPC_DEFAULT=1`` opts out of the default.  What a run used is reported in ``COMM_INFO``."""
    if env.get('DS_KEEP_HSA_IPC_DEFAULT') != '1':
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch_command(n: int, script: str, argv: Sequence[str], port: Optional[int] = None, python: Optional[str] = None) -> List[str]:
    """The exact command line that starts ``n`` ranks of ``script argv...`` on this node."""
    return [python or sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
            '--master-port', str(port or free_port()), script] + list(argv)


def env_world(env=None) -> Optional[Tuple[int, int, int]]:
    """(rank, world, local_rank) from a torchrun-style environment, or None when not launched by one."""
    env = os.environ if env is None else env
    if 'WORLD_SIZE' not in env:
        return None
    world = int(env['WORLD_SIZE'])
    rank = int(env.get('RANK', 0))
    local = int(env.get('LOCAL_RANK', rank))
    if not (0 <= rank < world):
        raise LaunchError(f'RANK={rank} outside WORLD_SIZE={world}')
    return rank, world, local


def resolve(gpus: int, script: str, argv: Sequence[str], env=None, spawn=None) -> Tuple[int, int, int]:
    """Decide what this process is.  Returns (rank, world, local_rank) for a worker (world == gpus, checked); when it is
    the un-launched parent of an N > 1 job it spawns the ranks, waits and exits with their status (never returns).
    ``spawn`` (tests) replaces ``subprocess.call``."""
    if gpus < 1:
        raise LaunchError(f'--gpus {gpus}: need at least one')
    ew = env_world(env)
    if ew is None:
        if gpus == 1:
            return 0, 1, 0
        cmd = launch_command(gpus, script, argv)
        child_env = dict(os.environ if env is None else env)
        ipc_default(child_env)
        rc = (spawn or subprocess.call)(cmd, env=child_env)
        raise SystemExit(rc)
    rank, world, local = ew
    if env is None and world > 1:
        ipc_default(os.environ)             # before the first HIP call of this rank (the HSA runtime reads it when it initialises)
    if world != gpus:
        raise LaunchError(f'--gpus {gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {gpus} (or drop the launcher and let '
                          f'`--gpus {gpus}` start the ranks itself)')
    return rank, world, local


def init_group(rank: int, world: int, local: int, backend: str, device=None):
    """env:// process group on the loopback address + the communicator's own head count.  Returns (dist, n_from_comm)."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    ver_file = None
    if backend == 'nccl':
        if torch.cuda.device_count() < world:
            raise LaunchError(f'{world} ranks need {world} GPUs on this node, {torch.cuda.device_count()} visible')
        # make the first real multi-GPU run diagnosable: RCCL's NCCL_DEBUG=VERSION line goes to a per-rank file (stdout carries the
        # one JSON line) and is reported back through COMM_INFO
        if 'NCCL_DEBUG' not in os.environ and 'NCCL_DEBUG_FILE' not in os.environ:
            import tempfile
            ver_file = os.path.join(tempfile.gettempdir(), f'ds_rccl_version_{os.getpid()}.log')
            os.environ['NCCL_DEBUG'], os.environ['NCCL_DEBUG_FILE'] = 'VERSION', ver_file
        if 'HSA_ENABLE_IPC_MODE_LEGACY' not in os.environ and rank == 0:
            print('launch: HSA_ENABLE_IPC_MODE_LEGACY is not set (DS_KEEP_HSA_IPC_DEFAULT=1); on hosts whose driver only supports dmabuf '
                  'IPC RCCL fails with "hipIpcGetMemHandle: invalid argument"', file=sys.stderr)
    dist.init_process_group(backend=backend, init_method='env://', rank=rank, world_size=world)
    ones = torch.ones(1, dtype=torch.float64, device=device if backend == 'nccl' else 'cpu')
    dist.all_reduce(ones)
    n = int(round(float(ones.item())))
    if n != world or dist.get_world_size() != world:
        raise LaunchError(f'communicator counts {n} ranks (get_world_size {dist.get_world_size()}), expected {world}')
    if backend == 'nccl':
        ids = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
        dist.all_gather(ids, torch.tensor([local], dtype=torch.int64, device=device))
        got = sorted(int(t.item()) for t in ids)
        if got != list(range(world)):
            raise LaunchError(f'ranks do not own distinct devices: LOCAL_RANKs {got}')
    COMM_INFO.clear()
    COMM_INFO.update(backend=('nccl (RCCL)' if backend == 'nccl' else backend), world=n,
                     HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'))
    if ver_file is not None:
        try:
            with open(ver_file) as f:
                lines = [l.strip() for l in f if 'version' in l.lower()]
            COMM_INFO['rccl_version_line'] = lines[0] if lines else None
            os.remove(ver_file)
        except OSError:
            COMM_INFO['rccl_version_line'] = None
    return dist, n
