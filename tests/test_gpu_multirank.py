"""The shipped multi-GPU path with MORE THAN ONE RANK in real RCCL, on the one GPU a test box has.

Each rank is a process of its own on cuda:0 with its own ``NCCL_HOSTID`` (``distributed.one_gpu_rank_env``): RCCL then
takes the ranks for different hosts, accepts the communicator and connects them through its socket transport on the
loopback interface.  The bytes travel over TCP instead of xGMI -- nothing here is a timing -- but every call is the
product's: ``zk_comm_init_file`` rendezvous, ``ncclCommInitRank`` with world > 1, the grouped ``ncclSend`` /
``ncclRecv`` schedule of ``zk_allgather_rows`` (and its ``ncclAllGather`` / ``ncclBroadcast`` forms), the event /
stream ordering against the kernels, ``zk_comm_allgather_host``, and on top of them the four sharded drivers with the
real kernels and the sharded consumers.  See tests/multirank_child.py for what a rank checks."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CHILD = os.path.join(ROOT, "tests", "multirank_child.py")


def _run_world(tmp_path, world, algo):
    from mtflearn_amd.distributed import one_gpu_rank_env
    id_file = str(tmp_path / "rccl_id")
    procs, logs = [], []
    for r in range(world):
        env = one_gpu_rank_env(r)
        env["ZK_COMM_ALGO"] = algo if algo != "auto" else ""
        log = open(tmp_path / f"rank{r}.log", "w")
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, CHILD, str(r), str(world), id_file, str(tmp_path)], env=env,
                                      stdout=log, stderr=subprocess.STDOUT))
    failed = []
    for r, p in enumerate(procs):
        try:
            rc = p.wait(timeout=420)
        except subprocess.TimeoutExpired:
            rc = "timeout"
            for q in procs:                                   # exact children only
                if q.poll() is None:
                    q.kill()
        if rc != 0:
            failed.append((r, rc))
    for log in logs:
        log.close()
    if failed:
        tails = "\n".join(f"--- rank {r} (exit {rc}) ---\n" + "\n".join(
            line for line in open(tmp_path / f"rank{r}.log").read().splitlines() if "NCCL WARN" not in line and line.strip())[-3000:]
            for r, rc in failed)
        pytest.fail(f"world {world}, ZK_COMM_ALGO={algo}:\n{tails}")
    assert all(os.path.exists(tmp_path / f"rank{r}.ok") for r in range(world))
    assert not os.path.exists(id_file)                        # rank 0 removed the rendezvous file after the join


@pytest.mark.parametrize("world,algo", [(2, "auto"), (3, "auto"), (4, "p2p"), (2, "bcast"), (3, "allgather")])
def test_real_rccl_ranks_on_one_gpu(tmp_path, world, algo):
    _run_world(tmp_path, world, algo)
