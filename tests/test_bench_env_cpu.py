"""bench.py at N > 1 without a GPU: what environment its ranks get, and how the RCCL transport that ends up in the line
(`allgather.transport`) is read from librccl's own INFO output."""
import importlib.util
import os
import sys

from conftest import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


REHEARSAL_KEYS = ("NCCL_HOSTID", "NCCL_P2P_DISABLE", "NCCL_SHM_DISABLE", "NCCL_SOCKET_IFNAME", "NCCL_IB_DISABLE")


def test_real_node_ranks_get_no_host_id_and_no_transport_switch():
    """One GPU per rank (the driver's 8-GPU node): nothing may tell RCCL that the ranks sit on different hosts or take its
    peer-to-peer transport away -- the gather would silently run over sockets instead of xGMI."""
    b = _bench()
    base = {"PATH": os.environ.get("PATH", ""), "HOME": "/root"}
    assert not b.is_rehearsal(base)
    for world in (2, 8):
        for r in range(world):
            env = b.rank_env(base, r, world, 29500)
            assert not any(k in env for k in REHEARSAL_KEYS), env
            assert env["RANK"] == env["LOCAL_RANK"] == str(r) and env["WORLD_SIZE"] == str(world)
            assert env["MASTER_ADDR"] == "127.0.0.1" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # the in-worker branch (an outside launcher such as torch.distributed.run) keys on the same predicate
    assert not b.is_rehearsal({"ZK_BENCH_BACKEND": "rccl"})
    assert not b.is_rehearsal({"ZK_BENCH_ONE_DEVICE": "1", "ZK_BENCH_BACKEND": "gloo"})
    assert not b.is_rehearsal({"ZK_BENCH_ONE_DEVICE": "1", "ZK_BENCH_SAME_HOSTID": "1"})


def test_one_gpu_rehearsal_ranks_get_distinct_host_ids():
    b = _bench()
    base = {"ZK_BENCH_ONE_DEVICE": "1"}
    assert b.is_rehearsal(base)
    ids = {b.rank_env(base, r, 6, 29500)["NCCL_HOSTID"] for r in range(6)}
    assert len(ids) == 6
    env = b.rank_env(base, 3, 6, 29500)
    assert env["NCCL_P2P_DISABLE"] == "1" and env["NCCL_SOCKET_IFNAME"] == "lo"


def test_transport_is_read_from_rccl_info_lines():
    from mtflearn_amd import distributed as D
    node = """
box:101:140 [0] NCCL INFO Channel 00/0 : 0[0] -> 1[1] via P2P/IPC/read
box:101:140 [0] NCCL INFO Channel 01/0 : 0[0] -> 1[1] via P2P/IPC/read
box:101:140 [0] NCCL INFO Channel 00/1 : 0[0] -> 7[7] via P2P/IPC/read
box:101:140 [0] NCCL INFO Connected all rings
"""
    sock = """
box:77:90 [0] NCCL INFO Channel 00/0 : 1[0] -> 0[0] [receive] via NET/Socket/0
box:77:90 [0] NCCL INFO Channel 00/0 : 0[0] -> 1[0] [send] via NET/Socket/0
box:77:90 [0] NCCL INFO Channel 01 : 0[3000] -> 1[3000] via SHM/direct/direct
"""
    assert D.rccl_transport_summary(node) == {"P2P/IPC": 3}
    assert D.rccl_transport_summary(sock) == {"NET/Socket": 2, "SHM/direct": 1}
    assert D.rccl_transport_summary("nothing of the kind") == {}
    assert D.describe_transport([{"P2P/IPC": 3}, {"P2P/IPC": 5}]) == "P2P/IPC"
    assert D.describe_transport([{"NET/Socket": 4}] * 2, rehearsal=True) == "NET/Socket (rehearsal)"
    assert D.describe_transport([{"P2P/IPC": 8}, {"NET/Socket": 2, "P2P/IPC": 1}]) == "P2P/IPC + NET/Socket"
    assert D.describe_transport([{}, {}]).startswith("unknown")
    env = D.rccl_debug_env(3, base={})
    assert env["NCCL_DEBUG"] == "INFO" and "INIT" in env["NCCL_DEBUG_SUBSYS"] and env["NCCL_DEBUG_FILE"].endswith("rank3.log")
    assert D.rccl_debug_env(0, base={"NCCL_DEBUG": "INFO", "NCCL_DEBUG_FILE": "/x"}) == {}
    raised = D.rccl_debug_env(0, base={"NCCL_DEBUG": "VERSION", "NCCL_DEBUG_FILE": "/x"})      # below INFO: no channel lines
    assert raised == {"NCCL_DEBUG": "INFO", "NCCL_DEBUG_SUBSYS": "INIT,P2P,NET,SHM"}
