"""The schedule of the shipped collective, checked without a GPU: ``zk_allgather_rows`` is a pure planner
(``zk_allgather_rows_plan``) plus an executor that hands the planner's list to RCCL in order.  Here every rank's
plan of a world is built on the CPU and the contract of include/zernike_hip.h is checked: sends and receives pair
up in issue order inside the same group, a rank receives exactly the other ranks' windows (once, disjoint), and
a simulated execution of all plans (FIFO channels per ordered rank pair, as RCCL and gloo match point-to-point
calls) leaves every rank with the complete array.  Worlds 2, 3, 8 (and others), 1 / 41 / 66 planes, whole blocks,
ragged tails, empty ranks, chunk windows, the three ZK_COMM_ALGO forms."""
import itertools

import numpy as np
import pytest

from mtflearn_amd import _native as N
from mtflearn_amd.distributed import _chunk_bounds, shard_bounds

SEND, RECV, ALLGATHER, BCAST = N.XFER_SEND, N.XFER_RECV, N.XFER_ALLGATHER, N.XFER_BCAST


def _windows(world, H, rpr, row_off, n_rows):
    """Rows [lo, hi) of every rank's block the window selects ((0, 0) when empty) -- restated here, not shared."""
    out = []
    for r in range(world):
        b0, b1 = r * rpr, min((r + 1) * rpr, H)
        lo, hi = b0 + row_off, min(b0 + row_off + n_rows, b1)
        out.append((lo, hi) if hi > lo else (0, 0))
    return out


def _plans(world, planes, H, W, rpr, row_off, n_rows, algo):
    return [N.allgather_rows_plan(r, world, planes, H, W, rpr, row_off, n_rows, algo) for r in range(world)]


def _check_contract(world, planes, H, W, rpr, row_off, n_rows, algo):
    plans = _plans(world, planes, H, W, rpr, row_off, n_rows, algo)
    wins = _windows(world, H, rpr, row_off, n_rows)
    for r, plan in enumerate(plans):
        groups = [g for _, _, g, _, _, _ in plan]
        assert groups == sorted(groups) and (not groups or groups[0] == 0)            # ascending from 0
        assert all(cnt > 0 and off >= 0 and off + cnt <= planes * H * W for _, _, _, _, off, cnt in plan)
        for g in set(groups):
            assert len({pl for _, _, gg, pl, _, _ in plan if gg == g}) <= 16         # ZK_COMM_PLANES_PER_GROUP
    ops = {op for plan in plans for op, *_ in plan}
    if ops <= {SEND, RECV}:
        # (1) pairing, in issue order, inside the same group
        for a, b in itertools.permutations(range(world), 2):
            sends = [(g, cnt) for op, peer, g, _, _, cnt in plans[a] if op == SEND and peer == b]
            recvs = [(g, cnt) for op, peer, g, _, _, cnt in plans[b] if op == RECV and peer == a]
            assert sends == recvs, (a, b)
        for r, plan in enumerate(plans):
            lo, hi = wins[r]
            # (2) a rank sends only its own window ...
            for op, peer, _, pl, off, cnt in plan:
                if op == SEND:
                    assert hi > lo and off == pl * H * W + lo * W and cnt == (hi - lo) * W and peer != r
            # (3) ... and receives exactly the other ranks' windows of every plane, once each
            got = sorted((off, cnt) for op, _, _, _, off, cnt in plan if op == RECV)
            want = sorted((pl * H * W + wins[o][0] * W, (wins[o][1] - wins[o][0]) * W)
                          for pl in range(planes) for o in range(world) if o != r and wins[o][1] > wins[o][0])
            assert got == want, r
    else:
        # collectives: the same entries in the same order on every rank (ALLGATHER differs in the own-block offset only)
        strip = lambda plan: [(op, peer, g, pl, cnt) + (() if op == ALLGATHER else (off,)) for op, peer, g, pl, off, cnt in plan]
        assert all(strip(p) == strip(plans[0]) for p in plans)
        for r, plan in enumerate(plans):
            for op, peer, _, _, off, cnt in plan:
                if op == ALLGATHER:
                    assert off == r * cnt and cnt * world == planes * H * W and peer == -1
    return plans


def _simulate(world, planes, H, W, plans, wins):
    """Execute every rank's plan on NumPy arrays: group by group, point-to-point messages through FIFO channels per
    ordered rank pair, collectives in list order.  Returns the arrays every rank ends with."""
    ref = np.arange(planes * H * W, dtype=np.float64) + 1.0
    arrays = []
    for r in range(world):
        a = np.full(planes * H * W, np.nan)
        lo, hi = wins[r]
        v = a.reshape(planes, H, W)
        v[:, lo:hi] = ref.reshape(planes, H, W)[:, lo:hi]
        arrays.append(a)
    n_groups = 1 + max([g for p in plans for _, _, g, _, _, _ in p], default=-1)
    for g in range(n_groups):
        chan = {}
        for r, plan in enumerate(plans):                       # all sends of the group are posted first (non-blocking)
            for op, peer, gg, _, off, cnt in plan:
                if gg == g and op == SEND:
                    chan.setdefault((r, peer), []).append(arrays[r][off:off + cnt].copy())
        for r, plan in enumerate(plans):
            for op, peer, gg, _, off, cnt in plan:
                if gg != g:
                    continue
                if op == RECV:
                    msg = chan[(peer, r)].pop(0)
                    assert msg.size == cnt
                    assert np.isnan(arrays[r][off:off + cnt]).all()          # nobody wrote here before: no overlap
                    arrays[r][off:off + cnt] = msg
                elif op == BCAST and r == peer:
                    for o in range(world):
                        if o != r:
                            arrays[o][off:off + cnt] = arrays[r][off:off + cnt]
                elif op == ALLGATHER:
                    base = off - r * cnt
                    for o in range(world):
                        arrays[o][base + r * cnt:base + (r + 1) * cnt] = arrays[r][off:off + cnt]
        assert all(not q for q in chan.values())                              # every message was received
    return ref, arrays


CASES = []
for world in (2, 3, 8):
    for planes in (1, 41, 66):
        for H in (world * 5, world * 5 - 1, world * 5 - (world - 1) * 5 + 1 if world > 2 else 7, 3):
            CASES.append((world, planes, H))


@pytest.mark.parametrize("algo", ["auto", "p2p", "allgather", "bcast"])
@pytest.mark.parametrize("world,planes,H", CASES)
def test_every_ranks_plan_reassembles_the_array(world, planes, H, algo):
    W = 3
    _, _, rpr = shard_bounds(H, 0, world)
    for c0, c1 in _chunk_bounds(rpr, 1) + _chunk_bounds(rpr, 3):
        plans = _check_contract(world, planes, H, W, rpr, c0, c1 - c0, algo)
        wins = _windows(world, H, rpr, c0, c1 - c0)
        ref, arrays = _simulate(world, planes, H, W, plans, wins)
        want = np.full((planes, H, W), np.nan)
        for lo, hi in wins:
            want[:, lo:hi] = ref.reshape(planes, H, W)[:, lo:hi]
        for r in range(world):
            np.testing.assert_array_equal(arrays[r].reshape(planes, H, W), want)


def test_form_selection_and_group_boundaries():
    # whole equal blocks of ONE plane -> one ncclAllGather, no group bracket needed
    plan = N.allgather_rows_plan(1, 4, 1, 32, 45, 8, 0, 8)
    assert plan == [(ALLGATHER, -1, 0, 0, 8 * 45, 8 * 45)]
    # a chunk window, a ragged tail or several planes -> point-to-point
    assert {op for op, *_ in N.allgather_rows_plan(1, 4, 1, 32, 45, 8, 0, 4)} == {SEND, RECV}
    assert {op for op, *_ in N.allgather_rows_plan(1, 4, 1, 31, 45, 8, 0, 8)} == {SEND, RECV}
    assert {op for op, *_ in N.allgather_rows_plan(1, 4, 2, 32, 45, 8, 0, 8)} == {SEND, RECV}
    # "allgather" forced on blocks that are not whole falls back to point-to-point
    assert N.allgather_rows_plan(0, 4, 1, 31, 45, 8, 0, 8, "allgather") == N.allgather_rows_plan(0, 4, 1, 31, 45, 8, 0, 8, "p2p")
    # 41 planes -> groups 0, 1, 2 of 16 + 16 + 9 planes; 7 peers x (send + receive) per plane at world 8
    plan = N.allgather_rows_plan(3, 8, 41, 64, 5, 8, 0, 8, "p2p")
    per_group = [sum(1 for e in plan if e[2] == g) for g in range(3)]
    assert per_group == [16 * 14, 16 * 14, 9 * 14] and len(plan) == 41 * 14
    # staggered peers: step s sends to rank + s and receives from rank - s
    first = [(op, peer) for op, peer, *_ in plan[:14]]
    assert first == [x for s in range(1, 8) for x in ((SEND, (3 + s) % 8), (RECV, (3 - s) % 8))]
    # world 1 and empty extents: nothing to do
    assert N.allgather_rows_plan(0, 1, 5, 10, 3, 10, 0, 10) == []
    assert N.allgather_rows_plan(0, 2, 0, 10, 3, 5, 0, 5) == [] and N.allgather_rows_plan(0, 2, 1, 10, 3, 5, 0, 0) == []
    # an empty rank (no rows of its own) only receives
    plan = N.allgather_rows_plan(7, 8, 2, 9, 4, 2, 0, 2, "p2p")                # rows 14.. do not exist
    assert plan and {op for op, *_ in plan} == {RECV}


def test_planner_rejects_bad_arguments():
    bad = [dict(rank=2, world=2), dict(rank=-1), dict(world=0), dict(rpr=3), dict(row_off=4, n_rows=2), dict(n_planes=-1),
           dict(algo=9)]
    base = dict(rank=0, world=2, n_planes=1, H=10, W=3, rpr=5, row_off=0, n_rows=5, algo=0)
    for change in bad:
        kw = dict(base, **change)
        with pytest.raises(RuntimeError, match="zk_allgather_rows_plan"):
            N.allgather_rows_plan(kw["rank"], kw["world"], kw["n_planes"], kw["H"], kw["W"], kw["rpr"], kw["row_off"],
                                  kw["n_rows"], kw["algo"])


def test_drivers_windows_cover_the_bench_configurations():
    """The exact (planes, H, W, rows_per_rank, window) sequences the sharded drivers issue for the BASELINE configs at
    world 8 (and 2, 4): every chunk's plans pair up and, executed in sequence, reassemble the result."""
    shapes = [("patch matrix", 1, 8 * 4068289, 45, 4), ("dense (45, 2048, 2048)", 45, 2048, 2048, 4),
              ("maps rot", 4, 4096, 4096, 4), ("maps abs", 36, 4096, 4096, 4), ("maps mirror", 1, 4096, 4096, 4),
              ("frames (64, 45*2048*2048)", 1, 64, 45 * 2048 * 2048, 8)]
    for world in (2, 4, 8):
        for name, planes, H, W, n_chunks in shapes:
            _, _, rpr = shard_bounds(H, 0, world)
            received = [[] for _ in range(world)]                              # row intervals of plane 0 a rank received
            for c0, c1 in _chunk_bounds(rpr, n_chunks):
                plans = _check_contract(world, planes, H, W, rpr, c0, c1 - c0, "auto")
                for r, plan in enumerate(plans):
                    for op, peer, _, pl, off, cnt in plan:
                        if op == RECV and pl == 0:
                            received[r].append((off // W, off // W + cnt // W))
                        elif op == ALLGATHER:
                            received[r] += [(o * rpr, (o + 1) * rpr) for o in range(world) if o != r]
            for r in range(world):
                lo, n, _ = shard_bounds(H, r, world)
                pos = 0                                                         # the intervals tile [0, H) minus the own block
                for a, b in sorted(received[r]):
                    if pos == lo:
                        pos = lo + n
                    assert a == pos and b > a, (name, world, r)
                    pos = b
                if pos == lo:
                    pos = lo + n
                assert pos == H, (name, world, r)
