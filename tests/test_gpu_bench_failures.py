"""bench.py at N > 1 must never end as "exit 0 with a plausible line" when something went wrong (VERDICT r2 / ADVICE r2):
a phase that overruns its limit, a communicator that cannot be created and a stopped peer all end NON-ZERO, rank 0 still
prints one JSON line (with `value` null and an "unmeasured" reason when the timed region was not completed), and stderr says
where the rank was.  Two ranks on the one GPU of the box (ZK_BENCH_ONE_DEVICE)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
BENCH = os.path.join(ROOT, "bench.py")


def _run(extra_env, *flags, timeout=400):
    env = dict(os.environ, ZK_BENCH_ONE_DEVICE="1", **extra_env)
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--frame", "256", "--no-cpu-baseline", *flags], env=env,
                       capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr


def test_a_phase_that_overruns_ends_nonzero_with_a_null_value():
    # gloo rehearsal backend (GPU tensors staged through the host): 100000 steps cannot finish within the 8-second limit
    rc, line, err = _run({"ZK_BENCH_BACKEND": "gloo", "ZK_BENCH_PHASE_TIMEOUT": "8"}, "--steps", "100000", "--warmup", "1",
                         "--only-timed-loop")
    assert rc != 0, err[-2000:]
    assert line is not None and line["value"] is None and "no progress" in line["unmeasured"] and "timed loop" in line["unmeasured"]
    assert line["n_gpus"] == 2 and line["metric"].startswith("patches/s")
    assert "stacks of every thread follow" in err and "File " in err                 # faulthandler's dump


def test_no_stand_in_when_the_product_communicator_is_unavailable():
    # both ranks on device 0 under the SAME host id: RCCL refuses the communicator ("Duplicate GPU detected")
    rc, line, err = _run({"ZK_BENCH_SAME_HOSTID": "1"}, "--steps", "2", "--warmup", "1", "--only-timed-loop")
    assert rc != 0, err[-2000:]
    assert line is not None and line["value"] is None and "product collective unavailable" in line["unmeasured"]
    assert "falling back" not in err


def test_two_rccl_ranks_deliver_a_verified_line():
    # the success path of the same harness: real RCCL between two ranks (a host id per rank), every section verified
    rc, line, err = _run({"ZK_BENCH_FRAMES_PER_RANK": "2", "ZK_BENCH_MAPS_FRAME": "512"}, "--steps", "3", "--warmup", "1")
    assert rc == 0, err[-3000:]
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["config"]["allgather_in_step"]
    assert line["allgather"]["verified"] and line["allgather"]["own_block_equals_recomputation"]
    assert line["allgather"]["backend"].startswith("rccl")
    assert line["multi_frame"]["verified"] and line["sharded_maps"]["verified"]
    assert line["clustering"]["ranks_agree"]
