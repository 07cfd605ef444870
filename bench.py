#!/usr/bin/env python3
"""bench.py -- Zernike-moment hot path on MI355X (contract: the task brief; numbers explained in DESIGN.md section 5).

A "step" is one pass of the batch-of-patches hot path (``ZPs.transform`` on a 3-D batch, reference
``mtflearn/features/_zps.py:146-157``) over every dense 32-px sliding window of one synthetic
2048 x 2048 STEM-like frame per GPU (BASELINE.json configs[1]: 4 068 289 patches, n_max = 8),
float32 patches resident in HBM, float64 moments out.

``python bench.py --gpus N`` starts its own N ranks (one process per GPU; the parent never touches the GPU);
under ``python -m torch.distributed.run`` the ranks it is given are used as they are.  With N > 1 every rank
owns its own frame (weak scaling, no data-path collective) and a step INCLUDES the one exchange north_star
names: the all-gather that reassembles the (N x 4 068 289, 45) moment matrix on every rank, issued chunk by
chunk on RCCL (``zk_allgather_rows`` inside libzernike_hip.so -- no torch.distributed) while the next
chunk's kernel runs.  `value` is that whole-job throughput; the sharded-kernel-only figure is reported
beside it (``kernel_only_patches_per_s``, ``allgather``).

Rank 0 prints ONE JSON line.  Besides the contract keys it carries
  roofline       -- the batch kernel: algorithmic bytes / HIP-event kernel time vs the 8 TB/s HBM peak
  cpu_baseline   -- the oracle's restatement of the reference CPU path (same NumPy call) on this host
  north_star_4096, config2_batch, config2_dense, config3_per_gpu, dense_frame, symmetry_pipeline -- the other
                    BASELINE configs as far as one GPU carries them, each with its own roofline object (N = 1)
  host_api       -- NumPy in / NumPy out through ZPs.transform (PCIe-inclusive; never `value`)
  clustering     -- the k-means consumer on the moment matrix where the timed kernel left it (N > 1: sharded, labels gathered)
  multi_frame, sharded_maps -- configs[3] / configs[4] on N > 1 ranks
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "motif-learn_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frame", type=int, default=2048, help="frame side (configs[1]: 2048)")
    ap.add_argument("--size", type=int, default=32)
    ap.add_argument("--n-max", type=int, default=8)
    ap.add_argument("--gather-chunks", type=int, default=4,
                    help="N>1: chunks a rank's block is cut into (kernel of chunk c+1 overlaps the transfer of chunk c)")
    ap.add_argument("--kernel-only-value", action="store_true",
                    help="N>1: take `value` from the loop WITHOUT the all-gather (default: with it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dense", action="store_true", help="skip the dense-frame side measurement")
    ap.add_argument("--no-maps", action="store_true", help="skip the configs[4] symmetry-map measurement")
    ap.add_argument("--no-4096", action="store_true",
                    help="skip the north star's own size (it launches the same kernel as the timed loop: use this "
                         "flag under rocprofv3 --stats so that the average is the timed loop's)")
    ap.add_argument("--no-config2", action="store_true", help="skip configs[2] (4096^2, 64-px, n_max 12)")
    ap.add_argument("--no-high-order", action="store_true", help="skip the matrix-core plain sum at (40 px, n_max 20)")
    ap.add_argument("--no-host-api", action="store_true", help="skip the NumPy-in / NumPy-out measurement")
    ap.add_argument("--no-multi-frame", action="store_true", help="N>1: skip configs[3] (8 frames per rank)")
    ap.add_argument("--only-timed-loop", action="store_true", help="skip every side measurement (profiling runs)")
    ap.add_argument("--no-clustering", action="store_true", help="skip the k-means consumer on the resident moment matrix")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------
# self-launch: one child process per GPU, started before anything in this process touches the GPU
# ---------------------------------------------------------------------------------------------------------
def is_rehearsal(env):
    """N RCCL ranks on ONE GPU (ZK_BENCH_ONE_DEVICE=1, a test aid): the only case in which a rank is given an NCCL_HOSTID and
    RCCL's peer-to-peer / shared-memory transports are switched off.  On a real node (one GPU per rank) nothing of the
    kind is set: RCCL picks its transports itself, and `allgather.transport` in the line says which."""
    return bool(env.get("ZK_BENCH_ONE_DEVICE")) and env.get("ZK_BENCH_BACKEND", "rccl") == "rccl" and not env.get("ZK_BENCH_SAME_HOSTID")


def rank_env(base, rank, world, port):
    """Environment of child `rank` of the self-launched job (tests/test_bench_env_cpu.py)."""
    env = dict(base, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if is_rehearsal(base):
        # rehearsal of N RCCL ranks on ONE GPU: a host id per rank (mtflearn_amd.distributed.one_gpu_rank_env)
        from mtflearn_amd.distributed import one_gpu_rank_env
        env = one_gpu_rank_env(rank, env)
    return env


def launch(args):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = rank_env(os.environ, r, args.gpus, port)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    code, kill_at = 0, None
    while procs:
        for p in list(procs):
            rc = p.poll()
            if rc is None:
                continue
            procs.remove(p)
            if rc != 0 and code == 0:
                code = rc if rc > 0 else 128 - rc                          # a signal's number, shell style
                print(f"[bench] a rank exited with {rc}: stopping the others", file=sys.stderr)
                for q in procs:                                           # one rank failed: stop the others (rank 0 still
                    q.terminate()                                         # delivers what it has, see Deadman)
                kill_at = time.time() + 20.0
        if kill_at is not None and time.time() > kill_at:
            for q in procs:                                               # exactly the children started above
                q.kill()
            kill_at = None
        time.sleep(0.05)
    return code                                                           # any rank failed or timed out -> non-zero


# ---------------------------------------------------------------------------------------------------------
# failure handling at N > 1: nothing may end as "exit 0 with a plausible line"
# ---------------------------------------------------------------------------------------------------------
class TracedComm:
    """Delegates to a communicator and remembers the call in flight, for the report of a stalled run."""

    def __init__(self, comm):
        self._c = comm
        self.in_flight = None
        self.n_calls = 0

    def __getattr__(self, name):
        attr = getattr(self._c, name)
        if not callable(attr):
            return attr

        def call(*a, **k):
            self.n_calls += 1
            self.in_flight = f"{name}({', '.join(_brief(x) for x in a)}) [call #{self.n_calls}]"
            out = attr(*a, **k)
            self.in_flight = None
            return out
        return call


def _brief(x):
    if hasattr(x, "shape") and hasattr(x, "dtype"):
        return f"<{tuple(x.shape)} {str(x.dtype).replace('torch.', '')}>"
    if isinstance(x, (bytes, bytearray)):
        return f"<{len(x)} bytes>"
    return repr(x)


class Deadman:
    """One per rank.  Knows the phase the rank is in, the line rank 0 could deliver so far, and ends the process
    NON-ZERO -- after saying on stderr where it was (phase, collective in flight, the stacks of every thread) -- when
    a phase overruns its limit, when SIGTERM arrives (the launcher stopping the survivors of a failed rank) or when the
    caller asks (verification failure, communicator unavailable).  Rank 0 first writes the line it has: with `value`
    null and an "unmeasured" reason if the timed region was not completed and verified, else the complete contract
    keys with "side_sections" naming what is missing."""

    def __init__(self, rank, world, json_fd, args):
        import threading
        self.rank, self.world, self.json_fd, self.args = rank, world, json_fd, args
        self.phase, self.result, self.comm = "start-up", None, None
        self._timer, self._lock, self._done = None, threading.Lock(), False
        if world > 1:
            # SIGTERM must be seen even while the main thread sits in a C call (a stalled collective): the C-level
            # handler writes the signal number to a pipe, a helper thread reads it
            import signal
            r, w = os.pipe()
            os.set_blocking(w, False)
            signal.set_wakeup_fd(w, warn_on_full_buffer=False)
            signal.signal(signal.SIGTERM, lambda *_: None)
            t = threading.Thread(target=self._on_signal, args=(r,), daemon=True)
            t.start()

    def _on_signal(self, fd):
        import signal
        while True:
            data = os.read(fd, 16)
            if not data:
                return
            if signal.SIGTERM in data:
                self.fire("stopped by SIGTERM (another rank failed or the launcher gave up)", 143)

    def enter(self, phase, limit_s=None):
        """Start a phase; with a limit, overrunning it ends the run."""
        import threading
        if self._timer is not None:
            self._timer.cancel()
            self._timer = None
        self.phase = phase
        if limit_s and self.world > 1:
            self._timer = threading.Timer(limit_s, self.fire, args=(f"no progress for {limit_s:.0f} s", 124))
            self._timer.daemon = True
            self._timer.start()

    def null_line(self, reason):
        a = self.args
        return {"metric": "patches/s (32x32, n_max=8) + achieved HBM GB/s vs roofline", "value": None, "unit": "patches/s",
                "n_gpus": self.world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": None, "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": f"configs[1]: synthetic {a.frame}x{a.frame} frame per GPU, dense {a.size}-px windows, "
                                       f"n_max={a.n_max}"},
                "unmeasured": reason}

    def fire(self, reason, code):
        import faulthandler
        with self._lock:
            if self._done:
                return
            self._done = True
        where = f"phase '{self.phase}'"
        if self.comm is not None and getattr(self.comm, "in_flight", None):
            where += f", collective in flight: {self.comm.in_flight}"
        msg = f"{reason} in {where}"
        try:
            sys.stderr.write(f"[bench] rank {self.rank}/{self.world}: {msg}; stacks of every thread follow\n")
            sys.stderr.flush()
            faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
            sys.stderr.flush()
        finally:
            if self.rank == 0:
                line = self.result
                if line is None:
                    line = self.null_line(msg)
                else:
                    line = dict(line, side_sections=f"incomplete: {msg}")
                os.write(self.json_fd, (json.dumps(line) + "\n").encode())
            os._exit(code)


# ---------------------------------------------------------------------------------------------------------
# CPU baselines (oracle = checker and baseline only; never inside a timed GPU region)
# ---------------------------------------------------------------------------------------------------------
def cpu_baseline(z, frame, size):
    """Reference CPU path (np.dot of the flattened batch with the float64 basis, _zps.py:151-155)
    restated by the oracle, timed on a bounded sample of the same workload: the first 400k sliding
    windows of the frame, repeated until ~10 s have elapsed."""
    from oracle import zernike_oracle as zo
    from mtflearn_amd.synthetic import sliding_patches
    n_side = frame.shape[0] - size + 1
    rows = max(1, min(n_side, 400000 // n_side))
    sample = sliding_patches(frame, size, rows=range(rows))
    zo.moments_patches(sample[:1000], z.polynomials)                    # warm BLAS
    done, t0 = 0, time.perf_counter()
    while True:
        zo.moments_patches(sample, z.polynomials)
        done += sample.shape[0]
        dt = time.perf_counter() - t0
        if dt > 10.0:
            break
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info() if p.get("user_api") == "blas"] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    return {"value": done / dt, "unit": "patches/s", "cores": int(threads), "kind": "port",
            "host_cpus": os.cpu_count(),
            "sample": f"oracle moments_patches (np.dot, reference _zps.py:151-155) on the first "
                      f"{sample.shape[0]} sliding {size}-px float32 windows of the frame, "
                      f"{done // sample.shape[0]} passes in {dt:.1f} s"}


def cpu_dense_baseline(z, frame):
    """The reference's dense path (fftconvolve, _zps.py:159-193) on a 1024 x 1024 crop (~2.6 GB RSS)."""
    import numpy as np
    from oracle import zernike_oracle as zo
    crop = np.ascontiguousarray(frame[:1024, :1024])
    t0 = time.perf_counter()
    zo.moments_frame_fft(crop, z.polynomials, z.n)
    dt = time.perf_counter() - t0
    return {"value": crop.size / dt, "unit": "patches/s", "cores": 1, "kind": "port",
            "sample": f"oracle moments_frame_fft (scipy fftconvolve, single-threaded) on a 1024x1024 crop, {dt:.2f} s"}


def traffic_record(key):
    """HBM bytes per launch from the PMC run recorded in profiles/traffic.json -- only if it was taken on the
    kernel sources this run was built from."""
    from mtflearn_amd import roofline as rl
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None, "no profiles/traffic.json"
    try:
        rec = json.load(open(path)).get(key)
    except Exception as exc:
        return None, f"unreadable profiles/traffic.json: {exc}"
    if not rec:
        return None, f"no PMC record for {key}"
    have = _kernel_source_hash()
    if rec.get("kernel_source_sha") != have:
        return None, (f"stale: {rec.get('source')} was taken at kernel_source_sha {rec.get('kernel_source_sha')} "
                      f"(git {rec.get('git')}), sources are now {have}")
    return rec["hbm_bytes_per_launch"], f"{rec.get('source')} @ git {rec.get('git')}, kernel_source_sha {have}"


def _kernel_source_hash():
    import glob
    import hashlib
    h = hashlib.sha256()
    # the sources that define the timed batch kernel and its tables
    for name in ("zk_sep_patches.hip", "zk_sep.h", "zk_sep.hip", "zk_fold.h", "zk_internal.h"):
        h.update(open(os.path.join(ROOT, "motif-learn_amd", "csrc", name), "rb").read())
    return h.hexdigest()[:16]


# ---------------------------------------------------------------------------------------------------------
def worker(args):
    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    # stdout carries exactly ONE line (the JSON): whatever libraries print on fd 1 (gloo's connection banner, RCCL
    # with NCCL_DEBUG set) goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    # Rehearsal knobs (a 1-GPU box cannot run RCCL between two ranks): ZK_BENCH_BACKEND=gloo with
    # ZK_BENCH_ONE_DEVICE=1 puts every rank on device 0 and runs the same drivers on the test-aid communicator.
    backend = os.environ.get("ZK_BENCH_BACKEND", "rccl")
    if os.environ.get("ZK_BENCH_ONE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    os.environ["MTFLEARN_AMD_DEVICE"] = str(local_rank)

    from mtflearn_amd import ZPs, _native, roofline as rl
    from mtflearn_amd import distributed as D
    from mtflearn_amd.synthetic import honeycomb_frame

    rehearsal = is_rehearsal(os.environ) and world > 1
    if rehearsal and "NCCL_HOSTID" not in os.environ:
        # the one-GPU rehearsal under an outside launcher (python -m torch.distributed.run ... bench.py --gpus N, the driver's
        # form): the launcher did not give the ranks their host ids, so each takes its own before RCCL is initialised
        os.environ.update(D.one_gpu_rank_env(rank, {}))
    rccl_log = None
    if world > 1 and backend == "rccl":
        # which transport RCCL connects the ranks over goes into the line (allgather.transport): INFO output of its INIT
        # phase into a file of this rank's own, parsed after the gathers have run
        os.environ.update(D.rccl_debug_env(rank))
        rccl_log = os.environ.get("NCCL_DEBUG_FILE")
    dead = Deadman(rank, world, json_fd, args)
    limit = float(os.environ.get("ZK_BENCH_PHASE_TIMEOUT", os.environ.get("ZK_BENCH_SIDE_TIMEOUT", "420")))
    comm = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dead.enter("communicator start-up", limit)
        if backend == "gloo":
            import torch.distributed as dist
            dist.init_process_group("gloo")
            comm = D.TorchComm()
        else:
            # ranks of one node meet through a file named after their common parent (the launcher) and port
            path = f"/tmp/zk_comm_{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}.id"
            try:
                comm = D.RcclComm(local_rank, rank, world, path=path)
            except RuntimeError as exc:
                # the product's collective is what this line measures: without it there is no `value`, and no stand-in
                # communicator is substituted
                dead.fire(f"product collective unavailable: {exc}", 3)
        comm = TracedComm(comm)
        dead.comm = comm

    K, H = args.size, args.frame
    z = ZPs(n_max=args.n_max, size=K)
    plan = z._device_plan()
    n_poly = len(z.n)
    frame = honeycomb_frame(H, seed=rank)                               # one frame per rank (weak scaling)
    f_dev = torch.from_numpy(frame).to(dev)
    patches = f_dev.unfold(0, K, 1).unfold(1, K, 1).reshape(-1, K, K).contiguous()
    n_local = patches.shape[0]
    n_total = world * n_local
    start = rank * n_local
    full = torch.empty((n_total, n_poly), dtype=torch.float64, device=dev)
    mine = full[start:start + n_local]
    fast = plan.has_path(0, _native.ZK_F32, _native.PATH_SEPARABLE)

    def fence():
        torch.cuda.synchronize()
        if comm is not None:
            comm.barrier()
        torch.cuda.synchronize()

    def step_kernel():
        D.patch_moments_device(plan, patches, out=mine)

    def step_gather():
        D.sharded_patch_moments(plan, comm, patches, n_total, out=full, n_chunks=args.gather_chunks)

    def timed_loop(step):
        """W untimed + exactly K timed steps, barrier + synchronize on both sides, max over ranks."""
        for _ in range(args.warmup):
            step()
        fence()
        plan.profile(True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        per_launch = plan.profile_read_launches(cap=max(65536, args.steps * (args.gather_chunks + 2)))
        plan.profile(False)
        launches, kernel_ms = len(per_launch), float(sum(per_launch))
        if comm is not None:
            dt = comm.max_over_ranks(dt)
            kernel_ms = comm.max_over_ranks(kernel_ms)
        # kernel time of every timed step (a step = launches / steps consecutive launches), this rank
        per = launches // args.steps if launches >= args.steps else 1
        step_ms = sorted(sum(per_launch[i * per:(i + 1) * per]) for i in range(len(per_launch) // per))
        spread = {"min": step_ms[0], "median": step_ms[len(step_ms) // 2], "max": step_ms[-1]} if step_ms else None
        return dt, launches, kernel_ms, spread

    dead.enter("timed loop (kernels only)", limit)
    el_kernel, l_kernel, ms_kernel, spread_kernel = timed_loop(step_kernel)
    gather = None
    verified = True
    if world > 1:
        dead.enter("timed loop (kernel + all-gather per step)", limit)
        el_gather, l_gather, ms_gather, spread_gather = timed_loop(step_gather)
        dead.enter("all-gather verification", limit)
        # verification: my block is what my kernel wrote; every other block carries its owner's checksum
        # (recomputed with the same launches as the timed step: a wave's summation order depends on its index within a
        #  launch -- the unit-order rotation of zk_sep_patches.hip -- so another chunking differs in the last bits)
        # -- on one more step into a matrix that starts as NaN, so that nothing an earlier step left behind can pass
        full.fill_(float("nan"))
        step_gather()
        torch.cuda.synchronize()
        check = torch.empty((n_local, n_poly), dtype=torch.float64, device=dev)
        for c0, c1 in D._chunk_bounds(n_local, args.gather_chunks):
            D.patch_moments_device(plan, patches[c0:c1], out=check[c0:c1])
        own_ok = ok = bool(torch.equal(check, mine))
        import struct
        # checksum = wrapping int64 sum of the bit patterns: exact and independent of the reduction order
        bits = lambda t: int(t.view(torch.int64).sum().item())
        sums = [struct.unpack("q", b)[0] for b in comm.allgather_host(struct.pack("q", bits(mine)))]
        for r in range(world):
            ok = ok and bits(full[r * n_local:(r + 1) * n_local]) == sums[r]
        if os.environ.get("ZK_BENCH_DEBUG"):
            print(f"[bench] rank {rank}: owners {sums} seen {[bits(full[r * n_local:(r + 1) * n_local]) for r in range(world)]} "
                  f"nan {[int(torch.isnan(full[r * n_local:(r + 1) * n_local]).sum().item()) for r in range(world)]}", file=sys.stderr)
        ok = comm.max_over_ranks(0.0 if ok else 1.0) == 0.0
        verified = ok
        del check
        fence()
        dead.enter("all-gather alone (3 passes)", limit)
        t1 = time.perf_counter()
        for _ in range(3):
            comm.allgather_rows(full, 1, n_total, n_poly, n_local, 0, n_local, D._current_stream_ptr(full))
            comm.join(D._current_stream_ptr(full))
        fence()
        alone_ms = comm.max_over_ranks((time.perf_counter() - t1) / 3 * 1e3)
        # what RCCL connected the ranks over, from every rank's own log (P2P/IPC = xGMI inside a node; NET/Socket in the
        # one-GPU rehearsal), and how many ranks the communicator holds
        counts = {}
        try:
            if rccl_log:
                with open(rccl_log.replace("%h", socket.gethostname()).replace("%p", str(os.getpid())), errors="replace") as fh:
                    counts = D.rccl_transport_summary(fh.read())
        except OSError:
            counts = {}
        blob = json.dumps(counts).encode()
        blob = (blob if len(blob) <= 256 else b"{}").ljust(256)          # equal-sized payloads on every rank
        all_counts = [json.loads(b.decode()) for b in comm.allgather_host(blob)]
        ranks_seen = int(comm.ranks_seen()) if backend == "rccl" else world
        link_GBps = rl.XGMI_LINKS * rl.XGMI_LINK_GBS
        received = (world - 1) * n_local * n_poly * 8 / (alone_ms * 1e-3) / 1e9
        gather = {"verified": ok, "own_block_equals_recomputation": own_ok, "ms_alone": alone_ms, "chunks": args.gather_chunks,
                  "transport": D.describe_transport(all_counts, rehearsal) if backend == "rccl" else "gloo (rehearsal)",
                  "transport_channels_per_rank": all_counts, "ranks_seen": ranks_seen,
                  "xgmi_bound_GBps_per_rank": link_GBps,
                  "frac_of_xgmi_bound": received / link_GBps,
                  "backend": {"rccl": "rccl (zk_allgather_rows in libzernike_hip.so)", "gloo": "gloo (rehearsal)"}.get(backend, backend),
                  "ms_per_step_with_allgather": el_gather / args.steps * 1e3,
                  "value_with_allgather": n_total / (el_gather / args.steps),
                  "ms_per_step_kernel_only": el_kernel / args.steps * 1e3,
                  "value_kernel_only": n_total / (el_kernel / args.steps),
                  "bytes_per_rank_out": n_local * n_poly * 8, "gathered_bytes": n_total * n_poly * 8,
                  "GBps_received_per_rank": (world - 1) * n_local * n_poly * 8 / (alone_ms * 1e-3) / 1e9}

    in_step = world > 1 and not args.kernel_only_value
    elapsed, launches, kernel_ms, spread = ((el_gather, l_gather, ms_gather, spread_gather) if in_step
                                            else (el_kernel, l_kernel, ms_kernel, spread_kernel))

    # ---- the line's contract keys are complete here; the side sections below only add to it -----------------------------
    ms_per_step = elapsed / args.steps * 1e3
    value = n_total / (elapsed / args.steps)
    per_patch = rl.batch_bytes_per_patch(K, args.n_max, 4)
    achieved = args.steps * n_local * per_patch / (kernel_ms * 1e-3) / 1e9      # all launches of the timed region
    traffic, traffic_source = traffic_record(f"patches_{K}_{args.n_max}_{H}")
    result = {
        "metric": "patches/s (32x32, n_max=8) + achieved HBM GB/s vs roofline",
        "value": value, "unit": "patches/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"configs[1]: synthetic {H}x{H} honeycomb STEM frame per GPU, all {n_local} dense "
                               f"{K}-px sliding windows as a float32 (N,{K},{K}) batch resident in HBM, n_max={args.n_max} "
                               f"({n_poly} moments), float64 out",
                   "patches_per_gpu": n_local, "patch_size": K, "n_max": args.n_max, "input_dtype": "f32",
                   "kernel": "zk_patch_sep_kernel (mirror-folded, row-separable, LDS-DMA staged)" if fast else "zk_generic_kernel",
                   "allgather_in_step": in_step,
                   "parallelism": f"dp{world} (one frame's patch batch per GPU, no data-path collective"
                                  + ("; every step ends with the all-gather of the moment matrix, chunked and "
                                     "overlapped with the kernels)" if in_step else ")")},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": rl.HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / rl.HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     "kernel_ms": kernel_ms / args.steps, "launches": launches,
                     "kernel_ms_min": spread["min"], "kernel_ms_median": spread["median"], "kernel_ms_max": spread["max"],
                     "frac_at_median": n_local * per_patch / (spread["median"] * 1e-3) / 1e9 / rl.HBM_PEAK_GBS,
                     "algorithmic_bytes_per_patch": per_patch, "patches_per_step": n_local},
        "kernel_only_patches_per_s": n_total / (ms_kernel / args.steps * 1e-3),
    }
    if gather is not None:
        result["allgather"] = gather
    if not args.only_timed_loop and world == 1:
        # what THIS box sustains on a plain stream over the very buffer the kernel reads (same physical pages), and on a copy
        # into the buffer it writes: a slow box or an unlucky placement shows here, a bad access pattern only in the kernel
        real_bytes = (traffic or n_local * (per_patch - 2 * K * 4))          # PMC traffic, else rows 0 / K-1 not fetched
        try:
            in_bytes = (patches.numel() * 4 // 262144) * 262144
            groups = in_bytes // 262144
            ms_read = _native.hbm_probe(local_rank, patches.data_ptr(), in_bytes)
            n_copy = min(in_bytes, (full.numel() * 8 // 262144) * 262144)
            ms_copy = _native.hbm_probe(local_rank, patches.data_ptr(), n_copy, dst_ptr=full.data_ptr())
            per_group = (full.numel() * 8 // groups) // 16 * 16               # the kernel's own ratio of written to read bytes
            ms_mix = _native.hbm_probe(local_rank, patches.data_ptr(), in_bytes, dst_ptr=full.data_ptr(), store_per_group=per_group)
            D.patch_moments_device(plan, patches, out=mine)                    # `full` was the probes' target: restore the moments
            torch.cuda.synchronize()
            mix_bytes = in_bytes + groups * per_group
            result["roofline"]["this_box"] = {
                "stream_read_GBps": in_bytes / ms_read / 1e6, "copy_GBps": 2 * n_copy / ms_copy / 1e6,
                "stream_with_the_kernels_store_ratio_GBps": mix_bytes / ms_mix / 1e6,
                "stream_with_stores_ms_scaled_to_kernel_traffic": ms_mix * real_bytes / mix_bytes,
                "kernel_ms_median": spread["median"], "kernel_real_traffic_GBps": real_bytes / (spread["median"] * 1e-3) / 1e9,
                "note": "zk_hbm_probe on the kernel's own buffers, HIP events: LDS-DMA stream read of the batch; 16-B-per-lane "
                        "copy; the same read stream with the kernel's ratio of stored to read bytes (no arithmetic).  The last "
                        "one, scaled to the bytes the kernel really moves, is what a bare stream of this traffic mix takes "
                        "on THIS box: mixing 8.6 % writes into the reads costs far more than their bytes, and by how much "
                        "differs from box to box (profiles/r03_stream_limits.txt)"}
        except RuntimeError as exc:
            result["roofline"]["this_box"] = {"error": str(exc)}
    if not verified:
        # a `value` whose gathered matrix is wrong is not a measurement: keep what was timed under another name, say why,
        # and leave non-zero (every rank takes this branch: `verified` went through max_over_ranks)
        result["value_unverified"], result["value"] = result["value"], None
        result["unmeasured"] = "all-gather verification failed (see allgather.verified / own_block_equals_recomputation)"
        dead.result = result
        dead.fire("the gathered moment matrix differs from its owners' blocks", 4)
    dead.result = result            # from here on a stalled or failed side section still delivers the contract keys

    # N > 1: every side section runs collectives.  A section that fails on one rank cannot be skipped by that rank alone
    # (the others are inside its collectives): the rank says what happened and leaves non-zero, the launcher stops the
    # others, rank 0 delivers the line it has (Deadman), and the job's exit code is non-zero.  A section that stalls
    # ends the same way through its phase limit.
    side = {}
    if not args.only_timed_loop and not args.no_clustering:
        dead.enter("side section: clustering", limit)
        try:
            side["clustering"] = clustering_section(comm, world, dev, mine, n_local, n_poly,
                                                    cpu_sample=0 if (args.no_cpu_baseline or world > 1) else 500000)
        except Exception as exc:
            if world > 1:
                import traceback
                traceback.print_exc()
                dead.fire(f"clustering section raised {type(exc).__name__}: {exc}", 5)
            side["clustering"] = {"error": f"{type(exc).__name__}: {exc}"}      # one rank: nobody waits for us
    if world > 1 and not args.only_timed_loop:
        side.update(multi_rank_sections(args, comm, rank, world, dev, plan, z, dead, limit))
        bad = [k for k in ("multi_frame", "sharded_maps") if k in side and not side[k].get("verified", True)]
        if bad:
            result.update(side)
            dead.result = result
            dead.fire(f"verification of {bad} failed", 4)
    dead.enter("finishing")

    if rank != 0:
        if comm is not None:
            comm.barrier()
            comm.close()
        return
    result.update(side)

    if world == 1 and not args.only_timed_loop:
        del patches, full, mine
        torch.cuda.empty_cache()
        single_gpu_sections(args, result, dev, plan, z, f_dev, frame)
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(z, frame, K)
            result["cpu_baseline_dense"] = cpu_dense_baseline(z, frame)
            result["gpu_over_cpu"] = value / result["cpu_baseline"]["value"]
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(result) + "\n").encode())
    if comm is not None:
        comm.barrier()
        comm.close()


# ---------------------------------------------------------------------------------------------------------
# downstream consumer on the moment matrix where the timed kernel left it (every rank takes part; rank 0 reports):
# k-means of ALL ranks' moments without gathering them -- the ranks exchange k x D sums, then 4 bytes of label per row
# ---------------------------------------------------------------------------------------------------------
def clustering_section(comm, world, dev, mine, n_local, n_poly, k=4, cpu_sample=0):
    import numpy as np
    import torch
    from mtflearn_amd.clustering import DeviceRows, kmeans_fit, gather_labels
    torch.cuda.synchronize()
    worst = (lambda v: comm.max_over_ranks(v)) if comm is not None else (lambda v: v)
    rows = DeviceRows.adopt(mine.data_ptr(), n_local, n_poly, device=dev.index)
    try:
        t0 = time.perf_counter()
        labels, _, n_iter = kmeans_fit(rows, k, random_state=0, comm=comm)
        s_fit = worst(time.perf_counter() - t0)
        t0 = time.perf_counter()
        whole = gather_labels(labels, comm)
        s_gather = worst(time.perf_counter() - t0)
        # one Lloyd pass on its own (the synchronous C-ABI call: centre table up, kernel, fixed-order reduction, k x (D+1) sums back)
        import statistics

        def timed_calls(call, warm=10, reps=12):
            """Kernel time of a synchronous C-ABI pass (HIP events inside the library): `warm` untimed calls, then `reps` timed
            ones; median, minimum, maximum (median of >= 10 after warm-up) and the wall clock per call."""
            for _ in range(warm):
                call()
            ms, t0 = [], time.perf_counter()
            for _ in range(reps):
                call()
                ms.append(rows.last_kernel_ms())
            wall = (time.perf_counter() - t0) / reps
            return {"median": worst(statistics.median(ms)), "min": min(ms), "max": max(ms), "timed_calls": reps, "warm_calls": warm}, worst(wall)

        centers0 = np.zeros((k, n_poly))
        rows.profile(True)
        lloyd_t, s_pass = timed_calls(lambda: rows.lloyd(centers0, True))
        kernel_ms = lloyd_t["median"]
        # the two passes of a Gaussian-mixture EM iteration (gmm_lbs), k = 6 full covariances, on the same resident block: kernels
        # on the matrix cores; local passes only (no collective), every rank its own block
        kg = 6
        rng = np.random.default_rng(0)
        prec = np.stack([np.triu(rng.standard_normal((n_poly, n_poly)) * 0.05, 1) + np.eye(n_poly) * (1.0 + 0.1 * c) for c in range(kg)])
        means_g = rng.standard_normal((kg, n_poly)) * 0.01
        log_det = np.log(np.einsum("kii->ki", prec)).sum(axis=1)
        log_w = np.log(np.full(kg, 1.0 / kg))
        e_t, _ = timed_calls(lambda: rows.estep(prec, means_g, log_det, log_w))
        m_t, _ = timed_calls(lambda: rows.moments(0, np.zeros(n_poly), count=kg))
        e_ms, m_ms = e_t["median"], m_t["median"]
        rows.profile(False)
    finally:
        rows.close()
    # correlation kNN of ForceGraph8.compute_graph on the first 100 000 moment vectors (its own resident copy)
    knn = None
    n_knn = min(100000, n_local)
    if n_knn >= 1024:
        from mtflearn_amd.manifold import _knn_affinities
        with DeviceRows(mine[:n_knn].cpu().numpy()) as sub:
            for _ in range(3):
                _knn_affinities(sub, 10, 1, 10)
            knn_s = []
            for _ in range(10):
                t0 = time.perf_counter()
                _knn_affinities(sub, 10, 1, 10)
                knn_s.append(time.perf_counter() - t0)
            s_knn = worst(sorted(knn_s)[len(knn_s) // 2])
        steps = 12 if n_poly <= 48 else (n_poly + 15) // 16 * 4
        knn = {"workload": f"zk_rows_knn_correlation: 10 nearest neighbours (correlation distance) + affinities of {n_knn} x {n_poly} rows, "
                           "N^2 scalar products on the matrix cores with the top-k kept in the result lanes; wall clock of the C-ABI call "
                           "(kernels + three result copies to the host)",
               "ms": s_knn * 1e3, "ms_min": min(knn_s) * 1e3, "ms_max": max(knn_s) * 1e3, "timed_calls": len(knn_s), "warm_calls": 3,
               "roofline": {"bound": "mfma-f64", "achieved": 2.0 * n_knn * n_knn * 4 * steps / s_knn / 1e12, "peak": 78.6, "unit": "TFLOP/s",
                            "frac": 2.0 * n_knn * n_knn * 4 * steps / s_knn / 1e12 / 78.6}}
    sizes = np.bincount(whole, minlength=k).astype(np.int64)
    agree = True
    if comm is not None:
        agree = all(b == sizes.tobytes() for b in comm.allgather_host(sizes.tobytes()))
    n_total = world * n_local
    cpu = None
    if cpu_sample:                                                   # what the reference's kmeans_lbs runs: scikit-learn on the host
        from sklearn.cluster import KMeans
        sample = mine[:cpu_sample].cpu().numpy()
        t0 = time.perf_counter()
        model = KMeans(n_clusters=k, random_state=0).fit(sample)
        s_cpu = time.perf_counter() - t0
        cpu = {"value": len(sample) * model.n_iter_ / s_cpu, "unit": "rows x Lloyd iterations / s", "cores": os.cpu_count(),
               "kind": "dependency (scikit-learn: the call inside the reference wrapper)", "sample": f"sklearn.cluster.KMeans(n_clusters={k}, random_state=0).fit on the first {len(sample)} "
               f"moment vectors ({model.n_iter_} iterations, {s_cpu:.2f} s) -- the call inside the reference's kmeans_lbs"}
    out = {"workload": f"kmeans_lbs flow (scikit-learn KMeans(n_clusters={k}, random_state=0): k-means++ seeding, Lloyd) on the "
                        f"({n_total}, {n_poly}) float64 moment matrix of the timed step, resident in HBM, "
                        + (f"one block per rank: the ranks exchange {k} x {n_poly + 1} sums per pass and gather 4-byte labels, "
                           "never the moments" if world > 1 else "adopted where the batch kernel wrote it"),
            "k": k, "lloyd_iterations": n_iter, "s_fit": s_fit, "s_label_gather": s_gather,
            "lloyd_pass": {"ms_per_call": s_pass * 1e3, "kernel_ms": kernel_ms, "kernel_ms_spread": lloyd_t, "bytes_per_rank": 8 * n_poly * n_local + 4 * n_local,
                           "roofline": {"bound": "hbm", "achieved": (8 * n_poly + 4) * n_local / (kernel_ms * 1e-3) / 1e9,
                                        "peak": 8000.0, "unit": "GB/s",
                                        "frac": (8 * n_poly + 4) * n_local / (kernel_ms * 1e-3) / 1e9 / 8000.0}},
            "rows_per_s_end_to_end": n_total / (s_fit + s_gather), "cluster_sizes": sizes.tolist(), "ranks_agree": bool(agree),
            "label_bytes_gathered": 4 * n_total, "moment_bytes_not_gathered": 8 * n_poly * n_total}
    out["rows_x_iterations_per_s"] = n_total * n_iter / s_fit
    # MFMA work of the mixture passes: E step k x (4 + 8 + .. + 4 NB) MFMAs of 2 048 flop per 16 rows; M step k x 6 blocks x 16 per 64 rows
    nb = (n_poly + 15) // 16
    if nb <= 3:
        e_flop = kg * 2 * nb * (nb + 1) * 2048.0 / 16 * n_local
        m_flop = kg * (nb * (nb + 1) // 2) * 16 * 2048.0 / 64 * n_local
        out["mixture_em"] = {"workload": f"one EM iteration of gmm_lbs (k = {kg}, full covariances) on the rank's ({n_local}, {n_poly}) block: "
                                         "E step (estep_mfma_kernel) and weighted second moments of all components (wgram_mfma_kernel)",
                             "e_step_kernel_ms": e_ms, "m_step_kernel_ms": m_ms, "e_step_spread": e_t, "m_step_spread": m_t,
                             "e_step_roofline": {"bound": "mfma-f64", "achieved": e_flop / (e_ms * 1e-3) / 1e12, "peak": 78.6,
                                                 "unit": "TFLOP/s", "frac": e_flop / (e_ms * 1e-3) / 1e12 / 78.6},
                             "m_step_roofline": {"bound": "mfma-f64", "achieved": m_flop / (m_ms * 1e-3) / 1e12, "peak": 78.6,
                                                 "unit": "TFLOP/s", "frac": m_flop / (m_ms * 1e-3) / 1e12 / 78.6}}
    if knn is not None:
        out["knn_graph"] = knn
    if cpu is not None:
        out["cpu_baseline"] = cpu
        out["gpu_over_cpu"] = out["rows_x_iterations_per_s"] / cpu["value"]
    return out


# ---------------------------------------------------------------------------------------------------------
# configs[3] / configs[4] at N > 1 (every rank takes part; rank 0 reports)
# ---------------------------------------------------------------------------------------------------------
def _bits(t):
    """Wrapping int64 sum of the bit patterns: exact, independent of the reduction order."""
    import torch
    return int(t.contiguous().view(torch.int64).sum().item())


def multi_rank_sections(args, comm, rank, world, dev, plan, z, dead, limit):
    import struct
    import numpy as np
    import torch
    from mtflearn_amd import ZPs, distributed as D, roofline as rl
    from mtflearn_amd.synthetic import honeycomb_frame
    out = {}
    K = args.size
    nan = float("nan")

    def fence():
        torch.cuda.synchronize()
        comm.barrier()
        torch.cuda.synchronize()

    def timed(fn, reps):
        fn()
        fence()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        fence()
        return comm.max_over_ranks((time.perf_counter() - t0) / reps)

    def owners_agree(mine_sums, seen_sums_of):
        """Every rank sends the bit-sums of ITS blocks; every rank compares what arrived with what the owner sent."""
        n = len(mine_sums)
        theirs = [struct.unpack(f"{n}q", b) for b in comm.allgather_host(struct.pack(f"{n}q", *mine_sums))]
        return all(tuple(seen_sums_of(r)) == theirs[r] for r in range(world))

    if not args.no_multi_frame:
        # configs[3]: 8 frames of 2048^2 per rank, moments of all 8 x world frames reassembled on every rank
        dead.enter("side section: multi_frame (configs[3])", limit)
        per_rank, Hf = int(os.environ.get("ZK_BENCH_FRAMES_PER_RANK", "8")), args.frame
        n_frames = per_rank * world
        frames = torch.stack([torch.from_numpy(honeycomb_frame(Hf, seed=1000 + rank * per_rank + i)) for i in range(per_rank)]).to(dev)
        full = torch.empty((n_frames, plan.n_poly, Hf, Hf), dtype=torch.float64, device=dev)
        sec = timed(lambda: D.sharded_frames_moments(plan, comm, frames, n_frames, out=full), 2)
        sec_k = timed(lambda: [D.frame_moments_device(plan, frames[i], out=full[rank * per_rank + i]) for i in range(per_rank)], 2)
        # verification as for the patch matrix: a pass into a result that starts as NaN; own frames bit-equal to a
        # recomputation, every other frame's bit-sum equal to its owner's
        dead.enter("side section: multi_frame verification", limit)
        full.fill_(nan)
        D.sharded_frames_moments(plan, comm, frames, n_frames, out=full)
        torch.cuda.synchronize()
        scratch = torch.empty((plan.n_poly, Hf, Hf), dtype=torch.float64, device=dev)
        own_ok = True
        for i in range(per_rank):
            D.frame_moments_device(plan, frames[i], out=scratch)
            own_ok = own_ok and bool(torch.equal(scratch, full[rank * per_rank + i]))
        del scratch
        ok = own_ok and owners_agree([_bits(full[rank * per_rank + i]) for i in range(per_rank)],
                                     lambda r: [_bits(full[r * per_rank + i]) for i in range(per_rank)])
        out["multi_frame"] = {
            "workload": f"configs[3]: {n_frames} synthetic {Hf}x{Hf} frames, {per_rank} per GPU, dense {K}-px moments, "
                        f"(F, {plan.n_poly}, H, W) reassembled on every rank (frame i gathered while frame i+1 is computed)",
            "positions_per_s_with_allgather": n_frames * Hf * Hf / sec, "s_per_pass_with_allgather": sec,
            "positions_per_s_kernels_only": n_frames * Hf * Hf / sec_k, "s_per_pass_kernels_only": sec_k,
            "gathered_bytes": full.numel() * 8, "own_block_equals_recomputation": comm.max_over_ranks(0.0 if own_ok else 1.0) == 0.0,
            "verified": comm.max_over_ranks(0.0 if ok else 1.0) == 0.0}
        del frames, full
        torch.cuda.empty_cache()
    if not args.no_maps:
        # configs[4]: one 4096^2 frame, n_max 10, row bands -> fused maps -> the 41 map planes gathered
        dead.enter("side section: sharded_maps (configs[4])", limit)
        Hm = int(os.environ.get("ZK_BENCH_MAPS_FRAME", "4096"))
        z10 = ZPs(n_max=10, size=K)
        plan10 = z10._device_plan()
        big = torch.from_numpy(honeycomb_frame(Hm, seed=1)).to(dev)
        theta = np.linspace(0, 2 * np.pi, 360, endpoint=False)
        n_c = rl.n_complex(10)
        maps = (torch.empty((4, Hm, Hm), dtype=torch.float64, device=dev), torch.empty((n_c, Hm, Hm), dtype=torch.float64, device=dev),
                torch.empty((Hm, Hm), dtype=torch.float64, device=dev))
        run = lambda: D.sharded_frame_maps(plan10, comm, big, n_c, theta=theta, n_chunks=args.gather_chunks, out=maps)
        sec = timed(run, 3)
        dead.enter("side section: sharded_maps verification", limit)
        for t in maps:
            t.fill_(nan)
        run()
        torch.cuda.synchronize()
        band = lambda t, r: t[..., D.shard_bounds(Hm, r, world)[0]:sum(D.shard_bounds(Hm, r, world)[:2]), :]
        start, count, _ = D.shard_bounds(Hm, rank, world)
        again = D.frame_maps_device(plan10, big, n_c, theta=theta, row0=start, n_rows=count)
        own_ok = all(bool(torch.equal(a, band(t, rank))) for a, t in zip(again, maps))
        del again
        ok = own_ok and owners_agree([_bits(band(t, rank)) for t in maps], lambda r: [_bits(band(t, r)) for t in maps])
        out["sharded_maps"] = {
            "workload": f"configs[4]: {Hm}x{Hm} frame, 32-px, n_max=10 -> rot_maps[2,3,4,6] + 36 |Z_nm| planes + mirror_map(360), "
                        f"row bands over {world} GPUs, 41 map planes gathered on every rank",
            "positions_per_s": Hm * Hm / sec, "s_per_pass": sec, "gathered_bytes": 41 * Hm * Hm * 8,
            "own_block_equals_recomputation": comm.max_over_ranks(0.0 if own_ok else 1.0) == 0.0,
            "verified": comm.max_over_ranks(0.0 if ok else 1.0) == 0.0}
        del big, maps
        torch.cuda.empty_cache()
    return out


# ---------------------------------------------------------------------------------------------------------
# the other single-GPU BASELINE configs, each with its own roofline (N = 1, rank 0)
# ---------------------------------------------------------------------------------------------------------
class _Timed(dict):
    """Result of _profiled: the median pass (``ms``) with its spread, how it was taken, and the shader clock meanwhile."""
    ms = property(lambda self: self["kernel_ms"])


def _profiled(plan, fn, reps=10, warm_ms=40.0):
    """Kernel time of one pass of ``fn`` (HIP events around every launch, zk_plan_profile), taken in the chip's STEADY state:
    untimed passes run back to back until >= ``warm_ms`` of kernel time have gone by (the power management moves the shader
    clock for the first ~15 ms of an FP64-heavy load: 2.05 GHz on the first launches, a dip to 1.7 around the 4th, 2.15-2.2
    from ~20 ms on -- profiles/r04_strip_clock_series.txt), then ``reps`` (>= 10) timed passes back to back.
    Returns the median pass with minimum / maximum, and the mean shader clock over the timed passes (zk_clock_monitor)."""
    import statistics
    import torch
    from mtflearn_amd import _native
    reps = max(10, int(reps))
    plan.profile(True)
    fn()
    torch.cuda.synchronize()
    launches, first_ms = plan.profile_read()
    warm = max(2, min(40, int(warm_ms / max(first_ms, 1e-3)) + 1))
    plan.profile(False)
    for _ in range(warm):
        fn()
    torch.cuda.current_stream().synchronize()          # (the monitor's window then holds the timed passes only)
    plan.profile(True)
    with _native.ClockMonitor(torch.cuda.current_device()) as clock:
        for _ in range(reps):
            fn()
        torch.cuda.current_stream().synchronize()      # (a DEVICE synchronisation would wait for the monitor itself)
    per_launch = plan.profile_read_launches()
    plan.profile(False)
    assert len(per_launch) == launches * reps, (len(per_launch), launches, reps)
    passes = [sum(per_launch[k * launches:(k + 1) * launches]) for k in range(reps)]
    return _Timed(kernel_ms=statistics.median(passes), kernel_ms_min=min(passes), kernel_ms_max=max(passes), timed_passes=reps,
                  warm_passes=warm + 1, launches_per_pass=launches, shader_clock_ghz=clock.ghz)


def _spread(t):
    """The measurement record every side section carries beside its median."""
    return {k: t[k] for k in ("kernel_ms_min", "kernel_ms_max", "timed_passes", "warm_passes", "shader_clock_ghz")}


def _hbm_roofline(bytes_per_pass, ms):
    from mtflearn_amd import roofline as rl
    a = bytes_per_pass / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": a, "peak": rl.HBM_PEAK_GBS, "unit": "GB/s", "frac": a / rl.HBM_PEAK_GBS}


def _fp64_roofline(flops_per_pass, bytes_per_pass, ms, clock_ghz=None):
    """Fraction of the FP64 vector peak (78.6 TFLOP/s at the nominal 2.4 GHz) of the MEDIAN pass; with the shader clock
    measured during the timed passes also against the peak at THAT clock: what the kernel makes of the cycles it is given."""
    from mtflearn_amd import roofline as rl
    t = flops_per_pass / (ms * 1e-3) / 1e12
    r = {"bound": "fp64-valu", "achieved": t, "peak": rl.FP64_VECTOR_PEAK_TF, "unit": "TFLOP/s (executed f64)",
         "frac": t / rl.FP64_VECTOR_PEAK_TF, "hbm_GBps_algorithmic": bytes_per_pass / (ms * 1e-3) / 1e9,
         "hbm_frac": bytes_per_pass / (ms * 1e-3) / 1e9 / rl.HBM_PEAK_GBS}
    if clock_ghz:
        r["shader_clock_ghz"] = clock_ghz
        r["peak_at_clock"] = rl.FP64_VECTOR_PEAK_TF * clock_ghz / 2.4
        r["frac_at_clock"] = t / r["peak_at_clock"]
    return r


def single_gpu_sections(args, result, dev, plan, z, f_dev, frame):
    import numpy as np
    import torch
    from mtflearn_amd import ZPs, _native, distributed as D, roofline as rl
    from mtflearn_amd.synthetic import honeycomb_frame
    K, H, n_max = args.size, args.frame, args.n_max
    n_poly = len(z.n)

    # ---- configs[1] in dense form: the dense-frame kernels on the same frame -------------------------------
    if not args.no_dense:
        out_f = D.frame_moments_device(plan, f_dev)
        npx = H * H
        strip = rl.strip2_available(K, n_max) and not os.environ.get("ZK_NO_STRIP")
        flops = rl.strip2_flops_per_unit(z.polynomials[0], n_max) if strip else rl.sep_flops_per_unit(z.polynomials[0], n_max)
        dense = {"positions": npx, "kernels": {}}
        for path in (_native.PATH_SEPARABLE, _native.PATH_FOLDED):
            if not plan.has_path(1, _native.ZK_F32, path):
                continue
            plan.set_path(path)
            t = _profiled(plan, lambda: D.frame_moments_device(plan, f_dev, out=out_f))
            ms = t.ms
            entry = {"patches_per_s": npx / (ms * 1e-3), "kernel_ms": ms, **_spread(t)}
            if path == _native.PATH_SEPARABLE:
                entry["kernel"] = "zk_frame_strip2_kernel" if strip else "zk_frame_sep_kernel"
                entry["fp64_flops_per_position"] = flops
                entry["roofline"] = _fp64_roofline(npx * flops, npx * rl.dense_bytes_per_position(n_max), ms, t["shader_clock_ghz"])
            else:
                entry["kernel"] = "zk_frame_fold_kernel"
                entry["roofline"] = _fp64_roofline(npx * (rl.direct_flops_per_unit(z.polynomials[0], n_max) / 4 + 8 * K * K / 4),
                                                   npx * rl.dense_bytes_per_position(n_max), ms, t["shader_clock_ghz"])
            dense["kernels"][_native.PATH_NAMES[path]] = entry
        plan.set_path(_native.PATH_AUTO)
        result["dense_frame"] = dense
        del out_f
        torch.cuda.empty_cache()

    # ---- moments at key points of the resident frame (reference features/_keypoint.py:60-78 + _zps.py:146-157) ----
    if not args.no_dense and plan.supports(_native.OP_POINTS, _native.ZK_F32):
        from ctypes import c_void_p
        n_pts = 1 << 20
        rng = np.random.default_rng(0)
        pts = rng.integers(K // 2, H - K // 2, size=(n_pts, 2)).astype(np.int32)
        out_p = torch.empty((n_pts, n_poly), dtype=torch.float64, device=dev)
        flops = rl.sep_flops_per_unit(z.polynomials[0], n_max)
        disk_px = int(np.count_nonzero(z.polynomials[0]))
        sec = {"workload": f"{n_pts} key points on the resident {H}x{H} frame -> ({n_pts}, {n_poly}) float64 moments, no (N, {K}, {K}) batch in memory; "
                           f"points bucketed on the device by frame row x 256 columns (bit-identical to the caller's order)",
               "kernel": "zk_points_sep_kernel (+ zk_points_hist / _scan / _scatter)", "fp64_flops_per_point": flops,
               "l2_gather_bytes_per_point": disk_px * 4, "algorithmic_hbm_bytes_per_point": 8 + 8 * n_poly}
        for order in ("random", "sorted"):
            d_pts = torch.from_numpy(pts if order == "random" else pts[np.lexsort((pts[:, 0], pts[:, 1]))]).to(dev)
            run = lambda: _native.check(plan._lib.zk_transform_points_dev(plan._h, c_void_p(f_dev.data_ptr()), _native.ZK_F32, H, H,
                                                                          c_void_p(d_pts.data_ptr()), n_pts, c_void_p(out_p.data_ptr()),
                                                                          c_void_p(D._current_stream_ptr(out_p))), "zk_transform_points_dev")
            t = _profiled(plan, run)
            # the whole call, bucketing kernels included (they carry no profiling bracket): stream events around 10 calls
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record()
            torch.cuda.current_stream().synchronize()
            call_ms = e0.elapsed_time(e1) / 10
            sec[order] = {"points_per_s": n_pts / (call_ms * 1e-3), "call_ms": call_ms, "moment_kernel_ms": t.ms, **_spread(t),
                          "roofline": _fp64_roofline(n_pts * flops, n_pts * (8 + 8 * n_poly), t.ms, t["shader_clock_ghz"]),
                          "l2_gather_GBps": n_pts * disk_px * 4 / (t.ms * 1e-3) / 1e9,
                          "l2_gather_frac_of_17_TBps": n_pts * disk_px * 4 / (t.ms * 1e-3) / 1e9 / 17000.0}
        result["key_points"] = sec
        del out_p, d_pts
        torch.cuda.empty_cache()

    # ---- configs[3], one GPU's share: 8 frames of 2048^2, dense moments of each into (8, N_poly, H, W) ----------------
    if not args.no_multi_frame:
        per_rank = 8
        frames = torch.stack([torch.from_numpy(honeycomb_frame(H, seed=1000 + i)) for i in range(per_rank)]).to(dev)
        full = torch.empty((per_rank, n_poly, H, H), dtype=torch.float64, device=dev)
        t = _profiled(plan, lambda: [D.frame_moments_device(plan, frames[i], out=full[i]) for i in range(per_rank)])
        ms, launches = t.ms, t["launches_per_pass"]
        strip = rl.strip2_available(K, n_max) and not os.environ.get("ZK_NO_STRIP")
        flops = rl.strip2_flops_per_unit(z.polynomials[0], n_max) if strip else rl.sep_flops_per_unit(z.polynomials[0], n_max)
        result["config3_per_gpu"] = {
            "workload": f"configs[3], one GPU's share of the 64-frame batch: {per_rank} synthetic {H}x{H} frames, dense {K}-px moments "
                        f"-> ({per_rank}, {n_poly}, {H}, {H}) float64 ({full.numel() * 8 / 1e9:.1f} GB); N > 1 adds the all-gather "
                        f"('multi_frame')",
            "kernel_ms_per_pass": ms, "launches_per_pass": launches, "positions_per_s": per_rank * H * H / (ms * 1e-3), **_spread(t),
            "roofline": _fp64_roofline(per_rank * H * H * flops, per_rank * H * H * rl.dense_bytes_per_position(n_max), ms,
                                       t["shader_clock_ghz"])}
        del frames, full
        torch.cuda.empty_cache()

    # ---- north star's own size: all dense 32-px windows of a 4096^2 frame as one batch ----------------------
    if not args.no_4096 and K == 32:
        n4 = (4096 - K + 1) ** 2
        need = n4 * rl.batch_bytes_per_patch(K, n_max) + 4096 * 4096 * 4
        if torch.cuda.mem_get_info()[0] > need * 1.1:
            f4 = torch.from_numpy(honeycomb_frame(4096, seed=2)).to(dev)
            p4 = f4.unfold(0, K, 1).unfold(1, K, 1).reshape(-1, K, K).contiguous()
            o4 = torch.empty((n4, n_poly), dtype=torch.float64, device=dev)
            t = _profiled(plan, lambda: D.patch_moments_device(plan, p4, out=o4))
            ms = t.ms
            result["north_star_4096"] = {
                "workload": f"north_star: all {n4} dense {K}-px windows of a 4096x4096 frame as one float32 batch "
                            f"({n4 * K * K * 4 / 1e9:.1f} GB), n_max={n_max}",
                "kernel_ms": ms, "patches_per_s": n4 / (ms * 1e-3), **_spread(t),
                "roofline": _hbm_roofline(n4 * rl.batch_bytes_per_patch(K, n_max), ms)}
            del p4, o4, f4
            torch.cuda.empty_cache()

    # ---- configs[2]: 4096^2 frame, 64-px windows, n_max 12 -- batch form (HBM) and dense form (FP64) --------
    if not args.no_config2:
        z12 = ZPs(n_max=12, size=64)
        plan12 = z12._device_plan()
        f2 = torch.from_numpy(honeycomb_frame(4096, seed=3)).to(dev)
        rows = 496                                                       # 496 x 4033 = 2.0 M windows = 32.8 GB
        p2 = f2[:rows + 63].unfold(0, 64, 1).unfold(1, 64, 1).reshape(-1, 64, 64).contiguous()
        o2 = torch.empty((p2.shape[0], 91), dtype=torch.float64, device=dev)
        t = _profiled(plan12, lambda: D.patch_moments_device(plan12, p2, out=o2))
        ms = t.ms
        result["config2_batch"] = {
            "workload": f"configs[2] in batch form: {p2.shape[0]} dense 64-px windows (the first {rows} window rows of a "
                        f"4096x4096 frame; all 16.3 M would be 266 GB) as one float32 batch, n_max=12 (91 moments)",
            "kernel_ms": ms, "patches_per_s": p2.shape[0] / (ms * 1e-3), **_spread(t),
            "roofline": _hbm_roofline(p2.shape[0] * rl.batch_bytes_per_patch(64, 12), ms)}
        del p2, o2
        torch.cuda.empty_cache()
        od = torch.empty((91, 4096, 4096), dtype=torch.float64, device=dev)
        t = _profiled(plan12, lambda: D.frame_moments_device(plan12, f2, out=od))
        ms = t.ms
        strip12 = rl.strip2_available(64, 12) and not os.environ.get("ZK_NO_STRIP") and not os.environ.get("ZK_STRIP_NO_SPLIT")
        flops = rl.strip2_flops_per_unit(z12.polynomials[0], 12) if strip12 else rl.sep_flops_per_unit(z12.polynomials[0], 12)
        result["config2_dense"] = {
            "workload": "configs[2]: 4096x4096 frame, every 64-px window (zero-padded 'same' positions), n_max=12 -> (91, 4096, 4096) float64",
            "kernel": "zk_frame_strip2_kernel<12> (two outputs per lane, two passes by x parity)" if strip12 else "zk_frame_sep_kernel<12>",
            "kernel_ms": ms, "positions_per_s": 4096 * 4096 / (ms * 1e-3), **_spread(t),
            "fp64_flops_per_position": flops,
            "roofline": _fp64_roofline(4096 * 4096 * flops, 4096 * 4096 * rl.dense_bytes_per_position(12), ms, t["shader_clock_ghz"])}
        del od, f2, plan12, z12
        torch.cuda.empty_cache()

    # ---- the orders the reference's estimator returns for larger windows: the plain sum on the matrix cores ------
    # (what ZK_PATH_AUTO runs from n_max 17: every disk pixel times the caller's own float64 basis value)
    if not args.no_high_order:
        z20 = ZPs(n_max=20, size=40)
        plan20 = z20._device_plan()
        fh = torch.from_numpy(honeycomb_frame(2048, seed=5)).to(dev)
        n_t = 1 << 18
        pt = fh.unfold(0, 40, 3).unfold(1, 40, 3).reshape(-1, 40, 40)[:n_t].contiguous()
        assert pt.shape[0] == n_t
        useful = rl.direct_flops_per_unit(z20.polynomials[0], 20)       # 2 x disk pixels x 231 functions
        kernels = {"batch": _native.PATH_NAMES[plan20.best_path(0, _native.ZK_F32, n_t)], "dense": _native.PATH_NAMES[plan20.best_path(1, _native.ZK_F32)]}

        def mfma_roofline(units, t):
            tf = units * useful / (t.ms * 1e-3) / 1e12
            r = {"bound": "mfma", "achieved": tf, "peak": rl.FP64_VECTOR_PEAK_TF, "unit": "TFLOP/s (useful f64: 2 x disk pixels x functions)",
                 "frac": tf / rl.FP64_VECTOR_PEAK_TF}
            if t["shader_clock_ghz"]:
                r["shader_clock_ghz"] = t["shader_clock_ghz"]
                r["frac_at_clock"] = r["frac"] * 2.4 / t["shader_clock_ghz"]
            return r

        oh = torch.empty((n_t, len(z20.n)), dtype=torch.float64, device=dev)
        t = _profiled(plan20, lambda: D.patch_moments_device(plan20, pt, out=oh))
        sec = {"workload": f"n_max=20 on 40-px windows (231 moments): {n_t} float32 windows as a batch; a 512-row band of a 2048x2048 frame densely",
               "kernels": kernels, "useful_fp64_flops_per_unit": useful,
               "batch": {"kernel_ms": t.ms, "patches_per_s": n_t / (t.ms * 1e-3), **_spread(t), "roofline": mfma_roofline(n_t, t)}}
        del oh, pt
        band = 512
        od = torch.empty((len(z20.n), band, 2048), dtype=torch.float64, device=dev)
        t = _profiled(plan20, lambda: D.frame_moments_device(plan20, fh, row0=256, n_rows=band, out=od))
        sec["dense"] = {"kernel_ms": t.ms, "positions_per_s": band * 2048 / (t.ms * 1e-3), **_spread(t), "roofline": mfma_roofline(band * 2048, t)}
        result["high_order"] = sec
        del od, fh, plan20, z20
        torch.cuda.empty_cache()

    # ---- configs[4]: full symmetry-map pipeline, 4096^2, n_max = 10, fused on device ---------------------------
    if not args.no_maps:
        z10 = ZPs(n_max=10, size=K)
        plan10 = z10._device_plan()
        big = torch.from_numpy(honeycomb_frame(4096, seed=1)).to(dev)
        theta = np.linspace(0, 2 * np.pi, 360, endpoint=False)
        n_c = rl.n_complex(10)
        if plan10.supports(_native.OP_MAPS, _native.ZK_F32):
            rot = torch.empty((4, 4096, 4096), dtype=torch.float64, device=dev)
            ab = torch.empty((n_c, 4096, 4096), dtype=torch.float64, device=dev)
            mir = torch.empty((4096, 4096), dtype=torch.float64, device=dev)
            t = _profiled(plan10, lambda: D.frame_maps_device(plan10, big, n_c, theta=theta, full=(rot, ab, mir)))
            ms = t.ms
            del rot, ab, mir
            mom = torch.empty((66, 4096, 4096), dtype=torch.float64, device=dev)
            t2 = _profiled(plan10, lambda: D.frame_moments_device(plan10, big, out=mom))
            ms2 = t2.ms
            del mom
            flops = rl.sep_flops_per_unit(z10.polynomials[0], 10) + rl.maps_tail_flops(10, 4, 360)
            result["symmetry_pipeline"] = {
                "workload": "configs[4]: 4096x4096 frame, 32-px, n_max=10 -> rot_maps[2,3,4,6] + 36 |Z_nm| planes + "
                            "mirror_map(360 angles), fused in one kernel",
                "kernel": "zk_frame_maps_kernel<10>", "fused_kernel_ms": ms, "positions_per_s": 4096 * 4096 / (ms * 1e-3), **_spread(t),
                "moments_only_kernel_ms": ms2, "moments_only": _spread(t2), "fp64_flops_per_position": flops,
                "out_bytes_fused": 41 * 4096 * 4096 * 8, "out_bytes_moments": 66 * 4096 * 4096 * 8,
                "roofline": _fp64_roofline(4096 * 4096 * flops, 4096 * 4096 * rl.dense_bytes_per_position(10, planes=41), ms,
                                           t["shader_clock_ghz"])}
        del big
        torch.cuda.empty_cache()

    # ---- NumPy in / NumPy out through the drop-in call (PCIe-inclusive; never `value`) ------------------------
    if not args.no_host_api:
        from mtflearn_amd.synthetic import sliding_patches
        n_side = H - K + 1
        rows = -(-500000 // n_side)
        batch = sliding_patches(frame, K, rows=range(rows))[:500000]
        z.transform(batch[:4096])
        best = 1e30
        for _ in range(3):
            t0 = time.perf_counter()
            zm = z.transform(batch)
            best = min(best, time.perf_counter() - t0)
        del zm
        # the same batch as a 16-bit detector image would deliver it: half the bytes over PCIe, widened on the device
        batch16 = np.round(batch * 60000).astype(np.uint16)
        z.transform(batch16[:4096])
        best16 = 1e30
        for _ in range(3):
            t0 = time.perf_counter()
            zm = z.transform(batch16)
            best16 = min(best16, time.perf_counter() - t0)
            del zm
        big = honeycomb_frame(4096, seed=2)
        z.transform(big)                                                 # first call: allocates the pinned result block
        bestf = 1e30
        for _ in range(2):
            t0 = time.perf_counter()
            zm = z.transform(big)
            bestf = min(bestf, time.perf_counter() - t0)
        del zm
        z.release()
        _native.pinned.trim()
        in_b, out_b = batch.nbytes, batch.shape[0] * n_poly * 8
        result["host_api"] = {
            "patches_per_s": batch.shape[0] / best, "batch": f"{batch.shape[0]} float32 {K}-px patches (pageable NumPy array) "
                                                             f"-> ZPs.transform -> (N, {n_poly}) float64",
            "batch_s": best, "batch_pcie_GBps": (in_b + out_b) / best / 1e9,
            "pcie_bound_patches_per_s": 63e9 / (K * K * 4),
            "patches_per_s_uint16": batch.shape[0] / best16, "batch_s_uint16": best16,
            "frame_4096_s": bestf, "frame": f"4096x4096 float32 frame -> ZPs.transform -> ({n_poly}, 4096, 4096) float64 "
                                            f"({n_poly * 4096 * 4096 * 8 / 1e9:.1f} GB, page-locked result from the pool)",
            "frame_pcie_GBps": n_poly * 4096 * 4096 * 8 / bestf / 1e9}


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch(args))
    worker(args)


if __name__ == "__main__":
    main()
