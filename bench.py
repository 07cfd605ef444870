#!/usr/bin/env python3
"""bench.py -- Zernike-moment hot path on MI355X (contract: see the task brief / DESIGN.md section 7).

A "step" is one pass of the batch-of-patches hot path (``ZPs.transform`` on a 3-D batch, reference
``mtflearn/features/_zps.py:146-157``) over every dense 32-px sliding window of one synthetic
2048 x 2048 STEM-like frame per GPU (BASELINE.json configs[1]: 4 068 289 patches, n_max = 8),
float32 patches resident in HBM, float64 moments out.  With N > 1 ranks every rank owns its own
frame (weak scaling, no data-path collective: every output depends on one window only).  The
single RCCL all-gather that reassembles the (N_total, 45) moment matrix on every rank is result
assembly, not part of the per-patch computation; it is measured in the same run by a second timed
loop of the same K steps with the all-gather inside each step (pipelined against the next step's
kernel on RCCL's stream) and reported under "allgather" -- ``--allgather-in-step`` makes that loop
the one `value` is taken from.

Rank 0 prints ONE JSON line.  Besides the contract keys it carries
  roofline      -- the batch kernel: algorithmic bytes / HIP-event kernel time vs the 8 TB/s HBM peak
  cpu_baseline  -- the oracle's restatement of the reference CPU path (same NumPy call) on this host
  dense_frame   -- the dense-frame kernel (reference _zps.py:159-193) on the same frame, for context
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "motif-learn_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VECTOR_PEAK_TF = 78.6     # public datasheet figure (not in the local guide); 62 TF measured,
                               # tools/micro_sfma.hip, profiles/r01_micro_sfma.txt


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frame", type=int, default=2048, help="frame side (configs[1]: 2048)")
    ap.add_argument("--size", type=int, default=32)
    ap.add_argument("--n-max", type=int, default=8)
    ap.add_argument("--allgather-in-step", action="store_true",
                    help="N>1: take `value` from the loop whose steps include the all-gather")
    ap.add_argument("--no-allgather", action="store_true", help="N>1: skip the all-gather measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dense", action="store_true", help="skip the dense-frame side measurement")
    ap.add_argument("--no-maps", action="store_true", help="skip the configs[4] symmetry-map side measurement")
    ap.add_argument("--with-4096", action="store_true",
                    help="also time the north star's own size (all windows of a 4096^2 frame, 67.7 GB) -- opt-in: it "
                         "launches the same kernel as the timed loop and would skew a rocprofv3 --stats average")
    return ap.parse_args()


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
    from oracle import zernike_oracle as zo
    crop = np.ascontiguousarray(frame[:1024, :1024])
    t0 = time.perf_counter()
    zo.moments_frame_fft(crop, z.polynomials, z.n)
    dt = time.perf_counter() - t0
    return {"value": crop.size / dt, "unit": "patches/s", "cores": 1, "kind": "port",
            "sample": f"oracle moments_frame_fft (scipy fftconvolve, single-threaded) on a 1024x1024 crop, {dt:.2f} s"}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with python -m torch.distributed.run --nproc-per-node N")
        args.gpus = world
    # Rehearsal knobs (a 1-GPU box cannot run RCCL between two ranks): ZK_BENCH_BACKEND=gloo with
    # ZK_BENCH_ONE_DEVICE=1 puts every rank on device 0 and exercises the same control flow.
    backend = os.environ.get("ZK_BENCH_BACKEND", "nccl")
    if os.environ.get("ZK_BENCH_ONE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    os.environ["MTFLEARN_AMD_DEVICE"] = str(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from mtflearn_amd import ZPs, _native
    from mtflearn_amd.synthetic import honeycomb_frame
    from mtflearn_amd.distributed import patch_moments_device, frame_moments_device

    K, H = args.size, args.frame
    z = ZPs(n_max=args.n_max, size=K)
    plan = z._device_plan()
    n_poly = len(z.n)
    frame = honeycomb_frame(H, seed=rank)                               # one frame per rank (weak scaling)
    f_dev = torch.from_numpy(frame).to(dev)
    patches = f_dev.unfold(0, K, 1).unfold(1, K, 1).reshape(-1, K, K).contiguous()
    n_local = patches.shape[0]
    gather = world > 1 and not args.no_allgather
    outs = [torch.empty((n_local, n_poly), dtype=torch.float64, device=dev) for _ in range(2 if gather else 1)]
    fulls = [torch.empty((world * n_local, n_poly), dtype=torch.float64, device=dev) for _ in range(2)] if gather else []
    fast = plan.has_path(0, _native.ZK_F32, _native.PATH_SEPARABLE)

    pending = [None, None]

    def step(i, with_gather):
        b = i & 1 if with_gather else 0
        if with_gather and pending[b] is not None:
            pending[b].wait()                                            # buffer pair b is free again
        patch_moments_device(plan, patches, out=outs[b])
        if with_gather:
            pending[b] = dist.all_gather_into_tensor(fulls[b], outs[b], async_op=True)

    def drain():
        for b in range(2):
            if pending[b] is not None:
                pending[b].wait()
                pending[b] = None

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_loop(with_gather):
        """W untimed + exactly K timed steps, barrier + synchronize on both sides, max over ranks."""
        for i in range(args.warmup):
            step(i, with_gather)
        drain()
        fence()
        plan.profile(True)
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i, with_gather)
        drain()
        fence()
        dt = time.perf_counter() - t0
        launches, kernel_ms = plan.profile_read()
        plan.profile(False)
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.item(), launches, kernel_ms

    elapsed, launches, kernel_ms = timed_loop(False)
    in_step = False
    gathered_ok = allgather_ms = elapsed_gather = None
    if gather:
        elapsed_gather, l2, k2 = timed_loop(True)
        last = (args.steps - 1) & 1
        mine = fulls[last][rank * n_local:(rank + 1) * n_local]
        ok = torch.tensor([int(torch.equal(mine, outs[last]))], device=dev)
        # every rank's block must have arrived: the first entry of each block is finite and non-zero
        heads = fulls[last][::n_local, 0]
        ok &= int(bool(torch.isfinite(heads).all() and (heads != 0).all()))
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        gathered_ok = bool(ok.item())
        fence()
        t1 = time.perf_counter()
        for _ in range(3):
            dist.all_gather_into_tensor(fulls[0], outs[0])
        fence()
        allgather_ms = (time.perf_counter() - t1) / 3 * 1e3
        if args.allgather_in_step:
            elapsed, launches, kernel_ms, in_step = elapsed_gather, l2, k2, True

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = world * n_local / (elapsed / args.steps)
    kern_ms = kernel_ms / max(launches, 1)
    alg_bytes = n_local * (K * K * 4 + 8 * n_poly)                       # K^2 s_in + 8 N_poly per patch
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            rec = json.load(open(tpath)).get(f"patches_{K}_{args.n_max}_{H}")
            traffic = rec and rec.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
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
                   "parallelism": f"dp{world} (one frame's patch batch per GPU, no data-path collective; moment "
                                  f"matrix reassembled by one RCCL all-gather, see 'allgather')"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel_ms": kern_ms, "launches": launches,
                     "algorithmic_bytes_per_patch": K * K * 4 + 8 * n_poly},
        "kernel_only_patches_per_s": world * n_local / (kern_ms * 1e-3),
    }
    if allgather_ms is not None:
        result["allgather"] = {"ms_alone": allgather_ms, "verified": gathered_ok,
                               "ms_per_step_with_allgather_in_step": elapsed_gather / args.steps * 1e3,
                               "value_with_allgather_in_step": world * n_local / (elapsed_gather / args.steps),
                               "bytes_per_rank_out": n_local * n_poly * 8,
                               "gathered_bytes": world * n_local * n_poly * 8}

    # ---- dense-frame kernels on the same frame (side measurement, not `value`) -----------------------
    if not args.no_dense:
        out_f = frame_moments_device(plan, f_dev)
        torch.cuda.synchronize()
        npx, disk = H * H, plan.disk_pixels
        # f64 operations the separable kernel executes per position (zk_sep.h): per quadrant disk pixel
        # 8 adds + 2(n_max+1) FMAs, per disk row pair N_poly FMAs, one class-blocked T product
        quad_px = int(np.count_nonzero(z.polynomials[0][:(K + 1) // 2, :(K + 1) // 2]))
        row_pairs = int(np.any(z.polynomials[0][:(K + 1) // 2] != 0, axis=1).sum())
        cls = [sum(1 for a in range(args.n_max + 1) for b in range(args.n_max + 1 - a) if (a % 2, b % 2) == pq)
               for pq in ((0, 0), (1, 0), (0, 1), (1, 1))]
        # (T is stored packed: entry (j, (a,b)) exists only for a + b <= n_j)
        t_terms = 0
        for pq, m_sel in (((0, 0), lambda m: m >= 0 and m % 2 == 0), ((1, 0), lambda m: m >= 0 and m % 2 == 1),
                          ((0, 1), lambda m: m < 0 and m % 2 == 1), ((1, 1), lambda m: m < 0 and m % 2 == 0)):
            degs = [a + b for a in range(args.n_max + 1) for b in range(args.n_max + 1 - a) if (a % 2, b % 2) == pq]
            zns = [n for n in range(args.n_max + 1) for m in range(-n, n + 1, 2) if m_sel(m)]
            t_terms += sum(1 for n in zns for d in degs if d <= n)
        sep_flops = quad_px * (8 + 4 * (args.n_max + 1)) + 2 * row_pairs * n_poly + 2 * t_terms
        # strip kernel (zk_sep_strip.hip; n_max <= 8, windows <= 65 px): per PAIR of outputs every frame row is swept
        # once from the centre to the wider of the two inner limits (2 adds + (n_max+1) FMAs per column pair), every
        # disk row of either output costs N_poly FMAs, and there are two T products
        strip = args.n_max <= 8 and (K + 7) * (K + 63) * 8 <= 80 * 1024 and not os.environ.get("ZK_NO_STRIP")
        if strip:
            Qh = (K + 1) // 2
            mask = z.polynomials[0] != 0
            cmin = [int(np.argmax(mask[r, :Qh])) if mask[r, :Qh].any() else Qh for r in range(K)]
            sweep_cols = sum(Qh - min(cmin[fr] if fr < K else Qh, cmin[fr - 1] if fr > 0 else Qh) for fr in range(K + 1))
            disk_rows = sum(1 for c in cmin if c < Qh)
            sep_flops = (sweep_cols * (2 + 2 * (args.n_max + 1)) + 2 * disk_rows * 2 * n_poly + 2 * 2 * t_terms) / 2
        dense = {"positions": npx, "bound": "fp64-valu", "fp64_vector_peak_TFLOPs": FP64_VECTOR_PEAK_TF,
                 "kernels": {}}
        for path in (_native.PATH_SEPARABLE, _native.PATH_FOLDED):
            if not plan.has_path(1, _native.ZK_F32, path):
                continue
            plan.set_path(path)
            frame_moments_device(plan, f_dev, out=out_f)
            torch.cuda.synchronize()
            plan.profile(True)
            for _ in range(5):
                frame_moments_device(plan, f_dev, out=out_f)
            torch.cuda.synchronize()
            ln, ms = plan.profile_read()
            plan.profile(False)
            fms = ms / ln
            dense["kernels"][_native.PATH_NAMES[path]] = {
                "patches_per_s": npx / (fms * 1e-3), "kernel_ms": fms,
                "hbm_GBps_algorithmic": npx * (4 + 8 * n_poly) / (fms * 1e-3) / 1e9,
                "hbm_frac": npx * (4 + 8 * n_poly) / (fms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "fp64_TFLOPs_direct_equiv": npx * 2.0 * disk * n_poly / (fms * 1e-3) / 1e12}
            if path == _native.PATH_SEPARABLE:
                ex = npx * sep_flops / (fms * 1e-3) / 1e12
                dense["kernels"]["separable"].update({"fp64_TFLOPs_executed": ex,
                                                      "fp64_frac_of_peak": ex / FP64_VECTOR_PEAK_TF,
                                                      "kernel": "zk_frame_strip_kernel" if strip else "zk_frame_sep_kernel",
                                                      "fp64_flops_per_position": sep_flops})
        plan.set_path(_native.PATH_AUTO)
        result["dense_frame"] = dense
        del out_f

    # ---- configs[4]: full symmetry-map pipeline, 4096^2, n_max = 10, fused on device (side measurement) ---
    if not args.no_maps and world == 1:
        from mtflearn_amd.distributed import frame_maps_device
        del patches, outs
        torch.cuda.empty_cache()
        z10 = ZPs(n_max=10, size=K)
        plan10 = z10._device_plan()
        big = torch.from_numpy(honeycomb_frame(4096, seed=1)).to(dev)
        theta = np.linspace(0, 2 * np.pi, 360, endpoint=False)
        n_c = sum(n // 2 + 1 for n in range(11))
        if plan10.has_path(1, _native.ZK_F32, _native.PATH_SEPARABLE):
            frame_maps_device(plan10, big, n_c, theta=theta)
            torch.cuda.synchronize()
            plan10.profile(True)
            for _ in range(3):
                frame_maps_device(plan10, big, n_c, theta=theta)
            torch.cuda.synchronize()
            ln, ms = plan10.profile_read()
            mom = frame_moments_device(plan10, big)
            torch.cuda.synchronize()
            for _ in range(3):
                frame_moments_device(plan10, big, out=mom)
            torch.cuda.synchronize()
            ln2, ms2 = plan10.profile_read()
            plan10.profile(False)
            result["symmetry_pipeline"] = {
                "workload": "configs[4]: 4096x4096 frame, 32-px, n_max=10 -> rot_maps[2,3,4,6] + 36 |Z_nm| planes + "
                            "mirror_map(360 angles), fused in one kernel",
                "fused_kernel_ms": ms / ln, "positions_per_s": 4096 * 4096 / (ms / ln * 1e-3),
                "moments_only_kernel_ms": ms2 / ln2,
                "out_bytes_fused": 41 * 4096 * 4096 * 8, "out_bytes_moments": 66 * 4096 * 4096 * 8}
            del mom
        del big

    # ---- north star's own size: all dense 32-px windows of a 4096^2 frame as one batch (side measurement) --
    if world == 1 and args.with_4096 and K == 32:
        try:
            del patches, outs                                            # (already gone if the maps section ran)
        except NameError:
            pass
        torch.cuda.empty_cache()
        free_b, _tot = torch.cuda.mem_get_info()
        n4 = (4096 - K + 1) ** 2
        need = n4 * (K * K * 4 + 8 * n_poly) + 4096 * 4096 * 4
        if free_b > need * 1.1:
            f4 = torch.from_numpy(honeycomb_frame(4096, seed=2)).to(dev)
            p4 = f4.unfold(0, K, 1).unfold(1, K, 1).reshape(-1, K, K).contiguous()
            o4 = torch.empty((n4, n_poly), dtype=torch.float64, device=dev)
            patch_moments_device(plan, p4, out=o4)
            torch.cuda.synchronize()
            plan.profile(True)
            for _ in range(5):
                patch_moments_device(plan, p4, out=o4)
            torch.cuda.synchronize()
            ln4, ms4 = plan.profile_read()
            plan.profile(False)
            k4 = ms4 / ln4
            result["north_star_4096"] = {
                "workload": f"all {n4} dense {K}-px windows of a 4096x4096 frame as one float32 batch ({n4 * K * K * 4 / 1e9:.1f} GB), n_max={args.n_max}",
                "kernel_ms": k4, "patches_per_s": n4 / (k4 * 1e-3),
                "hbm_GBps_algorithmic": n4 * (K * K * 4 + 8 * n_poly) / (k4 * 1e-3) / 1e9,
                "hbm_frac": n4 * (K * K * 4 + 8 * n_poly) / (k4 * 1e-3) / 1e9 / HBM_PEAK_GBS}
            del p4, o4, f4
            torch.cuda.empty_cache()

    if world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(z, frame, K)
        result["cpu_baseline_dense"] = cpu_dense_baseline(z, frame)
        result["gpu_over_cpu"] = value / result["cpu_baseline"]["value"]
    print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
