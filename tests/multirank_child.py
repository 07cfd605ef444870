"""One rank of tests/test_gpu_multirank.py: a process of its own on cuda:0 that joins an RCCL communicator of `world`
ranks through libzernike_hip.so (RcclComm, file rendezvous) and runs the PRODUCT's sharded drivers with the product's
kernels.  Every rank computes the whole result locally as well (one rank, no collective) and compares.

    python multirank_child.py RANK WORLD ID_FILE OUT_DIR
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "motif-learn_amd")):
    sys.path.insert(0, p)


def main(rank, world, id_file, out_dir):
    import warnings
    import torch
    from mtflearn_amd import ZPs, distributed as D, clustering as C
    from mtflearn_amd.features import pca

    comm = D.RcclComm(0, rank, world, path=id_file, timeout=150.0)
    report = {}
    try:
        assert (comm.rank, comm.world) == (rank, world)
        # ---- host-byte all-gather (timings, checksums, the k x D sums of sharded clustering) -----------------------
        assert comm.allgather_host(bytes([rank + 1]) * 7) == [bytes([r + 1]) * 7 for r in range(world)]
        big = lambda r: np.random.default_rng(r).integers(0, 255, 1_000_003, dtype=np.uint8).tobytes()   # grows the staging
        assert comm.allgather_host(big(rank)) == [big(r) for r in range(world)]
        assert comm.max_over_ranks(float(rank)) == float(world - 1)
        comm.barrier()

        dev = torch.device("cuda:0")
        stream = torch.cuda.current_stream().cuda_stream
        # ---- the collective alone: ragged rows, 19 planes (two group brackets), chunk windows, NaN where nobody wrote --
        planes, H, W = 19, 37, 11
        start, count, padded = D.shard_bounds(H, rank, world)
        ref = torch.arange(planes * H * W, dtype=torch.float64, device=dev).view(planes, H, W) + 0.5
        full = torch.full((planes, H, W), float("nan"), dtype=torch.float64, device=dev)
        full[:, start:start + count] = ref[:, start:start + count]
        for c0, c1 in D._chunk_bounds(padded, 3):
            comm.allgather_rows(full, planes, H, W, padded, c0, c1 - c0, stream)
        comm.join(stream)
        torch.cuda.synchronize()
        assert torch.equal(full, ref), "multi-plane ragged gather differs"
        # whole equal blocks of one plane (the ncclAllGather form under ZK_COMM_ALGO auto / allgather)
        H2 = 8 * world
        ref2 = torch.arange(H2 * 45, dtype=torch.float64, device=dev).view(H2, 45) - 3.0
        full2 = torch.full((H2, 45), float("nan"), dtype=torch.float64, device=dev)
        full2[8 * rank:8 * rank + 8] = ref2[8 * rank:8 * rank + 8]
        comm.allgather_rows(full2, 1, H2, 45, 8, 0, 8, stream)
        comm.join(stream)
        torch.cuda.synchronize()
        assert torch.equal(full2, ref2), "whole-block gather differs"

        # ---- the four drivers with the real kernels, against the same kernels on one rank ---------------------------
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            z = ZPs(8, 32)
        plan = z._device_plan()
        rng = np.random.default_rng(5)                                         # same data on every rank
        patches = torch.from_numpy(rng.random((1003, 32, 32), dtype=np.float32)).to(dev)
        one = D.patch_moments_device(plan, patches)
        start, count, _ = D.shard_bounds(1003, rank, world)
        for n_chunks in (1, 3):
            got = D.sharded_patch_moments(plan, comm, patches[start:start + count], 1003, n_chunks=n_chunks)
            torch.cuda.synchronize()
            # another batch size may select another kernel variant: equal to rounding, and own block bit-equal to a recomputation
            assert torch.allclose(got, one, rtol=0, atol=1e-13 * float(one.abs().max())), "patch moments differ"
        frame = torch.from_numpy(rng.random((150, 131), dtype=np.float32)).to(dev)
        one_f = D.frame_moments_device(plan, frame)
        for n_chunks in (1, 2):
            got = D.sharded_frame_moments(plan, comm, frame, n_chunks=n_chunks)
            torch.cuda.synchronize()
            assert torch.equal(got, one_f), "dense moments differ"          # one lane per output: bit-identical
        theta = np.linspace(0, 2 * np.pi, 360, endpoint=False)
        one_m = D.frame_maps_device(plan, frame, 25, folds=(2, 3, 4, 6), theta=theta)
        got_m = D.sharded_frame_maps(plan, comm, frame, 25, folds=(2, 3, 4, 6), theta=theta, n_chunks=3)
        torch.cuda.synchronize()
        for a, b, name in zip(got_m, one_m, ("rot", "abs", "mirror")):
            assert torch.equal(a, b), f"{name} maps differ"
        frames = torch.from_numpy(rng.random((5, 64, 80), dtype=np.float32)).to(dev)
        one_b = torch.stack([D.frame_moments_device(plan, f) for f in frames])
        start, count, _ = D.shard_bounds(5, rank, world)
        got_b = D.sharded_frames_moments(plan, comm, frames[start:start + count], 5)
        torch.cuda.synchronize()
        assert torch.equal(got_b, one_b), "frames differ"

        # ---- sharded consumers on the gathered matrix's blocks: labels of the whole matrix, no moment gather --------
        X = one.cpu().numpy()
        cuts = [D.shard_bounds(len(X), r, world)[0] for r in range(world)] + [len(X)]
        labels, centers, n_iter = C.kmeans_fit(X[cuts[rank]:cuts[rank + 1]], 4, random_state=0, comm=comm)
        whole = C.gather_labels(labels, comm)
        ref_labels, ref_centers, ref_iter = C.kmeans_fit(X, 4, random_state=0)
        assert np.array_equal(whole, ref_labels) and n_iter == ref_iter, "sharded k-means differs"
        scores = pca(X[cuts[rank]:cuts[rank + 1]], 3, comm=comm)
        ref_scores = pca(X, 3)
        np.testing.assert_allclose(scores, ref_scores[cuts[rank]:cuts[rank + 1]], rtol=0, atol=1e-10 * np.abs(ref_scores).max())
        comm.barrier()
        report["ok"] = True
    finally:
        comm.close()
    with open(os.path.join(out_dir, f"rank{rank}.ok"), "w") as f:
        f.write("ok\n")
    print(f"rank {rank}/{world} ok (ZK_COMM_ALGO={os.environ.get('ZK_COMM_ALGO', 'auto')})", flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4])
