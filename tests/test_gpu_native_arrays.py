"""The device entry points and the sharded drivers on the library's own device memory (``_native.DeviceArray``): no torch
anywhere in the process ("the device-resident and sharded helpers are hard-wired to torch tensors")."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent('''
    import os, sys, tempfile, warnings
    import numpy as np
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "motif-learn_amd"))
    from mtflearn_amd import ZPs, _native, distributed as D
    from mtflearn_amd.synthetic import honeycomb_frame, sliding_patches
    warnings.simplefilter("ignore")
    A = _native.DeviceArray
    frame = honeycomb_frame(160, 200, seed=4)
    z = ZPs(8, 32); plan = z._device_plan()
    patches = sliding_patches(frame, 32, rows=range(0, 128, 3), cols=range(0, 168, 5))
    ref_p, ref_f = z.transform(patches).data, z.transform(frame).data
    # views and copies
    a = A.from_numpy(np.arange(24.0).reshape(4, 3, 2))
    assert a[1:3].shape == (2, 3, 2) and a[2].shape == (3, 2) and a[-1][1].numpy().tolist() == [20.0, 21.0]
    assert np.array_equal(a[1:3].numpy(), np.arange(24.0).reshape(4, 3, 2)[1:3])
    # single-GPU entry points
    dp, df = A.from_numpy(patches), A.from_numpy(frame)
    assert np.array_equal(D.patch_moments_device(plan, dp).numpy(), ref_p)
    assert np.array_equal(D.frame_moments_device(plan, df).numpy(), ref_f)
    band = D.frame_moments_device(plan, df, row0=37, n_rows=50).numpy()
    assert np.array_equal(band, ref_f[:, 37:87])
    z10 = ZPs(10, 32); plan10 = z10._device_plan()
    theta = np.linspace(0, 2 * np.pi, 360, endpoint=False)
    rot, ab, mir = D.frame_maps_device(plan10, df, 36, theta=theta)
    maps = z10.symmetry_maps(frame)
    assert np.array_equal(rot.numpy(), maps["rot_maps"]) and np.array_equal(ab.numpy(), maps["abs"]) and np.array_equal(mir.numpy(), maps["mirror_map"])
    # the sharded drivers on the product's RCCL communicator (one rank)
    comm = D.RcclComm(0, 0, 1, path=os.path.join(tempfile.mkdtemp(), "id"))
    full = D.sharded_patch_moments(plan, comm, dp, len(patches), n_chunks=3)
    assert isinstance(full, A)
    np.testing.assert_allclose(full.numpy(), ref_p, rtol=1e-12, atol=1e-12 * np.abs(ref_p).max())   # (chunks may take another batch kernel)
    assert np.array_equal(D.sharded_frame_moments(plan, comm, df, n_chunks=5).numpy(), ref_f)
    rot, ab, mir = D.sharded_frame_maps(plan10, comm, df, 36, theta=theta, n_chunks=3)
    assert np.array_equal(rot.numpy(), maps["rot_maps"]) and np.array_equal(mir.numpy(), maps["mirror_map"])
    frames = np.stack([frame, frame[::-1].copy(), frame * 0.5])
    got = D.sharded_frames_moments(plan, comm, A.from_numpy(frames), 3).numpy()
    assert np.array_equal(got[1], z.transform(frames[1]).data) and np.array_equal(got[2], z.transform(frames[2]).data)
    comm.close()
    with np.testing.assert_raises(TypeError):
        D.patch_moments_device(plan, A.from_numpy(patches.astype(np.int32)))
    assert "torch" not in sys.modules, "torch was imported"
    print("native arrays ok")
''')


def test_drivers_run_without_torch():
    out = subprocess.run([sys.executable, "-c", f"ROOT = {ROOT!r}\n" + CHILD], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "native arrays ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
