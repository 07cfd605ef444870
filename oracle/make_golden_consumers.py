#!/usr/bin/env python3
"""Generate tests/golden/consumers_golden.npz from the REFERENCE's clustering wrappers (build container only).

TEST INFRASTRUCTURE.  ``mtflearn/clustering/_clustering_functions.py`` imports scikit-image at module level (for
``seg_lbs``), which is not installed here.  The module is loaded under a stub package with EMPTY stand-in modules for
``skimage.morphology`` / ``skimage.measure`` whose names raise if called (they only satisfy the import statement, as in
oracle/make_golden_pickers.py).  Recorded: the labels ``kmeans_lbs``, ``gmm_lbs`` and ``sort_lbs`` return on small seeded
matrices (scikit-learn does the arithmetic there; the reference adds the relabelling by cluster size).
No reference source or bytecode is copied; the fixtures are data.

Usage:  python oracle/make_golden_consumers.py
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "consumers_golden.npz")


def import_reference():
    sys.dont_write_bytecode = True
    for name in ("mtflearn", "mtflearn.clustering"):
        mod = types.ModuleType(name)
        mod.__path__ = [os.path.join(REF, *name.split("."))]
        sys.modules[name] = mod

    def absent(*a, **k):
        raise RuntimeError("scikit-image is not installed: this stand-in only satisfies the import")
    for name, attrs in (("skimage", ()), ("skimage.morphology", ("disk", "dilation")), ("skimage.measure", ("label",))):
        mod = types.ModuleType(name)
        for attr in attrs:
            setattr(mod, attr, absent)
        sys.modules[name] = mod
    from mtflearn.clustering import _clustering_functions
    return _clustering_functions


def blobs(rng, sizes, d, spread):
    """Gaussian blobs of unequal sizes around random centres (seeded; the matrix is stored in the fixture)."""
    centres = rng.standard_normal((len(sizes), d)) * 4.0
    parts = [c + rng.standard_normal((s, d)) * spread * (0.5 + rng.random(d)) for c, s in zip(centres, sizes)]
    x = np.concatenate(parts)
    return x[rng.permutation(len(x))]


def main():
    cf = import_reference()
    rng = np.random.default_rng(20261004)
    g = {}
    g["Xa"] = blobs(rng, (500, 300, 150, 50), 10, 0.6)                # 1000 x 10, four clusters of distinct sizes
    g["Xb"] = blobs(rng, (400, 250, 100), 45, 1.0)                    # 750 x 45: the width of an n_max = 8 moment matrix
    for key, n in (("Xa", 4), ("Xa", 3), ("Xb", 3), ("Xb", 5)):
        g[f"kmeans_{key}_{n}"] = cf.kmeans_lbs(g[key], n)
        g[f"kmeans_{key}_{n}_rs7"] = cf.kmeans_lbs(g[key], n, random_state=7)
    for key, n, kind in (("Xa", 4, "full"), ("Xa", 3, "full"), ("Xb", 3, "full"), ("Xa", 4, "diag"), ("Xa", 4, "tied"),
                         ("Xa", 4, "spherical")):
        g[f"gmm_{key}_{n}_{kind}"] = cf.gmm_lbs(g[key], n, type=kind)
    lbs = rng.integers(0, 6, 400) * 3 + 2
    g["sort_in"] = lbs
    g["sort_out"] = cf.sort_lbs(lbs)
    np.savez_compressed(OUT, **g)
    print("wrote", OUT, {k: v.shape for k, v in g.items()})


if __name__ == "__main__":
    main()
