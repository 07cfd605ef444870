"""Roofline accounting of the kernels in ``csrc/``: algorithmic bytes and the float64 operations
each kernel family actually executes per output unit (from the loop structure in ``zk_sep.h`` /
``zk_sep_strip.hip`` / ``zk_sep_maps.hip``).  Used by ``bench.py`` and the tools; no device code here."""
from __future__ import annotations

import numpy as np

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (~6.3 achievable)
XGMI_LINKS = 7                 # task statement / SURVEY 5: point-to-point xGMI, 7 links x ~153 GB/s per GPU
XGMI_LINK_GBS = 153.0
FP64_VECTOR_PEAK_TF = 78.6     # public datasheet figure (not in the local guide); 62 TF measured on register
                               # operands, 49-55 TF with SGPR-fed v_fma_f64 (profiles/r01_micro_sfma.txt)


def n_poly(n_max):
    return (n_max + 1) * (n_max + 2) // 2


def n_complex(n_max):
    return sum(n // 2 + 1 for n in range(n_max + 1))


def batch_bytes_per_patch(size, n_max, in_bytes=4):
    """Mode A (reference row 3): the patch is read once, every moment written once."""
    return size * size * in_bytes + 8 * n_poly(n_max)


def dense_bytes_per_position(n_max, in_bytes=4, planes=None):
    """Mode B (reference row 4): one pixel in, ``planes`` (default N_poly) float64 out."""
    return in_bytes + 8 * (n_poly(n_max) if planes is None else planes)


def _t_terms(n_max):
    """Entries of the packed class-blocked T matrix: (j, (a, b)) exists only for a + b <= n_j and matching parity."""
    total = 0
    for pq, sel in (((0, 0), lambda m: m >= 0 and m % 2 == 0), ((1, 0), lambda m: m >= 0 and m % 2 == 1),
                    ((0, 1), lambda m: m < 0 and m % 2 == 1), ((1, 1), lambda m: m < 0 and m % 2 == 0)):
        degs = [a + b for a in range(n_max + 1) for b in range(n_max + 1 - a) if (a % 2, b % 2) == pq]
        orders = [n for n in range(n_max + 1) for m in range(-n, n + 1, 2) if sel(m)]
        total += sum(1 for n in orders for d in degs if d <= n)
    return total


def _disk_geometry(mask):
    K = mask.shape[0]
    Q = (K + 1) // 2
    quad_px = int(np.count_nonzero(mask[:Q, :Q]))
    row_pairs = int(mask[:Q].any(axis=1).sum())
    return K, Q, quad_px, row_pairs


def sep_flops_per_unit(basis0, n_max):
    """Folded row-separable form (``zk_frame_sep_kernel``, ``zk_patch_sep_kernel``): per quadrant disk pixel
    8 adds (mirror folds) + 2 (n_max + 1) FMAs, per disk row pair N_poly FMAs, one packed T product."""
    _, _, quad_px, row_pairs = _disk_geometry(basis0 != 0)
    return quad_px * (8 + 4 * (n_max + 1)) + 2 * row_pairs * n_poly(n_max) + 2 * _t_terms(n_max)


def strip_flops_per_unit(basis0, n_max):
    """``zk_frame_strip_kernel`` (n_max <= 8): per PAIR of vertically adjacent outputs every frame row is swept
    once from the centre to the wider of the two inner limits (2 adds + (n_max + 1) FMAs per column pair), every
    disk row of either output costs N_poly FMAs, and there are two T products."""
    mask = basis0 != 0
    K, Q, _, _ = _disk_geometry(mask)
    cmin = [int(np.argmax(mask[r, :Q])) if mask[r, :Q].any() else Q for r in range(K)]
    sweep_cols = sum(Q - min(cmin[fr] if fr < K else Q, cmin[fr - 1] if fr > 0 else Q) for fr in range(K + 1))
    disk_rows = sum(1 for c in cmin if c < Q)
    return (sweep_cols * (2 + 2 * (n_max + 1)) + 2 * disk_rows * 2 * n_poly(n_max) + 2 * 2 * _t_terms(n_max)) / 2


def strip2_flops_per_unit(basis0, n_max):
    """``zk_frame_strip2_kernel`` (round 3; even window sizes, n_max <= 12): per PAIR of vertically adjacent outputs every
    frame row is swept once from the centre to the wider of the two inner limits -- per column pair a sum, a difference, one
    add for degree 0 (P_0 = 1 is implicit) and n_max FMAs; with n_max > 8 the even and the odd degrees are done in two passes
    over the same tile, which executes the same operations -- then every frame row that holds a disk pixel of either output
    costs two row steps of N_poly slots (n_max + 1 of them adds: P_0(y) = 1), and both outputs share one T product's operands."""
    mask = basis0 != 0
    K, Q, _, _ = _disk_geometry(mask)
    cmin = [int(np.argmax(mask[r, :Q])) if mask[r, :Q].any() else Q for r in range(K)]
    limits = [min(cmin[fr] if fr < K else Q, cmin[fr - 1] if fr > 0 else Q) for fr in range(K + 1)]
    sweep_cols = sum(Q - c for c in limits)
    live_rows = sum(1 for c in limits if c < Q)
    row_step = 2 * n_poly(n_max) - (n_max + 1)
    return (sweep_cols * (3 + 2 * n_max) + 2 * live_rows * row_step + 2 * 2 * _t_terms(n_max)) / 2


def strip2_available(size, n_max):
    """Mirror of ``zk_sep_strip_available`` for the round-3 form (bench labels only)."""
    return size % 2 == 0 and n_max <= 12 and (size + 7) * (size + 63) * 8 <= 80 * 1024


def stream_flops_per_unit(basis0, n_max):
    """``zk_patch_stream_kernel``: no mirror folds -- per disk pixel n_max FMAs + 1 add, per disk row N_poly FMAs."""
    mask = basis0 != 0
    return int(mask.sum()) * (2 * n_max + 1) + 2 * int(mask.any(axis=1).sum()) * n_poly(n_max) + 2 * _t_terms(n_max)


def maps_tail_flops(n_max, n_folds, n_theta, theta_sym=True, want_abs=True):
    """``zk_maps_tail`` (FMA = 2, add / max / sqrt = 1): per complex moment 10 operations (+ a square root when |Z|
    is written), the unselect / norm pass, ``n_folds`` weighted sums over |m|, and the mirror scan.  On a uniform
    angle grid with n_theta % 8 == 0 the scan visits n_theta/8 + 1 angles (executed: rounded up to a multiple of 4);
    per angle 2 (n_max - 1) FMAs of the cos / sin recurrence, 2 FMAs per even m and 4 per odd m, 8 + 8 adds and 4
    add-max pairs for its eight scores; otherwise 2 n_max FMAs and one select per angle."""
    nc = n_complex(n_max)
    flops = nc * (10 + (1 if want_abs else 0)) + 4 * (n_max + 1) + n_folds * (2 * (n_max + 1) + 1)
    if n_theta:
        if theta_sym and n_theta % 8 == 0:
            rows = (n_theta // 8 + 4) & ~3
            n_odd = (n_max + 1) // 2
            fmas = 2 * (n_max - 1) + 2 * (n_max - n_odd) + 4 * n_odd
            flops += rows * (2 * fmas + 1 + 8 + 8 + 8) + 4 * n_max
        else:
            flops += n_theta * (4 * n_max + 1)
    return flops


def direct_flops_per_unit(basis0, n_max):
    """What the definition costs without any folding: 2 * disk_px * N_poly."""
    return 2 * int((basis0 != 0).sum()) * n_poly(n_max)
