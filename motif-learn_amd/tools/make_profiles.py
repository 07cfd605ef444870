#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of tools/profile_round.sh (gpurun_out/<tag>_*) into the summaries under profiles/:
<tag>_kernel_stats_bench.csv, <tag>_bench_line_under_rocprof.json, <tag>_pmc_hbm.txt (+ profiles/traffic.json),
<tag>_sq_counters.txt.   python3 motif-learn_amd/tools/make_profiles.py r02"""
import collections
import csv
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
TOOL = os.path.join(ROOT, "motif-learn_amd", "tools", "pmc_summary.py")

rows = list(csv.DictReader(open(os.path.join(G, f"{TAG}_stats", "b_kernel_stats.csv"))))
with open(os.path.join(P, f"{TAG}_kernel_stats_bench.csv"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-4096 --no-cpu-baseline --no-host-api   (1 x MI355X)\n")
    f.write(f"# zk_* kernels only (torch copy / fill kernels of the harness omitted); bench line of the same run: profiles/{TAG}_bench_line_under_rocprof.json\n")
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
    for r in rows:
        if "zk_" in r["Name"]:
            m = re.search(r"zk_\w+(?:<[^(]*>)?", r["Name"])
            w.writerow([m.group(0) if m else r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"]])
json.dump(json.loads(open(os.path.join(G, f"{TAG}_bench_under_rocprof.json")).read()),
          open(os.path.join(P, f"{TAG}_bench_line_under_rocprof.json"), "w"))

out = subprocess.check_output([sys.executable, TOOL, os.path.join(G, f"{TAG}_pmc_FETCH_SIZE"), os.path.join(G, f"{TAG}_pmc_WRITE_SIZE"),
                               "--match", "zk_patch", "--traffic-json", os.path.join(P, "traffic.json"), "--key", "patches_32_8_2048",
                               "--kernel", "zk_patch_sep_kernel<8, 8",
                               "--source", f"profiles/{TAG}_pmc_hbm.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE doubled per the gfx950 correction)"],
                              text=True)
rec = json.load(open(os.path.join(P, "traffic.json")))["patches_32_8_2048"]
with open(os.path.join(P, f"{TAG}_pmc_hbm.txt"), "w") as f:
    f.write(f"""# {TAG}: HBM traffic of the timed batch kernel from rocprofv3 PMC, separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes
# command: rocprofv3 --pmc <COUNTER> --output-format csv -- python3 bench.py --only-timed-loop --steps 3 --warmup 1
#          (motif-learn_amd/tools/profile_round.sh {TAG}; summary by motif-learn_amd/tools/make_profiles.py)
# units: FETCH_SIZE / WRITE_SIZE in KiB.  gfx950 correction (MI355X_MICROARCH.md, "HBM"): FETCH_SIZE reports 1/2 of a wide
# coalesced read stream (calibrated in round 1 on a known 4-GiB stream, profiles/r01_pmc_hbm.txt) -> doubled below.
""")
    f.write("\n".join(l for l in out.splitlines() if l.startswith("zk_")) + "\n")
    n = 4068289
    f.write(f"""
# zk_patch_sep_kernel<8,8,float>, N = 4 068 289 patches (2048^2 frame, 32-px, n_max = 8):
#   read  = 2 x FETCH_SIZE x 1024 = {rec['read_bytes']:.4e} B  (= {rec['read_bytes'] / n:.1f} B/patch: 30 of 32 patch rows x 128 B; rows 0 and 31 lie outside the disk and are never fetched)
#   write = WRITE_SIZE x 1024     = {rec['write_bytes']:.4e} B  (= {rec['write_bytes'] / n:.1f} B/patch: 45 moments x 8 B)
#   traffic = {rec['hbm_bytes_per_launch']:.4e} B per launch vs algorithmic 4456 B x N = {4456 * n:.4e} B  -> no wasted re-reads ({rec['hbm_bytes_per_launch'] / (4456 * n):.2f} x algorithmic)
# recorded for bench.py in profiles/traffic.json together with the sha256 of the batch kernel sources (zk_sep_patches.hip, zk_sep.h, zk_sep.hip, zk_fold.h, zk_internal.h) it was taken on
# (kernel_source_sha {rec['kernel_source_sha']}, git {rec['git']}).
""")

sq = subprocess.check_output([sys.executable, TOOL, os.path.join(G, f"{TAG}_sq1"), os.path.join(G, f"{TAG}_sq2"), "--match", "zk_"], text=True)
vals = collections.defaultdict(dict)
for line in sq.splitlines():
    m = re.match(r"(\S.*?)\s+(SQ_\w+)\s+launches=\s*(\d+) mean=\s*([\d.]+)", line)
    if m:
        vals[m.group(1).strip()][m.group(2)] = float(m.group(4))
waves = {"zk_frame_strip_kernel<8, float>": 2, "zk_frame_strip2_kernel<8, float, 16>": 2, "zk_frame_strip2_kernel<12, float, 32>": 2,
         "zk_frame_sep_kernel<12, float, 15>": 2, "zk_frame_maps_kernel<10, float>": 2,
         "zk_patch_sep_kernel<12, 8, float, true, 15>": 1, "zk_patch_stream_kernel<12, float>": 2}
with open(os.path.join(P, f"{TAG}_sq_counters.txt"), "w") as f:
    f.write(f"""# {TAG}: SQ counters of the dense / maps / (64, 12) batch kernels (rocprofv3 --pmc, two counter sets in separate passes;
# motif-learn_amd/tools/profile_round.sh -> tools/run_dense.py; means over 3 launches, summed over the chip).
# Workloads: strip2<8>: 2048^2 frame, 32-px; strip2<12> (round 3: two passes by x parity; round 2 ran frame_sep<12> here):
#            4096^2, 64-px; maps<10>: 4096^2, 32-px, all outputs; patch_sep<12,...,wide> / stream<12>: 1.0 M 64-px float32 patches.
# Units: SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count in quad-cycles per wave (a wave64 VALU instruction occupies its
# SIMD for one quad = 4 clocks, so SQ_ACTIVE_INST_VALU == SQ_INSTS_VALU here).
# derived:  valu_per_wave = ACTIVE_INST_VALU / WAVE_CYCLES  (fraction of a wave's life spent issuing VALU)
#           simd_valu_busy ~= valu_per_wave x resident waves per SIMD (from the launch bounds)
#           salu:valu = INSTS_SALU / INSTS_VALU ;  wait_any = WAIT_ANY / WAVE_CYCLES ; wait_inst = WAIT_INST_ANY / WAVE_CYCLES
# Round 2 (profiles/r02_sq_counters.txt), same workloads: strip<8> simd_valu_busy 0.625, salu:valu 0.395, wait_any 0.326;
#   frame_sep<12> 0.81 / 0.18 / 0.30; maps<10> 0.81; patch_sep<12, wide> 0.53 (one wave per SIMD).
#
""")
    f.write(f"{'kernel':46s} {'waves/SIMD':>10s} {'valu_per_wave':>13s} {'simd_valu_busy':>14s} {'salu:valu':>9s} {'wait_any':>8s} {'wait_inst':>9s} {'lds_conflict':>12s} {'smem/valu':>9s}\n")
    for k, v in vals.items():
        wc, w = v["SQ_WAVE_CYCLES"], waves.get(k, 1)
        f.write(f"{k:46s} {w:10d} {v['SQ_ACTIVE_INST_VALU'] / wc:13.3f} {min(1.0, w * v['SQ_ACTIVE_INST_VALU'] / wc):14.3f} "
                f"{v['SQ_INSTS_SALU'] / v['SQ_INSTS_VALU']:9.3f} {v['SQ_WAIT_ANY'] / wc:8.3f} {v['SQ_WAIT_INST_ANY'] / wc:9.3f} "
                f"{v['SQ_LDS_BANK_CONFLICT']:12.0f} {v['SQ_INSTS_SMEM'] / v['SQ_INSTS_VALU']:9.3f}\n")
    f.write("\n# raw means\n" + sq)
cl = os.path.join(G, f"{TAG}_cluster", "c_kernel_stats.csv")
if os.path.exists(cl):
    log = [l.rstrip() for l in open(os.path.join(G, f"{TAG}_cluster.log")) if l.startswith(("matrix", "  ", "kmeans_fit", "gmm_fit"))]
    with open(os.path.join(P, f"{TAG}_cluster_kernel_stats.csv"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 motif-learn_amd/tools/time_clustering.py   (1 x MI355X)\n")
        f.write("# clustering consumers (csrc/zk_cluster.hip) on a resident 4 068 289 x 45 float64 matrix (1.46 GB), k = 6; the tool's own\n")
        f.write("# wall-clock lines (synchronous C-ABI calls: table upload + kernel + reduction + a few hundred bytes back):\n")
        for l in log:
            f.write("#   " + l + "\n")
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
        for r in csv.DictReader(open(cl)):
            if "_kernel" in r["Name"]:
                name = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
                w.writerow([name, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"]])
    print(open(os.path.join(P, f"{TAG}_cluster_kernel_stats.csv")).read())
print(open(os.path.join(P, f"{TAG}_kernel_stats_bench.csv")).read())
print(open(os.path.join(P, f"{TAG}_sq_counters.txt")).read().split("# raw means")[0])
