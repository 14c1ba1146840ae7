#!/usr/bin/env python3
"""Copies the outputs of profiles/tools/r03_refresh.sh (gpurun_out/r03_refresh/) into profiles/ under their r03_ names
and rewrites profiles/traffic_minsum.json, profiles/traffic_rs.json from the FETCH_SIZE / WRITE_SIZE passes."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "gpurun_out", "r03_refresh")
DST = os.path.join(ROOT, "profiles")


def cp(src, dst, filt=None):
    with open(os.path.join(SRC, src)) as f:
        text = f.read()
    lines = [l for l in text.splitlines() if "amdgpu.ids" not in l]
    if filt:
        lines = filt(lines)
    with open(os.path.join(DST, dst), "w") as f:
        f.write("\n".join(lines) + "\n")


def summary(sub, frame_iters, keep):
    files = sorted(glob.glob(os.path.join(SRC, sub, "**", "*counter_collection.csv"), recursive=True))
    out = subprocess.run([sys.executable, os.path.join(DST, "pmc_summary.py"), "--frame-iters", str(frame_iters)] + files,
                         capture_output=True, text=True, check=True).stdout
    blocks, cur = [], []
    for line in out.splitlines():
        if not line.startswith("   ") and cur:
            blocks.append(cur)
            cur = []
        cur.append(line)
    blocks.append(cur)
    return "\n".join("\n".join(b) for b in blocks if keep(b[0])) + "\n"


def counter_sum(sub, counter, keep):
    """mean over dispatches of the per-dispatch counter, summed over the kernels `keep` selects (one call's kernels)"""
    per = {}
    for path in glob.glob(os.path.join(SRC, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter and "ccamd" in r["Kernel_Name"] and keep(r["Kernel_Name"]):
                per.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return per


cp("final_bench.json", "r03_final_bench.json")
cp("headline_under_rocprof.json", "r03_final_headline_bench_under_rocprof.json", lambda ls: [l for l in ls if l.startswith("{")])
cp("final_headline_kernel_stats.csv", "r03_final_headline_kernel_stats.csv", lambda ls: ls[:28])
for name in ("hard_bench", "codes_bench", "variants_bench", "mc_bench", "host_path", "rs_bench", "rs_erasures"):
    cp(name + ".txt", "r03_" + name + ".txt")
bench = json.load(open(os.path.join(SRC, "final_bench.json")))
fi = bench["config"]["frames_per_gpu"] * bench["config"]["mean_iterations_run"]
general = lambda n: "false, true>" in n.replace("(anonymous namespace)::", "").replace("ccamd::", "")
with open(os.path.join(DST, "r03_pmc_minsum_diag.txt"), "w") as f:
    f.write("# rocprofv3 --pmc passes of `bench.py --no-secondary --no-cpu-baseline --steps 3 --warmup 1` (profiles/tools/"
            "r03_refresh.sh); the general kernel only\n# (SINGLE = false; the 4096-frame sample launches of the two-pass "
            "scheme are other kernels); per frame-iteration = / %d\n" % round(fi))
    f.write(summary("pmc_hl", fi, lambda h: "minsum_diag_kernel" in h and "false, true>" in h))
with open(os.path.join(DST, "r03_pmc_minsum_single_o0.txt"), "w") as f:
    f.write("# the message-free kernel (stop rule O0), `bench.py --stop-rule 0`; per frame = / 2^20\n")
    f.write(summary("pmc_o0", 1 << 20, lambda h: "minsum_diag_kernel" in h))
with open(os.path.join(DST, "r03_pmc_rs_decode.txt"), "w") as f:
    f.write("# RS(255,223) Berlekamp-Massey decode, 2^20 frames with 0..16 symbol errors (profiles/tools/rs_bench.py 20): "
            "every kernel of the call; per frame = / 2^20\n")
    f.write(summary("pmc_rs", 1 << 20, lambda h: True))
# traffic: FETCH_SIZE x2 on gfx950 for wide coalesced reads (MI355X_MICROARCH.md, HBM section) + WRITE_SIZE, KB
fetch = counter_sum("pmc_hl", "FETCH_SIZE", general)
write = counter_sum("pmc_hl", "WRITE_SIZE", general)
fk = sum(sum(v) / len(v) for v in fetch.values())
wk = sum(sum(v) / len(v) for v in write.values())
json.dump({"kernel": bench["roofline"]["kernel"], "batch_log2": 20, "round": 3, "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk,
           "correction": "gfx950: FETCH_SIZE x2 for wide coalesced reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported",
           "traffic_bytes_per_launch": int(2 * fk * 1024 + wk * 1024),
           "source": "profiles/r03_pmc_minsum_diag.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, "
                     "profiles/tools/r03_refresh.sh)"}, open(os.path.join(DST, "traffic_minsum.json"), "w"), indent=1)
fetch = counter_sum("pmc_rs", "FETCH_SIZE", lambda n: True)
write = counter_sum("pmc_rs", "WRITE_SIZE", lambda n: True)
# rs_bench.py runs the BM decode 1 + 5 times and other workloads: per-kernel MEAN per dispatch, summed over the kernels of one
# decode call (each launched once per call); median over the six decode calls of `rs_bench.py 20 only`
med = lambda v: sorted(v)[len(v) // 2]  # (the one encode call that makes the codewords uses two of the kernels too)
per_kernel = {k: (med(v), med(write.get(k, [0]))) for k, v in fetch.items()}
dec = {k: v for k, v in per_kernel.items()
       if "parity_kernel" not in k and "bitslice_fused_syndrome_kernel<false, true>" not in k}  # (the encoder's two kernels)
total = sum(2 * a + b for a, b in dec.values()) * 1024
json.dump({"workload": "rs255_223_bm_2^20", "round": 3, "traffic_bytes_per_launch": int(total),
           "traffic_bytes_per_frame": round(total / (1 << 20), 1), "algorithmic_bytes_per_frame": 518,
           "correction": "gfx950: FETCH_SIZE x2 (MI355X_MICROARCH.md, HBM section) + WRITE_SIZE; decode kernels only "
                         "(bitslice_fused_syndrome<false, RAW> / bitslice_parity belong to the encoder rs_bench.py also runs)",
           "kernels_KB_fetch_write": {k.replace("ccamd::(anonymous namespace)::", "")[:80]: [round(a), round(b)] for k, (a, b) in per_kernel.items()},
           "note": "per dispatch means; see r03_pmc_rs_decode.txt"}, open(os.path.join(DST, "traffic_rs.json"), "w"), indent=1)
print("copied; traffic headline: %.4f GB per launch" % ((2 * fk + wk) * 1024 / 1e9))
