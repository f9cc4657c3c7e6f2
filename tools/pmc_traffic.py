"""Derive profiles/rNN_traffic.json (HBM bytes + MFMA-busy of the dominant kernel) from the per-pass PMC
summaries written by tools/evidence.sh   usage: pmc_traffic.py <dir with pass{1,2,3}_summary.csv> <envs> <out.json>"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_hash  # noqa: E402  (stamps the record: bench.py drops it when the kernel source changes)

KERNEL = "sa_mlp_packed_kernel<64, 128, 128, 256, 8, true"  # (prefix: the row-map size is a further template argument since round 6)
d, envs, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]


def per_launch(path, counter):
    for r in csv.DictReader(open(path)):
        if KERNEL in r["kernel"] and r["counter"] == counter:
            return float(r.get("top_mean") or r["per_launch"])  # (mean over the full-size dispatches)
    raise SystemExit(f"{counter} of {KERNEL} not in {path}")


fetch = per_launch(f"{d}/pass1_summary.csv", "FETCH_SIZE")
write = per_launch(f"{d}/pass2_summary.csv", "WRITE_SIZE")
busy = per_launch(f"{d}/pass3_summary.csv", "SQ_VALU_MFMA_BUSY_CYCLES")
active = per_launch(f"{d}/pass3_summary.csv", "GRBM_GUI_ACTIVE")  # summed over the 8 XCDs
# algorithmic: every pre-activation row (512 points x 128 ch) and query row (128 x 128 ch) read once, pooled rows
# (128 x 256 ch) written once, plus the neighbour indices actually walked (~40 of 128 per query: counted as 128/3)
alg = envs * (512 * 128 * 4 + 128 * 128 * 4 + 128 * 256 * 4 + 128 * 128 * 4 // 3)


def whole_step(steps_in_pass=3):
    """Corrected HBM bytes of EVERY kernel of the headline step (the passes run 1 warm-up + 2 timed steps = 3 steps):
    sum over all dispatches / steps; the dominant kernels listed.  Algorithmic bytes of the step: SURVEY 8(d)."""
    f = {r["kernel"]: float(r["sum"]) for r in csv.DictReader(open(f"{d}/pass1_summary.csv")) if r["counter"] == "FETCH_SIZE"}
    w = {r["kernel"]: float(r["sum"]) for r in csv.DictReader(open(f"{d}/pass2_summary.csv")) if r["counter"] == "WRITE_SIZE"}
    per = {k: 1024.0 * (2 * f.get(k, 0.0) + w.get(k, 0.0)) / steps_in_pass for k in set(f) | set(w)}
    top = sorted(per.items(), key=lambda kv: -kv[1])[:8]
    return {"hbm_bytes_per_step": sum(per.values()), "algorithmic_bytes_per_step": envs * (1.3e5 + 49152),
            "top_kernels_gb_per_step": {k.replace("void ", "")[:60]: round(v / 1e9, 2) for k, v in top}}


json.dump({
    "kernel": "void " + KERNEL, "envs_per_gpu": envs, "whole_step": whole_step(),
    "source": "profiles/%s_pmc_pass{1,2,3}_envs%d.csv (rocprofv3 --pmc, separate passes)" % (os.path.basename(out).split("_")[0], envs),
    "kernel_source_sha256": kernel_source_hash(),
    "fetch_size_kib_per_launch": fetch, "write_size_kib_per_launch": write,
    "correction": "gfx950: FETCH_SIZE counts half of wide (16 B/lane) coalesced reads (MI355X_MICROARCH.md, HBM): reads doubled",
    "hbm_bytes_per_launch": 1024.0 * (2 * fetch + write), "algorithmic_bytes_per_launch": alg,
    "mfma_busy_frac": busy / (1024 * active / 8),
}, open(out, "w"), indent=1)
print(open(out).read())
