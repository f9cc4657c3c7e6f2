"""profiles/rNN_fast_traffic.json: counters of the bf16x3 mode's three matrix kernels at the bench size, from the per-pass
PMC summaries (tools/pmc_summary.py, `top_mean` = mean over the full-size dispatches) and the kernel-trace stats of
`tools/fast_timing.py 8192 N noref`.   usage: pmc_fast.py <dir with pmc_fast{1..5}.csv + stats_fast_kernel_stats.csv> <envs> <out.json>

Per kernel: time per launch, bf16 MFMA instructions and FLOPs (x 32768 per v_mfma_f32_32x32x16_bf16), achieved TFLOP/s,
the clock under the kernel (GRBM_GUI_ACTIVE summed over the 8 XCDs / 8 / time of the SAME profiled launch -- the counter
passes run the kernel slower than the trace run, so the clock is computed against the pass's own duration when the
pass reports it, else against the trace time), the matrix-pipe busy fraction SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x
cycles), and HBM bytes (FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE, KiB).  Stamped with the SHA-256 of
the kernel sources: bench.py drops the record when they change."""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import FAST_SOURCES, kernel_source_hash  # noqa: E402

KERNELS = {"sa2_bf16x3_persistent_kernel": "sa2_bf16x3_persistent_kernel",
           "sa_mlp_bf16_resident_kernel": "sa_mlp_bf16_resident_kernel<1, 64, 64, 64, 16, true>",
           "linear_bf16x3_pairs_kernel<3>": "linear_bf16x3_pairs_kernel<3>",   # 512 -> 1024 + pooling, pairs in / pairs out
           "sa3_front_bf16x3_kernel": "sa3_front_bf16x3_kernel<false>"}       # 272 -> 512 -> 512 fused, fp32 rows in / pairs out
d, envs, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]


def counter(pass_id, kernel, name, want_ms=False):
    for r in csv.DictReader(open(f"{d}/pmc_fast{pass_id}.csv")):
        if kernel in r["kernel"] and r["counter"] == name:
            if want_ms:  # duration of the same dispatches under the counter pass (None if rocprofv3 did not report it)
                return float(r["top_ms"]) if r.get("top_ms") else None
            return float(r["top_mean"])
    raise SystemExit(f"{name} of {kernel} not in pass {pass_id}")


def launch_ms(kernel):
    for r in csv.DictReader(open(f"{d}/stats_fast_kernel_stats.csv")):
        if kernel in r["Name"]:
            return float(r["MaxNs"]) / 1e6, float(r["AverageNs"]) / 1e6, int(r["Calls"])
    raise SystemExit(f"{kernel} not in the kernel stats")


rec = {"envs_per_gpu": envs, "kernel_source_sha256": kernel_source_hash(FAST_SOURCES), "sources": list(FAST_SOURCES),
       "source": "gpurun_out passes pmc_fast{1..5} (rocprofv3 --pmc, one counter group per pass) + stats_fast (kernel trace) of "
                 "tools/fast_timing.py %d 2 noref" % envs,
       "correction": "gfx950: FETCH_SIZE counts half of wide (16 B/lane) coalesced reads (MI355X_MICROARCH.md, HBM): reads doubled",
       "kernels": {}}
for key, name in KERNELS.items():
    mx, avg, calls = launch_ms(name)
    ms = avg  # (noref runs hold full-size launches only)
    mfma = counter(4, name, "SQ_INSTS_MFMA")
    busy = counter(3, name, "SQ_VALU_MFMA_BUSY_CYCLES")
    gui = counter(3, name, "GRBM_GUI_ACTIVE") / 8.0  # cycles of the launch (the counter is summed over the XCDs)
    fetch, write = counter(1, name, "FETCH_SIZE"), counter(2, name, "WRITE_SIZE")
    valu, salu, lds = counter(5, name, "SQ_INSTS_VALU"), counter(5, name, "SQ_INSTS_SALU"), counter(5, name, "SQ_INSTS_LDS")
    flops = mfma * 32768.0
    rec["kernels"][key] = {
        "ms_per_launch": ms, "ms_max": mx, "launches_in_trace": calls, "mfma_insts_per_launch": mfma,
        "bf16_mfma_flops_per_launch": flops, "achieved_tflops": flops / (ms * 1e-3) / 1e12,
        "frac_of_2500": flops / (ms * 1e-3) / 1e12 / 2500.0, "gui_cycles_per_launch": gui,
        "ms_under_counter_pass": counter(3, name, "GRBM_GUI_ACTIVE", True),
        "clock_ghz": gui / ((counter(3, name, "GRBM_GUI_ACTIVE", True) or ms) * 1e-3) / 1e9,
        "mfma_busy_frac": busy / (1024.0 * gui),
        "valu_insts_per_mfma": (valu - mfma) / mfma, "salu_insts_per_mfma": salu / mfma, "lds_insts_per_mfma": lds / mfma,
        "fetch_size_kib_per_launch": fetch, "write_size_kib_per_launch": write,
        "hbm_bytes_per_launch": 1024.0 * (2 * fetch + write)}
json.dump(rec, open(out, "w"), indent=1)
print(open(out).read())
