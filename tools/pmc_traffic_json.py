#!/usr/bin/env python3
"""Combine the two rocprofv3 --pmc passes over bench.py (tools/pmc_bench.sh: FETCH_SIZE, WRITE_SIZE per kernel) with a bench line's
`kernels` block into the table bench.py reads for `roofline.traffic`:

    python tools/pmc_traffic_json.py <pmc_bench_FETCH_SIZE.csv> <pmc_bench_WRITE_SIZE.csv> <bench_line.json> <out.json>

HBM bytes per entry-point launch = sum over the entry point's device kernels of launches x (2 x FETCH_SIZE + WRITE_SIZE) x 1024
(gfx950: FETCH_SIZE counts 2x units, KB = 1024 B; MI355X_MICROARCH.md) / launches of the entry point's primary kernel.  The PMC
passes run the same program as the bench line, so the launch mix (shapes) is the same."""
import csv
import json
import sys

ENTRY = {  # entry point -> (primary kernel substrings, secondary kernel substrings)
    "coin_conv_gemm_bf16": (["conv_gemm_p8_kernel", "conv_gemm_s4_kernel", "conv_gemm256_bf16_kernel", "conv_gemm_bf16_kernel"],
                            ["conv_gemm_p8_slab_sum_kernel", "conv_gemm_p8_tail_kernel", "conv_gemm_s4_tail_kernel"]),
    "coin_conv_wgrad_bf16": (["conv_wgrad_p8_kernel", "conv_wgrad_s4_kernel", "conv_wgrad_bf16_kernel"], ["tn_reduce_kernel", "tn_reduce_wide_kernel", "conv_wgrad_s4_reduce_kernel", "wgrad_reduce_kernel"]),
    "coin_bn_bwd": (["bn_bwd_reduce_kernel"], ["bn_bwd_finalize_kernel", "bn_bwd_dx_kernel"]),
    "coin_bn_apply_fwd": (["bn_apply_kernel", "bn_apply_mean_kernel"], []),
    "coin_bn_stats": (["bn_stats_kernel"], ["bn_finalize_kernel"]),
    "coin_roi_align_fwd": (["roi_align_fwd"], []),
    "coin_roi_align_bwd": (["roi_align_bwd"], []),
    "coin_gemm_nt": (["gemm_nt_bf16_kernel"], []),
}


def table(path):
    out = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            out[r["Kernel_Name"]] = (int(r["launches"]), float(r["mean_value"]))
    return out


def main():
    fetch, write, line, dst = sys.argv[1:5]
    ft, wt = table(fetch), table(write)
    with open(line) as f:
        bench = json.loads(f.read().strip().splitlines()[-1])
    res = {"_source": {"fetch": fetch, "write": write, "bench_value": bench["value"], "units": "FETCH_SIZE x2 x 1024 B + WRITE_SIZE x 1024 B"}}
    for entry, (prim, sec) in ENTRY.items():
        k = bench["kernels"].get(entry)
        if k is None:
            continue
        n_entry = sum(n for name, (n, _) in ft.items() if any(p in name for p in prim))
        if not n_entry:
            continue
        fb = sum(n * v for name, (n, v) in ft.items() if any(p in name for p in prim + sec))
        wb = sum(n * v for name, (n, v) in wt.items() if any(p in name for p in prim + sec))
        hbm = (2.0 * fb + wb) * 1024.0 / n_entry
        alg = k.get("alg_bytes", k.get("alg_flop"))
        rec = {"alg_bytes": alg, "alg_unit": "bytes" if "alg_bytes" in k else "flop", "pmc_launches": n_entry,
               "FETCH_SIZE_KB_per_launch": fb / n_entry, "WRITE_SIZE_KB_per_launch": wb / n_entry, "hbm_bytes": hbm,
               "device_kernels": prim + sec}
        if "alg_bytes" in k:
            rec["hbm_over_alg"] = hbm / alg
        else:
            rec["flop_per_hbm_byte"] = alg / hbm
        res[entry] = rec
    with open(dst, "w") as f:
        json.dump(res, f, indent=1)
    for e, r in res.items():
        if e[0] != "_":
            print(e, {x: (round(y, 3) if isinstance(y, float) else y) for x, y in r.items() if x != "device_kernels"})


if __name__ == "__main__":
    main()
