#!/usr/bin/env python3
"""profiles/rNN_pmc_traffic.json (what bench.py quotes as roofline.traffic) from the summaries of tools/prof_round.sh:
    python tools/pmc_traffic_json.py gpurun_out/prof_r06 profiles/r06_pmc_traffic.json
FETCH_SIZE is doubled (gfx950 counts 128-byte requests at 64 bytes: MI355X_MICROARCH.md, HBM), WRITE_SIZE taken as read; the
kernel sources' fingerprint goes in, so that bench.py can tell a profile older than the kernel."""
import hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fingerprint(files):
    return hashlib.sha256(b"".join(open(os.path.join(ROOT, "gpbayestools_hic_amd", "csrc", f), "rb").read() for f in files)).hexdigest()[:16]


def block(summary_path, kernel, algorithmic, files, note):
    rows = [json.loads(ln) for ln in open(summary_path) if ln.strip()]
    get = lambda c: next((r[c] for r in rows if c in r), None)
    fetch, write = get("FETCH_SIZE"), get("WRITE_SIZE")
    hit, miss = get("TCC_HIT_sum"), get("TCC_MISS_sum")
    out = {"kernel": kernel, "FETCH_SIZE_KB_raw": fetch, "WRITE_SIZE_KB": write,
           "bytes_per_launch_corrected": (2.0 * fetch + write) * 1024.0 if fetch is not None and write is not None else None,
           "algorithmic_bytes_per_launch": algorithmic, "TCC_HIT_sum": hit, "TCC_MISS_sum": miss,
           "l2_hit_rate": hit / (hit + miss) if hit and miss else None, "source_sha16": fingerprint(files), "note": note,
           "counters": {k: v for r in rows for k, v in r.items() if not k.startswith("sum_") and k not in ("kernel", "dispatches")}}
    if out["bytes_per_launch_corrected"]:
        out["ratio"] = out["bytes_per_launch_corrected"] / algorithmic
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    N, P, W = 2048, 10, 2048
    doc = {"source": "tools/prof_round.sh -> tools/pmc_passes.sh (separate rocprofv3 --pmc passes with --kernel-trace only; median over the "
                     "dispatches of the timed region, first 7 skipped) on `python3 bench.py [--sliced] --steps 10 --warmup 3 --preheat 0 "
                     "--no-cpu-baseline --no-extras --no-uniform`: burnt-in ensemble, every one of the 2048 proposal rows inside the prior box",
           "workload": {"config": 4, "N": N, "P": P, "W_per_launch": W, "burnt_in": True}}
    p = os.path.join(src, "pmc", "summary.jsonl")
    if os.path.exists(p):
        doc["k_predict"] = block(p, "k_predict<128, 4, 128, 16, true>", P * (N * N / 2 + N * W) * 8.0, ["gpb_predict.hip", "gemm_tile.h"],
                                 "algorithmic bytes: the lower half of L^-1 (168 MB) + K*^T of the 2048 rows (336 MB), fp64")
    p = os.path.join(src, "pmc8", "summary.jsonl")
    if os.path.exists(p):
        doc["k_predict_sliced"] = block(p, "k_predict_sliced", P * (N * N / 2 + N * W) * 6.0, ["gpb_sliced.hip"],
                                        "algorithmic bytes: six int8 digit planes of the lower half of L^-1 (126 MB) and of K*^T (252 MB)")
    json.dump(doc, open(dst, "w"), indent=1)
    print(json.dumps({k: {kk: v.get(kk) for kk in ("bytes_per_launch_corrected", "ratio", "l2_hit_rate")} for k, v in doc.items() if k.startswith("k_")}))


if __name__ == "__main__":
    main()
