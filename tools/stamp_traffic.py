"""Write profiles/grad_kernel_traffic.json from a per-kernel PMC table (tools/summarize_pmc.py output of the two
`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes over bench.py) and stamp it with the commit and the digest of the
kernel sources it was measured at -- bench.py reports `roofline.traffic` only while that digest matches the tree.

    python tools/stamp_traffic.py profiles/r06z_pmc_traffic_per_kernel_mean.csv "end of round 6" [profiles/r06z_ppo_pmc_per_kernel_mean.csv]

The optional second table (the SQ-counter pass of the same command) renews `mfma_busy`.  Both tables must come from runs in
which the kernel is ONE 256-workgroup launch per minibatch (TRL_PPO_CHAINS=joint), which is what bench.py's roofline entry times.
"""
import csv
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def grad_row(table):
    return next(r for r in csv.DictReader(open(table)) if r["kernel"].startswith("void ppo_grad_wave_kernel<17, 64, 6")
                or r["kernel"].startswith("ppo_grad_wave_kernel<17, 64, 6"))


def main(table, when, sq_table=None):
    row = grad_row(table)
    fetch_kb, write_kb = float(row["FETCH_SIZE"]), float(row["WRITE_SIZE"])
    try:
        commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=REPO, stdout=subprocess.PIPE, text=True).stdout.strip()
    except OSError:
        commit = None
    with open(bench.TRAFFIC_FILE) as f:
        rec = json.load(f)
    rec.update({
        "source": "%s (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, mean of %s launches at B=65536, %s)"
                  % (os.path.relpath(table, REPO), row["dispatches"], when),
        "fetch_size_kb_raw": fetch_kb, "write_size_kb": write_kb,
        "traffic_bytes_per_launch": int(round((2.0 * fetch_kb + write_kb) * 1024)),
        "measured_at_commit": commit, "kernel_source_sha256": bench.kernel_source_digest(),
        "kernel_sources": list(bench.TRAFFIC_SOURCES)})
    if sq_table:
        q = grad_row(sq_table)
        c = {k: float(v) for k, v in q.items() if k.startswith("SQ_")}
        waves = 1024.0
        rec["mfma_busy"] = {
            "frac": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["SQ_BUSY_CYCLES"] * 32.0),
            "definition": "SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES x 32): share of the launch's SIMD-cycles with an MFMA in the matrix pipe",
            "source": "%s (rocprofv3 --pmc, a run of its own: mean of %s launches, %s)" % (os.path.relpath(sq_table, REPO), q["dispatches"], when),
            "counters": c,
            "per_wave": {"mfma": c["SQ_VALU_MFMA_BUSY_CYCLES"] / 32.0 / waves,
                         "valu_non_mfma": c["SQ_INSTS_VALU"] / waves - c["SQ_VALU_MFMA_BUSY_CYCLES"] / 32.0 / waves,
                         "lds": c["SQ_INSTS_LDS"] / waves, "wave_cycles_x4": c["SQ_WAVE_CYCLES"] / waves}}
    with open(bench.TRAFFIC_FILE, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "", sys.argv[3] if len(sys.argv) > 3 else None)
