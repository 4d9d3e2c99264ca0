"""Per-kernel-family HBM traffic of one denoising step from the two rocprofv3 PMC passes of tools/profile_step.sh.

usage: python tools/traffic_table.py <dir with fetch/ and write/ subdirs> <tag>
Writes <dir>/<tag>_traffic_by_kernel.json (+ .md) and <dir>/<tag>_hbm_traffic_step1024.json.

Counters per MI355X_MICROARCH.md (HBM): read bytes = 2 * FETCH_SIZE * 1024 (gfx950 tallies 128-B requests of wide coalesced
reads at 64 B), written bytes = WRITE_SIZE * 1024. Each pass runs bench.py --steps 2 --warmup 1: 1 eager warm-up step (inside
capture()) + 1 graph warm-up + 2 timed = 4 executed steps; per-step figures divide by 4 (the model build / weight repack
kernels are listed separately under "setup" by name).
"""
import csv
import glob
import hashlib
import json
import os
import re
import sys
from collections import defaultdict

STEPS = 4
# not part of a denoising step: torch's own kernels (model build, weight repack, randn: at::native::*), runtime copies
SETUP = re.compile(r"^at::|^at_|^__amd_rocclr|elementwise_kernel|vectorized|distribution|fill|copy_|copyBuffer|Memcpy|index|"
                   r"cat|arange|philox|normal|random", re.I)


def family(name):
    """Kernel family = qualified name (namespaces kept: `at::native::...` must not collapse into `at`) + template arguments."""
    n = re.sub(r"^void\s+", "", name)
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.match(r"((?:[A-Za-z0-9_]+::)*[A-Za-z0-9_]+)(<[^(]*>)?", n)
    base = m.group(1) if m else n
    targs = (m.group(2) or "") if m else ""
    targs = targs.replace(" ", "")
    if base.startswith("at::"):
        return base                      # one family per torch kernel name; their template arguments are noise here
    return base + targs


def load(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    per = defaultdict(lambda: [0.0, 0])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = family(r["Kernel_Name"])
            per[k][0] += float(r["Counter_Value"])
            per[k][1] += 1
    return per


def main():
    O, tag = sys.argv[1], sys.argv[2]
    fe, wr = load(os.path.join(O, "fetch"), "FETCH_SIZE"), load(os.path.join(O, "write"), "WRITE_SIZE")
    rows = []
    for k in sorted(set(fe) | set(wr)):
        rd = 2.0 * fe.get(k, [0, 0])[0] * 1024
        wb = wr.get(k, [0, 0])[0] * 1024
        n = max(fe.get(k, [0, 0])[1], wr.get(k, [0, 0])[1])
        rows.append(dict(kernel=k, dispatches=n, setup=bool(SETUP.search(k)) and not k.startswith(("gemm", "flash", "gn_", "layernorm", "dc_", "ff_", "norm_", "ln_")),
                         read_gb_per_step=rd / STEPS / 1e9, write_gb_per_step=wb / STEPS / 1e9))
    rows.sort(key=lambda r: -(r["read_gb_per_step"] + r["write_gb_per_step"]))
    step_rows = [r for r in rows if not r["setup"]]
    tot_r = sum(r["read_gb_per_step"] for r in step_rows)
    tot_w = sum(r["write_gb_per_step"] for r in step_rows)
    # algorithmic bytes from the bench line of the same tree, if present next to the passes
    algo = {}
    for cand in (os.path.join(O, "bench_full.json"), os.path.join(O, "..", "bench_full.json")):
        if os.path.exists(cand):
            try:
                line = [l for l in open(cand) if l.startswith("{")][-1]
                for r in json.loads(line)["roofline"].get("per_kernel_eager_step", []):
                    algo[r["kernel"]] = r.get("algorithmic_gb")
            except Exception:
                pass
            break
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    h = bench.kernel_source_hash()
    out = dict(workload="inference_1024_v1.0.yaml denoising step (bench.py default)", steps_in_pass=STEPS,
               correction="read = 2*FETCH_SIZE*1024 B (gfx950), write = WRITE_SIZE*1024 B", kernel_source_hash=h,
               read_gb_per_step=tot_r, write_gb_per_step=tot_w, rows=rows, algorithmic_gb_by_tracer_family=algo)
    json.dump(out, open(os.path.join(O, f"{tag}_traffic_by_kernel.json"), "w"), indent=1)
    json.dump(dict(workload=out["workload"], collected="tools/profile_step.sh (two PMC passes), tools/traffic_table.py",
                   correction=out["correction"], hbm_bytes_per_step=int((tot_r + tot_w) * 1e9),
                   read_bytes_per_step=int(tot_r * 1e9), write_bytes_per_step=int(tot_w * 1e9), round=tag,
                   kernel_source_hash=h), open(os.path.join(O, f"{tag}_hbm_traffic_step1024.json"), "w"), indent=1)
    with open(os.path.join(O, f"{tag}_traffic_by_kernel.md"), "w") as f:
        f.write(f"HBM traffic per denoising step by kernel family ({tag}, kernel_source_hash {h})\n\n")
        f.write(f"total read {tot_r:.1f} GB + written {tot_w:.1f} GB = {tot_r + tot_w:.1f} GB per step\n\n")
        f.write("| kernel | dispatches (4 steps) | read GB/step | written GB/step |\n|---|---|---|---|\n")
        for r in rows:
            if r["read_gb_per_step"] + r["write_gb_per_step"] < 0.05:
                continue
            f.write(f"| {r['kernel']}{' (setup)' if r['setup'] else ''} | {r['dispatches']} | {r['read_gb_per_step']:.2f} | {r['write_gb_per_step']:.2f} |\n")
    print(open(os.path.join(O, f"{tag}_traffic_by_kernel.md")).read())


if __name__ == "__main__":
    main()
