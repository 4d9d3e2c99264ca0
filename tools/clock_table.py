"""Effective shader clock per kernel family of one denoising step, from ONE rocprofv3 pass:
    rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d <dir> -o out --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-ae --no-trace --no-cpu-baseline
usage: python tools/clock_table.py <dir> <tag>     -> <dir>/<tag>_clock_by_kernel.md
Effective clock of a dispatch = GRBM_GUI_ACTIVE / 8 / (End - Start): rocprofv3 reports the counter summed over the 8 XCDs
(MI355X_MICROARCH.md 'DVFS give-back'); it reads high on dispatches shorter than ~0.3 ms, so only families whose launches
average >= 150 us are listed and the figure is the duration-weighted mean. Profiled passes clock a little lower than free runs."""
import csv, glob, os, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from traffic_table import family

def main():
    O, tag = sys.argv[1], sys.argv[2]
    per = defaultdict(lambda: [0.0, 0.0, 0])          # cycles/8, ns, launches
    for f in glob.glob(os.path.join(O, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
                continue
            dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            if dur <= 0:
                continue
            k = family(r["Kernel_Name"])
            per[k][0] += float(r["Counter_Value"]) / 8.0
            per[k][1] += dur
            per[k][2] += 1
    rows = [(k, v[0] / v[1], v[1] / v[2] / 1e3, v[2], v[1] / 1e6) for k, v in per.items() if v[2] and v[1] / v[2] >= 150e3 and not k.startswith("at::")]
    rows.sort(key=lambda r: -r[4])
    with open(os.path.join(O, f"{tag}_clock_by_kernel.md"), "w") as f:
        f.write(f"Effective shader clock by kernel family ({tag}; GRBM_GUI_ACTIVE / 8 / duration, launches averaging >= 150 us)\n\n")
        f.write("| kernel | launches | avg us | total ms | effective GHz |\n|---|---|---|---|---|\n")
        for k, ghz, us, n, ms in rows:
            f.write(f"| {k} | {n} | {us:.0f} | {ms:.1f} | {ghz:.2f} |\n")
    print(open(os.path.join(O, f"{tag}_clock_by_kernel.md")).read())

if __name__ == "__main__":
    main()
