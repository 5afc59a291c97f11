"""HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE), per kernel:
   python tools/pmc_hbm.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> SPEC [SPEC ...]
SPEC = kernel-name substring, optionally `substring@GRID` (only launches of that Grid_Size) and `=label` (key in the JSON).
FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 reports half the bytes of wide coalesced reads); both counters are in KB."""
import csv, json, sys

csv.field_size_limit(1 << 30)


def mean(path, counter, kern, grid):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
         if r["Counter_Name"] == counter and kern in r["Kernel_Name"] and (grid is None or int(r["Grid_Size"]) == grid)]
    return (sum(v) / len(v), len(v)) if v else (0.0, 0)


out = {}
for spec in sys.argv[4:]:
    spec, _, label = spec.partition("=")
    kern, _, grid = spec.partition("@")
    grid = int(grid) if grid else None
    f, nf = mean(sys.argv[1], "FETCH_SIZE", kern, grid)
    w, nw = mean(sys.argv[2], "WRITE_SIZE", kern, grid)
    out[label or kern] = {"FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB": w, "launches_fetch_pass": nf, "launches_write_pass": nw,
                          "hbm_bytes_per_launch": (2 * f + w) * 1024, "grid": grid}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
