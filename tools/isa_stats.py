#!/usr/bin/env python3
"""Static VALU census of a kernel's gfx950 ISA, written to muzero_amd/lib/kernel_isa.json by __graft_entry__.build().

bench.py reports the env step kernel's VALU side next to its GB/s (at 46 B/step the HBM floor is below the
integer-ALU cost).  The counts are taken from the assembly hipcc emits for the sources as they are NOW, with the
Makefile's flags, so they cannot go stale.  k_env_step has forward branches only (no loop), so the static count is what
a wave executes when its lanes take both sides of every branch (the normal case: a wave mixes placing and passing games).
Issue cost per instruction from tools/valu_rate.hip on MI355X: 2.4 cycles per wave64 VALU instruction sustained, 4.2 for
the 64-bit shifts (half rate)."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "muzero_amd", "csrc")
OUT = os.path.join(ROOT, "muzero_amd", "lib", "kernel_isa.json")
FULL_RATE_CYCLES, HALF_RATE_CYCLES = 2.4, 4.2
HALF_RATE = re.compile(r"^v_(lshlrev_b64|lshrrev_b64|ashrrev_i64)$")
KERNELS = {"env_kernels.hip": ["k_env_step"], "mcts_kernels.hip": ["k_backup_select"]}   # k_backup_select has loops: its counts are STATIC (bench.py says so)


def makefile_flags():
    txt = open(os.path.join(CSRC, "Makefile")).read().replace("\\\n", " ")
    m = re.search(r"^FLAGS\s*=\s*(.*)$", txt, re.M)
    return [f for f in m.group(1).replace("$(ARCH)", "gfx950").split() if not f.startswith("-W") and f != "-fPIC"]


def census(asm, kernel):
    """VALU instructions between the kernel's entry label and its first s_endpgm"""
    lines = asm.splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%d%s\w*:" % (len(kernel), kernel), l))
    body = []
    for l in lines[start + 1:]:
        op = l.split(";")[0].strip().split()
        if not op:
            continue
        body.append(op[0])
        if op[0] == "s_endpgm":
            break
    valu = [o for o in body if o.startswith("v_")]
    half = [o for o in valu if HALF_RATE.match(o)]
    end = next(i for i, l in enumerate(lines[start:]) if l.split(";")[0].strip() == "s_endpgm")
    region = lines[start:start + end + 1]
    labels = {m.group(1): i for i, l in enumerate(region) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    backward = any(labels.get(m.group(1), 1 << 30) < i for i, l in enumerate(region)
                   for m in [re.match(r"^\s+s_cbranch\w*\s+(\.LBB\d+_\d+)", l)] if m)   # a loop would make the static count meaningless
    return {"valu_instructions": len(valu), "half_rate_64bit_shifts": len(half), "fp64_valu_instructions": sum(o.endswith("_f64") or "_f64_" in o for o in valu),
            "issue_cycles_per_wave": round((len(valu) - len(half)) * FULL_RATE_CYCLES + len(half) * HALF_RATE_CYCLES, 1),
            "salu_instructions": sum(o.startswith("s_") for o in body), "vmem_instructions": sum(o.startswith(("global_", "buffer_", "flat_")) for o in body),
            "has_loop": backward, "cycles_per_full_rate_valu": FULL_RATE_CYCLES, "cycles_per_half_rate_valu": HALF_RATE_CYCLES}


def main():
    out = {}
    for src, kernels in KERNELS.items():
        with tempfile.TemporaryDirectory() as td:
            s = os.path.join(td, "k.s")
            subprocess.check_call(["/opt/rocm/bin/hipcc"] + makefile_flags() + ["-S", "--cuda-device-only", "-o", s, os.path.join(CSRC, src)],
                                  stderr=subprocess.DEVNULL)
            asm = open(s).read()
        for k in kernels:
            out[k] = census(asm, k)
            out[k]["source"] = "muzero_amd/csrc/" + src
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    return out


if __name__ == "__main__":
    print(json.dumps(main(), indent=1))
    sys.exit(0)
