"""Per-kernel register / scratch / occupancy table of one csrc/*.hip file, from hipcc's own resource remarks
(-Rpass-analysis=kernel-resource-usage; cross-compiles without a GPU).

    python tools/kernel_resources.py tall.hip [name-filter] [--diff before.json] [--save after.json]

Used to check that an edit of a hot kernel did not move its register budget (VGPRs + AGPRs decide the wavefronts per SIMD),
did not spill, and did not grow scratch."""
import json
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pytorch_geometric_signed_directed_amd", "csrc")
KEYS = ("VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "VGPRs Spill")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return [re.sub(r"pygsd::\(anonymous namespace\)::", "", line).split("(")[0].replace("void ", "") for line in out.splitlines()]


def resources(src):
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
           "-Rpass-analysis=kernel-resource-usage"]
    err = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
    table, cur = {}, None
    for line in err.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            table[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+)", line)
        if m and cur and m.group(1).strip() in KEYS:
            table[cur][m.group(1).strip()] = m.group(2)
    names = list(table)
    return dict(zip(demangle(names), (table[n] for n in names)))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    src, pat = args[0], (args[1] if len(args) > 1 else "")
    table = {k: v for k, v in resources(src).items() if pat in k}
    before = None
    if "--diff" in sys.argv:
        with open(sys.argv[sys.argv.index("--diff") + 1]) as fh:
            before = json.load(fh)
    if "--save" in sys.argv:
        with open(sys.argv[sys.argv.index("--save") + 1], "w") as fh:
            json.dump(table, fh, indent=1)
    print(f"{'kernel':70s} vgpr agpr sgpr scratch occ spill")
    for name, r in table.items():
        row = [r.get(k, "?") for k in KEYS]
        mark = ""
        if before is not None and before.get(name) != r:
            b = before.get(name)
            mark = "   <- was " + (" ".join(b.get(k, "?") for k in KEYS) if b else "absent")
        print(f"{name[:70]:70s} {row[0]:>4} {row[1]:>4} {row[2]:>4} {row[3]:>7} {row[4]:>3} {row[5]:>5}{mark}")


if __name__ == "__main__":
    main()
