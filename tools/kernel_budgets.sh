#!/bin/bash
# Register / scratch / LDS budget and occupancy of EVERY kernel of librspt.so, from the compiler (no GPU needed):
# writes profiles/<round>_static_kernel_budgets.md.  usage: tools/kernel_budgets.sh [round prefix, default r03]
set -u
round=${1:-r03}; repo=$(cd "$(dirname "$0")/.." && pwd); work=$(mktemp -d)
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -Wno-unused-function -Wno-unused-result"
cd $repo/rs_pbrt_amd/csrc
ls *.hip | xargs -P 8 -I{} sh -c "/opt/rocm/bin/hipcc $F -Rpass-analysis=kernel-resource-usage -c -o /dev/null {} 2> $work/{}.ru"
python3 - $work $repo/profiles/${round}_static_kernel_budgets.md $(python3 -c "import sys; sys.path.insert(0, '$repo'); from rs_pbrt_amd import lib; print(lib.source_hash())") <<'PY'
import glob, os, re, subprocess, sys
work, out, h = sys.argv[1:4]
rows = []
for f in sorted(glob.glob(work + "/*.ru")):
    t = open(f).read()
    for m in re.finditer(r"Function Name: (\S+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?LDS Size \[bytes/block\]: (\d+)", t, re.S):
        n = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "").replace("rspt::", "")
        rows.append((os.path.basename(f)[:-7], n, int(m.group(2)), int(m.group(3)), int(m.group(5)), int(m.group(4))))
with open(out, "w") as o:
    o.write("# Static budgets of every kernel of librspt.so (hipcc -Rpass-analysis=kernel-resource-usage, gfx950; source hash %s)\n\n" % h)
    o.write("Generated on the build host by `tools/kernel_budgets.sh` (no GPU needed). Feature masks of `k_shade<F>`: 4098 = diffuse, 2101314 = plastic, 2232390 = textured, "
            "4282384383 = generic, 4286578687 = dynamic (lobe lists built per hit), 4294967295 = + moving instances"
            "4 path with dynamic materials. `k_trace_w4<ANY, OUT_MODE, INST, ALPHA>`.\n\n| unit | kernel | VGPRs | scratch B/lane | LDS B/workgroup | waves/SIMD |\n|---|---|---|---|---|---|\n")
    for r in rows:
        o.write("| %s | `%s` | %d | %d | %d | %d |\n" % r)
print(out, len(rows), "kernels")
PY
rm -rf $work
