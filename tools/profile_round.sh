#!/bin/bash
# One-shot profile of the headline workload on the GPU box (run through gpurun):
#   bash tools/profile_round.sh <tag>
# writes gpurun_out/<tag>_bench.json, <tag>_kernel_trace.txt, <tag>_pmc.txt, <tag>_traffic.json
# (copy the ones to keep into profiles/).  Counters are collected in their own --pmc passes, never
# together with trace domains.
tag=${1:-rXX}
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
if [ -z "$PMC_ONLY" ]; then
timeout 300 python bench.py 2>/dev/null | tail -1 > gpurun_out/${tag}_bench.json
rm -rf /tmp/prof_kt && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python bench.py --cpu-sample 0 > /tmp/kt.log 2>&1
python tools/rocprof_summary.py $(find /tmp/prof_kt -name "*.db" | head -1) > gpurun_out/${tag}_kernel_trace.txt
fi
{
  echo "# rocprofv3 --pmc passes, 65536 NAND / launch (bench.py --steps 1 --warmup 0); KiB units (gfx950: double FETCH_SIZE)"
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    rm -rf /tmp/prof_pmc; timeout 300 rocprofv3 --pmc $grp -d /tmp/prof_pmc -o pmc -- python bench.py --steps 1 --warmup 0 --cpu-sample 0 > /tmp/pmc.log 2>&1
    db=$(find /tmp/prof_pmc -name "*.db" 2>/dev/null | head -1)
    if [ -n "$db" ]; then python tools/rocprof_summary.py $db --pmc | grep -v "^#\|^ *[0-9]\|^ calls"; else echo "# pass failed: $grp"; tail -5 /tmp/pmc.log | sed 's/^/#   /'; fi
  done
} > gpurun_out/${tag}_pmc.txt
python - "$tag" <<'PY'
import json, re, sys
tag = sys.argv[1]
vals = {}
for line in open(f"gpurun_out/{tag}_pmc.txt"):
    m = re.match(r"(\w+)\s+([\d.]+)\s+n=\d+\s+(.*)", line)
    if m and "blind_rotate" in m.group(3):
        vals[m.group(1)] = float(m.group(2))
if "FETCH_SIZE" in vals:
    out = {
        "_doc": "HBM traffic of the dominant kernel from separate rocprofv3 --pmc passes: bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 "
                "per launch; the factor 2 is the gfx950 FETCH_SIZE correction prescribed by MI355X_MICROARCH.md (HBM section). "
                "bench.py copies traffic_bytes_per_launch into roofline.traffic when the workload matches.",
        "kernel": "blind_rotate_fp_kernel<Decomp<3,6,1>>",
        "workload": {"gates_per_launch": 65536, "params": "128bit", "op": "NAND"},
        "FETCH_SIZE_KiB": vals["FETCH_SIZE"], "WRITE_SIZE_KiB": vals.get("WRITE_SIZE", 0.0),
        "traffic_bytes_per_launch": int((2 * vals["FETCH_SIZE"] + vals.get("WRITE_SIZE", 0.0)) * 1024),
    }
    if "TCC_HIT_sum" in vals and "TCC_MISS_sum" in vals:
        out["l2_hit_rate"] = round(vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"]), 4)
    json.dump(out, open(f"gpurun_out/{tag}_traffic.json", "w"), indent=1)
    print(out)
PY
cat gpurun_out/${tag}_kernel_trace.txt | head -12
