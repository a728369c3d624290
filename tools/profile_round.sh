#!/bin/bash
# One-shot profile of the headline workload on the GPU box (run through gpurun):
#   bash tools/profile_round.sh <tag>
# writes gpurun_out/<tag>_bench.json, <tag>_kernel_trace.txt, <tag>_pmc.txt, <tag>_counters.json
# (copy the ones to keep into profiles/; bench.py reads profiles/r06_counters.json and uses it only while its build_id —
# the hash of the kernel sources, tools/src_hash.py — equals the loaded library's).  Counters are collected in their own
# --pmc passes, never together with trace domains.  PARAMS=80bit profiles the 80-bit set; DECOMP=direct its opt-in
# direct decomposition (bench.py --decomp).
tag=${1:-r06}
PARAMS=${PARAMS:-128bit}
DEC=${DECOMP:+--decomp $DECOMP}
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
if [ -z "$PMC_ONLY" ]; then
timeout 600 python bench.py --params $PARAMS $DEC 2>/dev/null | tail -1 > gpurun_out/${tag}_bench.json
rm -rf /tmp/prof_kt && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python bench.py --params $PARAMS $DEC --cpu-sample 0 > /tmp/kt.log 2>&1
python tools/rocprof_summary.py $(find /tmp/prof_kt -name "*.db" | head -1) > gpurun_out/${tag}_kernel_trace.txt
fi
{
  echo "# rocprofv3 --pmc passes, 65536 NAND / launch (bench.py --steps 1 --warmup 0); FETCH/WRITE in KiB (gfx950: double FETCH_SIZE)"
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
             "GRBM_GUI_ACTIVE" "SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES" \
             "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    rm -rf /tmp/prof_pmc; timeout 300 rocprofv3 --pmc $grp -d /tmp/prof_pmc -o pmc -- python bench.py --params $PARAMS $DEC --steps 1 --warmup 0 --cpu-sample 0 > /tmp/pmc.log 2>&1
    db=$(find /tmp/prof_pmc -name "*.db" 2>/dev/null | head -1)
    if [ -n "$db" ]; then
      python tools/rocprof_summary.py $db --pmc | grep -v "^#\|^ calls" | grep "blind_rotate\|keyswitch\|^ *[0-9]" | grep -v "copyBuffer\|at::native"
    else echo "# pass failed: $grp"; tail -5 /tmp/pmc.log | sed 's/^/#   /'; fi
  done
} > gpurun_out/${tag}_pmc.txt
python - "$tag" "$PARAMS" "${DECOMP:-default}" <<'PY'
import json, re, sys
sys.path.insert(0, ".")
from iyokan_amd import hip
tag, params_name = sys.argv[1], sys.argv[2]
import os
levels = {"128bit": 3, "80bit": 2 if (sys.argv[3] == "direct" or os.environ.get("IYK_HIP_NTT", "fft") == "fft") else 4}[params_name]   # iyk_hip_decomposition_levels of the profiled run
kname = None
vals, durs = {}, []
for line in open(f"gpurun_out/{tag}_pmc.txt"):
    m = re.match(r"(\w+)\s+([\d.]+)\s+n=\d+\s+(.*)", line)
    if m and "blind_rotate" in m.group(3):
        vals[m.group(1)] = float(m.group(2))
        kname = m.group(3).strip().split("(")[0].replace("void ", "")
    m = re.match(r"\s*(\d+)\s+(\d+)\s+(\d+)\s+[\d.]+\s+(.*)", line)   # calls total_ns avg_ns pct kernel
    if m and "blind_rotate" in m.group(4):
        durs.append(float(m.group(3)))
if "FETCH_SIZE" in vals:
    out = {
        "_doc": "Counters of the dominant kernel from separate rocprofv3 --pmc passes (tools/profile_round.sh), one "
                "65536-NAND launch.  traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes: the factor 2 is the gfx950 "
                "FETCH_SIZE correction prescribed by MI355X_MICROARCH.md (HBM section).  valu_insts_per_launch = "
                "SQ_INSTS_VALU (wave-instructions, deterministic for a given kernel build and workload).  "
                "sustained_clock_ghz = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration of that pass.  bench.py reads this "
                "file for roofline.traffic and roofline.valu when the workload AND build_id (tools/src_hash.py: hash of "
                "the kernel sources the profiled library was built from) match the loaded library.",
        "kernel": kname,
        "build_id": hip.build_id(),
        "workload": {"gates_per_launch": 65536, "params": params_name, "op": "NAND", "decomposition_levels": levels},
        "FETCH_SIZE_KiB": vals["FETCH_SIZE"], "WRITE_SIZE_KiB": vals.get("WRITE_SIZE", 0.0),
        "traffic_bytes_per_launch": int((2 * vals["FETCH_SIZE"] + vals.get("WRITE_SIZE", 0.0)) * 1024),
    }
    if "TCC_HIT_sum" in vals and "TCC_MISS_sum" in vals:
        out["l2_hit_rate"] = round(vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"]), 4)
    for k_src, k_dst in (("SQ_INSTS_VALU", "valu_insts_per_launch"), ("SQ_INSTS_LDS", "lds_insts_per_launch"),
                         ("SQ_INSTS_SALU", "salu_insts_per_launch"), ("SQ_INSTS_VMEM_RD", "vmem_rd_insts_per_launch"),
                         ("SQ_INSTS", "all_insts_per_launch"), ("SQ_INSTS_SMEM", "smem_insts_per_launch"),
                         ("SQ_INSTS_BRANCH", "branch_insts_per_launch"), ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VALU"),
                         ("SQ_WAIT_ANY", "SQ_WAIT_ANY"), ("SQ_WAIT_INST_ANY", "SQ_WAIT_INST_ANY"),
                         ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_ANY"), ("SQ_WAVE_CYCLES", "SQ_WAVE_CYCLES"),
                         ("GRBM_GUI_ACTIVE", "GRBM_GUI_ACTIVE"), ("SQ_BUSY_CYCLES", "SQ_BUSY_CYCLES"),
                         ("SQ_WAVES", "SQ_WAVES"), ("TCP_TOTAL_CACHE_ACCESSES_sum", "l1_requests_per_launch"),
                         ("TCP_TCC_READ_REQ_sum", "l1_to_l2_requests_per_launch"),
                         ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_BANK_CONFLICT"), ("SQ_LDS_IDX_ACTIVE", "SQ_LDS_IDX_ACTIVE")):
        if k_src in vals:
            out[k_dst] = vals[k_src]
    if durs:
        out["pmc_pass_avg_launch_ns"] = sum(durs) / len(durs)
        if "GRBM_GUI_ACTIVE" in vals:
            clk = vals["GRBM_GUI_ACTIVE"] / 8.0 / (sum(durs) / len(durs))
            if 0.5 < clk < 3.0:
                out["sustained_clock_ghz"] = round(clk, 3)
    try:  # micro-benchmark ceilings (tools/ubench/valu_occ.hip) are measured separately: carry them over
        prev = json.load(open("profiles/r02_counters.json"))  # the stream ceilings do not depend on the kernel build
        if "issue_ceiling" in prev:
            out["issue_ceiling"] = prev["issue_ceiling"]
    except (OSError, ValueError):
        pass
    json.dump(out, open(f"gpurun_out/{tag}_counters.json", "w"), indent=1)
    print(out)
PY
cat gpurun_out/${tag}_kernel_trace.txt | head -12
