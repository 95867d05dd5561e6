# counters of the scan-lane sweep kernel (one launch of 512 A4 scans per pass); usage on the GPU box: bash tools/pmc_lanes.sh [outdir] [passes: all | quick]
# (separate --pmc passes, --kernel-trace only, the program directly after --: the pool's rules for rocprofv3)
OUT=${1:-gpurun_out/pmc_lanes}
WHAT=${2:-all}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
run() { n=$1; shift; rm -rf "$OUT/$n"; timeout -k 10 280 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$n" -- python3 tools/klanes.py 512 512 2 > "$OUT/$n.log" 2>&1; echo "$n rc=$?" >> "$OUT/STATUS"; python3 tools/pmc_summary.py "$OUT/$n" slane_kernel >> "$OUT/summary.md" 2>&1; }
rm -f "$OUT/STATUS" "$OUT/summary.md"
run p2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES
run p3 SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_REQ SQC_TC_DATA_READ_REQ SQC_DCACHE_MISSES_DUPLICATE SQ_INSTS_SMEM SQ_BUSY_CU_CYCLES
if [ "$WHAT" = all ]; then
run p1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES
run p4 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum TCC_ATOMIC_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
run p5 FETCH_SIZE
run p6 WRITE_SIZE
fi
cat "$OUT/STATUS"
