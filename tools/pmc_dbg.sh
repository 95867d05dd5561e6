# counters of the sweep kernel with single phases switched off (debug library); see tools/kdbg.py
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
for dbg in 0 1 2 4; do
  export OMR_RUNS_DBG=$dbg
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/r02/pmc_dbg$dbg -- python3 tools/kdbg.py 4 > gpurun_out/r02/pmc_dbg$dbg.log 2>&1
  unset OMR_RUNS_DBG
done
for dbg in 0 1 2 4 6 7; do OMR_RUNS_DBG=$dbg python3 tools/kdbg.py 6 2>/dev/null | grep sweep; done
python3 - <<'PY'
import sys,os
sys.path.insert(0,'omr-img-corrector_amd')
from oics import pmc
for dbg in (0,1,2,4):
    c=pmc.read_counters('gpurun_out/r02/pmc_dbg%d'%dbg); d=pmc.read_durations('gpurun_out/r02/pmc_dbg%d'%dbg)
    k=pmc.pick(c.keys(),'runs_kernel')
    print('dbg',dbg,k,'us %.1f'%pmc.mean(d[k][1:]),{n.replace('SQ_',''):'%.4g'%pmc.mean(v[1:]) for n,v in c[k].items()})
PY
