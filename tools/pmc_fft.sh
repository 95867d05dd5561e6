# counters of the FFT pass kernel (two SQ passes), A4 and 4096^2; usage on the GPU box: bash tools/pmc_fft.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
for which in a4 c5; do
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/r02/pmc_fft_${which}_1 -- python3 tools/bench_fft.py 8 1 $which > gpurun_out/r02/pmc_fft_${which}_1.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/r02/pmc_fft_${which}_2 -- python3 tools/bench_fft.py 8 1 $which > gpurun_out/r02/pmc_fft_${which}_2.log 2>&1
done
python3 - <<'PY'
import sys
sys.path.insert(0,'omr-img-corrector_amd')
from oics import pmc
for which in ('a4','c5'):
    for ps in (1,2):
        root='gpurun_out/r02/pmc_fft_%s_%d'%(which,ps)
        c=pmc.read_counters(root); d=pmc.read_durations(root)
        k=pmc.pick(c.keys(),'fft_pass_kernel')
        print(which,ps,k,'dispatch us',['%.0f'%x for x in d[k]],{n.replace('SQ_',''):['%.4g'%x for x in v] for n,v in c[k].items()})
PY
