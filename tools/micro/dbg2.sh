mkdir -p gpurun_out/dbg2
timeout 600 python bench.py --workload uawarpc_align_512x512 --steps 50 --warmup 5 2>&1 | tail -3 > gpurun_out/dbg2/k2.txt
timeout 900 python -X faulthandler bench.py --precision fp32 --no-cpu --steps 4 --warmup 2 > gpurun_out/dbg2/fp32.txt 2>&1
echo "rc=$?" >> gpurun_out/dbg2/fp32.txt
RFN_HIP_GRAPH=0 timeout 900 python -X faulthandler bench.py --precision fp32 --no-cpu --steps 4 --warmup 2 > gpurun_out/dbg2/fp32_nograph.txt 2>&1
echo "rc=$?" >> gpurun_out/dbg2/fp32_nograph.txt
timeout 900 python -X faulthandler -c "
import sys; sys.argv=['bench.py','--precision','fp32','--no-cpu','--steps','4','--warmup','2']
from refign_amd import mfma; mfma.GROUP_WGRADS=False
import runpy; runpy.run_path('bench.py', run_name='__main__')" > gpurun_out/dbg2/fp32_nogroup.txt 2>&1
echo "rc=$?" >> gpurun_out/dbg2/fp32_nogroup.txt
