cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3
rm -rf gpurun_out/pmc_c
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -d gpurun_out/pmc_c -o d --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/r3/pmc_c.log 2>&1 || exit 2
python3 tools/pmc_summary.py $(find gpurun_out/pmc_c -name 'd_counter_collection.csv' | head -1) k_igemm > gpurun_out/r3/round3_gemm_sq_counters.txt
python3 tools/pmc_summary.py $(find gpurun_out/pmc_c -name 'd_counter_collection.csv' | head -1) k_wgrad >> gpurun_out/r3/round3_gemm_sq_counters.txt
python3 tools/pmc_summary.py $(find gpurun_out/pmc_c -name 'd_counter_collection.csv' | head -1) k_stem >> gpurun_out/r3/round3_gemm_sq_counters.txt
rm -rf gpurun_out/pmc_c
python3 - <<'PY'
import re
cur=None; d={}
for l in open('gpurun_out/r3/round3_gemm_sq_counters.txt'):
    if not l.startswith(' '): cur=l.split('  vgpr')[0]; d[cur]={}; continue
    p=l.split(); d[cur][p[0]]=float(p[1])
for k,v in d.items():
    m=v.get('SQ_INSTS_MFMA',0)
    if m>0: print(f"{k[:70]:70s} VALU-other/MFMA {(v['SQ_INSTS_VALU']-m)/m:5.2f}  LDS/MFMA {v['SQ_INSTS_LDS']/m:5.2f}  VMEM/MFMA {v.get('SQ_INSTS_VMEM',0)/m:5.2f}  SALU/MFMA {v['SQ_INSTS_SALU']/m:5.2f}  wait {v['SQ_WAIT_ANY']/v['SQ_WAVE_CYCLES']:.2f} stall {v['SQ_WAIT_INST_ANY']/v['SQ_WAVE_CYCLES']:.2f} us {v['~duration_us']:.0f}")
PY
