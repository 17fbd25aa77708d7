#!/bin/bash
# round-2 GPU session 12: node fetch as 8-byte loads (TEXIR_NODE_LD2 = 1: nodes, 2: nodes + triangles) A/B; tcp_node counters again
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r02_s12
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
ab() { label=$1; shift
  for W in "${WLS[@]}"; do
    v=$(env "$@" timeout 600 python bench.py --workload $W --steps 2 --warmup 1 --no-cpu --no-mat 2>>$out/ab.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel'])" 2>&1 | tail -1)
    echo "$label $W $v" | tee -a $out/ab.txt
  done
}
WLS=(c4 c2 c4_scan)
ab default X=1
ab ld2 TEXIR_HIP_LIB=$R/build_ab/libtexir_hip_ld2.so
ab ld2t TEXIR_HIP_LIB=$R/build_ab/libtexir_hip_ld2t.so
TEXIR_HIP_LIB=$R/build_ab/libtexir_hip_ld2t.so timeout 1200 python -m pytest tests/test_gpu_watertight.py tests/test_gpu_parity.py -m gpu -q -x > $out/pytest_ld2t.txt 2>&1
tail -n 3 $out/pytest_ld2t.txt | cut -c1-200
bash tools/r02_session11.sh > $out/s11_again.txt 2>&1
cp $R/gpurun_out/r02_s11/tcp_node_pmc.txt $out/
cat $out/tcp_node_pmc.txt | cut -c1-160
