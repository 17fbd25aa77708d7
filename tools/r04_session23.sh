#!/bin/bash
# session 23: 8-wide tree (TEXIR_BVH8=1, experiment) under texir_trace_shade: parity against the brute-force tests, then the specular rays' bare trace 4-wide vs 8-wide
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_s23
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
TEXIR_BVH8=1 timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "trace or query or watertight or misses or closest" > $out/pytest_bvh8.txt 2>&1
tail -n 4 $out/pytest_bvh8.txt | cut -c1-200
for b in 0 1 0 1; do
  echo "== TEXIR_BVH8=$b"
  TEXIR_BVH8=$b timeout 300 python tools/spec_split_probe.py 2>&1 | grep -E "^roughness|^fused" | cut -c1-330
done > $out/spec_bvh8_probe.txt 2>&1
cat $out/spec_bvh8_probe.txt
