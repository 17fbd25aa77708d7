#!/bin/bash
# round-4 GPU session 9: the pixel-sharded material step (1 and 2 ranks) and the 2-rank bench line; specular forward at 4 (shipped) / 5 / 6 / 8 waves per SIMD
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_s9
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
timeout 1500 python -m pytest tests/test_gpu_scale.py -m gpu -q -x -k "sharded or two_ranks or rccl" 2>&1 | tail -25 | tee $out/pytest.txt
for rep in 1 2; do
for lib in libtexir_hip build_ab/libtexir_specw5 build_ab/libtexir_specw6 build_ab/libtexir_specw8; do
  L=$R/texir_code_amd/$lib.so; [ -f $L ] || L=$R/$lib.so
  v=$(TEXIR_HIP_LIB=$L timeout 400 python bench.py --no-cpu --steps 1 --warmup 0 --extra none 2>>$out/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['material_step']['ms'])" 2>&1 | tail -1)
  echo "mat $lib $v" | tee -a $out/ab_spec_waves.txt
done
done
