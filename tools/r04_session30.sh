#!/bin/bash
# round-4 GPU session 30: texture.py adds the two gradients of a twice-consumed fetch itself beside the per-level reference folds (TEXIR_MIP_PER_LEVEL=1): the tests that
# exercise it + the material step's PMC profile on the changed material-side sources (profiles/pmc_mat_step.json carries their hash)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04_s30
mkdir -p $out
cd $R
export TEXIR_SYNTH_CACHE=/tmp/texir_synth
timeout 1500 python -m pytest tests/test_gpu_scan_and_configs.py tests/test_gpu_tex_batch.py tests/test_gpu_mat_step_oracle.py -m gpu -q -k "4k or batch or oracle" 2>&1 | tail -5 | tee $out/pytest.txt
bash tools/mat_step_pmc.sh r04_s30/matpmc > $out/mat_pmc.log 2>&1
cp $R/profiles/pmc_mat_step.json $out/ 2>/dev/null
head -n 1 $out/mat_pmc.log | cut -c1-300
