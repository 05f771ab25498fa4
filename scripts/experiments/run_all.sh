# full verification + the profiles the traffic manifest is built from (one gpurun call)
bash scripts/experiments/run_final.sh
rm -rf gpurun_out/prof_r02_*
bash scripts/experiments/run_prof.sh 2>&1 | tail -30
