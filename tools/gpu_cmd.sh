set -x
bash tools/collect_profiles.sh r02 1cd077a > gpurun_out/collect_r02.log 2>&1
tail -3 gpurun_out/collect_r02.log
cat gpurun_out/profiles_r02/kernel_counters.json
