bash tools/collect_profiles.sh r02 ed0a3ce > gpurun_out/collect_r02.log 2>&1
bash tools/collect_profiles.sh r02 ed0a3ce --workload sprot-like > gpurun_out/collect_r02s.log 2>&1
head -8 profiles/kernel_counters.json
bash tools/full_measurement.sh 2>&1 | tail -16
