A=cudasw4_amd/lib/align
$A --query tests/golden/allqueries.fasta --pseudodb 1000000 512 --top 0 --verbose --uploadFull --prefetchDBFile --mat blosum62 --singlePassType Half2 --manyPassType_small Half2 --manyPassType_large Float --overflowType Float > gpurun_out/align_peak2.txt 2>&1
grep "Scan time" gpurun_out/align_peak2.txt | head -3; tail -n 1 gpurun_out/align_peak2.txt
$A --query tests/golden/allqueries.fasta --pseudodb 1000000 512 --top 10 --verbose --uploadFull --dpx > gpurun_out/align_peak3.txt 2>&1
tail -n 1 gpurun_out/align_peak3.txt
timeout 600 python -m pytest tests/test_gpu_driver.py -m gpu -x -q 2>&1 | tail -3
