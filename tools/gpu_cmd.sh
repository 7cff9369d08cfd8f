for v in "" v2 v3 v4 v5 v6 v7 ""; do
  if [ -z "$v" ]; then L=cudasw4_amd/lib/libcudasw4_amd.so; else L=cudasw4_amd/lib_$v/libcudasw4_amd.so; fi
  echo "== ${v:-base}"
  CUDASW4_AMD_LIB=$L timeout 600 python tools/peak_sweep.py --kernels half2 --lengths 512 2>&1 | grep -v amdgpu.ids | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); pq=d['per_query_gcups']; print(d['gcups'], 'single(464,567):', pq[4], pq[5], 'multi(1500,5478):', pq[10], pq[19])"
done
