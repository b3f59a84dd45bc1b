bash scripts/r04_ks.sh $1;  timeout 300 python scripts/r04_ks_timing.py build/libcffm_kst.so 2>&1 | tail -12 | tee -a gpurun_out/r04_ks_$1.txt
