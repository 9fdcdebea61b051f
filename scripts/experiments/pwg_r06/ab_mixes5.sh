bash scripts/ab_libs.sh "B:10,0 E:10,0 B:10,1 C:10,0" 2>&1 | tee gpurun_out/ab_mixes5.txt
for n in 32768 16384 8192; do PWG_N=$n bash scripts/ab_libs.sh "A:- B:10,0 D:10,2 C:9,4" 2>&1 | tee -a gpurun_out/ab_mixes5.txt; done
