export TMPDIR=/tmp
ONLY="nar out_proj,nar linear2" DLN=1 NOATTN=1 bash tools/pmc_traffic.sh r4k_pmc > gpurun_out/r4k/pmc.log 2>&1
cat gpurun_out/r4k_pmc/summary.txt | head -40
