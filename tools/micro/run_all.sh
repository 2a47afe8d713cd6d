# diagnostic harnesses of the contraction kernels -> gpurun_out/micro/<round>_*.txt (copy the ones to keep into profiles/)
set -e
R=${1:-r04}; O=gpurun_out/micro; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPD_STAMP -w -Iprompt-diffusion_amd/csrc tools/micro/ring_stamp.hip -o /tmp/ring_stamp && timeout -k 10 120 /tmp/ring_stamp > $O/${R}_ring_stamp.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPD_STAMP -w -Iprompt-diffusion_amd/csrc tools/micro/conv_w4_stamp.hip -o /tmp/conv_w4_stamp && timeout -k 10 120 /tmp/conv_w4_stamp > $O/${R}_conv_w4_stamp.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/micro/mfma_issue.hip -o /tmp/mfma_issue && timeout -k 10 60 /tmp/mfma_issue > $O/${R}_mfma_issue.txt 2>&1
bash tools/micro/conv_cold.sh > $O/${R}_conv_cold.txt 2>&1
echo done
