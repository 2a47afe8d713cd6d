# knock-out builds of the ring GEMM (timing only): bash tools/micro/ring_ko.sh
set -e
for v in "" "-DPD_KO_COALESCED" "-DPD_KO_DMA" "-DPD_KO_DSREAD" "-DPD_KO_DMA -DPD_KO_DSREAD"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -w $v -Iprompt-diffusion_amd/csrc tools/micro/ring_stamp.hip -o /tmp/ring_ko
  echo "== build: ${v:-product}"; timeout -k 10 60 /tmp/ring_ko 2>&1 | tee /tmp/ring_ko.out; if grep -q "Memory access fault" /tmp/ring_ko.out; then exit 1; fi
done
