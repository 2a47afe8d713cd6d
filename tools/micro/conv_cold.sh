# conv_patch.hip in isolation: warm / cold inputs, random / constant data; with tools/micro/conv_patch_old.hip present also that version (A/B)
set -e
V=("" "-DCOLD" "-DCONST_DATA"); [ -f tools/micro/conv_patch_old.hip ] && V+=("-DCONV_SRC_OLD" "-DCOLD -DCONV_SRC_OLD")
for v in "${V[@]}"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -w $v -Iprompt-diffusion_amd/csrc tools/micro/conv_stamp.hip -o /tmp/conv_c
  echo "== build: ${v:-in-tree, warm, random data}"; timeout -k 10 100 /tmp/conv_c
done
