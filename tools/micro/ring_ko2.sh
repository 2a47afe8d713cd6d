# Knock-outs of the ring GEMM's epilogue on temporary copies of the source (timing only; results are garbage): where do the 7 us between the last
# MFMA and the end of a short-reduction launch go?   bash tools/micro/ring_ko2.sh > gpurun_out/micro/ring_ko2.txt
set -e
for v in base nostore nobias noepi; do
  d=/tmp/ko_$v; rm -rf $d; mkdir -p $d/tools/micro $d/prompt-diffusion_amd
  cp -r prompt-diffusion_amd/csrc $d/prompt-diffusion_amd/; cp tools/micro/ring_stamp.hip $d/tools/micro/
  f=$d/prompt-diffusion_amd/csrc/gemm_ring.hip
  case $v in
    nostore|noepi) sed -i 's|^\( *\)store4(p.C, (size_t)gm \* p.ldc + gn, p.c_dt, v);|\1if (p.M < 0) store4(p.C, (size_t)gm * p.ldc + gn, p.c_dt, v);|' $f;;
  esac
  case $v in
    nobias|noepi) sed -i 's|e_bias\[n\] = p.bias ? \*reinterpret_cast<const f32x4\*>(p.bias + gn)|e_bias[n] = p.M < 0 ? *reinterpret_cast<const f32x4*>(p.bias + gn)|' $f;;
  esac
  if [ $v != base ] && cmp -s $f prompt-diffusion_amd/csrc/gemm_ring.hip; then echo "knock-out $v did not apply"; exit 1; fi
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -I$d/prompt-diffusion_amd/csrc $d/tools/micro/ring_stamp.hip -o /tmp/ring_$v
done
for round in 1 2; do for v in base nostore nobias noepi; do echo "== $v (round $round)"; timeout -k 10 60 /tmp/ring_$v | head -6; done; done
