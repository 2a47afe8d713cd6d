# A/B of HIP runtime environment variables on the headline bench and on batch 1: bash tools/ab_env.sh (on the GPU box; writes gpurun_out/abenv_*.log)
B="python bench.py --no-cpu-baseline --no-f32 --no-profile --no-sd3 --no-parity --steps 2"
run() { name=$1; shift; env "$@" timeout -k 10 300 $B $EXTRA > gpurun_out/abenv_$name.log 2>&1 || { echo "$name FAILED"; tail -3 gpurun_out/abenv_$name.log; return; }; python - $name <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/abenv_{sys.argv[1]}.log").read().strip().split("\n")[-1])
print(sys.argv[1], round(d["value"],3), "img/s", round(d["ms_per_step"],1), "ms")
PY
}
EXTRA=""
run base X=1
run kernarg1 HIP_FORCE_DEV_KERNARG=1
run kernarg0 HIP_FORCE_DEV_KERNARG=0
run base2 X=1
run kernarg1b HIP_FORCE_DEV_KERNARG=1
EXTRA="--batch 1 --steps 3"
run bs1_base X=1
run bs1_kernarg1 HIP_FORCE_DEV_KERNARG=1
run bs1_kernarg0 HIP_FORCE_DEV_KERNARG=0
run bs1_noint HSA_ENABLE_INTERRUPT=0
