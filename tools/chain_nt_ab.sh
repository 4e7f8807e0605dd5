#!/bin/bash
# A/B: chain kernel with / without the non-temporal policy on the residual DMA and the t3 stores (compile-time: -DHPE_CHAIN_NO_NT restores the plain policy)
OUT=gpurun_out/nt; mkdir -p $OUT
for v in 0 1 0 1; do
  if [ $v = 0 ]; then export HPE_EXTRA_FLAGS="-DHPE_CHAIN_NO_NT"; else unset HPE_EXTRA_FLAGS; fi
  python -c "from hpe_amd import build; build.build()" || exit 1
  python tools/layer_times.py 256 5 --all -- on=dtype:bf16 2>/dev/null | grep -E "res2[bc]_branch2[bc]|res3[bc]_branch2[bc]|all conv" | tr '\n' ';' ; echo " <- NT=$v"
  python bench.py --encoder-dtype bf16 --steps 20 --warmup 5 --cpu-sample 0 --sustain 0 --no-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('NT=$v', d['value'], d['ms_per_step'], d['roofline']['serial']['sum_of_53_launch_ms'])"
done
unset HPE_EXTRA_FLAGS
python -c "from hpe_amd import build; build.build()"
