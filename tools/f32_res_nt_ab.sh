for v in 0 1 0 1; do
  if [ $v = 1 ]; then export HPE_EXTRA_FLAGS="-DHPE_F32_RES_NT"; else unset HPE_EXTRA_FLAGS; fi
  python -c "from hpe_amd import build; build.build()" || exit 1
  python bench.py --steps 30 --warmup 5 --cpu-sample 0 --sustain 0 --no-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RES_NT=$v', d['value'], d['ms_per_step'], d['roofline']['serial']['sum_of_53_launch_ms'])"
done
unset HPE_EXTRA_FLAGS; python -c "from hpe_amd import build; build.build()"
