for i in 1 2; do
  python bench.py --no-extras --no-cpu-baseline --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
  GPB_DEBUG_LIB=1 python bench.py --no-extras --no-cpu-baseline --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hint ', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done
