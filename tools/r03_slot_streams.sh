export TMPDIR=/tmp
python tools/gen_cache.py --pairs 64 > /dev/null 2>&1
for rep in 1 2; do for ss in 1 0; do
  timeout -k 10 400 python bench.py --cpu-seconds 0 --pairs-cache /tmp/pcm_pairs.npz --steps 50 --warmup 3 --slot-streams $ss 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('slot-streams $ss rep $rep:', round(d['value']), round(d['ms_per_step'],3), round(r['avg_launch_ms_two_slots'],4), d['config'].get('parity_ok'))"
done; done
