cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/emu; mkdir -p $O
for n in 8 4 1; do
for c in 1 0; do
  SIGAX_FIND_COOP=$c timeout -k 10 300 python3 bench.py --cpu-sample 0 --steps 60 --warmup 3 --isolated --emulate-world $n > $O/x$n$c.json 2> $O/x$n$c.err || exit 1
  python3 - $O/x$n$c.json $n $c <<EOT
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
print('world', sys.argv[2], 'coop', sys.argv[3], '%.2f Mreads/s step %.2f ms' % (d['value'] / 1e6, d['ms_per_step']), 'iso find %.2f' % d['roofline']['isolated']['kernel_ms_per_step']['k_find'])
EOT
done
done
