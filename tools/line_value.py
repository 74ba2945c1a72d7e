"""stdin: bench.py output -> 'workload value ms_per_step' of its JSON line (A/B loops of tools/gpu_session.sh run steps)."""
import json
import sys

z = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(z['config']['workload'][:60], z['value'], z['unit'], z['ms_per_step'], 'ms per call', flush=True)
