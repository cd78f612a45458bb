"""One short bench.py run reduced to the numbers an A/B needs (stdin: the JSON line)."""
import json, sys
d = json.loads([ln for ln in sys.stdin.read().splitlines() if ln.startswith('{')][-1])
rf = d.get('roofline', {})
print(f"{d['value']:.2f} steps/s  {d['ms_per_step']:.4f} ms/step  {rf.get('kernel')} {rf.get('avg_launch_ms', 0):.4f} ms  frac {rf.get('frac', 0):.4f}  build {d.get('build')}")
