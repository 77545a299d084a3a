#!/bin/bash
# clips/s of the default fp32 configuration against the batch per GPU (1 timed pass each)
for b in ${@:-1 2 4 8 16}; do
  python bench.py --batch $b --steps 1 --warmup 1 --no-cpu-baseline --profile-ddim-steps 2 2>/dev/null | tail -1 > /tmp/_b.json
  python - $b <<'PY'
import json, sys
d = json.load(open("/tmp/_b.json"))
print("B=%s: %.4f clips/s, %.2f s per pass" % (sys.argv[1], d["value"], d["ms_per_step"] / 1e3))
PY
done
