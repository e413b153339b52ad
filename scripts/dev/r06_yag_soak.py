"""Development aid (GPU box), round 6: tests/yag_soak.py for many seeds.    python3 scripts/dev/r06_yag_soak.py [first seed] [seeds] [calls]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import yag_soak
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 12
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 60
bad, tot = 0, {}
for s in range(first, first + seeds):
    b, t = yag_soak.run(s, calls)
    bad += b
    for k, v in t.items():
        tot[k] = tot.get(k, 0) + v
print("yagpy soak: %d seeds x %d calls, %d differed; %s" % (seeds, calls, bad, tot))
sys.exit(1 if bad else 0)
