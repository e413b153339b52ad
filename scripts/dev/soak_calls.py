"""Development aid (GPU box), round 5: tests/soak.py for many seeds and long sequences (the test suite runs two short ones).
    python3 scripts/dev/soak_calls.py [seed] [calls]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import soak
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ncalls = int(sys.argv[2]) if len(sys.argv) > 2 else 60
sys.exit(1 if soak.run(seed, ncalls) else 0)
