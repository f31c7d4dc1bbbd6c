"""The step-path fuzz of tests/test_gpu_step.py (random maps, vehicle parameters, fov, beam counts, agents, integrators
against independent oracle envs) over many more seeds than the suite runs:
    python tools/extended_fuzz.py [first_seed] [last_seed]
Prints the seeds that fail (none expected)."""
import os
import sys
import traceback

ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
lo = int(sys.argv[1]) if len(sys.argv) > 1 else 32
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 400
import test_gpu_step as t  # noqa: E402
assets = os.path.join(ROOT, 'red_gym_amd', 'assets')
bad = []
for seed in range(lo, hi):
    try:
        t.test_step_path_fuzz_vs_oracle(assets, seed)
    except Exception:  # a failing seed is the finding: report it and go on
        bad.append(seed)
        print('seed', seed, 'FAILED'); traceback.print_exc(limit=2)
    if seed % 50 == 0:
        print('... seed', seed, 'failures so far:', bad, flush=True)
print('seeds %d..%d: %d failures %s' % (lo, hi - 1, len(bad), bad))
