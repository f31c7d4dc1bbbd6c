"""The bounds-checked debug build of the library (-DF110_BOUNDS, SURVEY 5 "race detection / sanitizers"; GPU AddressSanitizer
is not available on this pool).  The suite is run against it once per round:
    tools/build_variant.sh bounds -DF110_BOUNDS
    F110_LIB=build_variants/bounds.so F110_CHECK_DEVICE_ERRORS=1 python -m pytest tests -m gpu -q
(every Engine.close() then fails if the handle's device error word is not clean).  With the product build the test below
checks what it can: the word exists, is clean after a rollout, and reports-and-clears."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_device_error_word_and_bounds_selftest(assets):
    import torch
    from red_gym_amd import F110VecEnv, _lib, workload
    env = F110VecEnv(64, map=os.path.join(assets, 'example_map'), num_agents=2, autoreset=True)
    env.reset(workload.spawn_poses(64, 2))
    acts = torch.as_tensor(workload.action_pool(4, 64, 2), device='cuda')
    for k in range(30):
        env.step(acts[k % 4])
    assert env.eng.device_errors() == 0
    lib = _lib.load()
    if hasattr(lib, 'f110_bounds_selftest'):           # the bounds-checked build: a deliberate violation must be reported
        lib.f110_bounds_selftest.argtypes = [_lib._VP]
        assert lib.f110_bounds_selftest(env.eng._h) == 0
        flags = env.eng.device_errors()
        assert flags & 0x2 and flags >> 8, hex(flags)  # F110_DEVERR_BOUNDS and the self-test's table bit
        assert env.eng.device_errors() == 0
    env.close()
