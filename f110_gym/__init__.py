"""Drop-in name of the reference package: `gym.make('f110_gym:f110-v0', **kw)` and
`from f110_gym.envs.base_classes import Integrator` resolve to the MI355X-native
implementation in red_gym_amd (reference: gym/f110_gym/__init__.py:1-5)."""
try:
    from gym.envs.registration import register
    register(id='f110-v0', entry_point='f110_gym.envs:F110Env')
except ImportError:  # gym absent: red_gym_amd.compat.install_missing() provides a minimal registry
    pass
