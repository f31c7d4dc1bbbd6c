"""Drop-in for the reference's `weap_util` package (weap_util/weap_util/__init__.py): the names resolve to the
MI355X implementation in red_gym_amd."""
