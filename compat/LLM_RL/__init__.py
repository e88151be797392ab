"""Import-path compatibility layer: the reference's module names re-exported from lmrl_gym_amd (see compat/README.md)."""
