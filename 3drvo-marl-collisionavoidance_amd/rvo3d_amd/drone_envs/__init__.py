"""Counterpart of the reference package `uaisa_env.drone_envs` (façade classes)."""
from .mdin import mdin  # noqa: F401  (uaisa_env/drone_envs/__init__.py:1)
