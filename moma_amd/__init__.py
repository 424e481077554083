from .hip_env import configure as _configure_hip_env

_configure_hip_env()          # before the first HIP call of the process (see hip_env.py)
