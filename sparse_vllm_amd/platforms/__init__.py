from .rocm import RocmPlatform, current_platform  # noqa: F401
