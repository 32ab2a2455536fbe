"""lungmask_amd -- MI355X-native engine behind the lungmask API (lungmask/__init__.py:1)."""
from .mask import LMInferer  # noqa: F401
