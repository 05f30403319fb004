"""Stand-in for torchnet: only meter.ClassErrorMeter is used by the reference
(packnet/main.py:152,184)."""
from . import meter  # noqa: F401
