"""Same logger name / format as the reference (lungmask/logger.py:4-13)."""
import logging
import sys

logger = logging.getLogger("lungmask")
if not logger.handlers:
    logger.setLevel(logging.INFO)
    _h = logging.StreamHandler(sys.stdout)
    _h.setFormatter(logging.Formatter("lungmask %(asctime)s %(message)s", datefmt="%Y-%m-%d %H:%M:%S"))
    logger.addHandler(_h)
    logger.propagate = False  # lungmask/logger.py:6
