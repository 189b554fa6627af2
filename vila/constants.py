"""reference vila/constants.py (same values as llava/constants.py) + the HALVA mask tags."""
from llava.constants import *  # noqa: F401,F403
